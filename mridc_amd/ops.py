"""Thin tensor-level wrappers over the C ABI (include/mridc_amd.h) used by the model mirrors.

Each wrapper validates shapes, allocates the output with torch (torch owns device memory), and launches the HIP
kernels on the current stream.  No arithmetic happens in Python/torch here.
"""
import ctypes
import os

import torch

from mridc_amd import _lib
from mridc_amd._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, PAD_REPLICATE, PAD_ZERO  # noqa: F401


def _norm(normalization):
    key = str(normalization).lower()
    if key not in _lib.NORM:
        raise ValueError(f"Unknown fft normalization '{normalization}'")
    return _lib.NORM[key]


def _check_last_two(spatial_dims, ndim_complex):
    dims = [-2, -1] if spatial_dims is None else [int(d) for d in spatial_dims]
    dims = sorted(d % ndim_complex for d in dims)
    if dims != [ndim_complex - 2, ndim_complex - 1]:
        raise NotImplementedError("the fused MRI operators transform the last two spatial dims only "
                                  f"(got spatial_dims={list(spatial_dims)})")


def _bchw(t):
    if t.dim() != 5 or t.shape[-1] != 2:
        raise ValueError(f"expected a [B,C,H,W,2] tensor, got {tuple(t.shape)}")
    return int(t.shape[0]), int(t.shape[1]), int(t.shape[2]), int(t.shape[3])


PFA372 = True            # W = 372 prime-factor row kernels (module attributes like this one are test hooks, not environment switches)


# Packed / transformed copies of model WEIGHTS are cached per (storage, version), and a pack made eagerly is reused inside a hipGraph capture: right
# for inference graphs (the weights do not change between replays).  A captured TRAINING step replays on weights the optimizer has updated in between:
# with WEIGHTS_DYNAMIC the pack kernels are launched into the graph (cache entries made in a capture are keyed on it and never served from, or to, eager
# calls), so every replay packs the current weights.  Set by training.GraphedModelStep around its capture.
WEIGHTS_DYNAMIC = False


def _wcap():
    """Key component of the weight-pack caches: 0, or the id of the running capture when its weights are dynamic."""
    return int(_lib.lib().mrx_stream_capture_id(_lib.stream_ptr())) if WEIGHTS_DYNAMIC else 0


class _PreparedCache:
    """Per-slice prepared operands (lane-ordered maps, column-tiled k-space) keyed on the source tensor's (storage, version) AND on the
    hipGraph capture the current stream is in (mrx_stream_capture_id; 0 = eager):
      * an eager entry is never used while capturing -- the prepare kernel is launched INTO the graph, so a replay on refilled static
        inputs prepares them again instead of reading operands of the data the capture happened to see;
      * an entry made inside a capture lives in that graph's memory pool and is used only by later calls of the same capture (the 64 RIM
        steps of a slice share one prepare launch); it is never evicted while its capture is running, and dropped when another capture or
        an eager call shows that it is over (the graph keeps its own reference to the pool);
      * eager entries are least-recently-made-first evicted beyond `keep`; every entry holds its source tensor so the address cannot be
        recycled under the same version."""

    def __init__(self, keep=4, parameters=False):
        self.keep = keep
        self.entries = {}
        # parameters=True: the sources are model weights, not per-slice data -- a pack made eagerly (the warm-up call every capture needs anyway) is
        # also right inside a capture; replaying a graph after an in-place weight update is wrong with or without this cache
        self.parameters = parameters

    def get(self, src, extra, make):
        cap = int(_lib.lib().mrx_stream_capture_id(_lib.stream_ptr()))
        stale = [k for k in self.entries if k[0] != 0 and k[0] != cap]
        for k in stale:
            del self.entries[k]
        key = (cap, src.data_ptr(), src._version, str(src.device), tuple(src.shape)) + tuple(extra)
        hit = self.entries.get(key)
        if hit is None and cap != 0 and self.parameters and not WEIGHTS_DYNAMIC:
            hit = self.entries.get((0,) + key[1:])
        if hit is None:
            if cap == 0:
                eager = [k for k in self.entries if k[0] == 0]
                if len(eager) >= self.keep:
                    del self.entries[eager[0]]
            hit = self.entries[key] = (make(), src.detach())
        return hit[0]


_SP372 = _PreparedCache()


def _sp372(sens, centered):
    """The sensitivity maps in the lane order of the W = 372 prime-factor kernels (mrx_pfa372_prepare_maps): constant over the cascades of a
    slice, so prepared once per (storage, version, capture) -- see _PreparedCache."""
    def make():
        B, C, H, W = _bchw(sens)
        sp = torch.empty(int(_lib.lib().mrx_llg372_operand_floats(B, C, H)), dtype=torch.float32, device=sens.device)
        _lib.check(_lib.lib().mrx_pfa372_prepare_maps(_lib.ptr(sens), _lib.ptr(sp), B, C, H, int(bool(centered)), _lib.stream_ptr()),
                   "mrx_pfa372_prepare_maps")
        return sp
    return _SP372.get(sens, (bool(centered),), make)


def _pfa372_ok(sens):
    return PFA372 and sens.dim() == 5 and int(sens.shape[3]) == 372


def sens_expand(x, sens, centered, normalization, spatial_dims=None, hybrid=False, reduce=False):
    """fft2(complex_mul(x, S)).  x [B,H,W,2] or [B,1,H,W,2]; S [B,C,H,W,2] -> [B,C,H,W,2].  `hybrid`: the W transform only
    (k-space kept as IFFT_H(k) for row-invariant masks: mrx_sens_expand_rows).  `reduce` (hybrid, W = 372): also returns
    sum_c conj(S) IFFT_W(result) -- the next cascade's sens_reduce -- from the same pass (mrx_pfa372_expand_reduce)."""
    sens = _lib.f32c(sens)
    B, C, H, W = _bchw(sens)
    _check_last_two(spatial_dims, 4)
    x = _lib.f32c(x)
    if x.dim() == 5 and x.shape[1] == 1:
        x = x.reshape(B, H, W, 2)
    if tuple(x.shape) != (B, H, W, 2):
        raise ValueError(f"sens_expand: image shape {tuple(x.shape)} does not match maps {tuple(sens.shape)}")
    out = torch.empty_like(sens)
    if reduce:
        if not (hybrid and _pfa372_ok(sens)):
            raise NotImplementedError("sens_expand(reduce=True): hybrid space at W = 372 only")
        L = _lib.lib()
        red = torch.empty(B, H, W, 2, dtype=torch.float32, device=sens.device)
        wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=sens.device)
        _lib.check(L.mrx_pfa372_expand_reduce(_lib.ptr(x), _lib.ptr(_sp372(sens, centered)), _lib.ptr(out), None, None, None, 0, None, None,
                                              _lib.ptr(red), _lib.ptr(wk), B, C, H, _norm(normalization), int(bool(centered)),
                                              _lib.stream_ptr()), "mrx_pfa372_expand_reduce")
        return out, red
    if hybrid and _pfa372_ok(sens):                                  # W = 372: wave-private prime-factor row transforms
        _lib.check(_lib.lib().mrx_pfa372_expand(_lib.ptr(x), _lib.ptr(_sp372(sens, centered)), _lib.ptr(out), None, None, None, 0, None, None,
                                                B, C, H, _norm(normalization), int(bool(centered)), _lib.stream_ptr()), "mrx_pfa372_expand")
        return out
    if not hybrid and _pfa372_ok(sens):      # W = 372: the prime-factor row pass, then the column pass in place (fft2 = FFT_H after FFT_W)
        _lib.check(_lib.lib().mrx_pfa372_expand(_lib.ptr(x), _lib.ptr(_sp372(sens, centered)), _lib.ptr(out), None, None, None, 0, None, None,
                                                B, C, H, _norm(normalization), int(bool(centered)), _lib.stream_ptr()), "mrx_pfa372_expand")
        _lib.check(_lib.lib().mrx_fft_cols(_lib.ptr(out), _lib.ptr(out), B * C, H, W, 0, _norm(normalization), int(bool(centered)),
                                           _lib.stream_ptr()), "mrx_fft_cols")
        return out
    fn = _lib.lib().mrx_sens_expand_rows if hybrid else _lib.lib().mrx_sens_expand
    _lib.check(fn(_lib.ptr(x), _lib.ptr(sens), _lib.ptr(out), B, C, H, W, _norm(normalization), int(bool(centered)), _lib.stream_ptr()),
               "mrx_sens_expand")
    return out


def sens_expand_dc_reduce_supported(sens):
    """The W = 372 kernel can hand the next cascade its sens_reduce in the same pass (mrx_pfa372_expand_reduce)."""
    return _pfa372_ok(sens)


def sens_expand_dc_hybrid(x, sens, pred, ref, mask, dc_weight, centered, normalization, reduce=False):
    """pred - where(mask, pred - ref, 0) * dc_weight - FFT_W(x * S), everything in hybrid space: the cascade's last two steps
    (sens_expand, soft data consistency) as one pass (mrx_sens_expand_rows_dc).  `reduce` (W = 372 only): also returns
    sum_c conj(S) IFFT_W(result) [B,H,W,2] -- the sens_reduce of the NEXT cascade -- computed on the rows while they are on chip."""
    sens, pred, ref = _lib.f32c(sens), _lib.f32c(pred), _lib.f32c(ref)
    B, C, H, W = _bchw(sens)
    x = _lib.f32c(x)
    if x.dim() == 5 and x.shape[1] == 1:
        x = x.reshape(B, H, W, 2)
    if tuple(x.shape) != (B, H, W, 2) or pred.shape != sens.shape or ref.shape != sens.shape:
        raise ValueError("sens_expand_dc_hybrid: inconsistent shapes")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    w = _lib.f32c(dc_weight.detach().reshape(-1))
    out = torch.empty_like(sens)
    if reduce:
        if not _pfa372_ok(sens):
            raise NotImplementedError("sens_expand_dc_hybrid(reduce=True): W = 372 only")
        L = _lib.lib()
        red = torch.empty(B, H, W, 2, dtype=torch.float32, device=sens.device)
        wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=sens.device)
        _lib.check(L.mrx_pfa372_expand_reduce(_lib.ptr(x), _lib.ptr(_sp372(sens, centered)), _lib.ptr(out), _lib.ptr(pred), _lib.ptr(ref),
                                              _lib.ptr(m), kind, ms, _lib.ptr(w), _lib.ptr(red), _lib.ptr(wk), B, C, H, _norm(normalization),
                                              int(bool(centered)), _lib.stream_ptr()), "mrx_pfa372_expand_reduce")
        return out, red
    if _pfa372_ok(sens):
        _lib.check(_lib.lib().mrx_pfa372_expand(_lib.ptr(x), _lib.ptr(_sp372(sens, centered)), _lib.ptr(out), _lib.ptr(pred), _lib.ptr(ref),
                                                _lib.ptr(m), kind, ms, _lib.ptr(w), B, C, H, _norm(normalization), int(bool(centered)),
                                                _lib.stream_ptr()), "mrx_pfa372_expand")
        return out
    _lib.check(_lib.lib().mrx_sens_expand_rows_dc(_lib.ptr(x), _lib.ptr(sens), _lib.ptr(pred), _lib.ptr(ref), _lib.ptr(m), kind, ms,
                                                  _lib.ptr(w), _lib.ptr(out), B, C, H, W, _norm(normalization), int(bool(centered)),
                                                  _lib.stream_ptr()), "mrx_sens_expand_rows_dc")
    return out


def sens_reduce(k, sens, centered, normalization, spatial_dims=None, work=None, hybrid=False):
    """sum_c ifft2(k) * conj(S) -> [B,H,W,2].  `hybrid`: k is IFFT_H of the k-space already (mrx_sens_reduce_rows)."""
    k, sens = _lib.f32c(k), _lib.f32c(sens)
    B, C, H, W = _bchw(k)
    if sens.shape != k.shape:
        raise ValueError(f"sens_reduce: k-space {tuple(k.shape)} vs maps {tuple(sens.shape)}")
    _check_last_two(spatial_dims, 4)
    if _pfa372_ok(sens):                      # W = 372: the prime-factor row pass (after the column pass when k is not in hybrid space yet)
        L = _lib.lib()
        if not hybrid:
            if work is None:
                work = torch.empty_like(k)
            _lib.check(L.mrx_fft_cols(_lib.ptr(k), _lib.ptr(work), B * C, H, W, 1, _norm(normalization), int(bool(centered)),
                                      _lib.stream_ptr()), "mrx_fft_cols")
            k = work
        out = torch.empty(B, H, W, 2, dtype=torch.float32, device=k.device)
        wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=k.device)
        _lib.check(L.mrx_pfa372_reduce(_lib.ptr(k), _lib.ptr(_sp372(sens, centered)), None, _lib.ptr(out), None, _lib.ptr(wk), B, C, H, 1.0,
                                       _norm(normalization), int(bool(centered)), _lib.stream_ptr()), "mrx_pfa372_reduce")
        return out
    if hybrid:
        out = torch.empty(B, H, W, 2, dtype=torch.float32, device=k.device)
        _lib.check(_lib.lib().mrx_sens_reduce_rows(_lib.ptr(k), _lib.ptr(sens), _lib.ptr(out), B, C, H, W, _norm(normalization),
                                                   int(bool(centered)), _lib.stream_ptr()), "mrx_sens_reduce_rows")
        return out
    if work is None:
        work = torch.empty_like(k)
    out = torch.empty(B, H, W, 2, dtype=torch.float32, device=k.device)
    _lib.check(_lib.lib().mrx_sens_reduce(_lib.ptr(k), _lib.ptr(sens), _lib.ptr(out), _lib.ptr(work), B, C, H, W,
                                          _norm(normalization), int(bool(centered)), _lib.stream_ptr()), "mrx_sens_reduce")
    return out


LLG_T4 = True
_Y_T4 = _PreparedCache()


def llg_t4_supported(y):
    """General-mask gradient at W = 372 on the column-tiled coil stack (mrx_pfa372_expand_t4 / mrx_llg_cols_dc_t4 / mrx_pfa372_reduce_t4)."""
    return (LLG_T4 and PFA372 and y.dim() == 5 and int(y.shape[3]) == 372
            and bool(_lib.lib().mrx_llg_cols_dc_t4_supported(int(y.shape[2]), int(y.shape[3]))))


def _y_t4(y):
    """The measured k-space in the column-tiled layout [B*C][W/4][H][4] (mrx_tile4_cols), prepared once per (storage, version, capture)
    like _sp372: it is constant over the steps and cascades of a slice."""
    def make():
        B, C, H, W = _bchw(y)
        t4 = torch.empty_like(y)
        _lib.check(_lib.lib().mrx_tile4_cols(_lib.ptr(y), _lib.ptr(t4), B * C, H, W, _lib.stream_ptr()), "mrx_tile4_cols")
        return t4
    return _Y_T4.get(y, (), make)


LLG_T4_NO_Y = True          # parts form: the measured data enters as one constant partial plane instead of being read by every column pass (test hook)
_LLG_CONST = _PreparedCache()


def _llg_parts_buffer(y, sens, mask, centered, normalization, spatial_dims, n):
    """The partial planes of the general-mask gradient at W = 372, [n + 1][B,H,W,2], one buffer per slice (storage, version, capture): planes
    0 .. n - 1 are rewritten by every step (mrx_pfa372_reduce_t4), plane n holds the constant term -A^H M y = -sum_c conj(S) ifft2(mask y)
    (rim_utils.py:53-62 is linear in (k - y)), so the column pass of a step never reads y and the first RIM layer's loader -- which keeps
    four partial planes in flight and has three at 15 coils -- adds it for free.

    LIFETIME: `llg(..., parts=True)` returns views of this buffer; planes 0 .. n - 1 are valid until the next gradient call for the same slice ON THE
    SAME STREAM (the buffer is keyed on the stream as well, so two streams evaluating one slice do not overwrite each other's planes).  The layer-1
    loader consumes them before that call; a caller that keeps a step's planes clones them."""
    def make():
        B, C, H, W = _bchw(y)
        buf = torch.empty(n + 1, B, H, W, 2, dtype=torch.float32, device=y.device)
        buf[n] = sens_reduce(y * mask, sens, centered, normalization, spatial_dims)
        buf[n].neg_()
        return buf
    return _LLG_CONST.get(y, (sens.data_ptr(), sens._version, mask.data_ptr(), mask._version, bool(centered), str(normalization), int(n),
                              int(torch.cuda.current_stream().cuda_stream)), make)


# General masks at W = 372: the tap gather rides in the next step's first gradient pass (mrx_pfa372_expand_t4_gather).  Built, bit-identical, and OFF by default:
# the 2-D-mask line ran 110.0-110.7 slices/s with it against 111.2-111.7 without (tools/runs/r05p.sh, three alternating runs on one box) -- the three
# coil-group tasks of an image row each repeat the row's 114 tap loads, which costs what the saved 5-us launch bought.
LLG_T4_GATHER = os.environ.get("MRIDC_AMD_LLG_T4_GATHER", "0") == "1"


def llg(eta, y, sens, mask, sigma, centered, normalization, spatial_dims=None, out=None, work=None, parts=False, gather=None):
    """log_likelihood_gradient -> [B,4,H,W].  `parts` (llg_t4_supported only): returns (part [n][B,H,W,2], n) -- the coil-group partial sums
    of the last pass, for rim_layer_indrnn_packed_llg, which adds them, scales by 1/sigma^2 and splits the channels in its tile loader.
    `gather` = (taps [B,18,H,W], b_final | None) with parts (the constant-plane form): eta is first replaced by eta + the nine-tap gather of the
    previous step's tap products inside the first pass (mrx_pfa372_expand_t4_gather, bit-identical to rim_final_gather); returns (part, n, eta_new)."""
    y, sens, eta = _lib.f32c(y), _lib.f32c(sens), _lib.f32c(eta)
    B, C, H, W = _bchw(y)
    if sens.shape != y.shape:
        raise ValueError(f"log_likelihood_gradient: k-space {tuple(y.shape)} vs maps {tuple(sens.shape)}")
    if tuple(eta.shape) != (B, H, W, 2):
        raise ValueError(f"log_likelihood_gradient: eta {tuple(eta.shape)} does not match {(B, H, W, 2)}")
    _check_last_two(spatial_dims, 4)
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    if out is None and not parts:
        out = torch.empty(B, 4, H, W, dtype=torch.float32, device=y.device)
    if work is None:
        work = torch.empty_like(y)
    if parts and not (_pfa372_ok(sens) and llg_t4_supported(y)):
        raise ValueError("llg(parts=True) needs the column-tiled W = 372 path (llg_t4_supported)")
    if _pfa372_ok(sens):
        # W = 372: the two row passes on the prime-factor kernels (maps in lane order, cached per slice), the column pass + DC between them
        L = _lib.lib()
        sp = _sp372(sens, centered)
        nrm, cen, st = _norm(normalization), int(bool(centered)), _lib.stream_ptr()
        if llg_t4_supported(y):
            # the coil stack between the passes column-tiled (contiguous 4-column blocks for the column pass); y tiled once per slice
            eta_new = None
            if gather is not None:
                if not (parts and LLG_T4_NO_Y):
                    raise ValueError("llg(gather=...) needs the constant-plane parts form")
                taps, bfin = gather
                eta_new = torch.empty_like(eta)
                _lib.check(L.mrx_pfa372_expand_t4_gather(_lib.ptr(eta), _lib.ptr(_lib.f32c(taps)), _lib.ptr(_lib.f32c(bfin.detach()) if bfin is not None else None),
                                                         _lib.ptr(eta_new), _lib.ptr(sp), _lib.ptr(work), B, C, H, nrm, cen, st), "mrx_pfa372_expand_t4_gather")
            else:
                _lib.check(L.mrx_pfa372_expand_t4(_lib.ptr(eta), _lib.ptr(sp), _lib.ptr(work), B, C, H, nrm, cen, st), "mrx_pfa372_expand_t4")
            if parts and LLG_T4_NO_Y:
                n0 = int(L.mrx_llg372_work_floats(B, C, H)) // (B * H * W * 2)           # coil-group partial planes of the last pass
                wk = _llg_parts_buffer(y, sens, mask, centered, normalization, spatial_dims, n0)
                _lib.check(L.mrx_llg_cols_dc_t4(_lib.ptr(work), None, _lib.ptr(m), kind, ms, B, C, H, W, nrm, cen, st), "mrx_llg_cols_dc_t4")
                n = ctypes.c_int(0)
                _lib.check(L.mrx_pfa372_reduce_t4(_lib.ptr(work), _lib.ptr(sp), None, None, _lib.ptr(wk), ctypes.byref(n), B, C, H,
                                                  float(1.0 / (float(sigma) ** 2.0)), nrm, cen, st), "mrx_pfa372_reduce_t4")
                if int(n.value) != n0:
                    raise RuntimeError(f"llg: {n.value} partial planes, expected {n0}")
                return (wk, n0 + 1, eta_new) if gather is not None else (wk, n0 + 1)
            yt4 = _y_t4(y)
            wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=y.device)
            _lib.check(L.mrx_llg_cols_dc_t4(_lib.ptr(work), _lib.ptr(yt4), _lib.ptr(m), kind, ms, B, C, H, W, nrm, cen, st), "mrx_llg_cols_dc_t4")
            if parts:
                n = ctypes.c_int(0)
                _lib.check(L.mrx_pfa372_reduce_t4(_lib.ptr(work), _lib.ptr(sp), None, None, _lib.ptr(wk), ctypes.byref(n), B, C, H,
                                                  float(1.0 / (float(sigma) ** 2.0)), nrm, cen, st), "mrx_pfa372_reduce_t4")
                return wk, int(n.value)
            _lib.check(L.mrx_pfa372_reduce_t4(_lib.ptr(work), _lib.ptr(sp), _lib.ptr(eta), _lib.ptr(out), _lib.ptr(wk), None, B, C, H,
                                              float(1.0 / (float(sigma) ** 2.0)), nrm, cen, st), "mrx_pfa372_reduce_t4")
            return out
        _lib.check(L.mrx_pfa372_expand(_lib.ptr(eta), _lib.ptr(sp), _lib.ptr(work), None, None, None, 0, None, None, B, C, H, nrm, cen, st),
                   "mrx_pfa372_expand")
        _lib.check(L.mrx_llg_cols_dc(_lib.ptr(work), _lib.ptr(y), _lib.ptr(m), kind, ms, B, C, H, W, nrm, cen, st), "mrx_llg_cols_dc")
        wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=y.device)
        _lib.check(L.mrx_pfa372_reduce(_lib.ptr(work), _lib.ptr(sp), _lib.ptr(eta), None, _lib.ptr(out), _lib.ptr(wk), B, C, H,
                                       float(1.0 / (float(sigma) ** 2.0)), nrm, cen, st), "mrx_pfa372_reduce")
        return out
    _lib.check(_lib.lib().mrx_llg(_lib.ptr(eta), _lib.ptr(y), _lib.ptr(sens), _lib.ptr(m), kind, ms, _lib.ptr(out),
                                  _lib.ptr(work), B, C, H, W, float(1.0 / (float(sigma) ** 2.0)), _norm(normalization),
                                  int(bool(centered)), _lib.stream_ptr()), "mrx_llg")
    return out


_ROW_INV = {}


def mask_is_row_invariant(mask):
    """True when the mask does not depend on the row (H) index: [B|1,1,1,W,1] 1-D column masks by shape, and full-size [.,.,H,W,1] masks
    whose rows are all equal (a column mask stored expanded) by content -- one device reduction per mask tensor, cached per
    (storage, version); never evaluated while a hipGraph is being captured (then only the shape decides)."""
    m = mask[..., 0] if (mask.dim() == 5 and mask.shape[-1] == 1) else mask
    if m.dim() < 2:
        return False
    if m.shape[-2] == 1:
        return True
    if not mask.is_cuda or torch.cuda.is_current_stream_capturing():
        return False
    key = (mask.data_ptr(), mask._version, tuple(mask.shape), tuple(mask.stride()), str(mask.dtype))
    hit = _ROW_INV.get(key)
    if hit is None:
        if len(_ROW_INV) >= 16:
            _ROW_INV.pop(next(iter(_ROW_INV)))
        hit = _ROW_INV[key] = (bool((m == m.narrow(-2, 0, 1)).all()), mask.detach())      # the entry keeps the tensor (and its address) alive
    return hit[0]


def row_invariant_view(mask):
    """The one-row view of a mask `mask_is_row_invariant` accepted (what the row-invariant kernels index)."""
    if mask.dim() == 5 and mask.shape[-1] == 1:
        return mask if mask.shape[-3] == 1 else mask.narrow(-3, 0, 1)
    return mask if mask.shape[-2] == 1 else mask.narrow(-2, 0, 1)


def llg_prepare(y, centered, normalization, spatial_dims=None):
    """yt = IFFT_H(y): the hybrid-space data the row-invariant-mask gradient needs (once per cascade)."""
    y = _lib.f32c(y)
    B, C, H, W = _bchw(y)
    _check_last_two(spatial_dims, 4)
    out = torch.empty_like(y)
    _lib.check(_lib.lib().mrx_fft_cols(_lib.ptr(y), _lib.ptr(out), B * C, H, W, 1, _norm(normalization), int(bool(centered)),
                                       _lib.stream_ptr()), "mrx_fft_cols")
    return out


class Llg372Operands:
    """The loop-invariant operands of the W = 372 one-launch gradient (mrx_llg372) in lane order: yt = IFFT_H(y), the sensitivity
    maps and the mask, laid out once per slice by `llg372_prepare`; `work` is the partial-sum workspace reused by every step.

    LIFETIME of the partial planes (`llg372(..., parts=True)`, `llg372_gather`, `adjoint_parts` return views of `work`): they are valid until the
    NEXT gradient call on the same operands object -- every call rewrites planes 0 .. n - 1 in place (only the last, constant plane survives).  The
    consumer (the first RIM layer's loader) runs on the same stream before that call; a caller that wants to keep a step's planes clones them,
    and two streams must not share one operands object (llg372_prepare per stream: the bench's streams each prepare their own)."""
    __slots__ = ("ytp", "sp", "maskp", "mask_batched", "B", "C", "H", "centered", "work", "const_norm", "linear")

    def __init__(self, ytp, sp, maskp, mask_batched, B, C, H, centered, work, linear=False):
        self.ytp, self.sp, self.maskp, self.mask_batched = ytp, sp, maskp, mask_batched
        self.B, self.C, self.H, self.centered, self.work = B, C, H, centered, work
        self.const_norm = None      # normalization for which the last plane of `work` holds the constant term -A^H M y (LLG372_NO_Y)
        self.linear = linear        # the linear part alone (the adjoint of training): own workspace whose constant plane is zero

    def linear_part(self):
        """eta -> A^H M A eta: the same maps and mask, a workspace of its own with a zero constant plane (self-adjoint: training's backward)."""
        work = torch.empty_like(self.work)
        work[-(self.B * self.H * 372 * 2):].zero_()
        return Llg372Operands(None, self.sp, self.maskp, self.mask_batched, self.B, self.C, self.H, self.centered, work, linear=True)


LLG372 = True
# The gradient is affine in eta: g = A^H M A eta - A^H M y.  True: the second term is one constant plane per slice (mrx_llg372_const_plane, the last
# plane of the operands' workspace) that the consumers of the partial planes add like one more coil group, and a step reads S only -- 34.5 MB
# instead of 63.1 MB per launch at 15 coils.  False (test hook): every step reads yt.
LLG372_NO_Y = os.environ.get("MRIDC_AMD_LLG372_NO_Y", "1") != "0"


def llg372_supported(yt, mask):
    """The prime-factor kernel covers W = 372 and masks that depend on the column (and batch) index only."""
    if not (LLG372 and yt.dim() == 5 and int(yt.shape[3]) == 372 and _lib.lib().mrx_llg372_supported(372)):
        return False
    m = mask[..., 0] if (mask.dim() == 5 and mask.shape[-1] == 1) else mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    return m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1


def llg372_prepare(yt, sens, mask, centered, normalization=None):
    """(yt, S, mask) -> Llg372Operands (mrx_llg372_prepare): once per slice, shared by every cascade and time-step.  With `normalization` the
    constant term of the gradient is prepared here too (otherwise by the first llg372 call)."""
    yt, sens = _lib.f32c(yt), _lib.f32c(sens)
    B, C, H, W = _bchw(yt)
    if sens.shape != yt.shape or W != 372:
        raise ValueError(f"llg372_prepare: yt {tuple(yt.shape)} vs maps {tuple(sens.shape)}")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    L = _lib.lib()
    n = int(L.mrx_llg372_operand_floats(B, C, H))
    ytp = torch.empty(n, dtype=torch.float32, device=yt.device)
    sp = torch.empty(n, dtype=torch.float32, device=yt.device)
    maskp = torch.empty(B * 372, dtype=torch.float32, device=yt.device)
    work = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)) + B * H * 372 * 2, dtype=torch.float32, device=yt.device)     # + the constant plane
    _lib.check(L.mrx_llg372_prepare(_lib.ptr(yt), _lib.ptr(sens), _lib.ptr(m), kind, ms, _lib.ptr(ytp), _lib.ptr(sp), _lib.ptr(maskp),
                                    B, C, H, int(bool(centered)), _lib.stream_ptr()), "mrx_llg372_prepare")
    op = Llg372Operands(ytp, sp, maskp, int(ms[0] != 0), B, C, H, bool(centered), work)
    if normalization is not None and LLG372_NO_Y:
        _llg372_const(op, normalization)
    return op


def _llg372_const(op, normalization):
    _lib.check(_lib.lib().mrx_llg372_const_plane(_lib.ptr(op.ytp), _lib.ptr(op.sp), _lib.ptr(op.maskp), op.mask_batched, _lib.ptr(op.work), op.B, op.C,
                                                 op.H, _norm(normalization), int(op.centered), _lib.stream_ptr()), "mrx_llg372_const_plane")
    op.const_norm = _norm(normalization)


def llg372(eta, op, sigma, normalization, out=None, parts=False):
    """log_likelihood_gradient from prepared operands.  parts=False -> out4 [B,4,H,372]; parts=True -> (work, nparts) for
    rim_layer_indrnn_packed_llg (the coil-group partial sums, 1/sigma^2 not yet applied)."""
    import ctypes
    eta = _lib.f32c(eta)
    if tuple(eta.shape) != (op.B, op.H, 372, 2):
        raise ValueError(f"llg372: eta {tuple(eta.shape)} does not match {(op.B, op.H, 372, 2)}")
    n = ctypes.c_int(0)
    if not parts and out is None:
        out = torch.empty(op.B, 4, op.H, 372, dtype=torch.float32, device=eta.device)
    no_y = op.linear or LLG372_NO_Y
    if no_y and not op.linear and op.const_norm != _norm(normalization):
        _llg372_const(op, normalization)                   # (first call, or another normalization than the one prepared for)
    _lib.check(_lib.lib().mrx_llg372(_lib.ptr(eta), None if no_y else _lib.ptr(op.ytp), _lib.ptr(op.sp), _lib.ptr(op.maskp), op.mask_batched,
                                     None if parts else _lib.ptr(out), _lib.ptr(op.work), ctypes.byref(n) if parts else None,
                                     op.B, op.C, op.H, float(1.0 / (float(sigma) ** 2.0)), _norm(normalization), int(op.centered),
                                     _lib.stream_ptr()), "mrx_llg372")
    return (op.work, int(n.value)) if parts else out


LLG372_GATHER = os.environ.get("MRIDC_AMD_LLG372_GATHER", "1") != "0"     # RIMBlock: the final convolution's tap gather rides in the next step's gradient launch


def llg372_gather(eta, taps, b_final, op, sigma, normalization):
    """The deferred form of llg372 on eta_new = rim_final_gather(taps, b_final, eta), in ONE launch (mrx_llg372_gather): returns (work, nparts, eta_new).
    Needs the constant-plane form (LLG372_NO_Y)."""
    import ctypes
    eta = _lib.f32c(eta)
    if tuple(eta.shape) != (op.B, op.H, 372, 2) or taps.numel() < 18 * op.B * op.H * 372:
        raise ValueError(f"llg372_gather: eta {tuple(eta.shape)}, taps {tuple(taps.shape)}")
    if op.linear or not LLG372_NO_Y:
        raise RuntimeError("llg372_gather needs the constant-plane form of the gradient (ops.LLG372_NO_Y)")
    if op.const_norm != _norm(normalization):
        _llg372_const(op, normalization)
    bf = _lib.f32c(b_final.detach()) if b_final is not None else None
    eta_new = torch.empty_like(eta)
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().mrx_llg372_gather(_lib.ptr(eta), _lib.ptr(taps), _lib.ptr(bf), _lib.ptr(eta_new), _lib.ptr(op.sp), _lib.ptr(op.maskp),
                                            op.mask_batched, None, _lib.ptr(op.work), ctypes.byref(n), op.B, op.C, op.H,
                                            float(1.0 / (float(sigma) ** 2.0)), _norm(normalization), int(op.centered), _lib.stream_ptr()),
               "mrx_llg372_gather")
    return op.work, int(n.value), eta_new


def llg372_gather_q(eta, taps_q, edges, b_final, op, sigma, normalization):
    """llg372_gather on the row-pre-summed tap planes of rim_layer2_f16_cb8_q (mrx_llg372_gather_q): returns (work, nparts, eta_new)."""
    import ctypes
    eta = _lib.f32c(eta)
    if tuple(eta.shape) != (op.B, op.H, 372, 2) or taps_q.numel() < 6 * op.B * op.H * 372:
        raise ValueError(f"llg372_gather_q: eta {tuple(eta.shape)}, taps_q {tuple(taps_q.shape)}")
    if op.linear or not LLG372_NO_Y:
        raise RuntimeError("llg372_gather_q needs the constant-plane form of the gradient (ops.LLG372_NO_Y)")
    _check_edges(edges, op.B, op.H, 372, eta.device, "llg372_gather_q")
    if op.const_norm != _norm(normalization):
        _llg372_const(op, normalization)
    bf = _lib.f32c(b_final.detach()) if b_final is not None else None
    eta_new = torch.empty_like(eta)
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().mrx_llg372_gather_q(_lib.ptr(eta), _lib.ptr(taps_q), _lib.ptr(edges), _lib.ptr(bf), _lib.ptr(eta_new), _lib.ptr(op.sp), _lib.ptr(op.maskp),
                                              op.mask_batched, None, _lib.ptr(op.work), ctypes.byref(n), op.B, op.C, op.H,
                                              float(1.0 / (float(sigma) ** 2.0)), _norm(normalization), int(op.centered), _lib.stream_ptr()),
               "mrx_llg372_gather_q")
    return op.work, int(n.value), eta_new


def llg_hinv(eta, yt, sens, mask, sigma, centered, normalization, out=None, work=None):
    """log_likelihood_gradient for a row-invariant mask (yt from llg_prepare): row transforms only."""
    yt, sens, eta = _lib.f32c(yt), _lib.f32c(sens), _lib.f32c(eta)
    B, C, H, W = _bchw(yt)
    if sens.shape != yt.shape or tuple(eta.shape) != (B, H, W, 2):
        raise ValueError("llg_hinv: inconsistent shapes")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    if out is None:
        out = torch.empty(B, 4, H, W, dtype=torch.float32, device=yt.device)
    if work is None:
        work = torch.empty(int(_lib.lib().mrx_llg_hinv_work_floats(B, C, H, W)), dtype=torch.float32, device=yt.device)
    _lib.check(_lib.lib().mrx_llg_hinv(_lib.ptr(eta), _lib.ptr(yt), _lib.ptr(sens), _lib.ptr(m), kind, ms, _lib.ptr(out),
                                       _lib.ptr(work), B, C, H, W, float(1.0 / (float(sigma) ** 2.0)), _norm(normalization),
                                       int(bool(centered)), _lib.stream_ptr()), "mrx_llg_hinv")
    return out


def llg_hinv_parts(eta, yt, sens, mask, sigma, centered, normalization, out=None, work=None):
    """llg_hinv with the coil-chunk partial sums left for the consumer (rim_layer_indrnn_packed_llg).  Returns (out4, work, nparts):
    nparts > 0: `work` holds that many partial planes and out4 was not written; nparts == 0: out4 is the complete gradient."""
    import ctypes
    yt, sens, eta = _lib.f32c(yt), _lib.f32c(sens), _lib.f32c(eta)
    B, C, H, W = _bchw(yt)
    if sens.shape != yt.shape or tuple(eta.shape) != (B, H, W, 2):
        raise ValueError("llg_hinv_parts: inconsistent shapes")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    if out is None:
        out = torch.empty(B, 4, H, W, dtype=torch.float32, device=yt.device)
    if work is None:
        work = torch.empty(int(_lib.lib().mrx_llg_hinv_work_floats(B, C, H, W)), dtype=torch.float32, device=yt.device)
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().mrx_llg_hinv_parts(_lib.ptr(eta), _lib.ptr(yt), _lib.ptr(sens), _lib.ptr(m), kind, ms, _lib.ptr(out),
                                             _lib.ptr(work), ctypes.byref(n), B, C, H, W, float(1.0 / (float(sigma) ** 2.0)),
                                             _norm(normalization), int(bool(centered)), _lib.stream_ptr()), "mrx_llg_hinv_parts")
    return out, work, int(n.value)


def rim_layer1_inplace_ok(Cin, F, k, dilation):
    """True when mrx_rim_layer_indrnn_packed[_llg] runs the split-bf16 first-layer kernel (rim_layer1_sb.hip), whose epilogue reads every
    h_prev element in the lane that writes h_new there: `out` may then be h_prev itself."""
    return (int(Cin) <= 4 and int(F) == 64 and int(k) == 5 and int(dilation) == 1
            and bool(_lib.lib().mrx_rim_layer1_xmax_supported(int(Cin), int(F), int(k), int(dilation))))


def rim_layer_indrnn_packed_llg(eta, part, nparts, sigma, packed, F, k, dilation, b_conv, b_ih, hh, h_prev, out=None, xmax=None):
    """The tuned fused first RIM layer reading log_likelihood_gradient's pieces (eta and the partial coil sums) directly.  `xmax` as in
    rim_layer_indrnn_packed."""
    eta = _lib.f32c(eta)
    B, H, W, _ = [int(v) for v in eta.shape]
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, F, H, W, dtype=torch.float32, device=eta.device)
    if xmax is not None:
        _lib.check(_lib.lib().mrx_rim_layer_indrnn_packed_llg_xmax(_lib.ptr(eta), _lib.ptr(part), int(nparts), float(1.0 / (float(sigma) ** 2.0)),
                                                                   _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp),
                                                                   _lib.ptr(out), _lib.ptr(xmax), B, int(F), H, W, int(k), int(dilation),
                                                                   _lib.stream_ptr()), "mrx_rim_layer_indrnn_packed_llg_xmax")
        return out
    _lib.check(_lib.lib().mrx_rim_layer_indrnn_packed_llg(_lib.ptr(eta), _lib.ptr(part), int(nparts), float(1.0 / (float(sigma) ** 2.0)),
                                                          _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp),
                                                          _lib.ptr(out), B, int(F), H, W, int(k), int(dilation), _lib.stream_ptr()),
               "mrx_rim_layer_indrnn_packed_llg")
    return out


def soft_dc(pred, ref, mask, dc_weight):
    """where(mask, pred - ref, 0) * dc_weight."""
    pred, ref = _lib.f32c(pred), _lib.f32c(ref)
    B, C, H, W = _bchw(pred)
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    w = _lib.f32c(dc_weight.detach().reshape(-1))
    out = torch.empty_like(pred)
    _lib.check(_lib.lib().mrx_soft_dc(_lib.ptr(pred), _lib.ptr(ref), _lib.ptr(m), kind, ms, _lib.ptr(w), _lib.ptr(out),
                                      B, C, H, W, _lib.stream_ptr()), "mrx_soft_dc")
    return out


def dc_combine(base, pred, ref, mask, dc_weight, eta_k):
    """base - where(mask, pred - ref, 0) * dc_weight - eta_k."""
    base, pred, ref, eta_k = _lib.f32c(base), _lib.f32c(pred), _lib.f32c(ref), _lib.f32c(eta_k)
    B, C, H, W = _bchw(pred)
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    w = _lib.f32c(dc_weight.detach().reshape(-1))
    out = torch.empty_like(pred)
    _lib.check(_lib.lib().mrx_dc_combine(_lib.ptr(base), _lib.ptr(pred), _lib.ptr(ref), _lib.ptr(m), kind, ms, _lib.ptr(w),
                                         _lib.ptr(eta_k), _lib.ptr(out), B, C, H, W, _lib.stream_ptr()), "mrx_dc_combine")
    return out


def hard_dc(pred, ref, mask, dc_weight):
    """VSNet DataConsistencyLayer: ((1 - mask) * pred + mask * ref) * dc_weight (vsnet_block.py:23-25)."""
    pred, ref = _lib.f32c(pred), _lib.f32c(ref)
    B, C, H, W = _bchw(pred)
    if ref.shape != pred.shape:
        raise ValueError(f"hard_dc: {tuple(pred.shape)} vs {tuple(ref.shape)}")
    if mask.dtype == torch.bool:   # `1 - mask` (vsnet_block.py:25) is not defined for bool tensors in torch either
        raise RuntimeError("Subtraction, the `-` operator, with a bool tensor is not supported. If you are trying to invert a mask, "
                           "use the `~` or `logical_not()` operator instead.")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    w = _lib.f32c(dc_weight.detach().reshape(-1))
    out = torch.empty_like(pred)
    _lib.check(_lib.lib().mrx_hard_dc(_lib.ptr(pred), _lib.ptr(ref), _lib.ptr(m), kind, ms, _lib.ptr(w), _lib.ptr(out), B, C, H, W,
                                      _lib.stream_ptr()), "mrx_hard_dc")
    return out


def vs_average(kspace, pred, sx, param):
    """VSNet WeightedAverageTerm as called by the block: param * (kspace + pred) + (1 - param) * sx, sx [B,H,W,2] broadcast
    over the coils (vsnet_block.py:35-36,145)."""
    kspace, pred, sx = _lib.f32c(kspace), _lib.f32c(pred), _lib.f32c(sx)
    B, C, H, W = _bchw(kspace)
    if pred.shape != kspace.shape or tuple(sx.shape) != (B, H, W, 2):
        raise ValueError(f"vs_average: {tuple(kspace.shape)}, {tuple(pred.shape)}, {tuple(sx.shape)}")
    p = _lib.f32c(param.detach().reshape(-1))
    out = torch.empty_like(kspace)
    _lib.check(_lib.lib().mrx_vs_average(_lib.ptr(kspace), _lib.ptr(pred), _lib.ptr(sx), _lib.ptr(p), _lib.ptr(out), B, C, H, W,
                                         _lib.stream_ptr()), "mrx_vs_average")
    return out


def coil_sum(k, mask=None):
    """sum_c k[b,c] * mask -> [B,H,W,2] (the sigmanet layers sum k-space over the coil axis, dc_layers.py:69-81)."""
    k = _lib.f32c(k)
    B, C, H, W = _bchw(k)
    m, kind, ms = _lib.mask_args(mask, B, C, H, W) if mask is not None else (None, 0, _lib.i64_array([0, 0, 0, 0]))
    out = torch.empty(B, H, W, 2, dtype=torch.float32, device=k.device)
    _lib.check(_lib.lib().mrx_coil_sum(_lib.ptr(k), _lib.ptr(m), kind, ms, _lib.ptr(out), B, C, H, W, _lib.stream_ptr()), "mrx_coil_sum")
    return out


def dc_bcast(a, y, mask, alpha=None):
    """alpha None: (a - y) * mask; else (1 - mask) * a + mask * (alpha * a + (1 - alpha) * y).  y [B,C,H,W,2]; a [B,H,W,2]
    (broadcast over the coils) or [B,C,H,W,2]."""
    a, y = _lib.f32c(a), _lib.f32c(y)
    B, C, H, W = _bchw(y)
    if tuple(a.shape) == (B, H, W, 2):
        a_coils = 0
    elif a.shape == y.shape:
        a_coils = 1
    else:
        raise ValueError(f"dc_bcast: {tuple(a.shape)} vs {tuple(y.shape)}")
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    al = None if alpha is None else _lib.f32c(alpha.detach().reshape(-1))
    out = torch.empty_like(y)
    _lib.check(_lib.lib().mrx_dc_bcast(_lib.ptr(a), a_coils, _lib.ptr(y), _lib.ptr(m), kind, ms, _lib.ptr(al), 0 if alpha is None else 1,
                                       _lib.ptr(out), B, C, H, W, _lib.stream_ptr()), "mrx_dc_bcast")
    return out


def lincomb(x, g, p, mode):
    """mode 0: x - p * g; mode 1: p * x + (1 - p) * g, with torch broadcasting of x against g restricted to "the smaller one
    repeats over the leading axes of the larger" (what the sigmanet layers need)."""
    x, g = _lib.f32c(x), _lib.f32c(g)
    shape = torch.broadcast_shapes(x.shape, g.shape)
    n = 1
    for v in shape:
        n *= int(v)
    for t in (x, g):
        sh = list(t.shape)
        while sh and sh[0] == 1:
            sh.pop(0)
        if tuple(shape[len(shape) - len(sh):]) != tuple(sh):     # the operand must repeat over the leading axes only
            raise NotImplementedError(f"lincomb: broadcast of {tuple(x.shape)} with {tuple(g.shape)}")
    pv = _lib.f32c(p.detach().reshape(-1))
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_lincomb(_lib.ptr(x), x.numel(), _lib.ptr(g), g.numel(), _lib.ptr(pv), int(mode), _lib.ptr(out), n,
                                      _lib.stream_ptr()), "mrx_lincomb")
    return out


def mul_mask(x, mask):
    """x * mask for a [B,C,H,W,2] tensor and a mask broadcastable to [B,C,H,W,1] (dc_layers.py:214,227: `x * mask`)."""
    x = _lib.f32c(x)
    B, C, H, W = _bchw(x)
    m, _, ms = _lib.mask_args(mask.float(), B, C, H, W)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_apply_mask(_lib.ptr(x), _lib.ptr(m), _lib.ptr(out), B, C, H, W, ms, _lib.stream_ptr()), "mrx_apply_mask")
    return out


def cdot(a, b):
    """Per-batch complex dot product sum(a * conj(b)) over everything but axis 0 -> [B,2] (dc_layers.py:160-165)."""
    a, b = _lib.f32c(a), _lib.f32c(b)
    if a.shape != b.shape or a.shape[-1] != 2:
        raise ValueError(f"cdot: {tuple(a.shape)} vs {tuple(b.shape)}")
    B = int(a.shape[0])
    n = a.numel() // (2 * B)
    out = torch.empty(B, 2, dtype=torch.float32, device=a.device)
    work = torch.empty(int(_lib.lib().mrx_cdot_work_floats(B)), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mrx_cdot(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(work), B, n, _lib.stream_ptr()), "mrx_cdot")
    return out


def cg_step(x, r, p, q, rr, pq):
    """In place: alpha = rr * conj(pq) / |pq|^2; x += alpha * p; r -= alpha * q (dc_layers.py:186-192)."""
    for t in (x, r, p, q):
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.shape == x.shape):
            raise ValueError("cg_step: contiguous fp32 tensors of one shape expected")
    _lib.require_gpu(x)
    B = int(x.shape[0])
    _lib.check(_lib.lib().mrx_cg_step(_lib.ptr(x), _lib.ptr(r), _lib.ptr(p), _lib.ptr(q), _lib.ptr(_lib.f32c(rr)), _lib.ptr(_lib.f32c(pq)),
                                      B, x.numel() // (2 * B), _lib.stream_ptr()), "mrx_cg_step")


def cg_dir(p, r, rr_new, rr):
    """In place: p = r + (rr_new / rr) * p (dc_layers.py:193-195)."""
    if not (p.is_contiguous() and r.is_contiguous() and p.shape == r.shape and p.dtype == torch.float32):
        raise ValueError("cg_dir: contiguous fp32 tensors of one shape expected")
    _lib.require_gpu(p)
    B = int(p.shape[0])
    _lib.check(_lib.lib().mrx_cg_dir(_lib.ptr(p), _lib.ptr(r), _lib.ptr(_lib.f32c(rr_new)), _lib.ptr(_lib.f32c(rr)), B,
                                     p.numel() // (2 * B), _lib.stream_ptr()), "mrx_cg_dir")


def _nchw(x):
    if x.dim() != 4:
        raise ValueError(f"expected a [B,C,H,W] tensor, got {tuple(x.shape)}")
    return [int(v) for v in x.shape]


# Winograd form of 3x3 convolutions into 64 channels (mrx_conv3x3_wino): transformed weights per (storage, version), oldest out
WINOGRAD_CONV = True
WINOGRAD_MIN_CIN = 16          # below this the 8-channel chunks are mostly padding and the direct kernels win
_WINO_PACKS = {}


def _wino_conv_pack(weight):
    """Transformed + packed weights of `weight`, cached per (storage address, version).  The entry keeps a detached alias of the tensor
    it was made from -- i.e. its STORAGE -- so the address cannot be recycled for other weights while the entry lives, even if the
    parameter's `.data` is re-pointed (training.FlatParameters does that)."""
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), tuple(weight.stride()), _wcap())
    hit = _WINO_PACKS.get(key)
    if hit is None:
        if len(_WINO_PACKS) >= 128:
            _WINO_PACKS.pop(next(iter(_WINO_PACKS)))
        w = _lib.f32c(weight.detach())
        F, Cin = int(w.shape[0]), int(w.shape[1])
        blk = int(_lib.lib().mrx_rim_layer_wino_pack_floats(Cin, 64))
        packed = torch.empty(blk * (F // 64), dtype=torch.float32, device=w.device)
        for ob in range(F // 64):                                        # one packed block per 64 output channels
            _lib.check(_lib.lib().mrx_rim_layer_wino_pack(_lib.ptr(w[ob * 64:(ob + 1) * 64]), None, _lib.ptr(packed[ob * blk:]), Cin, 64,
                                                          _lib.stream_ptr()), "mrx_rim_layer_wino_pack")
        hit = _WINO_PACKS[key] = (packed, weight.detach())   # the detached alias pins the STORAGE (p.data may be re-pointed later)
    return hit[0]


_PACKS_1X1 = {}


def _conv1x1_pack(weight):
    """Packed [C,C,1,1] weights (C = 64 | 128) for mrx_conv1x1_sq, cached like the Winograd packs (the entry keeps the source alive)."""
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), tuple(weight.stride()), _wcap())
    hit = _PACKS_1X1.get(key)
    if hit is None:
        if len(_PACKS_1X1) >= 128:
            _PACKS_1X1.pop(next(iter(_PACKS_1X1)))
        w = _lib.f32c(weight.detach())
        C = int(w.shape[0])
        packed = torch.empty(int(_lib.lib().mrx_conv1x1_sq_pack_floats(C)), dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().mrx_conv1x1_sq_pack(_lib.ptr(w), _lib.ptr(packed), C, _lib.stream_ptr()), "mrx_conv1x1_sq_pack")
        hit = _PACKS_1X1[key] = (packed, weight.detach())
    return hit[0]


def conv1x1_sq_supported(cin, cout):
    return bool(_lib.lib().mrx_conv1x1_sq_supported(int(cin), int(cout)))


def conv1x1_64(x, weight, bias=None, act=ACT_NONE, slope=0.0, hh=None, h_prev=None, out=None):
    """act(W x + bias [+ hh * h_prev]) for a 1x1 convolution C -> C, C = 64 or 128 (mrx_conv1x1_sq)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    if not conv1x1_sq_supported(Cin, int(weight.shape[0])) or tuple(weight.shape) != (Cin, Cin, 1, 1):
        raise ValueError(f"conv1x1: x {tuple(x.shape)}, weight {tuple(weight.shape)}")
    packed = _conv1x1_pack(weight)
    b = _lib.f32c(bias.detach()) if bias is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1)) if hh is not None and h_prev is not None else None
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty_like(x)
    if H3X3_CONV and _lib.arith() == "f16x2" and _lib.lib().mrx_conv1x1_sq_xmax_supported(Cin):
        # a 3x3 layer on two-term fp16 operands may read this next (qRIM: cell of stack 1 -> convolution of stack 2): the kernel keeps the
        # bound of its outputs, the consumer finds it on the tensor (ops._plain_bound) instead of running mrx_max_abs over it
        xmax = _zero_scalar(x.device)
        p16 = _precision16()            # inside inference_precision(16): x and W rounded to fp16 once (mrx_conv1x1_sq_p16)
        _lib.check((_lib.lib().mrx_conv1x1_sq_p16 if p16 else _lib.lib().mrx_conv1x1_sq_xmax)(
            _lib.ptr(x), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), _lib.ptr(xmax), B, Cin, H * W, int(act), float(slope),
            _lib.stream_ptr()), "mrx_conv1x1_sq_p16" if p16 else "mrx_conv1x1_sq_xmax")
        return _attach_bound(out, xmax)
    _lib.check(_lib.lib().mrx_conv1x1_sq(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), B, Cin,
                                         H * W, int(act), float(slope), _lib.stream_ptr()), "mrx_conv1x1_sq")
    return out


def conv3x3_wino_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv3x3_wino_supported(int(Cin), int(Cout), int(k), int(dilation)))


def conv3x3_wino(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None):
    """3x3 convolution into 64 (or a multiple of 64) channels as Winograd F(2x2,3x3) on the matrix cores (differs from the direct form by
    fp32 round-off, ~2e-7 of the output norm)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    Cout = int(weight.shape[0])
    packed = _wino_conv_pack(weight)
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_conv3x3_wino(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(out), B, Cin, Cout, H, W, int(dilation),
                                           int(pad_mode), int(act), float(slope), _lib.stream_ptr()), "mrx_conv3x3_wino")
    return out


SB_CONV = _lib.arith() != "fp32"       # 64 -> 64 3x3 convolutions on split operands (MRIDC_AMD_ARITH=fp32: the fp32 Winograd kernel)
_PACKS_SB = {}


def conv3x3_sb_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv3x3_sb_supported(int(Cin), int(Cout), int(k), int(dilation)))


SB_CHAIN = True                   # (module attribute: a test hook) 64-channel convolutions keep the bound of their outputs; a convolution fed by one
                                  # that did runs on two-term fp16 operands (mrx_conv3x3_sb_chain)
_PACKS_SB_F16 = {}


def conv3x3_sb(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None):
    """3x3 convolution 64 -> 64 (dilation 1 or 2, zero or replicate padding) + bias + activation as a direct convolution on the matrix pipe with
    fp32 results (the convolution stage of the dominant RIM layer on its own).  Operands: three bf16 terms (six term products per multiply) --
    or, when x carries the bound of its maximum (kept by the convolution that produced it: ops._attach_bound), two fp16 terms scaled by it (three
    term products): mrx_conv3x3_sb_chain.  The operand packs are cached per (storage, version)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    L = _lib.lib()
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), tuple(weight.stride()), _wcap())
    chain = SB_CHAIN and _lib.arith() == "f16x2"
    bound_in = _lib.bound_of(x) if chain else None                # (the caller's tensor, or a fresh fp32 copy nothing is known about)
    hit = hit16 = None
    if bound_in is None:
        hit = _PACKS_SB.get(key)
        if hit is None:
            if len(_PACKS_SB) >= 256:
                _PACKS_SB.clear()
            w = _lib.f32c(weight.detach())
            packed = torch.empty(int(L.mrx_rim_layer2_sb_pack_floats()), dtype=torch.float32, device=w.device)
            _lib.check(L.mrx_rim_layer2_sb_pack(_lib.ptr(w), None, None, _lib.ptr(packed), _lib.stream_ptr()), "mrx_rim_layer2_sb_pack")
            hit = (packed, weight.detach())                     # (the detached alias pins the STORAGE: its address cannot be recycled, even if p.data is re-pointed)
            _PACKS_SB[key] = hit
    else:
        hit16 = _PACKS_SB_F16.get(key)
        if hit16 is None:
            if len(_PACKS_SB_F16) >= 256:
                _PACKS_SB_F16.clear()
            hit16 = (rim_layer2_f16_pack(weight, None, None), weight.detach())
            _PACKS_SB_F16[key] = hit16
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, 64, H, W, dtype=torch.float32, device=x.device)
    if chain:
        xmax_out = _zero_scalar(x.device)
        _lib.check(L.mrx_conv3x3_sb_chain(_lib.ptr(x), _lib.ptr(hit[0]) if hit else None, _lib.ptr(hit16[0]) if hit16 else None, _lib.ptr(b),
                                          _lib.ptr(out), _lib.ptr(bound_in), _lib.ptr(xmax_out), B, H, W, int(dilation), int(pad_mode), int(act),
                                          float(slope), _lib.stream_ptr()), "mrx_conv3x3_sb_chain")
        return _attach_bound(out, xmax_out)
    _lib.check(L.mrx_conv3x3_sb(_lib.ptr(x), _lib.ptr(hit[0]), _lib.ptr(b), _lib.ptr(out), B, H, W, int(dilation), int(pad_mode),
                                int(act), float(slope), _lib.stream_ptr()), "mrx_conv3x3_sb")
    return out


SBS_CONV = _lib.arith() != "fp32"
SBS_MIN_COUT = 32
_PACKS_SBS = {}


def conv_sbs_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv_sbs_supported(int(Cin), int(Cout), int(k), int(dilation)))


def conv_sbs(x, weight, bias, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None):
    """3x3 / 5x5 convolution (dilation 1) of Cin <= 8 channels into Cout <= 128 + bias + activation on the bf16 matrix pipe with fp32 results
    (mrx_conv_sbs: three-term operand split, two taps per MFMA): the first layers of the cascades.  Pack cached per (storage, version)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    Cout, _, k, _ = [int(v) for v in weight.shape]
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), tuple(weight.stride()), _wcap())
    hit = _PACKS_SBS.get(key)
    if hit is None:
        if len(_PACKS_SBS) >= 256:
            _PACKS_SBS.clear()
        w = _lib.f32c(weight.detach())
        packed = torch.empty(int(_lib.lib().mrx_conv_sbs_pack_floats(Cout, k)), dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().mrx_conv_sbs_pack(_lib.ptr(w), _lib.ptr(packed), Cin, Cout, k, _lib.stream_ptr()), "mrx_conv_sbs_pack")
        hit = (packed, weight.detach())
        _PACKS_SBS[key] = hit
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    p16 = _precision16() and _lib.arith() == "f16x2"          # inside inference_precision(16): the first fp16 term only (mrx_conv_sbs_p16)
    _lib.check((_lib.lib().mrx_conv_sbs_p16 if p16 else _lib.lib().mrx_conv_sbs)(_lib.ptr(x), _lib.ptr(hit[0]), _lib.ptr(b), _lib.ptr(out), B, Cin, Cout, H, W, k,
                                                                                  int(pad_mode), int(act), float(slope), _lib.stream_ptr()),
               "mrx_conv_sbs_p16" if p16 else "mrx_conv_sbs")
    return out


TAPS_CONV = True
_TAPS_W = {}


def conv3x3_taps_supported(Cin, Cout):
    """The contraction runs on the square 1x1 kernel (C = 64 or 128, mrx_conv1x1_sq) with the 9 Cout tap rows padded to C."""
    return 9 * int(Cout) <= int(Cin) and int(Cout) <= 4 and conv1x1_sq_supported(int(Cin), int(Cin))


def conv3x3_taps(x, weight, bias, pad_mode=PAD_ZERO, out=None):
    """3x3 convolution (dilation 1) of C = 64 / 128 channels into Cout <= 4 as a per-pixel channel contraction C -> 9 Cout on the matrix cores
    (the square 1x1 kernel, tap rows [tap * Cout + co] padded to C) + a nine-tap gather (mrx_taps_gather): the direct form costs 18 Cout
    vector FMAs per (pixel, input channel) -- qRIM's 128 -> 4 final layer at 256 x 256: 75 us."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    Cout = int(weight.shape[0])
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), _wcap())
    w1 = _TAPS_W.get(key)
    if w1 is None:
        if len(_TAPS_W) >= 64:
            _TAPS_W.clear()
        wp = torch.zeros(Cin, Cin, 1, 1, dtype=torch.float32, device=weight.device)
        wp[:9 * Cout] = _lib.f32c(weight.detach()).reshape(Cout, Cin, 9).permute(2, 0, 1).reshape(9 * Cout, Cin, 1, 1)
        w1 = (wp, weight.detach())                          # (the detached alias pins the storage: its address cannot be recycled)
        _TAPS_W[key] = w1
    if Cin == 128:                           # only the first 64 of the 128 padded rows are needed (9 Cout <= 36)
        taps = torch.empty(B, 64, H, W, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mrx_conv1x1_sq_head128(_lib.ptr(x), _lib.ptr(_conv1x1_pack(w1[0])), _lib.ptr(taps), B, H * W, _lib.stream_ptr()),
                   "mrx_conv1x1_sq_head128")
    else:
        taps = conv1x1_64(x, w1[0])
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_taps_gather(_lib.ptr(taps), _lib.ptr(b), _lib.ptr(out), B, int(taps.shape[1]), Cout, H, W, int(pad_mode),
                                          _lib.stream_ptr()), "mrx_taps_gather")
    return out


# The reference's inference precision (`trainer.precision: 16` in every *_run.yaml of its model zoo, e.g. base_vn_run.yaml:98, base_qcirim_run.yaml:204): inside
# `inference_precision(16)` the 3x3 convolutions of the two-term fp16 route run on ONE fp16 term (mrx_unet_conv3x3_p16, mrx_conv3x3_p16).  Set by the models'
# inference forward (VarNet, UNet, qCIRIM); None inside / outside: fp32-class results.  (The RIM blocks of CIRIM have their own fp16 kernels: RIMBlock.precision.)
import threading as _threading

_INFERENCE_PRECISION = _threading.local()       # per thread: two threads may run models of different precision side by side


class inference_precision:
    def __init__(self, precision):
        self.precision = precision

    def __enter__(self):
        self.keep = getattr(_INFERENCE_PRECISION, "value", None)
        _INFERENCE_PRECISION.value = self.precision
        return self

    def __exit__(self, *exc):
        _INFERENCE_PRECISION.value = self.keep
        return False


def _precision16():
    p = getattr(_INFERENCE_PRECISION, "value", None)
    return p is not None and str(p).lower() in ("16", "fp16", "16-mixed")


def resolve_precision16(model_precision):
    """16 if the model's precision (its trainer's / cfg's; None: the process default, MRIDC_AMD_PRECISION) asks for the reference's `precision: 16`, else None."""
    prec = model_precision if model_precision is not None else _lib.precision()
    return 16 if str(prec).lower() in ("16", "fp16", "16-mixed") else None


H3X3_CONV = True                  # (module attribute: a test hook) 3x3 convolutions of wide layers on two-term fp16 operands (mrx_conv3x3_h)
H3X3_MIN_CIN = 32


def conv3x3_h_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv3x3_h_supported(int(Cin), int(Cout), int(k), int(dilation)))


def conv3x3_h(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None, bound=None):
    """act(conv3x3(x) + bias) for any channel counts on two-term fp16 operands (mrx_conv3x3_h; csrc/unet_f16.hip).  `bound`: device scalar
    >= max |x| (default: the one x carries, else measured by mrx_max_abs)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    weight = _lib.f32c(weight.detach())
    Cout = int(weight.shape[0])
    if tuple(weight.shape[1:]) != (Cin, 3, 3):
        raise RuntimeError(f"conv3x3_h: weight {tuple(weight.shape)} vs {Cin} input channels")
    L = _lib.lib()

    def make():
        pk = torch.empty(int(L.mrx_unet_conv3x3_pack_floats(Cout, Cin)), dtype=torch.float32, device=weight.device)
        _lib.check(L.mrx_unet_conv3x3_pack(_lib.ptr(weight), Cout, Cin, _lib.ptr(pk), _lib.stream_ptr()), "mrx_unet_conv3x3_pack")
        return pk
    packed = _UNET_PACKS.get(weight, (), make)
    bnd = bound if bound is not None else _plain_bound(x)
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    p16 = _precision16()                  # inside inference_precision(16): the first fp16 term only (mrx_conv3x3_p16)
    _lib.check((L.mrx_conv3x3_p16 if p16 else L.mrx_conv3x3_h)(_lib.ptr(x), _lib.ptr(bnd), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(out), B, Cin, Cout, H, W,
                                                              int(dilation), int(pad_mode), int(act), float(slope), _lib.stream_ptr()),
               "mrx_conv3x3_p16" if p16 else "mrx_conv3x3_h")
    return out


def conv2d(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None):
    """'same' conv, stride 1, square odd kernel (mrx_conv2d; 3x3 into 64 channels: mrx_conv3x3_wino)."""
    x = _lib.f32c(x)
    _lib.require_gpu(weight)
    B, Cin, H, W = _nchw(x)
    Cout, Cin_w, kh, kw = [int(v) for v in weight.shape]
    if Cin_w != Cin:
        raise RuntimeError(f"input has inconsistent input_size: got {Cin}, expected {Cin_w}")
    if kh != kw:
        raise NotImplementedError("square kernels only")
    # (5x5 only: the tuned 3x3 fp32 kernel is as fast for two input channels -- 30 vs 32 us at 2 -> 64, 640 x 372)
    if (SBS_CONV and kh == 5 and Cout >= SBS_MIN_COUT and act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and conv_sbs_supported(Cin, Cout, kh, dilation)
            and (out is None or out.data_ptr() != x.data_ptr())):
        return conv_sbs(x, weight, bias, pad_mode, act, slope, out)
    if (SB_CONV and kh == 3 and act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and conv3x3_sb_supported(Cin, Cout, kh, dilation)
            and (out is None or out.data_ptr() != x.data_ptr())):
        return conv3x3_sb(x, weight, bias, dilation, pad_mode, act, slope, out)
    if (H3X3_CONV and kh == 3 and Cin >= H3X3_MIN_CIN and Cout >= 16 and act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
            and conv3x3_h_supported(Cin, Cout, kh, dilation) and (out is None or out.data_ptr() != x.data_ptr())):
        return conv3x3_h(x, weight, bias, dilation, pad_mode, act, slope, out)
    if (WINOGRAD_CONV and kh == 3 and Cin >= WINOGRAD_MIN_CIN and act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
            and conv3x3_wino_supported(Cin, Cout, kh, dilation) and (out is None or out.data_ptr() != x.data_ptr())):
        return conv3x3_wino(x, weight, bias, dilation, pad_mode, act, slope, out)
    if (kh == 1 and conv1x1_sq_supported(Cin, Cout) and act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
            and (out is None or out.data_ptr() != x.data_ptr())):
        return conv1x1_64(x, weight, bias, act, slope, out=out)          # per-pixel 64x64 GEMM, HBM-bound
    if TAPS_CONV and kh == 3 and int(dilation) == 1 and act == ACT_NONE and conv3x3_taps_supported(Cin, Cout):
        return conv3x3_taps(x, weight, bias, pad_mode, out)
    weight = _lib.f32c(weight.detach())
    b = _lib.f32c(bias.detach()) if bias is not None else None
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_conv2d(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(b), _lib.ptr(out), B, Cin, Cout, H, W, kh,
                                     int(dilation), int(pad_mode), int(act), float(slope), _lib.stream_ptr()), "mrx_conv2d")
    return out


_PACKS_BF16 = {}


def conv_bf16_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv_bf16_supported(int(Cin), int(Cout), int(k), int(dilation)))


def _conv_bf16_pack(weight, transposed):
    """bf16 MFMA operand pack of a [Cout,Cin,k,k] weight (or of its flipped, channel-transposed form: the data gradient's weights),
    cached per (storage, version) like the other packs (the entry keeps the source tensor alive)."""
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), tuple(weight.stride()), bool(transposed), _wcap())
    hit = _PACKS_BF16.get(key)
    if hit is None:
        if len(_PACKS_BF16) >= 256:
            _PACKS_BF16.pop(next(iter(_PACKS_BF16)))
        w = _lib.f32c(weight.detach())
        d0, d1, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        cin, cout = (d0, d1) if transposed else (d1, d0)          # of the convolution the pack is for
        L = _lib.lib()
        packed = torch.empty(int(L.mrx_conv_bf16_pack_bytes(cin, cout, k)), dtype=torch.uint8, device=w.device)
        _lib.check(L.mrx_conv_bf16_pack(_lib.ptr(w), _lib.ptr(packed), cin, cout, k, int(bool(transposed)), _lib.stream_ptr()),
                   "mrx_conv_bf16_pack")
        hit = _PACKS_BF16[key] = (packed, weight.detach())
    return hit[0]


def conv2d_bf16(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, hh=None, h_prev=None, transposed=False):
    """'same' convolution with bf16 operands and fp32 accumulation (mrx_conv2d_bf16) on fp32 tensors: act(conv(x) + bias [+ hh * h_prev]).
    `transposed`: x is an output gradient and `weight` the forward weight [Cout,Cin,k,k] -- the zero-padded data-gradient convolution."""
    x = _lib.f32c(x)
    B, Cx, H, W = _nchw(x)
    d0, d1, kh, kw = [int(v) for v in weight.shape]
    cin, cout = (d0, d1) if transposed else (d1, d0)
    if cin != Cx:
        raise RuntimeError(f"input has inconsistent input_size: got {Cx}, expected {cin}")
    if kh != kw or not conv_bf16_supported(cin, cout, kh, dilation):
        raise NotImplementedError(f"conv2d_bf16: Cin={cin} Cout={cout} k={kh}x{kw} dilation={dilation}")
    packed = _conv_bf16_pack(weight, transposed)
    b = _lib.f32c(bias.detach()) if bias is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1)) if hh is not None and h_prev is not None else None
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_conv2d_bf16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), B, cin, cout,
                                          H, W, kh, int(dilation), int(pad_mode), int(act), float(slope), _lib.stream_ptr()),
               "mrx_conv2d_bf16")
    return out


def conv_dgrad_bf16_ext(dy, weight, dilation, ext):
    """Data gradient on the padded domain: the zero-padded convolution of dy (read as if zero-extended by `ext`) with the flipped,
    transposed `weight` [Cout,Cin,k,k] -> [B,Cin,H + 2 ext,W + 2 ext] (mrx_conv2d_bf16_ext; no extended copy of dy is made)."""
    dy = _lib.f32c(dy)
    B, C, H, W = _nchw(dy)
    cout_w, cin_w, k = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
    if C != cout_w:
        raise RuntimeError(f"input has inconsistent input_size: got {C}, expected {cout_w}")
    packed = _conv_bf16_pack(weight, True)
    out = torch.empty(B, cin_w, H + 2 * ext, W + 2 * ext, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().mrx_conv2d_bf16_ext(_lib.ptr(dy), _lib.ptr(packed), _lib.ptr(out), B, cout_w, cin_w, H, W, k, int(dilation), int(ext),
                                              _lib.stream_ptr()), "mrx_conv2d_bf16_ext")
    return out


def conv_wgrad_bf16_supported(Cin, Cout, k, dilation):
    return bool(_lib.lib().mrx_conv_wgrad_bf16_supported(int(Cin), int(Cout), int(k), int(dilation)))


def conv_wgrad_bf16_preferred(Cin, Cout, k, dilation):
    """Shapes where the bf16 weight gradient is also the faster one (measured at 640 x 372: 64 -> 64 3x3 d2 109 vs 254 us, 1x1 48 vs 95,
    64 -> 2 3x3 73 vs 106; the 5x5 4 -> 64 layer is correct in bf16 but slower -- 131 vs 94 us: 28 of its 32 channel-lanes carry zeros)."""
    return conv_wgrad_bf16_supported(Cin, Cout, k, dilation) and int(k) != 5


def conv_wgrad_bf16(x, dy, k, dilation=1, pad_mode=PAD_REPLICATE, out=None, accumulate=False):
    """Weight gradient of a 'same' convolution with bf16 operands (mrx_conv_wgrad_bf16_any): dw [Cout,Cin,k,k] for the shapes of
    conv_wgrad_bf16_supported (64 -> 64 1x1 / 3x3 d2; 3x3 64 -> <= 32; 5x5 <= 32 -> 64)."""
    x, dy = _lib.f32c(x), _lib.f32c(dy)
    B, Cin, H, W = _nchw(x)
    Cout = int(dy.shape[1])
    if tuple(dy.shape) != (B, Cout, H, W) or not conv_wgrad_bf16_supported(Cin, Cout, k, dilation):
        raise ValueError(f"conv_wgrad_bf16: x {tuple(x.shape)}, dy {tuple(dy.shape)}, k={k}, dilation={dilation}")
    if out is None:
        out = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=x.device)
        accumulate = False
    L = _lib.lib()
    work = torch.empty(int(L.mrx_conv_wgrad_bf16_any_work_floats(B, Cin, Cout, H, W, int(k))), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv_wgrad_bf16_any(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(out), _lib.ptr(work), B, Cin, Cout, H, W, int(k), int(dilation),
                                         int(pad_mode), int(bool(accumulate)), _lib.stream_ptr()), "mrx_conv_wgrad_bf16_any")
    return out


# ---- mixed-precision training with bf16 STORAGE (csrc/train_bf16.hip, conv_bf16.hip OUT 1 / 2; callers: mridc_amd/training.py) ---------------------
# "Pair tensors": what torch.autocast keeps in half precision (convolution results, the gradients flowing into them) lives in HBM as
# int32 [B, C / 2, H, W] = (bf16 of channel 2p, bf16 of channel 2p + 1).
TL_WGRAD_IN = True          # the first layer's weight gradient on its own kernel (mrx_tl_wgrad_in); False: the generic thin kernel (test hook)


def f32_to_pairs(x):
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    if C % 2:
        raise ValueError("f32_to_pairs: odd channel count")
    out = torch.empty(B, C // 2, H, W, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().mrx_tl_f32_to_pairs(_lib.ptr(x), _lib.ptr(out), B * (C // 2), H * W, _lib.stream_ptr()), "mrx_tl_f32_to_pairs")
    return out


def pairs_to_f32(p):
    B, C2, H, W = [int(v) for v in p.shape]
    out = torch.empty(B, 2 * C2, H, W, dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().mrx_tl_pairs_to_f32(_lib.ptr(p), _lib.ptr(out), B * C2, H * W, _lib.stream_ptr()), "mrx_tl_pairs_to_f32")
    return out


def _tl_pack(w_ih, w_fin):
    """mrx_tl_pack of an IndRNN layer's 1x1 weight (forward order, transposed order) and -- for the last layer -- the final convolution's weights,
    cached per (storage, version) like the other packs."""
    key = (w_ih.data_ptr(), w_ih._version, str(w_ih.device), None if w_fin is None else (w_fin.data_ptr(), w_fin._version), "tl", _wcap())
    hit = _PACKS_BF16.get(key)
    if hit is None:
        if len(_PACKS_BF16) >= 256:
            _PACKS_BF16.pop(next(iter(_PACKS_BF16)))
        if tuple(w_ih.shape) != (64, 64, 1, 1) or (w_fin is not None and tuple(w_fin.shape) != (2, 64, 3, 3)):
            raise NotImplementedError("training layer kernels: 64 features, 1x1 IndRNN, final 3x3 convolution into 2 channels")
        L = _lib.lib()
        packed = torch.empty(int(L.mrx_tl_pack_bytes()), dtype=torch.uint8, device=w_ih.device)
        wf = None if w_fin is None else _lib.f32c(w_fin.detach())
        _lib.check(L.mrx_tl_pack(_lib.ptr(_lib.f32c(w_ih.detach())), _lib.ptr(wf), _lib.ptr(packed), _lib.stream_ptr()), "mrx_tl_pack")
        hit = _PACKS_BF16[key] = (packed, w_ih.detach(), None if w_fin is None else w_fin.detach())
    return hit[0]


def tl_layer_supported(Cin, Cout, k, dilation, rnn_features, rnn_k):
    return Cout == 64 and rnn_features == 64 and rnn_k == 1 and ((k == 5 and dilation == 1 and 1 <= Cin <= 8) or (k == 3 and dilation == 2 and Cin == 64))


def tl_layer_fwd(x, conv_w, conv_b, w_ih, b_ih, hh, h_prev, w_fin=None, want_mask=False):
    """One RIM layer in the training arithmetic (mrx_tl_layer_fwd): a = ReLU(bf16(conv_reppad(x) + b)) as a pair tensor, h = ReLU(bf16(W_ih a + b_ih)
    + hh * h_prev) fp32; with w_fin also the final convolution's 18 tap-product planes.  Returns (a_pairs, h, taps | None) -- with want_mask
    (a_pairs, h, taps | None, hmask): (h > 0) as int32 [B,H,W,2] bit words, what tl_cell_bwd needs of h.
    Hidden states are CHANNEL-BLOCKED: h, h_prev and the 64-channel layer's x are [B,8,H,W,8] (cb8_from_nchw / cb8_to_nchw convert)."""
    x = _lib.f32c(x)
    k, dil = int(conv_w.shape[-1]), (1 if int(conv_w.shape[-1]) == 5 else 2)
    if int(conv_w.shape[1]) == 64:
        if x.dim() != 5 or int(x.shape[1]) != 8 or int(x.shape[4]) != 8:
            raise ValueError(f"tl_layer_fwd: the 64-channel layer takes a channel-blocked [B,8,H,W,8] input, got {tuple(x.shape)}")
        B, Cin, H, W = int(x.shape[0]), 64, int(x.shape[2]), int(x.shape[3])
    else:
        B, Cin, H, W = _nchw(x)
    if h_prev is not None and tuple(h_prev.shape) != (B, 8, H, W, 8):
        raise ValueError(f"tl_layer_fwd: h_prev {tuple(h_prev.shape)}, expected the channel-blocked {(B, 8, H, W, 8)}")
    cp = _conv_bf16_pack(conv_w, False)
    tp = _tl_pack(w_ih, w_fin)
    a = torch.empty(B, 32, H, W, dtype=torch.int32, device=x.device)
    h = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=x.device)
    taps = torch.empty(B, 18, H, W, dtype=torch.float32, device=x.device) if w_fin is not None else None
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1)) if h_prev is not None else None
    cb = _lib.f32c(conv_b.detach()) if conv_b is not None else None
    ib = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hm = torch.empty(B, H, W, 2, dtype=torch.int32, device=x.device) if want_mask else None
    _lib.check(_lib.lib().mrx_tl_layer_fwd(_lib.ptr(x), _lib.ptr(cp), _lib.ptr(cb), _lib.ptr(tp), _lib.ptr(ib), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(a),
                                           _lib.ptr(h), _lib.ptr(hm), _lib.ptr(taps), B, Cin, H, W, k, dil, _lib.stream_ptr()), "mrx_tl_layer_fwd")
    return (a, h, taps, hm) if want_mask else (a, h, taps)


def tl_final_gather(taps, eta):
    B, H, W = int(eta.shape[0]), int(eta.shape[1]), int(eta.shape[2])
    out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_tl_final_gather(_lib.ptr(taps), _lib.ptr(eta), _lib.ptr(out), B, H, W, _lib.stream_ptr()), "mrx_tl_final_gather")
    return out


def tl_cell_part(B, H, W, device):
    return torch.empty(int(_lib.lib().mrx_tl_cell_part_floats(B, H, W)), dtype=torch.float32, device=device)


def tl_cell_bwd(dh_pairs, dH, h, h_prev, a_pairs, w_ih, w_fin, hh, part, first):
    """mrx_tl_cell_bwd: (dh_prev | None, ga_pairs); the parameter-gradient partials accumulate in `part` (tl_cell_part) until tl_cell_reduce.
    h, h_prev, dH in and dh_prev out are CHANNEL-BLOCKED [B,8,H,W,8] (c = 8 q + j).  `h` may instead be the forward's mask words (int32 [B,H,W,2],
    tl_layer_fwd(want_mask=True)): the kernel needs the state as (h > 0) only and then reads 8 bytes per pixel instead of 256."""
    hmask = None
    if h.dtype == torch.int32:
        if h.dim() != 4 or int(h.shape[3]) != 2:
            raise ValueError(f"tl_cell_bwd: mask words {tuple(h.shape)}, expected [B,H,W,2]")
        hmask, B, H, W = h, int(h.shape[0]), int(h.shape[1]), int(h.shape[2])
        h = None
    else:
        if h.dim() != 5 or int(h.shape[1]) != 8 or int(h.shape[4]) != 8:
            raise ValueError(f"tl_cell_bwd: h {tuple(h.shape)}, expected channel-blocked [B,8,H,W,8]")
        B, H, W = int(h.shape[0]), int(h.shape[2]), int(h.shape[3])
    for nm, t in (("dH", dH), ("h_prev", h_prev)):
        if t is not None and tuple(t.shape) != (B, 8, H, W, 8):
            raise ValueError(f"tl_cell_bwd: {nm} {tuple(t.shape)}, expected the channel-blocked {(B, 8, H, W, 8)}")
    tp = _tl_pack(w_ih, w_fin)
    ga = torch.empty(B, 32, H, W, dtype=torch.int32, device=a_pairs.device)
    dhp = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=a_pairs.device) if h_prev is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1)) if h_prev is not None else None
    _lib.check(_lib.lib().mrx_tl_cell_bwd(_lib.ptr(dh_pairs), _lib.ptr(dH), _lib.ptr(h), _lib.ptr(hmask), _lib.ptr(h_prev), _lib.ptr(a_pairs), _lib.ptr(tp), _lib.ptr(hhc),
                                          _lib.ptr(dhp), _lib.ptr(ga), _lib.ptr(part), int(bool(first)), B, H, W, _lib.stream_ptr()), "mrx_tl_cell_bwd")
    return dhp, ga


def tl_cell_reduce(part, B, H, W, dw_ih, db_ih, dhh, db_conv):
    _lib.check(_lib.lib().mrx_tl_cell_reduce(_lib.ptr(part), B, H, W, _lib.ptr(dw_ih), _lib.ptr(db_ih), _lib.ptr(dhh), _lib.ptr(db_conv), _lib.stream_ptr()),
               "mrx_tl_cell_reduce")


def tl_dgrad(dy, weight, dilation, dx_pairs, weights_in_lds=True):
    """Data gradient of a replicate-padded convolution with bf16 results (mrx_tl_dgrad + mrx_tl_fold_edges): dy fp32 [B,Cout,H,W] or a pair tensor
    (int32 [B,Cout/2,H,W]); returns a pair tensor (dx_pairs) or fp32 holding bf16 values."""
    pairs_in = dy.dtype == torch.int32
    B, Cd, H, W = [int(v) for v in dy.shape]
    cout_w, cin_w, k = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
    if (2 * Cd if pairs_in else Cd) != cout_w:
        raise RuntimeError(f"tl_dgrad: gradient of {Cd} planes for a weight {tuple(weight.shape)}")
    if not pairs_in:
        dy = _lib.f32c(dy)
    pad = int(dilation) * (k - 1) // 2
    packed = _conv_bf16_pack(weight, True)
    dx = torch.empty(B, cin_w // 2 if dx_pairs else cin_w, H, W, dtype=torch.int32 if dx_pairs else torch.float32, device=dy.device)
    frame = torch.empty(B, cin_w, H + 2 * pad, W + 2 * pad, dtype=torch.float32, device=dy.device)
    L = _lib.lib()
    fn = L.mrx_tl_dgrad if weights_in_lds else L.mrx_tl_dgrad_l2w      # (False: every shape on the generic kernel -- the test's reference form)
    _lib.check(fn(_lib.ptr(dy), int(pairs_in), _lib.ptr(packed), _lib.ptr(dx), int(bool(dx_pairs)), _lib.ptr(frame), B, cout_w, cin_w, H, W, k,
                  int(dilation), _lib.stream_ptr()), "mrx_tl_dgrad")
    _lib.check(L.mrx_tl_fold_edges(_lib.ptr(frame), _lib.ptr(dx), int(bool(dx_pairs)), B, cin_w, H, W, pad, _lib.stream_ptr()), "mrx_tl_fold_edges")
    return dx


def conv_wgrad_bf16_pairs(x, dy_pairs, k, dilation, pad_mode=PAD_REPLICATE, out=None, accumulate=False):
    """Weight gradient with the output gradient given as a pair tensor (mrx_conv_wgrad_bf16_pairs): 3x3 dilation 2 64 -> 64 (x [B,64,H,W] or the
    channel-blocked [B,8,H,W,8] of the training tape), 5x5 Cin <= 32 -> 64."""
    x = _lib.f32c(x)
    blocked = x.dim() == 5
    if blocked:
        if int(x.shape[1]) != 8 or int(x.shape[4]) != 8 or int(k) != 3:
            raise ValueError(f"conv_wgrad_bf16_pairs: channel-blocked x {tuple(x.shape)} (k={k}): [B,8,H,W,8] for the 3x3 layer only")
        B, Cin, H, W = int(x.shape[0]), 64, int(x.shape[2]), int(x.shape[3])
    else:
        B, Cin, H, W = _nchw(x)
    if tuple(dy_pairs.shape) != (B, 32, H, W) or dy_pairs.dtype != torch.int32:
        raise ValueError(f"conv_wgrad_bf16_pairs: x {tuple(x.shape)}, dy {tuple(dy_pairs.shape)}")
    if out is None:
        out = torch.empty(64, Cin, k, k, dtype=torch.float32, device=x.device)
        accumulate = False
    L = _lib.lib()
    if int(k) == 5 and int(dilation) == 1 and Cin <= 5 and pad_mode == PAD_REPLICATE and TL_WGRAD_IN:
        work = torch.empty(int(L.mrx_tl_wgrad_in_work_floats(B, Cin, H, W)), dtype=torch.float32, device=x.device)
        _lib.check(L.mrx_tl_wgrad_in(_lib.ptr(x), _lib.ptr(dy_pairs), _lib.ptr(out), _lib.ptr(work), B, Cin, H, W, int(bool(accumulate)), _lib.stream_ptr()),
                   "mrx_tl_wgrad_in")
        return out
    work = torch.empty(int(L.mrx_conv_wgrad_bf16_any_work_floats(B, Cin, 64, H, W, int(k))), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv_wgrad_bf16_pairs(_lib.ptr(x), _lib.ptr(dy_pairs), _lib.ptr(out), _lib.ptr(work), B, Cin, H, W, int(k), int(dilation), int(pad_mode),
                                           int(bool(accumulate)), int(blocked), _lib.stream_ptr()), "mrx_conv_wgrad_bf16_pairs")
    return out


def conv_wgrad_bf16_xcb(x_cb8, dy, pad_mode=PAD_REPLICATE, out=None, accumulate=False):
    """Weight gradient of the final 3x3 convolution 64 -> Cout <= 32 with x channel-blocked [B,8,H,W,8] (mrx_conv_wgrad_bf16_xcb)."""
    x_cb8, dy = _lib.f32c(x_cb8), _lib.f32c(dy)
    B, Cout, H, W = _nchw(dy)
    if tuple(x_cb8.shape) != (B, 8, H, W, 8):
        raise ValueError(f"conv_wgrad_bf16_xcb: x {tuple(x_cb8.shape)}, dy {tuple(dy.shape)}")
    if out is None:
        out = torch.empty(Cout, 64, 3, 3, dtype=torch.float32, device=dy.device)
        accumulate = False
    L = _lib.lib()
    work = torch.empty(int(L.mrx_conv_wgrad_bf16_any_work_floats(B, 64, Cout, H, W, 3)), dtype=torch.float32, device=dy.device)
    _lib.check(L.mrx_conv_wgrad_bf16_xcb(_lib.ptr(x_cb8), _lib.ptr(dy), _lib.ptr(out), _lib.ptr(work), B, Cout, H, W, int(pad_mode), int(bool(accumulate)),
                                         _lib.stream_ptr()), "mrx_conv_wgrad_bf16_xcb")
    return out


def conv_to_complex(x, weight, bias, dilation=1, pad_mode=PAD_ZERO):
    """permute(conv(x), (0, 2, 3, 1)) for a convolution into 2 channels -> [B,H,W,2] (one complex image).  The tuned kernel covers
    3x3, dilation 1, W % 4 == 0, Cin % 4 == 0; other shapes run conv2d and permute."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    Cout, Cin_w, kh, kw = [int(v) for v in weight.shape]
    if Cout != 2 or Cin_w != Cin:
        raise ValueError(f"conv_to_complex: weight {tuple(weight.shape)} for input {tuple(x.shape)}")
    if not (kh == 3 and kw == 3 and int(dilation) == 1 and W % 4 == 0 and W >= 8 and Cin % 4 == 0 and x.data_ptr() % 16 == 0):
        return conv2d(x, weight, bias, dilation, pad_mode).permute(0, 2, 3, 1).contiguous()
    w = _lib.f32c(weight.detach())
    b = _lib.f32c(bias.detach()) if bias is not None else None
    out = torch.empty(B, H, W, 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_conv_to_complex(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Cin, H, W, 3, 1, int(pad_mode),
                                              _lib.stream_ptr()), "mrx_conv_to_complex")
    return out


def indrnn_cell(x, w_ih, b_ih, hh, h_prev, dilation=1, out=None):
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    F, Cin_w, k, _ = [int(v) for v in w_ih.shape]
    if Cin_w != Cin:
        raise RuntimeError(f"input has inconsistent input_size: got {Cin}, expected {Cin_w}")
    if k == 1 and conv1x1_sq_supported(Cin, F) and out is None:
        return conv1x1_64(x, w_ih, b_ih, ACT_RELU, 0.0, hh, h_prev)      # ReLU(W x + b + hh * h_prev) in one HBM-bound launch
    w_ih = _lib.f32c(w_ih.detach())
    b = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, F, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_indrnn_cell(_lib.ptr(x), _lib.ptr(w_ih), _lib.ptr(b), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out),
                                          B, Cin, F, H, W, k, int(dilation), _lib.stream_ptr()), "mrx_indrnn_cell")
    return out


def rim_layer_indrnn(x, w_conv, b_conv, k, dilation, w_ih, b_ih, hh, h_prev, out=None):
    """Fused ConvNonlinear(ReLU, replicate pad) + IndRNNCell(1x1)."""
    x, w_conv, w_ih = _lib.f32c(x), _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    B, Cin, H, W = _nchw(x)
    F = int(w_conv.shape[0])
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, F, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer_indrnn(_lib.ptr(x), _lib.ptr(w_conv), _lib.ptr(bc), _lib.ptr(w_ih), _lib.ptr(bi),
                                               _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), B, Cin, F, H, W, int(k),
                                               int(dilation), _lib.stream_ptr()), "mrx_rim_layer_indrnn")
    return out


def rim_layer_supported(Cin, F, k, dilation):
    return bool(_lib.lib().mrx_rim_layer_supported(int(Cin), int(F), int(k), int(dilation)))


def rim_layer_pack(w_conv, w_ih):
    """Pack conv [F,Cin,k,k] + ih [F,F,1,1] weights into the MFMA operand order of the tuned fused layer."""
    w_conv, w_ih = _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    F, Cin, k, _ = [int(v) for v in w_conv.shape]
    n = int(_lib.lib().mrx_rim_layer_pack_floats(Cin, F, k))
    if n < 0:
        raise ValueError("rim_layer_pack: unsupported shape")
    packed = torch.empty(n, dtype=torch.float32, device=w_conv.device)
    _lib.check(_lib.lib().mrx_rim_layer_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(packed), Cin, F, k, _lib.stream_ptr()),
               "mrx_rim_layer_pack")
    return packed


def rim_layer1_xmax_supported(Cin, F, k, dilation):
    """The first-layer kernel can keep the bound of its outputs that rim_layer2_f16 scales by (mrx_rim_layer1_xmax_supported)."""
    return bool(_lib.lib().mrx_rim_layer1_xmax_supported(int(Cin), int(F), int(k), int(dilation)))


def rim_layer_indrnn_packed(x, packed, F, k, dilation, b_conv, b_ih, hh, h_prev, out=None, xmax=None):
    """Tuned fused ConvNonlinear(ReLU, replicate pad) + IndRNNCell(1x1) on pre-packed weights.  `xmax` (one-element float32 device tensor):
    the maximum of the outputs is folded into it with an atomic max (mrx_rim_layer_indrnn_packed_xmax)."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, F, H, W, dtype=torch.float32, device=x.device)
    if xmax is not None:
        _lib.check(_lib.lib().mrx_rim_layer_indrnn_packed_xmax(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc),
                                                               _lib.ptr(hp), _lib.ptr(out), _lib.ptr(xmax), B, Cin, int(F), H, W, int(k),
                                                               int(dilation), _lib.stream_ptr()), "mrx_rim_layer_indrnn_packed_xmax")
        return out
    _lib.check(_lib.lib().mrx_rim_layer_indrnn_packed(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc),
                                                      _lib.ptr(hp), _lib.ptr(out), B, Cin, int(F), H, W, int(k), int(dilation),
                                                      _lib.stream_ptr()), "mrx_rim_layer_indrnn_packed")
    return out


def rim_layer_wino_supported(Cin, F, k, dilation):
    """Winograd F(2x2,3x3) fused layer: dilation-2 3x3 convolution into 64 features."""
    return int(k) == 3 and int(dilation) == 2 and int(F) == 64 and int(Cin) >= 1


def rim_layer2_sb_pack(w_conv, w_ih, w_final=None):
    """Split-bf16 operand pack of the second RIM layer (w_conv [64,64,3,3], w_ih [64,64,1,1]) for rim_layer2_sb; with w_final [2,64,3,3]
    also the operands of the final convolution's channel contraction (rim_layer2_sb_final)."""
    w_conv, w_ih = _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    if tuple(w_conv.shape) != (64, 64, 3, 3) or tuple(w_ih.shape) != (64, 64, 1, 1):
        raise NotImplementedError(f"rim_layer2_sb_pack: {tuple(w_conv.shape)} / {tuple(w_ih.shape)}")
    if w_final is not None:
        w_final = _lib.f32c(w_final.detach())
        if tuple(w_final.shape) != (2, 64, 3, 3):
            raise NotImplementedError(f"rim_layer2_sb_pack: final conv {tuple(w_final.shape)}")
    packed = torch.empty(int(_lib.lib().mrx_rim_layer2_sb_pack_floats()), dtype=torch.float32, device=w_conv.device)
    _lib.check(_lib.lib().mrx_rim_layer2_sb_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(w_final), _lib.ptr(packed), _lib.stream_ptr()),
               "mrx_rim_layer2_sb_pack")
    return packed


def rim_layer2_sb(x, packed, b_conv, b_ih, hh, h_prev, out=None):
    """ReLU(W_ih ReLU(conv3x3 dilation 2 (replicate pad)(x) + b_conv) + b_ih + hh * h_prev), 64 features, on the bf16 matrix pipe with fp32
    results (mrx_rim_layer2_sb).  `out` may be h_prev itself (state updated in place: every element is read by the lane that writes it)."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, 64, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer2_sb(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out),
                                            B, H, W, _lib.stream_ptr()), "mrx_rim_layer2_sb")
    return out


def rim_layer2_sb_taps(x, packed, b_conv, b_ih, hh, h_prev, taps=None, out=None):
    """rim_layer2_sb that also leaves the final convolution's per-pixel tap products [B,18,H,W] (mrx_rim_layer2_sb_taps); `packed` from
    rim_layer2_sb_pack(..., w_final).  `out` may be h_prev itself (state updated in place).  Returns (h_new, taps)."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if taps is None or taps.numel() < 18 * B * H * W:
        taps = torch.empty(B, 18, H, W, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(B, 64, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer2_sb_taps(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp),
                                                 _lib.ptr(out), _lib.ptr(taps), B, H, W, _lib.stream_ptr()), "mrx_rim_layer2_sb_taps")
    return out, taps


def rim_layer2_f16_pack(w_conv, w_ih, w_final=None):
    """Operand pack of the second RIM layer with the convolution's weights as two fp16 terms scaled by a power of two (mrx_rim_layer2_f16_pack;
    the 1x1 and final-conv operands as in rim_layer2_sb_pack)."""
    w_conv = _lib.f32c(w_conv.detach())
    w_ih = _lib.f32c(w_ih.detach()) if w_ih is not None else None      # (None: the convolution stage alone, mrx_conv3x3_sb_chain)
    if tuple(w_conv.shape) != (64, 64, 3, 3) or (w_ih is not None and tuple(w_ih.shape) != (64, 64, 1, 1)):
        raise NotImplementedError(f"rim_layer2_f16_pack: {tuple(w_conv.shape)} / {None if w_ih is None else tuple(w_ih.shape)}")
    if w_final is not None:
        w_final = _lib.f32c(w_final.detach())
        if tuple(w_final.shape) != (2, 64, 3, 3):
            raise NotImplementedError(f"rim_layer2_f16_pack: final conv {tuple(w_final.shape)}")
    packed = torch.empty(int(_lib.lib().mrx_rim_layer2_f16_pack_floats()), dtype=torch.float32, device=w_conv.device)
    _lib.check(_lib.lib().mrx_rim_layer2_f16_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(w_final), _lib.ptr(packed), _lib.stream_ptr()),
               "mrx_rim_layer2_f16_pack")
    return packed


def rim_layer2_f16(x, packed, b_conv, b_ih, hh, h_prev, xmax, taps=None, out=None, want_taps=False):
    """rim_layer2_sb / rim_layer2_sb_taps with the convolution's operands as two fp16 terms (mrx_rim_layer2_f16: half the MFMAs).  `xmax`: a
    one-element float32 device tensor holding an upper bound of max |x| (kept by the producer of x: rim_layer_indrnn_packed*(xmax=...)).
    Returns h_new, or (h_new, taps) with want_taps."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if want_taps and (taps is None or taps.numel() < 18 * B * H * W):
        taps = torch.empty(B, 18, H, W, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(B, 64, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer2_f16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out),
                                             _lib.ptr(taps) if want_taps else None, _lib.ptr(xmax), B, H, W, _lib.stream_ptr()),
               "mrx_rim_layer2_f16")
    return (out, taps) if want_taps else out


def cb8_from_nchw(x):
    """[B,C,H,W] -> the channel-blocked layout [B,C/8,H,W,8] of the cb8 RIM layer kernels (mrx_cb8_convert)."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    y = torch.empty(B, C // 8, H, W, 8, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_cb8_convert(_lib.ptr(x), _lib.ptr(y), B, C, H, W, 1, _lib.stream_ptr()), "mrx_cb8_convert")
    return y


def cb8_to_nchw(y):
    """The inverse of cb8_from_nchw."""
    y = _lib.f32c(y)
    B, Q, H, W, _ = [int(v) for v in y.shape]
    x = torch.empty(B, Q * 8, H, W, dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().mrx_cb8_convert(_lib.ptr(y), _lib.ptr(x), B, Q * 8, H, W, 0, _lib.stream_ptr()), "mrx_cb8_convert")
    return x


def rim_layer1_cb8(x, eta, part, nparts, sigma, packed, b_conv, b_ih, hh, h_prev, xmax, out=None):
    """First RIM layer on channel-blocked states (mrx_rim_layer1_cb8): input x [B,Cin<=4,H,W] (eta None) or (eta [B,H,W,2], coil-group partial
    sums) as rim_layer_indrnn_packed_llg; h_prev / result [B,8,H,W,8]; keeps the running bound `xmax` of its outputs."""
    if eta is not None:
        eta = _lib.f32c(eta)
        B, H, W, _ = [int(v) for v in eta.shape]
        Cin = 4
    else:
        x = _lib.f32c(x)
        B, Cin, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=(eta if eta is not None else x).device)
    _lib.check(_lib.lib().mrx_rim_layer1_cb8(_lib.ptr(x) if eta is None else None, int(Cin), _lib.ptr(eta), _lib.ptr(part), int(nparts),
                                             float(1.0 / (float(sigma) ** 2.0)), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc),
                                             _lib.ptr(hp), _lib.ptr(out), _lib.ptr(xmax), B, H, W, _lib.stream_ptr()), "mrx_rim_layer1_cb8")
    return out


def rim_layer2_f16_cb8(x, packed, b_conv, b_ih, hh, h_prev, xmax, taps=None, out=None, want_taps=False):
    """rim_layer2_f16 on channel-blocked tensors (mrx_rim_layer2_f16_cb8; x, h_prev, result [B,8,H,W,8]; taps [B,18,H,W] as rim_layer2_f16)."""
    x = _lib.f32c(x)
    B, Q, H, W, E = [int(v) for v in x.shape]
    if Q != 8 or E != 8:
        raise ValueError(f"rim_layer2_f16_cb8 expects x [B,8,H,W,8], got {tuple(x.shape)}")
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if want_taps and (taps is None or taps.numel() < 18 * B * H * W):
        taps = torch.empty(B, 18, H, W, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer2_f16_cb8(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out),
                                                 _lib.ptr(taps) if want_taps else None, _lib.ptr(xmax), B, H, W, _lib.stream_ptr()),
               "mrx_rim_layer2_f16_cb8")
    return (out, taps) if want_taps else out


# the fused RIM loop at W = 372: layer 2 leaves the final convolution's tap products pre-summed along x (env MRIDC_AMD_RIM_TAPS_Q=0: the 18-plane form)
RIM_TAPS_Q = os.environ.get("MRIDC_AMD_RIM_TAPS_Q", "1") != "0"


def rim_layer2_f16_cb8_q(x, packed, b_conv, b_ih, hh, h_prev, xmax, taps_q=None, edges=None, out=None):
    """rim_layer2_f16_cb8 with the final convolution's tap products pre-summed along x (mrx_rim_layer2_f16_cb8_q): returns (h_new [B,8,H,W,8], taps_q [B,3,H,W,2],
    edges [mrx_rim_taps_q_edge_floats]) -- for rim_final_gather_q / llg372_gather_q."""
    x = _lib.f32c(x)
    B, Q, H, W, E = [int(v) for v in x.shape]
    if Q != 8 or E != 8:
        raise ValueError(f"rim_layer2_f16_cb8_q expects x [B,8,H,W,8], got {tuple(x.shape)}")
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    L = _lib.lib()
    if taps_q is None or taps_q.numel() < 6 * B * H * W:
        taps_q = torch.empty(B, 3, H, W, 2, dtype=torch.float32, device=x.device)
    ne = int(L.mrx_rim_taps_q_edge_floats(B, H, W))
    if edges is None or edges.numel() < ne:
        edges = torch.empty(ne, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_rim_layer2_f16_cb8_q(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), _lib.ptr(taps_q),
                                          _lib.ptr(edges), _lib.ptr(xmax), B, H, W, _lib.stream_ptr()), "mrx_rim_layer2_f16_cb8_q")
    return out, taps_q, edges


def _check_edges(edges, B, H, W, device, who):
    """The gather kernels load `edges` unconditionally for every pixel and the C entry points receive no size for it: check it here."""
    ne = int(_lib.lib().mrx_rim_taps_q_edge_floats(B, H, W))
    if edges is None or edges.dtype != torch.float32 or edges.device != device or not edges.is_contiguous() or edges.numel() < ne:
        raise ValueError(f"{who}: edges must be a contiguous fp32 device tensor of >= {ne} elements (mrx_rim_taps_q_edge_floats({B}, {H}, {W}))")


def rim_final_gather_q(taps_q, edges, b_final, eta):
    """eta + permute(conv3x3_reppad(h) + b_final) [B,H,W,2] from the row-pre-summed tap planes of rim_layer2_f16_cb8_q (mrx_rim_final_gather_q)."""
    eta = _lib.f32c(eta)
    B, H, W, _ = [int(v) for v in eta.shape]
    if taps_q.numel() < 6 * B * H * W or int(eta.shape[-1]) != 2:
        raise ValueError("rim_final_gather_q expects taps_q [B,3,H,W,2] and eta [B,H,W,2]")
    _check_edges(edges, B, H, W, eta.device, "rim_final_gather_q")
    bf = _lib.f32c(b_final.detach()) if b_final is not None else None
    eta_out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_rim_final_gather_q(_lib.ptr(taps_q), _lib.ptr(edges), _lib.ptr(bf), _lib.ptr(eta), _lib.ptr(eta_out), B, H, W, _lib.stream_ptr()),
               "mrx_rim_final_gather_q")
    return eta_out


# ---- the reduced-precision inference route (csrc/rim_amp16.hip): the reference's `precision: 16` -----------------------------------------------------
def amp16_from_nchw(h):
    """[B,64,H,W] (any float dtype) -> the fp16 channel-blocked state [B,4,H,W,16] (h[b][c // 16][y][x][c % 16]) of the amp16 layer kernels (entry / exit of a block
    only: torch plumbing)."""
    _lib.require_gpu(h)
    B, C, H, W = _nchw(h)
    return h.reshape(B, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).to(torch.float16).contiguous()


def amp16_to_nchw(h):
    """The inverse of amp16_from_nchw, as fp32 [B,64,H,W] (rim_block.py hands fp32 states from call to call)."""
    B, Q, H, W, E = [int(v) for v in h.shape]
    return h.permute(0, 1, 4, 2, 3).reshape(B, Q * E, H, W).float().contiguous()


def amp16_layer1_pack(w_conv, w_ih):
    w_conv, w_ih = _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    if tuple(w_conv.shape[0:1] + w_conv.shape[2:]) != (64, 5, 5) or int(w_conv.shape[1]) > 4 or tuple(w_ih.shape) != (64, 64, 1, 1):
        raise NotImplementedError(f"amp16_layer1_pack: {tuple(w_conv.shape)} / {tuple(w_ih.shape)}")
    L = _lib.lib()
    packed = torch.empty(int(L.mrx_amp16_pack_floats(1)), dtype=torch.float32, device=w_conv.device)
    _lib.check(L.mrx_amp16_layer1_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(packed), int(w_conv.shape[1]), _lib.stream_ptr()), "mrx_amp16_layer1_pack")
    return packed


def amp16_layer2_pack(w_conv, w_ih, w_final=None):
    w_conv, w_ih = _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    if tuple(w_conv.shape) != (64, 64, 3, 3) or tuple(w_ih.shape) != (64, 64, 1, 1):
        raise NotImplementedError(f"amp16_layer2_pack: {tuple(w_conv.shape)} / {tuple(w_ih.shape)}")
    if w_final is not None:
        w_final = _lib.f32c(w_final.detach())
        if tuple(w_final.shape) != (2, 64, 3, 3):
            raise NotImplementedError(f"amp16_layer2_pack: final conv {tuple(w_final.shape)}")
    L = _lib.lib()
    packed = torch.empty(int(L.mrx_amp16_pack_floats(2)), dtype=torch.float32, device=w_conv.device)
    _lib.check(L.mrx_amp16_layer2_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(w_final), _lib.ptr(packed), _lib.stream_ptr()), "mrx_amp16_layer2_pack")
    return packed


def _amp16_state(h, B, H, W, what):
    if h is None:
        return None
    _lib.require_gpu(h)
    if h.dtype != torch.float16 or tuple(h.shape) != (B, 4, H, W, 16) or not h.is_contiguous():
        raise ValueError(f"{what}: states are contiguous fp16 [B,4,H,W,16] = {(B, 4, H, W, 16)}, got {h.dtype} {tuple(h.shape)}")
    return h


def amp16_layer1(x, eta, part, nparts, sigma, packed, b_conv, b_ih, hh, h_prev, out=None):
    """First RIM layer of the precision-16 route (mrx_amp16_layer1): input x [B,Cin<=4,H,W] (eta None) or (eta [B,H,W,2], nparts <= 4 coil-group partial
    planes) as rim_layer1_cb8; h_prev / result fp16 [B,4,H,W,16]."""
    if eta is not None:
        eta = _lib.f32c(eta)
        B, H, W, _ = [int(v) for v in eta.shape]
        Cin = 4
        if not 1 <= int(nparts) <= 4 or part is None or part.numel() < int(nparts) * B * H * W * 2 or part.dtype != torch.float32:
            raise ValueError("amp16_layer1: (eta, part) needs 1 .. 4 fp32 partial planes [nparts,B,H,W,2]")
    else:
        x = _lib.f32c(x)
        B, Cin, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _amp16_state(h_prev, B, H, W, "amp16_layer1")
    if out is None:
        out = torch.empty(B, 4, H, W, 16, dtype=torch.float16, device=(eta if eta is not None else x).device)
    _amp16_state(out, B, H, W, "amp16_layer1")
    _lib.check(_lib.lib().mrx_amp16_layer1(_lib.ptr(x) if eta is None else None, int(Cin), _lib.ptr(eta), _lib.ptr(part) if eta is not None else None, int(nparts),
                                           float(1.0 / (float(sigma) ** 2.0)), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc),
                                           _lib.ptr(hp), _lib.ptr(out), B, H, W, _lib.stream_ptr()), "mrx_amp16_layer1")
    return out


def amp16_layer2(x, packed, b_conv, b_ih, hh, h_prev, taps_q=None, edges=None, out=None, want_taps=True):
    """Second RIM layer of the precision-16 route (mrx_amp16_layer2): fp16 x / h_prev / result [B,4,H,W,16]; with `want_taps` also (taps_q [B,3,H,W,2],
    edges) of the final convolution, fp32, in the layout of rim_layer2_f16_cb8_q (for rim_final_gather_q / llg372_gather_q)."""
    _lib.require_gpu(x)
    B, Q, H, W, E = [int(v) for v in x.shape]
    _amp16_state(x, B, H, W, "amp16_layer2")
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _amp16_state(h_prev, B, H, W, "amp16_layer2")
    L = _lib.lib()
    if want_taps:
        if taps_q is None or taps_q.numel() < 6 * B * H * W:
            taps_q = torch.empty(B, 3, H, W, 2, dtype=torch.float32, device=x.device)
        ne = int(L.mrx_rim_taps_q_edge_floats(B, H, W))
        if edges is None or edges.numel() < ne:
            edges = torch.empty(ne, dtype=torch.float32, device=x.device)
    else:
        taps_q = edges = None
    if out is None:
        out = torch.empty(B, 4, H, W, 16, dtype=torch.float16, device=x.device)
    _amp16_state(out, B, H, W, "amp16_layer2")
    _lib.check(L.mrx_amp16_layer2(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out), _lib.ptr(taps_q),
                                  _lib.ptr(edges), B, H, W, _lib.stream_ptr()), "mrx_amp16_layer2")
    return (out, taps_q, edges) if want_taps else out


def rim_final_gather(taps, b_final, eta):
    """eta + permute(conv3x3_reppad(h) + b_final) [B,H,W,2] from the tap products of rim_layer2_sb_taps (mrx_rim_final_gather)."""
    eta = _lib.f32c(eta)
    B, H, W, _ = [int(v) for v in eta.shape]
    if taps.numel() < 18 * B * H * W or int(eta.shape[-1]) != 2:
        raise ValueError("rim_final_gather expects taps [B,18,H,W] and eta [B,H,W,2]")
    bf = _lib.f32c(b_final.detach()) if b_final is not None else None
    eta_out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_rim_final_gather(_lib.ptr(taps), _lib.ptr(bf), _lib.ptr(eta), _lib.ptr(eta_out), B, H, W, _lib.stream_ptr()),
               "mrx_rim_final_gather")
    return eta_out


def rim_layer2_sb_final(x, packed, b_conv, b_ih, hh, h_prev, b_final, eta, work=None, out=None):
    """Second RIM layer and the final convolution + eta update (rim_block.py:233-246): returns (h_new [B,64,H,W],
    eta + permute(conv3x3_reppad(h_new) + b_final) [B,H,W,2]).  `packed` from rim_layer2_sb_pack(..., w_final)."""
    if tuple(eta.shape) != (int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), 2):
        raise ValueError("rim_layer2_sb_final expects eta of shape [B,H,W,2]")
    h_new, taps = rim_layer2_sb_taps(x, packed, b_conv, b_ih, hh, h_prev, work, out)
    return h_new, rim_final_gather(taps, b_final, eta)


def rim_layer_wino_pack(w_conv, w_ih):
    """Transform conv [64,Cin,3,3] weights to G g G^T and pack them with the ih [64,64,1,1] weights for mrx_rim_layer_indrnn_wino."""
    w_conv, w_ih = _lib.f32c(w_conv.detach()), _lib.f32c(w_ih.detach())
    F, Cin, k, _ = [int(v) for v in w_conv.shape]
    n = int(_lib.lib().mrx_rim_layer_wino_pack_floats(Cin, F)) if k == 3 else -1
    if n < 0:
        raise ValueError("rim_layer_wino_pack: unsupported shape")
    packed = torch.empty(n, dtype=torch.float32, device=w_conv.device)
    _lib.check(_lib.lib().mrx_rim_layer_wino_pack(_lib.ptr(w_conv), _lib.ptr(w_ih), _lib.ptr(packed), Cin, F, _lib.stream_ptr()),
               "mrx_rim_layer_wino_pack")
    return packed


def rim_layer_indrnn_wino(x, packed, F, b_conv, b_ih, hh, h_prev, out=None):
    """Winograd fused ConvNonlinear(3x3, dilation 2, ReLU, replicate pad) + IndRNNCell(1x1) on pre-transformed weights."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if out is None:
        out = torch.empty(B, F, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer_indrnn_wino(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc),
                                                    _lib.ptr(hp), _lib.ptr(out), B, Cin, int(F), H, W, _lib.stream_ptr()),
               "mrx_rim_layer_indrnn_wino")
    return out


# ---- training path: backward pieces of the convolutional regulariser ---------------------------------------------------------
def conv_wgrad(x, dy, k, dilation=1, pad_mode=PAD_REPLICATE, out=None, accumulate=False):
    """Weight gradient of a 'same' convolution: dw[Cout,Cin,k,k] = sum_{b,pixel} dy * pad(x) (mrx_conv_wgrad)."""
    x, dy = _lib.f32c(x), _lib.f32c(dy)
    B, Cin, H, W = _nchw(x)
    Cout = int(dy.shape[1])
    if tuple(dy.shape) != (B, Cout, H, W):
        raise ValueError(f"conv_wgrad: dy {tuple(dy.shape)} vs x {tuple(x.shape)}")
    if out is None:
        out = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=x.device)
        accumulate = False
    L = _lib.lib()
    work = torch.empty(int(L.mrx_conv_wgrad_work_floats(B, Cin, Cout, H, W, int(k))), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(out), _lib.ptr(work), B, Cin, Cout, H, W, int(k), int(dilation),
                                int(pad_mode), int(bool(accumulate)), _lib.stream_ptr()), "mrx_conv_wgrad")
    return out


def conv_dgrad(dy, weight, dilation=1, pad_mode=PAD_REPLICATE):
    """Data gradient of y = conv(pad(x), weight): a zero-padded 'same' convolution of dy with the flipped, transposed weights; for
    replicate padding on the domain extended by the padding, folded back onto the image (mrx_reppad_fold)."""
    dy = _lib.f32c(dy)
    B, Cout, H, W = _nchw(dy)
    k = int(weight.shape[-1])
    pad = int(dilation) * (k - 1) // 2
    wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()          # [Cin,Cout,k,k] (parameter-sized host-side prep)
    if pad_mode == PAD_ZERO or pad == 0:
        return conv2d(dy, wt, None, dilation, PAD_ZERO)
    big = pad2d(dy, pad, pad, pad, pad, 0)                                # dy zero-extended to [H+2p, W+2p]
    g = conv2d(big, wt, None, dilation, PAD_ZERO)
    Cin = int(wt.shape[0])
    out = torch.empty(B, Cin, H, W, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().mrx_reppad_fold(_lib.ptr(g), _lib.ptr(out), B * Cin, H, W, pad, _lib.stream_ptr()), "mrx_reppad_fold")
    return out


def relu_bwd(dy, y, h_prev=None, hh=None):
    """dpre = dy * (y > 0) and the per-channel sums [C,2] = (sum dpre, sum dpre * h_prev); with h_prev also dh_prev = dpre * hh."""
    dy, y = _lib.f32c(dy), _lib.f32c(y)
    B, C, H, W = _nchw(y)
    hp = None if h_prev is None else _lib.f32c(h_prev)
    hhc = None if hh is None else _lib.f32c(hh.detach().reshape(-1))
    dpre = torch.empty_like(y)
    dhp = torch.empty_like(y) if hp is not None else None
    sums = torch.empty(C, 2, dtype=torch.float32, device=y.device)
    L = _lib.lib()
    work = torch.empty(int(L.mrx_relu_bwd_work_floats(C)), dtype=torch.float32, device=y.device)
    _lib.check(L.mrx_relu_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(hp), _lib.ptr(hhc), _lib.ptr(dpre), _lib.ptr(dhp), _lib.ptr(sums),
                              _lib.ptr(work), B, C, H * W, _lib.stream_ptr()), "mrx_relu_bwd")
    return dpre, dhp, sums


def rim_final(h, weight, bias, k, dilation, eta):
    """eta + permute(conv_reppad(h)) with a 2-channel conv -> [B,H,W,2]."""
    h, weight, eta = _lib.f32c(h), _lib.f32c(weight.detach()), _lib.f32c(eta)
    B, F, H, W = _nchw(h)
    if int(weight.shape[0]) != 2 or tuple(eta.shape) != (B, H, W, 2):
        raise ValueError("rim_final expects a 2-channel final conv and eta of shape [B,H,W,2]")
    b = _lib.f32c(bias.detach()) if bias is not None else None
    out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_rim_final(_lib.ptr(h), _lib.ptr(weight), _lib.ptr(b), _lib.ptr(eta), _lib.ptr(out), B, F, H, W,
                                        int(k), int(dilation), _lib.stream_ptr()), "mrx_rim_final")
    return out


def gru_gates(ih, hh, h):
    ih, hh, h = _lib.f32c(ih), _lib.f32c(hh), _lib.f32c(h)
    B, F, H, W = _nchw(h)
    out = torch.empty_like(h)
    _lib.check(_lib.lib().mrx_gru_gates(_lib.ptr(ih), _lib.ptr(hh), _lib.ptr(h), _lib.ptr(out), B, F, H * W, _lib.stream_ptr()),
               "mrx_gru_gates")
    return out


def _gates_bwd(name, gates, dy, ih, hh, h):
    dy, ih, hh, h = _lib.f32c(dy), _lib.f32c(ih), _lib.f32c(hh), _lib.f32c(h)
    B, F, H, W = _nchw(h)
    if tuple(ih.shape) != (B, gates * F, H, W) or tuple(hh.shape) != tuple(ih.shape) or tuple(dy.shape) != tuple(h.shape):
        raise ValueError(f"{name}: dy {tuple(dy.shape)}, ih {tuple(ih.shape)}, hh {tuple(hh.shape)}, h {tuple(h.shape)}")
    dih, dhh, dh = torch.empty_like(ih), torch.empty_like(hh), torch.empty_like(h)
    _lib.check(getattr(_lib.lib(), name)(_lib.ptr(dy), _lib.ptr(ih), _lib.ptr(hh), _lib.ptr(h), _lib.ptr(dih), _lib.ptr(dhh), _lib.ptr(dh), B, F, H * W,
                                         _lib.stream_ptr()), name)
    return dih, dhh, dh


def gru_gates_bwd(dy, ih, hh, h):
    """(d ih, d hh, d h) of gru_gates (mrx_gru_gates_bwd)."""
    return _gates_bwd("mrx_gru_gates_bwd", 3, dy, ih, hh, h)


def mgu_gates_bwd(dy, ih, hh, h):
    """(d ih, d hh, d h) of mgu_gates (mrx_mgu_gates_bwd)."""
    return _gates_bwd("mrx_mgu_gates_bwd", 2, dy, ih, hh, h)


def gated_cell_supported(cin, F, k, gates):
    return bool(_lib.lib().mrx_gated_cell_supported(int(cin), int(F), int(k), int(gates)))


def gated_cell_pack(w_ih, w_hh, gates):
    """Pack the 1x1 `ih` [gates*F,Cin,1,1] and `hh` [gates*F,F,1,1] weights of a ConvGRUCell / ConvMGUCell for gated_cell_1x1."""
    w_ih, w_hh = _lib.f32c(w_ih.detach()), _lib.f32c(w_hh.detach())
    F, Cin = int(w_hh.shape[1]), int(w_ih.shape[1])
    n = int(_lib.lib().mrx_gated_cell_pack_floats(Cin, F, int(gates)))
    if n < 0 or w_ih.shape[0] != gates * F or w_hh.shape[0] != gates * F:
        raise ValueError("gated_cell_pack: unsupported shape")
    packed = torch.empty(n, dtype=torch.float32, device=w_ih.device)
    _lib.check(_lib.lib().mrx_gated_cell_pack(_lib.ptr(w_ih), _lib.ptr(w_hh), _lib.ptr(packed), Cin, F, int(gates), _lib.stream_ptr()),
               "mrx_gated_cell_pack")
    return packed


def gated_cell_1x1(x, h, packed, b_ih, gates, F=64):
    """Whole ConvGRUCell (gates=3) / ConvMGUCell (gates=2) with 1x1 kernels in one launch.  h None = zero state."""
    x = _lib.f32c(x)
    B, Cin, H, W = _nchw(x)
    h = None if h is None else _lib.f32c(h)
    b_ih = None if b_ih is None else _lib.f32c(b_ih.detach())
    out = torch.empty((B, F, H, W), dtype=torch.float32, device=x.device)
    if SB_CHAIN and _lib.arith() == "f16x2":
        # the next stack's 64-channel convolution reads this state: keep the bound of it (mrx_conv3x3_sb_chain then runs two-term fp16 operands)
        xmax = _zero_scalar(x.device)
        _lib.check(_lib.lib().mrx_gated_cell_1x1_xmax(_lib.ptr(x), _lib.ptr(h), _lib.ptr(packed), _lib.ptr(b_ih), _lib.ptr(out), _lib.ptr(xmax), B,
                                                      Cin, F, H * W, int(gates), _lib.stream_ptr()), "mrx_gated_cell_1x1_xmax")
        return _attach_bound(out, xmax)
    _lib.check(_lib.lib().mrx_gated_cell_1x1(_lib.ptr(x), _lib.ptr(h), _lib.ptr(packed), _lib.ptr(b_ih), _lib.ptr(out), B, Cin, F,
                                             H * W, int(gates), _lib.stream_ptr()), "mrx_gated_cell_1x1")
    return out


def conv2dgru_supported(cin, F, k):
    return bool(_lib.lib().mrx_conv2dgru_supported(int(cin), int(F), int(k)))


def conv2dgru_pack(w_update, w_reset, w_out):
    """Pack the three 1x1 gate weights [64,128,1,1] of a Conv2dGRU layer for conv2dgru_cell_1x1."""
    ws = [_lib.f32c(w.detach()) for w in (w_update, w_reset, w_out)]
    F = int(ws[0].shape[0])
    n = int(_lib.lib().mrx_conv2dgru_pack_floats(F))
    if n < 0 or any(tuple(w.shape) != (F, 2 * F, 1, 1) for w in ws):
        raise ValueError("conv2dgru_pack: unsupported shape")
    packed = torch.empty(n, dtype=torch.float32, device=ws[0].device)
    _lib.check(_lib.lib().mrx_conv2dgru_pack(_lib.ptr(ws[0]), _lib.ptr(ws[1]), _lib.ptr(ws[2]), _lib.ptr(packed), F, _lib.stream_ptr()),
               "mrx_conv2dgru_pack")
    return packed


def conv2dgru_cell_1x1(x, h, packed, bias, relu_out=True):
    """One Conv2dGRU layer's GRU (1x1 gates, 64 features) in one launch -> (new state, ReLU(new state) | None)."""
    x = _lib.f32c(x)
    B, F, H, W = _nchw(x)
    h = None if h is None else _lib.f32c(h)
    bias = None if bias is None else _lib.f32c(bias)
    out = torch.empty_like(x)
    out_relu = torch.empty_like(x) if relu_out else None
    if relu_out and SB_CHAIN and _lib.arith() == "f16x2":
        # ReLU(new state) is the next layer's convolution input: keep its bound (mrx_conv3x3_sb_chain then runs two-term fp16 operands)
        xmax = _zero_scalar(x.device)
        _lib.check(_lib.lib().mrx_conv2dgru_cell_1x1_xmax(_lib.ptr(x), _lib.ptr(h), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(out),
                                                          _lib.ptr(out_relu), _lib.ptr(xmax), B, F, H * W, _lib.stream_ptr()),
                   "mrx_conv2dgru_cell_1x1_xmax")
        return out, _attach_bound(out_relu, xmax)
    _lib.check(_lib.lib().mrx_conv2dgru_cell_1x1(_lib.ptr(x), _lib.ptr(h), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(out),
                                                 _lib.ptr(out_relu), B, F, H * W, _lib.stream_ptr()), "mrx_conv2dgru_cell_1x1")
    return out, out_relu


def mul_sigmoid(h, pre):
    """h * sigmoid(pre); h None = zeros."""
    pre = _lib.f32c(pre)
    h = None if h is None else _lib.f32c(h)
    out = torch.empty_like(pre)
    _lib.check(_lib.lib().mrx_mul_sigmoid(_lib.ptr(h), _lib.ptr(pre), _lib.ptr(out), pre.numel(), _lib.stream_ptr()), "mrx_mul_sigmoid")
    return out


def gru_blend(h, pre_update, pre_out, relu_out=True):
    """h * (1 - sigmoid(pre_update)) + tanh(pre_out) * sigmoid(pre_update) -> (new state, ReLU(new state) | None); h None = zeros."""
    pu, po = _lib.f32c(pre_update), _lib.f32c(pre_out)
    h = None if h is None else _lib.f32c(h)
    out = torch.empty_like(pu)
    out_relu = torch.empty_like(pu) if relu_out else None
    _lib.check(_lib.lib().mrx_gru_blend(_lib.ptr(h), _lib.ptr(pu), _lib.ptr(po), _lib.ptr(out), _lib.ptr(out_relu), pu.numel(),
                                        _lib.stream_ptr()), "mrx_gru_blend")
    return out, out_relu


def mgu_gates(ih, hh, h):
    ih, hh, h = _lib.f32c(ih), _lib.f32c(hh), _lib.f32c(h)
    B, F, H, W = _nchw(h)
    out = torch.empty_like(h)
    _lib.check(_lib.lib().mrx_mgu_gates(_lib.ptr(ih), _lib.ptr(hh), _lib.ptr(h), _lib.ptr(out), B, F, H * W, _lib.stream_ptr()),
               "mrx_mgu_gates")
    return out


# ---- NormUnet pieces -------------------------------------------------------------------------------------------
def instance_norm_act(x, eps=1e-5, act=ACT_LEAKY, slope=0.2, inplace=True, return_work=False):
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    out = x if inplace else torch.empty_like(x)
    work = torch.empty(int(_lib.lib().mrx_norm_work_floats(B * C, H * W)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_instance_norm_act(_lib.ptr(x), _lib.ptr(out), _lib.ptr(work), B * C, H * W, float(eps), int(act),
                                                float(slope), _lib.stream_ptr()), "mrx_instance_norm_act")
    return (out, work) if return_work else out            # work: the per-plane partial sums / squared deviations (mrx_inorm_act_bwd reads rstd from them)


# ---- backward steps of the U-Net training path (csrc/diff_bwd.hip; callers: mridc_amd/diff.py) -----------------------------------------------
def act_bwd(dy, y, act, slope=0.0):
    """dy * act'(y) with y the activation's output (mrx_act_bwd)."""
    if act == ACT_NONE:
        return dy
    dy, y = _lib.f32c(dy), _lib.f32c(y)
    dx = torch.empty_like(dy)
    _lib.check(_lib.lib().mrx_act_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dx), dy.numel(), int(act), float(slope), _lib.stream_ptr()), "mrx_act_bwd")
    return dx


def instance_norm_act_bwd(dy, y, fwd_work, eps, act, slope):
    """Backward of act(InstanceNorm2d(x)) from the output y and the forward's work buffer (mrx_inorm_act_bwd)."""
    dy, y = _lib.f32c(dy), _lib.f32c(y)
    B, C, H, W = _nchw(y)
    L = _lib.lib()
    dx = torch.empty_like(y)
    work = torch.empty(int(L.mrx_inorm_act_bwd_work_floats(B * C, H * W)), dtype=torch.float32, device=y.device)
    _lib.check(L.mrx_inorm_act_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(fwd_work), _lib.ptr(dx), _lib.ptr(work), B * C, H * W, float(eps), int(act), float(slope),
                                   _lib.stream_ptr()), "mrx_inorm_act_bwd")
    return dx


def avg_pool2x2_bwd(dy, H, W):
    dy = _lib.f32c(dy)
    B, C = int(dy.shape[0]), int(dy.shape[1])
    dx = torch.empty(B, C, H, W, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().mrx_avgpool2x2_bwd(_lib.ptr(dy), _lib.ptr(dx), B * C, int(H), int(W), _lib.stream_ptr()), "mrx_avgpool2x2_bwd")
    return dx


def pixel_unshuffle2(x):
    """[B,C,2H,2W] -> [B,4C,H,W], channel (c, i, j) (mrx_pixel_unshuffle2)."""
    x = _lib.f32c(x)
    B, C, H2, W2 = _nchw(x)
    if H2 % 2 or W2 % 2:
        raise ValueError("pixel_unshuffle2: odd size")
    out = torch.empty(B, 4 * C, H2 // 2, W2 // 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_pixel_unshuffle2(_lib.ptr(x), _lib.ptr(out), B * C, H2 // 2, W2 // 2, _lib.stream_ptr()), "mrx_pixel_unshuffle2")
    return out


def cmul_bcast(a, v, conj_v=False, scale=1.0):
    """a [B,C,H,W,2] * v [B,H,W,2] (or its conjugate) * scale, complex (mrx_cmul_bcast)."""
    a, v = _lib.f32c(a), _lib.f32c(v)
    B, C, H, W = _bchw(a)
    if tuple(v.shape) != (B, H, W, 2):
        raise ValueError(f"cmul_bcast: {tuple(a.shape)} vs {tuple(v.shape)}")
    out = torch.empty_like(a)
    _lib.check(_lib.lib().mrx_cmul_bcast(_lib.ptr(a), _lib.ptr(v), _lib.ptr(out), B, C, H * W, int(bool(conj_v)), float(scale), _lib.stream_ptr()),
               "mrx_cmul_bcast")
    return out


def sens_expand_bwd_pointwise(G, sens, x, want_dx, want_ds, scale=1.0):
    """From G = adjoint-fft2(dy) [B,C,H,W,2]: (dx = scale * sum_c conj(S) G | None, dS = scale * conj(x) G | None) in one pass (mrx_sens_expand_bwd_pw)."""
    G, sens = _lib.f32c(G), _lib.f32c(sens)
    B, C, H, W = _bchw(G)
    if sens.shape != G.shape:
        raise ValueError(f"sens_expand_bwd_pointwise: {tuple(G.shape)} vs maps {tuple(sens.shape)}")
    xs = None
    if want_ds:
        xs = _lib.f32c(x).reshape(B, H, W, 2)
    dx = torch.empty(B, H, W, 2, dtype=torch.float32, device=G.device) if want_dx else None
    dS = torch.empty_like(G) if want_ds else None
    _lib.check(_lib.lib().mrx_sens_expand_bwd_pw(_lib.ptr(G), _lib.ptr(sens), _lib.ptr(xs), _lib.ptr(dx), _lib.ptr(dS), B, C, H * W, float(scale),
                                                 _lib.stream_ptr()), "mrx_sens_expand_bwd_pw")
    return dx, dS


def dc_combine_bwd(dy, pred, ref, mask, dc_weight, want_dpred, want_deta, want_dw, add_dy):
    """Backward of dc_combine (mrx_dc_combine_bwd): (dpred | None, deta | None, dw [1] | None)."""
    if not (want_dpred or want_deta or want_dw):
        return None, None, None
    dy = _lib.f32c(dy)
    B, C, H, W = _bchw(dy)
    m, kind, ms = _lib.mask_args(mask, B, C, H, W)
    w = _lib.f32c(dc_weight.detach().reshape(-1))
    L = _lib.lib()
    dpred = torch.empty_like(dy) if want_dpred else None
    deta = torch.empty_like(dy) if want_deta else None
    dw = work = p = r = None
    if want_dw:
        p, r = _lib.f32c(pred), _lib.f32c(ref)
        dw = torch.empty(1, dtype=torch.float32, device=dy.device)
        work = torch.empty(int(L.mrx_dc_combine_bwd_work_doubles()), dtype=torch.float64, device=dy.device)
    _lib.check(L.mrx_dc_combine_bwd(_lib.ptr(dy), _lib.ptr(p), _lib.ptr(r), _lib.ptr(m), kind, ms, _lib.ptr(w), _lib.ptr(dpred), _lib.ptr(deta), _lib.ptr(dw),
                                    _lib.ptr(work), int(bool(add_dy)), B, C, H, W, _lib.stream_ptr()), "mrx_dc_combine_bwd")
    return dpred, deta, dw


def conv_instance_norm_act(x, weight, eps=1e-5, act=ACT_LEAKY, slope=0.2, pad_mode=PAD_ZERO):
    """Conv2d(3x3, no bias) -> InstanceNorm2d -> activation (unet_block.py:251-254).  Tuned shapes: the statistics come out of the conv's
    own accumulators (mrx_conv2d_stats) and one more pass applies them; other shapes: conv, then the three-pass instance norm."""
    x, weight = _lib.f32c(x), _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cout, Cin_w, kh, kw = [int(v) for v in weight.shape]
    if Cin_w != Cin:
        raise RuntimeError(f"input has inconsistent input_size: got {Cin}, expected {Cin_w}")
    L = _lib.lib()
    if kh != kw or not L.mrx_conv2d_stats_supported(B, Cout, H, W, kh, 1):
        return instance_norm_act(conv2d(x, weight, None, 1, pad_mode), eps, act, slope)
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    work = torch.empty(int(L.mrx_conv2d_stats_work_floats(B, Cout, H, W)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv2d_stats(_lib.ptr(x), _lib.ptr(weight), None, _lib.ptr(y), None, _lib.ptr(work), B, Cin, Cout, H, W,
                                  kh, 1, int(pad_mode), _lib.stream_ptr()), "mrx_conv2d_stats")
    _lib.check(L.mrx_instance_norm_apply_tiles(_lib.ptr(y), _lib.ptr(y), _lib.ptr(work), B, Cout, H, W, float(eps), int(act),
                                               float(slope), _lib.stream_ptr()), "mrx_instance_norm_apply_tiles")
    return y


def conv2d_stats(x, weight, pad_mode=PAD_ZERO):
    """(conv3x3(x), per-plane (mean, sum of squared deviations) [B, Cout, 2]) in one pass (mrx_conv2d_stats, tuned shapes only)."""
    x, weight = _lib.f32c(x), _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cout, _, kh, _ = [int(v) for v in weight.shape]
    L = _lib.lib()
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    stats = torch.empty(B, Cout, 2, dtype=torch.float32, device=x.device)
    work = torch.empty(int(L.mrx_conv2d_stats_work_floats(B, Cout, H, W)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv2d_stats(_lib.ptr(x), _lib.ptr(weight), None, _lib.ptr(y), _lib.ptr(stats), _lib.ptr(work), B, Cin, Cout, H, W,
                                  kh, 1, int(pad_mode), _lib.stream_ptr()), "mrx_conv2d_stats")
    return y, stats


def instance_norm_apply(x, stats, eps=1e-5, act=ACT_LEAKY, slope=0.2):
    """act((x - mean) / sqrt(M2 / HW + eps)) from ready per-plane statistics [B, C, 2]."""
    x, stats = _lib.f32c(x), _lib.f32c(stats)
    B, C, H, W = _nchw(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_instance_norm_apply(_lib.ptr(x), _lib.ptr(out), _lib.ptr(stats), B * C, H * W, float(eps), int(act),
                                                  float(slope), _lib.stream_ptr()), "mrx_instance_norm_apply")
    return out


def group_norm(x, groups):
    """(x - mean)/std per (b, group) with the unbiased std.  Returns (normalised, mean[B,G,1], std[B,G,1])."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    n = (C * H * W) // groups
    mean = torch.empty(B, groups, 1, dtype=torch.float32, device=x.device)
    std = torch.empty_like(mean)
    out = torch.empty_like(x)
    L = _lib.lib()
    work = torch.empty(int(L.mrx_norm_work_floats(B * groups, n)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_group_norm_stats(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(work), B * groups, n,
                                      _lib.stream_ptr()), "mrx_group_norm_stats")
    _lib.check(L.mrx_group_norm_apply(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(out), B * groups, n, 0,
                                      _lib.stream_ptr()), "mrx_group_norm_apply")
    return out, mean, std


def group_unnorm(x, mean, std, groups):
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    n = (C * H * W) // groups
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_group_norm_apply(_lib.ptr(x), _lib.ptr(_lib.f32c(mean)), _lib.ptr(_lib.f32c(std)), _lib.ptr(out),
                                               B * groups, n, 1, _lib.stream_ptr()), "mrx_group_norm_apply")
    return out


def group_norm_bwd(dy, xhat, std, groups, dmean=None, dstd=None):
    """dx of group_norm given the gradients of its three results (mrx_group_norm_bwd, inverse = 0): dy of the normalised tensor `xhat`, dmean / dstd [B,G,1] or None."""
    dy, xhat, std = _lib.f32c(dy), _lib.f32c(xhat), _lib.f32c(std)
    B, C, H, W = _nchw(xhat)
    n = (C * H * W) // groups
    if tuple(dy.shape) != tuple(xhat.shape) or std.numel() != B * groups:
        raise ValueError(f"group_norm_bwd: dy {tuple(dy.shape)}, xhat {tuple(xhat.shape)}, std {tuple(std.shape)}, groups {groups}")
    dm = _lib.f32c(dmean) if dmean is not None else None
    ds = _lib.f32c(dstd) if dstd is not None else None
    dx = torch.empty_like(xhat)
    L = _lib.lib()
    work = torch.empty(int(L.mrx_norm_work_floats(B * groups, n)), dtype=torch.float32, device=dy.device)
    _lib.check(L.mrx_group_norm_bwd(_lib.ptr(dy), _lib.ptr(xhat), _lib.ptr(std), _lib.ptr(dm), _lib.ptr(ds), _lib.ptr(dx), None, None, _lib.ptr(work),
                                    B * groups, n, 0, _lib.stream_ptr()), "mrx_group_norm_bwd")
    return dx


def group_unnorm_bwd(dy, x, std, groups):
    """(dx, dmean [B,G,1], dstd [B,G,1]) of group_unnorm (mrx_group_norm_bwd, inverse = 1); x: the un-normalisation's input."""
    dy, x, std = _lib.f32c(dy), _lib.f32c(x), _lib.f32c(std)
    B, C, H, W = _nchw(x)
    n = (C * H * W) // groups
    if tuple(dy.shape) != tuple(x.shape) or std.numel() != B * groups:
        raise ValueError(f"group_unnorm_bwd: dy {tuple(dy.shape)}, x {tuple(x.shape)}, std {tuple(std.shape)}, groups {groups}")
    dx = torch.empty_like(x)
    dmean = torch.empty(B, groups, 1, dtype=torch.float32, device=dy.device)
    dstd = torch.empty_like(dmean)
    L = _lib.lib()
    work = torch.empty(int(L.mrx_norm_work_floats(B * groups, n)), dtype=torch.float32, device=dy.device)
    _lib.check(L.mrx_group_norm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(std), None, None, _lib.ptr(dx), _lib.ptr(dmean), _lib.ptr(dstd), _lib.ptr(work),
                                    B * groups, n, 1, _lib.stream_ptr()), "mrx_group_norm_bwd")
    return dx, dmean, dstd


def pad2d(x, top, bottom, left, right, mode=0):
    """Zero (mode 0; negative values crop) or reflect (mode 1) padding of the last two dims."""
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    out = torch.empty(B, C, H + top + bottom, W + left + right, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_pad2d(_lib.ptr(x), _lib.ptr(out), B * C, H, W, int(top), int(bottom), int(left), int(right),
                                    int(mode), _lib.stream_ptr()), "mrx_pad2d")
    return out


def avg_pool2x2(x):
    x = _lib.f32c(x)
    B, C, H, W = _nchw(x)
    out = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_avg_pool2x2(_lib.ptr(x), _lib.ptr(out), B * C, H, W, _lib.stream_ptr()), "mrx_avg_pool2x2")
    return out


def conv_transpose2x2(x, weight):
    x, weight = _lib.f32c(x), _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cin_w, Cout, kh, kw = [int(v) for v in weight.shape]
    if Cin_w != Cin or kh != 2 or kw != 2:
        raise ValueError("conv_transpose2x2 expects weight [Cin,Cout,2,2]")
    out = torch.empty(B, Cout, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_conv_transpose2x2(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(out), B, Cin, Cout, H, W,
                                                _lib.stream_ptr()), "mrx_conv_transpose2x2")
    return out


def conv_transpose2x2_instance_norm_act(x, weight, eps=1e-5, act=ACT_LEAKY, slope=0.2):
    """ConvTranspose2d(k 2, s 2, no bias) -> InstanceNorm2d -> activation (unet_block.py:296-299): the statistics come out of the transposed
    convolution's accumulators (mrx_conv_transpose2x2_stats), one more pass applies them; shapes without the tuned kernel take the three-pass norm."""
    x, weight = _lib.f32c(x), _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cin_w, Cout, kh, kw = [int(v) for v in weight.shape]
    if Cin_w != Cin or kh != 2 or kw != 2:
        raise ValueError("conv_transpose2x2 expects weight [Cin,Cout,2,2]")
    L = _lib.lib()
    if Cout % 2 or Cin * 14 * 16 > 48 * 1024:
        return instance_norm_act(conv_transpose2x2(x, weight), eps, act, slope)
    out = torch.empty(B, Cout, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    stats = torch.empty(B, Cout, 2, dtype=torch.float32, device=x.device)
    work = torch.empty(int(L.mrx_conv_transpose2x2_stats_work_floats(B, Cout, H, W)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_conv_transpose2x2_stats(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(out), _lib.ptr(stats), _lib.ptr(work), B, Cin, Cout, H, W,
                                             _lib.stream_ptr()), "mrx_conv_transpose2x2_stats")
    _lib.check(L.mrx_instance_norm_apply(_lib.ptr(out), _lib.ptr(out), _lib.ptr(stats), B * Cout, 4 * H * W, float(eps), int(act),
                                         float(slope), _lib.stream_ptr()), "mrx_instance_norm_apply")
    return out


def concat_channels(a, b):
    """torch.cat([a, b], dim=1) in one launch."""
    a, b = _lib.f32c(a), _lib.f32c(b)
    B, Ca, H, W = _nchw(a)
    Cb = int(b.shape[1])
    out = torch.empty(B, Ca + Cb, H, W, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mrx_concat_channels(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), B, Ca, Cb, H * W, _lib.stream_ptr()),
               "mrx_concat_channels")
    return out


# ---- U-Net without normalisation passes (csrc/unet_fused.hip) ----------------------------------------------------
# A "lazy" tensor is the pair (raw, norm): raw [B,C,H,W] convolution output, norm [B,C,2] per-plane (mean, 1/std) of the InstanceNorm2d that
# follows it in the reference (unet_block.py:251-258, 296-299); whoever reads it applies leaky((raw - mean) / std) while loading.
def _lazy(t):
    """(raw, norm or None) of a plain tensor or a (raw, norm) pair."""
    if isinstance(t, tuple):
        return _lib.f32c(t[0]), _lib.f32c(t[1])
    return _lib.f32c(t), None


_UNET_PACKS = _PreparedCache(keep=512, parameters=True)
_UNET_BOUNDS = {}
UNET_F16 = True                   # (module attribute: a test hook for the fp32-input MFMA kernel; MRIDC_AMD_ARITH != f16x2 turns the fp16 form off too)


# Merge the tile statistics inside the convolution launch (mrx_unet_conv3x3_hc) instead of a k_unorm_finalize launch behind it.  Built and measured in round 5,
# OFF by default: E2EVN-6 at 8 slices x 2 streams 1 105-1 168 slices/s without it, 148 with an agent-scope release fence per tile (every fence writes the
# XCD's L2 back), 590-600 with write-through statistics and 112 adjacent tickets (four cache lines: 107 k read-modify-writes one after the other), 900-920
# with one 128-byte line per ticket and the tile stored behind the tickets -- 960 returning atomics per plane address still cost more than the 5-us launch
# they replace (tools/runs/r05i.sh).  MRX_UNET_FOLD=1 / this attribute switch it on (tests/test_gpu_unet_fused.py covers it).
UNET_FOLD_FINALIZE = os.environ.get("MRX_UNET_FOLD", "0") == "1"
_UNET_TICKETS = {}


def _unet_tickets(n, device):
    """A zeroed int32 buffer of >= n plane tickets for mrx_unet_conv3x3_hc, one per (device, stream, hipGraph capture): the kernel leaves it zeroed, so
    consecutive calls on a stream share it; calls that may overlap (other streams) get their own; a buffer first made INSIDE a capture lives in that
    graph's pool -- its zero-fill is part of the graph -- and is never served to eager calls or other captures (the rule of _analytic_bound)."""
    cap = int(_lib.lib().mrx_stream_capture_id(_lib.stream_ptr()))
    for k in [k for k in _UNET_TICKETS if k[2] != 0 and k[2] != cap]:
        del _UNET_TICKETS[k]
    key = (str(device), int(torch.cuda.current_stream().cuda_stream), cap)
    t = _UNET_TICKETS.get(key)
    if t is None or t.numel() < n:
        t = _UNET_TICKETS[key] = torch.zeros(max(int(n), 4096), dtype=torch.int32, device=device)
    return t


def _analytic_bound(n, device):
    """Device scalar sqrt(n): the bound of an instance- / group-normalised tensor whose statistics ran over n values (|z| <= sqrt(n - 1))."""
    # keyed on the hipGraph capture like _PreparedCache: a scalar first filled INSIDE a capture exists only in that graph's pool and only after a
    # replay -- an eager call (or another capture) must not read it
    cap = int(_lib.lib().mrx_stream_capture_id(_lib.stream_ptr()))
    for k in [k for k in _UNET_BOUNDS if k[2] != 0 and k[2] != cap]:
        del _UNET_BOUNDS[k]
    key = (int(n), str(device), cap)
    t = _UNET_BOUNDS.get(key)
    if t is None:
        t = _UNET_BOUNDS[key] = torch.full((1,), float(n) ** 0.5, dtype=torch.float32, device=device)
    return t


_ZERO_SCALARS = {}


def _zero_scalar(device):
    """A fresh zeroed 1-element device tensor for a kernel that folds a maximum into it (atomic max): handed out of a block of 256 zeros made by ONE fill
    launch per (device, stream, hipGraph capture) -- `torch.zeros(1)` per call was a 3-5 us fill kernel each (127 of them per 7 qCIRIM steps, 5 % of that
    configuration's kernel time).  Every scalar is handed out once; a block made inside a capture belongs to that graph (the rule of _analytic_bound)."""
    cap = int(_lib.lib().mrx_stream_capture_id(_lib.stream_ptr()))
    for k in [k for k in _ZERO_SCALARS if k[2] != 0 and k[2] != cap]:
        del _ZERO_SCALARS[k]
    key = (str(device), int(torch.cuda.current_stream().cuda_stream), cap)
    blk = _ZERO_SCALARS.get(key)
    if blk is None or blk[1] >= blk[0].numel():
        blk = _ZERO_SCALARS[key] = [torch.zeros(256, dtype=torch.float32, device=device), 0]
    i = blk[1]
    blk[1] = i + 1
    return blk[0][i:i + 1]


def _attach_bound(t, bound):
    """Leave a device scalar >= max |t| for the consumers of t (_lib.bound_attach: a table of memory ranges owned by the binding; every library
    call that may write a range drops the entries overlapping it, a torch write bumps the version the entry remembers)."""
    return _lib.bound_attach(t, bound)


def _plain_bound(x):
    """Device scalar >= max |x| of a plain tensor: the bound its producer left (unet_cnorm_pad, unet_avg_pool2x2, the 128-channel 1x1 cell
    kernel) if nothing has written the tensor's memory since, else measured (mrx_max_abs)."""
    b = _lib.bound_of(x)
    return b if b is not None else max_abs(x).reshape(1)


def unet_conv3x3(src_a, src_b, weight, eps=1e-5, slope=0.2):
    """Conv2d(3x3, zero pad, no bias) over the channels of src_a then src_b (None: one source; each plain or lazy) -> lazy output: the InstanceNorm
    statistics come out of the accumulators, nothing is normalised or concatenated in memory.  Default arithmetic: two-term fp16 operands
    (mrx_unet_conv3x3_h, csrc/unet_f16.hip); MRIDC_AMD_ARITH=bf16x3 / fp32: the fp32-input MFMA kernel (mrx_unet_conv3x3)."""
    xa, na = _lazy(src_a)
    xb, nb = _lazy(src_b) if src_b is not None else (None, None)
    weight = _lib.f32c(weight.detach())
    B, Ca, H, W = _nchw(xa)
    Cb = int(xb.shape[1]) if xb is not None else 0
    Cout = int(weight.shape[0])
    if tuple(weight.shape[1:]) != (Ca + Cb, 3, 3) or (xb is not None and (xb.shape[0], xb.shape[2], xb.shape[3]) != (B, H, W)):
        raise RuntimeError(f"unet_conv3x3: weight {tuple(weight.shape)} vs sources with {Ca} + {Cb} channels")
    L = _lib.lib()
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=xa.device)
    norm = torch.empty(B, Cout, 2, dtype=torch.float32, device=xa.device)
    work = torch.empty(int(L.mrx_unet_conv3x3_work_floats(B, Cout, H, W)), dtype=torch.float32, device=xa.device)
    if UNET_F16 and _lib.arith() == "f16x2":
        def make():
            pk = torch.empty(int(L.mrx_unet_conv3x3_pack_floats(Cout, Ca + Cb)), dtype=torch.float32, device=weight.device)
            _lib.check(L.mrx_unet_conv3x3_pack(_lib.ptr(weight), Cout, Ca + Cb, _lib.ptr(pk), _lib.stream_ptr()), "mrx_unet_conv3x3_pack")
            return pk
        packed = _UNET_PACKS.get(weight, (), make)
        ba = None if na is not None else _plain_bound(src_a if not isinstance(src_a, tuple) else xa)
        bb = None if (xb is None or nb is not None) else _plain_bound(src_b if not isinstance(src_b, tuple) else xb)
        if _precision16():
            _lib.check(L.mrx_unet_conv3x3_p16(_lib.ptr(xa), _lib.ptr(na), _lib.ptr(ba), Ca, _lib.ptr(xb), _lib.ptr(nb), _lib.ptr(bb), Cb, _lib.ptr(packed),
                                              _lib.ptr(y), _lib.ptr(norm), _lib.ptr(work), B, Cout, H, W, float(eps), float(slope), _lib.stream_ptr()),
                       "mrx_unet_conv3x3_p16")
            return y, norm
        if UNET_FOLD_FINALIZE:
            tickets = _unet_tickets(int(L.mrx_unet_conv3x3_hc_ticket_ints(B, Cout)), xa.device)
            _lib.check(L.mrx_unet_conv3x3_hc(_lib.ptr(xa), _lib.ptr(na), _lib.ptr(ba), Ca, _lib.ptr(xb), _lib.ptr(nb), _lib.ptr(bb), Cb, _lib.ptr(packed),
                                             _lib.ptr(y), _lib.ptr(norm), _lib.ptr(work), _lib.ptr(tickets), B, Cout, H, W, float(eps), float(slope),
                                             _lib.stream_ptr()), "mrx_unet_conv3x3_hc")
            return y, norm
        _lib.check(L.mrx_unet_conv3x3_h(_lib.ptr(xa), _lib.ptr(na), _lib.ptr(ba), Ca, _lib.ptr(xb), _lib.ptr(nb), _lib.ptr(bb), Cb, _lib.ptr(packed),
                                        _lib.ptr(y), _lib.ptr(norm), _lib.ptr(work), B, Cout, H, W, float(eps), float(slope), _lib.stream_ptr()),
                   "mrx_unet_conv3x3_h")
        return y, norm
    _lib.check(L.mrx_unet_conv3x3(_lib.ptr(xa), _lib.ptr(na), Ca, _lib.ptr(xb), _lib.ptr(nb), Cb, _lib.ptr(weight), _lib.ptr(y), _lib.ptr(norm),
                                  _lib.ptr(work), B, Cout, H, W, float(eps), float(slope), _lib.stream_ptr()), "mrx_unet_conv3x3")
    return y, norm


def unet_conv_transpose2x2_supported(Cin, Cout):
    return int(Cout) % 2 == 0 and int(Cin) * (2 * 16 + 8) <= 48 * 1024


def unet_conv_transpose2x2(src, weight, eps=1e-5, slope=0.2):
    """ConvTranspose2d(k 2, s 2, no bias) of a plain or lazy tensor -> lazy output (mrx_unet_conv_transpose2x2)."""
    x, nrm = _lazy(src)
    weight = _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cin_w, Cout, kh, kw = [int(v) for v in weight.shape]
    if Cin_w != Cin or kh != 2 or kw != 2:
        raise ValueError("unet_conv_transpose2x2 expects weight [Cin,Cout,2,2]")
    L = _lib.lib()
    out = torch.empty(B, Cout, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    norm = torch.empty(B, Cout, 2, dtype=torch.float32, device=x.device)
    work = torch.empty(int(L.mrx_unet_conv_transpose2x2_work_floats(B, Cout, H, W)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_unet_conv_transpose2x2(_lib.ptr(x), _lib.ptr(nrm), _lib.ptr(weight), _lib.ptr(out), _lib.ptr(norm), _lib.ptr(work), B, Cin, Cout,
                                   H, W, float(eps), float(slope), _lib.stream_ptr()), "mrx_unet_conv_transpose2x2")
    return out, norm


def unet_avg_pool2x2(src, slope=0.2):
    """avg_pool2d(2) of a plain or lazy tensor -> plain tensor (mrx_unet_avgpool)."""
    x, nrm = _lazy(src)
    B, C, H, W = _nchw(x)
    out = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_unet_avgpool(_lib.ptr(x), _lib.ptr(nrm), _lib.ptr(out), B * C, H, W, float(slope), _lib.stream_ptr()),
               "mrx_unet_avgpool")
    # an average is bounded by what it averages: sqrt(n) of the normalised planes, or the plain source's own bound (if it has one)
    if nrm is not None:
        _attach_bound(out, _analytic_bound(H * W, x.device))
    elif _lib.bound_of(x) is not None:
        _attach_bound(out, _lib.bound_of(x))
    return out


def unet_apply(src, slope=0.2):
    """The plain tensor leaky((raw - mean) / std) of a lazy one (mrx_unet_apply)."""
    x, nrm = _lazy(src)
    if nrm is None:
        return x
    B, C, H, W = _nchw(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_unet_apply(_lib.ptr(x), _lib.ptr(nrm), _lib.ptr(out), B * C, H * W, float(slope), _lib.stream_ptr()),
               "mrx_unet_apply")
    return out


def unet_conv1x1(src, weight, bias, slope=0.2):
    """1x1 convolution (+ bias) of a plain or lazy tensor into <= 4 channels -> plain tensor (mrx_unet_conv1x1)."""
    x, nrm = _lazy(src)
    weight = _lib.f32c(weight.detach())
    B, Cin, H, W = _nchw(x)
    Cout = int(weight.shape[0])
    if tuple(weight.shape[1:]) != (Cin, 1, 1):
        raise RuntimeError(f"unet_conv1x1: weight {tuple(weight.shape)} vs {Cin} input channels")
    b = _lib.f32c(bias.detach()) if bias is not None else None
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_unet_conv1x1(_lib.ptr(x), _lib.ptr(nrm), _lib.ptr(weight), _lib.ptr(b), _lib.ptr(out), B, Cin, Cout, H * W,
                                           float(slope), _lib.stream_ptr()), "mrx_unet_conv1x1")
    return out


def unet_cnorm_pad(x, h_pad, w_pad):
    """pad(norm(complex_to_chan_dim(x))) of NormUnet.forward (unet_block.py:46-112, norm_groups = 2) for x [B,c,H,W,2]: two statistics passes
    and one normalise + permute + zero-pad pass (mrx_unet_cnorm_pad).  Returns (out [B,2c,H',W'], mean [B,2,1], std [B,2,1])."""
    x = _lib.f32c(x)
    B, c, H, W, two = [int(v) for v in x.shape]
    if two != 2:
        raise AssertionError
    L = _lib.lib()
    out = torch.empty(B, 2 * c, H + h_pad[0] + h_pad[1], W + w_pad[0] + w_pad[1], dtype=torch.float32, device=x.device)
    mean = torch.empty(B, 2, 1, dtype=torch.float32, device=x.device)
    std = torch.empty_like(mean)
    work = torch.empty(int(L.mrx_unet_cnorm_work_floats(B)), dtype=torch.float32, device=x.device)
    _lib.check(L.mrx_unet_cnorm_pad(_lib.ptr(x), _lib.ptr(out), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(work), B, c, H, W, int(h_pad[0]),
                                    int(h_pad[1]), int(w_pad[0]), int(w_pad[1]), _lib.stream_ptr()), "mrx_unet_cnorm_pad")
    _attach_bound(out, _analytic_bound(c * H * W, x.device))    # normalised per (batch, component) over c H W values (unbiased std: smaller still)
    return out, mean, std


def unet_conv1x1_cunnorm(src, weight, bias, mean, std, top, left, H, W, slope=0.2):
    """chan_complex_to_last_dim(unnorm(unpad(conv1x1(src)))) (unet_block.py:186-189, 114-136) -> [B,c,H,W,2] in one launch
    (mrx_unet_conv1x1_cunnorm); src plain or lazy [B,Cin,H',W'], weight [2c,Cin,1,1] with c <= 2."""
    x, nrm = _lazy(src)
    weight = _lib.f32c(weight.detach())
    B, Cin, OH, OW = _nchw(x)
    Cout = int(weight.shape[0])
    if tuple(weight.shape[1:]) != (Cin, 1, 1) or Cout % 2:
        raise RuntimeError(f"unet_conv1x1_cunnorm: weight {tuple(weight.shape)} vs {Cin} input channels")
    b = _lib.f32c(bias.detach()) if bias is not None else None
    out = torch.empty(B, Cout // 2, H, W, 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_unet_conv1x1_cunnorm(_lib.ptr(x), _lib.ptr(nrm), _lib.ptr(weight), _lib.ptr(b), _lib.ptr(_lib.f32c(mean)),
                                                   _lib.ptr(_lib.f32c(std)), _lib.ptr(out), B, Cin, Cout // 2, OH, OW, int(top), int(left), int(H),
                                                   int(W), float(slope), _lib.stream_ptr()), "mrx_unet_conv1x1_cunnorm")
    return out


# ---- quantitative MRI (A19) -----------------------------------------------------------------------------------
def _tes_host(TEs):
    import ctypes
    vals = [float(t) for t in TEs]
    return (ctypes.c_float * len(vals))(*vals), len(vals)


def qmri_signal(r2, s0, b0, phi, TEs, scaling=1e-3):
    """MEGRE signal model: maps [N,H,W] -> [N,E,H,W,2]."""
    r2, s0, b0, phi = (_lib.f32c(t) for t in (r2, s0, b0, phi))
    N, H, W = [int(v) for v in r2.shape]
    tes, E = _tes_host(TEs)
    out = torch.empty(N, E, H, W, 2, dtype=torch.float32, device=r2.device)
    _lib.check(_lib.lib().mrx_qmri_signal(_lib.ptr(r2), _lib.ptr(s0), _lib.ptr(b0), _lib.ptr(phi), tes, E, _lib.ptr(out), N, H * W,
                                          float(scaling), _lib.stream_ptr()), "mrx_qmri_signal")
    return out


def dc_residual(x, y, sens, mask, sdiv, centered, normalization):
    """sum_c conj(S) ifft2(mask (fft2(x S) - y)) with maps shared by groups of `sdiv` batch entries.
    x [B',H,W,2]; y [B',C,H,W,2]; sens [B'/sdiv,C,H,W,2]; mask broadcastable to [B',C,H,W,1] -> [B',H,W,2]."""
    x, y, sens = _lib.f32c(x), _lib.f32c(y), _lib.f32c(sens)
    Bp, C, H, W = _bchw(y)
    if tuple(x.shape) != (Bp, H, W, 2) or tuple(sens.shape) != (Bp // sdiv, C, H, W, 2) or Bp % sdiv:
        raise ValueError("dc_residual: inconsistent shapes")
    m, kind, ms = _lib.mask_args(mask, Bp, C, H, W)
    out = torch.empty(Bp, H, W, 2, dtype=torch.float32, device=y.device)
    work = torch.empty_like(y)
    _lib.check(_lib.lib().mrx_dc_residual(_lib.ptr(x), _lib.ptr(y), _lib.ptr(sens), _lib.ptr(m), kind, ms, _lib.ptr(out),
                                          _lib.ptr(work), Bp, C, H, W, int(sdiv), _norm(normalization), int(bool(centered)),
                                          _lib.stream_ptr()), "mrx_dc_residual")
    return out


def qmri_grad(dinv, r2, s0, b0, phi, TEs, scaling=1e-3, post=1.0):
    """Analytic gradient from the coil-combined residual dinv [N,E,H,W,2] -> [N,4,H,W]."""
    dinv = _lib.f32c(dinv)
    r2, s0, b0, phi = (_lib.f32c(t) for t in (r2, s0, b0, phi))
    N, H, W = [int(v) for v in r2.shape]
    tes, E = _tes_host(TEs)
    if tuple(dinv.shape) != (N, E, H, W, 2):
        raise ValueError("qmri_grad: inconsistent shapes")
    out = torch.empty(N, 4, H, W, dtype=torch.float32, device=r2.device)
    _lib.check(_lib.lib().mrx_qmri_grad(_lib.ptr(dinv), _lib.ptr(r2), _lib.ptr(s0), _lib.ptr(b0), _lib.ptr(phi), tes, E,
                                        _lib.ptr(out), N, H * W, float(scaling), float(post), _lib.stream_ptr()), "mrx_qmri_grad")
    return out


def scale(x, s, take_abs=False, divide=False):
    x = _lib.f32c(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_scale(_lib.ptr(x), _lib.ptr(out), x.numel(), float(s), int(bool(take_abs)) | (2 if divide else 0),
                                    _lib.stream_ptr()), "mrx_scale")
    return out


def max_abs(x, complex_modulus=False):
    """0-dim device tensor: max |x| over every float of x, or (complex_modulus) max modulus over the complex values x[..., 2]."""
    x = _lib.f32c(x)
    n = x.numel() // 2 if complex_modulus else x.numel()
    if n < 1 or (complex_modulus and x.shape[-1] != 2):
        raise ValueError("max_abs: empty tensor or not a [..., 2] complex view")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    work = torch.empty(int(_lib.lib().mrx_max_abs_work_floats()), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_max_abs(_lib.ptr(x), n, int(bool(complex_modulus)), _lib.ptr(out), _lib.ptr(work), _lib.stream_ptr()),
               "mrx_max_abs")
    return out.reshape(())


def div_by_device_scalar(x, d, modulus=False):
    """x / d with d a 1-element device tensor (no host read); `modulus`: |x_c / d| of the complex view x[..., 2] -> real tensor."""
    x, d = _lib.f32c(x), _lib.f32c(d.reshape(1))
    if modulus:
        if x.shape[-1] != 2:
            raise ValueError("div_by_device_scalar: modulus needs a [..., 2] complex view")
        out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        n = out.numel()
    else:
        out = torch.empty_like(x)
        n = x.numel()
    _lib.check(_lib.lib().mrx_div_by_device_scalar(_lib.ptr(x), _lib.ptr(d), _lib.ptr(out), n, int(bool(modulus)), _lib.stream_ptr()),
               "mrx_div_by_device_scalar")
    return out


def recon_metrics(target, output):
    """Per-slice harness metrics of two `abs / max` images of one shape (models/base.py:427-436) -> device tensor
    [MSE, NMSE, maxval = max(output) - min(output), PSNR, sum target^2] (mrx_recon_metrics; no host read)."""
    target, output = _lib.f32c(target), _lib.f32c(output)
    if target.shape != output.shape or target.numel() < 1:
        raise ValueError(f"recon_metrics: target {tuple(target.shape)} vs output {tuple(output.shape)}")
    L = _lib.lib()
    out5 = torch.empty(5, dtype=torch.float32, device=target.device)
    work = torch.empty(int(L.mrx_recon_metrics_work_floats()), dtype=torch.float32, device=target.device)
    _lib.check(L.mrx_recon_metrics(_lib.ptr(target), _lib.ptr(output), _lib.ptr(out5), _lib.ptr(work), target.numel(), _lib.stream_ptr()),
               "mrx_recon_metrics")
    return out5


def qrim_update(eta, delta):
    eta, delta = _lib.f32c(eta), _lib.f32c(delta)
    B, Cc, H, W = _nchw(eta)
    out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_qrim_update(_lib.ptr(eta), _lib.ptr(delta), _lib.ptr(out), B, Cc, H * W, _lib.stream_ptr()),
               "mrx_qrim_update")
    return out
