"""ops wrappers of the channel-blocked RIM kernels (experiment, not part of the product library)."""
import torch

from mridc_amd import _lib
from mridc_amd.ops import _nchw


def cb8_from_nchw(x):
    """[B,C,H,W] -> the channel-blocked layout [B,C/8,H,W,8] of the cb8 RIM kernels (mrx_cb8_convert)."""
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


def rim_layer2_cb8(x, packed, b_conv, b_ih, hh, h_prev, xmax, taps=None, out=None, want_taps=False):
    """Second RIM layer on channel-blocked states (mrx_rim_layer2_cb8; x, h_prev, result [B,8,H,W,8]; `packed` from rim_layer2_f16_pack).
    Returns h_new, or (h_new, taps9 [B,9,H,W,2]) with want_taps (the final convolution's tap products for rim_final_gather9)."""
    x = _lib.f32c(x)
    B, Q, H, W, E = [int(v) for v in x.shape]
    if Q != 8 or E != 8:
        raise ValueError(f"rim_layer2_cb8 expects x [B,8,H,W,8], got {tuple(x.shape)}")
    bc = _lib.f32c(b_conv.detach()) if b_conv is not None else None
    bi = _lib.f32c(b_ih.detach()) if b_ih is not None else None
    hhc = _lib.f32c(hh.detach().reshape(-1))
    hp = _lib.f32c(h_prev) if h_prev is not None else None
    if want_taps and (taps is None or taps.numel() < 18 * B * H * W):
        taps = torch.empty(B, 9, H, W, 2, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(B, 8, H, W, 8, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rim_layer2_cb8(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bc), _lib.ptr(bi), _lib.ptr(hhc), _lib.ptr(hp), _lib.ptr(out),
                                             _lib.ptr(taps) if want_taps else None, _lib.ptr(xmax), B, H, W, _lib.stream_ptr()),
               "mrx_rim_layer2_cb8")
    return (out, taps) if want_taps else out


def rim_final_gather9(taps9, b_final, eta):
    """eta + permute(conv3x3_reppad(h) + b_final) [B,H,W,2] from the tap products [B,9,H,W,2] of rim_layer2_cb8 (mrx_rim_final_gather9)."""
    eta = _lib.f32c(eta)
    B, H, W, _ = [int(v) for v in eta.shape]
    if taps9.numel() < 18 * B * H * W or int(eta.shape[-1]) != 2:
        raise ValueError("rim_final_gather9 expects taps [B,9,H,W,2] and eta [B,H,W,2]")
    bf = _lib.f32c(b_final.detach()) if b_final is not None else None
    eta_out = torch.empty_like(eta)
    _lib.check(_lib.lib().mrx_rim_final_gather9(_lib.ptr(taps9), _lib.ptr(bf), _lib.ptr(eta), _lib.ptr(eta_out), B, H, W, _lib.stream_ptr()),
               "mrx_rim_final_gather9")
    return eta_out


