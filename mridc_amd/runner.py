"""The evaluation harness contract of `BaseMRIReconstructionModel` (reference models/base.py:347-438 `test_step`, :155-186
`process_inputs`, :490-517 metric aggregation, :35-53 `DistributedMetricSum`) for the model mirrors of this package -- SURVEY 8a row H.

The reference runs this inside pytorch-lightning; here it is a plain class around any of the models (CIRIM / VarNet / UNet / ZF / ...):

    runner = ReconstructionRunner(model)
    name, slice_num, pred = runner.test_step((kspace, y, sensitivity_maps, mask, init_pred, target, fname, slice_num, acc))
    metrics = runner.test_epoch_end()          # {"MSE", "NMSE", "SSIM", "PSNR"} averaged like base.py:490-517, summed over ranks

Post-processing (`abs`, divide by the maximum) and the four metrics (MSE / NMSE / PSNR / SSIM with `maxval = output.max() - output.min()`,
base.py:415-436) run on the device through libmridc_amd (`mrx_complex_abs`, `mrx_max_abs`, `mrx_div_by_device_scalar`,
`mrx_recon_metrics`, `mrx_ssim_loss`); nothing is read back to the host until `test_epoch_end`, so a stream of slices never
synchronises.  The cross-rank reduction is one 5-scalar all-reduce (the four metric sums + the example count)."""
from collections import defaultdict

import numpy as np
import torch

from mridc_amd import _lib, ops, sharding

__all__ = ["ReconstructionRunner", "postprocess", "slice_metrics", "metrics_to_dict"]


def _magnitude(x):
    """torch.abs of the reference for a complex tensor, a [..., 2] real view of one, or a real tensor -> real fp32 device tensor."""
    if x.is_complex():
        x = torch.view_as_real(x)
        cplx = True
    else:
        cplx = False
    x = _lib.f32c(x)
    if not cplx:
        out = torch.empty_like(x)
        _lib.check(_lib.lib().mrx_scale(_lib.ptr(x), _lib.ptr(out), x.numel(), 1.0, 1, _lib.stream_ptr()), "mrx_scale")
        return out
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_complex_abs(_lib.ptr(x), _lib.ptr(out), out.numel(), 0, _lib.stream_ptr()), "mrx_complex_abs")
    return out


def postprocess(preds, target):
    """base.py:415-419: `output = abs(preds); output /= output.max()` and the same for the target.  Device tensors in, device tensors out."""
    out = _magnitude(preds)
    out = ops.div_by_device_scalar(out, ops.max_abs(out))
    tgt = _magnitude(target)
    tgt = ops.div_by_device_scalar(tgt, ops.max_abs(tgt))
    return out, tgt


def slice_metrics(output, target):
    """base.py:427-436 on post-processed [B,h,w] images -> device tensor [MSE, NMSE, maxval, PSNR, sum target^2, 1 - SSIM]
    (two launches + the SSIM pair; `metrics_to_dict` reads it on the host)."""
    if output.dim() != 3 or output.shape != target.shape:
        # reconstruction_metrics.ssim raises for anything but [slices, h, w] (:30-33)
        raise ValueError("Unexpected number of dimensions in ground truth." if target.dim() != 3
                         else "Ground truth dimensions does not match pred.")
    target, output = _lib.f32c(target), _lib.f32c(output)
    B, h, w = [int(v) for v in output.shape]
    L = _lib.lib()
    m6 = torch.empty(6, dtype=torch.float32, device=output.device)
    work = torch.empty(max(int(L.mrx_recon_metrics_work_floats()), int(L.mrx_ssim_work_floats(B, h, w))), dtype=torch.float32,
                       device=output.device)
    _lib.check(L.mrx_recon_metrics(_lib.ptr(target), _lib.ptr(output), _lib.ptr(m6), _lib.ptr(work), target.numel(), _lib.stream_ptr()),
               "mrx_recon_metrics")
    dr = m6[2:3].expand(B)                                     # one data range for the whole batch (reconstruction_metrics.py:35-39)
    dr = dr if B == 1 else dr.contiguous()
    _lib.check(L.mrx_ssim_loss(_lib.ptr(target), _lib.ptr(output), _lib.ptr(dr), _lib.ptr(m6[5:6]), _lib.ptr(work), B, h, w, 7,
                               0.01, 0.03, _lib.stream_ptr()), "mrx_ssim_loss")
    return m6


def metrics_to_dict(m6):
    """Host view of one slice_metrics result."""
    v = m6.detach().double().cpu()
    return {"MSE": float(v[0]), "NMSE": float(v[1]), "SSIM": 1.0 - float(v[5]), "PSNR": float(v[3]), "maxval": float(v[2])}


class ReconstructionRunner:
    """`test_step` / `test_epoch_end` of the reference's base model around one of this package's models."""

    def __init__(self, model, to_numpy=True):
        self.model = model
        self.to_numpy = to_numpy
        self.accumulate_estimates = bool(getattr(model, "accumulate_estimates", False))
        self.use_sens_net = bool(getattr(model, "use_sens_net", False))
        self.metric_vals = defaultdict(dict)                   # fname -> slice -> device tensor of slice_metrics

    @staticmethod
    def process_inputs(y, mask, init_pred):
        """base.py:155-186: lists of accelerations -> one randomly selected entry."""
        if isinstance(y, list):
            r = np.random.randint(len(y))
            y, mask, init_pred = y[r], mask[r], init_pred[r]
        else:
            r = 0
        return y, mask, init_pred, r

    def predict(self, y, sensitivity_maps, mask, init_pred, target, kspace=None):
        """forward + the unwrapping of base.py:394-407 -> complex [B,h,w] on the device."""
        with torch.no_grad():
            if self.use_sens_net:
                # (under the reference's `precision: 16` the sensitivity network runs inside the same autocast as the model: base.py:392 in test_step)
                with ops.inference_precision(ops.resolve_precision16(self.model.precision) if hasattr(self.model, "precision") else None):
                    sensitivity_maps = self.model.sens_net(kspace, mask)
            preds = self.model.forward(y, sensitivity_maps, mask, init_pred, target)
            if self.accumulate_estimates:
                try:
                    preds = next(preds)
                except StopIteration:
                    pass
        if isinstance(preds, list):                            # cascades
            preds = preds[-1]
        if isinstance(preds, list):                            # time-steps
            preds = preds[-1]
        return preds

    def test_step(self, batch, batch_idx=0):
        kspace, y, sensitivity_maps, mask, init_pred, target, fname, slice_num, _ = batch
        y, mask, init_pred, _r = self.process_inputs(y, mask, init_pred)
        preds = self.predict(y, sensitivity_maps, mask, init_pred, target, kspace)
        slice_num = int(slice_num)
        name = str(fname[0]) if isinstance(fname, (list, tuple)) else str(fname)
        output, tgt = postprocess(preds, target)
        self.metric_vals[name][slice_num] = slice_metrics(output, tgt)
        preds = preds.detach()
        return name, slice_num, (preds.cpu().numpy() if self.to_numpy else preds)

    @staticmethod
    def save_outputs(outputs, out_dir):
        """base.py:575-587: the (fname, slice, output) triples of `test_step`, stacked per volume in slice order and written as
        `<out_dir>/reconstructions/<fname>` with one `reconstruction` dataset each (HDF5 through this package's writer)."""
        import os

        from mridc_amd.collections.common.parts.utils import save_reconstructions
        reconstructions = defaultdict(list)
        for fname, slice_num, output in outputs:
            reconstructions[fname].append((slice_num, output.detach().cpu().numpy() if torch.is_tensor(output) else np.asarray(output)))
        stacked = {fname: np.stack([out for _, out in sorted(v, key=lambda t: t[0])]) for fname, v in reconstructions.items()}
        save_reconstructions(stacked, os.path.join(str(out_dir), "reconstructions"))
        return stacked

    def test_epoch_end(self, device=None, outputs=None, out_dir=None):
        """base.py:490-517: mean over the slices of every volume, summed over volumes, all-reduced (sum) over the ranks together
        with the volume count, divided by the total count.  With `outputs` (the list of `test_step` results) and `out_dir`, the
        reconstructions are written as the reference does at the end of the same method (:575-587)."""
        if outputs is not None and out_dir is not None:
            self.save_outputs(outputs, out_dir)
        names = sorted(self.metric_vals)
        sums = torch.zeros(5, dtype=torch.float64)
        for fname in names:
            vals = torch.stack([v for _, v in sorted(self.metric_vals[fname].items())]).double().cpu()
            v = vals.mean(0)
            sums += torch.tensor([v[0], v[1], 1.0 - v[5], v[3], 1.0], dtype=torch.float64)
        tot = sharding.gather_metric_sums(sums.tolist(), device=device).cpu()
        n = float(tot[4]) if float(tot[4]) > 0 else 1.0
        return {"MSE": float(tot[0]) / n, "NMSE": float(tot[1]) / n, "SSIM": float(tot[2]) / n, "PSNR": float(tot[3]) / n,
                "TotExamples": int(tot[4])}
