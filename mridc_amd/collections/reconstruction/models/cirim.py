"""Drop-in for `mridc.collections.reconstruction.models.cirim.CIRIM` (reference cirim.py:25-249), inference path.

The reference class subclasses a pytorch-lightning base (training/validation harness, out of scope: SURVEY 8a row H);
this mirror is a plain nn.Module with the same constructor config keys, parameter names (`cirim.{i}.*`, `dc_weight`)
and `forward` contract (a generator yielding list[cascade][time_step] of complex [B,h,w]).
"""
import math
from typing import Generator, Union

import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd import ops
from mridc_amd.collections.reconstruction.models import _cfg
from mridc_amd.collections.reconstruction.models.rim import rim_block

__all__ = ["CIRIM"]

from mridc_amd.collections.reconstruction.models.base import build_sens_net


class CIRIM(torch.nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        self.recurrent_filters = cfg_dict.get("recurrent_filters")
        # make time-steps size divisible by 8 (cirim.py:50-51)
        self.time_steps = 8 * math.ceil(cfg_dict.get("time_steps") / 8)
        self.no_dc = cfg_dict.get("no_dc")
        self.fft_centered = cfg_dict.get("fft_centered")
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.use_sens_net = cfg_dict.get("use_sens_net")
        if self.use_sens_net:                                          # models/base.py:81-95 (applied by the caller's step, :234)
            self.sens_net = build_sens_net(cfg_dict, self.fft_centered, self.fft_normalization, self.spatial_dims, self.coil_dim)
        self.num_cascades = cfg_dict.get("num_cascades")
        self.cirim = torch.nn.ModuleList([
            rim_block.RIMBlock(
                recurrent_layer=cfg_dict.get("recurrent_layer"), conv_filters=cfg_dict.get("conv_filters"),
                conv_kernels=cfg_dict.get("conv_kernels"), conv_dilations=cfg_dict.get("conv_dilations"),
                conv_bias=cfg_dict.get("conv_bias"), recurrent_filters=self.recurrent_filters,
                recurrent_kernels=cfg_dict.get("recurrent_kernels"), recurrent_dilations=cfg_dict.get("recurrent_dilations"),
                recurrent_bias=cfg_dict.get("recurrent_bias"), depth=cfg_dict.get("depth"), time_steps=self.time_steps,
                conv_dim=cfg_dict.get("conv_dim"), no_dc=self.no_dc, fft_centered=self.fft_centered,
                fft_normalization=self.fft_normalization, spatial_dims=self.spatial_dims, coil_dim=self.coil_dim,
                dimensionality=cfg_dict.get("dimensionality"))
            for _ in range(self.num_cascades)])                      # cirim.py:60-84
        # the reference hands `trainer.precision` to pytorch-lightning (base_cirim_run.yaml:132: 16 = native AMP around forward); here the trainer
        # (or a plain `precision` key of cfg) selects the precision-16 kernels of the RIM blocks.  None: the process default (MRIDC_AMD_PRECISION).
        prec = getattr(trainer, "precision", None) if trainer is not None else None
        prec = cfg_dict.get("precision", None) if prec is None else prec
        if prec is not None:
            for blk in self.cirim:
                blk.precision = prec
        self.keep_eta = cfg_dict.get("keep_eta")
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        # cirim.py:91-93 applies rnn_weights_init, a no-op for conv layers (common/parts/rnn_utils.py:21-32)
        self.train_loss_fn = _cfg.make_loss(cfg_dict.get("train_loss_fn", "l1"))
        self.val_loss_fn = _cfg.make_loss(cfg_dict.get("val_loss_fn", "l1"))
        self.dc_weight = torch.nn.Parameter(torch.ones(1))          # cirim.py:112
        self.accumulate_estimates = True

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, init_pred: torch.Tensor,
                target: torch.Tensor) -> Union[Generator, torch.Tensor]:
        """cirim.py:115-165."""
        prediction = y.clone()
        init_pred = None if init_pred is None or init_pred.dim() < 4 else init_pred
        hx = None
        sigma = 1.0
        cascades_etas = []
        # hybrid-space data of the one-launch gradient (row-invariant masks): the same for every cascade, computed once
        hybrid = None
        if ops.mask_is_row_invariant(mask) and self.coil_dim == 1:
            hybrid = ops.llg_prepare(y, self.fft_centered, self.fft_normalization, self.spatial_dims)
            m1 = ops.row_invariant_view(mask)
            if ops.llg372_supported(hybrid, m1):            # W = 372: maps, data and mask in the lane order of mrx_llg372, once per slice
                hybrid = (hybrid, ops.llg372_prepare(hybrid, sensitivity_maps, m1, self.fft_centered, self.fft_normalization))
        for i, cascade in enumerate(self.cirim):
            prediction, _ = cascade(prediction, y, sensitivity_maps, mask, init_pred, hx, sigma,
                                    keep_eta=False if i == 0 else self.keep_eta, _hybrid=hybrid, _want_hx=False)
            time_steps_etas = [self.process_intermediate_pred(pred, sensitivity_maps, target) for pred in prediction]
            cascades_etas.append(time_steps_etas)
        yield cascades_etas

    # fastMRI/ATOMMIC-style alias named by BASELINE.json
    def forward_step(self, y, sensitivity_maps, mask, init_pred=None, target=None):
        return next(self.forward(y, sensitivity_maps, mask, init_pred, target))

    def process_loss(self, target, pred, _loss_fn=None, mask=None):
        """cirim.py:199-249: the (validation / training) loss of the estimates against the target for the three configurable losses
        (`l1`, `mse`, `ssim`: cirim.py:95-110), evaluated on the device by libmridc_amd (mrx_absl1_loss, mrx_recon_metrics,
        mrx_ssim_loss) -- a generator, like the reference.  With accumulate_estimates every time-step loss is multiplied by the whole
        logspace(-1, 0, time_steps) vector and summed (the reference's weighting), cascades are averaged.  Gradients of the l1 form are
        what mridc_amd.training back-propagates; this method itself records no tape."""
        from mridc_amd import _lib
        from mridc_amd import runner
        _loss_fn = self.train_loss_fn if _loss_fn is None else _loss_fn
        kind = "ssim" if "ssim" in str(_loss_fn).lower() else ("mse" if "mse" in str(_loss_fn).lower() else "l1")
        tgt = runner._magnitude(target)
        tgt = ops.div_by_device_scalar(tgt, ops.max_abs(tgt))                  # |target / max |target||
        L = _lib.lib()

        def loss_fn(y):
            yc = y if y.is_complex() else torch.view_as_complex(_lib.f32c(y))  # estimates are complex [B,h,w] (or their [..., 2] view)
            y = _lib.f32c(torch.view_as_real(yc))
            if kind == "l1":
                m = ops.max_abs(y, complex_modulus=True).reshape(1)
                out2 = torch.empty(2, dtype=torch.float32, device=y.device)
                work = torch.empty(int(L.mrx_absl1_work_floats()), dtype=torch.float32, device=y.device)
                _lib.check(L.mrx_absl1_loss(_lib.ptr(y), _lib.ptr(tgt), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(work), tgt.numel(),
                                            _lib.stream_ptr()), "mrx_absl1_loss")
                return out2[0]
            yn = runner._magnitude(yc)
            yn = ops.div_by_device_scalar(yn, ops.max_abs(yn))
            if kind == "mse":
                return ops.recon_metrics(tgt, yn)[0]
            x4, y4 = tgt.unsqueeze(self.coil_dim), yn.unsqueeze(self.coil_dim)   # [B,1,h,w]
            return _loss_fn(x4, y4, data_range=ops.max_abs(tgt).reshape(1))

        if self.accumulate_estimates:
            w = float(torch.logspace(-1, 0, steps=self.time_steps).sum()) / self.time_steps
            cascades_loss = []
            for cascade_pred in pred:
                terms = torch.stack([loss_fn(p) for p in cascade_pred])
                cascades_loss.append(terms.sum() * w)
            yield sum(cascades_loss) / len(self.cirim)
        else:
            yield loss_fn(pred)

    def process_intermediate_pred(self, pred, sensitivity_maps, target, do_coil_combination=False):
        """cirim.py:167-197."""
        if not self.no_dc or do_coil_combination:
            pred = fft.ifft2(pred, centered=self.fft_centered, normalization=self.fft_normalization,
                             spatial_dims=self.spatial_dims)
            pred = utils.coil_combination(pred, sensitivity_maps, method=self.coil_combination_method, dim=self.coil_dim)
        pred = torch.view_as_complex(pred)
        _, pred = utils.center_crop_to_smallest(target, pred)
        return pred
