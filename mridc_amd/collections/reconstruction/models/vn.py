"""Drop-in for `mridc.collections.reconstruction.models.vn.VarNet` (reference vn.py:22-142): inference on the fused HIP path, training
(gradients recorded) on the differentiable operator forms of `mridc_amd.diff`."""
import torch

from mridc_amd import diff, ops
import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd.collections.reconstruction.models import _cfg
from mridc_amd.collections.reconstruction.models.unet_base import unet_block
from mridc_amd.collections.reconstruction.models.varnet import vn_block

__all__ = ["VarNet"]

from mridc_amd.collections.reconstruction.models.base import build_sens_net


class VarNet(torch.nn.Module):
    # hybrid cascades at W = 372: each block's data-consistency pass also yields the next block's sens_reduce (MRIDC_AMD_CHAIN_REDUCE=0: off)
    chain_reduce = __import__("os").environ.get("MRIDC_AMD_CHAIN_REDUCE", "1") != "0"

    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        self.no_dc = cfg_dict.get("no_dc")
        self.fft_centered = cfg_dict.get("fft_centered")
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.use_sens_net = cfg_dict.get("use_sens_net")
        if self.use_sens_net:                                          # models/base.py:81-95 (applied by the caller's step, :234)
            self.sens_net = build_sens_net(cfg_dict, self.fft_centered, self.fft_normalization, self.spatial_dims, self.coil_dim)
        self.num_cascades = cfg_dict.get("num_cascades")
        self.cascades = torch.nn.ModuleList([
            vn_block.VarNetBlock(
                unet_block.NormUnet(chans=cfg_dict.get("channels"), num_pools=cfg_dict.get("pooling_layers"),
                                    padding_size=cfg_dict.get("padding_size"), normalize=cfg_dict.get("normalize")),
                fft_centered=self.fft_centered, fft_normalization=self.fft_normalization, spatial_dims=self.spatial_dims,
                coil_dim=self.coil_dim, no_dc=self.no_dc)
            for _ in range(self.num_cascades)])                      # vn.py:50-67
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        self.train_loss_fn = _cfg.make_loss(cfg_dict.get("train_loss_fn", "l1"))
        self.val_loss_fn = _cfg.make_loss(cfg_dict.get("val_loss_fn", "l1"))
        self.dc_weight = torch.nn.Parameter(torch.ones(1))          # vn.py:91
        self.accumulate_estimates = False
        # `trainer.precision` (base_vn_run.yaml:98: 16 = pytorch-lightning's native AMP around the forward pass) or a plain `precision` key of cfg: 16
        # selects the one-term fp16 convolutions of the regulariser at inference (mrx_unet_conv3x3_p16); None: the process default (MRIDC_AMD_PRECISION)
        prec = getattr(trainer, "precision", None) if trainer is not None else None
        self.precision = cfg_dict.get("precision", None) if prec is None else prec

    def _inference_precision(self):
        return ops.resolve_precision16(self.precision)

    def _hybrid_ok(self, mask):
        """Row-invariant (1-D column) mask + SENSE combination: the masked data consistency commutes with the H transform, so the
        cascades can run on IFFT_H(k) with row transforms only (`MRIDC_AMD_HYBRID=0` turns this off)."""
        import os
        return (os.environ.get("MRIDC_AMD_HYBRID", "1") != "0" and self.coil_dim == 1
                and str(self.coil_combination_method).upper() == "SENSE" and ops.mask_is_row_invariant(mask))

    def _forward_hybrid(self, y, sensitivity_maps, mask):
        """The cascades on kh = IFFT_H(k): every block is sens_reduce_rows -> regulariser -> sens_expand_rows -> dc_combine, and the
        SENSE combination of ifft2(k) at the end is one more sens_reduce_rows.  Same function as the k-space form, half the FFT work."""
        yh = ops.llg_prepare(y, self.fft_centered, self.fft_normalization, self.spatial_dims)
        est = yh
        red = None                               # sum_c conj(S) IFFT_W(est), handed from one block's data-consistency pass to the next block
        for cascade in self.cascades:
            cascade._hybrid, cascade._reduced_in, cascade._want_reduced = True, red, self.chain_reduce
            try:
                est = cascade(est, yh, sensitivity_maps, mask)
                red = cascade._reduced_out
            finally:
                cascade._hybrid, cascade._reduced_in, cascade._want_reduced, cascade._reduced_out = False, None, False, None
        if red is not None:                      # the SENSE combination of ifft2(k) (vn.py:125-142) is that same reduction
            return red
        return ops.sens_reduce(est, sensitivity_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=True)

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, init_pred: torch.Tensor,
                target: torch.Tensor) -> torch.Tensor:
        """vn.py:94-142."""
        if diff.active(y, *self.parameters(), training=self.training):                # training: k-space formulation on the differentiable forms
            estimation = y.clone()
            for cascade in self.cascades:
                estimation = cascade(estimation, y, sensitivity_maps, mask)
            estimation = diff.ifft2(estimation, self.fft_centered, self.fft_normalization, self.spatial_dims)
            estimation = diff.coil_combination(estimation, sensitivity_maps, method=self.coil_combination_method, dim=self.coil_dim)
        elif self._hybrid_ok(mask):
            with ops.inference_precision(self._inference_precision()):
                estimation = self._forward_hybrid(y, sensitivity_maps, mask)
        else:
            estimation = y.clone()
            with ops.inference_precision(self._inference_precision()):
                for cascade in self.cascades:
                    estimation = cascade(estimation, y, sensitivity_maps, mask)
            estimation = fft.ifft2(estimation, centered=self.fft_centered, normalization=self.fft_normalization,
                                   spatial_dims=self.spatial_dims)
            estimation = utils.coil_combination(estimation, sensitivity_maps, method=self.coil_combination_method,
                                                dim=self.coil_dim)
        estimation = torch.view_as_complex(estimation)
        _, estimation = utils.center_crop_to_smallest(target, estimation)
        return estimation

    forward_step = forward
