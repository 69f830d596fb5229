"""Drop-in for `mridc.collections.reconstruction.models.ccnn.CascadeNet` (reference ccnn.py:22-142), inference path."""
import torch

from mridc_amd import ops
import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd.collections.reconstruction.models import _cfg
from mridc_amd.collections.reconstruction.models.cascadenet import ccnn_block
from mridc_amd.collections.reconstruction.models.conv import conv2d

__all__ = ["CascadeNet"]


class CascadeNet(torch.nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        self.fft_centered = cfg_dict.get("fft_centered")
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.cascades = torch.nn.ModuleList([
            ccnn_block.CascadeNetBlock(
                conv2d.Conv2d(in_channels=2, out_channels=2, hidden_channels=cfg_dict.get("hidden_channels"),
                              n_convs=cfg_dict.get("n_convs"), batchnorm=cfg_dict.get("batchnorm")),
                fft_centered=self.fft_centered, fft_normalization=self.fft_normalization, spatial_dims=self.spatial_dims,
                coil_dim=self.coil_dim, no_dc=cfg_dict.get("no_dc"))
            for _ in range(cfg_dict.get("num_cascades"))])             # ccnn.py:47-67
        self.train_loss_fn = _cfg.make_loss(cfg_dict.get("train_loss_fn", "l1"))
        self.val_loss_fn = _cfg.make_loss(cfg_dict.get("val_loss_fn", "l1"))
        self.accumulate_estimates = False
        self.dc_weight = torch.nn.Parameter(torch.ones(1))            # ccnn.py:90


    chain_reduce = __import__("os").environ.get("MRIDC_AMD_CHAIN_REDUCE", "1") != "0"

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
        red = None                               # sum_c conj(S) IFFT_W(est), handed from one block's last pass to the next block (W = 372)
        for cascade in self.cascades:
            cascade._hybrid, cascade._reduced_in, cascade._want_reduced = True, red, self.chain_reduce
            try:
                est = cascade(est, yh, sensitivity_maps, mask)
                red = cascade._reduced_out
            finally:
                cascade._hybrid, cascade._reduced_in, cascade._want_reduced, cascade._reduced_out = False, None, False, None
        if red is not None:
            return red
        return ops.sens_reduce(est, sensitivity_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=True)

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, init_pred: torch.Tensor,
                target: torch.Tensor) -> torch.Tensor:
        """ccnn.py:93-142."""
        if self._hybrid_ok(mask):
            pred = torch.view_as_complex(self._forward_hybrid(y, sensitivity_maps, mask))
        else:
            pred = y.clone()
            for cascade in self.cascades:
                pred = cascade(pred, y, sensitivity_maps, mask)
            pred = fft.ifft2(pred, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims)
            pred = torch.view_as_complex(utils.coil_combination(pred, sensitivity_maps, method=self.coil_combination_method,
                                                                dim=self.coil_dim))
        _, pred = utils.center_crop_to_smallest(target, pred)
        return pred

    forward_step = forward
