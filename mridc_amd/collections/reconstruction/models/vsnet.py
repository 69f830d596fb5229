"""Drop-in for `mridc.collections.reconstruction.models.vsnet.VSNet` (reference vsnet.py:23-167), inference path."""
import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd.collections.reconstruction.models import _cfg
from mridc_amd.collections.reconstruction.models.base import build_sens_net
from mridc_amd.collections.reconstruction.models.conv import conv2d
from mridc_amd.collections.reconstruction.models.unet_base import unet_block
from mridc_amd.collections.reconstruction.models.variablesplittingnet import vsnet_block

__all__ = ["VSNet"]


class VSNet(torch.nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        num_cascades = cfg_dict.get("num_cascades")
        self.fft_centered = cfg_dict.get("fft_centered")             # set by the reference's base class (models/base.py)
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.num_cascades = num_cascades
        self.use_sens_net = cfg_dict.get("use_sens_net")
        if self.use_sens_net:
            self.sens_net = build_sens_net(cfg_dict, self.fft_centered, self.fft_normalization, self.spatial_dims, self.coil_dim)
        arch = cfg_dict.get("imspace_model_architecture")
        if arch == "CONV":                                            # vsnet.py:45-52
            image_model = conv2d.Conv2d(in_channels=2, out_channels=2, hidden_channels=cfg_dict.get("imspace_conv_hidden_channels"),
                                        n_convs=cfg_dict.get("imspace_conv_n_convs"), batchnorm=cfg_dict.get("imspace_conv_batchnorm"))
        elif arch in ["UNET", "NORMUNET"]:                            # vsnet.py:61-70
            image_model = unet_block.NormUnet(cfg_dict.get("imspace_unet_num_filters"), cfg_dict.get("imspace_unet_num_pool_layers"),
                                              in_chans=2, out_chans=2, drop_prob=cfg_dict.get("imspace_unet_dropout_probability"),
                                              padding_size=cfg_dict.get("imspace_unet_padding_size"),
                                              normalize=cfg_dict.get("imspace_unet_normalize"))
        elif arch == "MWCNN":
            raise NotImplementedError("mridc_amd VSNet: the MWCNN denoiser is not on the HIP path (CONV and UNET are)")
        else:
            raise NotImplementedError("VSNet is currently implemented only with image_model_architecture == 'MWCNN' or 'UNet'."
                                      f"Got {arch}.")
        # one instance of each module, listed num_cascades times: the cascades share their weights (vsnet.py:81-83)
        image_model = torch.nn.ModuleList([image_model] * num_cascades)
        data_consistency_model = torch.nn.ModuleList([vsnet_block.DataConsistencyLayer()] * num_cascades)
        weighted_average_model = torch.nn.ModuleList([vsnet_block.WeightedAverageTerm()] * num_cascades)
        self.model = vsnet_block.VSNetBlock(denoiser_block=image_model, data_consistency_block=data_consistency_model,
                                            weighted_average_block=weighted_average_model, num_cascades=num_cascades,
                                            fft_centered=self.fft_centered, fft_normalization=self.fft_normalization,
                                            spatial_dims=self.spatial_dims, coil_dim=self.coil_dim)
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        self.train_loss_fn = _cfg.make_loss(cfg_dict.get("train_loss_fn", "l1"))
        self.val_loss_fn = _cfg.make_loss(cfg_dict.get("val_loss_fn", "l1"))
        self.accumulate_estimates = False

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, init_pred: torch.Tensor,
                target: torch.Tensor) -> torch.Tensor:
        """vsnet.py:115-167."""
        sensitivity_maps = self.sens_net(y, mask) if self.use_sens_net else sensitivity_maps
        # (no hybrid-space form here: the block adds the coil-combined IMAGE sx to K-SPACE, vsnet_block.py:145, which does not
        # commute with a transform along H)
        image = self.model(y, sensitivity_maps, mask)
        image = torch.view_as_complex(utils.coil_combination(
            fft.ifft2(image, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims),
            sensitivity_maps, method=self.coil_combination_method, dim=self.coil_dim))
        _, image = utils.center_crop_to_smallest(target, image)
        return image

    forward_step = forward
