"""Drop-in for `mridc.collections.reconstruction.models.sigmanet.sensitivity_net` (reference sensitivity_net.py:9-310): the complex
instance normalisation around a regulariser and the unrolled network x <- D(x - R(x)), inference path.

The statistics (a mean and a 2x2 pseudo-covariance per image: a handful of reductions) are torch reductions on the device; the
regulariser and the data layers they surround run on the HIP kernels.  Shapes are the reference's: a 4-D image [B,H,W,2] leaves the
wrapper as [B,B,H,W,2] through the broadcast of its [B,1,1,1] statistics (B = 1 in the reference's own use), the covariance divisor is
`shape[2] * shape[3] - 1` of whatever rank arrives."""
import numpy as np
import torch


def matrix_invert(xx, xy, yx, yy):
    """sensitivity_net.py:9-12."""
    det = xx * yy - xy * yx
    return yy.div(det), -xy.div(det), -yx.div(det), xx.div(det)


class ComplexInstanceNorm(torch.nn.Module):
    """sensitivity_net.py:15-118."""

    def __init__(self):
        super().__init__()
        self.mean = 0
        self.cov_xx_half = 1 / np.sqrt(2)
        self.cov_xy_half = 0
        self.cov_yx_half = 0
        self.cov_yy_half = 1 / np.sqrt(2)

    def complex_instance_norm(self, x, eps=1e-5):
        x_combined = torch.sum(x, dim=1, keepdim=True)
        mean = x_combined.mean(dim=(1, 2, 3), keepdim=True)
        self.mean = mean
        self.complex_pseudocovariance(x - mean)

    def complex_pseudocovariance(self, data):
        """Mean-free data in; sets the square root of the 2x2 covariance (closed-form eigen-decomposition)."""
        if data.size(-1) != 2:
            raise AssertionError
        shape = data.shape
        N = shape[2] * shape[3]
        re, im = torch.unbind(data, dim=-1)
        dim = list(range(1, len(shape) - 1))
        cxx = (re * re).sum(dim=dim, keepdim=True) / (N - 1)
        cyy = (im * im).sum(dim=dim, keepdim=True) / (N - 1)
        cxy = (re * im).sum(dim=dim, keepdim=True) / (N - 1)
        root = torch.sqrt((cxx + cyy) ** 2 / 4 - cxx * cyy + cxy ** 2)
        s1 = (cxx + cyy) / 2 - root
        s2 = (cxx + cyy) / 2 + root
        v1x, v1y, v2x, v2y = s1 - cyy, cxy, s2 - cyy, cxy
        norm1 = torch.sqrt(torch.sum(v1x * v1x + v1y * v1y, dim=dim, keepdim=True))
        norm2 = torch.sqrt(torch.sum(v2x * v2x + v2y * v2y, dim=dim, keepdim=True))
        v1x, v1y, v2x, v2y = v1x.div(norm1), v1y.div(norm1), v2x.div(norm2), v2y.div(norm2)
        det = v1x * v2y - v2x * v1y
        s1 = torch.sqrt(s1).div(det)
        s2 = torch.sqrt(s2).div(det)
        self.cov_xx_half = v1x * v2y * s1 - v1y * v2x * s2
        self.cov_yy_half = v1x * v2y * s2 - v1y * v2x * s1
        self.cov_xy_half = v1x * v2x * (s2 - s1)
        self.cov_yx_half = v1y * v2y * (s1 - s2)

    def forward(self, input):
        return self.normalize(input)

    def set_normalization(self, input):
        mean = torch.mean(input).reshape(1)          # stays on the device (the reference round-trips it through .item())
        self.complex_pseudocovariance(input - mean)
        self.mean = mean.unsqueeze(1).unsqueeze(1).unsqueeze(1)
        self.cov_xx_half = self.cov_xx_half.view(-1, 1, 1, 1)
        self.cov_xy_half = self.cov_xy_half.view(-1, 1, 1, 1)
        self.cov_yx_half = self.cov_yx_half.view(-1, 1, 1, 1)
        self.cov_yy_half = self.cov_yy_half.view(-1, 1, 1, 1)

    def normalize(self, x):
        x_m = x - self.mean
        re, im = torch.unbind(x_m, dim=-1)
        ixx, ixy, iyx, iyy = matrix_invert(self.cov_xx_half, self.cov_xy_half, self.cov_yx_half, self.cov_yy_half)
        img = torch.stack([ixx * re + ixy * im, iyx * re + iyy * im], dim=-1)
        return img.clamp(-6, 6)

    def unnormalize(self, x):
        re, im = torch.unbind(x, dim=-1)
        return torch.stack([self.cov_xx_half * re + self.cov_xy_half * im, self.cov_yx_half * re + self.cov_yy_half * im], dim=-1) + self.mean


class ComplexNormWrapper(torch.nn.Module):
    """sensitivity_net.py:121-139."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.complex_instance_norm = ComplexInstanceNorm()

    def forward(self, input):
        self.complex_instance_norm.set_normalization(input)
        output = self.complex_instance_norm.normalize(input)
        shp = output.shape
        output = output.reshape(shp[0] * shp[1], *shp[2:]).permute(0, 3, 1, 2)
        output = self.model(output)
        output = output.permute(0, 2, 3, 1).reshape(*shp)
        return self.complex_instance_norm.unnormalize(output)


class SensitivityNetwork(torch.nn.Module):
    """sensitivity_net.py:142-310: x <- gradD(x - gradR(x), y, smaps, mask), num_iter times."""

    def __init__(self, num_iter, model, datalayer, shared_params=True, save_space=False, reset_cache=False):
        super().__init__()
        self.shared_params = shared_params
        self.num_iter = 1 if self.shared_params else num_iter
        self.num_iter_total = num_iter
        self.is_trainable = [True] * num_iter
        self.gradR = torch.nn.ModuleList([ComplexNormWrapper(model) for _ in range(self.num_iter)])
        self.gradD = torch.nn.ModuleList([datalayer for _ in range(self.num_iter)])
        self.save_space = save_space
        self.reset_cache = reset_cache

    def _iterations(self):
        if self.shared_params:
            return self.num_iter_total
        return min(np.where(self.is_trainable)[0][-1] + 1, self.num_iter)

    def forward(self, x, y, smaps, mask):
        for i in range(self._iterations()):                # forward and forward_save_space compute the same values
            x_thalf = x - self.gradR[i % self.num_iter](x)
            x = self.gradD[i % self.num_iter](x_thalf, y, smaps, mask)
        return x

    forward_save_space = forward

    def freeze(self, i):
        for param in self.gradR[i].parameters():
            param.require_grad_ = False
        self.is_trainable[i] = False

    def unfreeze(self, i):
        for param in self.gradR[i].parameters():
            param.require_grad_ = True
        self.is_trainable[i] = True

    def freeze_all(self):
        for i in range(self.num_iter):
            self.freeze(i)

    def unfreeze_all(self):
        for i in range(self.num_iter):
            self.unfreeze(i)

    def copy_params(self, src_i, trg_j):
        for trg_param, src_param in zip(self.gradR[trg_j].parameters(), self.gradR[src_i].parameters()):
            trg_param.data.copy_(src_param.data)

    def stage_training_init(self):
        self.freeze_all()
        self.unfreeze(0)

    def stage_training_transition_i(self, copy=False):
        if self.shared_params:
            return
        if not np.all(self.is_trainable):
            for i in range(self.num_iter):
                if i == self.num_iter - 1:
                    self.unfreeze_all()
                    break
                if self.is_trainable[i]:
                    self.freeze(i)
                    self.unfreeze(i + 1)
                    if copy:
                        self.copy_params(i, i + 1)
                    break
