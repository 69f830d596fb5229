"""GPU parity of the training-path backward kernels against torch autograd of the same fp32 operations on
the CPU (the reference trains through torch autograd: rim_block.py:217-249).  Tolerance: rel-L2 <= 1e-5 per operator."""
import pytest
import torch
import torch.nn.functional as F

from tests._util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref_conv(x, w, b, k, dil, pad_mode):
    p = dil * (k - 1) // 2
    xp = F.pad(x, (p, p, p, p), mode="replicate" if pad_mode == "replicate" else "constant")
    return F.conv2d(xp, w, b, dilation=dil)


CASES = [  # B, Cin, Cout, H, W, k, dil
    (1, 4, 64, 16, 32, 5, 1), (2, 64, 64, 13, 37, 3, 2), (1, 64, 64, 9, 70, 1, 1), (1, 64, 2, 12, 40, 3, 1), (1, 16, 64, 8, 33, 3, 1),
    (1, 24, 4, 7, 9, 3, 1), (1, 64, 64, 40, 372, 3, 2),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("pad_mode", ["replicate", "zero"])
def test_conv_backward_vs_autograd(dev, case, pad_mode):
    from mridc_amd import ops
    B, Cin, Cout, H, W, k, dil = case
    g = torch.Generator().manual_seed(H * W + Cin + k)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)).requires_grad_(True)
    dy = torch.randn(B, Cout, H, W, generator=g)
    _ref_conv(x, w, None, k, dil, pad_mode).backward(dy)
    pm = ops.PAD_REPLICATE if pad_mode == "replicate" else ops.PAD_ZERO
    dw = ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm)
    assert_close(dw, w.grad, 1e-5, f"weight gradient {case} {pad_mode}")
    acc = torch.full_like(dw, 0.5)
    ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm, out=acc, accumulate=True)
    assert_close(acc, w.grad + 0.5, 1e-5, "accumulating form")
    dx = ops.conv_dgrad(dy.to(dev), w.detach().to(dev), dil, pm)
    assert_close(dx, x.grad, 1e-5, f"data gradient {case} {pad_mode}")
    # determinism: the reductions have a fixed order
    assert torch.equal(dw, ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm))


@pytest.mark.parametrize("shape", [(2, 64, 13, 18), (1, 5, 9, 370), (1, 64, 40, 372)])
def test_relu_and_indrnn_backward_vs_autograd(dev, shape):
    from mridc_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B + C + H)
    pre = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    bias = torch.randn(C, generator=g, requires_grad=True)
    hh = torch.randn(1, C, 1, 1, generator=g, requires_grad=True)
    hp = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    dy = torch.randn(B, C, H, W, generator=g)
    y = F.relu(pre + bias.view(1, -1, 1, 1) + hh * hp)       # the IndRNN cell's output stage (rnn_cells.py:390)
    y.backward(dy)
    dpre, dhp, sums = ops.relu_bwd(dy.to(dev), y.detach().to(dev), hp.detach().to(dev), hh.detach().to(dev))
    assert_close(dpre, pre.grad, 1e-6, "dpre")
    assert_close(dhp, hp.grad, 1e-6, "dh_prev")
    assert_close(sums[:, 0], bias.grad, 1e-5, "bias gradient")
    assert_close(sums[:, 1], hh.grad.reshape(-1), 1e-5, "hh gradient")
    dpre2, none, sums2 = ops.relu_bwd(dy.to(dev), y.detach().to(dev))
    assert none is None and torch.equal(dpre2, dpre) and torch.equal(sums2[:, 0], sums[:, 0])
