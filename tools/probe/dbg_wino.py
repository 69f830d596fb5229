import torch, sys
sys.path.insert(0, '/root/repo')
from mridc_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for (B, Cin, H, W) in [(1, 64, 8, 32), (1, 8, 8, 32), (1, 16, 8, 32), (1, 64, 16, 64), (1, 64, 640, 372)]:
    F = 64
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    wc = (torch.randn(F, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
    wi = (torch.randn(F, F, 1, 1, generator=g) / 8).to(dev)
    bc, bi, hh = torch.randn(F, generator=g).to(dev) * 0.1, torch.randn(F, generator=g).to(dev) * 0.1, torch.randn(1, F, 1, 1, generator=g).to(dev) * 0.5
    hp = torch.randn(B, F, H, W, generator=g).to(dev)
    d = ops.rim_layer_indrnn_packed(x, ops.rim_layer_pack(wc, wi), F, 3, 2, bc, bi, hh, hp)
    pk = ops.rim_layer_wino_pack(wc, wi)
    for rep in range(3):
        w = ops.rim_layer_indrnn_wino(x, pk, F, bc, bi, hh, hp)
        err = (w - d).abs()
        bad = err > 1e-3 * d.abs().max()
        print((B, Cin, H, W), 'rep', rep, 'rel', ((w - d).norm() / d.norm()).item(), 'bad frac', bad.float().mean().item())
        if bad.any() and H * W <= 64 * 64:
            bm = bad[0].any(0)
            print('bad rows', bm.any(1).nonzero().flatten().tolist()[:40], 'bad cols', bm.any(0).nonzero().flatten().tolist()[:40])
            print('bad channels', bad[0].flatten(1).any(1).nonzero().flatten().tolist())
