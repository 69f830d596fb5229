"""Round-off of a Winograd F(2x2, 3x3) form of the dilated 3x3 64 -> 64 convolution on TWO-TERM fp16 operands (three term products, fp32 accumulation)
against the direct form on the same operands and against float64 -- numpy emulation, CPU only.  The dilated convolution is four interleaved unit-dilation
convolutions; one of them is emulated here (the others are the same arithmetic).  Input transform in fp32 BEFORE the split (V = B^T d B), weight transform in
float64 then split (U = G g G^T), products per transform position accumulated in fp32 over the 64 input channels, output transform (A^T M A) in fp32."""
import numpy as np

rng = np.random.default_rng(0)
C, K, H, W = 64, 64, 24, 24


def split2(x, scale_pow):
    """x (fp32) * 2^scale_pow as two fp16 terms (values returned in fp32, unscaled by the caller)."""
    s = np.float32(2.0 ** scale_pow)
    xs = (x * s).astype(np.float32)
    h1 = xs.astype(np.float16).astype(np.float32)
    h2 = (xs - h1).astype(np.float16).astype(np.float32)
    return h1, h2


def pow_for(m):
    # the kernels' rule: scale so that the maximum lands below 2^15 (headroom for fp16's range), power of two
    return int(np.floor(14 - np.log2(max(m, 1e-30))))


def mm3(a1, a2, b1, b2):
    """three term products, fp32 accumulation over the contraction axis (einsum in float32 is accumulated in float32 pairwise; emulate with float32 matmul)"""
    f = np.float32
    return (a1.astype(f) @ b2.astype(f)) + (a2.astype(f) @ b1.astype(f)) + (a1.astype(f) @ b1.astype(f))


x = np.maximum(rng.standard_normal((C, H, W)), 0).astype(np.float32) * 3.0      # ReLU states
w = (rng.standard_normal((K, C, 3, 3)) / 24).astype(np.float32)
# float64 reference (valid convolution)
ref = np.zeros((K, H - 2, W - 2))
for dy in range(3):
    for dx in range(3):
        ref += np.einsum("kc,chw->khw", w[:, :, dy, dx].astype(np.float64), x[:, dy:dy + H - 2, dx:dx + W - 2].astype(np.float64))

# direct form on two-term operands
px, pw = pow_for(np.abs(x).max()), pow_for(np.abs(w).max())
x1, x2 = split2(x, px)
w1, w2 = split2(w, pw)
acc = np.zeros((K, H - 2, W - 2), np.float32)
for dy in range(3):
    for dx in range(3):
        a1, a2 = w1[:, :, dy, dx], w2[:, :, dy, dx]
        b1 = x1[:, dy:dy + H - 2, dx:dx + W - 2].reshape(C, -1)
        b2 = x2[:, dy:dy + H - 2, dx:dx + W - 2].reshape(C, -1)
        acc += mm3(a1, a2, b1, b2).reshape(K, H - 2, W - 2)
direct = acc * np.float32(2.0 ** (-px - pw))

# Winograd F(2x2, 3x3)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
U = np.einsum("ai,kcij,bj->kcab", G, w.astype(np.float64), G)                  # [K, C, 4, 4] float64
pu = pow_for(np.abs(U).max())
U1, U2 = split2(U.astype(np.float32), pu)                                      # (the fp32 rounding of U is part of the scheme)
th, tw = (H - 2) // 2, (W - 2) // 2
tiles = np.stack([x[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] for i in range(th) for j in range(tw)], 1)   # [C, T, 4, 4]
V = np.einsum("ai,ctij,bj->ctab", Bt, tiles, Bt).astype(np.float32)            # fp32 adds
pv = pow_for(np.abs(V).max())
V1, V2 = split2(V, pv)
M = np.zeros((K, th * tw, 4, 4), np.float32)
for a in range(4):
    for b in range(4):
        M[:, :, a, b] = mm3(U1[:, :, a, b], U2[:, :, a, b], V1[:, :, a, b], V2[:, :, a, b])
M *= np.float32(2.0 ** (-pu - pv))
Y = np.einsum("ai,ktij,bj->ktab", At, M, At).astype(np.float32)                # [K, T, 2, 2]
wino = np.zeros((K, H - 2, W - 2), np.float32)
t = 0
for i in range(th):
    for j in range(tw):
        wino[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2] = Y[:, t]
        t += 1
# plain fp32 direct (what the fp32-MFMA kernels do)
f32 = np.zeros((K, H - 2, W - 2), np.float32)
for dy in range(3):
    for dx in range(3):
        f32 += (w[:, :, dy, dx] @ x[:, dy:dy + H - 2, dx:dx + W - 2].reshape(C, -1)).reshape(K, H - 2, W - 2)
e = lambda a: float(np.linalg.norm(a.astype(np.float64) - ref) / np.linalg.norm(ref))  # noqa: E731
print(f"rel-L2 against float64:  fp32 direct {e(f32):.2e}   two-term fp16 direct {e(direct):.2e}   two-term fp16 Winograd F(2x2,3x3) {e(wino):.2e}")
print(f"max |V| / max |x| = {np.abs(V).max() / np.abs(x).max():.2f}   max |U| / max |w| = {np.abs(U).max() / np.abs(w).max():.2f}")
