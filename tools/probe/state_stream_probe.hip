// state_stream_probe.hip -- round 6: how fast can the fp16 hidden state of the precision-16 route (8 slices x 64 channels x 640 x 372) be read and written back with
// the access pattern of the layer kernels (persistent workgroups on 16 x 32 tiles, a wave = two image rows, lane = pixel + channel sub-block, next tile's loads in
// flight), (a) as [B][8][H][W][8] halves: 8 planes x 8 bytes per lane = 512-byte pieces, (b) as [B][4][H][W][16]: 4 planes x 16 bytes per lane = 1 KB pieces?
// hipcc --offload-arch=gfx950 -O3 tools/probe/state_stream_probe.hip -o tools/probe/state_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int NP, class V, bool NT>   // NP planes per pixel row piece, V = the per-lane vector
__global__ __launch_bounds__(512, 1) void k_stream(const V* __restrict__ in, V* __restrict__ out, int B, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int tiles_x = (W + 31) / 32, ntiles = tiles_x * (H / 16), total = ntiles * B;
    const long long plane = (long long)H * W;      // pixels
    V cur[2][NP], nxt[2][NP];
    auto addr = [&](int t, int rw, int q) {
        const int b = t / ntiles, tile = t - b * ntiles, ty0 = tile / tiles_x, w0 = (tile - ty0 * tiles_x) * 32;
        const int oy = ty0 * 16 + 2 * wave + rw, ox = w0 + l31 < W ? w0 + l31 : W - 1;
        return ((long long)(b * NP + q) * plane + (long long)oy * W + ox) * 2 + lhi;      // two V per pixel and plane
    };
    auto load = [&](int t, V (&v)[2][NP]) {
        const int tc = t < total ? t : total - 1;
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int q = 0; q < NP; ++q) v[rw][q] = NT ? __builtin_nontemporal_load(in + addr(tc, rw, q)) : in[addr(tc, rw, q)];
    };
    load(blockIdx.x, cur);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        load(t + gridDim.x, nxt);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                V v = cur[rw][q];
                v[0] ^= 0x3c00u;
                if (NT) __builtin_nontemporal_store(v, out + addr(t, rw, q));
                else out[addr(t, rw, q)] = v;
            }
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int q = 0; q < NP; ++q) cur[rw][q] = nxt[rw][q];
    }
}

int main() {
    const int B = 8, H = 640, W = 372;
    const size_t bytes = (size_t)B * 64 * H * W * 2;
    void *a, *b;
    hipMalloc(&a, bytes), hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-52s %7.1f us  %6.0f GB/s read + written (%.2f of 8 TB/s)\n", name, best * 1e3, 2.0 * bytes / best / 1e6, 2.0 * bytes / best / 1e6 / 8000.0);
    };
    run("[B][8][H][W][8]  8 B per lane, out of place", [&] { hipLaunchKernelGGL((k_stream<8, u32x2, false>), dim3(256), dim3(512), 0, 0, (const u32x2*)a, (u32x2*)b, B, H, W); });
    run("[B][8][H][W][8]  8 B per lane, in place", [&] { hipLaunchKernelGGL((k_stream<8, u32x2, false>), dim3(256), dim3(512), 0, 0, (const u32x2*)a, (u32x2*)a, B, H, W); });
    run("[B][8][H][W][8]  8 B per lane, in place, nt", [&] { hipLaunchKernelGGL((k_stream<8, u32x2, true>), dim3(256), dim3(512), 0, 0, (const u32x2*)a, (u32x2*)a, B, H, W); });
    run("[B][4][H][W][16] 16 B per lane, out of place", [&] { hipLaunchKernelGGL((k_stream<4, u32x4, false>), dim3(256), dim3(512), 0, 0, (const u32x4*)a, (u32x4*)b, B, H, W); });
    run("[B][4][H][W][16] 16 B per lane, in place", [&] { hipLaunchKernelGGL((k_stream<4, u32x4, false>), dim3(256), dim3(512), 0, 0, (const u32x4*)a, (u32x4*)a, B, H, W); });
    run("[B][4][H][W][16] 16 B per lane, in place, nt", [&] { hipLaunchKernelGGL((k_stream<4, u32x4, true>), dim3(256), dim3(512), 0, 0, (const u32x4*)a, (u32x4*)a, B, H, W); });
    return 0;
}
