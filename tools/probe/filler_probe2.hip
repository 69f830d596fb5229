// Probe 2: marginal cost (cycles of fp32-MFMA issue per SIMD) of one extra instruction of each kind, 2 waves/SIMD, 16x16x4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>
__global__ __launch_bounds__(512, 2) void k_probe(const float* in, float* out, unsigned long long* ticks, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 4095];
    __syncthreads();
    float a = in[tid], b = in[tid + 512];
    float f[8];
    f32x2 g[8];
    f32x4 h4[4];
    for (int i = 0; i < 8; ++i) { f[i] = in[tid + i]; g[i] = (f32x2){in[tid + i], in[tid + 8 + i]}; }
    for (int i = 0; i < 4; ++i) h4[i] = (f32x4){0, 0, 0, 0};
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float* lp = lds + (tid & 63) * 4;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int o = ((i * 4 + k) * 256) % 6144;
                if (KIND == 0) f[k & 7] = f[k & 7] + a;
                if (KIND == 1) g[k & 7] = g[k & 7] + g[(k + 1) & 7];                        // v_pk_add_f32
                if (KIND == 2) f[k & 7] += lp[o];                                            // ds_read_b32 (+ v_add)
                if (KIND == 3) g[k & 7] += *reinterpret_cast<const f32x2*>(lp + o);          // ds_read_b64 (+ pk add)
                if (KIND == 4) h4[k & 3] += *reinterpret_cast<const f32x4*>(lp + o);         // ds_read_b128 (+ 2 pk add)
                if (KIND == 5) lp[o] = f[k & 7];                                             // ds_write_b32
                if (KIND == 6) *reinterpret_cast<f32x2*>(lp + o) = g[k & 7];                 // ds_write_b64
                if (KIND == 7) *reinterpret_cast<f32x4*>(lp + o) = h4[k & 3];                // ds_write_b128
                if (KIND == 8) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");             // SALU
                if (KIND == 9) f[k & 7] = __builtin_fmaf(f[k & 7], a, b);                    // v_fma_f32
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i] + g[i][0] + g[i][1];
    for (int i = 0; i < 4; ++i) s += h4[i][0] + h4[i][1] + h4[i][2] + h4[i][3];
    out[blockIdx.x * 512 + tid] = s + lds[tid];
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

static const char* kinds[] = {"v_add_f32", "v_pk_add_f32", "ds_read_b32+v_add", "ds_read_b64+v_pk_add", "ds_read_b128+2 v_pk_add", "ds_write_b32",
                              "ds_write_b64", "ds_write_b128", "s_add_u32", "v_fma_f32"};
template <int K, int KIND>
double run(float* d_in, float* d_out, unsigned long long* d_t) {
    const int blocks = 256, iters = 300;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_probe<K, KIND>), dim3(blocks), dim3(512), 0, 0, d_in, d_out, d_t, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mt = 0;
    for (auto v : h) mt += (double)v;
    return mt / blocks / ((double)iters * 16 * 2);
}
template <int KIND>
void kind(float* d_in, float* d_out, unsigned long long* d_t) {
    const double c0 = run<0, KIND>(d_in, d_out, d_t), c1 = run<1, KIND>(d_in, d_out, d_t), c2 = run<2, KIND>(d_in, d_out, d_t), c4 = run<4, KIND>(d_in, d_out, d_t);
    printf("%-26s cycles/MFMA/SIMD with 0,1,2,4 per MFMA: %5.1f %5.1f %5.1f %5.1f   marginal per instruction at 2: %.1f, at 4: %.1f\n", kinds[KIND], c0, c1, c2, c4,
           (c2 - c0) / 2, (c4 - c0) / 4);
}
int main() {
    float *d_in, *d_out;
    unsigned long long* d_t;
    hipMalloc(&d_in, 8192 * 4);
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_t, 4096 * 8);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(d_in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    kind<0>(d_in, d_out, d_t); kind<1>(d_in, d_out, d_t); kind<9>(d_in, d_out, d_t); kind<2>(d_in, d_out, d_t); kind<3>(d_in, d_out, d_t); kind<4>(d_in, d_out, d_t);
    kind<5>(d_in, d_out, d_t); kind<6>(d_in, d_out, d_t); kind<7>(d_in, d_out, d_t); kind<8>(d_in, d_out, d_t);
    return 0;
}
