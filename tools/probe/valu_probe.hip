// Probe: fp32 VALU issue rates on gfx950 -- v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32, 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(const float* in, float* out, int iters) {
    float c = in[0], d = in[1];
    float a[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = in[2 + i] + threadIdx.x;
    for (int i = 0; i < 8; ++i) p[i] = (f2){a[2 * i], a[2 * i + 1]};
    f2 cc = (f2){c, c}, dd = (f2){d, d};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
        } else if (MODE == 3) {   // fma with an SGPR multiplier (the DFT's constant coefficients)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(d));
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 5) {   // packed fma with an SGPR-pair multiplier
            unsigned long long cs = ((unsigned long long)__float_as_uint(c) << 32) | __float_as_uint(c);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(cs), "v"(dd));
        } else if (MODE == 6) {   // fma with a literal constant
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[i]) : "v"(d));
        } else if (MODE == 7) {   // fma with an inline constant (0.5)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(a[i]) : "v"(d));
        } else {   // v_mul with sgpr
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int instr_per_iter, int lanes_ops, float* din, float* dout) {
    const int iters = 4000, blocks = 256, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, din, dout, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * instr_per_iter * waves_per_simd;
    printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s-equivalent\n", name,
           waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4,
           instr_per_simd * 1024 * 64 * lanes_ops / (ms * 1e-3) / 1e12);
}

int main() {
    float *din, *dout;
    hipMalloc(&din, 4096);
    hipMemset(din, 0, 4096);
    hipMalloc(&dout, 256 * 1024 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32", w, 64, 2, din, dout);
        run<1>("v_pk_fma_f32", w, 32, 4, din, dout);
        run<2>("v_pk_add_f32", w, 32, 2, din, dout);
        run<3>("v_fmac sgpr", w, 64, 2, din, dout);
        run<4>("v_add_f32", w, 64, 1, din, dout);
        run<5>("pk_fma sgpr", w, 32, 4, din, dout);
        run<6>("fmaak literal", w, 64, 2, din, dout);
        run<7>("fma inline .5", w, 64, 2, din, dout);
        run<8>("v_mul sgpr", w, 64, 1, din, dout);
    }
    return 0;
}
