// rim_layer1_sb.h -- internal interface of the split-bf16 first RIM layer (rim_layer1_sb.hip), used by rim_layer.hip's C entry points.
#pragma once
#include <hip/hip_runtime.h>

#define MRX_L1SB_PACK_FLOATS 28164   // three bf16 terms of the 5x5 (28 taps x 4 channels) and 1x1 weights in A-operand lane order (16896 floats), then the same
                                     // weights as two scaled fp16 terms (7 x 2 x 2 x 64 + 4 x 2 x 2 x 64 operands) and one header element with the two scale exponents

struct MrxL1sbArgs {
    const float* x;        // [B,Cin,H,W], Cin <= 4 (unused when eta2 is set)
    const float* packed;   // mrx_l1sb_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const float* hprev;    // [B,64,H,W] or null
    float* hnew;           // [B,64,H,W]
    int B, Cin, H, W, tiles_x, ntiles;
    const float2* eta2;    // [B,H,W] complex or null: input = (eta, post * sum_k part_k), the gradient's last pass done by the tile loader
    const float2* part;    // [nparts][B][H][W] complex
    long long part_stride;
    int nparts;
    float post;
    int f16;               // 1: two-term fp16 operands (per-unit / per-pixel scales), 0: three-term bf16
    int cb8;               // 1: h_prev / h_new channel-blocked [B][8][H][W][8], 0: [B,64,H,W]
    unsigned* xmax;        // not null: atomic max of the bits of every output (outputs are >= 0: ReLU) -- the bound mrx_rim_layer2_f16 scales by
};

int mrx_l1sb_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, hipStream_t st);
#define MRX_L1SB_TH 16   // image rows per workgroup tile (ntiles / tiles_x are counted in 16 x 32 tiles)
int mrx_l1sb_launch(const MrxL1sbArgs& a, hipStream_t st);
