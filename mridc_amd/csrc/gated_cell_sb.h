// gated_cell_sb.h -- internal interface of the split-bf16 gated cell (gated_cell_sb.hip), used by gated_cell.hip's C entry points.
#pragma once
#include <hip/hip_runtime.h>

// three bf16 terms of the 2 * gates 64 x 64 matrices in A-operand lane order, then the same matrices as two fp16 terms scaled by one power of two,
// then a header element holding that exponent
#define MRX_GATED_SB_PACK_FLOATS(gates) ((2 * (gates) * 2 * 4 * 5 * 64 + 1) * 4)

struct MrxGatedSbArgs {
    const float* x;       // [B,64,P]
    const float* h;       // [B,64,P] or null (= zeros)
    const float* packed;  // mrx_gated_sb_pack
    const float* b_ih;    // [GATES*64] or null
    float* out;           // [B,64,P]
    long long P, nsegb, nseg;  // pixels per image, 32-pixel segments per image, segments in total
    float* xmax;          // or null: max |out| is folded into this device scalar (atomic max, never reset here: mrx_conv3x3_sb_chain's xmax_in)
};

int mrx_gated_sb_pack(const float* w_ih, const float* w_hh, float* packed, int gates, hipStream_t st);
int mrx_gated_sb_launch(const MrxGatedSbArgs& a, int gates, hipStream_t st);

#define MRX_CONV2DGRU_SB_PACK_FLOATS ((6 * 2 * 4 * 5 * 64 + 1) * 4)   // three bf16 terms, two scaled fp16 terms, header (the scale exponent)

struct MrxConv2dGruSbArgs {
    const float* x;       // [B,64,P] layer input (after its conv + ReLU)
    const float* h;       // [B,64,P] previous state of this layer or null (= zeros)
    const float* packed;  // mrx_conv2dgru_sb_pack
    const float* bias;    // [3][64]: update, reset, out
    float* out;           // [B,64,P] new state
    float* out_relu;      // [B,64,P] ReLU(new state) or null
    long long P, nsegb, nseg;
    float* xmax;          // or null: max of out_relu is folded into this device scalar (atomic max, never reset here: mrx_conv3x3_sb_chain's xmax_in)
};

int mrx_conv2dgru_sb_pack(const float* wu, const float* wr, const float* wo, float* packed, hipStream_t st);
int mrx_conv2dgru_sb_launch(const MrxConv2dGruSbArgs& a, hipStream_t st);

// three bf16 terms of W in A-operand lane order, then (round 6) W rounded to fp16 once in the same order: the precision-16 form
#define MRX_CONV1X1_SB128_PACK_FLOATS ((2 * 2 * 2 * 4 * 3 * 64 + 2 * 2 * 2 * 4 * 64) * 4)

struct MrxConv1x1SbArgs {
    const float* x;       // [B,128,P]
    const float* packed;  // mrx_conv1x1_sb128_pack
    const float* bias;    // [128] or null
    const float* hh;      // [128] or null
    const float* hprev;   // [B,128,P] or null
    float* out;           // [B,128,P]
    long long P, nsegb, nseg;
    int act;
    float slope;
    int head;             // 1: only the first 64 output channels, out [B,64,P]
    int p16;              // 1: the reference's `precision: 16` arithmetic (x and W rounded to fp16 once -- as torch.autocast casts them, no scale --, fp32 sums)
    float* xmax;          // or null: device scalar, max |out| is folded in with an atomic max (never reset here): the bound a two-term fp16
                          // consumer of `out` scales its operands by (mrx_conv3x3_h)
};

int mrx_conv1x1_sb128_pack(const float* w, float* packed, hipStream_t st);
int mrx_conv1x1_sb128_launch(const MrxConv1x1SbArgs& a, hipStream_t st);
