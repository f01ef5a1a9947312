// Kernel-side aliases of the public argument structs (include/cdnet_hip.h).
#pragma once
#include "../../include/cdnet_hip.h"

struct ConvSrc {
    const unsigned short *x;
    const unsigned short *res;
    const float *scale;
    const float *shift;
    int C, Hs, Ws, pool, relu, off_y, off_x;
    int f16, row_stride;
};

struct ConvArgs {
    ConvSrc src[2];
    int nsrc;
    const unsigned short *w;
    const float *bias;
    const float *oscale;
    const float *oshift;
    int orelu;
    unsigned short *out;
    int Cout, out_cstride, out_coff;
    float *stats;
    int N, H, W;
    int taps, npar, ostride, nchunk;
    int tile, CK, BN;
    int out_f16;
    int debug;
    int ws;
    int f32;
    const unsigned short *eres;
    const float *eres_scale, *eres_shift;
    int eres_f16, eres_relu;
    int taps1, pad_;
    unsigned short *pool_out;
    const float *dot_w, *dot_b;
    float *dot_out;
};

static_assert(sizeof(ConvSrc) == sizeof(cdnet_conv_src), "ConvSrc layout");
static_assert(sizeof(ConvArgs) == sizeof(cdnet_conv_args), "ConvArgs layout");

namespace cdnet {
// conv32.hip: the fp32-storage / split-bf16x3 variant
int conv_forward_f32(const ConvArgs &A, hipStream_t st);
// conv32ws.hip: its wave-specialised persistent form (3x3, full 16x16 tiles, >= 4 chunks); -1 = not eligible
int conv_forward_f32_ws(const ConvArgs &A, hipStream_t st, bool dry_run = false);
// conv16ws.hip: the 16-bit path's persistent kernel for launches without statistics (any chunk count, direct stores); -1 = not eligible
int conv_forward_ws16(const ConvArgs &A, hipStream_t st, bool dry_run = false);
int materialize_f32(const ConvSrc &s, int N, int H, int W, void *out, hipStream_t st);
}
