// Implicit-GEMM convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), NHWC bf16 activations,
// fp32 accumulation.  One kernel family serves
//   3x3 pad 1 convolution   (models/dam/model_unet_rev1.py: VGG16-BN encoder :40-41, UpsampleBlock.conv2 :112-114,
//                            ResidualUnit :146-170; models/unet.py encoder/decoder :8-50)
//   1x1 convolution         (ResidualUnit.conv_1x1 :158)
//   ConvTranspose2d k4 s2 p1 as four 2x2 sub-pixel convolutions (UpsampleBlock.up :100-101)
//   ConvTranspose2d k2 s2   as four 1x1 sub-pixel convolutions  (models/unet.py decoder.up :30)
// and, with transposed/flipped weight packs, their backward-data passes.
//
// Tiling: one workgroup (4 waves) = TH x TW output pixels x BN output channels.  The (TH+2)x(TW+2) input halo of a
// CK-channel chunk is staged through LDS once and reused by all taps; the chunk's weights are staged in the exact
// MFMA B-fragment order (a linear copy of the host-side pack).  While staging, the producer layer's BatchNorm+ReLU
// (per-channel scale/shift), an optional residual add, a 2x2 max-pool and the decoder's zero-pad + concat are
// applied on the fly, so none of those ever makes its own pass over HBM.  The epilogue either applies a folded
// eval-mode BN (+ReLU) or emits the raw convolution output together with per-tile channel sums / sums of squares
// (training-mode BN statistics, reduced deterministically by bn_finalize).
#include <type_traits>
#include <vector>
#include "common.h"
#include "conv_args.h"
#include "xform.h"
#include "stage16.h"
#include <stdlib.h>

using namespace cdnet;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

__device__ int g_dbg_dummy;

// ------------------------------------------------------------------------------------------------------
// weight packing: fp32 master weights -> bf16 MFMA B-fragment order
//   packed[cout_tile][chunk][tap][kc][half][col BN][8]   value = w(cout = tile*BN+col, cin = chunk*CK+kc*16+half*8+j, tap)
// mode 0: Conv2d weight [Cout][Cin][KH][KW], forward       tap = kh*KW+kw
// mode 1: Conv2d weight, backward-data: roles of cin/cout swap, taps flip: "cout" := original cin,
//         "cin" := original cout, tap (kh,kw) := original (KH-1-kh, KW-1-kw)
// mode 2: ConvTranspose2d weight [Cin][Cout][4][4] forward, sub-pixel parity (a,b) in `parity`:
//         tap t=(ty,tx) in 2x2: rows {a==0: kh 1 (dy 0), kh 3 (dy -1); a==1: kh 0 (dy +1), kh 2 (dy 0)}
// mode 3: ConvTranspose2d weight [Cin][Cout][2][2] forward, parity (a,b): single tap (kh=a, kw=b)
// Channels beyond the real Cin/Cout (padding to CK / BN multiples) are zero.
// ------------------------------------------------------------------------------------------------------
struct PackDesc {            // one (layer, sub-pixel parity) packing job
    const float *w;
    unsigned short *out;
    int Cout, Cin, KH, KW, CK, BN, nchunk, ntile, TAPS, mode, parity;
    unsigned block0, nblocks; // its slice of the batched launch
    int split;                // 1: fp32-precision pack - per chunk the bf16(w) image followed by the bf16(w - bf16(w)) image
    const float *scale;       // optional per-output-channel multiplier applied in fp32 before the rounding (eval-mode BatchNorm fold; mode 0)
};

__device__ __forceinline__ void pack_range(const PackDesc &d, size_t first, size_t stride) {
    const float *__restrict__ w = d.w;
    unsigned short *__restrict__ out = d.out;
    const int Cout = d.Cout, Cin = d.Cin, KH = d.KH, KW = d.KW, CK = d.CK, BN = d.BN, nchunk = d.nchunk, TAPS = d.TAPS, mode = d.mode,
              parity = d.parity;
    // one thread = the 8 consecutive input channels of one (tile, chunk, tap, k-half, column): the index is decoded once per
    // 16-byte vector (decoding every element made this kernel 0.1 ms of integer divisions on the step's critical chain)
    const size_t total = (size_t)d.ntile * nchunk * TAPS * (CK / 16) * 2 * BN * (d.split ? 2 : 1);
    for (size_t i = first; i < total; i += stride) {
        // work order: the tap runs fastest - neighbouring lanes then read neighbouring floats of the [..][kh][kw] weight tensors (a wave's
        // load touches ~8 cache lines instead of 64; the scattered side is the 16-byte stores, an eighth as many) - the pack layout is
        // [tile][chunk][hi|lo][tap][k-chunk][k-half][column][8 channels]
        size_t r = i;
        int tap = r % TAPS; r /= TAPS;
        int col = r % BN; r /= BN;
        int half = r % 2; r /= 2;
        int kc = r % (CK / 16); r /= (CK / 16);
        int hl = 0;
        if (d.split) { hl = r % 2; r /= 2; }
        int chunk = r % nchunk; r /= nchunk;
        int tile = (int)r;
        int co = tile * BN + col;
        const size_t ovec = ((((((size_t)tile * nchunk + chunk) * (d.split ? 2 : 1) + hl) * TAPS + tap) * (CK / 16) + kc) * 2 + half) * BN + col;
        unsigned short o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
        int ci = chunk * CK + kc * 16 + half * 8 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            if (mode == 0) {
                v = w[(((size_t)co * Cin + ci) * KH + tap / KW) * KW + tap % KW];
            } else if (mode == 1) {
                int kh = KH - 1 - tap / KW, kw = KW - 1 - tap % KW;
                // here Cout/Cin are the ROLES in the backward GEMM: co indexes original cin, ci original cout
                v = w[(((size_t)ci * Cout + co) * KH + kh) * KW + kw];
            } else if (mode == 2) {
                int a = parity >> 1, b = parity & 1, ty = tap >> 1, tx = tap & 1;
                int kh = a == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2);
                int kw = b == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 0 : 2);
                v = w[(((size_t)ci * Cout + co) * 4 + kh) * 4 + kw];
            } else if (mode == 3) {
                int a = parity >> 1, b = parity & 1;
                v = w[(((size_t)ci * Cout + co) * 2 + a) * 2 + b];
            } else if (mode == 7) {
                // backward-data of mode 6: GEMM co = (a, b, c) of the view (Cout = 4*Ct), ci = the conv's out channel,
                // taps flipped: dV[y'][x'][(a,b,c)] = sum_t dY[y' - dy_t][x' - dx_t][ci] * W[ci][c][ky][kx]
                const int Ct = Cout / 4;
                const int a = co / (2 * Ct), b = (co / Ct) & 1, c = co % Ct;
                const int tf = 8 - tap;
                const int kh = 2 * (tf / 3 - 1) + a + 1, kw = 2 * (tf % 3 - 1) + b + 1;
                if (kh >= 0 && kh < 3 && kw >= 0 && kw < 3) v = w[(((size_t)ci * Ct + c) * 3 + kh) * 3 + kw];
            } else {
                // modes 4/5: backward-data of a stride-2 transposed convolution as a convolution over the space-to-depth
                // view of the output gradient.  GEMM "cin" ci = a*(2*Ct) + b*Ct + c (row parity a = source, column parity
                // b, channel c of the transposed conv's Ct out_channels); GEMM "cout" co = its in_channel.
                const int Ct = Cin / 4;
                const int a = ci / (2 * Ct), b = (ci / Ct) & 1, c = ci % Ct;
                if (mode == 4) {
                    const int kh = 2 * (tap / 3) + a - 1, kw = 2 * (tap % 3) + b - 1;      // 3x3 taps (ky, kx)
                    if (kh >= 0 && kh < 4 && kw >= 0 && kw < 4) v = w[(((size_t)co * Ct + c) * 4 + kh) * 4 + kw];
                } else if (mode == 5) {
                    v = w[(((size_t)co * Ct + c) * 2 + a) * 2 + b];
                } else {
                    // mode 6: FORWARD 3x3 stride-2 pad-1 convolution [Cout][Ct][3][3] as a 3x3 convolution over the
                    // space-to-depth view of its input: in(2y+ky-1, 2x+kx-1) = view(y+dy, x+dx; parity a, b) with
                    // ky = 2*dy + a + 1 (tap row r = dy + 1), only dy in {-1, 0} contribute
                    const int kh = 2 * (tap / 3 - 1) + a + 1, kw = 2 * (tap % 3 - 1) + b + 1;
                    if (kh >= 0 && kh < 3 && kw >= 0 && kw < 3) v = w[(((size_t)co * Ct + c) * 3 + kh) * 3 + kw];
                }
            }
        }
        if (d.scale && co < Cout) v *= d.scale[co];
        const unsigned short hi = f2bf(v);
        o8[j] = hl ? f2bf(v - bf2f(hi)) : hi;
        }
        uint4 ov;
        ov.x = o8[0] | ((unsigned)o8[1] << 16); ov.y = o8[2] | ((unsigned)o8[3] << 16);
        ov.z = o8[4] | ((unsigned)o8[5] << 16); ov.w = o8[6] | ((unsigned)o8[7] << 16);
        reinterpret_cast<uint4 *>(out)[ovec] = ov;
    }
}

__global__ void pack_weights_kernel(PackDesc d) {
    pack_range(d, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// every layer's forward and backward-data packs in ONE launch (the training step re-packs ~80 small tensors after each
// Adam update; separate launches cost more than the work)
__global__ void pack_weights_batch_kernel(const PackDesc *__restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                          // last descriptor whose block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = table[lo];
    pack_range(d, (size_t)(blockIdx.x - d.block0) * blockDim.x + threadIdx.x, (size_t)d.nblocks * blockDim.x);
}

// ------------------------------------------------------------------------------------------------------
// input staging
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ V16 max8(V16 a, V16 b, bool nonneg) {
    V16 o;
    if (nonneg) {                                // post-ReLU values: integer order == float order
        o.u = __builtin_bit_cast(uint4, xf_max_nonneg_bf8(__builtin_bit_cast(xf_u32x4, a.u), __builtin_bit_cast(xf_u32x4, b.u)));
        return o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h[j] = bf2f(a.h[j]) >= bf2f(b.h[j]) ? a.h[j] : b.h[j];
    return o;
}

// the 8 channels' scale / shift of one thread; `xf` is the LDS copy of the source's tables ([0] scale, [xfs] shift) made
// once per workgroup (a global load here would sit in front of the chunk pipeline: vmcnt retires in order), or null
// for callers without one
constexpr int XF_MAX = 2048;      // source channels (both concat sources) whose scale/shift fit the LDS table

__device__ __forceinline__ void load_chan_xf(ChanXf &t, const ConvSrc &s, const float *xf, int xfs, int c) {
    t.on = s.scale != nullptr;
    if (!t.on) return;
    const float4 *ps = reinterpret_cast<const float4 *>(xf ? xf + c : s.scale + c);
    const float4 *ph = reinterpret_cast<const float4 *>(xf ? xf + xfs + c : s.shift + c);
    float4 a = ps[0], b = ps[1], cc = ph[0], d = ph[1];
    t.sc[0] = a.x; t.sc[1] = a.y; t.sc[2] = a.z; t.sc[3] = a.w; t.sc[4] = b.x; t.sc[5] = b.y; t.sc[6] = b.z; t.sc[7] = b.w;
    t.sh[0] = cc.x; t.sh[1] = cc.y; t.sh[2] = cc.z; t.sh[3] = cc.w; t.sh[4] = d.x; t.sh[5] = d.y; t.sh[6] = d.z; t.sh[7] = d.w;
}

// Halo image of one K chunk in LDS.  16x16-pixel tiles with 16-channel chunks: 32 B per pixel, unpadded, the two 16-byte k-halves of
// a pixel at (half ^ (halo row & 1)) * 16 - an A-fragment ds_read_b128 serves 16 lanes per LDS cycle (8 pixels of one tile row and 8
// of the next, same k-half), and the row-parity swizzle puts them on 16 distinct 16-byte bank groups (tools/micro/lds_read_patterns.hip:
// 4 LDS cycles per read against 7-8 for 48-byte padded pixels).  Other tile shapes keep padded pixels (CK * 2 + 16 bytes).
template <int TH, int CK>
struct HaloImg {
    static constexpr bool SWZ = (TH == 16 && CK == 16);
    static constexpr int PSTR = SWZ ? CK * 2 : CK * 2 + 16;
    // byte offset of 16-byte slot `slot` of halo pixel `pix` in halo row `hy`
    __device__ __forceinline__ static int off(int pix, int hy, int slot) { return pix * PSTR + (SWZ ? ((slot ^ (hy & 1)) * 16) : slot * 16); }
};

template <int TH, int TW, int CK>
__device__ __forceinline__ void stage_input(const ConvSrc &s, int cc0, int n, int y0, int x0, int H, int W,
                                            unsigned char *lds_a, int tid, const float *xf = nullptr, int xfs = 0) {
    constexpr int VPP = CK / 8;                  // 16-byte vectors per pixel
    constexpr int HW_ = TW + 2, NPIX = (TH + 2) * (TW + 2);
    const int slot = tid % VPP;                  // constant per thread because 256 % VPP == 0
    ChanXf t;
    load_chan_xf(t, s, xf, xfs, cc0 + slot * 8);
    const bool relu = s.relu != 0;
    const bool f16 = s.f16 != 0;
    const bool plain = !t.on && !relu && s.res == nullptr && !f16;
    // logical source extent (after the optional 2x2 pool)
    // pool: 1 = floor mode (torchvision VGG 'M'), 2 = ceil mode (models/unet.py:19, partial windows at the edge)
    const int Hl = s.pool ? (s.Hs + (s.pool == 2)) / 2 : s.Hs, Wl = s.pool ? (s.Ws + (s.pool == 2)) / 2 : s.Ws;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;      // elements per row
    const size_t img = (size_t)n * s.Hs * rs;
    for (int v = tid; v < NPIX * VPP; v += 256) {
        const int pix = v / VPP;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        V16 val;
        val.u = make_uint4(0, 0, 0, 0);
        const int ys = y - s.off_y, xs = x - s.off_x;
        if (y >= 0 && y < H && x >= 0 && x < W && ys >= 0 && ys < Hl && xs >= 0 && xs < Wl) {
            if (!s.pool) {
                const size_t e = img + (size_t)ys * rs + (size_t)xs * s.C + cc0 + slot * 8;
                V16 raw;
                raw.u = *reinterpret_cast<const uint4 *>(s.x + e);
                if (plain) val = raw;
                else if (s.res) { V16 r; r.u = *reinterpret_cast<const uint4 *>(s.res + e); val = xform8(raw, &r, t, relu, f16); }
                else val = xform8(raw, nullptr, t, relu, f16);
            } else {
                // maxpool 2x2 stride 2 of the transformed source (torchvision VGG 'M' layers / unet.py :19)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int yy = 2 * ys + (q >> 1), xx = 2 * xs + (q & 1);
                    if (q != 0 && (yy >= s.Hs || xx >= s.Ws)) continue;          // ceil-mode partial window
                    const size_t e = img + (size_t)yy * rs + (size_t)xx * s.C + cc0 + slot * 8;
                    V16 raw;
                    raw.u = *reinterpret_cast<const uint4 *>(s.x + e);
                    V16 tv = plain ? raw : xform8(raw, nullptr, t, relu, f16);
                    val = q == 0 ? tv : max8(val, tv, relu);
                }
            }
        }
        *reinterpret_cast<uint4 *>(lds_a + HaloImg<TH, CK>::off(pix, hy, slot)) = val.u;
    }
}

// ------------------------------------------------------------------------------------------------------
// software pipeline: the global loads of chunk c+1 (input halo + weights) are ISSUED before the MFMA loop of chunk c
// and only transformed / written to LDS after it, so their HBM/L2 latency hides under the matrix work (guide T14).
// Pooled sources (4 loads + max per element) keep the synchronous path.
// ------------------------------------------------------------------------------------------------------
#ifndef CDNET_CONV_XCD
#define CDNET_CONV_XCD 1
#endif
#ifndef CDNET_CONV_GLDS
#define CDNET_CONV_GLDS 1
#endif
#ifndef CDNET_CONV_ADEPTH
#define CDNET_CONV_ADEPTH 1
#endif
// LDS layout of conv_fwd_kernel (dynamic): [A halo tile][B chunk (x2 when the weights arrive by LDS-DMA)] aliased by the
// out tile + the BatchNorm partial sums, followed by the scale|shift table of the source channels.
template <int TH, int TW, int CK, int BN, int TAPS>
struct ConvLds {
    static constexpr bool DEEP = (CK == 16 && TH == 16);      // 16-channel chunks: prefetch ring of ADEPTH halo chunks
    static constexpr bool GLDS = CDNET_CONV_GLDS && DEEP;     // weights by global_load_lds into a double buffer
    static constexpr int ADEPTH = DEEP ? CDNET_CONV_ADEPTH : 1;
    static constexpr int PSTR = HaloImg<TH, CK>::PSTR;
    static constexpr int A_BYTES = (TH + 2) * (TW + 2) * PSTR;
    static constexpr int B_BYTES = TAPS * CK * BN * 2;
    static constexpr int STAGE = A_BYTES + (GLDS ? 2 : 1) * B_BYTES;
    static constexpr int OUT_BYTES = TH * TW * (BN * 2 + 8);
    static constexpr int STATS_BYTES = 4 * 2 * BN * 4;
    static constexpr int MAIN = ((STAGE > OUT_BYTES + STATS_BYTES ? STAGE : OUT_BYTES + STATS_BYTES) + 15) / 16 * 16;
    static int bytes(int ctot) { return MAIN + 2 * ((ctot + 7) / 8 * 8) * 4; }
};

template <int TH, int TW, int CK, int BN, int TAPS>
struct Prefetch {
    static constexpr int VPP = CK / 8;
    static constexpr int NPIX = (TH + 2) * (TW + 2);
    static constexpr int NA = (NPIX * VPP + 255) / 256;
    static constexpr bool GLDS = ConvLds<TH, TW, CK, BN, TAPS>::GLDS;
    static constexpr int NB = (TAPS * CK * BN * 2 / 16 + 255) / 256;
    static constexpr int ADEPTH = ConvLds<TH, TW, CK, BN, TAPS>::ADEPTH;
    struct ASlot {
        uint4 a[NA];
        unsigned valid;      // snapshot of the source's tvalid taken at issue time
        bool pooled;         // synchronous path at commit time
    };
    ASlot s[ADEPTH];         // ring of halo-chunk prefetches: chunk c lives in slot c % ADEPTH
    uint4 b[GLDS ? 1 : NB];
    // per (tile, source) staging geometry, computed once by prep_source and reused by every chunk of that source:
    int eoff[2][NA];         // element offset of the thread's i-th vector inside the image (channel 0 of its slot), -1 = zero fill
    unsigned tvalid[2];      // bit i: eoff[i] >= 0
};

template <int SI, int TH, int TW, int CK, int BN, int TAPS>
__device__ __forceinline__ void prep_source(Prefetch<TH, TW, CK, BN, TAPS> &P, const ConvSrc &s, int y0, int x0, int H, int W, int tid) {
    using PF = Prefetch<TH, TW, CK, BN, TAPS>;
    constexpr int HW_ = TW + 2;
    const int slot = tid % PF::VPP;
    const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
    P.tvalid[SI] = 0;
#pragma unroll
    for (int i = 0; i < PF::NA; ++i) {
        const int v = tid + i * 256;
        const int pix = v / PF::VPP;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const int ys = y - s.off_y, xs = x - s.off_x;
        const bool inr = v < PF::NPIX * PF::VPP;
        const bool ok = inr && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (unsigned)ys < (unsigned)s.Hs && (unsigned)xs < (unsigned)s.Ws;
        P.eoff[SI][i] = ok ? ys * rs + xs * s.C + slot * 8 : -1;
        P.tvalid[SI] |= (ok ? 1u : 0u) << i;
    }
}

template <int TH, int TW, int CK, int BN, int TAPS>
__device__ __forceinline__ void issue_b(Prefetch<TH, TW, CK, BN, TAPS> &P, const unsigned short *wchunk, int tid, unsigned char *lds_b_next) {
    using PF = Prefetch<TH, TW, CK, BN, TAPS>;
    if (PF::GLDS) {
        // weights: the packed chunk is the LDS image - LDS-DMA, one 1 KB wave-instruction per 64 vectors, no registers.
        // (destination = wave-uniform base + lane * 16; drained by the vmcnt(0) of the barrier before the next commit)
        constexpr int NVEC = TAPS * CK * BN * 2 / 16;
        static_assert(NVEC % 64 == 0, "whole wave-instructions");
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int i = 0; i < PF::NB; ++i) {
            const int v0 = (i * 4 + wave) * 64;                          // first vector of this wave-instruction
            if (v0 < NVEC)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wchunk + (size_t)(v0 + lane) * 8),
                                                 (__attribute__((address_space(3))) void *)(lds_b_next + v0 * 16), 16, 0, 0);
        }
    } else {   // weights: linear, through registers
        const uint4 *src = reinterpret_cast<const uint4 *>(wchunk);
#pragma unroll
        for (int i = 0; i < PF::NB; ++i) {
            const int v = tid + i * 256;
            if (v < TAPS * CK * BN * 2 / 16) P.b[i] = src[v];
        }
    }
}

// halo chunk of source `si` (channels cc0..cc0+CK) -> the registers of one ring slot
template <typename PF>
__device__ __forceinline__ void issue_a(PF &P, typename PF::ASlot &S, const ConvSrc &s, int si, int cc0, int n) {
    S.pooled = s.pool != 0;
    S.valid = 0;
    if (S.pooled) return;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    const unsigned short *base = s.x + (size_t)n * s.Hs * rs + cc0;
    S.valid = si ? P.tvalid[1] : P.tvalid[0];
#pragma unroll
    for (int i = 0; i < PF::NA; ++i) {
        const int e = si ? P.eoff[1][i] : P.eoff[0][i];
        if (e >= 0) S.a[i] = *reinterpret_cast<const uint4 *>(base + e);
    }
}

template <int TH, int TW, int CK, int BN, int TAPS>
__device__ __forceinline__ void commit_chunk(const Prefetch<TH, TW, CK, BN, TAPS> &P, const typename Prefetch<TH, TW, CK, BN, TAPS>::ASlot &S,
                                             const ConvSrc &s, int si, int cc0, int n, int y0,
                                             int x0, int H, int W, unsigned char *lds_a, unsigned char *lds_b, int tid,
                                             const float *xf, int xfs) {
    using PF = Prefetch<TH, TW, CK, BN, TAPS>;
    if (!PF::GLDS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_b);
#pragma unroll
        for (int i = 0; i < PF::NB; ++i) {
            const int v = tid + i * 256;
            if (v < TAPS * CK * BN * 2 / 16) dst[v] = P.b[i];
        }
    }
    if (S.pooled) { stage_input<TH, TW, CK>(s, cc0, n, y0, x0, H, W, lds_a, tid, xf, xfs); return; }
    const int slot = tid % PF::VPP;
    ChanXf t;
    load_chan_xf(t, s, xf, xfs, cc0 + slot * 8);
    const bool relu = s.relu != 0, f16 = s.f16 != 0;
    const bool plain = !t.on && !relu && s.res == nullptr && !f16;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    // the residual operand is read here (not prefetched: registers are better spent on more waves per SIMD); the
    // loads of all vectors are issued before the first use
    V16 rr[PF::NA];
    if (s.res) {
        const unsigned short *rbase = s.res + (size_t)n * s.Hs * rs + cc0;
#pragma unroll
        for (int i = 0; i < PF::NA; ++i) rr[i].u = *reinterpret_cast<const uint4 *>(rbase + ((S.valid & (1u << i)) ? (si ? P.eoff[1][i] : P.eoff[0][i]) : 0));
    }
    // LDS address of vector i: pixel (tid / VPP + i * 256 / VPP), slot tid % VPP
#pragma unroll
    for (int i = 0; i < PF::NA; ++i) {
        if (tid + i * 256 >= PF::NPIX * PF::VPP) continue;
        V16 val;
        val.u = make_uint4(0, 0, 0, 0);
        if (S.valid & (1u << i)) {
            V16 raw;
            raw.u = S.a[i];
            if (plain) val = raw;
            else if (s.res) val = xform8(raw, &rr[i], t, relu, f16);
            else val = xform8(raw, nullptr, t, relu, f16);
        }
        const int pix = (tid + i * 256) / PF::VPP;
        *reinterpret_cast<uint4 *>(lds_a + HaloImg<TH, CK>::off(pix, pix / (TW + 2), slot)) = val.u;
    }
}

// ------------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------------
template <int TH, int TW, int CK, int BN, int WM, int WN, int TAPS>
__global__ __launch_bounds__(256, (BN <= 64 ? (CK == 16 && TH == 16 && CDNET_CONV_ADEPTH <= 2 ? 3 : 2) : 1)) void conv_fwd_kernel(ConvArgs A) {
    constexpr int PSTR = HaloImg<TH, CK>::PSTR;
    constexpr bool SWZ = HaloImg<TH, CK>::SWZ;
    constexpr int HW_ = TW + 2;
    constexpr int KC = CK / 16;
    constexpr int MT = TH * TW / 32, NT = BN / 32;
    constexpr int MPW = MT / WM, NPW = NT / WN;
    using LDS = ConvLds<TH, TW, CK, BN, TAPS>;
    constexpr bool GLDS = LDS::GLDS;
    constexpr int A_BYTES = LDS::A_BYTES;
    constexpr int B_BYTES = LDS::B_BYTES;
    constexpr int OSTR = BN * 2 + 8;             // out staging row stride (bytes)
    static_assert(MT % WM == 0 && NT % WN == 0 && WM * WN == 4, "wave tiling");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_a = smem;
    unsigned char *lds_b0 = smem + A_BYTES;
    float (*s_stats)[2][BN] = reinterpret_cast<float (*)[2][BN]>(smem + LDS::OUT_BYTES);      // epilogue only
    float *s_xf = reinterpret_cast<float *>(smem + LDS::MAIN);          // scale | shift of source 0 then source 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int xfs = ((A.src[0].C + (A.nsrc > 1 ? A.src[1].C : 0)) + 7) / 8 * 8;      // table stride (floats)
    {
        const int c0n = A.src[0].C, ctot = c0n + (A.nsrc > 1 ? A.src[1].C : 0);
        for (int c = tid; c < ctot; c += 256) {
            const ConvSrc &S = c < c0n ? A.src[0] : A.src[1];
            const int cc = c < c0n ? c : c - c0n;
            s_xf[c] = S.scale ? S.scale[cc] : 1.f;
            s_xf[xfs + c] = S.shift ? S.shift[cc] : 0.f;
        }
        // (published by the barrier in front of the first commit)
    }

    // one workgroup per (image, parity, tile_y, tile_x) tile
    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total_tiles = A.N * A.npar * tiles_img;
    const int nchunk_total = A.nchunk;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in dispatch order (x fastest).  XCD k takes
    // the k-th contiguous eighth of the (tile, cout block) sequence with the cout blocks of one tile next to each other: neighbouring
    // tiles - which share halo rows / columns - and the cout blocks of one tile - which read the same input - sit behind one L2 at about
    // the same time (a 64 -> 256 1x1 layer read its input from HBM four times when the cout blocks were dealt a whole grid apart)
#if CDNET_CONV_XCD
    int tile, cout_tile;
    {
        const int NC = (int)gridDim.y, lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        const int T = (int)gridDim.x * NC, q = T >> 3, rem = T & 7;
        const int xcd = lin & 7, idx = lin >> 3;
        const int seq = xcd < rem ? xcd * (q + 1) + idx : rem * (q + 1) + (xcd - rem) * q + idx;
        tile = seq / NC;
        cout_tile = seq - tile * NC;
        if (A.debug & 16) { tile = blockIdx.x; cout_tile = blockIdx.y; }          // ablation (CDNET_CONV_DEBUG=16): dispatch order
    }
#else
    const int tile = blockIdx.x, cout_tile = blockIdx.y;
#endif

    // per-lane A base for each of this wave's M tiles
    // (swizzled image: the k-half position follows the parity of the halo row = tile row + the tap's row offset)
    int abase[MPW][SWZ ? 2 : 1];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int m = (wm * MPW + mi) * 32 + l31;
        const int py = m / TW, px = m % TW;
#pragma unroll
        for (int rp = 0; rp < (SWZ ? 2 : 1); ++rp) abase[mi][rp] = (py * HW_ + px) * PSTR + (SWZ ? ((half ^ ((py + rp) & 1)) * 16) : half * 16);
    }
    const int bbase = half * BN * 16 + (wn * NPW * 32 + l31) * 16;

    // chunk c -> (source, first channel)
    auto chunk_src = [&](int c, int &si, int &cc0) {
        const int n0 = A.src[0].C / CK;
        if (c < n0) { si = 0; cc0 = c * CK; } else { si = 1; cc0 = (c - n0) * CK; }
    };
    auto decode = [&](int tile, int &n, int &par, int &y0, int &x0) {
        const int z = tile / tiles_img, r = tile - z * tiles_img;
        n = z / A.npar; par = z - n * A.npar;
        const int ty_ = r / tiles_x;
        y0 = ty_ * TH; x0 = (r - ty_ * tiles_x) * TW;
    };
    auto wchunk = [&](int par, int chunk) {
        return A.w + ((size_t)(par * gridDim.y + cout_tile) * nchunk_total + chunk) * (B_BYTES / 2);
    };

    Prefetch<TH, TW, CK, BN, TAPS> P;
    (void)total_tiles;
    {
        int n, par, y0, x0;
        decode(tile, n, par, y0, x0);
        using PF = Prefetch<TH, TW, CK, BN, TAPS>;
        constexpr int AD = PF::ADEPTH;
        prep_source<0>(P, A.src[0], y0, x0, A.H, A.W, tid);
        if (A.nsrc > 1) prep_source<1>(P, A.src[1], y0, x0, A.H, A.W, tid);
        // prologue: the first AD halo chunks and the first weight chunk
        issue_b(P, wchunk(par, 0), tid, lds_b0);
#pragma unroll
        for (int k = 0; k < AD; ++k)
            if (k < nchunk_total) {
                int si, cc0;
                chunk_src(k, si, cc0);
                issue_a(P, P.s[k], A.src[si], si, cc0, n);
            }
        // tap offsets inside the halo tile (rows, cols): 3x3 / 1x1 fixed, sub-pixel 2x2 depends on the parity
        int toff[TAPS], tpar[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            int r, c;
            if (TAPS == 9) { r = t / 3; c = t % 3; }
            else if (TAPS == 4) {
                const int a = par >> 1, b = par & 1, ty = t >> 1, tx = t & 1;
                r = a == 0 ? (ty == 0 ? 1 : 0) : (ty == 0 ? 2 : 1);
                c = b == 0 ? (tx == 0 ? 1 : 0) : (tx == 0 ? 2 : 1);
            } else { r = 1; c = 1; }
            toff[t] = (r * HW_ + c) * PSTR;
            tpar[t] = SWZ ? (r & 1) : 0;
        }
        f32x16 acc[MPW][NPW];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // one chunk step; K = ring slot of this chunk (compile-time so that the prefetch registers stay registers)
        auto step = [&](auto kc_, int chunk) {
            constexpr int K = decltype(kc_)::value;
            int si, cc0;
            chunk_src(chunk, si, cc0);
            __syncthreads();                               // previous chunk's fragment reads / previous tile's out-tile reads are done
            unsigned char *lds_b = lds_b0 + (GLDS ? (chunk & 1) * B_BYTES : 0);
            commit_chunk<TH, TW, CK, BN, TAPS>(P, P.s[K], A.src[si], si, cc0, n, y0, x0, A.H, A.W, lds_a, lds_b, tid, s_xf + (si ? A.src[0].C : 0), xfs);
            __syncthreads();
            // refill: the weights of the next chunk and the halo chunk AD steps ahead fly during the MFMA loops
            if (chunk + 1 < nchunk_total) issue_b(P, wchunk(par, chunk + 1), tid, lds_b0 + ((chunk + 1) & 1) * B_BYTES);
            if (chunk + AD < nchunk_total) {
                int sj, cj;
                chunk_src(chunk + AD, sj, cj);
                issue_a(P, P.s[K], A.src[sj], sj, cj, n);
            }
            if (A.debug & 4) return;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    bf16x8 af[MPW], bfr[NPW];
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi)
                        af[mi] = *reinterpret_cast<const bf16x8 *>(lds_a + (SWZ ? (tpar[t] ? abase[mi][SWZ ? 1 : 0] : abase[mi][0]) : abase[mi][0]) + toff[t] + kc * 32);
#pragma unroll
                    for (int ni = 0; ni < NPW; ++ni)
                        bfr[ni] = *reinterpret_cast<const bf16x8 *>(lds_b + bbase + ((t * KC + kc) * 2) * BN * 16 + ni * 512);
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NPW; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        };
        for (int c0 = 0; c0 < nchunk_total; c0 += AD) {
            step(std::integral_constant<int, 0>{}, c0);
            if (AD > 1 && c0 + 1 < nchunk_total) step(std::integral_constant<int, (AD > 1 ? 1 : 0)>{}, c0 + 1);
            if (AD > 2 && c0 + 2 < nchunk_total) step(std::integral_constant<int, (AD > 2 ? 2 : 0)>{}, c0 + 2);
            if (AD > 3 && c0 + 3 < nchunk_total) step(std::integral_constant<int, (AD > 3 ? 3 : 0)>{}, c0 + 3);
        }
        static_assert(AD <= 4, "ring depth");
        __syncthreads();

        // ---------------- epilogue ----------------
        if (A.debug & 8) { if (acc[0][0][0] == 123.456f) g_dbg_dummy = 1; return; }
        const int cout0 = cout_tile * BN;
        unsigned char *s_out = smem;
        const bool full = (y0 + TH <= A.H) && (x0 + TW <= A.W);
        float ssum[NPW], ssq[NPW];
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) { ssum[ni] = 0.f; ssq[ni] = 0.f; }
        // BatchNorm statistics of the raw accumulators (before bias / scale).  Full tiles take the branch-free loop.
        if (A.stats) {
            if (full) {
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { const float v = acc[mi][ni][r]; ssum[ni] += v; ssq[ni] = fmaf(v, v, ssq[ni]); }
            } else {
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = (wm * MPW + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            const float v = ((y0 + m / TW) < A.H && (x0 + m % TW) < A.W) ? acc[mi][ni][r] : 0.f;
                            ssum[ni] += v;
                            ssq[ni] = fmaf(v, v, ssq[ni]);
                        }
            }
        }
        // accumulators -> 16-bit out tile in LDS.  A lane owns one output channel (column) of 16 pixel rows; two
        // neighbouring lanes swap one value per register pair through DPP (quad_perm [1,0,3,2]) so that each lane
        // ends up with the (even column, odd column) pair of ONE row and stores a packed dword: no LDS permutes, half
        // the stores.  bias, scale and shift fold into one fma; the ReLU is taken on the packed 16-bit patterns.
        auto write_tile = [&](auto f16c) {
            constexpr bool F16 = decltype(f16c)::value;
            const bool odd = (l31 & 1) != 0;
            const xf_s16x2 lo = A.orelu ? xf_s16x2{0, 0} : xf_s16x2{(short)-32768, (short)-32768};
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                const int col = (wn * NPW + ni) * 32 + l31;
                const int co = cout0 + col;
                const bool cok = co < A.Cout;
                const float osc = (A.oscale && cok) ? A.oscale[co] : 1.f;
                const float osh = fmaf((A.bias && cok) ? A.bias[co] : 0.f, osc, (A.oshift && cok) ? A.oshift[co] : 0.f);
                unsigned char *dst = s_out + (col & ~1) * 2 + (odd ? OSTR : 0);
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) {
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        const int r = 2 * rp;
                        const int m = (wm * MPW + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // row of register r
                        const float v0 = fmaf(acc[mi][ni][r], osc, osh), v1 = fmaf(acc[mi][ni][r + 1], osc, osh);
                        const float send = odd ? v0 : v1;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                        const xf_f32x2 pr = {odd ? got : v0, odd ? v1 : got};      // (even column, odd column) of row m (+1 on odd lanes)
                        xf_s16x2 pk;
                        if (F16) pk = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(pr, xf_h16x2));
                        else pk = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(pr, xf_bf16x2));
                        pk = __builtin_elementwise_max(pk, lo);
                        *reinterpret_cast<unsigned *>(dst + m * OSTR) = __builtin_bit_cast(unsigned, pk);
                    }
                }
            }
        };
        if (A.out_f16 || A.eres) write_tile(std::true_type{});        // the fused residual epilogue stages r in fp16
        else write_tile(std::false_type{});
        if (A.stats) {
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                ssum[ni] += __shfl_xor(ssum[ni], 32);
                ssq[ni] += __shfl_xor(ssq[ni], 32);
                if (half == 0) {
                    s_stats[wave][0][(wn * NPW + ni) * 32 + l31] = ssum[ni];
                    s_stats[wave][1][(wn * NPW + ni) * 32 + l31] = ssq[ni];
                }
            }
        }
        __syncthreads();
        if (A.stats && tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            const int wn_of = col / (NPW * 32);
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < WM; ++k) v += s_stats[k * WN + wn_of][which][col];
            const int co = cout0 + col;
            if (co < A.Cout) A.stats[((size_t)tile * 2 + which) * A.Cout + co] = v;
        }
        // coalesced store of the tile: 16-byte vectors, BN/8 per pixel
        {
            constexpr int VO = BN / 8;
            const int Ho = A.H * A.ostride, Wo = A.W * A.ostride;
            const int pa = par >> 1, pb = par & 1;
            ChanXf et;                                       // fused residual epilogue: the other branch's per-channel affine
            et.on = false;
            if (A.eres && A.eres_scale) {
                const int c8 = cout0 + (tid % VO) * 8;          // 256 % VO == 0: a thread always stores the same 8 channels
                et.on = true;
#pragma unroll
                for (int j = 0; j < 8; ++j) { et.sc[j] = c8 + j < A.Cout ? A.eres_scale[c8 + j] : 1.f; et.sh[j] = c8 + j < A.Cout ? A.eres_shift[c8 + j] : 0.f; }
            }
            // one 16-byte vector of the tile: optional fused residual epilogue, then the store
            auto store_vec = [&](int v, const uint4 *pre) {
                const int m = v / VO, q = v % VO;
                const int y = y0 + m / TW, x = x0 + m % TW;
                const int co = cout0 + q * 8;
                if (y < A.H && x < A.W && co < A.Cout) {
                    const int oy = y * A.ostride + pa, ox = x * A.ostride + pb;
                    uint4 val = *reinterpret_cast<const uint4 *>(s_out + m * OSTR + q * 16);
                    const size_t opix = ((size_t)n * Ho + oy) * Wo + ox;
                    if (A.eres) {
                        // out = bf16([relu]((eres * scale + shift) + r)), r = this convolution's fp16-rounded result: the same
                        // arithmetic as the staging transform of a (raw, residual) source pair (xform8)
                        V16 e, r;
                        e.u = pre ? *pre : *reinterpret_cast<const uint4 *>(A.eres + opix * A.Cout + co);
                        r.u = val;
                        if (A.eres_f16) val = xform8(e, &r, et, A.eres_relu != 0, true).u;
                        else {                                   // bf16 other branch (identity shortcut of an HRNet block)
                            V16 o;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                float t = bf2f(e.h[j]);
                                if (et.on) t = fmaf(t, et.sc[j], et.sh[j]);
                                t += ld16(r.h[j], true);
                                o.h[j] = f2bf(A.eres_relu ? fmaxf(t, 0.f) : t);
                            }
                            val = o.u;
                        }
                    }
                    unsigned short *dst = A.out + opix * A.out_cstride + A.out_coff + co;
                    *reinterpret_cast<uint4 *>(dst) = val;               // Cout % 8 == 0 (checked by the ABI entry)
                }
            };
            constexpr bool PRE = (BN == 64 && TH == 16 && CK == 16);      // the configuration the residual units run
            if constexpr (PRE) {
                if (A.eres) {
                    // the other branch's vectors are requested up front (clamped address): their latency overlaps instead of
                    // sitting in front of every store
                    constexpr int NV = TH * TW * VO / 256;
                    uint4 ev[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int v = tid + i * 256, m = v / VO, q = v % VO;
                        int y = y0 + m / TW, x = x0 + m % TW, co = cout0 + q * 8;
                        y = y < A.H ? y : A.H - 1; x = x < A.W ? x : A.W - 1; co = co < A.Cout ? co : 0;
                        ev[i] = *reinterpret_cast<const uint4 *>(A.eres + (((size_t)n * A.H + y) * A.W + x) * A.Cout + co);
                    }
#pragma unroll
                    for (int i = 0; i < NV; ++i) store_vec(tid + i * 256, &ev[i]);
                } else {
                    for (int v = tid; v < TH * TW * VO; v += 256) store_vec(v, nullptr);
                }
            } else {
                for (int v = tid; v < TH * TW * VO; v += 256) store_vec(v, nullptr);
            }
        }
    }
}

template <int TH, int TW, int CK, int BN, int WM, int WN, int TAPS>
int launch_conv(const ConvArgs &A, hipStream_t st) {
    using LDS = ConvLds<TH, TW, CK, BN, TAPS>;
    int ctot = 0;
    for (int i = 0; i < A.nsrc; ++i) ctot += A.src[i].C;
    const int smem = LDS::bytes(ctot);
    auto kern = conv_fwd_kernel<TH, TW, CK, BN, WM, WN, TAPS>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS::bytes(XF_MAX)) != hipSuccess)
            return check_launch("hipFuncSetAttribute(conv)");
        attr_done = true;
    }
    const int total_tiles = cdiv(A.W, TW) * cdiv(A.H, TH) * A.N * A.npar;
    const int ctiles = cdiv(A.Cout, BN);
    dim3 grid(total_tiles, ctiles, 1);
    kern<<<grid, 256, smem, st>>>(A);
    return check_launch("conv_fwd_kernel");
}

// ------------------------------------------------------------------------------------------------------
// Wave-specialised, weight-stationary, persistent variant for the 3x3 layers with 64 input channels (four 16-channel chunks;
// the whole weight tile of a cout block - 74 KB for 64 couts - stays in LDS): the full-resolution layers of the encoder stem
// and of the three residual units, forward and backward-data.
//
//   * one 8-wave workgroup per CU, persistent over a contiguous run of tiles (XCD-contiguous, like conv_fwd_kernel);
//   * the weights of the workgroup's cout tile are loaded ONCE and stay in LDS;
//   * waves 4..7 (movers) only move data.  In: four 16-channel halo chunks in flight in registers (>= 2 us of HBM latency
//     covered; straight-line code so that the compiler counts the loads in flight instead of draining them), transform
//     (BatchNorm scale/shift, residual, ReLU), LDS ring of two slots.  Out: the finished tile's out image -> global memory,
//     spread over the first three chunk steps of the NEXT tile, so that no matrix wave ever waits for the write path;
//   * waves 0..3 (consumers) only read fragments, issue MFMAs, and at the end of a tile convert their 64 pixels x BN couts and
//     park them in LDS as [cout][pixel] blocks (8-byte writes straight from the accumulator registers; the movers read them
//     back with the transposing ds_read_b64_tr_b16, which yields pixel-major 16-byte vectors for NHWC stores);
//   * one barrier per 16-channel chunk step.
// Accumulation order, MFMA shapes and epilogue arithmetic are those of conv_fwd_kernel<16,16,16,BN,4,1,TAPS>: the results are
// bit-identical (tests/test_gpu_conv.py compares the two).
// Timeline of one workgroup (debug build -DCDNET_WS_STAMPS, tools/ws_stamps.py) before the write path moved to the movers: chunk
// step 0.80 us (36 MFMAs per wave: the matrix pipe at the clock the chip holds under this load), barrier 0.22 us, conversion +
// LDS writes 1.2 us and global stores 2.5 us per tile, both in the consumers' serial path: 8.0 us per tile, 128 us per launch.
// ------------------------------------------------------------------------------------------------------
#ifdef CDNET_WS_STAMPS
// debug build only (CDNET_HIPCC_FLAGS=-DCDNET_WS_STAMPS): wall-clock stamps (100 MHz) of one consumer and one mover wave of one
// workgroup, parked in LDS and dumped at the end of the kernel; read back with cdnet_debug_ws_stamps
__device__ unsigned long long g_ws_stamps[2 * 1024];      // consumer stamps from 0, mover stamps from 1024 (384 each)
extern "C" __attribute__((visibility("default"))) int cdnet_debug_ws_stamps(unsigned long long *dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_stamps), sizeof(g_ws_stamps)) == hipSuccess ? 0 : 1;
}
#define WS_STAMP(id) do { if (stamp_on && sn < 380) { s_stamp[sn++] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(id); } } while (0)
#define WS_STAMP_CYC(id) do { if (stamp_on && sn < 380) { s_stamp[sn++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (unsigned long long)(id); } } while (0)
#else
#define WS_STAMP(id) do { } while (0)
#define WS_STAMP_CYC(id) do { } while (0)
#endif

template <int BN, int TAPS>
struct WsLds {
    static constexpr int TH = 16, TW = 16, CK = 16;
    // halo image: 32 B per pixel, unpadded; the two 16-byte k-halves of a pixel sit at (half ^ (halo row & 1)) * 16.  An A-fragment
    // ds_read_b128 serves 16 lanes per LDS cycle - 8 pixels of one tile row and 8 of the next, same k-half - and with the row-parity
    // swizzle these land on 16 distinct 16-byte bank groups (tools/micro/lds_read_patterns.hip: 4 LDS cycles per read; the padded
    // 48-byte layout of conv_fwd_kernel takes 7-8)
    static constexpr int PSTR = CK * 2;
    static constexpr int NPIX = (TH + 2) * (TW + 2);
    static constexpr int A_BYTES = NPIX * PSTR;                   // one ring slot: the halo tile of a 16-channel chunk
    static constexpr int NSLOT = 4;                               // ring slots = the four chunks of a tile
    static constexpr int B_CHUNK = TAPS * CK * BN * 2;            // packed weights of one chunk
    static constexpr int IROW = 72;                               // a cout row of an out-image block: 32 pixels x 2 B + 8 (conflict-free 8-byte writes)
    static constexpr int IBLK = 32 * IROW;                        // one block: 32 couts x 32 pixels
    static constexpr int OUT_WAVE = 2 * (BN / 32) * IBLK;         // a consumer wave's 64 pixels x BN couts
    static constexpr int STATS_BYTES = 2 * 4 * 2 * BN * 4;        // double-buffered [wave][sum|sumsq][BN]
    __host__ __device__ static int bytes(int nchunk, int ctot, int xf_rows = 2) {      // (xf_rows: scale | shift)
        return nchunk * B_CHUNK + NSLOT * A_BYTES + 4 * OUT_WAVE + STATS_BYTES + xf_rows * ((ctot + 7) / 8 * 8) * 4;
    }
};

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));      // register staging type (HIP's uint4 struct copies defeat SROA)
typedef short ws_s16x4 __attribute__((ext_vector_type(4)));

// XF: input transform of every source, decided by the launcher - 0 plain bf16, 1 fp16 raw x scale + shift -> ReLU (training-mode
// BatchNorm source, packed math), 2 anything (run-time flags).
// STATS: per-tile channel sums of the unrounded accumulators.
// STREAM: more than four chunks per tile (128+ input channels, two-source layers) - the weight tile no longer fits the LDS; its
// chunks then stream through a four-slot ring by LDS-DMA, two chunks ahead of the consumers, beside the halo ring.
// The launcher guarantees nchunk % 4 == 0 (== 4 without STREAM) and full tiles (H, W multiples of 16).
// (Round 2's mover-side BatchNorm-backward experiments - the second pass applied while staging, XF 3, and the channel sums beside the
// stores, cdnet_conv_args.ws = 2 - were correct and not faster: a vector instruction of a mover wave costs the matrix pipe issue time.
// They were removed in round 3; the fp32 kernel carries the sums in the consumers' gaps instead, conv32ws.hip.)
template <int BN, int TAPS, int XF, bool STATS, bool STREAM>
__global__ __launch_bounds__(512) void conv_ws_kernel(ConvArgs A) {
    using L = WsLds<BN, TAPS>;
    constexpr int TH = 16, TW = 16, CK = 16, PSTR = L::PSTR, HW_ = TW + 2, NPIX = L::NPIX;
    constexpr int NT = BN / 32, NPW = NT, MPW = 2;               // consumer wave wm: M tiles 2wm, 2wm+1 (64 pixels), all N tiles
    constexpr int VPP = CK / 8, NA = (NPIX * VPP + 255) / 256;
    constexpr int PF = 4;                                         // halo chunks in flight per mover thread
    constexpr int NWS = 4;                                        // weight slots in LDS (the whole tile without STREAM)
    const int NCH = STREAM ? A.nchunk : 4;                        // chunks per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the halo ring first: its fragment addresses (slot, tap) then fit the 16-bit offset field of ds_read from one base register per M tile
    unsigned char *lds_a = smem;
    unsigned char *lds_w = smem + L::NSLOT * L::A_BYTES;
    unsigned char *lds_o = lds_w + NWS * L::B_CHUNK;
    float *s_stats = reinterpret_cast<float *>(lds_o + 4 * L::OUT_WAVE);          // [2][4][2][BN]
    float *s_xf = reinterpret_cast<float *>(lds_o + 4 * L::OUT_WAVE + L::STATS_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0n = A.src[0].C, ctot = c0n + (A.nsrc > 1 ? A.src[1].C : 0);
    const int xfs = (ctot + 7) / 8 * 8;
    const int cout_tile = blockIdx.y;
    const int cout0 = cout_tile * BN;
#ifdef CDNET_WS_STAMPS
    unsigned long long *s_stamp = reinterpret_cast<unsigned long long *>(smem + L::bytes(NWS, ctot, 2)) + (wave >= 4 ? 384 : 0);
    const bool stamp_on = blockIdx.x == 17 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4);
    int sn = 0;
#endif

    // this workgroup's contiguous run of tiles; XCD k (workgroups k, k+8, ...) serves the k-th eighth of the tiles
    const int tiles_x = A.W / TW, tiles_y = A.H / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int T = A.N * tiles_img;
    int t_lo, t_hi;
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        if ((G & 7) == 0) {
            const int xcd = b & 7, idx = b >> 3, nw = G >> 3;
            const long long x0 = (long long)T * xcd / 8, x1 = (long long)T * (xcd + 1) / 8;
            t_lo = (int)(x0 + (x1 - x0) * idx / nw);
            t_hi = (int)(x0 + (x1 - x0) * (idx + 1) / nw);
        } else {
            t_lo = (int)((long long)T * b / G);
            t_hi = (int)((long long)T * (b + 1) / G);
        }
    }
    const int ntl = t_hi - t_lo;
    const int S = ntl * NCH;                                      // chunk steps of this workgroup

    if (S == 0) return;
    // start-up: the movers put their first four halo chunks in flight and fill the scale/shift table while the consumers bring in the
    // resident weights; one barrier, then the movers stage chunks 0 and 1

    auto chunk_src = [&](int k, int &si, int &cc0) {
        const int n0 = A.src[0].C / CK;
        if (k < n0) { si = 0; cc0 = k * CK; } else { si = 1; cc0 = (k - n0) * CK; }
    };

    if (wave >= 4) {
        // ================================ movers ================================
        // Straight-line code only: every load is issued unconditionally (clamped address, clamped chunk index past the end of
        // the run) and out-of-range vectors are zeroed by a mask, so that the compiler can count the loads in flight
        // (s_waitcnt vmcnt(N)) instead of draining them at a join.
        const int ptid = tid - 256;
        // Round 5 (conv_ws16_kernel's finding, conv16ws.hip): a request takes a chunk PAIR - 32 channels = 64 contiguous bytes of a pixel, a whole
        // sector of the memory side - instead of one chunk's 32 bytes (every sector was requested twice).  Lane -> pixel v / 4, 16-byte segment
        // v % 4 = chunk (v % 4) / 2 of the pair, k-half v % 2; register set R of a pair holds half of the pair's vectors (group R & 1), both sets
        // are written into the pair's two ring slots in the same interval.  The launcher guarantees pairs inside one source (n0 even).
        constexpr int VPQ = 2 * VPP, NG = 2;
        const int slot = ptid % VPQ;
        const int khalf = slot & 1, c2 = slot >> 1;
        u32x4v pa[PF][NA];
        unsigned eo[PF][NA];                     // byte offsets of the requests (read again for a residual operand); bit 31 = zero fill
        // per-thread constants: halo coordinates and LDS offsets of its vectors
        int hyx[NG][NA], doff[NG][NA];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int v = ptid + (g * NA + i) * 256;
                const int pix = v / VPQ;
                const int hy = pix / HW_, hx = pix - hy * HW_;
                hyx[g][i] = v < NPIX * VPQ ? ((hy << 8) | hx) : 0x1f1f;      // (31 = a vector that never exists: bit 31 of the masks is always set)
                doff[g][i] = pix * PSTR + ((khalf ^ (hy & 1)) * 16);
            }
        // Requests through buffer descriptors (conv_ws32_kernel's recipe, round 3): a vector outside the image / the source window carries
        // bit 31 in its byte offset and reads as zeros (the launcher admits tensors below 2 GB) - no select, no 64-bit address arithmetic and,
        // for plain sources, no mask per vector: every vector instruction of a mover wave costs the consumers' MFMA stream issue time.
        const unsigned src_bytes0 = (unsigned)(A.N * A.npar) * A.src[0].Hs * (A.src[0].row_stride ? A.src[0].row_stride : A.src[0].Ws * A.src[0].C) * 2u;
        const unsigned src_bytes1 = A.nsrc > 1 ? (unsigned)(A.N * A.npar) * A.src[1].Hs * (A.src[1].row_stride ? A.src[1].row_stride : A.src[1].Ws * A.src[1].C) * 2u : 0u;
        const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 ? A.src[1].x : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].res ? A.src[0].res : A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 && A.src[1].res ? A.src[1].res : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff) -> u32x4v {
            return __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
        };
        // bits [lo, hi) clear, everything else set (lo, hi clamped to [0, 31])
        auto bad_mask = [](int lo, int hi) -> unsigned {
            lo = lo < 0 ? 0 : (lo > 31 ? 31 : lo);
            hi = hi < lo ? lo : (hi > 31 ? 31 : hi);
            return ~(((1u << hi) - 1u) & ~((1u << lo) - 1u));
        };
        // cursors (no integer division per chunk): the issue cursor walks chunks 0, 1, 2, ... of the run and stops on the last
        // one; the commit cursor only needs the chunk-in-tile index
        int ik = 0, ic = 0, in_, iy0, ix0;
        {
            in_ = t_lo / tiles_img;
            const int r = t_lo - in_ * tiles_img, ty = r / tiles_x;
            iy0 = ty * TH; ix0 = (r - ty * tiles_x) * TW;
        }
        int o_n = in_, o_y0 = iy0, o_x0 = ix0;   // out cursor: the tile the consumers are accumulating
        int s_n = 0, s_y0 = 0, s_x0 = 0;         // the finished tile whose out image the consumers park during interval A
        bool s_ok = false;
        int ck = 0;
        // staging geometry of the current tile and source (the chunks of one source share it): element offset of each vector
        // without the chunk's channel offset, validity mask
        unsigned ge[NG][NA];                     // byte offset of channel 0 of the tile's vectors in the current source, bit 31 = zero fill
        const int n0 = A.src[0].C / CK;
        auto issue = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            constexpr int G = R & 1;                              // which half of the pair's vectors this register set holds
            int si, cc0;
            chunk_src(ik & ~1, si, cc0);                          // (the pair's first chunk: both sets request from its base)
            const ConvSrc &s = A.src[si];
            if ((ik & ~1) == 0 || (ik & ~1) == n0) {
                const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
                // halo row r <-> y = iy0 - 1 + r, halo column c <-> x = ix0 - 1 + c; valid = inside the image and the source window
                const int ylo = s.off_y > 0 ? s.off_y : 0, yhi = A.H < s.off_y + s.Hs ? A.H : s.off_y + s.Hs;
                const int xlo = s.off_x > 0 ? s.off_x : 0, xhi = A.W < s.off_x + s.Ws ? A.W : s.off_x + s.Ws;
                const unsigned rowbad = bad_mask(ylo - (iy0 - 1), yhi - (iy0 - 1)), colbad = bad_mask(xlo - (ix0 - 1), xhi - (ix0 - 1));
                const unsigned img_b = (unsigned)((in_ * s.Hs + (iy0 - 1 - s.off_y)) * rs + (ix0 - 1 - s.off_x) * s.C) * 2u;
                const unsigned rs_b = (unsigned)rs * 2u, c_b = (unsigned)s.C * 2u;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const unsigned hy = (unsigned)hyx[G][i] >> 8, hx = (unsigned)hyx[G][i] & 0xffu;
                    const unsigned t = (rowbad >> hy) | (colbad >> hx);
                    ge[G][i] = ((img_b + hy * rs_b + hx * c_b + (unsigned)slot * 16u) & 0x7fffffffu) | (t << 31);
                }
            }
            const __amdgpu_buffer_rsrc_t rsx = si ? rsx1 : rsx0;
            const unsigned cc0_b = (unsigned)cc0 * 2u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned voff = ge[G][i] + cc0_b;                // (a zero-fill vector keeps bit 31: beyond every tensor the launcher admits)
                eo[R][i] = voff;
                pa[R][i] = bload(rsx, voff);
            }
            // advance (saturating at the last chunk of the run)
            if (ic + 1 < S) {
                ++ic;
                if (++ik == NCH) {
                    ik = 0;
                    ix0 += TW;
                    if (ix0 >= A.W) { ix0 = 0; iy0 += TH; if (iy0 >= A.H) { iy0 = 0; ++in_; } }
                }
            }
        };
        // chunk c_ of the run (register set R = c_ % 4) -> ring slot c_ % 4
        auto commit = [&](auto rc, int c_) {
            constexpr int R = decltype(rc)::value;
            constexpr int G = R & 1;
            int si, cc0;
            chunk_src(ck & ~1, si, cc0);
            if (c_ + 1 < S) { if (++ck == NCH) ck = 0; }
            const ConvSrc &s = A.src[si];
            const float *xf = s_xf + (si ? c0n : 0) + cc0 + slot * 8;
            float sc[8], sh[8];
            if (XF != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = xf[j]; sh[j] = xf[xfs + j]; }
            }
            unsigned char *dst0 = lds_a + ((R & 2) + c2) * L::A_BYTES;      // (this thread's vectors belong to chunk c2 of the pair, whichever set R is)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                u32x4v val;
                if (XF == 0) val = pa[R][i];
                else if (XF == 1) val = xf_bnrelu_f16<false>(pa[R][i], pa[R][i], sc, sh);
                else {
                    ChanXf t;
                    t.on = s.scale != nullptr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { t.sc[j] = sc[j]; t.sh[j] = sh[j]; }
                    V16 raw, rr;
                    raw.u = __builtin_bit_cast(uint4, pa[R][i]);
                    const bool relu = s.relu != 0, f16 = s.f16 != 0;
                    if (s.res) {
                        rr.u = __builtin_bit_cast(uint4, bload(si ? rsr1 : rsr0, eo[R][i]));
                        val = __builtin_bit_cast(u32x4v, xform8(raw, &rr, t, relu, f16).u);
                    } else if (!t.on && !relu && !f16) val = pa[R][i];
                    else val = __builtin_bit_cast(u32x4v, xform8(raw, nullptr, t, relu, f16).u);
                }
                if (XF != 0) {                                   // (plain sources: the zero fill arrived as zeros)
                    const unsigned keep = (int)eo[R][i] < 0 ? 0u : 0xffffffffu;
                    val &= keep;
                }
#ifdef CDNET_WS_STAMPS
                if (A.debug & 1024) { if (val[0] == 0x12345678u) *reinterpret_cast<u32x4v *>(dst0 + doff[G][i]) = val; continue; }
#endif
                if (ptid + (G * NA + i) * 256 < NPIX * VPQ)
                    *reinterpret_cast<u32x4v *>(dst0 + doff[G][i]) = val;
            }
        };
        // ---- out path: mover wave w stores the region of consumer wave w.  Piece pc = (M tile mi, pixel half ch, cout pass kk): the four
        // ---- 16-lane groups take four cout octets, lane i of a group receives pixel i of tile row 4w + 2mi + ch (transposing read:
        // ---- lane 4q+p supplies the address of cout row q, pixels 4p..4p+3) -> one 16-byte NHWC store per lane
        constexpr int KO = BN / 32, NP = 4 * KO;
        typedef ws_s16x4 __attribute__((address_space(3))) * lptr;
        const int pw = wave - 4;
        const int lg = lane >> 4, li = lane & 15;
        const unsigned char *s_img = lds_o + pw * L::OUT_WAVE + (li >> 2) * L::IROW + (li & 3) * 8;
        auto store_pieces = [&](bool now) {
            // all transposing reads first, into registers of their own, then the stores: with the pair of reads, the wait for them and
            // the store piece by piece through one register quad (what the compiler makes of a single loop) every piece paid the LDS
            // latency - 1.2 us per tile in the movers' second interval, where the consumers then waited 0.9 us (stamp build)
            ws_s16x4 t1[NP], t2[NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const int mi = pc / (2 * KO), ch = (pc / KO) % 2, kk = pc % KO;
                const int o = lg + 4 * kk;
                const int ni = o >> 2, r0 = 8 * (o & 3);
                const unsigned char *p = s_img + ((mi * NPW + ni) * 32 + r0) * L::IROW + ch * 32;
                t1[pc] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
                t2[pc] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 4 * L::IROW));
            }
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const int mi = pc / (2 * KO), ch = (pc / KO) % 2, kk = pc % KO;
                const int o = lg + 4 * kk;
                const int y = s_y0 + pw * 4 + mi * 2 + ch, x = s_x0 + li, co = cout0 + 8 * o;
                if (now && s_ok && co < A.Cout) {
                    const uint2 a = __builtin_bit_cast(uint2, t1[pc]), b2 = __builtin_bit_cast(uint2, t2[pc]);
                    *reinterpret_cast<uint4 *>(A.out + (((size_t)s_n * A.H + y) * A.W + x) * A.out_cstride + A.out_coff + co) = make_uint4(a.x, a.y, b2.x, b2.y);
                }
            }
        };
        // STREAM: weight chunk wk of the tile -> weight slot wk & 3 by LDS-DMA (the packed chunk is the LDS image; one 1 KB
        // wave-instruction per 64 vectors, the four mover waves take them in turn); the cursor wraps at the end of a tile
        int wk = 0;
        auto dma_w = [&]() {
            constexpr int NVEC = L::B_CHUNK / 16;
            static_assert(NVEC % 64 == 0, "whole wave-instructions");
            const unsigned short *wsrc = A.w + ((size_t)cout_tile * NCH + wk) * (L::B_CHUNK / 2);
            unsigned char *wdst = lds_w + (wk & 3) * L::B_CHUNK;
#pragma unroll
            for (int i = 0; i < (NVEC / 64 + 3) / 4; ++i) {
                const int v0 = (i * 4 + pw) * 64;
                if (v0 < NVEC)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + (size_t)(v0 + lane) * 8),
                                                     (__attribute__((address_space(3))) void *)(wdst + v0 * 16), 16, 0, 0);
            }
            if (++wk == NCH) wk = 0;
        };
        // vmcnt(N) alone (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14): wait until only the N youngest
        // vector-memory operations of this wave are outstanding - the halo requests issued after the DMA stay in flight
        auto wait_vm = [](auto n_c) {
            constexpr int N = decltype(n_c)::value;
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
        };
        using NHL = std::integral_constant<int, 2 * NA>;          // halo requests of one interval (two chunks)
        // the movers' barrier of the streaming loop by hand: __syncthreads() carries a workgroup release fence, and with an LDS-DMA in the
        // wave's history the compiler turns that fence into s_waitcnt vmcnt(0) - draining the halo requests in flight every interval.
        // What the consumers need is: this wave's LDS writes done (lgkmcnt(0)), its weight DMA landed (everything older than the N
        // youngest vector-memory operations), then the barrier.
        auto stream_sync = [](auto n_c) {
            constexpr int N = decltype(n_c)::value;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        issue(I0{});
        issue(I1{});
        issue(I2{});
        issue(I3{});
        for (int c = ptid; c < ctot; c += 256) {
            const ConvSrc &Sx = c < c0n ? A.src[0] : A.src[1];
            const int cc = c < c0n ? c : c - c0n;
            s_xf[c] = Sx.scale ? Sx.scale[cc] : 1.f;
            s_xf[xfs + c] = Sx.shift ? Sx.shift[cc] : 0.f;
        }
        __syncthreads();                                         // table + weights
        commit(I0{}, 0);
        commit(I1{}, 1);
        if (STREAM) { dma_w(); dma_w(); }
        issue(I0{});
        issue(I1{});
        if (STREAM) wait_vm(NHL{});                              // weight chunks 0, 1 have landed (the two halo chunks requested after them stay in flight)
        __syncthreads();
        const int NIT = NCH / 4;                                 // loop iterations (four chunks = two barrier intervals) per tile
        int it = 0;
        // tile j (chunks q0 .. q0+3).  Interval A: the consumers work on chunks q0, q0+1 (slots 0, 1) and park the out image of tile
        // j-1; slots 2, 3 take chunks q0+2, q0+3.  Interval B: the consumers work on slots 2, 3; slots 0, 1 take the first two chunks
        // of tile j+1 and the out image of tile j-1 leaves for global memory.
        if (!STREAM) {
            for (int q0 = 0; q0 < S; q0 += 4) {
                WS_STAMP(1); commit(I2{}, q0 + 2); issue(I2{}); commit(I3{}, q0 + 3); issue(I3{}); WS_STAMP(3); __syncthreads();
                WS_STAMP(1); commit(I0{}, q0 + 4); issue(I0{}); commit(I1{}, q0 + 5); issue(I1{}); WS_STAMP(2); store_pieces(true); WS_STAMP(3); __syncthreads();
                s_n = o_n; s_y0 = o_y0; s_x0 = o_x0; s_ok = true;
                o_x0 += TW;
                if (o_x0 >= A.W) { o_x0 = 0; o_y0 += TH; if (o_y0 >= A.H) { o_y0 = 0; ++o_n; } }
            }
        } else {
            // per interval: the weight DMA of the two chunks after the ones being consumed first, (second interval of a tile: the
            // previous tile's out image), the halo commits and requests last - the explicit wait then leaves exactly the halo
            // requests of this interval in flight
            // (every LDS access of the interval - out-image reads, halo commits - comes BEFORE the DMA: the compiler cannot tell that
            //  the DMA's destination does not alias them and guards any later LDS access of this wave with s_waitcnt vmcnt(0),
            //  which would drain the halo requests in flight)
            for (int q0 = 0; q0 < S; q0 += 4) {
                commit(I2{}, q0 + 2); commit(I3{}, q0 + 3);
                dma_w(); dma_w();
                issue(I2{}); issue(I3{});
                stream_sync(NHL{});
                store_pieces(it == 0);
                commit(I0{}, q0 + 4); commit(I1{}, q0 + 5);
                dma_w(); dma_w();
                issue(I0{}); issue(I1{});
                stream_sync(NHL{});
                if (++it == NIT) {
                    it = 0;
                    s_n = o_n; s_y0 = o_y0; s_x0 = o_x0; s_ok = true;
                    o_x0 += TW;
                    if (o_x0 >= A.W) { o_x0 = 0; o_y0 += TH; if (o_y0 >= A.H) { o_y0 = 0; ++o_n; } }
                }
            }
        }
        __syncthreads();                                         // the consumers have parked the last tile's out image
        store_pieces(true);
#ifdef CDNET_WS_STAMPS
        if (stamp_on) { for (int i = 0; i < sn; ++i) g_ws_stamps[1024 + i] = s_stamp[i]; g_ws_stamps[1024 + sn] = 0; }
#endif
        return;
    }

    // ================================ consumers ================================
    if (!STREAM) {
        const u32x4v *src = reinterpret_cast<const u32x4v *>(A.w + (size_t)cout_tile * 4 * (L::B_CHUNK / 2));
        u32x4v *dst = reinterpret_cast<u32x4v *>(lds_w);
        constexpr int nv = 4 * (L::B_CHUNK / 16), NW = (nv + 255) / 256;
        // all of a thread's vectors in flight at once (one memory round trip for the 74 KB; the accumulators are not live yet)
        u32x4v wv[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = src[tid + i * 256 < nv ? tid + i * 256 : 0];
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (tid + i * 256 < nv) dst[tid + i * 256] = wv[i];
    }
    __syncthreads();                                             // table + weights
    const int wm = wave;
    const int half = lane >> 5, l31 = lane & 31;
    int abase[MPW][2];                                           // [.][parity of the tap's row offset]: the k-half swizzle follows the halo row
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int m = (wm * MPW + mi) * 32 + l31;
#pragma unroll
        for (int par = 0; par < 2; ++par) abase[mi][par] = ((m / TW) * HW_ + m % TW) * PSTR + ((half ^ ((m / TW + par) & 1)) * 16);
    }
    auto tpar = [](int t) { return (TAPS == 9 ? t / 3 : 1) & 1; };
    const int bbase = half * BN * 16 + l31 * 16;
    auto toff = [](int t) { return ((TAPS == 9 ? t / 3 : 1) * HW_ + (TAPS == 9 ? t % 3 : 1)) * PSTR; };      // folds to immediates
    // two accumulator sets: while one takes the current tile's MFMAs, the other (the finished tile) is converted and parked in the
    // shadow of those MFMAs
    f32x16 accA[MPW][NPW], accB[MPW][NPW];
    // this wave's out image: blocks [mi][ni] of 32 cout rows x 32 pixels; this lane owns cout row l31 of every block
    unsigned char *s_out = lds_o + wave * L::OUT_WAVE + l31 * L::IROW + half * 8;
    // epilogue constants of this lane (the same for every tile of the run)
    float e_osc[NPW], e_osh[NPW];
#pragma unroll
    for (int ni = 0; ni < NPW; ++ni) {
        const int co = cout0 + ni * 32 + l31;
        const bool cok = co < A.Cout;
        e_osc[ni] = (A.oscale && cok) ? A.oscale[co] : 1.f;
        e_osh[ni] = fmaf((A.bias && cok) ? A.bias[co] : 0.f, e_osc[ni], (A.oshift && cok) ? A.oshift[co] : 0.f);
    }
    const bool f16out = A.out_f16 != 0;
    const xf_s16x2 lo_clamp = A.orelu ? xf_s16x2{0, 0} : xf_s16x2{(short)-32768, (short)-32768};

    // ---- epilogue units of a finished accumulator set (full tiles only).  Statistics: registers 4g..4g+3 of block (mi, ni) into the
    // ---- lane's running sums, blocks in the order (ni, mi) - the summation order of conv_fwd_kernel.  Image: bias/scale/shift,
    // ---- 16-bit conversion, ReLU; registers 4g..4g+3 are four consecutive pixels of this lane's cout: one 8-byte LDS write
    constexpr int NU = MPW * NPW * 4;
    float st_sum = 0.f, st_sq = 0.f;
    auto stat_unit = [&](const f32x16 (&P)[MPW][NPW], int u, int par) {
        const int ni = u / (MPW * 4), mi = (u / 4) % MPW, g = u % 4;
        if (mi == 0 && g == 0) { st_sum = 0.f; st_sq = 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float v = P[mi][ni][4 * g + j]; st_sum += v; st_sq = fmaf(v, v, st_sq); }
        if (mi == MPW - 1 && g == 3) {
            float *sp = s_stats + par * (4 * 2 * BN);
            const float a = st_sum + __shfl_xor(st_sum, 32), b2 = st_sq + __shfl_xor(st_sq, 32);
            if (half == 0) {
                sp[(wave * 2 + 0) * BN + ni * 32 + l31] = a;
                sp[(wave * 2 + 1) * BN + ni * 32 + l31] = b2;
            }
        }
    };
    auto img_unit = [&](const f32x16 (&P)[MPW][NPW], int u) {
        const int mi = u / (NPW * 4), ni = (u / 4) % NPW, g = u % 4;
        const float osc = e_osc[ni], osh = e_osh[ni];
        const xf_f32x2 p0 = {fmaf(P[mi][ni][4 * g], osc, osh), fmaf(P[mi][ni][4 * g + 1], osc, osh)};
        const xf_f32x2 p1 = {fmaf(P[mi][ni][4 * g + 2], osc, osh), fmaf(P[mi][ni][4 * g + 3], osc, osh)};
        const xf_s16x2 h0 = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(p0, xf_h16x2)), h1 = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(p1, xf_h16x2));
        const xf_s16x2 b0 = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(p0, xf_bf16x2)), b1 = __builtin_bit_cast(xf_s16x2, __builtin_convertvector(p1, xf_bf16x2));
        const xf_s16x2 k0 = __builtin_elementwise_max(f16out ? h0 : b0, lo_clamp), k1 = __builtin_elementwise_max(f16out ? h1 : b1, lo_clamp);
        *reinterpret_cast<uint2 *>(s_out + (mi * NPW + ni) * L::IBLK + g * 16) = make_uint2(__builtin_bit_cast(unsigned, k0), __builtin_bit_cast(unsigned, k1));
    };
    // slot sl (0 .. 2 * 36 - 1) of interval A -> unit: statistics on the even slots from 0, then the image on the odd slots
    constexpr int SLOTS = TAPS * MPW * NPW;                      // MFMAs of one chunk step
    constexpr int IMG0 = STATS ? 2 * NU + 1 : 0;
    static_assert(IMG0 + 2 * NU <= 2 * SLOTS, "the deferred epilogue fits the MFMA slots of interval A");

    int stats_tile = -1, stats_par = 0;                         // statistics parked in LDS by a finished epilogue
    auto flush_stats = [&]() {
        if (STATS && stats_tile >= 0 && tid < 2 * BN) {
            // per-tile channel sums: the four waves' partials in a fixed order (deterministic)
            const int which = tid / BN, col = tid % BN;
            const float *sp = s_stats + stats_par * (4 * 2 * BN);
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += sp[(w4 * 2 + which) * BN + col];
            const int co = cout0 + col;
            if (co < A.Cout) A.stats[((size_t)stats_tile * 2 + which) * A.Cout + co] = v;
        }
        stats_tile = -1;
    };

    // one barrier interval on accumulator set C: two chunks = 2 * TAPS taps of MPW x NPW MFMAs; the fragments of a tap are requested
    // two taps ahead (three fragment sets): with four consumer waves reading, an LDS read issued only one tap (4 MFMAs = 128 cycles)
    // ahead is not back in time and every tap stalls.  FIRST: the tile's first chunk starts from zero.  EPI: the interval carries the
    // epilogue units of the finished set P after its MFMAs (program order = issue order: vector and LDS instructions ride in the MFMA shadow)
    auto interval = [&](auto first_c, auto epi_c, f32x16 (&C)[MPW][NPW], const f32x16 (&P)[MPW][NPW], int par, int c0) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr bool EPI = decltype(epi_c)::value;
        constexpr int NTAP = 2 * TAPS;
        bf16x8 af[3][MPW], bfr[3][NPW];
        const unsigned char *la = lds_a + c0 * L::A_BYTES, *lw = lds_w + c0 * L::B_CHUNK;
        auto request = [&](int tau) {
            const int ch = tau / TAPS, t = tau % TAPS;
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) af[tau % 3][mi] = *reinterpret_cast<const bf16x8 *>(la + ch * L::A_BYTES + abase[mi][tpar(t)] + toff(t));
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) bfr[tau % 3][ni] = *reinterpret_cast<const bf16x8 *>(lw + ch * L::B_CHUNK + bbase + (t * 2) * BN * 16 + ni * 512);
        };
        request(0);
        request(1);
#pragma unroll
        for (int tau = 0; tau < NTAP; ++tau) {
            if (tau + 2 < NTAP) request(tau + 2);
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni) {
                    if (FIRST && tau == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tau % 3][mi], bfr[tau % 3][ni], z, 0, 0, 0);
                    } else {
                        C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tau % 3][mi], bfr[tau % 3][ni], C[mi][ni], 0, 0, 0);
                    }
                    // the gap behind this MFMA: in an epilogue interval, one unit of the finished tile
                    if (EPI) {
                        const int sl = (tau * MPW + mi) * NPW + ni;
                        if (STATS && sl < 2 * NU && (sl & 1) == 0) { __builtin_amdgcn_sched_barrier(0); stat_unit(P, sl / 2, par); __builtin_amdgcn_sched_barrier(0); }
                        if (sl >= IMG0 && sl < IMG0 + 2 * NU && ((sl - IMG0) & 1) == 0) { __builtin_amdgcn_sched_barrier(0); img_unit(P, (sl - IMG0) / 2); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
        }
    };
    using F_ = std::false_type;
    using T_ = std::true_type;
    // tile j on set C; P = the finished tile j-1 (its epilogue rides in interval A)
    auto tile_step = [&](auto has_prev, f32x16 (&C)[MPW][NPW], const f32x16 (&P)[MPW][NPW], int j) {
        constexpr bool HP = decltype(has_prev)::value;
        const int par = (j + 1) & 1;
        WS_STAMP(10);
#ifdef CDNET_WS_STAMPS
        if (A.debug & 512) interval(T_{}, F_{}, C, P, par, 0); else
#endif
        interval(T_{}, has_prev, C, P, par, 0);
        if (HP && STATS) { stats_tile = t_lo + j - 1; stats_par = par; }
        WS_STAMP(11);
        __syncthreads();
        WS_STAMP(14);
        flush_stats();
        interval(F_{}, F_{}, C, P, par, 2);
        WS_STAMP(12);
        __syncthreads();
        WS_STAMP(13);
        if (STREAM) {
            // the rest of a long tile: intervals alternate between the two halves of the rings
            const int NI = NCH / 2;
            for (int i = 2; i < NI; ++i) {
                interval(F_{}, F_{}, C, P, par, (i & 1) * 2);
                __syncthreads();
            }
        }
    };
    // the last tile of the run: nothing left to hide behind
    auto serial_epilogue = [&](const f32x16 (&P)[MPW][NPW], int j) {
        const int par = j & 1;
        if (STATS) {
#pragma unroll
            for (int u = 0; u < NU; ++u) stat_unit(P, u, par);
            stats_tile = t_lo + j;
            stats_par = par;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) img_unit(P, u);
    };

    __syncthreads();                                             // chunks 0, 1 are staged
    tile_step(F_{}, accA, accB, 0);
    int j = 1;
    for (; j + 1 < ntl; j += 2) {
        tile_step(T_{}, accB, accA, j);
        tile_step(T_{}, accA, accB, j + 1);
    }
    if (j < ntl) {
        tile_step(T_{}, accB, accA, j);
        serial_epilogue(accB, j);
    } else {
        serial_epilogue(accA, ntl - 1);
    }
    __syncthreads();
    flush_stats();
#ifdef CDNET_WS_STAMPS
    if (stamp_on) { for (int i = 0; i < sn; ++i) g_ws_stamps[i] = s_stamp[i]; g_ws_stamps[sn] = 0; }
#endif
}

// eligibility + launch of the wave-specialised kernel; returns -1 when the layer must take conv_fwd_kernel
template <int BN, int TAPS>
int try_launch_conv_ws(const ConvArgs &A, hipStream_t st, bool dry_run = false) {
    using L = WsLds<BN, TAPS>;
    int ctot = 0;
    if (A.ws == 2) return -1;                                    // (BatchNorm-backward sums beside the stores: the fp32 kernel only, conv32ws.hip)
    if (A.eres) return -1;                                       // fused residual epilogues stay on conv_fwd_kernel
    if (A.nchunk < 4 || A.nchunk % 4 != 0) return -1;            // whole pairs of barrier intervals (two chunks each) per tile
    if ((A.src[0].C / 16) & 1) return -1;                        // (the movers request by chunk pairs: every pair inside one source)
    const bool stream = A.nchunk != 4;                           // 128+ input channels / two sources: the weight chunks stream through the LDS
    if (A.H % 16 != 0 || A.W % 16 != 0) return -1;               // full tiles only
    for (int i = 0; i < A.nsrc; ++i) {
        if (A.src[i].pool || A.src[i].relu == 3) return -1;
        ctot += A.src[i].C;
        // the movers' requests: 31-bit byte offsets from the source's base (bit 31 marks a zero-fill vector)
        const long long rs_ = A.src[i].row_stride ? A.src[i].row_stride : (long long)A.src[i].Ws * A.src[i].C;
        if ((long long)A.N * A.npar * A.src[i].Hs * rs_ * 2 >= (1LL << 31)) return -1;
    }
#ifdef CDNET_WS_STAMPS
    const int smem = L::bytes(4, ctot, 2) + 2 * 384 * 8;
#else
    const int smem = L::bytes(4, ctot, 2);
#endif
    if (smem > 160 * 1024) return -1;
    const int T = (A.W / 16) * (A.H / 16) * A.N;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return check_launch("hipGetDeviceProperties");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ctiles = cdiv(A.Cout, BN);
    // a persistent workgroup pays ~3 us of start-up (weights + first chunks): worth it from a few tiles' worth of chunks per workgroup on
    const int Gmax = n_cu / ctiles > 0 ? n_cu / ctiles : 1;
    // (few tiles but long channel loops - the 16x16-pixel bottleneck layers - still take it: the workgroups that exist keep the memory
    //  pipeline full, which the one-tile-per-workgroup kernel does not)
    // (12, not 16: the decoder's first block - 768 -> 256 channels on 16 tiles of 16 x 16 pixels, 64 workgroups either way - 120 us on the one-tile kernel)
    if (!(A.debug & 64) && ((long long)T * A.nchunk < 12LL * Gmax || (T < Gmax && A.nchunk < 16))) return -1;
    bool all_plain = true, all_fast = true;
    for (int i = 0; i < A.nsrc; ++i) {
        const ConvSrc &s = A.src[i];
        all_plain = all_plain && !s.scale && !s.relu && !s.res && !s.f16;
        all_fast = all_fast && s.scale && s.relu && !s.res && s.f16 == 1;
    }
    int G = n_cu / ctiles;
    G = G > T ? T : G;
    if (G >= 8) G &= ~7;
    if (G < 1) G = 1;
    dim3 grid(G, ctiles, 1);
    auto launch2 = [&](auto xf_c, auto st_c, auto sm_c) -> int {
        constexpr int XF = decltype(xf_c)::value;
        constexpr bool STATS = decltype(st_c)::value;
        constexpr bool STREAM = decltype(sm_c)::value;
        auto kern = conv_ws_kernel<BN, TAPS, XF, STATS, STREAM>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return check_launch("hipFuncSetAttribute(conv_ws)");
            attr_done = true;
        }
        kern<<<grid, 512, smem, st>>>(A);
        return CDNET_OK;
    };
    auto launch = [&](auto xf_c, auto st_c) -> int {
        return stream ? launch2(xf_c, st_c, std::true_type{}) : launch2(xf_c, st_c, std::false_type{});
    };
    const int xf = all_plain ? 0 : (all_fast ? 1 : 2);
    if (dry_run) return CDNET_OK;
    using X0 = std::integral_constant<int, 0>;
    using X1 = std::integral_constant<int, 1>;
    using X2 = std::integral_constant<int, 2>;
    int rc;
    if (A.stats) rc = xf == 0 ? launch(X0{}, std::true_type{}) : (xf == 1 ? launch(X1{}, std::true_type{}) : launch(X2{}, std::true_type{}));
    else rc = xf == 0 ? launch(X0{}, std::false_type{}) : (xf == 1 ? launch(X1{}, std::false_type{}) : launch(X2{}, std::false_type{}));
    if (rc != CDNET_OK) return rc;
    return check_launch("conv_ws_kernel");
}

template <int TAPS>
int dispatch_conv(const ConvArgs &A, hipStream_t st) {
    // configuration key: (tile, CK, BN); chosen by the host (cdnet_amd/engine.py) per layer
    const int key = A.tile * 10000 + A.CK * 100 + (A.BN == 128 ? 99 : A.BN);
    switch (key) {
        case 16 * 10000 + 16 * 100 + 32: return launch_conv<16, 16, 16, 32, 4, 1, TAPS>(A, st);
        case 16 * 10000 + 16 * 100 + 64: return launch_conv<16, 16, 16, 64, 4, 1, TAPS>(A, st);
        case 16 * 10000 + 32 * 100 + 32: return launch_conv<16, 16, 32, 32, 4, 1, TAPS>(A, st);
        case 16 * 10000 + 32 * 100 + 64: return launch_conv<16, 16, 32, 64, 4, 1, TAPS>(A, st);
        case 16 * 10000 + 32 * 100 + 99: return launch_conv<16, 16, 32, 128, 2, 2, TAPS>(A, st);
        case 16 * 10000 + 64 * 100 + 64: return launch_conv<16, 16, 64, 64, 4, 1, TAPS>(A, st);
        case 8 * 10000 + 32 * 100 + 64: return launch_conv<8, 8, 32, 64, 2, 2, TAPS>(A, st);
        case 8 * 10000 + 32 * 100 + 99: return launch_conv<8, 8, 32, 128, 2, 2, TAPS>(A, st);
        case 8 * 10000 + 64 * 100 + 64: return launch_conv<8, 8, 64, 64, 2, 2, TAPS>(A, st);
        default:
            set_error("cdnet_conv: unsupported configuration tile=%d CK=%d BN=%d", A.tile, A.CK, A.BN);
            return CDNET_E_ARG;
    }
}

// ------------------------------------------------------------------------------------------------------
// A source with its pending transform (BatchNorm scale/shift, residual, ReLU, 2x2 max-pool, pad offset) written out as a
// plain bf16 tensor - exactly the values stage_input would put into LDS.  Used where the on-the-fly transform costs more
// than a pass over HBM: max-pooled sources (4 loads + 4 transforms per staged element, repeated by every output-channel
// block and halo; the pooled staging path is not pipelined).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void materialize_kernel(const ConvSrc s, int N, int H, int W, unsigned short *__restrict__ out) {
    const int VPP = s.C / 8;
    const size_t total = (size_t)N * H * W * VPP;
    const bool relu = s.relu != 0, f16 = s.f16 != 0;
    const int Hl = s.pool ? (s.Hs + (s.pool == 2)) / 2 : s.Hs, Wl = s.pool ? (s.Ws + (s.pool == 2)) / 2 : s.Ws;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int slot = (int)(i % VPP);
        const size_t pix = i / VPP;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        ChanXf t;
        load_chan_xf(t, s, nullptr, 0, slot * 8);
        const bool plain = !t.on && !relu && s.res == nullptr && !f16;
        const size_t img = (size_t)n * s.Hs * rs;
        V16 val;
        val.u = make_uint4(0, 0, 0, 0);
        const int ys = y - s.off_y, xs = x - s.off_x;
        if (ys >= 0 && ys < Hl && xs >= 0 && xs < Wl) {
            if (!s.pool) {
                const size_t e = img + (size_t)ys * rs + (size_t)xs * s.C + slot * 8;
                V16 raw;
                raw.u = *reinterpret_cast<const uint4 *>(s.x + e);
                if (plain) val = raw;
                else if (s.res) { V16 r; r.u = *reinterpret_cast<const uint4 *>(s.res + e); val = xform8(raw, &r, t, relu, f16); }
                else val = xform8(raw, nullptr, t, relu, f16);
            } else {
                V16 raw[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                     // all four loads first (clamped address), then the math
                    int yy = 2 * ys + (q >> 1), xx = 2 * xs + (q & 1);
                    ok[q] = q == 0 || (yy < s.Hs && xx < s.Ws);   // ceil-mode partial window
                    yy = yy < s.Hs ? yy : s.Hs - 1;
                    xx = xx < s.Ws ? xx : s.Ws - 1;
                    raw[q].u = *reinterpret_cast<const uint4 *>(s.x + img + (size_t)yy * rs + (size_t)xx * s.C + slot * 8);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const V16 tv = plain ? raw[q] : xform8(raw[q], nullptr, t, relu, f16);
                    val = q == 0 ? tv : (ok[q] ? max8(val, tv, relu) : val);
                }
            }
        }
        *reinterpret_cast<uint4 *>(out + pix * s.C + slot * 8) = val.u;
    }
}

}  // namespace

extern "C" size_t cdnet_conv_packed_weight_elems(int Cout, int Cin_padded_chunks, int taps, int CK, int BN, int npar) {
    return (size_t)npar * cdiv(Cout, BN) * Cin_padded_chunks * taps * CK * BN;
}

static int fill_pack_desc(PackDesc &d, const float *w, void *packed, int Cout, int Cin, int KH, int KW, int CK, int BN, int mode, int p,
                          const char *who) {
    const int split = (mode & 16) ? 1 : 0;            // CDNET_PACK_SPLIT: hi | lo images (fp32-precision convolutions)
    mode &= 15;
    CDNET_REQUIRE(w && packed, "%s: null pointer", who);
    CDNET_REQUIRE(CK % 16 == 0 && BN % 32 == 0 && Cin % CK == 0 && mode >= 0 && mode <= 7, "%s: Cin=%d CK=%d BN=%d mode=%d", who, Cin, CK, BN, mode);
    const int taps = mode == 2 ? 4 : (mode == 3 ? 1 : ((mode == 4 || mode == 6 || mode == 7) ? 9 : (mode == 5 ? 1 : KH * KW)));
    const int nchunk = Cin / CK, ntile = cdiv(Cout, BN);
    const size_t per = (size_t)ntile * nchunk * taps * CK * BN * (split ? 2 : 1);
    d.split = split;
    d.scale = nullptr;
    d.w = w; d.out = (unsigned short *)packed + (size_t)p * per;
    d.Cout = Cout; d.Cin = Cin; d.KH = KH; d.KW = KW; d.CK = CK; d.BN = BN; d.nchunk = nchunk; d.ntile = ntile; d.TAPS = taps; d.mode = mode;
    d.parity = p;
    d.block0 = 0;
    d.nblocks = (unsigned)((per + 2047) / 2048);          // 8 elements per thread
    if (d.nblocks > 4096) d.nblocks = 4096;
    return CDNET_OK;
}

extern "C" int cdnet_pack_conv_weights(const float *w, void *packed, int Cout, int Cin, int KH, int KW, int CK, int BN,
                                       int mode, void *stream) {
    const int npar = ((mode & 15) == 2 || (mode & 15) == 3) ? 4 : 1;
    for (int p = 0; p < npar; ++p) {
        PackDesc d;
        int rc = fill_pack_desc(d, w, packed, Cout, Cin, KH, KW, CK, BN, mode, p, "cdnet_pack_conv_weights");
        if (rc) return rc;
        pack_weights_kernel<<<d.nblocks, 256, 0, (hipStream_t)stream>>>(d);
    }
    return check_launch("cdnet_pack_conv_weights");
}

extern "C" int cdnet_pack_conv_weights_scaled(const float *w, const float *cout_scale, void *packed, int Cout, int Cin, int KH, int KW, int CK,
                                              int BN, int mode, void *stream) {
    CDNET_REQUIRE(cout_scale && (mode & 15) == 0, "cdnet_pack_conv_weights_scaled: forward Conv2d packs (mode 0) with a scale vector");
    PackDesc d;
    int rc = fill_pack_desc(d, w, packed, Cout, Cin, KH, KW, CK, BN, mode, 0, "cdnet_pack_conv_weights_scaled");
    if (rc) return rc;
    d.scale = cout_scale;
    pack_weights_kernel<<<d.nblocks, 256, 0, (hipStream_t)stream>>>(d);
    return check_launch("cdnet_pack_conv_weights_scaled");
}

extern "C" size_t cdnet_pack_batch_table_bytes(int n_jobs) { return (size_t)n_jobs * 4 * sizeof(PackDesc); }

extern "C" int cdnet_pack_conv_weights_batch(const cdnet_pack_job *jobs, int n_jobs, void *table, size_t table_bytes, int upload,
                                             void *stream) {
    CDNET_REQUIRE(jobs && n_jobs > 0 && table, "cdnet_pack_conv_weights_batch: null pointer / no jobs");
    CDNET_REQUIRE(table_bytes >= cdnet_pack_batch_table_bytes(n_jobs), "cdnet_pack_conv_weights_batch: table too small");
    static thread_local std::vector<PackDesc> host;
    host.clear();
    unsigned blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const cdnet_pack_job &J = jobs[j];
        const int npar = ((J.mode & 15) == 2 || (J.mode & 15) == 3) ? 4 : 1;
        for (int p = 0; p < npar; ++p) {
            PackDesc d;
            int rc = fill_pack_desc(d, J.w, J.packed, J.Cout, J.Cin, J.KH, J.KW, J.CK, J.BN, J.mode, p, "cdnet_pack_conv_weights_batch");
            if (rc) return rc;
            d.block0 = blocks;
            blocks += d.nblocks;
            host.push_back(d);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (upload) {
        // (synchronous with respect to the host buffer: the table is rebuilt on every call)
        if (hipMemcpyAsync(table, host.data(), host.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return check_launch("cdnet_pack_conv_weights_batch(upload)");
    }
    pack_weights_batch_kernel<<<blocks, 256, 0, st>>>((const PackDesc *)table, (int)host.size());
    return check_launch("cdnet_pack_conv_weights_batch");
}

extern "C" int cdnet_conv_forward(const cdnet_conv_args *args, void *stream) {
    CDNET_REQUIRE(args, "cdnet_conv_forward: null args");
    const ConvArgs &A = *reinterpret_cast<const ConvArgs *>(args);
    CDNET_REQUIRE(A.nsrc >= 1 && A.nsrc <= 2 && A.w && (A.out || A.dot_out), "cdnet_conv_forward: bad pointers / nsrc=%d", A.nsrc);
    CDNET_REQUIRE(!A.dot_out || (A.dot_w && !A.f32 && !A.pool_out), "cdnet_conv_forward: dot_out needs dot_w, the 16-bit path and no pool_out");
    CDNET_REQUIRE(A.N > 0 && A.H > 0 && A.W > 0 && A.Cout > 0 && A.Cout % 8 == 0, "cdnet_conv_forward: bad size (Cout must be a multiple of 8)");
    {
        int ctot_xf = 0;
        for (int i = 0; i < A.nsrc; ++i) ctot_xf += A.src[i].C;
        CDNET_REQUIRE(ctot_xf <= XF_MAX, "cdnet_conv_forward: %d source channels exceed the %d-entry scale/shift table", ctot_xf, XF_MAX);
    }
    CDNET_REQUIRE(A.ws == 0 || A.ws == 2, "cdnet_conv_forward: ws must be 0 or 2 (BatchNorm-backward statistics epilogue)");
    if (A.f32) {
        CDNET_REQUIRE(A.f32 == 1, "cdnet_conv_forward: f32 must be 0 or 1");
        for (int i = 0; i < A.nsrc; ++i)
            CDNET_REQUIRE(A.src[i].f16 == 2, "cdnet_conv_forward(f32): every source must be fp32 (f16 = 2)");
    } else {
        for (int i = 0; i < A.nsrc; ++i)
            CDNET_REQUIRE(A.src[i].f16 == 0 || A.src[i].f16 == 1, "cdnet_conv_forward: fp32 sources need args.f32 = 1");
    }
    if (A.eres && A.ws != 2) {
        CDNET_REQUIRE(A.ostride == 1 && A.npar == 1 && A.out_coff == 0 && A.out_cstride == A.Cout && !A.stats && !A.orelu &&
                      !A.out_f16 && ((A.eres_scale == nullptr) == (A.eres_shift == nullptr)),
                      "cdnet_conv_forward: fused residual epilogue needs a dense bf16 output, no statistics and no ReLU before the add");
    }
    int nchunk = 0;
    for (int i = 0; i < A.nsrc; ++i) {
        CDNET_REQUIRE(A.src[i].x && A.src[i].C % A.CK == 0, "cdnet_conv_forward: source %d channels %d not a multiple of CK=%d",
                      i, A.src[i].C, A.CK);
        CDNET_REQUIRE(!(A.src[i].pool && A.src[i].res), "cdnet_conv_forward: pool+residual source unsupported");
        nchunk += A.src[i].C / A.CK;
    }
    // (the second source of a two-source 3x3 launch of the 16-bit path may carry ONE padding chunk of zero weights beyond its channels - an even
    //  chunk count for conv_ws16_kernel's out-image form with pair requests; what a kernel reads for that chunk is the neighbouring pixel's
    //  channels, or zeros past the tensor's end: finite values times zero weights)
    const bool padded = A.nchunk == nchunk + 1 && A.taps == 9 && A.nsrc == 2 && !A.f32;
    CDNET_REQUIRE(nchunk == A.nchunk || padded, "cdnet_conv_forward: nchunk %d != %d", A.nchunk, nchunk);
    CDNET_REQUIRE((A.taps == 9 && A.npar == 1 && A.ostride == 1) || (A.taps == 1 && A.npar == 1 && A.ostride == 1) ||
                  (A.taps == 4 && A.npar == 4 && A.ostride == 2) || (A.taps == 1 && A.npar == 4 && A.ostride == 2),
                  "cdnet_conv_forward: taps=%d npar=%d ostride=%d", A.taps, A.npar, A.ostride);
    CDNET_REQUIRE(A.out_cstride % 8 == 0 && A.out_coff % 8 == 0, "cdnet_conv_forward: output channel slice must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    CDNET_REQUIRE(A.taps1 == 0 || A.taps1 == A.taps || (A.taps1 == 1 && A.taps == 9 && A.nsrc == 2), "cdnet_conv_forward: taps1 = %d (0, taps, or 1 beside a nine-tap first source)", A.taps1);
    if (A.f32) return conv_forward_f32(A, st);
    {
        const int rc = conv_forward_ws16(A, st);
        if (rc >= 0) return rc;
        CDNET_REQUIRE(A.taps1 == 0 || A.taps1 == A.taps, "cdnet_conv_forward: a one-tap second source runs on conv_ws16_kernel only (ask cdnet_conv_ws_eligible)");
        CDNET_REQUIRE(!A.pool_out, "cdnet_conv_forward: the fused max-pool output needs conv_ws16_kernel's out-image form (ask cdnet_conv_ws_eligible)");
        CDNET_REQUIRE(!A.dot_out, "cdnet_conv_forward: the fused 1x1 classifier (dot_out) needs conv_ws16_kernel's out-image form with resident weights (ask cdnet_conv_ws_eligible)");
        // (only conv_ws16_kernel / conv_ws_kernel read through bounded buffer descriptors: the one-tile kernels would index the scale / shift
        //  table and the tensor past the second source's channels)
        CDNET_REQUIRE(!padded, "cdnet_conv_forward: a padding chunk (nchunk = real + 1) runs on conv_ws16_kernel only (ask cdnet_conv_ws_eligible)");
    }
    static const int dbg = getenv("CDNET_CONV_DEBUG") ? atoi(getenv("CDNET_CONV_DEBUG")) : 0;
    if (!(A.debug & 32) && A.taps == 9 && A.npar == 1 && A.ostride == 1 && A.tile == 16 && A.CK == 16 && (A.BN == 64 || A.BN == 32)) {
        const int rc = A.BN == 64 ? try_launch_conv_ws<64, 9>(A, st) : try_launch_conv_ws<32, 9>(A, st);
        if (rc >= 0) return rc;
    }
    for (int i = 0; i < A.nsrc; ++i)
        CDNET_REQUIRE(A.src[i].relu >= 0 && A.src[i].relu <= 2, "cdnet_conv_forward: source %d: relu = %d (the BatchNorm-backward source of round 2, relu = 3, was removed)", i, A.src[i].relu);
    CDNET_REQUIRE(A.ws != 2, "cdnet_conv_forward: the BatchNorm-backward statistics epilogue (ws = 2) exists in fp32 mode on conv_ws32_kernel only "
                             "(ask cdnet_conv_ws_eligible first)");
    if (dbg) { ConvArgs B = A; B.debug = dbg; if (B.taps == 9) return dispatch_conv<9>(B, st); }
    if (A.debug & 32) { ConvArgs B = A; B.debug = 0; if (B.taps == 9) return dispatch_conv<9>(B, st); if (B.taps == 4) return dispatch_conv<4>(B, st); return dispatch_conv<1>(B, st); }
    if (A.taps == 9) return dispatch_conv<9>(A, st);
    if (A.taps == 4) return dispatch_conv<4>(A, st);
    return dispatch_conv<1>(A, st);
}

/* non-zero when cdnet_conv_forward would run these arguments on a producer / consumer kernel: 2 = conv_ws16_kernel, 1 = conv_ws_kernel /
 * conv_ws32_kernel */
extern "C" int cdnet_conv_ws_eligible(const cdnet_conv_args *args) {
    if (!args) return 0;
    const ConvArgs &A = *reinterpret_cast<const ConvArgs *>(args);
    if (A.f32) return (!A.dot_out && conv_forward_f32_ws(A, nullptr, true) == CDNET_OK) ? 1 : 0;
    if (conv_forward_ws16(A, nullptr, true) == CDNET_OK) return 2;
    if ((A.taps1 != 0 && A.taps1 != A.taps) || A.pool_out || A.dot_out) return 0;
    if (A.debug & 32) return 0;
    if (!(A.taps == 9 && A.npar == 1 && A.ostride == 1 && A.tile == 16 && A.CK == 16 && (A.BN == 64 || A.BN == 32))) return 0;
    const int rc = A.BN == 64 ? try_launch_conv_ws<64, 9>(A, nullptr, true) : try_launch_conv_ws<32, 9>(A, nullptr, true);
    return rc == CDNET_OK ? 1 : 0;
}

extern "C" int cdnet_src_materialize(const cdnet_conv_src *src, int N, int H, int W, uint16_t *out, void *stream) {
    CDNET_REQUIRE(src && out && src->x && N > 0 && H > 0 && W > 0, "cdnet_src_materialize: bad args");
    const ConvSrc &s = *reinterpret_cast<const ConvSrc *>(src);
    CDNET_REQUIRE(s.C >= 8 && s.C % 8 == 0 && s.Hs > 0 && s.Ws > 0, "cdnet_src_materialize: C=%d must be a multiple of 8", s.C);
    CDNET_REQUIRE(!(s.pool && s.res), "cdnet_src_materialize: pooled sources carry no residual");
    CDNET_REQUIRE((s.scale == nullptr) == (s.shift == nullptr), "cdnet_src_materialize: scale and shift come together");
    if (s.f16 == 2) return materialize_f32(s, N, H, W, out, (hipStream_t)stream);
    const size_t total = (size_t)N * H * W * (s.C / 8);
    size_t g = (total + 255) / 256;
    g = g > 8192 ? 8192 : g;
    materialize_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(s, N, H, W, out);
    return check_launch("cdnet_src_materialize");
}
