// fp32-storage variant of the implicit-GEMM convolutions (the reference's precision): NHWC fp32 activations in HBM,
// every product a*b evaluated on the matrix cores as three bf16 MFMAs over split operands
//     a = a_hi + a_lo,  b = b_hi + b_lo  (hi = bf16(x), lo = bf16(x - hi): 16 significant bits each side)
//     a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (the dropped a_lo*b_lo term is 2^-16 of the product)
// accumulated in fp32 (v_mfma_f32_32x32x16_bf16).  gfx950 has no TF32/xf32 matrix path and its fp32 MFMA runs at 1/16 of
// the bf16 rate; three bf16 MFMAs cost 3/16 of that and keep ~fp32 products (relative error <= 2^-16 per product before
// fp32 accumulation - tighter than cuDNN's TF32 default for the reference on recent NVIDIA parts).
//
// Same cdnet_conv_args / cdnet_conv_src contract as conv.hip with args.f32 = 1: x / res / eres / out point to fp32
// tensors, `w` to the split pack (cdnet_pack_conv_weights, mode | 16: per chunk the hi image followed by the lo image).
// Differences from the 16-bit kernel: 16-channel chunks only; no pooled sources (max-pools are materialised:
// cdnet_src_materialize with f16 = 2); the tile leaves from the accumulators with 128-byte row segments (no LDS out tile).
//
// Replaces the same reference lines as conv.hip: models/dam/model_unet_rev1.py:86-170,244-287, models/unet.py:8-50,90-106.
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include <stdlib.h>

using namespace cdnet;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));     // register staging types (HIP's float4 / uint4 structs defeat SROA)

namespace {

// 8 fp32 values -> hi / lo bf16 vectors (packed pairs: v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ void split8(const float *v, u32x4 &hi, u32x4 &lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        const f32x2 hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        const bf16x2 l = __builtin_convertvector(x - hf, bf16x2);
        hi[k] = hb;
        lo[k] = __builtin_bit_cast(unsigned, l);
    }
}

constexpr int CK = 16;
constexpr int PSTR = CK * 2 + 16;          // padded pixel stride of one LDS plane (bank-conflict-free b128 reads)
constexpr int XF_MAX = 2048;

template <int TH, int TW, int BN, int TAPS>
struct Lds32 {
    static constexpr int NPIX = (TH + 2) * (TW + 2);
    static constexpr int A_PLANE = NPIX * PSTR;
    static constexpr int B_PLANE = TAPS * CK * BN * 2;
    static constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    static constexpr int STATS_BYTES = 4 * 2 * BN * 4;
    static constexpr int MAIN = ((STAGE > STATS_BYTES ? STAGE : STATS_BYTES) + 15) / 16 * 16;
    static int bytes(int ctot) { return MAIN + 2 * ((ctot + 7) / 8 * 8) * 4; }
};


// The finished tile leaves from the accumulators: lane (l31, half) of a 32 x 32 block owns output channel co and the pixels
// m0 + (r & 3) + 8 (r >> 2) + 4 half, r = 0..15.  Fast path for a full 16 x 16 tile, a full channel block and no residual input: one base
// pointer per block and 32-bit tile-relative offsets (the general path's 64-bit index arithmetic and bounds tests per element made the
// epilogue as long as the tile's MFMAs: 7.7 of 15 us per tile on the 64 -> 64 layer).
template <int MPW, int NPW, int TW>
__device__ __forceinline__ void store_tile_f32(const ConvArgs &A, const f32x16 (&acc)[MPW][NPW], int n, int y0, int x0, int mblock0, int nblock0,
                                               int half, int l31, int cout0, int pa_, int pb_) {
    const int Ho = A.H * A.ostride, Wo = A.W * A.ostride;
    float *out = reinterpret_cast<float *>(A.out);
    const float *eres = reinterpret_cast<const float *>(A.eres);
    const bool full = y0 + 16 <= A.H && x0 + TW <= A.W && cout0 + (nblock0 + NPW) * 32 <= A.Cout && A.ostride == 1 && TW == 16 &&
                      !(A.debug & 16);                       // (debug bit 16: tests compare the two epilogues bit for bit)
    if (full) {
        const int cs = A.out_cstride, ce = A.Cout;                    // (the residual input is a dense [N][H][W][Cout] tensor)
        float *tb = out + (((size_t)n * A.H + y0) * A.W + x0) * cs + A.out_coff;
        const float *te = eres ? eres + (((size_t)n * A.H + y0) * A.W + x0) * ce : nullptr;
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) {
            const int co = cout0 + (nblock0 + ni) * 32 + l31;
            const float osc = A.oscale ? A.oscale[co] : 1.f;
            const float osh = fmaf(A.bias ? A.bias[co] : 0.f, osc, A.oshift ? A.oshift[co] : 0.f);
            const float esc = (eres && A.eres_scale) ? A.eres_scale[co] : 1.f;
            const float esh = (eres && A.eres_shift) ? A.eres_shift[co] : 0.f;
            float *ob = tb + co;
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                // block (mblock0 + mi): rows 2 (mblock0 + mi), +1 of the tile; this lane: pixel 4 half + (r & 3) + 8 (r >> 2) of the 32
                const int rowb = ((mblock0 + mi) * 2) * A.W + 4 * half;
                float ev[16];
                if (eres) {                        // the other branch of the residual unit, requested up front
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = (r & 3) + 8 * (r >> 2);
                        ev[r] = te[(rowb + (p / 16) * A.W + (p % 16)) * ce + co];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (r & 3) + 8 * (r >> 2);            // + 4 half: 0..31 -> (row p / 16, column p % 16); 4 half never carries
                    const int off = (rowb + (p / 16) * A.W + (p % 16)) * cs;
                    float v = fmaf(acc[mi][ni][r], osc, osh);
                    if (A.orelu) v = fmaxf(v, 0.f);
                    if (eres) {
                        v = fmaf(ev[r], esc, esh) + v;
                        if (A.eres_relu) v = fmaxf(v, 0.f);
                    }
                    ob[off] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NPW; ++ni) {
        const int co = cout0 + (nblock0 + ni) * 32 + l31;
        const bool cok = co < A.Cout;
        const float osc = (A.oscale && cok) ? A.oscale[co] : 1.f;
        const float osh = fmaf((A.bias && cok) ? A.bias[co] : 0.f, osc, (A.oshift && cok) ? A.oshift[co] : 0.f);
        const float esc = (A.eres && A.eres_scale && cok) ? A.eres_scale[co] : 1.f;
        const float esh = (A.eres && A.eres_shift && cok) ? A.eres_shift[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            float ev[16];
            if (A.eres) {                          // the other branch of the residual unit, requested up front
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (mblock0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    int y = y0 + m / TW, x = x0 + m % TW;
                    y = y < A.H ? y : A.H - 1; x = x < A.W ? x : A.W - 1;
                    ev[r] = eres[(((size_t)n * A.H + y) * A.W + x) * A.Cout + (cok ? co : 0)];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (mblock0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int y = y0 + m / TW, x = x0 + m % TW;
                float v = fmaf(acc[mi][ni][r], osc, osh);
                if (A.orelu) v = fmaxf(v, 0.f);
                if (A.eres) {
                    v = fmaf(ev[r], esc, esh) + v;
                    if (A.eres_relu) v = fmaxf(v, 0.f);
                }
                if (cok && y < A.H && x < A.W) {
                    const size_t opix = ((size_t)n * Ho + (y * A.ostride + pa_)) * Wo + (x * A.ostride + pb_);
                    out[opix * A.out_cstride + A.out_coff + co] = v;
                }
            }
        }
    }
}

template <int TH, int TW, int BN, int WM, int WN, int TAPS>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(ConvArgs A) {
    using L = Lds32<TH, TW, BN, TAPS>;
    constexpr int HW_ = TW + 2, NPIX = L::NPIX;
    constexpr int MT = TH * TW / 32, NT = BN / 32;
    constexpr int MPW = MT / WM, NPW = NT / WN;
    constexpr int VPP = CK / 8;                                   // 8-channel vectors per pixel
    constexpr int NA = (NPIX * VPP + 255) / 256;                  // staged vectors per thread
    constexpr int NBV = 2 * L::B_PLANE / 16;                      // 16-byte vectors of one weight chunk (hi | lo)
    constexpr int NB = (NBV + 255) / 256;
    static_assert(MT % WM == 0 && NT % WN == 0 && WM * WN == 4, "wave tiling");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_a = smem;                                  // [hi plane][lo plane]
    unsigned char *lds_b = smem + 2 * L::A_PLANE;                 // [hi image][lo image]
    float (*s_stats)[2][BN] = reinterpret_cast<float (*)[2][BN]>(smem);      // epilogue only
    float *s_xf = reinterpret_cast<float *>(smem + L::MAIN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int c0n = A.src[0].C, ctot = c0n + (A.nsrc > 1 ? A.src[1].C : 0);
    const int xfs = (ctot + 7) / 8 * 8;
    for (int c = tid; c < ctot; c += 256) {
        const ConvSrc &S = c < c0n ? A.src[0] : A.src[1];
        const int cc = c < c0n ? c : c - c0n;
        s_xf[c] = S.scale ? S.scale[cc] : 1.f;
        s_xf[xfs + c] = S.shift ? S.shift[cc] : 0.f;
    }

    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    // XCD-aware order (see conv.hip): XCD k serves the k-th contiguous eighth of the (tile, cout block) sequence, the cout blocks of a
    // tile next to each other
    int tile, cout_tile;
    {
        const int NC = (int)gridDim.y, lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        const int T = (int)gridDim.x * NC, q = T >> 3, rem = T & 7;
        const int xcd = lin & 7, idx = lin >> 3;
        const int seq = xcd < rem ? xcd * (q + 1) + idx : rem * (q + 1) + (xcd - rem) * q + idx;
        tile = seq / NC;
        cout_tile = seq - tile * NC;
    }
    const int z = tile / tiles_img, rt = tile - z * tiles_img;
    const int n = z / A.npar, par = z - n * A.npar;
    const int ty_ = rt / tiles_x;
    const int y0 = ty_ * TH, x0 = (rt - ty_ * tiles_x) * TW;

    // staging geometry of this thread's vectors, per source (element offset inside the image, -1 = zero fill)
    int eoff[2][NA];
    const int slot = tid % VPP;
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        if (si >= A.nsrc) break;
        const ConvSrc &s = A.src[si];
        const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * 256;
            const int pix = v / VPP;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const int ys = y - s.off_y, xs = x - s.off_x;
            const bool ok = v < NPIX * VPP && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W && (unsigned)ys < (unsigned)s.Hs &&
                            (unsigned)xs < (unsigned)s.Ws;
            eoff[si][i] = ok ? ys * rs + xs * s.C + slot * 8 : -1;
        }
    }
    auto chunk_src = [&](int c, int &si, int &cc0) {
        const int n0 = A.src[0].C / CK;
        if (c < n0) { si = 0; cc0 = c * CK; } else { si = 1; cc0 = (c - n0) * CK; }
    };
    const unsigned short *wbase = A.w + ((size_t)(par * gridDim.y + cout_tile) * A.nchunk) * (size_t)(L::B_PLANE);      // u16 elements: 2 planes x B_PLANE/2

    // register prefetch of one chunk: halo vectors (two float4 each) and the weight image
    f32x4 pa[NA][2];
    u32x4 pb[NB];
    auto issue = [&](int chunk) {
        int si, cc0;
        chunk_src(chunk, si, cc0);
        const ConvSrc &s = A.src[si];
        const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
        const float *base = reinterpret_cast<const float *>(s.x) + (size_t)n * s.Hs * rs + cc0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = si ? eoff[1][i] : eoff[0][i];
            const f32x4 *p = reinterpret_cast<const f32x4 *>(base + (e >= 0 ? e : 0));
            pa[i][0] = p[0];
            pa[i][1] = p[1];
        }
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wbase + (size_t)chunk * L::B_PLANE);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            if (v < NBV) pb[i] = wsrc[v];
        }
    };
    auto commit = [&](int chunk) {
        int si, cc0;
        chunk_src(chunk, si, cc0);
        const ConvSrc &s = A.src[si];
        const bool on = s.scale != nullptr, relu = s.relu != 0;
        const float *xf = s_xf + (si ? c0n : 0) + cc0 + slot * 8;
        float sc[8], sh[8];
        if (on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = xf[j]; sh[j] = xf[xfs + j]; }
        }
        f32x4 rr[NA][2];
        if (s.res) {
            const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
            const float *rbase = reinterpret_cast<const float *>(s.res) + (size_t)n * s.Hs * rs + cc0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e = si ? eoff[1][i] : eoff[0][i];
                const f32x4 *p = reinterpret_cast<const f32x4 *>(rbase + (e >= 0 ? e : 0));
                rr[i][0] = p[0];
                rr[i][1] = p[1];
            }
        }
        unsigned char *dst0 = lds_a + (tid / VPP) * PSTR + slot * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (tid + i * 256 >= NPIX * VPP) continue;
            const int e = si ? eoff[1][i] : eoff[0][i];
            u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
            if (e >= 0) {
                float v[8] = {pa[i][0][0], pa[i][0][1], pa[i][0][2], pa[i][0][3], pa[i][1][0], pa[i][1][1], pa[i][1][2], pa[i][1][3]};
                if (on) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
                }
                if (s.res) {
                    const float r[8] = {rr[i][0][0], rr[i][0][1], rr[i][0][2], rr[i][0][3], rr[i][1][0], rr[i][1][1], rr[i][1][2], rr[i][1][3]};
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                split8(v, hi, lo);
            }
            unsigned char *d = dst0 + i * (256 / VPP) * PSTR;
            *reinterpret_cast<u32x4 *>(d) = hi;
            *reinterpret_cast<u32x4 *>(d + L::A_PLANE) = lo;
        }
        u32x4 *bdst = reinterpret_cast<u32x4 *>(lds_b);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            if (v < NBV) bdst[v] = pb[i];
        }
    };

    int abase[MPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int m = (wm * MPW + mi) * 32 + l31;
        abase[mi] = ((m / TW) * HW_ + m % TW) * PSTR + half * 16;
    }
    const int bbase = half * BN * 16 + (wn * NPW * 32 + l31) * 16;
    int toff[TAPS];
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
    }
    f32x16 acc[MPW][NPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue(0);
    for (int chunk = 0; chunk < A.nchunk; ++chunk) {
        __syncthreads();                          // the previous chunk's fragment reads are done (first pass: the xf table is written)
        commit(chunk);
        __syncthreads();
        if (chunk + 1 < A.nchunk) issue(chunk + 1);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            bf16x8 ah[MPW], al[MPW], bh[NPW], bl[NPW];
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                ah[mi] = *reinterpret_cast<const bf16x8 *>(lds_a + abase[mi] + toff[t]);
                al[mi] = *reinterpret_cast<const bf16x8 *>(lds_a + L::A_PLANE + abase[mi] + toff[t]);
            }
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                bh[ni] = *reinterpret_cast<const bf16x8 *>(lds_b + bbase + t * 2 * BN * 16 + ni * 512);
                bl[ni] = *reinterpret_cast<const bf16x8 *>(lds_b + L::B_PLANE + bbase + t * 2 * BN * 16 + ni * 512);
            }
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni) {
                    // small terms first
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
    }
    __syncthreads();

    // ---------------- epilogue ----------------
    const int cout0 = cout_tile * BN;
    if (A.stats) {
        float ssum[NPW], ssq[NPW];
        const bool full_tile = y0 + TH <= A.H && x0 + TW <= A.W && !(A.debug & 16);      // (no per-element bounds tests then; the sums are the same)
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) {
            ssum[ni] = 0.f; ssq[ni] = 0.f;
            if (full_tile) {
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mi][ni][r];
                        ssum[ni] += v;
                        ssq[ni] = fmaf(v, v, ssq[ni]);
                    }
            } else {
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
            ssum[ni] += __shfl_xor(ssum[ni], 32);
            ssq[ni] += __shfl_xor(ssq[ni], 32);
            if (half == 0) {
                s_stats[wave][0][(wn * NPW + ni) * 32 + l31] = ssum[ni];
                s_stats[wave][1][(wn * NPW + ni) * 32 + l31] = ssq[ni];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            const int wn_of = col / (NPW * 32);
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < WM; ++k) v += s_stats[k * WN + wn_of][which][col];
            const int co = cout0 + col;
            if (co < A.Cout) A.stats[((size_t)tile * 2 + which) * A.Cout + co] = v;
        }
    }
    // a lane owns one output channel; lanes 0..31 / 32..63 write two 128-byte row segments per instruction
    store_tile_f32<MPW, NPW, TW>(A, acc, n, y0, x0, wm * MPW, wn * NPW, half, l31, cout0, par >> 1, par & 1);
}

template <int TH, int TW, int BN, int WM, int WN, int TAPS>
int launch_conv32(const ConvArgs &A, hipStream_t st) {
    using L = Lds32<TH, TW, BN, TAPS>;
    int ctot = 0;
    for (int i = 0; i < A.nsrc; ++i) ctot += A.src[i].C;
    const int smem = L::bytes(ctot);
    auto kern = conv_f32_kernel<TH, TW, BN, WM, WN, TAPS>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes(XF_MAX)) != hipSuccess)
            return check_launch("hipFuncSetAttribute(conv32)");
        attr_done = true;
    }
    dim3 grid(cdiv(A.W, TW) * cdiv(A.H, TH) * A.N * A.npar, cdiv(A.Cout, BN), 1);
    kern<<<grid, 256, smem, st>>>(A);
    return check_launch("conv_f32_kernel");
}

template <int TAPS>
int dispatch_conv32(const ConvArgs &A, hipStream_t st) {
    if (A.tile == 16 && A.BN == 64) return launch_conv32<16, 16, 64, 4, 1, TAPS>(A, st);
    if (A.tile == 16 && A.BN == 32) return launch_conv32<16, 16, 32, 4, 1, TAPS>(A, st);
    if (A.tile == 8 && A.BN == 64) return launch_conv32<8, 8, 64, 2, 2, TAPS>(A, st);
    set_error("cdnet_conv_forward(f32): unsupported configuration tile=%d CK=%d BN=%d", A.tile, A.CK, A.BN);
    return CDNET_E_ARG;
}

// fp32 variant of materialize_kernel (conv.hip): the source with its pending transform (scale/shift, residual, ReLU, 2x2
// max-pool floor / ceil mode, pad offset) written out as a plain fp32 tensor
__global__ __launch_bounds__(256) void materialize_f32_kernel(const ConvSrc s, int N, int H, int W, float *__restrict__ out) {
    const int VPP = s.C / 4;
    const size_t total = (size_t)N * H * W * VPP;
    const bool relu = s.relu != 0, on = s.scale != nullptr;
    const int Hl = s.pool ? (s.Hs + (s.pool == 2)) / 2 : s.Hs, Wl = s.pool ? (s.Ws + (s.pool == 2)) / 2 : s.Ws;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    const float *sx = reinterpret_cast<const float *>(s.x), *sr = reinterpret_cast<const float *>(s.res);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % VPP) * 4;
        const size_t pix = i / VPP;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) { sc = *reinterpret_cast<const float4 *>(s.scale + c); sh = *reinterpret_cast<const float4 *>(s.shift + c); }
        const size_t img = (size_t)n * s.Hs * rs;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ys = y - s.off_y, xs = x - s.off_x;
        auto xf = [&](float4 v, const float4 *r) {
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (r) { v.x += r->x; v.y += r->y; v.z += r->z; v.w += r->w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            return v;
        };
        if (ys >= 0 && ys < Hl && xs >= 0 && xs < Wl) {
            if (!s.pool) {
                const size_t e = img + (size_t)ys * rs + (size_t)xs * s.C + c;
                const float4 raw = *reinterpret_cast<const float4 *>(sx + e);
                if (sr) { const float4 r = *reinterpret_cast<const float4 *>(sr + e); val = xf(raw, &r); }
                else val = xf(raw, nullptr);
            } else {
                float4 raw[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int yy = 2 * ys + (q >> 1), xx = 2 * xs + (q & 1);
                    ok[q] = q == 0 || (yy < s.Hs && xx < s.Ws);
                    yy = yy < s.Hs ? yy : s.Hs - 1;
                    xx = xx < s.Ws ? xx : s.Ws - 1;
                    raw[q] = *reinterpret_cast<const float4 *>(sx + img + (size_t)yy * rs + (size_t)xx * s.C + c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = xf(raw[q], nullptr);
                    if (q == 0) val = t;
                    else if (ok[q]) { val.x = fmaxf(val.x, t.x); val.y = fmaxf(val.y, t.y); val.z = fmaxf(val.z, t.z); val.w = fmaxf(val.w, t.w); }
                }
            }
        }
        *reinterpret_cast<float4 *>(out + pix * s.C + c) = val;
    }
}

}  // namespace

namespace cdnet {

int conv_forward_f32(const ConvArgs &A, hipStream_t st) {
    CDNET_REQUIRE(A.CK == 16, "cdnet_conv_forward(f32): CK must be 16 (got %d)", A.CK);
    for (int i = 0; i < A.nsrc; ++i)
        CDNET_REQUIRE(!A.src[i].pool, "cdnet_conv_forward(f32): pooled sources must be materialised (cdnet_src_materialize)");
    CDNET_REQUIRE(!A.out_f16, "cdnet_conv_forward(f32): out_f16 must be 0");
    {
        const int rc = conv_forward_f32_ws(A, st);               // the producer / consumer kernel where it applies
        if (rc >= 0) return rc;
    }
    CDNET_REQUIRE(A.ws != 2, "cdnet_conv_forward(f32): the BatchNorm-backward statistics epilogue (ws = 2) needs the producer / consumer kernel "
                             "(ask cdnet_conv_ws_eligible first)");
    CDNET_REQUIRE(A.taps1 == 0 || A.taps1 == A.taps, "cdnet_conv_forward(f32): a one-tap second source runs on conv_ws32_kernel only (ask cdnet_conv_ws_eligible)");
    CDNET_REQUIRE(!A.pool_out, "cdnet_conv_forward(f32): the fused max-pool output rides in conv_ws32_kernel's epilogue only (ask cdnet_conv_ws_eligible)");
    if (A.taps == 9) return dispatch_conv32<9>(A, st);
    if (A.taps == 4) return dispatch_conv32<4>(A, st);
    return dispatch_conv32<1>(A, st);
}

int materialize_f32(const ConvSrc &s, int N, int H, int W, void *out, hipStream_t st) {
    const size_t total = (size_t)N * H * W * (s.C / 4);
    size_t g = (total + 255) / 256;
    g = g > 8192 ? 8192 : g;
    materialize_f32_kernel<<<(int)g, 256, 0, st>>>(s, N, H, W, reinterpret_cast<float *>(out));
    return check_launch("cdnet_src_materialize(f32)");
}

}  // namespace cdnet
