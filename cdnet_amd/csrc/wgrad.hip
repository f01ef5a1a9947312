// Backward-weight of the convolutions on the gfx950 matrix cores.
//   dW[tap][ci][co] = sum over pixels p of  A[p + tap][ci] * G[p][co]
// A = the layer's (transformed) input, G = gradient w.r.t. the layer's raw output (bf16).  Both live in LDS as
// [pixel][channel] tiles - exactly what the forward staging produces - and the MFMA operands, which need eight
// consecutive PIXELS of one channel per lane, are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16):
// no transposed copy of any activation ever exists in HBM.
//
// One workgroup (4 waves) owns a (CI_T*32 input channels) x (CO_T*32 output channels) block of dW for ALL taps
// (CI_T*CO_T = 4: each wave one 32x32 quadrant x TAPS accumulator tiles) and walks a slice of the 8x16-pixel tiles
// (split-K over pixels).  Partial slabs are written fp32 and summed in a fixed order by wgrad_reduce (deterministic),
// which also scatters into the PyTorch weight layout.
//
// Replaces the weight-gradient half of loss.backward() (train_util_dam.py:307) for nn.Conv2d / nn.ConvTranspose2d.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include "xform.h"

using namespace cdnet;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

union V16 {
    uint4 u;
    unsigned short h[8];
};

constexpr int TH = 8, TW = 16, HALO_W = TW + 2, NPIX_A = (TH + 2) * (TW + 2), NPIX_G = TH * TW;
constexpr int NT = 512;       // threads per workgroup: 8 waves = 4 dW quadrants x 2 halves of the tile's k-steps

// LDS pixel stride for C channels such that four consecutive pixel rows x 64 B (one transposing read of a 32-lane
// half) never share a bank: stride == 64 (mod 256) bytes
__host__ __device__ constexpr int pstride(int C) { return C == 32 ? 64 : (C == 64 ? 192 : 320); }

struct WgradArgs {
    ConvSrc src;              // the input source this launch differentiates (one launch per concat source)
    int src_coff;             // first input channel of this source inside the layer's weight tensor
    const unsigned short *g;  // bf16 NHWC gradient of the raw output [N][Ho][Wo][Cout]
    float *slab;              // f32 [ksplit][npar][ci_blocks][co_blocks][TAPS][CI][CO]
    int N, H, W;              // logical input size
    int Cout;
    int taps, npar, ostride;
    int ksplit;
    int debug;                // ablation bits for tools/bench_wgrad.py (env CDNET_WGRAD_DEBUG): 1 no MFMA loop, 2 no loads, 4 no LDS staging
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // register staging type (HIP's uint4 struct copies defeat SROA)
union W16 {
    u32x4 u;
    unsigned short h[8];
};

// ---- register prefetch of a pixel tile.  Everything that does not depend on the tile (which halo pixel / channel slot a
// ---- thread stages, its LDS address) is computed once per kernel and kept packed in one register per staged vector;
// ---- the loads are issued with clamped addresses and no branches, one tile ahead of their use.
template <int CI, bool RES, int NTH = NT>
struct APref {
    static constexpr int VPP = CI / 8;
    static constexpr int NA = (NPIX_A * VPP + NTH - 1) / NTH;
    u32x4 a[NA];
    u32x4 r[RES ? NA : 1];
    int inv[NA];              // (hy << 16) | hx of the halo pixel, -1 = this thread has no i-th vector
    unsigned valid;
};
template <int CO, int NTH = NT>
struct GPref {
    static constexpr int VPP = CO / 8;
    static constexpr int NG = NPIX_G * VPP / NTH;
    u32x4 g[NG];
    unsigned valid;
};

template <int CI, bool RES, int NTH>
__device__ __forceinline__ void init_a(APref<CI, RES, NTH> &P, int tid) {
    constexpr int VPP = CI / 8, NA = APref<CI, RES, NTH>::NA;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int v = tid + i * NTH;
        const int pix = v / VPP;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        P.inv[i] = v < NPIX_A * VPP ? ((hy << 16) | hx) : -1;
    }
    P.valid = 0;
}

template <bool NOPOOL = false, int CI, bool RES, int NTH>
__device__ __forceinline__ void issue_a(APref<CI, RES, NTH> &P, const ConvSrc &s, int cc0, int n, int y0, int x0, int H, int W, int tid) {
    constexpr int VPP = CI / 8, NA = APref<CI, RES, NTH>::NA;
    P.valid = 0;
    if (!NOPOOL && !RES && s.pool) return;       // pooled sources are gathered in commit_a (4 loads per element)
    const int cbase = cc0 + (tid % VPP) * 8;
    const bool cok = cbase < s.C;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    const unsigned short *base = s.x + (size_t)n * s.Hs * rs + cbase;
    const unsigned short *rbase = RES ? s.res + (size_t)n * s.Hs * rs + cbase : nullptr;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int hy = P.inv[i] >> 16, hx = P.inv[i] & 0xffff;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const int ys = y - s.off_y, xs = x - s.off_x;
        const bool ok = P.inv[i] >= 0 && cok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (unsigned)ys < (unsigned)s.Hs &&
                        (unsigned)xs < (unsigned)s.Ws;
        const size_t e = ok ? (size_t)ys * rs + (size_t)(xs * s.C) : 0;
        P.a[i] = *reinterpret_cast<const u32x4 *>((ok ? base : s.x) + e);
        if (RES) P.r[i] = *reinterpret_cast<const u32x4 *>((ok ? rbase : s.res) + e);
        P.valid |= (ok ? 1u : 0u) << i;
    }
}

__device__ __forceinline__ u32x4 xform_a(u32x4 raw, u32x4 res, bool has_res, const float *sc, const float *sh, bool on, bool relu, bool f16) {
    if (f16 && on && relu)                       // the training-mode combination: packed math (xform.h)
        return has_res ? xf_bnrelu_f16<true>(raw, res, sc, sh) : xf_bnrelu_f16<false>(raw, raw, sc, sh);
    W16 r, q, o;
    r.u = raw; q.u = res;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = f16 ? h2f(r.h[j]) : bf2f(r.h[j]);
        if (on) t = fmaf(t, sc[j], sh[j]);
        if (has_res) t += f16 ? h2f(q.h[j]) : bf2f(q.h[j]);
        if (relu) t = fmaxf(t, 0.f);
        o.h[j] = f2bf(t);
    }
    return o.u;
}

__device__ __forceinline__ u32x4 max_bf8(u32x4 a, u32x4 b, bool nonneg) {
    if (nonneg) return xf_max_nonneg_bf8(a, b);
    W16 x, y, o;
    x.u = a; y.u = b;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h[j] = bf2f(y.h[j]) > bf2f(x.h[j]) ? y.h[j] : x.h[j];
    return o.u;
}

// XF selects the input transform at compile time (run-time flag tests inside the unrolled staging loops multiplied the
// code far past the instruction cache): XF_PLAIN bf16 as stored, XF_FAST fp16 raw * scale + shift [+ residual] -> ReLU
// (training mode, packed math), XF_GEN anything, decided per element group from the source's flags.
enum { XF_PLAIN = 0, XF_FAST = 1, XF_GEN = 2 };

template <int XF, int CI, bool RES, int NTH>
__device__ __forceinline__ void commit_a(const APref<CI, RES, NTH> &P, const ConvSrc &s, int cc0, int n, int y0, int x0, int H, int W,
                                         unsigned char *lds, int tid, const float *xf) {
    constexpr int VPP = CI / 8, PSTR = pstride(CI), NA = APref<CI, RES, NTH>::NA;
    const int slot = tid % VPP;
    const int cbase = cc0 + slot * 8;
    const bool cok = cbase < s.C;
    float sc[8], sh[8];
    const bool on = XF == XF_FAST || (XF == XF_GEN && s.scale != nullptr);
    if (XF != XF_PLAIN && on) {                  // LDS copy of this block's scale | shift (see fill_xf): no vmcnt traffic here
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = xf[slot * 8 + j]; sh[j] = xf[CI + slot * 8 + j]; }
    }
    const bool relu = s.relu != 0, f16 = s.f16 != 0;
    const bool plain = XF == XF_PLAIN || (XF == XF_GEN && !on && !relu && !f16 && !RES);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    unsigned char *dst = lds + slot * 16;
    if (XF != XF_GEN || RES || !s.pool) {        // (a source with a residual branch is never pooled: checked by the ABI entry)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (P.inv[i] < 0) continue;
            const int hy = P.inv[i] >> 16, hx = P.inv[i] & 0xffff;
            u32x4 val = zero;
            if (P.valid & (1u << i)) {
                if (XF == XF_PLAIN) val = P.a[i];
                else if (XF == XF_FAST) val = xf_bnrelu_f16<RES>(P.a[i], RES ? P.r[i] : P.a[i], sc, sh);
                else val = plain ? P.a[i] : xform_a(P.a[i], RES ? P.r[i] : zero, RES, sc, sh, on, relu, f16);
            }
            *reinterpret_cast<u32x4 *>(dst + (hy * HALO_W + hx) * PSTR) = val;
        }
        return;
    }
    // pooled source (encoder stage inputs): window (2ys..2ys+1, 2xs..2xs+1) of the stored tensor, transform then max;
    // ceil-mode windows that stick out repeat their first pixel
    const int Hl = (s.Hs + (s.pool == 2)) / 2, Wl = (s.Ws + (s.pool == 2)) / 2;
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    const unsigned short *base = s.x + (size_t)n * s.Hs * rs + cbase;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if (P.inv[i] < 0) continue;
        const int hy = P.inv[i] >> 16, hx = P.inv[i] & 0xffff;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const int ys = y - s.off_y, xs = x - s.off_x;
        const bool ok = cok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (unsigned)ys < (unsigned)Hl && (unsigned)xs < (unsigned)Wl;
        u32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int yy = 2 * ys + (q >> 1), xx = 2 * xs + (q & 1);
            if (yy >= s.Hs) yy = 2 * ys;
            if (xx >= s.Ws) xx = 2 * xs;
            const size_t e = ok ? (size_t)yy * rs + (size_t)(xx * s.C) : 0;
            w[q] = *reinterpret_cast<const u32x4 *>((ok ? base : s.x) + e);
        }
        u32x4 val = zero;
        if (ok) {
            val = plain ? w[0] : xform_a(w[0], zero, false, sc, sh, on, relu, f16);
#pragma unroll
            for (int q = 1; q < 4; ++q) val = max_bf8(val, plain ? w[q] : xform_a(w[q], zero, false, sc, sh, on, relu, f16), relu);
        }
        *reinterpret_cast<u32x4 *>(dst + (hy * HALO_W + hx) * PSTR) = val;
    }
}

template <int CO, int NTH>
__device__ __forceinline__ void issue_g(GPref<CO, NTH> &P, const unsigned short *g, int co0, int Cout, int n, int y0, int x0, int H, int W,
                                        int ostride, int pa, int pb, int tid) {
    constexpr int VPP = CO / 8, NG = GPref<CO, NTH>::NG, PPI = NTH / VPP;      // PPI pixels per i step
    const int Ho = H * ostride, Wo = W * ostride;
    const int co = co0 + (tid % VPP) * 8;
    const bool cok = co < Cout;                                  // Cout % 8 == 0 (checked by the ABI entry)
    const int pix0 = tid / VPP;
    const unsigned short *base = g + (size_t)n * Ho * Wo * Cout + co;
    P.valid = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int pix = pix0 + i * PPI;
        const int y = y0 + pix / TW, x = x0 + pix % TW;
        const bool ok = cok && y < H && x < W;
        const size_t e = ok ? ((size_t)(y * ostride + pa) * Wo + (x * ostride + pb)) * Cout : 0;
        P.g[i] = *reinterpret_cast<const u32x4 *>((ok ? base : g) + e);
        P.valid |= (ok ? 1u : 0u) << i;
    }
}

template <int CO, int NTH>
__device__ __forceinline__ void commit_g(const GPref<CO, NTH> &P, unsigned char *lds, int tid) {
    constexpr int VPP = CO / 8, PSTR = pstride(CO), NG = GPref<CO, NTH>::NG, PPI = NTH / VPP;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    unsigned char *dst = lds + (tid / VPP) * PSTR + (tid % VPP) * 16;
#pragma unroll
    for (int i = 0; i < NG; ++i) *reinterpret_cast<u32x4 *>(dst + i * PPI * PSTR) = (P.valid & (1u << i)) ? P.g[i] : zero;
}

// scale | shift of the CI input channels of this block -> LDS (zero-extended past the source's last channel)
template <int CI>
__device__ __forceinline__ void fill_xf(float *xf, const ConvSrc &s, int cc0, int tid) {
    if (tid < CI) {
        const bool ok = s.scale != nullptr && cc0 + tid < s.C;
        xf[tid] = ok ? s.scale[cc0 + tid] : 1.f;
        xf[CI + tid] = ok ? s.shift[cc0 + tid] : 0.f;
    }
    __syncthreads();
}

template <int CI_T, int CO_T, int TAPS, bool RES>
__global__ __launch_bounds__(NT) void wgrad_kernel(WgradArgs A) {
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int PA = pstride(CI), PG = pstride(CO);
    constexpr int A_BYTES = NPIX_A * PA, G_BYTES = NPIX_G * PG, STAGE = A_BYTES + G_BYTES;
    static_assert(CI_T * CO_T == 4, "one 32x32 quadrant per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // two staging buffers [A | G]
    typedef s16x4 __attribute__((address_space(3))) * lptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = wave & 3, khalf = wave >> 2;
    const int wci = quad / CO_T, wco = quad % CO_T;
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;

    const int co_blocks = (A.Cout + CO - 1) / CO;
    const int ci_blocks = (A.src.C + CI - 1) / CI;
    const int cb = blockIdx.x % co_blocks, ib = blockIdx.x / co_blocks;
    const int par = blockIdx.y;
    const int ks = blockIdx.z;
    const int pa = par >> 1, pb = par & 1;

    auto tap_off = [&](int t) -> int {
        if (TAPS == 9) return ((t / 3) * HALO_W + t % 3) * PA;
        if (TAPS == 4) {
            const int ty = t >> 1, tx = t & 1;
            const int r = pa == 0 ? (ty == 0 ? 1 : 0) : (ty == 0 ? 2 : 1);
            const int c = pb == 0 ? (tx == 0 ? 1 : 0) : (tx == 0 ? 2 : 1);
            return (r * HALO_W + c) * PA;
        }
        return (HALO_W + 1) * PA;
    };
    // per-lane byte offsets of the transposing reads: pixel row (8*(grp>>1) + q) of the k-step, channel 16*(grp&1)+4p;
    // this wave's k-steps are tile rows khalf*4 .. khalf*4+3
    const int a_lane = (8 * (grp >> 1) + q) * PA + (wci * 32 + 16 * (grp & 1) + 4 * p) * 2 + khalf * (TH / 2) * HALO_W * PA;
    const int g_lane = A_BYTES + (8 * (grp >> 1) + q) * PG + (wco * 32 + 16 * (grp & 1) + 4 * p) * 2 + khalf * (TH / 2) * TW * PG;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH - 1) / TH;
    const int tiles_img = tiles_y * tiles_x;
    const int ntiles = A.N * tiles_img;
    auto coords = [&](int tile, int &n, int &y0, int &x0) {
        n = tile / tiles_img;
        const int rem = tile - n * tiles_img;
        const int ty = rem / tiles_x;
        y0 = ty * TH;
        x0 = (rem - ty * tiles_x) * TW;
    };
    __shared__ __attribute__((aligned(16))) float s_xf[2 * CI];
    fill_xf<CI>(s_xf, A.src, ib * CI, tid);
    APref<CI, RES> PA_;
    GPref<CO> PG_;
    init_a(PA_, tid);
    // prologue: tile ks -> buffer 0, loads of tile ks+ksplit in flight
    int n, y0, x0;
    if (ks < ntiles) {
        coords(ks, n, y0, x0);
        issue_a(PA_, A.src, ib * CI, n, y0, x0, A.H, A.W, tid);
        issue_g(PG_, A.g, cb * CO, A.Cout, n, y0, x0, A.H, A.W, A.ostride, pa, pb, tid);
        commit_a<XF_GEN>(PA_, A.src, ib * CI, n, y0, x0, A.H, A.W, smem, tid, s_xf);
        commit_g(PG_, smem + A_BYTES, tid);
        if (ks + A.ksplit < ntiles) {
            coords(ks + A.ksplit, n, y0, x0);
            issue_a(PA_, A.src, ib * CI, n, y0, x0, A.H, A.W, tid);
            issue_g(PG_, A.g, cb * CO, A.Cout, n, y0, x0, A.H, A.W, A.ostride, pa, pb, tid);
        }
    }
    __syncthreads();
    int cur = 0;
    for (int tile = ks; tile < ntiles; tile += A.ksplit) {
        // stage the next tile into the other buffer (its previous reader finished before the last barrier), then
        // start the loads of the tile after it; (n, y0, x0) still hold the next tile's coordinates
        if (tile + A.ksplit < ntiles) {
            unsigned char *nb = smem + (cur ^ 1) * STAGE;
            if (!(A.debug & 4)) {
                commit_a<XF_GEN>(PA_, A.src, ib * CI, n, y0, x0, A.H, A.W, nb, tid, s_xf);
                commit_g(PG_, nb + A_BYTES, tid);
            }
            if (tile + 2 * A.ksplit < ntiles && !(A.debug & 2)) {
                coords(tile + 2 * A.ksplit, n, y0, x0);
                issue_a(PA_, A.src, ib * CI, n, y0, x0, A.H, A.W, tid);
                issue_g(PG_, A.g, cb * CO, A.Cout, n, y0, x0, A.H, A.W, A.ostride, pa, pb, tid);
            }
        }
        const unsigned char *buf = smem + cur * STAGE;
        if (!(A.debug & 1))
#pragma unroll
        for (int ky = 0; ky < TH / 2; ++ky) {             // one k-step = one tile row of 16 pixels
            const int gaddr = g_lane + ky * TW * PG;
            s16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + gaddr));
            s16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + gaddr + 4 * PG));
            s16x8 gv;
            gv[0] = g0[0]; gv[1] = g0[1]; gv[2] = g0[2]; gv[3] = g0[3];
            gv[4] = g1[0]; gv[5] = g1[1]; gv[6] = g1[2]; gv[7] = g1[3];
            const bf16x8 gf = __builtin_bit_cast(bf16x8, gv);
            const int abase = a_lane + ky * HALO_W * PA;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + abase + tap_off(t)));
                s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + abase + tap_off(t) + 4 * PA));
                s16x8 av;
                av[0] = a0[0]; av[1] = a0[1]; av[2] = a0[2]; av[3] = a0[3];
                av[4] = a1[0]; av[5] = a1[1]; av[6] = a1[2]; av[7] = a1[3];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), gf, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // fold the two k-halves through LDS (the staging tiles are dead now): waves 4..7 park their accumulators,
    // waves 0..3 add them in a fixed order (deterministic) and write the slab
    float *s_acc = reinterpret_cast<float *>(smem);
    if (khalf == 1) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[((quad * TAPS + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (khalf == 1) return;
    // slab [ks][par][ib][cb][tap][CI][CO]; D rows = ci (regs + lane half), cols = co (lane & 31)
    float *slab = A.slab + ((((size_t)ks * A.npar + par) * ci_blocks + ib) * co_blocks + cb) * (size_t)(TAPS * CI * CO);
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[((size_t)t * CI + ci) * CO + wco * 32 + l31] = acc[t][r] + s_acc[((quad * TAPS + t) * 16 + r) * 64 + lane];
        }
}

// ------------------------------------------------------------------------------------------------------
// Specialised-wave variant for un-pooled sources.  Waves 0..3 only run the matrix cores (one dW quadrant each, all eight
// k-steps of a tile), waves 4..7 only move data: they keep the global loads of two tiles in flight in registers,
// transform them and fill the other LDS buffer while the consumers work.  One barrier per tile; each SIMD hosts one
// consumer and one producer wave, so staging VALU work and MFMAs overlap instead of alternating.
// ------------------------------------------------------------------------------------------------------
constexpr int NTP = 256;      // producer threads

#ifdef CDNET_WS_STAMPS
// debug build only (CDNET_HIPCC_FLAGS=-DCDNET_WS_STAMPS): wall-clock stamps (100 MHz) of one consumer and one producer wave
__device__ unsigned long long g_wg_stamps[2 * 1024];
extern "C" __attribute__((visibility("default"))) int cdnet_debug_wgrad_stamps(unsigned long long *dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_stamps), sizeof(g_wg_stamps)) == hipSuccess ? 0 : 1;
}
#define WG_STAMP(id) do { if (stamp_on && sn < 1000) { g_wg_stamps[sbase + sn++] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(id); } } while (0)
#else
#define WG_STAMP(id) do { } while (0)
#endif

template <int CI_T, int CO_T, int TAPS, bool RES, int XF>
__global__ __launch_bounds__(NT) void wgrad_ws_kernel(WgradArgs A) {
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int PA = pstride(CI), PG = pstride(CO);
    constexpr int A_BYTES = NPIX_A * PA, G_BYTES = NPIX_G * PG, STAGE = A_BYTES + G_BYTES;
    static_assert(CI_T * CO_T == 4 || CI_T * CO_T == 1, "one 32x32 quadrant per consumer wave, or one 32x32 block for all four");
    // KQ (<1, 1>: layers with at most 32 output channels - HRNet's 18-channel branch at 512 x 512, the decoder's 16-channel block): ONE
    // 32 x 32 block of dW, and the four consumer waves split the tile's eight rows (the k dimension) instead of the block - on 64 x 64 blocks
    // three of the four quadrants multiplied zero padding and the consumers' 72 MFMAs per tile, not the 16 KB the tile brings, set the tile
    // period (1.0-1.3 TB/s).  The four partial blocks are folded through the LDS in a fixed order at the end.  The slab keeps the <1, 4>
    // layout the caller sized it for (128 columns per row, the first 32 written; the reduce skips columns beyond Cout).
    constexpr bool KQ = CI_T * CO_T == 1;
    constexpr int CO_SLAB = KQ ? 128 : CO;
    static_assert(!KQ || 2 * STAGE >= TAPS * 16 * 64 * 4, "the fold of the four partial blocks goes through the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // two staging buffers [A | G]
    typedef s16x4 __attribute__((address_space(3))) * lptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_blocks = (A.Cout + CO - 1) / CO;
    const int ci_blocks = (A.src.C + CI - 1) / CI;
    const int cb = blockIdx.x % co_blocks, ib = blockIdx.x / co_blocks;
    const int par = blockIdx.y;
    const int ks = blockIdx.z;
    const int pa = par >> 1, pb = par & 1;
    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH - 1) / TH;
    const int tiles_img = tiles_y * tiles_x;
    const int ntiles = A.N * tiles_img;
    const int ntl = ks < ntiles ? (ntiles - ks + A.ksplit - 1) / A.ksplit : 0;      // tiles of this workgroup
    const int ntl2 = (ntl + 1) & ~1;                                                // barrier steps of both roles (padded to even)

    __shared__ __attribute__((aligned(16))) float s_xf[2 * CI];
    fill_xf<CI>(s_xf, A.src, ib * CI, tid);
#ifdef CDNET_WS_STAMPS
    const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 17 && lane == 0 && (wave == 0 || wave == 4);
    const int sbase = wave >= 4 ? 1024 : 0;
    int sn = 0;
#endif
    if (wave >= 4) {
        // ------------------------------- producers -------------------------------
        const int ptid = tid - NTP;
        APref<CI, RES, NTP> P0, P1;
        GPref<CO, NTP> G0, G1;
        init_a(P0, ptid);
        init_a(P1, ptid);
        // Straight-line loop body: the tile index is clamped instead of tested (the tail re-requests and re-stages the last tile into
        // the buffer nobody reads), the source is never pooled here - with branches around the requests the compiler cannot count the
        // loads in flight and drains them (s_waitcnt vmcnt(0) between the input and the gradient requests: one exposed memory round
        // trip, 2.3 us, per tile - measured with the stamp build)
        // Request addresses without per-vector multiplications: what depends on the thread only (its halo pixels' and gradient pixels'
        // offsets inside a tile) is computed once, a tile contributes one scalar base (the stamp build showed the request phase at
        // 2.0 us per tile - 45 quarter-rate integer multiplies and 64-bit mads per thread beside the consumers' MFMAs - against 2.0 us
        // of matrix work: the producers, not the matrix pipe, set the tile period)
        const ConvSrc &sA = A.src;
        const int rs_ = sA.row_stride ? sA.row_stride : sA.Ws * sA.C;
        constexpr int VA_ = CI / 8, VG_ = CO / 8, NA_ = APref<CI, RES, NTP>::NA, NG_ = GPref<CO, NTP>::NG, PPI_ = NTP / VG_;
        const int cbase_a = ib * CI + (ptid % VA_) * 8;
        const bool cok_a = cbase_a < sA.C;
        const int co_g = cb * CO + (ptid % VG_) * 8;
        const bool cok_g = co_g < A.Cout;
        const int Ho_ = A.H * A.ostride, Wo_ = A.W * A.ostride;
        int aoff[NA_], goff[NG_], gyx[NG_];
#pragma unroll
        for (int i = 0; i < NA_; ++i) {
            const int hy = P0.inv[i] >> 16, hx = P0.inv[i] & 0xffff;
            aoff[i] = P0.inv[i] >= 0 ? hy * rs_ + hx * sA.C : 0;
        }
#pragma unroll
        for (int i = 0; i < NG_; ++i) {
            const int pix = ptid / VG_ + i * PPI_;
            const int py = pix / TW, px = pix % TW;
            gyx[i] = (py << 16) | px;
            goff[i] = (py * A.ostride * Wo_ + px * A.ostride) * A.Cout;
        }
        auto issue = [&](APref<CI, RES, NTP> &P, GPref<CO, NTP> &G, int j) {
            j = j < ntl ? j : ntl - 1;
            const int tile = ks + j * A.ksplit;
            const int n = tile / tiles_img, rem = tile - n * tiles_img;
            const int ty = rem / tiles_x;
            const int y0 = ty * TH, x0 = (rem - ty * tiles_x) * TW;
            // (scalar) element offset of halo pixel (0, 0), channel cbase_a; may point before the image at the border - used only when ok
            const long long abase = ((long long)n * sA.Hs + (y0 - 1 - sA.off_y)) * rs_ + (long long)(x0 - 1 - sA.off_x) * sA.C + cbase_a;
            const unsigned short *pa0 = sA.x + abase;
            const unsigned short *pr0 = RES ? sA.res + abase : nullptr;
            unsigned valid = 0;
#pragma unroll
            for (int i = 0; i < NA_; ++i) {
                const int hy = P.inv[i] >> 16, hx = P.inv[i] & 0xffff;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool ok = P.inv[i] >= 0 && cok_a && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W &&
                                (unsigned)(y - sA.off_y) < (unsigned)sA.Hs && (unsigned)(x - sA.off_x) < (unsigned)sA.Ws;
                P.a[i] = *reinterpret_cast<const u32x4 *>(ok ? pa0 + aoff[i] : sA.x);
                if (RES) P.r[i] = *reinterpret_cast<const u32x4 *>(ok ? pr0 + aoff[i] : sA.res);
                valid |= (ok ? 1u : 0u) << i;
            }
            P.valid = valid;
            const unsigned short *pg0 = A.g + (((long long)n * Ho_ + (y0 * A.ostride + pa)) * Wo_ + (x0 * A.ostride + pb)) * A.Cout + co_g;
            unsigned gvalid = 0;
#pragma unroll
            for (int i = 0; i < NG_; ++i) {
                const bool ok = cok_g && y0 + (gyx[i] >> 16) < A.H && x0 + (gyx[i] & 0xffff) < A.W;
                G.g[i] = *reinterpret_cast<const u32x4 *>(ok ? pg0 + goff[i] : A.g);
                gvalid |= (ok ? 1u : 0u) << i;
            }
            G.valid = gvalid;
        };
        auto commit = [&](const APref<CI, RES, NTP> &P, const GPref<CO, NTP> &G, int bufi) {
            unsigned char *nb = smem + bufi * STAGE;
            commit_a<XF>(P, A.src, ib * CI, 0, 0, 0, A.H, A.W, nb, ptid, s_xf);
            commit_g(G, nb + A_BYTES, ptid);
        };
        if (ntl == 0) {
            __syncthreads();
            if (KQ) for (int i = 0; i < 6; ++i) __syncthreads();      // (the consumers' fold)
            return;
        }
        issue(P0, G0, 0);
        issue(P1, G1, 1);
        commit(P0, G0, 0);
        issue(P0, G0, 2);
        __syncthreads();
        for (int j = 0; j < ntl2; j += 2) {
            // consumers are on tile j (buffer 0): fill buffer 1 with tile j+1, then start the loads of tile j+3
            WG_STAMP(1);
            commit(P1, G1, 1);
            WG_STAMP(2);
            issue(P1, G1, j + 3);
            WG_STAMP(3);
            __syncthreads();
            // consumers are on tile j+1 (buffer 1): fill buffer 0 with tile j+2, start the loads of tile j+4
            WG_STAMP(1);
            commit(P0, G0, 0);
            WG_STAMP(2);
            issue(P0, G0, j + 4);
            WG_STAMP(3);
            __syncthreads();
        }
#ifdef CDNET_WS_STAMPS
        if (stamp_on) g_wg_stamps[sbase + sn] = 0;
#endif
        if (KQ) for (int i = 0; i < 6; ++i) __syncthreads();          // (the consumers' fold: every wave meets every barrier)
        return;
    }
    // ------------------------------- consumers -------------------------------
    const int quad = wave;
    const int wci = KQ ? 0 : quad / CO_T, wco = KQ ? 0 : quad % CO_T;
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    auto tap_off = [&](int t) -> int {
        if (TAPS == 9) return ((t / 3) * HALO_W + t % 3) * PA;
        if (TAPS == 4) {
            const int ty = t >> 1, tx = t & 1;
            const int r = pa == 0 ? (ty == 0 ? 1 : 0) : (ty == 0 ? 2 : 1);
            const int c = pb == 0 ? (tx == 0 ? 1 : 0) : (tx == 0 ? 2 : 1);
            return (r * HALO_W + c) * PA;
        }
        return (HALO_W + 1) * PA;
    };
    const int a_lane = (8 * (grp >> 1) + q) * PA + (wci * 32 + 16 * (grp & 1) + 4 * p) * 2;
    const int g_lane = A_BYTES + (8 * (grp >> 1) + q) * PG + (wco * 32 + 16 * (grp & 1) + 4 * p) * 2;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();
    for (int j = 0; j < ntl2; ++j) {
        if (j >= ntl) { __syncthreads(); continue; }          // padding step
        const unsigned char *buf = smem + (j & 1) * STAGE;
        WG_STAMP(10);
        auto frag = [&](int off, int pstr) -> bf16x8 {
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + off));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + off + 4 * pstr));
            s16x8 av;
            av[0] = a0[0]; av[1] = a0[1]; av[2] = a0[2]; av[3] = a0[3];
            av[4] = a1[0]; av[5] = a1[1]; av[6] = a1[2]; av[7] = a1[3];
            return __builtin_bit_cast(bf16x8, av);
        };
        if (A.debug & 1) {
        } else if (TAPS == 9 && KQ) {
            // this wave's two tile rows r0, r0 + 1: halo rows r0 .. r0 + 3 (12 fragments), two gradient fragments, 18 MFMAs
            const int r0 = 2 * quad;
            bf16x8 row[4][3], gq[2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) row[r][kx] = frag(a_lane + ((r0 + r) * HALO_W + kx) * PA, PA);
            gq[0] = frag(g_lane + r0 * TW * PG, PG);
            gq[1] = frag(g_lane + (r0 + 1) * TW * PG, PG);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row[k + t / 3][t % 3], gq[k], acc[t], 0, 0, 0);
        } else if (TAPS == 9) {
            // 3x3: the k-step of tile row ky multiplies halo rows ky, ky+1, ky+2 (three column shifts each) - consecutive k-steps share
            // two of the three rows.  A ring of four halo rows of fragments: every k-step reads ONE new halo row (3 fragments) and one
            // gradient fragment for its 9 MFMAs (0.9 KB of LDS per MFMA instead of 2), one k-step ahead of their use.
            bf16x8 row[4][3], gq[2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) row[r][kx] = frag(a_lane + (r * HALO_W + kx) * PA, PA);
            gq[0] = frag(g_lane, PG);
#pragma unroll
            for (int ky = 0; ky < TH; ++ky) {
                if (ky + 1 < TH) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) row[(ky + 3) & 3][kx] = frag(a_lane + ((ky + 3) * HALO_W + kx) * PA, PA);
                    gq[(ky + 1) & 1] = frag(g_lane + (ky + 1) * TW * PG, PG);
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this k-step's MFMAs (the scheduler sinks them to their use)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row[(ky + t / 3) & 3][t % 3], gq[ky & 1], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 2
            for (int k = 0; k < (KQ ? 2 : TH); ++k) {     // one k-step = one tile row of 16 pixels
                const int ky = (KQ ? 2 * quad : 0) + k;
                const bf16x8 gf = frag(g_lane + ky * TW * PG, PG);
                const int abase = a_lane + ky * HALO_W * PA;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(abase + tap_off(t), PA), gf, acc[t], 0, 0, 0);
            }
        }
        WG_STAMP(11);
        __syncthreads();
        WG_STAMP(12);
    }
    if (KQ) {
        // ((wave 0 + wave 1) + wave 2) + wave 3 through the (dead) staging buffers: deterministic
        float *s_acc = reinterpret_cast<float *>(smem);
#pragma unroll 1
        for (int w = 1; w < 4; ++w) {
            if (quad == w) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_acc[(t * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (quad == 0) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += s_acc[(t * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (quad != 0) return;
    }
    // slab [ks][par][ib][cb][tap][CI][CO]; D rows = ci (regs + lane half), cols = co (lane & 31)
    const int co_blocks_slab = (A.Cout + CO_SLAB - 1) / CO_SLAB;
    float *slab = A.slab + ((((size_t)ks * A.npar + par) * ci_blocks + ib) * co_blocks_slab + (KQ ? 0 : cb)) * (size_t)(TAPS * CI * CO_SLAB);
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[((size_t)t * CI + ci) * CO_SLAB + wco * 32 + l31] = acc[t][r];
        }
    WG_STAMP(13);
#ifdef CDNET_WS_STAMPS
    if (stamp_on) g_wg_stamps[sbase + sn] = 0;
#endif
}

// ------------------------------------------------------------------------------------------------------
// fp32-precision variant (src.f16 == 2: input, residual and output gradient are fp32 tensors).  Both MFMA operands are
// split into (hi, lo) bf16 planes while they are staged - [pixel][channel] LDS tiles A_hi | A_lo | G_hi | G_lo, read with
// the same transposing loads - and every k-step runs three MFMAs per tap: a_lo*g_hi + a_hi*g_lo + a_hi*g_hi, fp32
// accumulation (see conv32.hip for the error bound).  8 waves = 4 dW quadrants x 2 halves of the tile's k-steps; one LDS
// stage, the next tile's global loads in flight in registers during the MFMA phase.  Pooled sources are materialised first.
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 wg_bf16x2;
typedef __attribute__((ext_vector_type(2))) float wg_f32x2;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;

// hi = bf16(x) (RNE), lo = bf16(x - hi) for eight values.  The two subtractions of a pair are kept scalar (inline assembly): left to
// the compiler they become one v_pk_add_f32, and a packed fp32 instruction costs a mover wave far more than two plain ones beside
// the consumers' MFMA stream (wgrad_ws32_kernel: 257 -> 241 us on the 64 -> 64 layer at 256 x 256 x 16).
__device__ __forceinline__ void wg_split8(const float *v, u32x4 &hi, u32x4 &lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const wg_f32x2 x = {v[2 * k], v[2 * k + 1]};
        const wg_bf16x2 h = __builtin_convertvector(x, wg_bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float d0, d1;
        asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x[0]), "v"(__builtin_bit_cast(float, hb << 16)));
        asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x[1]), "v"(__builtin_bit_cast(float, hb & 0xffff0000u)));
        const wg_f32x2 d = {d0, d1};
        hi[k] = hb;
        lo[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, wg_bf16x2));
    }
}

constexpr int NT32 = 256;      // 4 waves, one per SIMD (512 registers each): one dW quadrant x all k-steps of a tile per wave

// QM (64 x 64 blocks, the transposed convolutions of the decoder's 32- and 16-channel blocks): as wgrad_ws32_kernel's - 1: at most 32
// channels on both sides, one quadrant, every wave two of the tile's eight rows; 2: at most 32 output channels, the two input-channel
// quadrants, a wave pair each, four rows per wave.  The zero half of the staged vectors is neither requested nor split.
template <int CI_T, int CO_T, int TAPS, int QM = 0>
__global__ __launch_bounds__(NT32) void wgrad_f32_kernel(WgradArgs A) {
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int PA = pstride(CI), PG = pstride(CO);
    constexpr int A_BYTES = NPIX_A * PA, G_BYTES = NPIX_G * PG;
    static_assert(QM == 0 || (CI_T == 2 && CO_T == 2 && TAPS != 9 && (QM == 1 || QM == 2)), "quadrant modes: 64 x 64 blocks, the plain k loop");
    constexpr int VA = QM == 1 ? 4 : CI / 8, VG = QM != 0 ? 4 : CO / 8;
    constexpr int NA = (NPIX_A * VA + NT32 - 1) / NT32, NG = NPIX_G * VG / NT32;
    static_assert(CI_T * CO_T == 4 && NPIX_G * VG % NT32 == 0, "one 32x32 quadrant per wave");
    constexpr int KROWS = QM == 0 ? TH : (QM == 1 ? TH / 4 : TH / 2);        // tile rows (k-steps) per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [A_hi][A_lo][G_hi][G_lo]
    typedef s16x4 __attribute__((address_space(3))) * lptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = wave;
    const int wci = QM == 0 ? quad / CO_T : (QM == 2 ? (quad & 1) : 0), wco = QM == 0 ? quad % CO_T : 0;
    const int row0 = QM == 0 ? 0 : (QM == 1 ? quad : (quad >> 1)) * KROWS;    // this wave's first tile row
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int co_blocks = (A.Cout + CO - 1) / CO;
    const int ci_blocks = (A.src.C + CI - 1) / CI;
    const int cb = blockIdx.x % co_blocks, ib = blockIdx.x / co_blocks;
    const int par = blockIdx.y;
    const int ks = blockIdx.z;
    const int pa = par >> 1, pb = par & 1;
    auto tap_off = [&](int t) -> int {
        if (TAPS == 9) return ((t / 3) * HALO_W + t % 3) * PA;
        if (TAPS == 4) {
            const int ty = t >> 1, tx = t & 1;
            const int r = pa == 0 ? (ty == 0 ? 1 : 0) : (ty == 0 ? 2 : 1);
            const int c = pb == 0 ? (tx == 0 ? 1 : 0) : (tx == 0 ? 2 : 1);
            return (r * HALO_W + c) * PA;
        }
        return (HALO_W + 1) * PA;
    };
    const int a_lane = (8 * (grp >> 1) + q) * PA + (wci * 32 + 16 * (grp & 1) + 4 * p) * 2;
    const int g_lane = 2 * A_BYTES + (8 * (grp >> 1) + q) * PG + (wco * 32 + 16 * (grp & 1) + 4 * p) * 2;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH - 1) / TH;
    const int tiles_img = tiles_y * tiles_x;
    const int ntiles = A.N * tiles_img;
    const int ntl = ks < ntiles ? (ntiles - ks + A.ksplit - 1) / A.ksplit : 0;
    __shared__ __attribute__((aligned(16))) float s_xf[2 * CI];
    fill_xf<CI>(s_xf, A.src, ib * CI, tid);

    const ConvSrc &s = A.src;
    const float *sx = reinterpret_cast<const float *>(s.x), *sr = reinterpret_cast<const float *>(s.res);
    const float *gx = reinterpret_cast<const float *>(A.g);
    const size_t rs = s.row_stride ? (size_t)s.row_stride : (size_t)s.Ws * s.C;
    const int slot_a = tid % VA, cbase = ib * CI + slot_a * 8;
    const bool cok_a = cbase < s.C;
    const int slot_g = tid % VG, co = cb * CO + slot_g * 8;
    const bool cok_g = co < A.Cout;
    const int Ho = A.H * A.ostride, Wo = A.W * A.ostride;
    const bool on = s.scale != nullptr, relu = s.relu != 0;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_xf[slot_a * 8 + j]; sh[j] = s_xf[CI + slot_a * 8 + j]; }

    wg_f32x4 pa_[NA][2], pg_[NG][2];
    int ea[NA];                          // element offset of the thread's i-th input vector, -1 = zero fill (tensors < 2^31 elements)
    unsigned gvalid = 0;
    // what depends on the thread only - halo coordinates and in-tile offsets of its vectors - is computed once; a tile contributes one
    // base offset (no integer divisions or 64-bit multiplies per vector and tile between the MFMA phases)
    int hyx[NA], aoff[NA], gyx[NG], goff[NG];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int v = tid + i * NT32, pix = v / VA;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        hyx[i] = v < NPIX_A * VA ? ((hy << 16) | hx) : -1;
        aoff[i] = hy * (int)rs + hx * s.C;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int pix = tid / VG + i * (NT32 / VG);
        const int py = pix / TW, px = pix % TW;
        gyx[i] = (py << 16) | px;
        goff[i] = (py * A.ostride * Wo + px * A.ostride) * A.Cout;
    }
    auto issue = [&](int j) {
        const int tile = ks + j * A.ksplit;
        const int n = tile / tiles_img, rem = tile - n * tiles_img;
        const int ty = rem / tiles_x;
        const int y0 = ty * TH, x0 = (rem - ty * tiles_x) * TW;
        const int abase_e = (n * s.Hs + (y0 - 1 - s.off_y)) * (int)rs + (x0 - 1 - s.off_x) * s.C + cbase;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int hy = hyx[i] >> 16, hx = hyx[i] & 0xffff;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const int ys = y - s.off_y, xs = x - s.off_x;
            const bool ok = hyx[i] >= 0 && cok_a && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W &&
                            (unsigned)ys < (unsigned)s.Hs && (unsigned)xs < (unsigned)s.Ws;
            ea[i] = ok ? abase_e + aoff[i] : -1;
            const wg_f32x4 *pp = reinterpret_cast<const wg_f32x4 *>(sx + (ok ? ea[i] : 0));
            pa_[i][0] = pp[0];
            pa_[i][1] = pp[1];
        }
        gvalid = 0;
        const size_t gbase_e = (((size_t)n * Ho + (y0 * A.ostride + pa)) * Wo + (x0 * A.ostride + pb)) * A.Cout + co;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const bool ok = cok_g && y0 + (gyx[i] >> 16) < A.H && x0 + (gyx[i] & 0xffff) < A.W;
            const size_t e = ok ? gbase_e + goff[i] : 0;
            const wg_f32x4 *pp = reinterpret_cast<const wg_f32x4 *>(gx + e);
            pg_[i][0] = pp[0];
            pg_[i][1] = pp[1];
            gvalid |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit = [&]() {
        wg_f32x4 rr[NA][2];
        if (sr) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const wg_f32x4 *pp = reinterpret_cast<const wg_f32x4 *>(sr + (ea[i] >= 0 ? ea[i] : 0));
                rr[i][0] = pp[0];
                rr[i][1] = pp[1];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * NT32;
            if (v >= NPIX_A * VA) continue;
            u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
            if (ea[i] >= 0) {
                float x[8] = {pa_[i][0][0], pa_[i][0][1], pa_[i][0][2], pa_[i][0][3], pa_[i][1][0], pa_[i][1][1], pa_[i][1][2], pa_[i][1][3]};
                if (on) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], sc[j], sh[j]);
                }
                if (sr) {
                    const float r[8] = {rr[i][0][0], rr[i][0][1], rr[i][0][2], rr[i][0][3], rr[i][1][0], rr[i][1][1], rr[i][1][2], rr[i][1][3]};
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] += r[j];
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                }
                wg_split8(x, hi, lo);
            }
            unsigned char *d = smem + (v / VA) * PA + slot_a * 16;
            *reinterpret_cast<u32x4 *>(d) = hi;
            *reinterpret_cast<u32x4 *>(d + A_BYTES) = lo;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int pix = tid / VG + i * (NT32 / VG);
            u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
            if (gvalid & (1u << i)) {
                const float x[8] = {pg_[i][0][0], pg_[i][0][1], pg_[i][0][2], pg_[i][0][3], pg_[i][1][0], pg_[i][1][1], pg_[i][1][2], pg_[i][1][3]};
                wg_split8(x, hi, lo);
            }
            unsigned char *d = smem + 2 * A_BYTES + pix * PG + slot_g * 16;
            *reinterpret_cast<u32x4 *>(d) = hi;
            *reinterpret_cast<u32x4 *>(d + G_BYTES) = lo;
        }
    };
    auto frag2 = [&](int off, int pstr) -> bf16x8 {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(smem + off));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(smem + off + 4 * pstr));
        s16x8 av;
        av[0] = a0[0]; av[1] = a0[1]; av[2] = a0[2]; av[3] = a0[3];
        av[4] = a1[0]; av[5] = a1[1]; av[6] = a1[2]; av[7] = a1[3];
        return __builtin_bit_cast(bf16x8, av);
    };

    if (ntl > 0) issue(0);
    for (int j = 0; j < ntl; ++j) {
        __syncthreads();                              // the previous tile's fragment reads are done
        commit();
        __syncthreads();
        if (j + 1 < ntl) issue(j + 1);
        if (TAPS == 9) {
            // consecutive k-steps share two of their three halo rows: a ring of four rows of (hi, lo) fragments, every k-step reads one
            // new row and one gradient fragment pair for its 27 MFMAs, one k-step ahead of their use (as wgrad_ws_kernel)
            bf16x8 rh[4][3], rl[4][3], gqh[2], gql[2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    rh[r][kx] = frag2(a_lane + (r * HALO_W + kx) * PA, PA);
                    rl[r][kx] = frag2(a_lane + (r * HALO_W + kx) * PA + A_BYTES, PA);
                }
            gqh[0] = frag2(g_lane, PG);
            gql[0] = frag2(g_lane + G_BYTES, PG);
#pragma unroll
            for (int ky = 0; ky < TH; ++ky) {
                if (ky + 1 < TH) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        rh[(ky + 3) & 3][kx] = frag2(a_lane + ((ky + 3) * HALO_W + kx) * PA, PA);
                        rl[(ky + 3) & 3][kx] = frag2(a_lane + ((ky + 3) * HALO_W + kx) * PA + A_BYTES, PA);
                    }
                    gqh[(ky + 1) & 1] = frag2(g_lane + (ky + 1) * TW * PG, PG);
                    gql[(ky + 1) & 1] = frag2(g_lane + (ky + 1) * TW * PG + G_BYTES, PG);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const bf16x8 ah = rh[(ky + t / 3) & 3][t % 3], al = rl[(ky + t / 3) & 3][t % 3];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gqh[ky & 1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gql[ky & 1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gqh[ky & 1], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 2
            for (int k = 0; k < KROWS; ++k) {
                const int ky = row0 + k;
                const int gaddr = g_lane + ky * TW * PG;
                const bf16x8 gh = frag2(gaddr, PG), gl = frag2(gaddr + G_BYTES, PG);
                const int abase = a_lane + ky * HALO_W * PA;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const bf16x8 ah = frag2(abase + tap_off(t), PA), al = frag2(abase + tap_off(t) + A_BYTES, PA);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc[t], 0, 0, 0);
                }
            }
        }
    }
    if (QM != 0) {
        // the waves that share a quadrant, in a fixed order through the (dead) staging buffer: QM 1 ((w0 + w1) + w2) + w3, QM 2 w0 + w2, w1 + w3
        static_assert(QM == 0 || 2 * A_BYTES + 2 * G_BYTES >= 3 * TAPS * 16 * 64 * 4, "the parked accumulators fit the staging buffer");
        float *s_acc = reinterpret_cast<float *>(smem);
        __syncthreads();                                  // the last tile's fragment reads are done
        if (QM == 1 ? quad != 0 : quad >= 2) {
            float *d = s_acc + (QM == 1 ? quad - 1 : (quad & 1)) * TAPS * 16 * 64;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) d[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (QM == 1 ? quad != 0 : quad >= 2) return;
#pragma unroll 1
        for (int w = 0; w < (QM == 1 ? 3 : 1); ++w) {
            const float *d = s_acc + (QM == 1 ? w : (quad & 1)) * TAPS * 16 * 64;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += d[(t * 16 + r) * 64 + lane];
        }
    }
    float *slab = A.slab + ((((size_t)ks * A.npar + par) * ci_blocks + ib) * co_blocks + cb) * (size_t)(TAPS * CI * CO);
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[((size_t)t * CI + ci) * CO + wco * 32 + l31] = acc[t][r];
        }
}

// ------------------------------------------------------------------------------------------------------
// Wave-specialised form of the fp32-precision weight gradient (3x3, 64 x 64 channel blocks): wgrad_f32_kernel's four waves stage a
// tile (transform, hi / lo split, LDS writes: ~450 vector instructions per thread) and then multiply it - the matrix pipe idles a
// third of the time.  Here waves 0..3 only read fragments and issue MFMAs (one 32 x 32 dW quadrant x 9 taps each), waves 4..7 only
// move data: fp32 input halo + gradient vectors of two tiles in flight in registers (straight-line code, clamped tile index),
// source transform, split into hi | lo bf16 planes, the other LDS stage.  Two stages of A_hi | A_lo | G_hi | G_lo only fit the LDS
// for 4 x 16-pixel tiles (66 KB per stage); a tile is then four k-steps of 27 MFMAs - one barrier per 108 MFMAs per wave, as in
// conv_ws32_kernel.  The consumers' fragment reads (ds_read_b64_tr_b16 pairs) are issued three taps ahead of their MFMAs, at most one
// pair per MFMA gap, and pinned there by a scheduling fence behind every MFMA; no fragment ring - 2 waves per SIMD leave 256
// registers, 144 of them accumulators - so a k-step reads its 9 (hi, lo) input fragments again (40 transposing reads per 27 MFMAs).
// Same products and the same three-MFMA order per tap as wgrad_f32_kernel; the pixels are summed in another order (other tiles),
// so the results agree to fp32 rounding, not bit for bit.
// ------------------------------------------------------------------------------------------------------
constexpr int TH32 = 4, NPIX_A32 = (TH32 + 2) * HALO_W, NPIX_G32 = TH32 * TW;

// XF: source transform: 0 plain fp32, 1 x * scale + shift -> ReLU, 2 run-time flags (scale / shift, residual, ReLU).  TAPS: 9 (3x3), or 1
// (the residual units' 1x1 convolutions: the same pipeline on the tile's inner 4 x 16 pixels - the halo vectors are never requested;
// wgrad_f32_kernel took 250-470 us per launch for this HBM-bound product, one 4-wave workgroup per CU waiting out every load)
// QM (layers with at most 32 input or output channels - the decoder's 32- and 16-channel blocks, the 16-channel input of the first
// residual unit): quadrants of the 64 x 64 block that hold nothing but zero padding are not multiplied; the consumer waves they free split
// the tile's four rows (the k dimension) instead.  0: four quadrants, one per wave (the 64-channel layers); 1: one quadrant (at most 32
// channels on both sides), every wave one tile row; 2: the two input-channel quadrants (Cout <= 32), 3: the two output-channel quadrants
// (C <= 32) - a wave pair per quadrant, two tile rows each.  The partial blocks of the waves that share a quadrant are folded through the
// LDS in a fixed order when the run ends.  (A 16 -> 16 layer at 256 x 256 x 16 took the 64 -> 64 layer's 233 us on quadrants of zeros.)
template <int XF, int TAPS = 9, int QM = 0>
__global__ __launch_bounds__(512) void wgrad_ws32_kernel(WgradArgs A) {
    constexpr int CI = 64, CO = 64, CO_T = 2;
    static_assert(TAPS == 9 || TAPS == 1, "3x3 or 1x1");
    constexpr int NQ = QM == 0 ? 4 : (QM == 1 ? 1 : 2);          // quadrants multiplied
    // (QM 1 - 32 channels on both sides, 8 KB of operands per 4-row tile - takes 8-row tiles: its rate was the barrier period, 94 us for
    //  a 16 -> 16 layer at 256 x 256 x 16 against ~35 us of traffic; the LDS rows shrink to the 32-channel pitch)
    constexpr int TH_ = QM == 1 ? 8 : TH32, NPA = (TH_ + 2) * HALO_W, NPG = TH_ * TW;
    constexpr int KROWS = TH_ * NQ / 4;                          // tile rows (k-steps) per consumer wave
    constexpr int FOLD_BARRIERS = QM == 0 ? 0 : (QM == 1 ? 6 : 2);
    constexpr int PA = QM == 1 ? pstride(32) : pstride(CI), PG = QM == 1 ? pstride(32) : pstride(CO);
    constexpr int A_PLANE = NPA * PA, G_PLANE = NPG * PG, STAGE = 2 * A_PLANE + 2 * G_PLANE;      // [A_hi][A_lo][G_hi][G_lo]
    // (QM 1 / 3: nobody reads input channels 32..63 of the staged tile, QM 1 / 2: nor gradient channels 32..63 - the movers stage four
    //  8-channel vectors per pixel instead of eight; the LDS rows keep their 64-channel pitch)
    constexpr int VA = (QM == 1 || QM == 3) ? 4 : CI / 8, VG = (QM == 1 || QM == 2) ? 4 : CO / 8;
    constexpr int NA = (NPA * VA + 255) / 256, NG = NPG * VG / 256;
    static_assert(NPG * VG % 256 == 0, "whole gradient vectors per mover thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef s16x4 __attribute__((address_space(3))) * lptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co_blocks = (A.Cout + CO - 1) / CO;
    const int ci_blocks = (A.src.C + CI - 1) / CI;
    const int cb = blockIdx.x % co_blocks, ib = blockIdx.x / co_blocks;
    const int par = blockIdx.y;
    const int ks = blockIdx.z;
    const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + TH_ - 1) / TH_;
    const int tiles_img = tiles_y * tiles_x;
    const int ntiles = A.N * tiles_img;
    const int ntl = ks < ntiles ? (ntiles - ks + A.ksplit - 1) / A.ksplit : 0;      // tiles of this workgroup
    const int ntl2 = (ntl + 1) & ~1;                                                // barrier steps of both roles (padded to even)

    __shared__ __attribute__((aligned(16))) float s_xf[2 * CI];
    fill_xf<CI>(s_xf, A.src, ib * CI, tid);
#ifdef CDNET_WS_STAMPS
    const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (A.debug >> 8 ? A.debug >> 8 : 17) && lane == 0 && (wave == 0 || wave == 4);
    const int sbase = wave >= 4 ? 1024 : 0;
    int sn = 0;
#endif
    if (wave >= 4) {
        // ------------------------------- movers -------------------------------
        if (ntl == 0) {                                          // (a slice without tiles: the consumers store a slab of zeros)
            for (int i = 0; i < FOLD_BARRIERS; ++i) __syncthreads();
            return;
        }
        if (A.debug & 2) {                                       // ablation: consumers alone (whatever the LDS holds)
            for (int j = -1; j < ntl2; ++j) __syncthreads();
            for (int i = 0; i < FOLD_BARRIERS; ++i) __syncthreads();
            return;
        }
        const int ptid = tid - 256;
        const ConvSrc &s = A.src;
        const float *sx = reinterpret_cast<const float *>(s.x), *sr = reinterpret_cast<const float *>(s.res);
        const float *gx = reinterpret_cast<const float *>(A.g);
        const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
        const int slot_a = ptid % VA, cbase = ib * CI + slot_a * 8;
        const bool cok_a = cbase < s.C;
        const int slot_g = ptid % VG, co = cb * CO + slot_g * 8;
        const bool cok_g = co < A.Cout;
        const bool on = XF == 1 || (XF == 2 && s.scale != nullptr), relu = XF == 1 || (XF == 2 && s.relu != 0);
        const bool has_res = XF == 2 && sr != nullptr;
        float sc[8], sh[8];
        if (XF != 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = s_xf[slot_a * 8 + j]; sh[j] = s_xf[CI + slot_a * 8 + j]; }
        }
        // Requests go through buffer descriptors: a vector outside the image / source window / channels gets the byte offset ~0, the
        // range check returns zeros for it - no test, no select and no 64-bit address arithmetic per vector (a vector instruction of a
        // mover wave costs ~10 ns beside the consumers' MFMA stream, stamped: the movers, not the matrix pipe, bounded this kernel).
        // Validity of a tile's halo rows / columns = two scalar bit masks per tile; per vector: two shifts, or, bit extract.
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sx), 0, (int)((unsigned)A.N * s.Hs * rs * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sr ? sr : sx), 0, (int)((unsigned)A.N * s.Hs * rs * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gx), 0, (int)((unsigned)A.N * A.H * A.W * A.Cout * 4u), 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff) -> wg_f32x4 {
            return __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
        };
        // what depends on the thread only: halo coordinates (31 = a vector that never exists: the masks' bit 31 is always set), byte
        // offsets inside a tile and LDS addresses of its vectors
        int ahy[NA], ahx[NA], adst[NA], gpy[NG], gpx[NG], gdst[NG];
        unsigned aoffb[NA], goffb[NG];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = ptid + i * 256, pix = v / VA;
            const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
            const bool exists = v < NPA * VA && cok_a && (TAPS == 9 || (hy >= 1 && hy <= TH_ && hx >= 1 && hx <= TW));
            ahy[i] = exists ? hy : 31;
            ahx[i] = exists ? hx : 31;
            aoffb[i] = (unsigned)(hy * rs + hx * s.C + slot_a * 8) * 4u;
            adst[i] = pix * PA + slot_a * 16;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int pix = ptid / VG + i * (256 / VG);
            const int py = pix / TW, px = pix % TW;
            gpy[i] = cok_g ? py : 31;
            gpx[i] = cok_g ? px : 31;
            goffb[i] = (unsigned)((py * A.W + px) * A.Cout + slot_g * 8) * 4u;
            gdst[i] = 2 * A_PLANE + pix * PG + slot_g * 16;
        }
        const int ylo = s.off_y > 0 ? s.off_y : 0, yhi = A.H < s.off_y + s.Hs ? A.H : s.off_y + s.Hs;       // valid rows / columns of
        const int xlo = s.off_x > 0 ? s.off_x : 0, xhi = A.W < s.off_x + s.Ws ? A.W : s.off_x + s.Ws;       // the input: image and source window
        // bits [lo, hi) clear, everything else set (lo, hi clamped to [0, 31])
        auto bad_mask = [](int lo, int hi) -> unsigned {
            lo = lo < 0 ? 0 : (lo > 31 ? 31 : lo);
            hi = hi < lo ? lo : (hi > 31 ? 31 : hi);
            return ~(((1u << hi) - 1u) & ~((1u << lo) - 1u));
        };
        // tile cursor of the request stream: tile ks + j * ksplit as (image, tile row, tile column), advanced by ksplit without divisions
        int c_n, c_ty, c_tx, c_j = 0;
        {
            c_n = ks / tiles_img;
            const int rem = ks - c_n * tiles_img;
            c_ty = rem / tiles_x; c_tx = rem - c_ty * tiles_x;
        }
        const int d_n = A.ksplit / tiles_img, d_rem = A.ksplit - d_n * tiles_img, d_ty = d_rem / tiles_x, d_tx = d_rem - d_ty * tiles_x;
        wg_f32x4 pa_[2][NA][2], pg_[2][NG][2];
        unsigned av[2][NA];                  // byte offsets of the input vectors (read again for a residual operand), ~0 = zero fill
        auto issue = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            const int y0 = c_ty * TH_, x0 = c_tx * TW;
            // halo row r <-> y = y0 - 1 + r, halo column c <-> x = x0 - 1 + c
            const unsigned rowbad = bad_mask(ylo - (y0 - 1), yhi - (y0 - 1)), colbad = bad_mask(xlo - (x0 - 1), xhi - (x0 - 1));
            const unsigned abase_b = (unsigned)((c_n * s.Hs + (y0 - 1 - s.off_y)) * rs + (x0 - 1 - s.off_x) * s.C + ib * CI) * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned t = (rowbad >> ahy[i]) | (colbad >> ahx[i]);
                const unsigned voff = (abase_b + aoffb[i]) | (unsigned)(-(int)(t & 1u));
                av[R][i] = voff;
                pa_[R][i][0] = bload(rs_a, voff);
                pa_[R][i][1] = bload(rs_a, voff | 16u);          // (| 16: the ~0 of a zero-fill vector stays out of range)
            }
            const unsigned growbad = bad_mask(0, A.H - y0), gcolbad = bad_mask(0, A.W - x0);
            const unsigned gbase_b = (unsigned)(((c_n * A.H + y0) * A.W + x0) * A.Cout + cb * CO) * 4u;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const unsigned t = (growbad >> gpy[i]) | (gcolbad >> gpx[i]);
                const unsigned voff = (gbase_b + goffb[i]) | (unsigned)(-(int)(t & 1u));
                pg_[R][i][0] = bload(rs_g, voff);
                pg_[R][i][1] = bload(rs_g, voff | 16u);
            }
            if (c_j + 1 < ntl) {             // (past the end the last tile is requested and staged again, into the buffer nobody reads)
                ++c_j;
                c_tx += d_tx; if (c_tx >= tiles_x) { c_tx -= tiles_x; ++c_ty; }
                c_ty += d_ty; if (c_ty >= tiles_y) { c_ty -= tiles_y; ++c_n; }
                c_n += d_n;
            }
        };
        auto commit = [&](auto rc, int bufi) {
            constexpr int R = decltype(rc)::value;
            unsigned char *nb = smem + bufi * STAGE;
            wg_f32x4 rr[NA][2];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    rr[i][0] = bload(rs_r, av[R][i]);
                    rr[i][1] = bload(rs_r, av[R][i] | 16u);
                }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float x[8] = {pa_[R][i][0][0], pa_[R][i][0][1], pa_[R][i][0][2], pa_[R][i][0][3], pa_[R][i][1][0], pa_[R][i][1][1], pa_[R][i][1][2], pa_[R][i][1][3]};
                u32x4 hi, lo;
                if (XF == 1) {
                    // BatchNorm + ReLU with the zero fill folded into the clamp: med3(v, 0, lim) = max(v, 0) for lim = +inf, 0 for lim = 0
                    const float lim = __builtin_bit_cast(float, av[R][i] == ~0u ? 0u : 0x7f800000u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_fmed3f(fmaf(x[j], sc[j], sh[j]), 0.f, lim);
                    wg_split8(x, hi, lo);
                } else {
                    if (XF != 0 && on) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], sc[j], sh[j]);
                    }
                    if (has_res) {
                        const float r[8] = {rr[i][0][0], rr[i][0][1], rr[i][0][2], rr[i][0][3], rr[i][1][0], rr[i][1][1], rr[i][1][2], rr[i][1][3]};
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] += r[j];
                    }
                    if (XF != 0 && relu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                    }
                    wg_split8(x, hi, lo);
                    if (XF != 0) {                                           // (plain sources: the zero fill arrived as zeros)
                        const unsigned keep = av[R][i] == ~0u ? 0u : 0xffffffffu;
                        hi &= keep;
                        lo &= keep;
                    }
                }
                if (i < NA - 1 || ptid + i * 256 < NPA * VA) {
                    *reinterpret_cast<u32x4 *>(nb + adst[i]) = hi;
                    *reinterpret_cast<u32x4 *>(nb + A_PLANE + adst[i]) = lo;
                }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const float x[8] = {pg_[R][i][0][0], pg_[R][i][0][1], pg_[R][i][0][2], pg_[R][i][0][3], pg_[R][i][1][0], pg_[R][i][1][1], pg_[R][i][1][2], pg_[R][i][1][3]};
                u32x4 hi, lo;
                wg_split8(x, hi, lo);                                        // (no mask: the range check zero-filled it)
                *reinterpret_cast<u32x4 *>(nb + gdst[i]) = hi;
                *reinterpret_cast<u32x4 *>(nb + G_PLANE + gdst[i]) = lo;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        issue(I0{});                                             // tile 0
        issue(I1{});                                             // tile 1
        commit(I0{}, 0);
        issue(I0{});                                             // tile 2
        __syncthreads();
        for (int j = 0; j < ntl2; j += 2) {
            // consumers are on tile j (stage 0): fill stage 1 with tile j+1, then request tile j+3; then the other way round
            WG_STAMP(1);
#ifdef CDNET_WS_STAMPS
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // (stamped build: the older tile's loads have landed)
            WG_STAMP(2);
#endif
            commit(I1{}, 1);
            WG_STAMP(3);
            issue(I1{});
            WG_STAMP(4);
            __syncthreads();
            WG_STAMP(1);
#ifdef CDNET_WS_STAMPS
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            WG_STAMP(2);
#endif
            commit(I0{}, 0);
            WG_STAMP(3);
            issue(I0{});
            WG_STAMP(4);
            __syncthreads();
        }
#ifdef CDNET_WS_STAMPS
        if (stamp_on) g_wg_stamps[sbase + sn] = 0;
#endif
        for (int i = 0; i < FOLD_BARRIERS; ++i) __syncthreads();       // (the consumers' fold: every wave meets every barrier)
        return;
    }
    // ------------------------------- consumers -------------------------------
    const int quad = wave;
    const int qi = QM <= 1 ? 0 : (quad & 1), kgrp = QM == 0 ? 0 : (QM == 1 ? quad : (quad >> 1));
    const int wci = QM == 0 ? quad / CO_T : (QM == 2 ? qi : 0), wco = QM == 0 ? quad % CO_T : (QM == 3 ? qi : 0);
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int row0 = kgrp * KROWS;                               // this wave's first tile row
    const int a_lane = (8 * (grp >> 1) + q) * PA + (wci * 32 + 16 * (grp & 1) + 4 * p) * 2 + row0 * HALO_W * PA;
    const int g_lane = 2 * A_PLANE + (8 * (grp >> 1) + q) * PG + (wco * 32 + 16 * (grp & 1) + 4 * p) * 2 + row0 * TW * PG;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (ntl) __syncthreads();
    for (int j = 0; j < ntl2; ++j) {
        if (j >= ntl) { __syncthreads(); continue; }              // padding step
        WG_STAMP(10);
        const unsigned char *buf = smem + (j & 1) * STAGE;
        auto frag = [&](int off, int pstr) -> bf16x8 {
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + off));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(buf + off + 4 * pstr));
            s16x8 av;
            av[0] = a0[0]; av[1] = a0[1]; av[2] = a0[2]; av[3] = a0[3];
            av[4] = a1[0]; av[5] = a1[1]; av[6] = a1[2]; av[7] = a1[3];
            return __builtin_bit_cast(bf16x8, av);
        };
        // tap tau = ky * 9 + t of the tile (36 of them): input fragments of halo row ky + t / 3, column shift t % 3
        auto a_off = [&](int tau) {
            return TAPS == 1 ? a_lane + ((tau + 1) * HALO_W + 1) * PA                      // 1x1: the pixel itself
                             : a_lane + ((tau / TAPS + (tau % TAPS) / 3) * HALO_W + (tau % TAPS) % 3) * PA;
        };
        constexpr int NTAU = KROWS * TAPS;
        constexpr int PFD = 3, RING = PFD + 1;                   // fragment pairs requested PFD taps (3 PFD MFMAs) ahead of their use
        bf16x8 ah[RING], al[RING], gh[2], gl[2];
        if (!(A.debug & 1)) {
            gh[0] = frag(g_lane, PG); gl[0] = frag(g_lane + G_PLANE, PG);
#pragma unroll
            for (int k = 0; k < PFD && k < NTAU; ++k) { ah[k] = frag(a_off(k), PA); al[k] = frag(a_off(k) + A_PLANE, PA); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tau = 0; tau < NTAU; ++tau) {
                const int ky = tau / TAPS, t = tau % TAPS;
                // small terms first (wgrad_f32_kernel's order); behind every MFMA a scheduling fence and at most one fragment pair of a
                // later tap: program order is issue order
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tau % RING], gh[ky & 1], acc[t], 0, 0, 0);
                if (tau + PFD < NTAU) ah[(tau + PFD) % RING] = frag(a_off(tau + PFD), PA);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tau % RING], gl[ky & 1], acc[t], 0, 0, 0);
                if (tau + PFD < NTAU) al[(tau + PFD) % RING] = frag(a_off(tau + PFD) + A_PLANE, PA);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tau % RING], gh[ky & 1], acc[t], 0, 0, 0);
                if (ky + 1 < KROWS && t == (TAPS == 1 ? 0 : 2)) gh[(ky + 1) & 1] = frag(g_lane + (ky + 1) * TW * PG, PG);
                if (ky + 1 < KROWS && t == (TAPS == 1 ? 0 : 5)) gl[(ky + 1) & 1] = frag(g_lane + (ky + 1) * TW * PG + G_PLANE, PG);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WG_STAMP(11);
        __syncthreads();
    }
    WG_STAMP(12);
#ifdef CDNET_WS_STAMPS
    if (stamp_on) g_wg_stamps[sbase + sn] = 0;
#endif
    if (QM != 0) {
        // the waves that share a quadrant: QM 1 ((w0 + w1) + w2) + w3, QM 2 / 3 w0 + w2 and w1 + w3 - through the (dead) staging buffers
        static_assert(QM == 0 || 2 * STAGE >= 2 * TAPS * 16 * 64 * 4, "two parked accumulator sets fit the staging buffers");
        float *s_acc = reinterpret_cast<float *>(smem) + (QM == 1 ? 0 : (quad & 1) * TAPS * 16 * 64);
#pragma unroll 1
        for (int rd = 1; rd <= FOLD_BARRIERS / 2; ++rd) {
            const bool park = QM == 1 ? quad == rd : quad >= 2, take = QM == 1 ? quad == 0 : quad < 2;
            if (park) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_acc[(t * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (take) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += s_acc[(t * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (QM == 1 ? quad != 0 : quad >= 2) return;
    }
    float *slab = A.slab + ((((size_t)ks * A.npar + par) * ci_blocks + ib) * co_blocks + cb) * (size_t)(TAPS * CI * CO);
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[((size_t)t * CI + ci) * CO + wco * 32 + l31] = acc[t][r];
        }
}

// sum the split-K slabs in a fixed order and scatter into the PyTorch parameter-gradient layout
//   mode 0: Conv2d  dW[Cout][Cin][KH][KW]  (taps = KH*KW)
//   mode 2: ConvTranspose2d k4 s2 p1  dW[Cin][Cout][4][4]   (npar 4 x taps 4)
//   mode 3: ConvTranspose2d k2 s2     dW[Cin][Cout][2][2]   (npar 4 x taps 1)
typedef float rf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ slab, int ksplit, int npar, int ci_blocks, int co_blocks,
                                                  int TAPS, int CI, int CO, int Csrc_real, int Cin_real, int src_coff, int Cout,
                                                  int mode, float *__restrict__ dw, unsigned bidx, rf4 (*s_part)[64]) {
    // block = 256 consecutive slab elements (64 threads x one 16-byte vector) x 4 interleaved k-lanes (k = kq, kq+4, ...); every thread
    // keeps eight slabs' vectors in flight (the pass is pure streaming: 256 slabs x 147 KB for a 64 x 64 layer; with one 4-byte load per
    // thread and four in flight it ran at 0.8 TB/s, 49 us per launch on the weight-gradient stream); fixed-order combine
    const size_t per_ks = (size_t)npar * ci_blocks * co_blocks * TAPS * CI * CO;      // a multiple of 1024 (CI * CO = 4096)
    const int e = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const size_t i = ((size_t)bidx * 64 + e) * 4;
    rf4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = rf4{0.f, 0.f, 0.f, 0.f};
    if (i < per_ks) {
        const float *p = slab + i;
        int k = kq;
        for (; k + 28 < ksplit; k += 32) {
            rf4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const rf4 *>(p + (size_t)(k + 4 * u) * per_ks);
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += v[u];
        }
        for (; k < ksplit; k += 4) a[0] += *reinterpret_cast<const rf4 *>(p + (size_t)k * per_ks);
    }
    s_part[kq][e] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (kq != 0 || i >= per_ks) return;
    const rf4 sv = (s_part[0][e] + s_part[1][e]) + (s_part[2][e] + s_part[3][e]);
    size_t r = i;
    const int co_l0 = r % CO; r /= CO;                           // four consecutive output channels (CO is a multiple of 32)
    const int ci_l = r % CI; r /= CI;
    const int tap = r % TAPS; r /= TAPS;
    const int cb = r % co_blocks; r /= co_blocks;
    const int ib = r % ci_blocks; r /= ci_blocks;
    const int par = (int)r;
    const int ci = src_coff + ib * CI + ci_l;
    if (ib * CI + ci_l >= Csrc_real) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = cb * CO + co_l0 + j;
        if (co >= Cout) continue;
        size_t o;
        if (mode == 0) {
            o = ((size_t)co * Cin_real + ci) * TAPS + tap;
        } else if (mode == 6) {
            // stride-2 3x3 conv through the space-to-depth view: GEMM ci = (a, b, c), Cin_real = 4*Ct; dW[co][c][ky][kx] with
            // ky = 2*(tap/3 - 1) + a + 1 (the other tap / parity combinations multiply structural zeros)
            const int Ct = Cin_real / 4;
            const int a_ = ci / (2 * Ct), b_ = (ci / Ct) & 1, c = ci % Ct;
            const int kh = 2 * (tap / 3 - 1) + a_ + 1, kw = 2 * (tap % 3 - 1) + b_ + 1;
            if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
            o = (((size_t)co * Ct + c) * 3 + kh) * 3 + kw;
        } else if (mode == 2) {
            const int a_ = par >> 1, b_ = par & 1, ty = tap >> 1, tx = tap & 1;
            const int kh = a_ == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2);
            const int kw = b_ == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 0 : 2);
            o = (((size_t)ci * Cout + co) * 4 + kh) * 4 + kw;
        } else {
            o = (((size_t)ci * Cout + co) * 2 + (par >> 1)) * 2 + (par & 1);
        }
        dw[o] = sv[j];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slab, int ksplit, int npar, int ci_blocks,
                                                           int co_blocks, int TAPS, int CI, int CO, int Csrc_real, int Cin_real,
                                                           int src_coff, int Cout, int mode, float *__restrict__ dw) {
    __shared__ rf4 s_part[4][64];
    wgrad_reduce_body(slab, ksplit, npar, ci_blocks, co_blocks, TAPS, CI, CO, Csrc_real, Cin_real, src_coff, Cout, mode, dw, blockIdx.x, s_part);
}

// the reduces of several layers in one launch (cdnet_wgrad_reduce_batch): the descriptors sit in device memory in launch order,
// `block0` ascending; a workgroup finds its layer by bisection over uniform (scalar) loads and runs the same body
struct ReduceDesc {
    const float *slab;
    float *dw;
    int ksplit, npar, ci_blocks, co_blocks, taps, CI, CO, Csrc_real, Cin_real, src_coff, Cout, mode;
    int block0, blocks;
};
static_assert(sizeof(ReduceDesc) == sizeof(cdnet_wgrad_reduce_desc), "cdnet_wgrad_reduce_desc layout");

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceDesc *__restrict__ tab, int n) {
    __shared__ rf4 s_part[4][64];
    const unsigned b = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((unsigned)tab[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const ReduceDesc d = tab[lo];
    wgrad_reduce_body(d.slab, d.ksplit, d.npar, d.ci_blocks, d.co_blocks, d.taps, d.CI, d.CO, d.Csrc_real, d.Cin_real, d.src_coff, d.Cout,
                      d.mode, d.dw, b - (unsigned)d.block0, s_part);
}

template <int CI_T, int CO_T, int TAPS, bool RES, int XF>
int launch_wgrad_ws(const WgradArgs &A, hipStream_t st) {
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int smem = 2 * (NPIX_A * pstride(CI) + NPIX_G * pstride(CO));
    auto kern = wgrad_ws_kernel<CI_T, CO_T, TAPS, RES, XF>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return check_launch("hipFuncSetAttribute(wgrad_ws)");
        attr_done = true;
    }
    dim3 grid(cdiv(A.src.C, CI) * cdiv(A.Cout, CO), A.npar, A.ksplit);
    kern<<<grid, NT, smem, st>>>(A);
    return check_launch("wgrad_ws_kernel");
}

template <int CI_T, int CO_T, int TAPS, bool RES>
int launch_wgrad_gen(const WgradArgs &A, hipStream_t st) {
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int stage_bytes = NPIX_A * pstride(CI) + NPIX_G * pstride(CO);
    constexpr int fold_bytes = 4 * TAPS * 16 * 64 * 4;
    constexpr int smem = 2 * stage_bytes > fold_bytes ? 2 * stage_bytes : fold_bytes;
    auto kern = wgrad_kernel<CI_T, CO_T, TAPS, RES>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return check_launch("hipFuncSetAttribute(wgrad)");
        attr_done = true;
    }
    dim3 grid(cdiv(A.src.C, CI) * cdiv(A.Cout, CO), A.npar, A.ksplit);
    kern<<<grid, NT, smem, st>>>(A);
    return check_launch("wgrad_kernel");
}

// the wave-specialised fp32 kernel: 3x3 layers on 64 x 64 channel blocks (ostride 1), tensors below 4 GB (32-bit byte offsets)
template <int TAPS>
int launch_wgrad_ws32(const WgradArgs &A, hipStream_t st) {
    const ConvSrc &s = A.src;
    const bool plain = !s.scale && !s.relu && !s.res, fast = s.scale && s.shift && s.relu == 1 && !s.res;
    auto go2 = [&](auto xf_c, auto qm_c) -> int {
        constexpr int XF = decltype(xf_c)::value, QM = decltype(qm_c)::value;
        constexpr int smem = QM == 1 ? 2 * (2 * (8 + 2) * HALO_W * pstride(32) + 2 * 8 * TW * pstride(32))
                                     : 2 * (2 * NPIX_A32 * pstride(64) + 2 * NPIX_G32 * pstride(64));
        auto kern = wgrad_ws32_kernel<XF, TAPS, QM>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
                return check_launch("hipFuncSetAttribute(wgrad_ws32)");
            attr_done = true;
        }
        dim3 grid(cdiv(A.src.C, 64) * cdiv(A.Cout, 64), A.npar, A.ksplit);
        kern<<<grid, 512, smem, st>>>(A);
        return check_launch("wgrad_ws32_kernel");
    };
    // quadrants of zero padding are not multiplied (QM): at most 32 channels on the input side, the output side, or both
    const bool csmall = s.C <= 32 && !(A.debug & 16), gsmall = A.Cout <= 32 && !(A.debug & 16);       // (16: tests / A-B - all four quadrants)
    auto go = [&](auto xf_c) -> int {
        if (csmall && gsmall) return go2(xf_c, std::integral_constant<int, 1>{});
        if (gsmall) return go2(xf_c, std::integral_constant<int, 2>{});
        if (csmall) return go2(xf_c, std::integral_constant<int, 3>{});
        return go2(xf_c, std::integral_constant<int, 0>{});
    };
    if (plain) return go(std::integral_constant<int, 0>{});
    if (fast) return go(std::integral_constant<int, 1>{});
    return go(std::integral_constant<int, 2>{});
}

template <int CI_T, int CO_T, int TAPS>
int launch_wgrad_f32(const WgradArgs &A, hipStream_t st) {
    static const int use_ws32 = getenv("CDNET_WGRAD_WS32") ? atoi(getenv("CDNET_WGRAD_WS32")) : 1;
    if (CI_T == 2 && CO_T == 2 && (TAPS == 9 || TAPS == 1) && use_ws32 && !(A.debug & 8) && A.ostride == 1 && A.npar == 1 && !A.src.pool &&
        (long long)A.N * A.H * A.W * (A.src.C > A.Cout ? A.src.C : A.Cout) < (1LL << 30) &&
        (long long)A.N * A.src.Hs * (A.src.row_stride ? A.src.row_stride : A.src.Ws * A.src.C) < (1LL << 30))
        return launch_wgrad_ws32<(TAPS == 1 ? 1 : 9)>(A, st);
    constexpr int CI = CI_T * 32, CO = CO_T * 32;
    constexpr int smem = 2 * (NPIX_A * pstride(CI) + NPIX_G * pstride(CO));
    auto go = [&](auto qm_c) -> int {
        constexpr int QM = decltype(qm_c)::value;
        auto kern = wgrad_f32_kernel<CI_T, CO_T, TAPS, QM>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
                return check_launch("hipFuncSetAttribute(wgrad_f32)");
            attr_done = true;
        }
        dim3 grid(cdiv(A.src.C, CI) * cdiv(A.Cout, CO), A.npar, A.ksplit);
        kern<<<grid, NT32, smem, st>>>(A);
        return check_launch("wgrad_f32_kernel");
    };
    if constexpr (CI_T == 2 && CO_T == 2 && TAPS != 9) {
        // quadrants of zero padding are not multiplied (the transposed convolutions into the decoder's 32- and 16-channel blocks)
        if (A.Cout <= 32 && !A.src.pool && !(A.debug & 16)) return A.src.C <= 32 ? go(std::integral_constant<int, 1>{}) : go(std::integral_constant<int, 2>{});
    }
    return go(std::integral_constant<int, 0>{});
}

// kernel choice: un-pooled sources with CI <= 64 take the specialised-wave kernel with a compile-time transform
// (plain / training-mode fast path / generic); pooled sources, CI = 128 blocks and generic-with-residual the 8-wave one
template <int CI_T, int CO_T, int TAPS>
int launch_wgrad(const WgradArgs &A, hipStream_t st) {
    const ConvSrc &s = A.src;
    if (s.f16 == 2) return launch_wgrad_f32<CI_T, CO_T, TAPS>(A, st);
    const bool res = s.res != nullptr;
    if constexpr (CI_T == 1 && CO_T == 4) {
        // at most 32 output channels: one 32 x 32 block per workgroup, the consumer waves split the tile's rows (wgrad_ws_kernel<1, 1>)
        if (A.Cout <= 32 && !s.pool && !(A.debug & (8 | 16))) {           // (16: tests - the 32 x 128 form)
            const bool plain = !s.scale && !s.relu && !s.f16 && !res;
            const bool fast = s.scale && s.relu && s.f16;
            if (plain) return launch_wgrad_ws<1, 1, TAPS, false, XF_PLAIN>(A, st);
            if (fast) return res ? launch_wgrad_ws<1, 1, TAPS, true, XF_FAST>(A, st) : launch_wgrad_ws<1, 1, TAPS, false, XF_FAST>(A, st);
            if (!res) return launch_wgrad_ws<1, 1, TAPS, false, XF_GEN>(A, st);
        }
    }
    if (CI_T != 4 && !s.pool && !(A.debug & 8)) {
        const bool plain = !s.scale && !s.relu && !s.f16 && !res;
        const bool fast = s.scale && s.relu && s.f16;
        if (plain) return launch_wgrad_ws<CI_T, CO_T, TAPS, false, XF_PLAIN>(A, st);
        if (fast) return res ? launch_wgrad_ws<CI_T, CO_T, TAPS, true, XF_FAST>(A, st) : launch_wgrad_ws<CI_T, CO_T, TAPS, false, XF_FAST>(A, st);
        if (!res) return launch_wgrad_ws<CI_T, CO_T, TAPS, false, XF_GEN>(A, st);
    }
    return res ? launch_wgrad_gen<CI_T, CO_T, TAPS, true>(A, st) : launch_wgrad_gen<CI_T, CO_T, TAPS, false>(A, st);
}

template <int TAPS>
int dispatch_wgrad(const WgradArgs &A, int ci_t, hipStream_t st) {
    if (ci_t == 1) return launch_wgrad<1, 4, TAPS>(A, st);
    if (ci_t == 2) return launch_wgrad<2, 2, TAPS>(A, st);
    return launch_wgrad<4, 1, TAPS>(A, st);
}

}  // namespace

extern "C" size_t cdnet_conv_wgrad_slab_floats(int C_src, int Cout, int taps, int npar, int ci_tiles, int ksplit) {
    const int CI = ci_tiles * 32, CO = (4 / ci_tiles) * 32;
    return (size_t)ksplit * npar * cdiv(C_src, CI) * cdiv(Cout, CO) * taps * CI * CO;
}

extern "C" int cdnet_conv_backward_weight(const cdnet_conv_src *src, int src_coff, int Csrc_real, int Cin_real,
                                          const uint16_t *grad_out,
                                          int Cout, int N, int H, int W, int taps, int npar, int ostride, int ci_tiles,
                                          int ksplit, float *slab, float *dw, int mode, void *stream) {
    CDNET_REQUIRE(src && src->x && grad_out && slab && dw, "cdnet_conv_backward_weight: null pointer");
    CDNET_REQUIRE(ci_tiles == 1 || ci_tiles == 2 || ci_tiles == 4, "cdnet_conv_backward_weight: ci_tiles=%d", ci_tiles);
    CDNET_REQUIRE(src->C % 8 == 0, "cdnet_conv_backward_weight: source channels %d not a multiple of 8", src->C);
    CDNET_REQUIRE(!(src->res && src->pool), "cdnet_conv_backward_weight: a pooled source with a residual branch is not supported");
    CDNET_REQUIRE(!(src->f16 == 2 && src->pool), "cdnet_conv_backward_weight: fp32 pooled sources must be materialised");
    CDNET_REQUIRE(ksplit >= 1 && N > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "cdnet_conv_backward_weight: bad size (Cout %% 8)");
    const bool defer = (mode & CDNET_WGRAD_DEFER_REDUCE) != 0;       // leave the slabs: cdnet_wgrad_reduce_batch sums them later
    mode &= ~CDNET_WGRAD_DEFER_REDUCE;
    CDNET_REQUIRE((taps == 9 && npar == 1 && ostride == 1 && mode == 0) || (taps == 1 && npar == 1 && ostride == 1 && mode == 0) ||
                  (taps == 9 && npar == 1 && ostride == 1 && mode == 6) ||
                  (taps == 4 && npar == 4 && ostride == 2 && mode == 2) || (taps == 1 && npar == 4 && ostride == 2 && mode == 3),
                  "cdnet_conv_backward_weight: taps=%d npar=%d ostride=%d mode=%d", taps, npar, ostride, mode);
    hipStream_t st = (hipStream_t)stream;
    WgradArgs A;
    A.src = *reinterpret_cast<const ConvSrc *>(src);
    A.src_coff = src_coff;
    A.g = grad_out;
    A.slab = slab;
    A.N = N; A.H = H; A.W = W; A.Cout = Cout;
    A.taps = taps; A.npar = npar; A.ostride = ostride; A.ksplit = ksplit;
    static const int dbg = getenv("CDNET_WGRAD_DEBUG") ? atoi(getenv("CDNET_WGRAD_DEBUG")) : 0;
    A.debug = dbg;
    int rc;
    if (taps == 9) rc = dispatch_wgrad<9>(A, ci_tiles, st);
    else if (taps == 4) rc = dispatch_wgrad<4>(A, ci_tiles, st);
    else rc = dispatch_wgrad<1>(A, ci_tiles, st);
    if (rc != CDNET_OK || defer) return rc;
    const int CI = ci_tiles * 32, CO = (4 / ci_tiles) * 32;
    const int ci_blocks = cdiv(src->C, CI), co_blocks = cdiv(Cout, CO);
    const size_t per_ks = (size_t)npar * ci_blocks * co_blocks * taps * CI * CO;
    const int blocks = (int)((per_ks + 255) / 256);
    wgrad_reduce_kernel<<<blocks, 256, 0, st>>>(slab, ksplit, npar, ci_blocks, co_blocks, taps, CI, CO, Csrc_real, Cin_real, src_coff, Cout,
                                                mode, dw);
    return check_launch("wgrad_reduce_kernel");
}

extern "C" int cdnet_wgrad_reduce_desc_fill(int C_src, int src_coff, int Csrc_real, int Cin_real, int Cout, int taps, int npar,
                                            int ci_tiles, int ksplit, const float *slab, float *dw, int mode, int block0,
                                            cdnet_wgrad_reduce_desc *out) {
    CDNET_REQUIRE(out && slab && dw, "cdnet_wgrad_reduce_desc_fill: null pointer");
    CDNET_REQUIRE(ci_tiles == 1 || ci_tiles == 2 || ci_tiles == 4, "cdnet_wgrad_reduce_desc_fill: ci_tiles=%d", ci_tiles);
    CDNET_REQUIRE(ksplit >= 1 && C_src > 0 && Cout > 0 && block0 >= 0, "cdnet_wgrad_reduce_desc_fill: bad size");
    mode &= ~CDNET_WGRAD_DEFER_REDUCE;
    CDNET_REQUIRE(mode == 0 || mode == 2 || mode == 3 || mode == 6, "cdnet_wgrad_reduce_desc_fill: mode=%d", mode);
    const int CI = ci_tiles * 32, CO = (4 / ci_tiles) * 32;
    out->slab = slab; out->dw = dw;
    out->ksplit = ksplit; out->npar = npar;
    out->ci_blocks = cdiv(C_src, CI); out->co_blocks = cdiv(Cout, CO);
    out->taps = taps; out->CI = CI; out->CO = CO;
    out->Csrc_real = Csrc_real; out->Cin_real = Cin_real; out->src_coff = src_coff; out->Cout = Cout; out->mode = mode;
    const size_t per_ks = (size_t)npar * out->ci_blocks * out->co_blocks * taps * CI * CO;
    out->block0 = block0;
    out->blocks = (int)((per_ks + 255) / 256);
    return CDNET_OK;
}

extern "C" int cdnet_wgrad_reduce_batch(const cdnet_wgrad_reduce_desc *table_dev, int n, int total_blocks, void *stream) {
    CDNET_REQUIRE(table_dev && n >= 1 && total_blocks >= 1, "cdnet_wgrad_reduce_batch: empty table");
    wgrad_reduce_batch_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const ReduceDesc *>(table_dev), n);
    return check_launch("wgrad_reduce_batch_kernel");
}
