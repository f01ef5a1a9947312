// Training-only streaming kernels (HBM-bound): BatchNorm/ReLU/max-pool backward, the DAM head backward, the five-term
// CDNet loss with its gradient, and the fused Adam step.  Everything reduces through per-block partials that are
// summed in a fixed order, so a training step is bit-reproducible.
//
// Replaces, in the reference's train_util_dam.py:
//   loss terms :167-276 with loss.py:131-260 (dice / weighted cyclic dice), nn.NLLLoss(reduction='none') x weight map,
//   nn.MSELoss; loss.backward() :307 (non-convolution parts); optimizer.step() :308 with utils.py:915-918 (Adam).
#include "common.h"
#include "xform.h"

using namespace cdnet;

namespace {

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ float ld16(unsigned short u, bool f16) { return f16 ? h2f(u) : bf2f(u); }

union V16 {
    uint4 u;
    unsigned short h[8];
};

inline int lin_grid(size_t total, int cap = 2048) {
    size_t g = (total + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

// out[k] = sum_b partial[b][k]: one wave per output, lanes stride over b, fixed butterfly -> deterministic
// first pixel (or pool window) of a thread in the "VPP threads per pixel, 256 / VPP pixels per block" layouts below; when VPP does
// not divide 256 (HRNet's 48 / 80 / 144 channels) the left-over threads of the block sit the loop out
__device__ __forceinline__ unsigned first_pixel(unsigned ppb, int VPP) {
    return (int)threadIdx.x < (int)ppb * VPP ? blockIdx.x * ppb + threadIdx.x / VPP : 0xffffffffu;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial, int nb, int K, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    float s = 0.f;
    for (int b = lane; b < nb; b += 64) s += partial[(size_t)b * K + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[k] = s;
}

// ======================================================================================================
// BatchNorm + ReLU (+ residual) (+ consumers' max-pool / F.pad) backward
// ======================================================================================================
struct GradIn {
    const unsigned short *g;     // bf16 NHWC [N][Hg][Wg][C]: gradient w.r.t. this tensor as seen by one consumer
    int Hg, Wg;
    int oy, ox;                  // consumer read (y - oy, x - ox) of this tensor  => gradient sits at (y + oy, x + ox)
    int pooled;                  // consumer read maxpool2x2 of this tensor (1 floor / 2 ceil): route to the argmax
    int coff, cstride;           // channel slice of a wider gradient tensor
};

struct BnBwdArgs {
    const unsigned short *raw;   // stored forward output [N][H][W][C]
    const unsigned short *res;   // optional residual added before the ReLU
    int f16;                     // storage format of raw/res
    const float *scale, *shift;  // forward affine (NULL: identity)
    int relu;
    const float *mean, *invstd;  // saved batch statistics
    GradIn gin[3];
    int ngin;
    int N, H, W, C;
    float *partial;              // reduce: [nblocks][2][C]
    const float *k1, *k2, *k3;   // apply: draw = k1*(dz - k2 - xhat*k3)
    unsigned short *draw;        // bf16 [N][H][W][C]
    unsigned short *dz_out;      // optional bf16 copy of dz (gradient of the residual branch)
    int rev;                     // apply pass walks the tensor back to front (see cdnet_bn_backward)
};

__device__ __forceinline__ void ldf8(const void *base, size_t e, float *v) {
    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + e);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void stf8(void *base, size_t e, const float *v) {
    float4 *p = reinterpret_cast<float4 *>(reinterpret_cast<float *>(base) + e);
    p[0] = make_float4(v[0], v[1], v[2], v[3]);
    p[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// fp32 tensors (f16 == 2): the activated value in plain fp32 arithmetic; relu == 2: `res` is the stored post-ReLU output of the
// unit (fused residual epilogue) and the mask is read from it
__device__ __forceinline__ void act8_f32(const BnBwdArgs &A, size_t e, const float *sc, const float *sh, float *a, float *rawf) {
    float x[8], r[8];
    ldf8(A.raw, e, x);
    if (A.res) ldf8(A.res, e, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (rawf) rawf[j] = x[j];
        float v = A.scale ? fmaf(x[j], sc[j], sh[j]) : x[j];
        if (A.relu == 2) v = r[j];
        else {
            if (A.res) v += r[j];
            if (A.relu) v = fmaxf(v, 0.f);
        }
        a[j] = v;
    }
}

// activated value (rounded to bf16 like the forward staging does) of 8 channels at one pixel
template <bool F32 = false>
__device__ __forceinline__ void act8(const BnBwdArgs &A, size_t e, const float *sc, const float *sh, float *a, float *rawf) {
    if (F32) { act8_f32(A, e, sc, sh, a, rawf); return; }
    V16 r, rr;
    r.u = *reinterpret_cast<const uint4 *>(A.raw + e);
    rr.u = make_uint4(0, 0, 0, 0);
    if (A.res) rr.u = *reinterpret_cast<const uint4 *>(A.res + e);
    const bool f16 = A.f16 != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = ld16(r.h[j], f16);
        if (rawf) rawf[j] = x;
        float v = A.scale ? fmaf(x, sc[j], sh[j]) : x;
        if (A.res) v += ld16(rr.h[j], f16);
        if (A.relu) v = fmaxf(v, 0.f);
        a[j] = bf2f(f2bf(v));
    }
}

// dz for 8 channels of pixel (n,y,x); also returns xhat
template <bool WANT_XHAT, bool F32 = false>
__device__ __forceinline__ void dz8(const BnBwdArgs &A, int n, int y, int x, int c0, const float *sc, const float *sh,
                                    const float *mu, const float *is, float *dz, float *xhat) {
    const size_t e = (((size_t)n * A.H + y) * A.W + x) * A.C + c0;
    float a[8], rawf[8];
    act8<F32>(A, e, sc, sh, a, rawf);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    for (int k = 0; k < A.ngin; ++k) {
        const GradIn &gi = A.gin[k];
        if (!gi.pooled) {
            const int yy = y + gi.oy, xx = x + gi.ox;
            if (yy >= 0 && yy < gi.Hg && xx >= 0 && xx < gi.Wg) {
                const size_t ge = (((size_t)n * gi.Hg + yy) * gi.Wg + xx) * gi.cstride + gi.coff + c0;
                if (F32) {
                    float t[8];
                    ldf8(gi.g, ge, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += t[j];
                } else {
                    V16 v;
                    v.u = *reinterpret_cast<const uint4 *>(gi.g + ge);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += bf2f(v.h[j]);
                }
            }
        } else {
            const int py = y >> 1, px = x >> 1;
            if (py < gi.Hg && px < gi.Wg) {
                // is (y,x) the first maximum of its 2x2 window?  (nn.MaxPool2d backward)
                bool sel[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sel[j] = true;
                const int q0 = (y & 1) * 2 + (x & 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == q0) continue;
                    const int yy = (y & ~1) + (q >> 1), xx = (x & ~1) + (q & 1);
                    if (yy >= A.H || xx >= A.W) continue;
                    float b[8];
                    act8<F32>(A, (((size_t)n * A.H + yy) * A.W + xx) * A.C + c0, sc, sh, b, nullptr);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sel[j] = sel[j] && (q < q0 ? a[j] > b[j] : a[j] >= b[j]);
                }
                const size_t ge = (((size_t)n * gi.Hg + py) * gi.Wg + px) * gi.cstride + gi.coff + c0;
                if (F32) {
                    float t[8];
                    ldf8(gi.g, ge, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += sel[j] ? t[j] : 0.f;
                } else {
                    V16 v;
                    v.u = *reinterpret_cast<const uint4 *>(gi.g + ge);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += sel[j] ? bf2f(v.h[j]) : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dz[j] = (A.relu && !(a[j] > 0.f)) ? 0.f : g[j];
        if (WANT_XHAT) xhat[j] = A.mean ? (rawf[j] - mu[j]) * is[j] : 0.f;
    }
}

__device__ __forceinline__ void load8(const float *p, int c0, float *o, float dflt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = p ? p[c0 + j] : dflt;
}

// Both passes are pure streaming (HBM bound): every thread keeps BN_U independent pixels in flight per iteration.
constexpr int BN_U = 4;
constexpr int BN_MAX_BLOCKS = 2048;      // workspace rows
// blocks actually launched by the streaming passes: two 256-thread workgroups per CU keep 64-98 KB of loads in flight per CU, and the
// finalize pass (one workgroup per channel walking the partial rows with a 2 * C stride) has a quarter of the rows to sum - measured
// (bench.py, 16 tiles): 2048 blocks 1 641 tiles/s, 1024 1 645-1 655, 768 1 652-1 655, 512 1 654-1 663, 256 1 605-1 609
static int bn_blocks_cap() {
    return 512 > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : 512;
}

template <bool F32 = false>
__device__ __forceinline__ void dz8_at(const BnBwdArgs &A, unsigned p, unsigned HW, int c0, const float *sc, const float *sh, const float *mu,
                                       const float *is, float *dz, float *xh) {
    const unsigned n = p / HW, r = p - n * HW;
    const unsigned y = r / (unsigned)A.W, x = r - y * (unsigned)A.W;
    dz8<true, F32>(A, (int)n, (int)y, (int)x, c0, sc, sh, mu, is, dz, xh);
}

// The storage format of the raw tensor (fp16 / bf16) and the ReLU mode (0 none, 1 mask from the recomputed activation, 2 mask from the
// stored output in `res`) are wave-uniform run-time fields: the pixel loop is instantiated per combination and chosen once at the top
// (tested per element they cost a scalar branch + an exec-mask save around every dz: ~25 instructions per element, the reduce pass ran at
// 3.9 TB/s beside the apply pass's 5.1).
template <typename Body>
__device__ __forceinline__ void bn_bwd_dispatch(bool f16, int relu, Body &&body) {
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (f16) {
        if (relu == 0) body(T_{}, std::integral_constant<int, 0>{});
        else if (relu == 1) body(T_{}, std::integral_constant<int, 1>{});
        else body(T_{}, std::integral_constant<int, 2>{});
    } else {
        if (relu == 0) body(F_{}, std::integral_constant<int, 0>{});
        else if (relu == 1) body(F_{}, std::integral_constant<int, 1>{});
        else body(F_{}, std::integral_constant<int, 2>{});
    }
}

// Window kernels: the layer's consumers are one 2x2 max-pool plus NF same-size un-shifted tensors (the skip connection).
// One pooling window per thread: the four activations are read once, the pooled gradient goes to the first maximum
// (nn.MaxPool2d backward).  All loads are unconditional (clamped coordinates) and issued before any arithmetic.
template <int NF, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_window_kernel(BnBwdArgs A, int kp) {
    __shared__ float s_red[APPLY ? 1 : 256][17];
    const int VPP = A.C / 8;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 8;
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8], k3[8], s1[8], s2[8];
    load8(A.scale, c0, sc, 1.f); load8(A.shift, c0, sh, 0.f); load8(A.mean, c0, mu, 0.f); load8(A.invstd, c0, is, 1.f);
    if (APPLY) { load8(A.k1, c0, k1, 1.f); load8(A.k2, c0, k2, 0.f); load8(A.k3, c0, k3, 0.f); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const unsigned H = A.H, W = A.W, Hp = (H + 1) / 2, Wp = (W + 1) / 2, nwin = (unsigned)A.N * Hp * Wp;
    const unsigned ppb = 256 / VPP;
    const GradIn gp = A.gin[kp];
    int kf[2] = {0, 0};
    {
        int m = 0;
        for (int k = 0; k < A.ngin && m < NF; ++k)
            if (k != kp) kf[m++] = k;
    }
    bn_bwd_dispatch(A.f16 != 0, A.relu != 0 ? 1 : 0, [&](auto f16_c, auto relu_c) {
    constexpr bool f16 = decltype(f16_c)::value, relu = decltype(relu_c)::value != 0;
    for (unsigned w0 = first_pixel(ppb, VPP); w0 < nwin; w0 += gridDim.x * ppb) {
        const unsigned w = (APPLY && A.rev) ? nwin - 1 - w0 : w0;
        const unsigned n = w / (Hp * Wp), r = w - n * Hp * Wp;
        const unsigned py = r / Wp, px = r - py * Wp;
        V16 raw[4], g[4][NF > 0 ? NF : 1], gv;
        unsigned pix[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned yy = 2 * py + (q >> 1), xx = 2 * px + (q & 1);
            ok[q] = yy < H && xx < W;
            yy = yy < H ? yy : H - 1;
            xx = xx < W ? xx : W - 1;
            pix[q] = (n * H + yy) * W + xx;
            raw[q].u = *reinterpret_cast<const uint4 *>(A.raw + (size_t)pix[q] * A.C + c0);
#pragma unroll
            for (int m = 0; m < NF; ++m)
                g[q][m].u = *reinterpret_cast<const uint4 *>(A.gin[kf[m]].g + (size_t)pix[q] * A.gin[kf[m]].cstride + A.gin[kf[m]].coff + c0);
        }
        const bool pok = py < (unsigned)gp.Hg && px < (unsigned)gp.Wg;
        {
            const unsigned cy = pok ? py : 0, cx = pok ? px : 0;
            gv.u = *reinterpret_cast<const uint4 *>(gp.g + (((size_t)n * gp.Hg + cy) * gp.Wg + cx) * gp.cstride + gp.coff + c0);
        }
        V16 o[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x[4], a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x[q] = ld16(raw[q].h[j], f16);
                float v = fmaf(x[q], sc[j], sh[j]);
                if (relu) v = fmaxf(v, 0.f);
                a[q] = bf2f(f2bf(v));
            }
            int bi = 0;
            float best = a[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (ok[q] && a[q] > best) { best = a[q]; bi = q; }
            const float gpool = pok ? bf2f(gv.h[j]) : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float gs = (bi == q) ? gpool : 0.f;
#pragma unroll
                for (int m = 0; m < NF; ++m) gs += bf2f(g[q][m].h[j]);
                const float dz = (!ok[q] || (relu && !(a[q] > 0.f))) ? 0.f : gs;
                const float xh = (x[q] - mu[j]) * is[j];
                if (APPLY) o[q].h[j] = f2bf(k1[j] * (dz - k2[j] - xh * k3[j]));
                else { s1[j] += dz; s2[j] = fmaf(dz, xh, s2[j]); }
            }
        }
        if (APPLY) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) *reinterpret_cast<uint4 *>(A.draw + (size_t)pix[q] * A.C + c0) = o[q].u;
        }
    }
    });
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s_red[tid][j] = s1[j]; s_red[tid][8 + j] = s2[j]; }
        __syncthreads();
        for (int q = tid; q < 16 * VPP; q += 256) {
            const int sl = q % VPP, j = q / VPP;
            float t = 0.f;
            for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
            float *op = A.partial + (size_t)blockIdx.x * 2 * A.C;
            op[(j >> 3) * A.C + sl * 8 + (j & 7)] = t;
        }
    }
}

// Flat kernels: every gradient source is a same-size un-shifted tensor, so dz needs only the pixel index.  Loads of BN_U
// pixels are issued back to back (clamped index, no branch), then each pixel is folded into the sums / written out.
template <int NG, bool RES>
__global__ __launch_bounds__(256) void bn_bwd_reduce_flat_kernel(BnBwdArgs A) {
    __shared__ float s_red[256][17];
    const int VPP = A.C / 8;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 8;
    float sc[8], sh[8], mu[8], is[8];
    load8(A.scale, c0, sc, 1.f); load8(A.shift, c0, sh, 0.f); load8(A.mean, c0, mu, 0.f); load8(A.invstd, c0, is, 1.f);
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const unsigned npix = (unsigned)(A.N * A.H * A.W);
    const unsigned ppb = 256 / VPP, step = gridDim.x * ppb;
    bn_bwd_dispatch(A.f16 != 0, A.relu, [&](auto f16_c, auto relu_c) {
        constexpr bool F16 = decltype(f16_c)::value;
        constexpr int RELU = decltype(relu_c)::value;
        for (unsigned p0 = first_pixel(ppb, VPP); p0 < npix; p0 += step * BN_U) {
            V16 raw[BN_U], res[BN_U], g[BN_U][NG];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                unsigned p = p0 + u * step;
                p = p < npix ? p : npix - 1;
                raw[u].u = *reinterpret_cast<const uint4 *>(A.raw + (size_t)p * A.C + c0);
                if (RES) res[u].u = *reinterpret_cast<const uint4 *>(A.res + (size_t)p * A.C + c0);
#pragma unroll
                for (int k = 0; k < NG; ++k)
                    g[u][k].u = *reinterpret_cast<const uint4 *>(A.gin[k].g + (size_t)p * A.gin[k].cstride + A.gin[k].coff + c0);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const bool valid = p0 + u * step < npix;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = ld16(raw[u].h[j], F16);
                    float v = fmaf(x, sc[j], sh[j]);
                    if (RES) v = RELU == 2 ? bf2f(res[u].h[j]) : v + ld16(res[u].h[j], F16);     // RELU 2: res IS the stored output
                    float gs = bf2f(g[u][0].h[j]);
#pragma unroll
                    for (int k = 1; k < NG; ++k) gs += bf2f(g[u][k].h[j]);
                    // the forward rounds the activation to bf16 before the ReLU; rounding keeps the sign
                    const bool keep = valid & (RELU == 0 || bf2f(f2bf(v)) > 0.f);
                    const float dz = keep ? gs : 0.f;
                    s1[j] += dz;
                    s2[j] = fmaf(dz, (x - mu[j]) * is[j], s2[j]);
                    if (RES) res[u].h[j] = f2bf(dz);                             // (the register is free: dz for the store below)
                }
                // a residual unit's bn2: dz is the 1x1 branch's gradient and is stored anyway - by this pass, so that the second pass reads
                // one tensor instead of the NG gradient sources and the mask again (cdnet_bn_backward; it then sees dz rounded to bf16)
                if (RES && A.dz_out && valid) *reinterpret_cast<uint4 *>(A.dz_out + (size_t)(p0 + u * step) * A.C + c0) = res[u].u;
            }
        }
    });
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_red[tid][j] = s1[j]; s_red[tid][8 + j] = s2[j]; }
    __syncthreads();
    for (int q = tid; q < 16 * VPP; q += 256) {
        const int sl = q % VPP, j = q / VPP;
        float t = 0.f;
        for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
        float *o = A.partial + (size_t)blockIdx.x * 2 * A.C;
        o[(j >> 3) * A.C + sl * 8 + (j & 7)] = t;
    }
}

template <int NG, bool RES>
__global__ __launch_bounds__(256) void bn_bwd_apply_flat_kernel(BnBwdArgs A) {
    const int VPP = A.C / 8;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 8;
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8], k3[8];
    load8(A.scale, c0, sc, 1.f); load8(A.shift, c0, sh, 0.f); load8(A.mean, c0, mu, 0.f); load8(A.invstd, c0, is, 1.f);
    load8(A.k1, c0, k1, 1.f); load8(A.k2, c0, k2, 0.f); load8(A.k3, c0, k3, 0.f);
    const unsigned npix = (unsigned)(A.N * A.H * A.W);
    const unsigned ppb = 256 / VPP, step = gridDim.x * ppb;
    const bool rev = A.rev != 0;
    bn_bwd_dispatch(A.f16 != 0, A.relu, [&](auto f16_c, auto relu_c) {
        constexpr bool F16 = decltype(f16_c)::value;
        constexpr int RELU = decltype(relu_c)::value;
        for (unsigned p0 = first_pixel(ppb, VPP); p0 < npix; p0 += step * BN_U) {
            V16 raw[BN_U], res[BN_U], g[BN_U][NG];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                unsigned p = p0 + u * step;
                p = p < npix ? p : npix - 1;
                p = rev ? npix - 1 - p : p;
                raw[u].u = *reinterpret_cast<const uint4 *>(A.raw + (size_t)p * A.C + c0);
                if (RES) res[u].u = *reinterpret_cast<const uint4 *>(A.res + (size_t)p * A.C + c0);
#pragma unroll
                for (int k = 0; k < NG; ++k)
                    g[u][k].u = *reinterpret_cast<const uint4 *>(A.gin[k].g + (size_t)p * A.gin[k].cstride + A.gin[k].coff + c0);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const unsigned p = p0 + u * step;
                V16 o, z;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = ld16(raw[u].h[j], F16);
                    float v = fmaf(x, sc[j], sh[j]);
                    if (RES) v = RELU == 2 ? bf2f(res[u].h[j]) : v + ld16(res[u].h[j], F16);
                    float gs = bf2f(g[u][0].h[j]);
#pragma unroll
                    for (int k = 1; k < NG; ++k) gs += bf2f(g[u][k].h[j]);
                    const bool keep = RELU == 0 || bf2f(f2bf(v)) > 0.f;
                    const float dz = keep ? gs : 0.f;
                    o.h[j] = f2bf(k1[j] * (dz - k2[j] - (x - mu[j]) * is[j] * k3[j]));
                    z.h[j] = f2bf(dz);
                }
                if (p < npix) {
                    const unsigned pw = rev ? npix - 1 - p : p;
                    *reinterpret_cast<uint4 *>(A.draw + (size_t)pw * A.C + c0) = o.u;
                    if (RES) *reinterpret_cast<uint4 *>(A.dz_out + (size_t)pw * A.C + c0) = z.u;
                }
            }
        }
    });
}

// fp32 variants of the flat kernels: 4 channels per thread (one float4 per tensor and pixel), BN_U pixels in flight, plain
// fp32 arithmetic (no 16-bit rounding of the activation); relu == 2 reads the mask from the stored output in `res`.
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load4(const float *p, int c0, float *o, float dflt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = p ? p[c0 + j] : dflt;
}

template <int NG, bool RES, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_flat32_kernel(BnBwdArgs A) {
    __shared__ float s_red[APPLY ? 1 : 256][9];
    const int VPP = A.C / 4;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 4;
    float sc[4], sh[4], mu[4], is[4], k1[4], k2[4], k3[4], s1[4], s2[4];
    load4(A.scale, c0, sc, 1.f); load4(A.shift, c0, sh, 0.f); load4(A.mean, c0, mu, 0.f); load4(A.invstd, c0, is, 1.f);
    if (APPLY) { load4(A.k1, c0, k1, 1.f); load4(A.k2, c0, k2, 0.f); load4(A.k3, c0, k3, 0.f); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const unsigned npix = (unsigned)(A.N * A.H * A.W);
    const unsigned ppb = 256 / VPP, step = gridDim.x * ppb;
    const bool relu = A.relu != 0, outmask = A.relu == 2;
    const float *raw = reinterpret_cast<const float *>(A.raw), *resp = reinterpret_cast<const float *>(A.res);
    float *draw = reinterpret_cast<float *>(A.draw), *dzo = reinterpret_cast<float *>(A.dz_out);
    for (unsigned p0 = first_pixel(ppb, VPP); p0 < npix; p0 += step * BN_U) {
        bn_f32x4 x[BN_U], r[BN_U], g[BN_U][NG];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            unsigned p = p0 + u * step;
            p = p < npix ? p : npix - 1;
            if (APPLY && A.rev) p = npix - 1 - p;
            x[u] = *reinterpret_cast<const bn_f32x4 *>(raw + (size_t)p * A.C + c0);
            if (RES) r[u] = *reinterpret_cast<const bn_f32x4 *>(resp + (size_t)p * A.C + c0);
#pragma unroll
            for (int k = 0; k < NG; ++k)
                g[u][k] = *reinterpret_cast<const bn_f32x4 *>(reinterpret_cast<const float *>(A.gin[k].g) + (size_t)p * A.gin[k].cstride + A.gin[k].coff + c0);
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const unsigned p = p0 + u * step;
            const bool valid = p < npix;
            bn_f32x4 o, z;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fmaf(x[u][j], sc[j], sh[j]);
                if (RES) v = outmask ? r[u][j] : v + r[u][j];
                float gs = g[u][0][j];
#pragma unroll
                for (int k = 1; k < NG; ++k) gs += g[u][k][j];
                const float dz = (!valid || (relu && !(v > 0.f))) ? 0.f : gs;
                const float xh = (x[u][j] - mu[j]) * is[j];
                if (APPLY) { o[j] = k1[j] * (dz - k2[j] - xh * k3[j]); z[j] = dz; }
                else { s1[j] += dz; s2[j] = fmaf(dz, xh, s2[j]); z[j] = dz; }
            }
            if (APPLY && valid) {
                const unsigned pw = A.rev ? npix - 1 - p : p;
                *reinterpret_cast<bn_f32x4 *>(draw + (size_t)pw * A.C + c0) = o;
                if (RES) *reinterpret_cast<bn_f32x4 *>(dzo + (size_t)pw * A.C + c0) = z;
            }
            // the sums pass of a residual unit's bn2 already leaves dz (the 1x1 branch's gradient): the second pass then reads one
            // tensor instead of the NG gradient sources and the mask again (cdnet_bn_backward)
            if (!APPLY && RES && dzo && valid) *reinterpret_cast<bn_f32x4 *>(dzo + (size_t)p * A.C + c0) = z;
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_red[tid][j] = s1[j]; s_red[tid][4 + j] = s2[j]; }
        __syncthreads();
        for (int q = tid; q < 8 * VPP; q += 256) {
            const int sl = q % VPP, j = q / VPP;
            float t = 0.f;
            for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
            float *op = A.partial + (size_t)blockIdx.x * 2 * A.C;
            op[(j >> 2) * A.C + sl * 4 + (j & 3)] = t;
        }
    }
}

template <bool APPLY>
static void launch_flat32(const BnBwdArgs &A, int nb, hipStream_t st) {
    switch (A.ngin * 2 + (A.res ? 1 : 0)) {
        case 2: bn_bwd_flat32_kernel<1, false, APPLY><<<nb, 256, 0, st>>>(A); break;
        case 3: bn_bwd_flat32_kernel<1, true, APPLY><<<nb, 256, 0, st>>>(A); break;
        case 4: bn_bwd_flat32_kernel<2, false, APPLY><<<nb, 256, 0, st>>>(A); break;
        case 5: bn_bwd_flat32_kernel<2, true, APPLY><<<nb, 256, 0, st>>>(A); break;
        case 6: bn_bwd_flat32_kernel<3, false, APPLY><<<nb, 256, 0, st>>>(A); break;
        default: bn_bwd_flat32_kernel<3, true, APPLY><<<nb, 256, 0, st>>>(A); break;
    }
}

// fp32 variant of the window kernels (one 2x2 max-pool consumer + NF same-size un-shifted ones: the encoder's conv1_2 ... conv5_3):
// one pooling window x 4 channels per thread, the four raw vectors, their flat gradients and the pooled gradient requested back to back,
// plain fp32 arithmetic, the pooled gradient to the first maximum of relu(bn(raw)) (nn.MaxPool2d backward) - bit-identical to the
// generic per-pixel path (bn_bwd_reduce_kernel / bn_bwd_apply_kernel<true>), which reads a window's raw vectors once per pixel.
template <int NF, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_window32_kernel(BnBwdArgs A, int kp) {
    __shared__ float s_red[APPLY ? 1 : 256][9];
    const int VPP = A.C / 4;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 4;
    float sc[4], sh[4], mu[4], is[4], k1[4], k2[4], k3[4], s1[4], s2[4];
    load4(A.scale, c0, sc, 1.f); load4(A.shift, c0, sh, 0.f); load4(A.mean, c0, mu, 0.f); load4(A.invstd, c0, is, 1.f);
    if (APPLY) { load4(A.k1, c0, k1, 1.f); load4(A.k2, c0, k2, 0.f); load4(A.k3, c0, k3, 0.f); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const unsigned H = A.H, W = A.W, Hp = (H + 1) / 2, Wp = (W + 1) / 2, nwin = (unsigned)A.N * Hp * Wp;
    const unsigned ppb = 256 / VPP;
    const bool relu = A.relu != 0;
    const GradIn gp = A.gin[kp];
    const float *gpool_p = reinterpret_cast<const float *>(gp.g);
    const float *raw = reinterpret_cast<const float *>(A.raw);
    float *draw = reinterpret_cast<float *>(A.draw);
    const float *gf[NF > 0 ? NF : 1];
    int gcs[NF > 0 ? NF : 1], gco[NF > 0 ? NF : 1];
    {
        int m = 0;
        for (int k = 0; k < A.ngin && m < NF; ++k)
            if (k != kp) { gf[m] = reinterpret_cast<const float *>(A.gin[k].g); gcs[m] = A.gin[k].cstride; gco[m] = A.gin[k].coff; ++m; }
    }
    for (unsigned w0 = first_pixel(ppb, VPP); w0 < nwin; w0 += gridDim.x * ppb) {
        const unsigned w = (APPLY && A.rev) ? nwin - 1 - w0 : w0;
        const unsigned n = w / (Hp * Wp), r = w - n * Hp * Wp;
        const unsigned py = r / Wp, px = r - py * Wp;
        bn_f32x4 x[4], g[4][NF > 0 ? NF : 1], gv;
        unsigned pix[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned yy = 2 * py + (q >> 1), xx = 2 * px + (q & 1);
            ok[q] = yy < H && xx < W;
            yy = yy < H ? yy : H - 1;
            xx = xx < W ? xx : W - 1;
            pix[q] = (n * H + yy) * W + xx;
            x[q] = *reinterpret_cast<const bn_f32x4 *>(raw + (size_t)pix[q] * A.C + c0);
#pragma unroll
            for (int m = 0; m < NF; ++m)
                g[q][m] = *reinterpret_cast<const bn_f32x4 *>(gf[m] + (size_t)pix[q] * gcs[m] + gco[m] + c0);
        }
        const bool pok = py < (unsigned)gp.Hg && px < (unsigned)gp.Wg;
        {
            const unsigned cy = pok ? py : 0, cx = pok ? px : 0;
            gv = *reinterpret_cast<const bn_f32x4 *>(gpool_p + (((size_t)n * gp.Hg + cy) * gp.Wg + cx) * gp.cstride + gp.coff + c0);
        }
        bn_f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = fmaf(x[q][j], sc[j], sh[j]);
                a[q] = relu ? fmaxf(v, 0.f) : v;
            }
            int bi = 0;
            float best = a[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (ok[q] && a[q] > best) { best = a[q]; bi = q; }
            const float gpool = pok ? gv[j] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // the generic path's order: 0 + the sources in argument order (kp = the pooled one's position)
                const float pt = (bi == q) ? gpool : 0.f;
                float gsum;
                if (NF == 0) gsum = 0.f + pt;
                else if (NF == 1) gsum = (0.f + (kp == 0 ? pt : g[q][0][j])) + (kp == 0 ? g[q][0][j] : pt);
                else gsum = ((0.f + (kp == 0 ? pt : g[q][0][j])) + (kp == 0 ? g[q][0][j] : (kp == 1 ? pt : g[q][NF - 1][j]))) +
                            (kp == 2 ? pt : g[q][NF - 1][j]);
                const float dz = (!ok[q] || (relu && !(a[q] > 0.f))) ? 0.f : gsum;
                const float xh = (x[q][j] - mu[j]) * is[j];
                if (APPLY) o[q][j] = k1[j] * (dz - k2[j] - xh * k3[j]);
                else { s1[j] += dz; s2[j] = fmaf(dz, xh, s2[j]); }
            }
        }
        if (APPLY) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) *reinterpret_cast<bn_f32x4 *>(draw + (size_t)pix[q] * A.C + c0) = o[q];
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_red[tid][j] = s1[j]; s_red[tid][4 + j] = s2[j]; }
        __syncthreads();
        for (int q = tid; q < 8 * VPP; q += 256) {
            const int sl = q % VPP, j = q / VPP;
            float t = 0.f;
            for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
            float *op = A.partial + (size_t)blockIdx.x * 2 * A.C;
            op[(j >> 2) * A.C + sl * 4 + (j & 3)] = t;
        }
    }
}

template <bool APPLY>
static void launch_window32(const BnBwdArgs &A, int nflat, int kp, int nb, hipStream_t st) {
    if (nflat == 0) bn_bwd_window32_kernel<0, APPLY><<<nb, 256, 0, st>>>(A, kp);
    else if (nflat == 1) bn_bwd_window32_kernel<1, APPLY><<<nb, 256, 0, st>>>(A, kp);
    else bn_bwd_window32_kernel<2, APPLY><<<nb, 256, 0, st>>>(A, kp);
}

template <bool F32>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs A) {
    __shared__ float s_red[256][17];
    const int VPP = A.C / 8;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 8;
    float sc[8], sh[8], mu[8], is[8];
    load8(A.scale, c0, sc, 1.f); load8(A.shift, c0, sh, 0.f); load8(A.mean, c0, mu, 0.f); load8(A.invstd, c0, is, 1.f);
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const unsigned HW = (unsigned)(A.H * A.W), npix = (unsigned)A.N * HW;
    const unsigned ppb = 256 / VPP;                                 // pixels per block per sub-iteration
    const unsigned step = gridDim.x * ppb;
    for (unsigned p0 = first_pixel(ppb, VPP); p0 < npix; p0 += step * BN_U) {
        float dz[BN_U][8], xh[BN_U][8];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const unsigned p = p0 + u * step;
            if (p < npix) dz8_at<F32>(A, p, HW, c0, sc, sh, mu, is, dz[u], xh[u]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { dz[u][j] = 0.f; xh[u][j] = 0.f; }
            }
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += dz[u][j]; s2[j] = fmaf(dz[u][j], xh[u][j], s2[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_red[tid][j] = s1[j]; s_red[tid][8 + j] = s2[j]; }
    __syncthreads();
    // threads with the same slot: tid, tid+VPP, ...  -> fixed-order sum; 16 values x VPP slots spread over the block
    for (int q = tid; q < 16 * VPP; q += 256) {
        const int sl = q % VPP, j = q / VPP;
        float t = 0.f;
        for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
        float *o = A.partial + (size_t)blockIdx.x * 2 * A.C;
        o[(j >> 3) * A.C + sl * 8 + (j & 7)] = t;
    }
}

// sums [nb][2][C] -> dgamma, dbeta and the apply coefficients; one workgroup per channel, fixed LDS tree (deterministic)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nb, int C, float M, const float *gamma,
                                                              const float *invstd, float *dgamma, float *dbeta, float *k1, float *k2,
                                                              float *k3) {
    __shared__ double s_a[256], s_b[256];
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nb; b += 256) { s1 += (double)partial[((size_t)b * 2) * C + c]; s2 += (double)partial[((size_t)b * 2 + 1) * C + c]; }
    s_a[lane] = s1; s_b[lane] = s2;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if (lane < o) { s_a[lane] += s_a[lane + o]; s_b[lane] += s_b[lane + o]; }
        __syncthreads();
    }
    s1 = s_a[0]; s2 = s_b[0];
    if (lane == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        k1[c] = gamma[c] * invstd[c];
        k2[c] = (float)(s1 / M);
        k3[c] = (float)(s2 / M);
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs A) {
    const int VPP = A.C / 8;
    const int tid = threadIdx.x;
    const int slot = tid % VPP, c0 = slot * 8;
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8], k3[8];
    load8(A.scale, c0, sc, 1.f); load8(A.shift, c0, sh, 0.f); load8(A.mean, c0, mu, 0.f); load8(A.invstd, c0, is, 1.f);
    load8(A.k1, c0, k1, 1.f); load8(A.k2, c0, k2, 0.f); load8(A.k3, c0, k3, 0.f);
    const unsigned HW = (unsigned)(A.H * A.W), npix = (unsigned)A.N * HW;
    const unsigned ppb = 256 / VPP;
    const unsigned step = gridDim.x * ppb;
    for (unsigned p0 = first_pixel(ppb, VPP); p0 < npix; p0 += step * BN_U) {
        float dz[BN_U][8], xh[BN_U][8];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const unsigned p = p0 + u * step;
            if (p < npix) dz8_at<F32>(A, p, HW, c0, sc, sh, mu, is, dz[u], xh[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const unsigned p = p0 + u * step;
            if (p >= npix) continue;
            if (F32) {
                float o32[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o32[j] = k1[j] * (dz[u][j] - k2[j] - xh[u][j] * k3[j]);
                if (A.draw) stf8(A.draw, (size_t)p * A.C + c0, o32);
                if (A.dz_out) stf8(A.dz_out, (size_t)p * A.C + c0, dz[u]);
                continue;
            }
            V16 o, z;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o.h[j] = f2bf(k1[j] * (dz[u][j] - k2[j] - xh[u][j] * k3[j]));
                z.h[j] = f2bf(dz[u][j]);
            }
            if (A.draw) *reinterpret_cast<uint4 *>(A.draw + (size_t)p * A.C + c0) = o.u;
            if (A.dz_out) *reinterpret_cast<uint4 *>(A.dz_out + (size_t)p * A.C + c0) = z.u;
        }
    }
}

// ======================================================================================================
// DAM head backward
// ======================================================================================================
struct HeadFeat {
    const unsigned short *raw;
    const unsigned short *res;
    const float *scale;
    const float *shift;
    int relu;
    int f16;
};

struct HeadW {
    float wp[64], wd[9][64], wm[3][64];
    float bp, bd[9], bm[3];
    float a1;
    float a2[9];
};
constexpr int HEADW_FLOATS = sizeof(HeadW) / 4;      // 855; the gradient block has the same layout

// fp32-stored feature (f16 == 2): 8 channels, plain fp32 arithmetic
__device__ __forceinline__ void feat8_f32(const HeadFeat &f, size_t pix, int c0, const float *s_sc, const float *s_sh, float *v) {
    float x[8], r[8];
    ldf8(f.raw, pix * 64 + c0, x);
    if (f.res) ldf8(f.res, pix * 64 + c0, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = x[j];
        if (f.scale) t = fmaf(t, s_sc[c0 + j], s_sh[c0 + j]);
        if (f.res) t += r[j];
        if (f.relu) t = fmaxf(t, 0.f);
        v[j] = t;
    }
}

__device__ __forceinline__ float feat1(const HeadFeat &f, size_t pix, int c, const float *s_sc, const float *s_sh) {
    float x = f.f16 ? h2f(f.raw[pix * 64 + c]) : bf2f(f.raw[pix * 64 + c]);
    if (f.scale || f.res || f.relu) {
        if (f.scale) x = fmaf(x, s_sc[c], s_sh[c]);
        if (f.res) x += f.f16 ? h2f(f.res[pix * 64 + c]) : bf2f(f.res[pix * 64 + c]);
        if (f.relu) x = fmaxf(x, 0.f);
        x = bf2f(f2bf(x));
    }
    return x;
}

// 8 channels [c0, c0+8) of the feature at one pixel
__device__ __forceinline__ void feat8(const HeadFeat &f, size_t pix, int c0, const float *s_sc, const float *s_sh, float *v) {
    if (f.f16 == 2) { feat8_f32(f, pix, c0, s_sc, s_sh, v); return; }
    V16 r, rr;
    r.u = *reinterpret_cast<const uint4 *>(f.raw + pix * 64 + c0);
    rr.u = make_uint4(0, 0, 0, 0);
    if (f.res) rr.u = *reinterpret_cast<const uint4 *>(f.res + pix * 64 + c0);
    if (f.f16 && f.scale && f.relu) {            // training-mode feature: packed math (xform.h)
        const xf_u32x4 a = __builtin_bit_cast(xf_u32x4, r.u), b = __builtin_bit_cast(xf_u32x4, rr.u);
        if (f.res) xf_bnrelu_f16_to_f32<true>(a, b, s_sc + c0, s_sh + c0, v);
        else xf_bnrelu_f16_to_f32<false>(a, a, s_sc + c0, s_sh + c0, v);
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = f.f16 ? h2f(r.h[j]) : bf2f(r.h[j]);
        if (f.scale || f.res || f.relu) {
            if (f.scale) x = fmaf(x, s_sc[c0 + j], s_sh[c0 + j]);
            if (f.res) x += f.f16 ? h2f(rr.h[j]) : bf2f(rr.h[j]);
            if (f.relu) x = fmaxf(x, 0.f);
            x = bf2f(f2bf(x));
        }
        v[j] = x;
    }
}

// the same in two steps for 16-bit features: the raw vectors first (a kernel puts the loads of all its features in flight before it
// touches any of them - one memory round trip per pixel instead of one per feature), the lazily applied transform second.
// FM: 0 the feature is plain bf16 (a materialised tensor), 1 fp16 raw x scale + shift + residual -> ReLU (a lazily transformed
// training-mode residual-unit output), 2 anything (run-time flags)
struct FeatRaw8 {
    uint4 r, s;
};
template <int FM>
__device__ __forceinline__ void load_raw8(const HeadFeat &f, size_t pix, int c0, FeatRaw8 &R) {
    R.r = *reinterpret_cast<const uint4 *>(f.raw + pix * 64 + c0);
    R.s = make_uint4(0, 0, 0, 0);
    if (FM == 1 || (FM == 2 && f.res)) R.s = *reinterpret_cast<const uint4 *>(f.res + pix * 64 + c0);
}
template <int FM>
__device__ __forceinline__ void feat_from_raw8(const HeadFeat &f, const FeatRaw8 &R, int c0, const float *s_sc, const float *s_sh, float *v) {
    const xf_u32x4 a = __builtin_bit_cast(xf_u32x4, R.r), b = __builtin_bit_cast(xf_u32x4, R.s);
    if (FM == 1) { xf_bnrelu_f16_to_f32<true>(a, b, s_sc + c0, s_sh + c0, v); return; }
    if (FM == 0) {
        V16 r0;
        r0.u = R.r;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(r0.h[j]);
        return;
    }
    if (f.f16 && f.scale && f.relu) {
        if (f.res) xf_bnrelu_f16_to_f32<true>(a, b, s_sc + c0, s_sh + c0, v);
        else xf_bnrelu_f16_to_f32<false>(a, a, s_sc + c0, s_sh + c0, v);
        return;
    }
    V16 r, rr;
    r.u = R.r; rr.u = R.s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = f.f16 ? h2f(r.h[j]) : bf2f(r.h[j]);
        if (f.scale || f.res || f.relu) {
            if (f.scale) x = fmaf(x, s_sc[c0 + j], s_sh[c0 + j]);
            if (f.res) x += f.f16 ? h2f(rr.h[j]) : bf2f(rr.h[j]);
            if (f.relu) x = fmaxf(x, 0.f);
            x = bf2f(f2bf(x));
        }
        v[j] = x;
    }
}

__device__ __forceinline__ void feat64(const HeadFeat &f, size_t pix, const float *s_sc, const float *s_sh, float *v) {
    if (f.f16 == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) feat8_f32(f, pix, q * 8, s_sc, s_sh, v + q * 8);
        return;
    }
    const uint4 *pr = reinterpret_cast<const uint4 *>(f.raw + pix * 64);
    const uint4 *ps = f.res ? reinterpret_cast<const uint4 *>(f.res + pix * 64) : nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        V16 r, rr;
        r.u = pr[q];
        rr.u = make_uint4(0, 0, 0, 0);
        if (ps) rr.u = ps[q];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = f.f16 ? h2f(r.h[j]) : bf2f(r.h[j]);
            if (f.scale || ps || f.relu) {
                if (f.scale) x = fmaf(x, s_sc[q * 8 + j], s_sh[q * 8 + j]);
                if (ps) x += f.f16 ? h2f(rr.h[j]) : bf2f(rr.h[j]);
                if (f.relu) x = fmaxf(x, 0.f);
                x = bf2f(f2bf(x));
            }
            v[q * 8 + j] = x;
        }
    }
}

__device__ __forceinline__ void store64_bf16(unsigned short *dst, const float *v) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        V16 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = f2bf(v[q * 8 + j]);
        reinterpret_cast<uint4 *>(dst)[q] = o.u;
    }
}

__device__ __forceinline__ void feat16(const HeadFeat &f, size_t pix, int q, const float *s_sc, const float *s_sh, float *v) {
    feat8(f, pix, q * 16, s_sc, s_sh, v);
    feat8(f, pix, q * 16 + 8, s_sc, s_sh, v + 8);
}
__device__ __forceinline__ float quad_sum(float v) { return xf_quad_sum(v); }
// 16 gradient values of one lane: element offset e of a bf16 (f32 = false) or fp32 tensor
__device__ __forceinline__ void store16_grad(unsigned short *base, size_t e, const float *v, bool f32);
__device__ __forceinline__ void store16_bf16(unsigned short *dst, const float *v) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        V16 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = f2bf(v[h2 * 8 + j]);
        reinterpret_cast<uint4 *>(dst)[h2] = o.u;
    }
}

__device__ __forceinline__ void store16_grad(unsigned short *base, size_t e, const float *v, bool f32) {
    if (f32) { stf8(base, e, v); stf8(base, e + 8, v + 8); }
    else store16_bf16(base + e, v);
}

// Kernel 1: four lanes per pixel (16 channels each).  Recomputes the head, writes the three feature gradients, the 13
// per-pixel coefficients {dpt, du[9], dm[3]} (f32 [px][16]) for the weight-gradient kernel, and per-block partial sums of
// the 23 scalar parameter gradients (biases and gate weights).
// FM: the storage of all three features (see FeatRaw8) - with a compile-time transform all three features' loads and the pixel's
// upstream gradients are in flight together; the run-time format tests of FM 2 make 6 000 instructions of branches whose joins
// drain the loads
template <int FM>
__global__ __launch_bounds__(256) void dam_head_bwd_kernel(HeadFeat f1, HeadFeat f2, HeadFeat f3, const HeadW *__restrict__ hw,
                                                           const float *__restrict__ dmask, const float *__restrict__ dpoint,
                                                           const float *__restrict__ ddir, int N, int plane,
                                                           unsigned short *__restrict__ df1, unsigned short *__restrict__ df2,
                                                           unsigned short *__restrict__ df3, float *__restrict__ coef,
                                                           float *__restrict__ partial) {
    __shared__ HeadW w;
    __shared__ float s_sc[3][64], s_sh[3][64];
    __shared__ float s_red[64][24];
    const int tid = threadIdx.x;
    {
        const float *src = reinterpret_cast<const float *>(hw);
        float *dst = reinterpret_cast<float *>(&w);
        for (int i = tid; i < HEADW_FLOATS; i += 256) dst[i] = src[i];
        if (tid < 64) {
            const HeadFeat *fs[3] = {&f1, &f2, &f3};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_sc[k][tid] = fs[k]->scale ? fs[k]->scale[tid] : 1.f;
                s_sh[k][tid] = fs[k]->scale ? fs[k]->shift[tid] : 0.f;
            }
        }
    }
    __syncthreads();
    const size_t total = (size_t)N * plane;
    const int q = tid & 3;
    const bool f32 = f1.f16 == 2;                // fp32 features -> fp32 feature gradients
    // scalar gradients accumulated by the q == 0 lane of each pixel: dbm[3] | dbd[9] | dbp | da1 | da2[9]
    float sg[23];
#pragma unroll
    for (int j = 0; j < 23; ++j) sg[j] = 0.f;
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        const size_t i = base + (tid >> 2);
        const bool ok = i < total;
        const size_t ii = ok ? i : total - 1;
        const size_t n = ii / plane, p = ii - n * plane;
        // (the head weights stay in LDS: without this fence the compiler hoists a lane's 208 weight reads out of the loop)
        asm volatile("" ::: "memory");
        float v[16];
        FeatRaw8 R3[2], R2[2], R1[2];
        constexpr bool all16 = FM != 2;             // the generic instantiation keeps one feature at a time (registers)
        if (all16) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                load_raw8<FM>(f3, ii, q * 16 + h2 * 8, R3[h2]);
                load_raw8<FM>(f2, ii, q * 16 + h2 * 8, R2[h2]);
                load_raw8<FM>(f1, ii, q * 16 + h2 * 8, R1[h2]);
            }
        }
        // upstream gradients of this pixel: in flight with the features
        float go_m[3], go_d[9], go_p;
#pragma unroll
        for (int k = 0; k < 3; ++k) go_m[k] = ok ? dmask[(n * 3 + k) * plane + p] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) go_d[k] = ok ? ddir[(n * 9 + k) * plane + p] : 0.f;
        go_p = ok ? dpoint[n * plane + p] : 0.f;
        auto feat = [&](const HeadFeat &f, const FeatRaw8 (&R)[2], int k) {
            if (all16) {
                feat_from_raw8<FM>(f, R[0], q * 16, s_sc[k], s_sh[k], v);
                feat_from_raw8<FM>(f, R[1], q * 16 + 8, s_sc[k], s_sh[k], v + 8);
            } else {
                feat16(f, ii, q, s_sc[k], s_sh[k], v);
            }
        };
        feat(f3, R3, 2);
        const float pt = quad_sum(xf_dot16(w.wp + q * 16, v)) + w.bp;
        const float sg1 = 1.f / (1.f + expf(-(w.a1 * pt)));
        const float g1 = 1.f + sg1;
        feat(f2, R2, 1);
        float u[9], d[9], q2 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            u[k] = quad_sum(xf_dot16(w.wd[k] + q * 16, v));
            d[k] = fmaf(g1, u[k], w.bd[k]);
            q2 = fmaf(w.a2[k], d[k], q2);
        }
        const float sg2 = 1.f / (1.f + expf(-q2));
        const float g2 = 1.f + sg2;
        feat(f1, R1, 0);
        float dm[3], dg2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mk = quad_sum(xf_dot16(w.wm[k] + q * 16, v));
            const float go = go_m[k];
            dg2 = fmaf(go, mk, dg2);
            dm[k] = go * g2;
            sg[k] += go;
        }
        // dF1 = sum_k dm_k * wm_k   (this lane's 16 channels)
        xf_axpy16(dm[0], w.wm[0] + q * 16, v, false);
        xf_axpy16(dm[1], w.wm[1] + q * 16, v, true);
        xf_axpy16(dm[2], w.wm[2] + q * 16, v, true);
        if (ok) store16_grad(df1, ii * 64 + q * 16, v, f32);
        const float dq2 = dg2 * sg2 * (1.f - sg2);
        float du[9], dg1 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float dd = go_d[k] + dq2 * w.a2[k];
            sg[3 + k] += dd;
            sg[14 + k] = fmaf(dq2, d[k], sg[14 + k]);
            dg1 = fmaf(dd, u[k], dg1);
            du[k] = dd * g1;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) xf_axpy16(du[k], w.wd[k] + q * 16, v, k != 0);
        if (ok) store16_grad(df2, ii * 64 + q * 16, v, f32);
        const float dsg1 = dg1 * sg1 * (1.f - sg1);
        const float dpt = go_p + dsg1 * w.a1;
        sg[12] += dpt;
        sg[13] = fmaf(dsg1, pt, sg[13]);
        xf_axpy16(dpt, w.wp + q * 16, v, false);
        if (ok) {
            store16_grad(df3, ii * 64 + q * 16, v, f32);
            // coefficient row [dpt | du[9] | dm[3] | 0 0 0]: lane q writes floats 4q..4q+3
            float4 cf;
            if (q == 0) cf = make_float4(dpt, du[0], du[1], du[2]);
            else if (q == 1) cf = make_float4(du[3], du[4], du[5], du[6]);
            else if (q == 2) cf = make_float4(du[7], du[8], dm[0], dm[1]);
            else cf = make_float4(dm[2], 0.f, 0.f, 0.f);
            reinterpret_cast<float4 *>(coef + ii * 16)[q] = cf;
        }
    }
    // every lane of a pixel accumulated identical scalar sums: take the q == 0 lanes, fixed-order block reduction
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < 23; ++j) s_red[tid >> 2][j] = sg[j];
    }
    __syncthreads();
    if (tid < 23) {
        float s = 0.f;
        for (int k = 0; k < 64; ++k) s += s_red[k][tid];
        int dst;                                     // HeadW tail: bp, bd[9], bm[3], a1, a2[9]
        if (tid < 3) dst = 832 + 10 + tid;
        else if (tid < 12) dst = 832 + 1 + (tid - 3);
        else if (tid == 12) dst = 832;
        else if (tid == 13) dst = 832 + 13;
        else dst = 832 + 14 + (tid - 14);
        partial[(size_t)blockIdx.x * HEADW_FLOATS + dst] = s;
    }
}

// Kernel 2: weight gradients  dW[row][c] = sum_px coef[px][row] * F_row(px, c),  rows = {dpt x F3, du[9] x F2, dm[3] x F1}.
// thread = (8 channels, pixel group); 104 accumulators; per-block partials in the HeadW layout.
template <int FM>
__global__ __launch_bounds__(256) void dam_head_wgrad_kernel(HeadFeat f1, HeadFeat f2, HeadFeat f3, const float *__restrict__ coef,
                                                             int N, int plane, float *__restrict__ partial) {
    __shared__ float s_sc[3][64], s_sh[3][64];
    __shared__ float s_w[4][13][64];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const HeadFeat *fs[3] = {&f1, &f2, &f3};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_sc[k][tid] = fs[k]->scale ? fs[k]->scale[tid] : 1.f;
            s_sh[k][tid] = fs[k]->scale ? fs[k]->shift[tid] : 0.f;
        }
    }
    __syncthreads();
    float gw[13][8];
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) gw[j][q] = 0.f;
    const int c8 = (tid & 7) * 8, pg = tid >> 3;
    const size_t total = (size_t)N * plane;
    for (size_t ip = (size_t)blockIdx.x * 32 + pg; ip < total; ip += (size_t)gridDim.x * 32) {
        const float4 *cr = reinterpret_cast<const float4 *>(coef + ip * 16);
        const float4 c0 = cr[0], c1 = cr[1], c2 = cr[2], c3 = cr[3];
        const float cf[13] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x};
        float x[8];
        FeatRaw8 R3, R2, R1;
        constexpr bool all16 = FM != 2;
        if (all16) { load_raw8<FM>(f3, ip, c8, R3); load_raw8<FM>(f2, ip, c8, R2); load_raw8<FM>(f1, ip, c8, R1); }
        if (all16) feat_from_raw8<FM>(f3, R3, c8, s_sc[2], s_sh[2], x);
        else feat8(f3, ip, c8, s_sc[2], s_sh[2], x);
#pragma unroll
        for (int q = 0; q < 8; ++q) gw[0][q] = fmaf(cf[0], x[q], gw[0][q]);
        if (all16) feat_from_raw8<FM>(f2, R2, c8, s_sc[1], s_sh[1], x);
        else feat8(f2, ip, c8, s_sc[1], s_sh[1], x);
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) gw[1 + j][q] = fmaf(cf[1 + j], x[q], gw[1 + j][q]);
        if (all16) feat_from_raw8<FM>(f1, R1, c8, s_sc[0], s_sh[0], x);
        else feat8(f1, ip, c8, s_sc[0], s_sh[0], x);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) gw[10 + j][q] = fmaf(cf[10 + j], x[q], gw[10 + j][q]);
    }
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float t = gw[j][q];
            t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
            gw[j][q] = t;
        }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 13; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) s_w[tid >> 6][j][c8 + q] = gw[j][q];
    }
    __syncthreads();
    float *o = partial + (size_t)blockIdx.x * HEADW_FLOATS;
    for (int idx = tid; idx < 13 * 64; idx += 256) {
        const int row = idx / 64, cc = idx % 64;
        o[idx] = (s_w[0][row][cc] + s_w[1][row][cc]) + (s_w[2][row][cc] + s_w[3][row][cc]);
    }
}

// ======================================================================================================
// Loss (train_util_dam.py:167-276) - two passes over the logits
// ======================================================================================================
// per-sample sums (lsums<ND>() floats; ND = number of direction classes, 5 / 9 / 17 - options.py:45 "4 8 16" + background):
//   0..2  I_c   sum p_c [label==c]      3..5  P_c   sum p_c          6..8  T_c   sum [label==c]
//   PW+i  Pw_i  sum w q_i               TW+j  Tw_j sum w t_j
//   SS+j  S[j][j]  SN+j  S[next(j)][j]  SP+j  S[prev(j)][j]   (S[i][j] = sum w q_i t_j, j = target class)
//   SC+0 ce  +1 dce  +2 mse
//   SC+3 tp  +4 fp  +5 fn   of the pixel-level metric (argmax direction == 1 vs direction label == 1, train_util_dam.py:279-281)
// For ND = 9 this is the 60-float layout the first version fixed (PW 9, TW 18, SS 27, SN 36, SP 45, SC 54).
template <int ND> struct LossLay {
    static constexpr int PW = 9, TW = 9 + ND, SS = 9 + 2 * ND, SN = 9 + 3 * ND, SP = 9 + 4 * ND, SC = 9 + 5 * ND, SUMS = SC + 6;
    // coefficient block per sample: dice alpha[3], beta[3]; wdice: bsum[ND], a_self[ND], a_next[ND], a_prev[ND]
    //   a_self[j]  multiplies row i=j,        a_next[j] row i=next(j),  a_prev[j] row i=prev(j)  when the pixel's target is j
    static constexpr int COEF = 6 + 4 * ND;
    static constexpr int TPB = ND > 9 ? 128 : 256;      // reduce kernel: SUMS x TPB floats of LDS (<= 64 KB)
};
template <int ND> __device__ __forceinline__ int dnext(int i) { return i == ND - 1 ? 1 : i + 1; }      // cyclic over 1..ND-1 (loss.py:231-258)
template <int ND> __device__ __forceinline__ int dprev(int i) { return i == 1 ? ND - 1 : i - 1; }

struct LossIn {
    const float *mask, *point, *dirn;        // f32 NCHW logits [B][3][P], [B][1][P], [B][ND][P]
    const unsigned char *label, *dirlab;     // u8 [B][P]
    const unsigned short *point_t;           // f16 [B][P]
    const unsigned char *weight;             // u8 [B][P]  (png weight map; /20 on the fly)
    const int *single;                       // [B]: 1 if the sample's direction map is constant (train_util_dam.py:133,141)
    int B, P;
    int quirk0;                              // mask the direction one-hot with SAMPLE 0's foreground (:139)
};

// per sample: is the direction label constant (the one-hot of a single class, train_util_dam.py:131-137)?  The same scan
// validates the label content: a mask class > 2 or a direction class >= nd would index past the per-class accumulators, so
// it raises *err (the finalize kernel then poisons every loss with NaN - the reference's NLLLoss fails loudly on such targets)
// and the accumulation kernels clamp their indices.
// one workgroup of 1024 threads per sample, 16 label bytes per thread and load (the first version walked them a byte at a time with
// 256 threads: 64 us on the step's critical chain for 2 MB)
__global__ __launch_bounds__(1024) void loss_single_kernel(const unsigned char *dirlab, const unsigned char *label, int P, int nd, int *single, int *err) {
    __shared__ int s_min[16], s_max[16], s_lmax[16];
    const unsigned char *d = dirlab + (size_t)blockIdx.x * P;
    const unsigned char *l = label + (size_t)blockIdx.x * P;
    int mn = 255, mx = 0, lm = 0;
    const int tid = threadIdx.x;
    const bool vec = (P % 16 == 0) && ((reinterpret_cast<size_t>(d) | reinterpret_cast<size_t>(l)) % 16 == 0);
    if (vec) {
        const uint4 *d4 = reinterpret_cast<const uint4 *>(d), *l4 = reinterpret_cast<const uint4 *>(l);
        for (int i = tid; i < P / 16; i += 1024) {
            const uint4 dv = d4[i], lv = l4[i];
            const unsigned dw[4] = {dv.x, dv.y, dv.z, dv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b8 = 0; b8 < 4; ++b8) {
                    const int v = (dw[k] >> (8 * b8)) & 0xff, w = (lw[k] >> (8 * b8)) & 0xff;
                    mn = v < mn ? v : mn; mx = v > mx ? v : mx; lm = w > lm ? w : lm;
                }
        }
    } else {
        for (int i = tid; i < P; i += 1024) {
            int v = d[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
            v = l[i]; lm = v > lm ? v : lm;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int a = __shfl_xor(mn, m), b2 = __shfl_xor(mx, m), c = __shfl_xor(lm, m);
        mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx; lm = c > lm ? c : lm;
    }
    if ((tid & 63) == 0) { s_min[tid >> 6] = mn; s_max[tid >> 6] = mx; s_lmax[tid >> 6] = lm; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) {
            mn = s_min[i] < mn ? s_min[i] : mn; mx = s_max[i] > mx ? s_max[i] : mx; lm = s_lmax[i] > lm ? s_lmax[i] : lm;
        }
        single[blockIdx.x] = (mn == mx) ? 1 : 0;
        if (mx > nd - 1 || lm > 2) atomicOr(err, 1);
    }
}

template <int NC>
__device__ __forceinline__ void softmax_n(const float *l, float *p, float *logp) {
    float m = l[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) m = fmaxf(m, l[c]);
    float e[NC], s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { e[c] = expf(l[c] - m); s += e[c]; }
    const float ls = logf(s);
#pragma unroll
    for (int c = 0; c < NC; ++c) { p[c] = e[c] / s; logp[c] = l[c] - m - ls; }
}
__device__ __forceinline__ void softmax3(const float *l, float *p, float *logp) {
    const float m = fmaxf(l[0], fmaxf(l[1], l[2]));
    float e[3], s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { e[c] = expf(l[c] - m); s += e[c]; }
    const float ls = logf(s);
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = e[c] / s; logp[c] = l[c] - m - ls; }
}
__device__ __forceinline__ void softmax9(const float *l, float *p, float *logp) { softmax_n<9>(l, p, logp); }

// target class of the weighted dice for pixel i of sample b: -1 = all one-hot channels are zero
template <int ND>
__device__ __forceinline__ int dice_target(const LossIn &L, int b, int i) {
    if (L.single[b]) return 0;
    int t = L.dirlab[(size_t)b * L.P + i];
    t = t > ND - 1 ? ND - 1 : t;
    if (L.quirk0) return L.label[i] != 0 ? t : -1;               // sample 0's label
    return L.label[(size_t)b * L.P + i] != 0 ? t : -1;
}

// grid (chunks, B); private accumulators live in LDS ([k][tid]) because several are indexed by the target class
template <int ND>
__global__ __launch_bounds__(LossLay<ND>::TPB) void loss_reduce_kernel(LossIn L, float *__restrict__ partial) {
    using Y = LossLay<ND>;
    constexpr int TPB = Y::TPB;
    __shared__ float acc[Y::SUMS][TPB];
    const int tid = threadIdx.x, b = blockIdx.y;
#pragma unroll
    for (int k = 0; k < Y::SUMS; ++k) acc[k][tid] = 0.f;
    const size_t ob = (size_t)b * L.P;
    for (int i = blockIdx.x * TPB + tid; i < L.P; i += gridDim.x * TPB) {
        float l3[3], p3[3], lp3[3], l9[ND], p9[ND], lp9[ND];
#pragma unroll
        for (int c = 0; c < 3; ++c) l3[c] = L.mask[((size_t)b * 3 + c) * L.P + i];
#pragma unroll
        for (int c = 0; c < ND; ++c) l9[c] = L.dirn[((size_t)b * ND + c) * L.P + i];
        softmax3(l3, p3, lp3);
        softmax_n<ND>(l9, p9, lp9);
        const float w = (float)L.weight[ob + i] / 20.f;
        int lab = L.label[ob + i], dl = L.dirlab[ob + i];
        lab = lab > 2 ? 2 : lab; dl = dl > ND - 1 ? ND - 1 : dl;   // (out-of-range content is reported through *err, see loss_single_kernel)
        acc[lab][tid] += p3[lab];
        acc[6 + lab][tid] += 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[3 + c][tid] += p3[c];
#pragma unroll
        for (int c = 0; c < ND; ++c) acc[Y::PW + c][tid] = fmaf(w, p9[c], acc[Y::PW + c][tid]);
        const int t = dice_target<ND>(L, b, i);
        if (t >= 0) {
            acc[Y::TW + t][tid] += w;
            acc[Y::SS + t][tid] = fmaf(w, p9[t], acc[Y::SS + t][tid]);
            if (t >= 1) {
                acc[Y::SN + t][tid] = fmaf(w, p9[dnext<ND>(t)], acc[Y::SN + t][tid]);
                acc[Y::SP + t][tid] = fmaf(w, p9[dprev<ND>(t)], acc[Y::SP + t][tid]);
            }
        }
        acc[Y::SC][tid] -= lp3[lab] * w;
        acc[Y::SC + 1][tid] -= lp9[dl] * w;
        const float dpt = L.point[ob + i] - h2f(L.point_t[ob + i]);
        acc[Y::SC + 2][tid] = fmaf(dpt, dpt, acc[Y::SC + 2][tid]);
        {   // np.argmax over the direction classes (first maximum), "inside" = class 1 (utils.py:76-78)
            int am = 0;
            float best = l9[0];
#pragma unroll
            for (int c = 1; c < ND; ++c) if (l9[c] > best) { best = l9[c]; am = c; }
            const bool pi = am == 1, ti = dl == 1;
            if (pi && ti) acc[Y::SC + 3][tid] += 1.f;
            if (pi && !ti) acc[Y::SC + 4][tid] += 1.f;
            if (!pi && ti) acc[Y::SC + 5][tid] += 1.f;
        }
    }
    __syncthreads();
    if (tid < Y::SUMS) {
        float s = 0.f;
        for (int k = 0; k < TPB; ++k) s += acc[tid][k];
        partial[((size_t)b * gridDim.x + blockIdx.x) * Y::SUMS + tid] = s;
    }
}

// single block: per-sample sums -> loss terms (5 + total) and the pass-2 coefficients
template <int ND>
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float *__restrict__ partial, int nchunk, int B, int P,
                                                            float *__restrict__ sums, float *__restrict__ coef,
                                                            float *__restrict__ losses, const int *__restrict__ err) {
    using Y = LossLay<ND>;
    constexpr int NS = Y::SUMS;
    __shared__ float s_sum[64 * NS];      // B <= 64
    const int tid = threadIdx.x;
    for (int idx = tid; idx < B * NS; idx += 256) {
        const int b = idx / NS, k = idx % NS;
        // four independent chains of loads (a single dependent chain of nchunk L2 round trips dominated this kernel)
        const float *pp = partial + (size_t)b * nchunk * NS + k;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int ch = 0;
        for (; ch + 3 < nchunk; ch += 4) {
            s0 += pp[(size_t)ch * NS]; s1 += pp[(size_t)(ch + 1) * NS];
            s2 += pp[(size_t)(ch + 2) * NS]; s3 += pp[(size_t)(ch + 3) * NS];
        }
        for (; ch < nchunk; ++ch) s0 += pp[(size_t)ch * NS];
        const float s = (s0 + s1) + (s2 + s3);
        s_sum[idx] = s;
        if (sums) sums[idx] = s;
    }
    __syncthreads();
    const float fB = (float)B;
    for (int b = tid; b < B; b += 256) {
        const float *S = s_sum + b * NS;
        float *cf = coef + (size_t)b * Y::COEF;
        for (int c = 0; c < 3; ++c) {
            const float I = S[c], U = S[3 + c] + S[6 + c];
            cf[c] = -2.f / (fB * (U + 1.f));
            cf[3 + c] = 2.f * (I + 1.f) / (fB * (U + 1.f) * (U + 1.f));
        }
        // row i, column j terms: alpha_ij = -2/(B (U_ij+1)), beta_ij = 2 (S_ij+1)/(B (U_ij+1)^2), U_ij = Pw_i + Tw_j
        float bsum[ND];
        for (int i = 0; i < ND; ++i) bsum[i] = 0.f;
        for (int j = 0; j < ND; ++j) {
            {   // (i=j, j)
                const float U = S[Y::PW + j] + S[Y::TW + j], Sij = S[Y::SS + j], m = j == 0 ? 2.f : 1.f;
                cf[6 + ND + j] = m * -2.f / (fB * (U + 1.f));
                bsum[j] += m * 2.f * (Sij + 1.f) / (fB * (U + 1.f) * (U + 1.f));
            }
            if (j >= 1) {
                const int in = dnext<ND>(j), ip = dprev<ND>(j);
                {   const float U = S[Y::PW + in] + S[Y::TW + j], Sij = S[Y::SN + j];
                    cf[6 + 2 * ND + j] = -2.f / (fB * (U + 1.f));
                    bsum[in] += 2.f * (Sij + 1.f) / (fB * (U + 1.f) * (U + 1.f)); }
                {   const float U = S[Y::PW + ip] + S[Y::TW + j], Sij = S[Y::SP + j];
                    cf[6 + 3 * ND + j] = -2.f / (fB * (U + 1.f));
                    bsum[ip] += 2.f * (Sij + 1.f) / (fB * (U + 1.f) * (U + 1.f)); }
            } else { cf[6 + 2 * ND] = 0.f; cf[6 + 3 * ND] = 0.f; }
        }
        for (int i = 0; i < ND; ++i) cf[6 + i] = bsum[i];
    }
    // the 3 ND + 1 batch-mean dice ratios, one thread each (they were ~450 serial divisions on thread 0):
    // [0,3) mask dice c; [3,3+ND) wdice(i,i); then wdice(i,prev(i)), i=1..ND-1; then wdice(i,next(i)), i=1..ND-1
    constexpr int T_PREV = 3 + ND, T_NEXT = T_PREV + ND - 1, NT = T_NEXT + ND - 1;
    __shared__ float s_term[NT];
    if (tid < NT) {
        int num, da, db;                      // mean_b 2 (S[num] + 1) / (S[da] + S[db] + 1)
        if (tid < 3) { num = tid; da = 3 + tid; db = 6 + tid; }
        else if (tid < T_PREV) { const int i = tid - 3; num = Y::SS + i; da = Y::PW + i; db = Y::TW + i; }
        else if (tid < T_NEXT) { const int i = tid - T_PREV + 1, j = dprev<ND>(i); num = Y::SN + j; da = Y::PW + i; db = Y::TW + j; }
        else { const int i = tid - T_NEXT + 1, j = dnext<ND>(i); num = Y::SP + j; da = Y::PW + i; db = Y::TW + j; }
        float acc = 0.f;
        for (int b = 0; b < B; ++b) { const float *S = s_sum + b * NS; acc += 2.f * (S[num] + 1.f) / (S[da] + S[db] + 1.f); }
        s_term[tid] = 1.f - acc / fB;
    }
    __syncthreads();
    if (tid == 0) {
        const float n = (float)B * (float)P;
        float ce = 0.f, dce = 0.f, mse = 0.f;
        for (int b = 0; b < B; ++b) { ce += s_sum[b * NS + Y::SC]; dce += s_sum[b * NS + Y::SC + 1]; mse += s_sum[b * NS + Y::SC + 2]; }
        ce /= n; dce /= n; mse /= n;
        float dice = 0.f;
        for (int c = 0; c < 3; ++c) dice += s_term[c];
        float wd = 0.f;
        for (int i = 0; i < ND; ++i) {
            if (i == 0) wd += 2.f * s_term[3];
            else wd += s_term[3 + i] - (1.f - s_term[T_PREV + i - 1]) - (1.f - s_term[T_NEXT + i - 1]);
        }
        wd /= (float)ND;
        losses[0] = ce + dice + dce + wd + mse;
        losses[1] = dce; losses[2] = wd; losses[3] = mse; losses[4] = ce; losses[5] = dice;
        // pixel-level metrics, mean over the samples (utils.py:67-110): accuracy, IoU, recall, precision, F1
        double m[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < B; ++b) {
            const double tp = s_sum[b * NS + Y::SC + 3], fp = s_sum[b * NS + Y::SC + 4], fn = s_sum[b * NS + Y::SC + 5];
            const double tn = (double)P - tp - fp - fn;
            const double precision = tp / (tp + fp + 1e-10), recall = tp / (tp + fn + 1e-10);
            m[0] += (tp + tn) / (tp + fp + tn + fn + 1e-10);
            m[1] += tp / (tp + fp + fn + 1e-10);
            m[2] += recall;
            m[3] += precision;
            m[4] += 2 * precision * recall / (precision + recall + 1e-10);
        }
        for (int k = 0; k < 5; ++k) losses[6 + k] = (float)(m[k] / B);
        if (*err) {                          // label content out of range: no silent garbage
            for (int k = 0; k < 11; ++k) losses[k] = __builtin_nanf("");
        }
    }
}

// pass 2: gradients w.r.t. the logits (f32 NCHW, same layout as the logits)
template <int ND>
__global__ __launch_bounds__(256) void loss_grad_kernel(LossIn L, const float *__restrict__ coef, float *__restrict__ dmask,
                                                        float *__restrict__ dpoint, float *__restrict__ ddir) {
    const int b = blockIdx.y;
    const float *cf = coef + (size_t)b * LossLay<ND>::COEF;
    const float inv_n = 1.f / ((float)L.B * (float)L.P);
    const size_t ob = (size_t)b * L.P;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L.P; i += gridDim.x * 256) {
        float l3[3], p3[3], lp3[3], l9[ND], p9[ND], lp9[ND];
#pragma unroll
        for (int c = 0; c < 3; ++c) l3[c] = L.mask[((size_t)b * 3 + c) * L.P + i];
#pragma unroll
        for (int c = 0; c < ND; ++c) l9[c] = L.dirn[((size_t)b * ND + c) * L.P + i];
        softmax3(l3, p3, lp3);
        softmax_n<ND>(l9, p9, lp9);
        const float w = (float)L.weight[ob + i] / 20.f;
        int lab = L.label[ob + i], dl = L.dirlab[ob + i];
        lab = lab > 2 ? 2 : lab; dl = dl > ND - 1 ? ND - 1 : dl;   // (out-of-range content is reported through *err, see loss_single_kernel)
        // mask: dice gradient w.r.t. probabilities, through the softmax, plus the weighted CE
        float gp[3], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { gp[c] = cf[3 + c] + (c == lab ? cf[c] : 0.f); dot = fmaf(p3[c], gp[c], dot); }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dmask[((size_t)b * 3 + c) * L.P + i] = p3[c] * (gp[c] - dot) + w * inv_n * (p3[c] - (c == lab ? 1.f : 0.f));
        // direction: weighted cyclic dice (average over the ND classes) + weighted CE
        const int t = dice_target<ND>(L, b, i);
        float gq[ND], dotq = 0.f;
#pragma unroll
        for (int c = 0; c < ND; ++c) gq[c] = cf[6 + c];
        if (t >= 0) {
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                float a = 0.f;
                if (c == t) a = cf[6 + ND + t];
                else if (t >= 1 && c == dnext<ND>(t)) a = cf[6 + 2 * ND + t];
                else if (t >= 1 && c == dprev<ND>(t)) a = cf[6 + 3 * ND + t];
                gq[c] += a;
            }
        }
#pragma unroll
        for (int c = 0; c < ND; ++c) { gq[c] *= w * (1.f / (float)ND); dotq = fmaf(p9[c], gq[c], dotq); }
#pragma unroll
        for (int c = 0; c < ND; ++c)
            ddir[((size_t)b * ND + c) * L.P + i] = p9[c] * (gq[c] - dotq) + w * inv_n * (p9[c] - (c == dl ? 1.f : 0.f));
        if (dpoint) dpoint[ob + i] = 2.f * inv_n * (L.point[ob + i] - h2f(L.point_t[ob + i]));
    }
}

// ======================================================================================================
// Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias correction)
// ======================================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
    const float step = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        gi = fmaf(wd, pi, gi);
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - b1);
        vi = vi * b2 + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step * (mi / denom);
    }
}


// ======================================================================================================
// validate() loss mix (train_util_dam.py:499-580) - per-sample sums, one pass over the logits.  The host combines them:
//   0..2 I_c = sum p_c [label==c]   3..5 P_c = sum p_c   6..8 T_c = sum [label==c]   9 sum -log p_label (UNweighted, :499-505)
//   10..18 Iq_i = sum q'_i t_i   19..27 Pq_i = sum q'_i   28..36 Tq_i = sum t_i   with q' = softmax(direction), q'_0 *= p_0 (:564-566)
//            and t = one-hot of the direction class RANK (lut) masked by SAMPLE 0's foreground (:463-470)
//   37 sum w * -log q_dir (:553-559)   38 sum (point - target / 255)^2 (:575-580)
//   39 tp  40 fp  41 fn  of (argmax mask == 1) vs (label == 1)  (utils.accuracy_pixel_level, :585-590)
// ======================================================================================================
// For ND direction classes (5 / 9 / 17) the three direction blocks are ND wide: IQ = 10, PQ = 10 + ND, TQ = 10 + 2 ND, then the five
// scalars at VS = 10 + 3 ND (42 floats for ND = 9, the layout above).
template <int ND> struct ValLay {
    static constexpr int IQ = 10, PQ = 10 + ND, TQ = 10 + 2 * ND, VS = 10 + 3 * ND, SUMS = VS + 5;
    static constexpr int TPB = ND > 9 ? 128 : 256;
};
static_assert(ValLay<9>::SUMS == CDNET_VAL_SUMS, "cdnet_dam_val_sums row layout");

struct ValIn {
    const float *mask, *point, *dirn;
    const unsigned char *label, *dirlab, *weight;
    const unsigned short *point_t;
    int lut[17];                             // direction class value -> channel (rank among the batch's unique values), -1 = absent
    int B, P;
};

template <int ND>
__global__ __launch_bounds__(ValLay<ND>::TPB) void val_sums_kernel(ValIn L, float *__restrict__ partial) {
    using Y = ValLay<ND>;
    constexpr int TPB = Y::TPB;
    __shared__ float acc[Y::SUMS][TPB];
    const int tid = threadIdx.x, b = blockIdx.y;
#pragma unroll
    for (int k = 0; k < Y::SUMS; ++k) acc[k][tid] = 0.f;
    const size_t ob = (size_t)b * L.P;
    for (int i = blockIdx.x * TPB + tid; i < L.P; i += gridDim.x * TPB) {
        float l3[3], p3[3], lp3[3], l9[ND], p9[ND], lp9[ND];
#pragma unroll
        for (int c = 0; c < 3; ++c) l3[c] = L.mask[((size_t)b * 3 + c) * L.P + i];
#pragma unroll
        for (int c = 0; c < ND; ++c) l9[c] = L.dirn[((size_t)b * ND + c) * L.P + i];
        softmax3(l3, p3, lp3);
        softmax_n<ND>(l9, p9, lp9);
        int lab = L.label[ob + i], dl = L.dirlab[ob + i];
        lab = lab > 2 ? 2 : lab; dl = dl > ND - 1 ? ND - 1 : dl;
        const float w = (float)L.weight[ob + i] / 20.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[3 + c][tid] += p3[c];
            if (c == lab) { acc[c][tid] += p3[c]; acc[6 + c][tid] += 1.f; }
        }
        acc[9][tid] -= lp3[lab];
        p9[0] *= p3[0];
        const bool fg0 = L.label[i] != 0;                     // sample 0's foreground (the reference indexes target[0])
        const int tch = (fg0 && L.lut[dl] >= 0) ? L.lut[dl] : -1;
#pragma unroll
        for (int c = 0; c < ND; ++c) {
            acc[Y::PQ + c][tid] += p9[c];
            if (c == tch) { acc[Y::IQ + c][tid] += p9[c]; acc[Y::TQ + c][tid] += 1.f; }
        }
        acc[Y::VS][tid] -= w * lp9[dl];
        const float dpt = L.point[ob + i] - h2f(L.point_t[ob + i]) / 255.f;
        acc[Y::VS + 1][tid] = fmaf(dpt, dpt, acc[Y::VS + 1][tid]);
        int am = 0;                                            // np.argmax: first maximum
        if (l3[1] > l3[am]) am = 1;
        if (l3[2] > l3[am]) am = 2;
        const bool pi = am == 1, ti = lab == 1;
        if (pi && ti) acc[Y::VS + 2][tid] += 1.f;
        if (pi && !ti) acc[Y::VS + 3][tid] += 1.f;
        if (!pi && ti) acc[Y::VS + 4][tid] += 1.f;
    }
    __syncthreads();
    if (tid < Y::SUMS) {
        float s = 0.f;
        for (int k = 0; k < TPB; ++k) s += acc[tid][k];
        partial[((size_t)b * gridDim.x + blockIdx.x) * Y::SUMS + tid] = s;
    }
}

// sums[b][k] = sum over chunks, fixed order
__global__ void val_sums_reduce_kernel(const float *__restrict__ partial, int nchunk, int B, int nsums, float *__restrict__ sums) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * nsums) return;
    const int b = idx / nsums, k = idx % nsums;
    double s = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) s += (double)partial[((size_t)b * nchunk + ch) * nsums + k];
    sums[idx] = (float)s;
}

// ======================================================================================================
// Backward of the plain UNet's 64 -> K classifier (models/unet.py:75,104).  Four lanes per pixel (16 channels each):
//   dF[c] = sum_k dlogit_k * w[k][c];  dW[k][c] = sum_px dlogit_k * F[c];  db[k] = sum_px dlogit_k.
// Per-block partial sums [K*64 | K], reduced in a fixed order by reduce_partials_kernel (deterministic).
// ======================================================================================================
constexpr int FC_KMAX = 4;            // plain UNet classifier (3 classes): 4 lanes per pixel x 16 channels
constexpr int FC_KWIDE = 12;          // the ablation heads' 9-class direction classifier: 8 lanes per pixel x 8 channels
constexpr int FC_KMOST = 20;          // model_unet_MandD16's 17-class direction classifier: same 8 x 8 split, 20 accumulator rows

// KM: classes the instantiation holds accumulators for; LPP lanes share a pixel (64 / LPP channels each)
template <int KM, int LPP>
__global__ __launch_bounds__(256) void final_conv_bwd_kernel(HeadFeat f, const float *__restrict__ w, const float *__restrict__ dl,
                                                             int K, int N, int plane, unsigned short *__restrict__ df,
                                                             float *__restrict__ partial) {
    constexpr int CH = 64 / LPP, PPB = 256 / LPP, ROW = KM * 64 + KM;
    static_assert(CH == 16 || CH == 8, "16 or 8 channels per lane");
    __shared__ float s_w[KM * 64], s_sc[64], s_sh[64];
    __shared__ float s_red[4][ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * 64; i += 256) s_w[i] = w[i];
    if (tid < 64) { s_sc[tid] = f.scale ? f.scale[tid] : 1.f; s_sh[tid] = f.scale ? f.shift[tid] : 0.f; }
    __syncthreads();
    const int q = tid % LPP;
    float gw[KM][CH], gb[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        gb[k] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) gw[k][c] = 0.f;
    }
    const size_t total = (size_t)N * plane;
    for (size_t base = (size_t)blockIdx.x * PPB; base < total; base += (size_t)gridDim.x * PPB) {
        const size_t i = base + (tid / LPP);
        const bool ok = i < total;
        const size_t ii = ok ? i : total - 1;
        const size_t n = ii / plane, p = ii - n * plane;
        float v[CH], d[KM];
        if (CH == 16) feat16(f, ii, q, s_sc, s_sh, v);
        else feat8(f, ii, q * 8, s_sc, s_sh, v);
#pragma unroll
        for (int k = 0; k < KM; ++k) d[k] = (ok && k < K) ? dl[(n * K + k) * plane + p] : 0.f;
        float o[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                t = fmaf(d[k], s_w[(k < K ? k : 0) * 64 + q * CH + c], t);
                gw[k][c] = fmaf(d[k], v[c], gw[k][c]);
            }
            o[c] = t;
        }
#pragma unroll
        for (int k = 0; k < KM; ++k) gb[k] += d[k];
        if (ok) {
            if (CH == 16) store16_grad(df, ii * 64 + q * 16, o, f.f16 == 2);
            else if (f.f16 == 2) stf8(df, ii * 64 + q * 8, o);
            else {
                V16 ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) ov.h[j] = f2bf(o[j]);
                *reinterpret_cast<uint4 *>(df + ii * 64 + q * 8) = ov.u;
            }
        }
    }
    // lanes with the same q own the same channels: butterfly over the pixel slots of the wave, then one row per wave
#pragma unroll
    for (int k = 0; k < KM; ++k) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float t = gw[k][c];
#pragma unroll
            for (int m = LPP; m < 64; m <<= 1) t += __shfl_xor(t, m);
            if (lane < LPP) s_red[wave][k * 64 + lane * CH + c] = t;
        }
        float t = gb[k];
#pragma unroll
        for (int m = LPP; m < 64; m <<= 1) t += __shfl_xor(t, m);
        if (lane == 0) s_red[wave][KM * 64 + k] = t;
    }
    __syncthreads();
    for (int j = tid; j < ROW; j += 256)
        partial[(size_t)blockIdx.x * ROW + j] = (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]);
}

__global__ void final_conv_scatter_kernel(const float *__restrict__ sums, int K, int KM, float *__restrict__ dw, float *__restrict__ db) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t < K * 64) dw[t] = sums[t];
    if (t < K) db[t] = sums[KM * 64 + t];
}


// bias gradient of a BatchNorm-less convolution (the plain UNet's ConvTranspose2d, models/unet.py:30):
// db[c] = sum over pixels of the bf16 NHWC output gradient.  thread = (8 channels, pixel group); per-block partials.
template <bool F32>
__global__ __launch_bounds__(256) void bias_grad_kernel(const unsigned short *__restrict__ g, unsigned npix, int C,
                                                        float *__restrict__ partial) {
    __shared__ float s_red[256][9];
    const int VPP = C / 8, tid = threadIdx.x;
    const int slot = tid % VPP;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const unsigned ppb = 256 / VPP;
    for (unsigned p = first_pixel(ppb, VPP); p < npix; p += gridDim.x * ppb) {
        if (F32) {
            float t[8];
            ldf8(g, (size_t)p * C + slot * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t[j];
        } else {
            V16 v;
            v.u = *reinterpret_cast<const uint4 *>(g + (size_t)p * C + slot * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v.h[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[tid][j] = acc[j];
    __syncthreads();
    for (int q = tid; q < C; q += 256) {
        const int sl = q / 8, j = q % 8;
        float t = 0.f;
        for (int k = sl; k < 256; k += VPP) t += s_red[k][j];
        partial[(size_t)blockIdx.x * C + q] = t;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
static int fill_bn_args(const cdnet_bn_bwd_args *a, BnBwdArgs &A, const char *who) {
    CDNET_REQUIRE(a && a->raw, "%s: null pointer", who);
    CDNET_REQUIRE(a->C % 8 == 0 && a->C >= 8 && a->C <= 2048 , "%s: C=%d unsupported", who, a->C);
    CDNET_REQUIRE(a->ngin >= 1 && a->ngin <= 3, "%s: ngin=%d", who, a->ngin);
    A.raw = a->raw; A.res = a->res; A.f16 = a->f16; A.scale = a->scale; A.shift = a->shift; A.relu = a->relu;
    A.mean = a->mean; A.invstd = a->invstd;
    for (int k = 0; k < 3; ++k) {
        A.gin[k].g = a->gin[k].g; A.gin[k].Hg = a->gin[k].Hg; A.gin[k].Wg = a->gin[k].Wg;
        A.gin[k].oy = a->gin[k].oy; A.gin[k].ox = a->gin[k].ox; A.gin[k].pooled = a->gin[k].pooled;
        A.gin[k].coff = a->gin[k].coff; A.gin[k].cstride = a->gin[k].cstride ? a->gin[k].cstride : a->C;
        if (k < a->ngin) CDNET_REQUIRE(a->gin[k].g, "%s: null gradient input %d", who, k);
    }
    A.ngin = a->ngin; A.N = a->N; A.H = a->H; A.W = a->W; A.C = a->C;
    A.partial = nullptr; A.k1 = A.k2 = A.k3 = nullptr; A.draw = nullptr; A.dz_out = nullptr;
    return CDNET_OK;
}

extern "C" int cdnet_bn_backward(const cdnet_bn_bwd_args *a, const float *gamma, float *dgamma, float *dbeta, float *workspace,
                                 size_t workspace_floats, uint16_t *draw, uint16_t *dz_out, void *stream) {
    BnBwdArgs A;
    int rc = fill_bn_args(a, A, "cdnet_bn_backward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)A.N * A.H * A.W;
    CDNET_REQUIRE(npix * (size_t)A.C < ((size_t)1 << 32) && npix < ((size_t)1 << 31), "cdnet_bn_backward: tensor too large for 32-bit pixel indexing");
    const int ppb = 256 / (A.C / 8);
    int nb = (int)((npix + (size_t)ppb * BN_U - 1) / ((size_t)ppb * BN_U));
    if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
    if (nb < 1) nb = 1;
    bool simple = true, window = false;               // every gradient source a same-size, un-shifted tensor?
    int npool = 0, kp = 0, nflat = 0;
    for (int k = 0; k < A.ngin; ++k) {
        const GradIn &g = A.gin[k];
        const bool same = !g.pooled && g.oy == 0 && g.ox == 0 && g.Hg == A.H && g.Wg == A.W;
        simple = simple && same;
        if (g.pooled) { ++npool; kp = k; }
        else if (same) ++nflat;
    }
    // window path: exactly one pooled consumer, every other one flat, a BatchNorm + ReLU layer without residual branch
    window = npool == 1 && nflat == A.ngin - 1 && nflat <= 2 && A.mean && A.scale && !A.res && draw && !dz_out;
    if (window) {
        const size_t nwin = (size_t)A.N * ((A.H + 1) / 2) * ((A.W + 1) / 2);
        nb = (int)((nwin + ppb - 1) / ppb);
        if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
    }
    A.draw = draw; A.dz_out = dz_out;
    static const bool dz_reuse = !(getenv("CDNET_BN_DZ_REUSE") && atoi(getenv("CDNET_BN_DZ_REUSE")) == 0);
    if (!dz_reuse) A.dz_out = nullptr;                // (the sums pass stores dz only when the second pass is going to read it)
    // The apply pass re-reads what the reduce pass just streamed (raw + gradients, up to 2 x 134 MB against 256 MB of
    // Infinity Cache): walking it back to front meets the most recently cached lines first instead of chasing the LRU tail.
    A.rev = 1;
    bool flat = !window && simple && A.mean && A.scale && draw && ((A.res != nullptr) == (dz_out != nullptr));
    CDNET_REQUIRE(A.relu != 2 || (flat && A.res), "cdnet_bn_backward: relu = 2 (mask from the stored output) needs same-size gradient sources and res");
    const bool f32 = A.f16 == 2;                      // fp32 tensors: flat32 kernels (same-size sources) or the generic ones (pool / pad routing)
    const bool flat32 = f32 && flat && A.C <= 1024;
    const bool window32 = f32 && window && A.C % 4 == 0 && A.C <= 1024 && A.shift && A.invstd;
    if (window32) {
        const int ppb4 = 256 / (A.C / 4);
        const size_t nwin = (size_t)A.N * ((A.H + 1) / 2) * ((A.W + 1) / 2);
        nb = (int)((nwin + ppb4 - 1) / ppb4);
        if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
    }
    if (f32) { window = false; flat = false; }
    if (flat32) {
        const int ppb4 = 256 / (A.C / 4);
        nb = (int)((npix + (size_t)ppb4 * BN_U - 1) / ((size_t)ppb4 * BN_U));
        if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
        if (nb < 1) nb = 1;
    }
    if (A.mean) {
        CDNET_REQUIRE(gamma && A.invstd && workspace, "cdnet_bn_backward: BatchNorm layer needs gamma/invstd/workspace");
        const size_t need = (size_t)nb * 2 * A.C + 3 * (size_t)A.C;
        if (workspace_floats < need) { set_error("cdnet_bn_backward: workspace %zu < %zu floats", workspace_floats, need); return CDNET_E_WORKSPACE; }
        A.partial = workspace;
        float *k = workspace + (size_t)nb * 2 * A.C;
        if (window) {
            if (nflat == 0) bn_bwd_window_kernel<0, false><<<nb, 256, 0, st>>>(A, kp);
            else if (nflat == 1) bn_bwd_window_kernel<1, false><<<nb, 256, 0, st>>>(A, kp);
            else bn_bwd_window_kernel<2, false><<<nb, 256, 0, st>>>(A, kp);
        } else if (flat) {
            switch (A.ngin * 2 + (A.res ? 1 : 0)) {
                case 2: bn_bwd_reduce_flat_kernel<1, false><<<nb, 256, 0, st>>>(A); break;
                case 3: bn_bwd_reduce_flat_kernel<1, true><<<nb, 256, 0, st>>>(A); break;
                case 4: bn_bwd_reduce_flat_kernel<2, false><<<nb, 256, 0, st>>>(A); break;
                case 5: bn_bwd_reduce_flat_kernel<2, true><<<nb, 256, 0, st>>>(A); break;
                case 6: bn_bwd_reduce_flat_kernel<3, false><<<nb, 256, 0, st>>>(A); break;
                default: bn_bwd_reduce_flat_kernel<3, true><<<nb, 256, 0, st>>>(A); break;
            }
        } else if (window32) launch_window32<false>(A, nflat, kp, nb, st);
        else if (flat32) launch_flat32<false>(A, nb, st);
        else if (f32) bn_bwd_reduce_kernel<true><<<nb, 256, 0, st>>>(A);
        else bn_bwd_reduce_kernel<false><<<nb, 256, 0, st>>>(A);
        bn_bwd_finalize_kernel<<<A.C, 256, 0, st>>>(A.partial, nb, A.C, (float)npix, gamma, A.invstd, dgamma, dbeta, k,
                                                              k + A.C, k + 2 * A.C);
        A.k1 = k; A.k2 = k + A.C; A.k3 = k + 2 * A.C;
    }
    A.dz_out = dz_out;
    if (window) {
        if (nflat == 0) bn_bwd_window_kernel<0, true><<<nb, 256, 0, st>>>(A, kp);
        else if (nflat == 1) bn_bwd_window_kernel<1, true><<<nb, 256, 0, st>>>(A, kp);
        else bn_bwd_window_kernel<2, true><<<nb, 256, 0, st>>>(A, kp);
    } else if (flat && dz_reuse && A.mean && A.res && dz_out) {
        BnBwdArgs B = A;
        B.ngin = 1;
        B.gin[0].g = dz_out; B.gin[0].Hg = A.H; B.gin[0].Wg = A.W; B.gin[0].oy = 0; B.gin[0].ox = 0; B.gin[0].pooled = 0;
        B.gin[0].coff = 0; B.gin[0].cstride = A.C;
        B.res = nullptr; B.relu = 0; B.dz_out = nullptr;
        bn_bwd_apply_flat_kernel<1, false><<<nb, 256, 0, st>>>(B);
    } else if (flat) {
        switch (A.ngin * 2 + (A.res ? 1 : 0)) {
            case 2: bn_bwd_apply_flat_kernel<1, false><<<nb, 256, 0, st>>>(A); break;
            case 3: bn_bwd_apply_flat_kernel<1, true><<<nb, 256, 0, st>>>(A); break;
            case 4: bn_bwd_apply_flat_kernel<2, false><<<nb, 256, 0, st>>>(A); break;
            case 5: bn_bwd_apply_flat_kernel<2, true><<<nb, 256, 0, st>>>(A); break;
            case 6: bn_bwd_apply_flat_kernel<3, false><<<nb, 256, 0, st>>>(A); break;
            default: bn_bwd_apply_flat_kernel<3, true><<<nb, 256, 0, st>>>(A); break;
        }
    } else if (window32) launch_window32<true>(A, nflat, kp, nb, st);
    else if (flat32 && A.mean && A.res && dz_out && dz_reuse) {
        // (the sums pass stored dz: one plain source, no mask, no second dz store - bit-identical, 9 instead of 12 tensor passes)
        BnBwdArgs B = A;
        B.ngin = 1;
        B.gin[0].g = dz_out; B.gin[0].Hg = A.H; B.gin[0].Wg = A.W; B.gin[0].oy = 0; B.gin[0].ox = 0; B.gin[0].pooled = 0;
        B.gin[0].coff = 0; B.gin[0].cstride = A.C;
        B.res = nullptr; B.relu = 0; B.dz_out = nullptr;
        launch_flat32<true>(B, nb, st);
    } else if (flat32) launch_flat32<true>(A, nb, st);
    else if (f32) bn_bwd_apply_kernel<true><<<nb, 256, 0, st>>>(A);
    else bn_bwd_apply_kernel<false><<<nb, 256, 0, st>>>(A);
    return check_launch("cdnet_bn_backward");
}

// The two passes of cdnet_bn_backward as separate calls, for the plain case (one same-size gradient source, BatchNorm + ReLU, no
// residual, 16-bit tensors): `stats` = reduce + finalize and the [7][C] table scale | shift | mean | invstd | k1 | k2 | k3 that both
// the apply pass and a fused consumer (cdnet_conv_src.relu = 3) read; `apply` = the second pass alone.  The trainer runs `stats`
// on the main chain, backward-data with the fused source right behind it, and `apply` + the weight gradient on the side stream.
static bool bn_plain_case(const BnBwdArgs &A) {
    const GradIn &g = A.gin[0];
    return A.ngin == 1 && !g.pooled && g.oy == 0 && g.ox == 0 && g.Hg == A.H && g.Wg == A.W && A.mean && A.scale && A.shift && A.invstd && !A.res &&
           A.relu == 1 && (g.cstride == 0 || g.cstride == A.C) && g.coff == 0;
}

__global__ void bn_ktab_copy_kernel(const float *scale, const float *shift, const float *mean, const float *invstd, int C, float *ktab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { ktab[c] = scale[c]; ktab[C + c] = shift[c]; ktab[2 * C + c] = mean[c]; ktab[3 * C + c] = invstd[c]; }
}

extern "C" int cdnet_bn_backward_stats(const cdnet_bn_bwd_args *a, const float *gamma, float *dgamma, float *dbeta, float *workspace,
                                       size_t workspace_floats, float *ktab, void *stream) {
    BnBwdArgs A;
    int rc = fill_bn_args(a, A, "cdnet_bn_backward_stats");
    if (rc) return rc;
    CDNET_REQUIRE(bn_plain_case(A) && A.f16 != 2 && gamma && workspace && ktab, "cdnet_bn_backward_stats: plain 16-bit case only (one same-size gradient, BatchNorm + ReLU, no residual)");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)A.N * A.H * A.W;
    CDNET_REQUIRE(npix * (size_t)A.C < ((size_t)1 << 32) && npix < ((size_t)1 << 31), "cdnet_bn_backward_stats: tensor too large for 32-bit pixel indexing");
    const int ppb = 256 / (A.C / 8);
    int nb = (int)((npix + (size_t)ppb * BN_U - 1) / ((size_t)ppb * BN_U));
    if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
    if (nb < 1) nb = 1;
    const size_t need = (size_t)nb * 2 * A.C;
    if (workspace_floats < need) { set_error("cdnet_bn_backward_stats: workspace %zu < %zu floats", workspace_floats, need); return CDNET_E_WORKSPACE; }
    A.partial = workspace;
    A.draw = nullptr; A.dz_out = nullptr; A.rev = 0;
    bn_ktab_copy_kernel<<<cdiv(A.C, 256), 256, 0, st>>>(A.scale, A.shift, A.mean, A.invstd, A.C, ktab);
    bn_bwd_reduce_flat_kernel<1, false><<<nb, 256, 0, st>>>(A);
    bn_bwd_finalize_kernel<<<A.C, 256, 0, st>>>(A.partial, nb, A.C, (float)npix, gamma, A.invstd, dgamma, dbeta, ktab + 4 * A.C, ktab + 5 * A.C,
                                                 ktab + 6 * A.C);
    return check_launch("cdnet_bn_backward_stats");
}

/* finalize pass alone over partial rows f32 [nb][2][C] that somebody else produced (the backward-data kernel with
 * cdnet_conv_args.ws = 2): dgamma, dbeta and rows 4..6 (k1 | k2 | k3) of ktab - all cdnet_bn_backward_apply reads; rows 0..3 are left
 * alone (their only reader, round 2's fused convolution source, is gone: nine copy launches per training step less) */
extern "C" int cdnet_bn_backward_finalize(const cdnet_bn_bwd_args *a, const float *gamma, float *dgamma, float *dbeta, const float *partial,
                                          int nb, float *ktab, void *stream) {
    BnBwdArgs A;
    int rc = fill_bn_args(a, A, "cdnet_bn_backward_finalize");
    if (rc) return rc;
    CDNET_REQUIRE(bn_plain_case(A) && gamma && partial && ktab && nb >= 1, "cdnet_bn_backward_finalize: plain case only");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)A.N * A.H * A.W;
    bn_bwd_finalize_kernel<<<A.C, 256, 0, st>>>(partial, nb, A.C, (float)npix, gamma, A.invstd, dgamma, dbeta, ktab + 4 * A.C, ktab + 5 * A.C,
                                                 ktab + 6 * A.C);
    return check_launch("cdnet_bn_backward_finalize");
}

extern "C" int cdnet_bn_backward_apply(const cdnet_bn_bwd_args *a, const float *ktab, uint16_t *draw, void *stream) {
    BnBwdArgs A;
    int rc = fill_bn_args(a, A, "cdnet_bn_backward_apply");
    if (rc) return rc;
    CDNET_REQUIRE(bn_plain_case(A) && ktab && draw, "cdnet_bn_backward_apply: plain case only");
    const size_t npix = (size_t)A.N * A.H * A.W;
    const bool f32 = A.f16 == 2;                                 // fp32 tensors (gradient, raw output, dRaw): the flat32 kernel
    CDNET_REQUIRE(!f32 || A.C <= 1024, "cdnet_bn_backward_apply(f32): C=%d > 1024", A.C);
    const int ppb = 256 / (A.C / (f32 ? 4 : 8));
    int nb = (int)((npix + (size_t)ppb * BN_U - 1) / ((size_t)ppb * BN_U));
    if (nb > bn_blocks_cap()) nb = bn_blocks_cap();
    if (nb < 1) nb = 1;
    A.rev = 1;
    A.draw = draw; A.dz_out = nullptr;
    A.k1 = const_cast<float *>(ktab) + 4 * A.C; A.k2 = const_cast<float *>(ktab) + 5 * A.C; A.k3 = const_cast<float *>(ktab) + 6 * A.C;
    if (f32) launch_flat32<true>(A, nb, (hipStream_t)stream);
    else bn_bwd_apply_flat_kernel<1, false><<<nb, 256, 0, (hipStream_t)stream>>>(A);
    return check_launch("cdnet_bn_backward_apply");
}

extern "C" size_t cdnet_bn_backward_workspace_floats(int C) { return (size_t)BN_MAX_BLOCKS * 2 * C + 3 * (size_t)C; }

static HeadFeat mk_hf(const cdnet_head_feat &f) {
    HeadFeat h;
    h.raw = f.raw; h.res = f.res; h.scale = f.scale; h.shift = f.shift; h.relu = f.relu; h.f16 = f.f16;
    return h;
}

extern "C" int cdnet_dam_head_backward(const cdnet_head_feat *f1, const cdnet_head_feat *f2, const cdnet_head_feat *f3,
                                       const float *head_weights, const float *dmask, const float *dpoint, const float *ddir,
                                       int N, int H, int W, uint16_t *df1, uint16_t *df2, uint16_t *df3, float *workspace,
                                       size_t workspace_floats, float *dhead_weights, void *stream) {
    CDNET_REQUIRE(f1 && f2 && f3 && head_weights && dmask && dpoint && ddir && df1 && df2 && df3 && workspace && dhead_weights,
                  "cdnet_dam_head_backward: null pointer");
    static_assert(HEADW_FLOATS == CDNET_HEAD_WEIGHT_FLOATS, "head weight block");
    const size_t total = (size_t)N * H * W;
    const int nb = 1024;                                    // both kernels write partial[nb][855] (disjoint column ranges)
    const size_t need = (size_t)nb * HEADW_FLOATS + total * 16;
    if (workspace_floats < need) { set_error("cdnet_dam_head_backward: workspace %zu < %zu floats", workspace_floats, need); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float *partial = workspace, *coef = workspace + (size_t)nb * HEADW_FLOATS;
    const HeadFeat a = mk_hf(*f1), b = mk_hf(*f2), c = mk_hf(*f3);
    auto plain = [](const HeadFeat &f) { return f.f16 == 0 && !f.scale && !f.relu && !f.res; };
    auto train = [](const HeadFeat &f) { return f.f16 == 1 && f.scale && f.relu && f.res; };
    const HeadW *hw = reinterpret_cast<const HeadW *>(head_weights);
    if (plain(a) && plain(b) && plain(c)) {
        dam_head_bwd_kernel<0><<<nb, 256, 0, st>>>(a, b, c, hw, dmask, dpoint, ddir, N, H * W, df1, df2, df3, coef, partial);
        dam_head_wgrad_kernel<0><<<nb, 256, 0, st>>>(a, b, c, coef, N, H * W, partial);
    } else if (train(a) && train(b) && train(c)) {
        dam_head_bwd_kernel<1><<<nb, 256, 0, st>>>(a, b, c, hw, dmask, dpoint, ddir, N, H * W, df1, df2, df3, coef, partial);
        dam_head_wgrad_kernel<1><<<nb, 256, 0, st>>>(a, b, c, coef, N, H * W, partial);
    } else {
        dam_head_bwd_kernel<2><<<nb, 256, 0, st>>>(a, b, c, hw, dmask, dpoint, ddir, N, H * W, df1, df2, df3, coef, partial);
        dam_head_wgrad_kernel<2><<<nb, 256, 0, st>>>(a, b, c, coef, N, H * W, partial);
    }
    reduce_partials_kernel<<<cdiv(HEADW_FLOATS, 4), 256, 0, st>>>(workspace, nb, HEADW_FLOATS, dhead_weights);
    return check_launch("cdnet_dam_head_backward");
}

extern "C" size_t cdnet_dam_head_backward_workspace_floats(int N, int H, int W) {
    return (size_t)1024 * HEADW_FLOATS + (size_t)N * H * W * 16;
}

static int loss_nchunk(int P, int tpb) {
    int nchunk = cdiv(P, tpb * 8);
    return nchunk > 64 ? 64 : nchunk;
}

template <int ND>
static size_t dam_loss_ws(int B, int P) {
    using Y = LossLay<ND>;
    const int nchunk = loss_nchunk(P, Y::TPB);
    return (size_t)B * nchunk * Y::SUMS + (size_t)B * Y::COEF + (size_t)B * Y::SUMS + 16 + (size_t)B;   // partial | coef | sums | pad | single(int)
}

template <int ND>
static int dam_loss_impl(LossIn L, int B, int P, float *workspace, float *losses, float *dmask, float *dpoint, float *ddir, hipStream_t st) {
    using Y = LossLay<ND>;
    const int nchunk = loss_nchunk(P, Y::TPB);
    float *partial = workspace;
    float *coef = partial + (size_t)B * nchunk * Y::SUMS;
    float *sums = coef + (size_t)B * Y::COEF;
    int *err = reinterpret_cast<int *>(sums + (size_t)B * Y::SUMS);          // first word of the 16-float pad
    int *single = reinterpret_cast<int *>(sums + (size_t)B * Y::SUMS + 16);
    if (hipMemsetAsync(err, 0, sizeof(int), st) != hipSuccess) return check_launch("cdnet_dam_loss(memset)");
    L.single = single;
    loss_single_kernel<<<B, 1024, 0, st>>>(L.dirlab, L.label, P, ND, single, err);
    loss_reduce_kernel<ND><<<dim3(nchunk, B), Y::TPB, 0, st>>>(L, partial);
    loss_finalize_kernel<ND><<<1, 256, 0, st>>>(partial, nchunk, B, P, sums, coef, losses, err);
    if (dmask) loss_grad_kernel<ND><<<dim3(lin_grid((size_t)P, 256), B), 256, 0, st>>>(L, coef, dmask, dpoint, ddir);
    return check_launch("cdnet_dam_loss");
}

extern "C" size_t cdnet_dam_loss_classes_workspace_floats(int B, int P, int direction_classes) {
    return direction_classes == 5 ? dam_loss_ws<5>(B, P) : direction_classes == 17 ? dam_loss_ws<17>(B, P) : dam_loss_ws<9>(B, P);
}
extern "C" size_t cdnet_dam_loss_workspace_floats(int B, int P) { return dam_loss_ws<9>(B, P); }

extern "C" int cdnet_dam_loss_classes(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                                      const uint16_t *point_target_f16, const uint8_t *weight_u8, int B, int H, int W, int direction_classes,
                                      int quirk_sample0, float *workspace, size_t workspace_floats, float *losses, float *dmask,
                                      float *dpoint, float *ddir, void *stream) {
    CDNET_REQUIRE(mask && point && dirn && label && dirlab && point_target_f16 && weight_u8 && workspace && losses,
                  "cdnet_dam_loss: null pointer");
    CDNET_REQUIRE(B >= 1 && B <= 64 && H > 0 && W > 0, "cdnet_dam_loss: batch %d not in [1,64]", B);
    CDNET_REQUIRE(direction_classes == 5 || direction_classes == 9 || direction_classes == 17,
                  "cdnet_dam_loss: direction_classes %d must be 5, 9 or 17 (options.py:45)", direction_classes);
    const int P = H * W;
    if (workspace_floats < cdnet_dam_loss_classes_workspace_floats(B, P, direction_classes)) { set_error("cdnet_dam_loss: workspace too small"); return CDNET_E_WORKSPACE; }
    if (dmask) CDNET_REQUIRE(dpoint && ddir, "cdnet_dam_loss: all three gradient outputs or none");
    LossIn L;
    L.mask = mask; L.point = point; L.dirn = dirn; L.label = label; L.dirlab = dirlab; L.point_t = point_target_f16;
    L.weight = weight_u8; L.single = nullptr; L.B = B; L.P = P; L.quirk0 = quirk_sample0;
    hipStream_t st = (hipStream_t)stream;
    if (direction_classes == 5) return dam_loss_impl<5>(L, B, P, workspace, losses, dmask, dpoint, ddir, st);
    if (direction_classes == 17) return dam_loss_impl<17>(L, B, P, workspace, losses, dmask, dpoint, ddir, st);
    return dam_loss_impl<9>(L, B, P, workspace, losses, dmask, dpoint, ddir, st);
}

extern "C" int cdnet_dam_loss(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                              const uint16_t *point_target_f16, const uint8_t *weight_u8, int B, int H, int W, int quirk_sample0,
                              float *workspace, size_t workspace_floats, float *losses, float *dmask, float *dpoint, float *ddir,
                              void *stream) {
    return cdnet_dam_loss_classes(mask, point, dirn, label, dirlab, point_target_f16, weight_u8, B, H, W, 9, quirk_sample0, workspace,
                                  workspace_floats, losses, dmask, dpoint, ddir, stream);
}

template <int ND>
static int val_sums_impl(ValIn L, int B, int P, float *workspace, float *sums, hipStream_t st) {
    using Y = ValLay<ND>;
    const int nchunk = loss_nchunk(P, Y::TPB);
    val_sums_kernel<ND><<<dim3(nchunk, B), Y::TPB, 0, st>>>(L, workspace);
    val_sums_reduce_kernel<<<cdiv(B * Y::SUMS, 256), 256, 0, st>>>(workspace, nchunk, B, Y::SUMS, sums);
    return check_launch("cdnet_dam_val_sums");
}

extern "C" size_t cdnet_dam_val_sums_classes_workspace_floats(int B, int P, int direction_classes) {
    if (direction_classes == 5) return (size_t)B * loss_nchunk(P, ValLay<5>::TPB) * ValLay<5>::SUMS;
    if (direction_classes == 17) return (size_t)B * loss_nchunk(P, ValLay<17>::TPB) * ValLay<17>::SUMS;
    return (size_t)B * loss_nchunk(P, ValLay<9>::TPB) * ValLay<9>::SUMS;
}
extern "C" size_t cdnet_dam_val_sums_workspace_floats(int B, int P) { return cdnet_dam_val_sums_classes_workspace_floats(B, P, 9); }

extern "C" int cdnet_dam_val_sums_classes(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                                          const uint16_t *point_target_f16, const uint8_t *weight_u8, const int *dir_rank_host,
                                          int direction_classes, int B, int H, int W, float *workspace, size_t workspace_floats, float *sums,
                                          void *stream) {
    CDNET_REQUIRE(mask && point && dirn && label && dirlab && point_target_f16 && weight_u8 && dir_rank_host && workspace && sums,
                  "cdnet_dam_val_sums: null pointer");
    CDNET_REQUIRE(B >= 1 && B <= 64 && H > 0 && W > 0, "cdnet_dam_val_sums: batch %d not in [1,64]", B);
    CDNET_REQUIRE(direction_classes == 5 || direction_classes == 9 || direction_classes == 17,
                  "cdnet_dam_val_sums: direction_classes %d must be 5, 9 or 17 (options.py:45)", direction_classes);
    const int P = H * W;
    if (workspace_floats < cdnet_dam_val_sums_classes_workspace_floats(B, P, direction_classes)) { set_error("cdnet_dam_val_sums: workspace too small"); return CDNET_E_WORKSPACE; }
    ValIn L;
    L.mask = mask; L.point = point; L.dirn = dirn; L.label = label; L.dirlab = dirlab; L.weight = weight_u8; L.point_t = point_target_f16;
    for (int k = 0; k < 17; ++k) L.lut[k] = k < direction_classes ? dir_rank_host[k] : -1;
    L.B = B; L.P = P;
    hipStream_t st = (hipStream_t)stream;
    if (direction_classes == 5) return val_sums_impl<5>(L, B, P, workspace, sums, st);
    if (direction_classes == 17) return val_sums_impl<17>(L, B, P, workspace, sums, st);
    return val_sums_impl<9>(L, B, P, workspace, sums, st);
}

extern "C" int cdnet_dam_val_sums(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                                  const uint16_t *point_target_f16, const uint8_t *weight_u8, const int *dir_rank_host, int B, int H, int W,
                                  float *workspace, size_t workspace_floats, float *sums, void *stream) {
    return cdnet_dam_val_sums_classes(mask, point, dirn, label, dirlab, point_target_f16, weight_u8, dir_rank_host, 9, B, H, W, workspace,
                                      workspace_floats, sums, stream);
}

extern "C" int cdnet_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale, void *stream) {
    CDNET_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "cdnet_adam_step: bad args");
    if (n == 0) return CDNET_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<lin_grid(n, 4096), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                                    weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return check_launch("cdnet_adam_step");
}

extern "C" size_t cdnet_final_conv1x1_backward_workspace_floats(void) { return (size_t)1025 * (FC_KMOST * 64 + FC_KMOST); }

extern "C" int cdnet_final_conv1x1_backward(const cdnet_head_feat *f, const float *w, const float *dlogits, int K, int N, int H, int W,
                                            uint16_t *df, float *workspace, size_t workspace_floats, float *dw, float *db, void *stream) {
    CDNET_REQUIRE(f && f->raw && w && dlogits && df && workspace && dw && db, "cdnet_final_conv1x1_backward: null pointer");
    CDNET_REQUIRE(K >= 1 && K <= FC_KMOST && N > 0 && H > 0 && W > 0, "cdnet_final_conv1x1_backward: K=%d must be in [1,%d]", K, FC_KMOST);
    if (workspace_floats < cdnet_final_conv1x1_backward_workspace_floats()) { set_error("cdnet_final_conv1x1_backward: workspace too small"); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)N * H * W;
    const bool wide = K > FC_KMAX;
    const bool most = K > FC_KWIDE;
    const int KM = most ? FC_KMOST : wide ? FC_KWIDE : FC_KMAX;
    int nb = (int)((npix + (wide ? 31 : 63)) / (wide ? 32 : 64));
    if (nb > 1024) nb = 1024;
    const int ROW = KM * 64 + KM;
    float *sums = workspace + (size_t)1024 * ROW;
    if (most) final_conv_bwd_kernel<FC_KMOST, 8><<<nb, 256, 0, st>>>(mk_hf(*f), w, dlogits, K, N, H * W, df, workspace);
    else if (wide) final_conv_bwd_kernel<FC_KWIDE, 8><<<nb, 256, 0, st>>>(mk_hf(*f), w, dlogits, K, N, H * W, df, workspace);
    else final_conv_bwd_kernel<FC_KMAX, 4><<<nb, 256, 0, st>>>(mk_hf(*f), w, dlogits, K, N, H * W, df, workspace);
    reduce_partials_kernel<<<cdiv(ROW, 4), 256, 0, st>>>(workspace, nb, ROW, sums);
    final_conv_scatter_kernel<<<cdiv(K * 64, 256), 256, 0, st>>>(sums, K, KM, dw, db);
    return check_launch("cdnet_final_conv1x1_backward");
}

extern "C" size_t cdnet_bias_grad_workspace_floats(int C) { return (size_t)512 * C; }

static int bias_grad_impl(const uint16_t *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db, void *stream, bool f32);
extern "C" int cdnet_bias_grad(const uint16_t *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db,
                               void *stream) {
    return bias_grad_impl(grad_out, npix, C, workspace, workspace_floats, db, stream, false);
}
extern "C" int cdnet_bias_grad_f32(const float *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db,
                                   void *stream) {
    return bias_grad_impl(reinterpret_cast<const uint16_t *>(grad_out), npix, C, workspace, workspace_floats, db, stream, true);
}
static int bias_grad_impl(const uint16_t *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db, void *stream, bool f32) {
    CDNET_REQUIRE(grad_out && workspace && db, "cdnet_bias_grad: null pointer");
    CDNET_REQUIRE(C >= 8 && C % 8 == 0 && C <= 2048 && npix > 0 && npix < ((size_t)1 << 31), "cdnet_bias_grad: C=%d unsupported", C);
    if (workspace_floats < cdnet_bias_grad_workspace_floats(C)) { set_error("cdnet_bias_grad: workspace too small"); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const unsigned ppb = 256 / (C / 8);
    int nb = (int)((npix + ppb * 8 - 1) / (ppb * 8));
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    if (f32) bias_grad_kernel<true><<<nb, 256, 0, st>>>(grad_out, (unsigned)npix, C, workspace);
    else bias_grad_kernel<false><<<nb, 256, 0, st>>>(grad_out, (unsigned)npix, C, workspace);
    reduce_partials_kernel<<<cdiv(C, 4), 256, 0, st>>>(workspace, nb, C, db);
    return check_launch("cdnet_bias_grad");
}
