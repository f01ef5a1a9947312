// Streaming (HBM-bound) kernels of the model path that are not convolutions:
//   input pack (ToTensor'd NCHW f32 -> NHWC bf16, channels zero-padded to 16)
//   BatchNorm bookkeeping: eval fold (model_unet_rev1.py BN layers in eval()), training finalize from the
//     per-tile statistics the convolution epilogue emits (nn.BatchNorm2d training semantics: biased variance for
//     normalisation, unbiased for running_var, momentum 0.1, eps 1e-5)
//   DAM head: point_conv, directionAtt, direction_conv, maskAtt, mask_conv fused per pixel
//     (models/dam/model_unet_rev1.py:8-17, 227-231, 258-263)
#include <algorithm>
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

// x f32 [N][C][H][W] -> out bf16 [N][H][W][16]  (C <= 16)
__global__ __launch_bounds__(256) void input_pack_kernel(const float *__restrict__ x, int N, int C, int plane,
                                                         unsigned short *__restrict__ out) {
    const size_t total = (size_t)N * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, p = i - n * plane;
        unsigned short v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = c < C ? f2bf(x[(n * C + c) * plane + p]) : (unsigned short)0;
        uint4 *dst = reinterpret_cast<uint4 *>(out + i * 16);
        dst[0] = *reinterpret_cast<const uint4 *>(v);
        dst[1] = *reinterpret_cast<const uint4 *>(v + 8);
    }
}

// x f32 [N][C][H][W] -> out f32 [N][H][W][16]  (fp32-precision path)
__global__ __launch_bounds__(256) void input_pack_f32_kernel(const float *__restrict__ x, int N, int C, int plane, float *__restrict__ out) {
    const size_t total = (size_t)N * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, p = i - n * plane;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = c < C ? x[(n * C + c) * plane + p] : 0.f;
        float4 *dst = reinterpret_cast<float4 *>(out + i * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
}

// eval-mode fold: scale = g / sqrt(rv + eps); shift = b + (bias - rm) * scale
__global__ void bn_fold_eval_kernel(const float *g, const float *b, const float *rm, const float *rv, const float *bias,
                                    float eps, int C, float *scale, float *shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = g[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = b[c] + ((bias ? bias[c] : 0.f) - rm[c]) * s;
}

// training finalize: stats f32 [T][2][C] (sum, sumsq of the bias-free conv output over `count` elements per channel)
// -> scale = g*invstd, shift = b - mean*scale (bias cancels), mean/invstd saved for backward,
//    running_mean = (1-m)*rm + m*(mean + bias), running_var = (1-m)*rv + m*var*count/(count-1)
__global__ __launch_bounds__(256) void bn_finalize_train_kernel(const float *__restrict__ stats, int T, int C, float count,
                                                                const float *g, const float *b, const float *bias,
                                                                float eps, float momentum, float *rm, float *rv,
                                                                float *scale, float *shift, float *mean_out,
                                                                float *invstd_out) {
    // one workgroup per channel (C workgroups keep every CU busy; the 4-byte reads of different channels share cache
    // lines in L2): thread i sums tiles i, i+256, ... in order, then a fixed LDS tree - deterministic
    __shared__ double s_s[256], s_q[256];
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int t = lane; t < T; t += 256) {
        s += (double)stats[((size_t)t * 2) * C + c];
        q += (double)stats[((size_t)t * 2 + 1) * C + c];
    }
    s_s[lane] = s; s_q[lane] = q;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if (lane < o) { s_s[lane] += s_s[lane + o]; s_q[lane] += s_q[lane + o]; }
        __syncthreads();
    }
    s = s_s[0]; q = s_q[0];
    if (lane == 0) {
        double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0) var = 0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float sc = g[c] * invstd;
        scale[c] = sc;
        shift[c] = b[c] - (float)mean * sc;
        if (mean_out) mean_out[c] = (float)mean;
        if (invstd_out) invstd_out[c] = invstd;
        if (rm) {
            float mb = (float)mean + (bias ? bias[c] : 0.f);
            rm[c] = (1.f - momentum) * rm[c] + momentum * mb;
            float unb = (float)(var * (count / (count > 1.f ? count - 1.f : 1.f)));
            rv[c] = (1.f - momentum) * rv[c] + momentum * unb;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// DAM head.  Each feature F_k = relu(raw*scale + shift + res) is recomputed from its stored pieces
// (scale==NULL: the tensor already holds the activated feature).
// ------------------------------------------------------------------------------------------------------
struct HeadFeat {
    const unsigned short *raw;
    const unsigned short *res;
    const float *scale;
    const float *shift;
    int relu;
    int f16;
};

struct HeadW {            // 64-channel 1x1 heads, fp32
    float wp[64], wd[9][64], wm[3][64];
    float bp, bd[9], bm[3];
    float a1;             // directionAtt.Conv1x1 (1->1, no bias)
    float a2[9];          // maskAtt.Conv1x1 (9->1, no bias)
};

// fp32-stored feature (f16 == 2): NC channels starting at c0, plain fp32 arithmetic (no 16-bit rounding anywhere)
template <int NC>
__device__ __forceinline__ void load_feat_f32(const HeadFeat &f, size_t pix, int c0, const float *s_sc, const float *s_sh, float *v) {
    const float4 *pr = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(f.raw) + pix * 64 + c0);
    const float4 *ps = f.res ? reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(f.res) + pix * 64 + c0) : nullptr;
#pragma unroll
    for (int q = 0; q < NC / 4; ++q) {
        const float4 r = pr[q];
        float x[4] = {r.x, r.y, r.z, r.w};
        float rr[4] = {0.f, 0.f, 0.f, 0.f};
        if (ps) { const float4 t = ps[q]; rr[0] = t.x; rr[1] = t.y; rr[2] = t.z; rr[3] = t.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = x[j];
            if (f.scale) t = fmaf(t, s_sc[c0 + q * 4 + j], s_sh[c0 + q * 4 + j]);
            if (ps) t += rr[j];
            if (f.relu) t = fmaxf(t, 0.f);
            v[q * 4 + j] = t;
        }
    }
}

__device__ __forceinline__ void load_feat64(const HeadFeat &f, size_t pix, const float *s_sc, const float *s_sh, float *v) {
    if (f.f16 == 2) { load_feat_f32<64>(f, pix, 0, s_sc, s_sh, v); return; }
    const uint4 *pr = reinterpret_cast<const uint4 *>(f.raw + pix * 64);
    const uint4 *ps = f.res ? reinterpret_cast<const uint4 *>(f.res + pix * 64) : nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint4 r = pr[q];
        const unsigned short *h = reinterpret_cast<const unsigned short *>(&r);
        uint4 rr = make_uint4(0, 0, 0, 0);
        if (ps) rr = ps[q];
        const unsigned short *hr = reinterpret_cast<const unsigned short *>(&rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = f.f16 ? h2f(h[j]) : bf2f(h[j]);
            if (f.scale || ps || f.relu) {
                if (f.scale) x = fmaf(x, s_sc[q * 8 + j], s_sh[q * 8 + j]);
                if (ps) x += f.f16 ? h2f(hr[j]) : bf2f(hr[j]);
                if (f.relu) x = fmaxf(x, 0.f);
                x = bf2f(f2bf(x));       // the convolutions consume the feature rounded to bf16; keep the head consistent
            }
            v[q * 8 + j] = x;
        }
    }
}

// 16 channels [16q, 16q+16) of the feature at one pixel (q = lane & 3): four lanes share a pixel
// 16 channels (slot q of 4) of one pixel of a 16-bit feature: the raw vectors first (so that a kernel can put the loads of all its
// features in flight before it touches any of them), the lazily applied transform second
struct FeatRaw16 {
    uint4 r[2], s[2];
};
__device__ __forceinline__ void load_raw16(const HeadFeat &f, size_t pix, int q, FeatRaw16 &R) {
    const uint4 *pr = reinterpret_cast<const uint4 *>(f.raw + pix * 64 + q * 16);
    R.r[0] = pr[0];
    R.r[1] = pr[1];
    R.s[0] = R.s[1] = make_uint4(0, 0, 0, 0);
    if (f.res) {
        const uint4 *ps = reinterpret_cast<const uint4 *>(f.res + pix * 64 + q * 16);
        R.s[0] = ps[0];
        R.s[1] = ps[1];
    }
}
__device__ __forceinline__ void feat_from_raw16(const HeadFeat &f, const FeatRaw16 &R, int q, const float *s_sc, const float *s_sh, float *v) {
    const bool has_res = f.res != nullptr;
    const bool fast = f.f16 && f.scale && f.relu;        // training-mode feature: packed math (xform.h)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 r = R.r[h2], rr = R.s[h2];
        if (f.f16 && !f.scale && f.relu && has_res) {      // eval-mode residual unit output: relu(raw + res)
            xf_addrelu_f16_to_f32(__builtin_bit_cast(xf_u32x4, r), __builtin_bit_cast(xf_u32x4, rr), v + h2 * 8);
            continue;
        }
        if (fast) {
            const xf_u32x4 a = __builtin_bit_cast(xf_u32x4, r), b = __builtin_bit_cast(xf_u32x4, rr);
            const int c0 = q * 16 + h2 * 8;
            if (has_res) xf_bnrelu_f16_to_f32<true>(a, b, s_sc + c0, s_sh + c0, v + h2 * 8);
            else xf_bnrelu_f16_to_f32<false>(a, a, s_sc + c0, s_sh + c0, v + h2 * 8);
            continue;
        }
        const unsigned short *h = reinterpret_cast<const unsigned short *>(&r);
        const unsigned short *hr = reinterpret_cast<const unsigned short *>(&rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = q * 16 + h2 * 8 + j;
            float x = f.f16 ? h2f(h[j]) : bf2f(h[j]);
            if (f.scale || has_res || f.relu) {
                if (f.scale) x = fmaf(x, s_sc[c], s_sh[c]);
                if (has_res) x += f.f16 ? h2f(hr[j]) : bf2f(hr[j]);
                if (f.relu) x = fmaxf(x, 0.f);
                x = bf2f(f2bf(x));
            }
            v[h2 * 8 + j] = x;
        }
    }
}
__device__ __forceinline__ void load_feat16(const HeadFeat &f, size_t pix, int q, const float *s_sc, const float *s_sh, float *v) {
    if (f.f16 == 2) { load_feat_f32<16>(f, pix, q * 16, s_sc, s_sh, v); return; }
    FeatRaw16 R;
    load_raw16(f, pix, q, R);
    feat_from_raw16(f, R, q, s_sc, s_sh, v);
}

__device__ __forceinline__ float quad_sum(float v) { return xf_quad_sum(v); }        // sum over the 4 lanes of a pixel

// four lanes per pixel, 16 channels each: 4x the parallelism and a quarter of the registers of one-thread-per-pixel.
// FM: the storage / transform of all three features, decided by the launcher - 1 fp16 raw x scale + shift + residual -> ReLU
// (training-mode residual-unit outputs of the unfused form), 2 anything (run-time flags; with them the kernel is 9 000 instructions of
// branches and every join drains the loads in flight).  Plain stored features - eval mode and the fused training forward - take
// dam_head_mfma_kernel below.
template <int FM>
__global__ __launch_bounds__(256, (FM == 2 ? 1 : 4)) void dam_head_fwd_kernel(HeadFeat f1, HeadFeat f2, HeadFeat f3, const HeadW *__restrict__ hw,
                                                           int N, int plane, float *__restrict__ mask,
                                                           float *__restrict__ point, float *__restrict__ dirn) {
    __shared__ HeadW w;
    __shared__ float s_sc[3][64], s_sh[3][64];
    {
        const float *src = reinterpret_cast<const float *>(hw);
        float *dst = reinterpret_cast<float *>(&w);
        for (int i = threadIdx.x; i < (int)(sizeof(HeadW) / 4); i += 256) dst[i] = src[i];
        if (threadIdx.x < 64) {
            const HeadFeat *fs[3] = {&f1, &f2, &f3};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_sc[k][threadIdx.x] = fs[k]->scale ? fs[k]->scale[threadIdx.x] : 1.f;
                s_sh[k][threadIdx.x] = fs[k]->scale ? fs[k]->shift[threadIdx.x] : 0.f;
            }
        }
    }
    __syncthreads();
    const size_t total = (size_t)N * plane;
    const int q = threadIdx.x & 3;
    const bool all16 = FM != 2 || (f1.f16 != 2 && f2.f16 != 2 && f3.f16 != 2);
    // FM 0 / 1: fixed conversions of the raw vectors
    auto conv = [&](const HeadFeat &f, const FeatRaw16 &R, int k, float *v) {
        if (FM == 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                xf_bnrelu_f16_to_f32<true>(__builtin_bit_cast(xf_u32x4, R.r[h2]), __builtin_bit_cast(xf_u32x4, R.s[h2]), s_sc[k] + q * 16 + h2 * 8,
                                           s_sh[k] + q * 16 + h2 * 8, v + h2 * 8);
        } else {
            feat_from_raw16(f, R, q, s_sc[k], s_sh[k], v);
        }
    };
    auto fetch = [&](const HeadFeat &f, size_t pix, FeatRaw16 &R) {
        if (FM == 2) { load_raw16(f, pix, q, R); return; }
        const uint4 *pr = reinterpret_cast<const uint4 *>(f.raw + pix * 64 + q * 16);
        R.r[0] = pr[0];
        R.r[1] = pr[1];
        if (FM == 1) {
            const uint4 *ps = reinterpret_cast<const uint4 *>(f.res + pix * 64 + q * 16);
            R.s[0] = ps[0];
            R.s[1] = ps[1];
        }
    };
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        // the head weights stay in LDS: without this fence the compiler hoists all 208 of a lane's weight reads out of the loop
        // (286-402 registers, one wave per SIMD - or, under a register bound, spills them)
        asm volatile("" ::: "memory");
        const size_t i = base + (threadIdx.x >> 2);
        const bool ok = i < total;
        const size_t ii = ok ? i : total - 1;                 // keep all lanes alive for the shuffles
        const size_t n = ii / plane, p = ii - n * plane;
        float v[16];
        // 16-bit features: all three features' vectors in flight at once (one memory round trip per pixel instead of three)
        FeatRaw16 R1, R2, R3;
        if (all16) {
            fetch(f3, ii, R3);
            fetch(f2, ii, R2);
            fetch(f1, ii, R1);
            conv(f3, R3, 2, v);
        } else {
            load_feat16(f3, ii, q, s_sc[2], s_sh[2], v);
        }
        const float pt = quad_sum(xf_dot16(w.wp + q * 16, v)) + w.bp;
        const float g1 = 1.f + 1.f / (1.f + expf(-(w.a1 * pt)));
        if (all16) conv(f2, R2, 1, v);
        else load_feat16(f2, ii, q, s_sc[1], s_sh[1], v);
        float d[9], q2 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            d[k] = fmaf(g1, quad_sum(xf_dot16(w.wd[k] + q * 16, v)), w.bd[k]);
            q2 = fmaf(w.a2[k], d[k], q2);
        }
        const float g2 = 1.f + 1.f / (1.f + expf(-q2));
        if (all16) conv(f1, R1, 0, v);
        else load_feat16(f1, ii, q, s_sc[0], s_sh[0], v);
        float mk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mk[k] = fmaf(g2, quad_sum(xf_dot16(w.wm[k] + q * 16, v)), w.bm[k]);
        }
        if (ok) {
            // the 13 outputs of a pixel are spread over its 4 lanes: lane q writes outputs q, q+4, q+8, (q+12)
            if (q == 0) { point[n * plane + p] = pt; }
#pragma unroll
            for (int k = 0; k < 9; ++k) if ((k & 3) == q) dirn[(n * 9 + k) * plane + p] = d[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k + 1 == q) mask[(n * 3 + k) * plane + p] = mk[k];
        }
    }
}

// The head on the matrix cores - eval mode, plain bf16 features (what the fused epilogues of the 16-bit path leave).  The three 1x1
// classifiers are 64 -> {1, 9, 3} GEMMs per pixel: `v_mfma_f32_16x16x32_bf16` with the WEIGHTS as the A operand (row = output, padded to
// 16) and 16 pixels as the columns of B - a lane's B fragment is 16 contiguous bytes of a pixel (channels 8 kg .. 8 kg + 7 of the k-step),
// loaded straight from the NHWC tensor, no conversion.  The fp32 weights enter as hi + lo bf16 pairs (two MFMAs per product, 2^-16
// relative), built once per wave and kept in registers: the vector-unit kernel above reads its 208 weights per lane from LDS for every pixel
// (the LDS pipe, not HBM, set its 416 us per 64 tiles).  Lane (col, kg) then holds outputs 4 kg .. 4 kg + 3 of pixel `col`: the gates
// (revAttention, model_unet_rev1.py:8-17) need the point logit - broadcast from the kg = 0 lane - and the 9-term sum over the direction
// logits - a partial sum per lane, two cross-lane adds.  HAS_F3 = false: `point` is given (cdnet_conv_args.dot_out).
typedef float hd_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 hd_bf16x8;

template <bool HAS_F3, bool F32>
__global__ __launch_bounds__(256) void dam_head_mfma_kernel(const unsigned short *__restrict__ f1, const unsigned short *__restrict__ f2,
                                                            const unsigned short *__restrict__ f3, const HeadW *__restrict__ hw, size_t total,
                                                            int plane, float *__restrict__ mask, float *__restrict__ point,
                                                            float *__restrict__ dirn) {
    // F32: fp32-stored features (the fp32 precision mode) - split into hi | lo bf16 like the weights, three MFMAs per product (the
    // convolutions' arithmetic); one 16-pixel group in flight per wave (the fragments take twice the registers)
    constexpr int U = F32 ? 1 : 2;                               // 16-pixel groups in flight per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kg = lane >> 4;
    union Frag { hd_bf16x8 v; unsigned short h[8]; };
    // row `col` of a classifier (zero rows beyond its outputs), channels 32 ks + 8 kg .. + 7, as hi | lo
    auto split = [&](const float *row, int ks, Frag &hi, Frag &lo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = row ? row[ks * 32 + kg * 8 + j] : 0.f;
            const unsigned short h = f2bf(x);
            hi.h[j] = h;
            lo.h[j] = f2bf(x - bf2f(h));
        }
    };
    Frag wd_h[2], wd_l[2], wm_h[2], wm_l[2], wp_h[2], wp_l[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        split(col < 9 ? hw->wd[col] : nullptr, ks, wd_h[ks], wd_l[ks]);
        split(col < 3 ? hw->wm[col] : nullptr, ks, wm_h[ks], wm_l[ks]);
        if (HAS_F3) split(col < 1 ? hw->wp : nullptr, ks, wp_h[ks], wp_l[ks]);
    }
    float bd4[4], a24[4], bm3[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 4 * kg + i;
        bd4[i] = o < 9 ? hw->bd[o] : 0.f;
        a24[i] = o < 9 ? hw->a2[o] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) bm3[i] = hw->bm[i];
    const float bp = hw->bp, a1 = hw->a1;
    const hd_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (size_t g0 = ((size_t)blockIdx.x * 4 + wave) * (16 * U); g0 < total; g0 += (size_t)gridDim.x * 4 * (16 * U)) {
        Frag b1[U][2], b2[U][2], b3[U][2];                       // 16-bit features: the fragments themselves; F32: their hi parts ...
        Frag l1[F32 ? U : 1][2], l2[F32 ? U : 1][2], l3[F32 ? U : 1][2];      // ... and lo parts
        float ptin[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t px = g0 + u * 16 + col;
            px = px < total ? px : total - 1;
            const size_t e = px * 64 + kg * 8;
            if constexpr (F32) {
                hd_f32x4 r[3][2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        if (HAS_F3) r[2][ks][hq] = *reinterpret_cast<const hd_f32x4 *>(reinterpret_cast<const float *>(f3) + e + ks * 32 + hq * 4);
                        r[1][ks][hq] = *reinterpret_cast<const hd_f32x4 *>(reinterpret_cast<const float *>(f2) + e + ks * 32 + hq * 4);
                        r[0][ks][hq] = *reinterpret_cast<const hd_f32x4 *>(reinterpret_cast<const float *>(f1) + e + ks * 32 + hq * 4);
                    }
                auto cut = [&](const hd_f32x4 (&q)[2], Frag &hi, Frag &lo) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = q[j >> 2][j & 3];
                        const unsigned short h = f2bf(x);
                        hi.h[j] = h;
                        lo.h[j] = f2bf(x - bf2f(h));
                    }
                };
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (HAS_F3) cut(r[2][ks], b3[u][ks], l3[u][ks]);
                    cut(r[1][ks], b2[u][ks], l2[u][ks]);
                    cut(r[0][ks], b1[u][ks], l1[u][ks]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (HAS_F3) b3[u][ks].v = *reinterpret_cast<const hd_bf16x8 *>(f3 + e + ks * 32);
                    b2[u][ks].v = *reinterpret_cast<const hd_bf16x8 *>(f2 + e + ks * 32);
                    b1[u][ks].v = *reinterpret_cast<const hd_bf16x8 *>(f1 + e + ks * 32);
                }
            }
            if (!HAS_F3) ptin[u] = point[px];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            hd_f32x4 aP = zero, aD = zero, aM = zero;
            // (small terms first)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (HAS_F3) aP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_l[ks].v, b3[u][ks].v, aP, 0, 0, 0);
                aD = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd_l[ks].v, b2[u][ks].v, aD, 0, 0, 0);
                aM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm_l[ks].v, b1[u][ks].v, aM, 0, 0, 0);
                if constexpr (F32) {
                    if (HAS_F3) aP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_h[ks].v, l3[u][ks].v, aP, 0, 0, 0);
                    aD = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd_h[ks].v, l2[u][ks].v, aD, 0, 0, 0);
                    aM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm_h[ks].v, l1[u][ks].v, aM, 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (HAS_F3) aP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_h[ks].v, b3[u][ks].v, aP, 0, 0, 0);
                aD = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd_h[ks].v, b2[u][ks].v, aD, 0, 0, 0);
                aM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm_h[ks].v, b1[u][ks].v, aM, 0, 0, 0);
            }
            float pt;
            if (HAS_F3) pt = __shfl(aP[0], col) + bp;            // output row 0 lives in the kg = 0 lanes
            else pt = ptin[u];
            const float g1 = 1.f + 1.f / (1.f + expf(-(a1 * pt)));
            float d[4], q2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = fmaf(g1, aD[i], bd4[i]);
                q2 = fmaf(a24[i], d[i], q2);
            }
            q2 += __shfl_xor(q2, 16);
            q2 += __shfl_xor(q2, 32);
            const float g2 = 1.f + 1.f / (1.f + expf(-q2));
            const size_t idx = g0 + u * 16 + col;
            if (idx < total) {
                const size_t n = idx / plane, p = idx - n * plane;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * kg + i < 9) dirn[(n * 9 + 4 * kg + i) * plane + p] = d[i];
                if (kg == 0) {
                    if (HAS_F3) point[idx] = pt;
#pragma unroll
                    for (int i = 0; i < 3; ++i) mask[(n * 3 + i) * plane + p] = fmaf(g2, aM[i], bm3[i]);
                }
            }
        }
    }
}

// final 1x1 classifier of the plain UNet (models/unet.py:75,104): logits f32 NCHW from a 64-channel feature
__global__ __launch_bounds__(256) void final_conv1x1_kernel(HeadFeat f, const float *__restrict__ w, const float *__restrict__ b,
                                                            int K, int N, int plane, float *__restrict__ out) {
    __shared__ float s_w[32 * 64], s_b[32], s_sc[64], s_sh[64];
    for (int i = threadIdx.x; i < K * 64; i += 256) s_w[i] = w[i];
    if (threadIdx.x < K) s_b[threadIdx.x] = b[threadIdx.x];
    if (threadIdx.x < 64) {
        s_sc[threadIdx.x] = f.scale ? f.scale[threadIdx.x] : 1.f;
        s_sh[threadIdx.x] = f.scale ? f.shift[threadIdx.x] : 0.f;
    }
    __syncthreads();
    const size_t total = (size_t)N * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, p = i - n * plane;
        float v[64];
        load_feat64(f, i, s_sc, s_sh, v);
        for (int k = 0; k < K; ++k) {
            float s = s_b[k];
#pragma unroll
            for (int c = 0; c < 64; ++c) s = fmaf(s_w[k * 64 + c], v[c], s);
            out[(n * K + k) * plane + p] = s;
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Sliding windows (utils.split_forward_dam, utils.py:658-726) with the TTA view transform folded in
// (test_dam.py:313-385: PIL FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM / rotate(90, expand=True)).
// view (vy,vx) -> image (iy,ix): undo vertical / horizontal flips in the view frame, then the ccw rotation.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void view_to_image(int xf, int vy, int vx, int H, int W, int &iy, int &ix) {
    const int hv = (xf & 4) ? W : H, wv = (xf & 4) ? H : W;
    if (xf & 2) vy = hv - 1 - vy;
    if (xf & 1) vx = wv - 1 - vx;
    if (xf & 4) { iy = vx; ix = W - 1 - vy; }       // r[y'][x'] = img[x'][W-1-y']
    else { iy = vy; ix = vx; }
}

// img f32 [C][H][W] (one image) -> tiles bf16 NHWC [ny*nx][th][tw][16]; window (ky,kx) starts at (ky*stride, kx*stride)
// of the (zero-padded) view
__global__ __launch_bounds__(256) void window_pack_f32_kernel(const float *__restrict__ img, int C, int H, int W, int xf, int th,
                                                              int tw, int stride, int ny, int nx, float *__restrict__ out) {
    const int hv = (xf & 4) ? W : H, wv = (xf & 4) ? H : W;
    const size_t total = (size_t)ny * nx * th * tw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int tx = r % tw; r /= tw;
        const int ty = r % th; r /= th;
        const int kx = r % nx; const int ky = (int)(r / nx);
        const int vy = ky * stride + ty, vx = kx * stride + tx;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0.f;
        if (vy < hv && vx < wv) {
            int iy, ix;
            view_to_image(xf, vy, vx, H, W, iy, ix);
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < C) v[c] = img[((size_t)c * H + iy) * W + ix];
        }
        float4 *dst = reinterpret_cast<float4 *>(out + i * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
}

__global__ __launch_bounds__(256) void window_pack_kernel(const float *__restrict__ img, int C, int H, int W, int xf, int th,
                                                          int tw, int stride, int ny, int nx, unsigned short *__restrict__ out) {
    const int hv = (xf & 4) ? W : H, wv = (xf & 4) ? H : W;
    const size_t total = (size_t)ny * nx * th * tw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int tx = r % tw; r /= tw;
        const int ty = r % th; r /= th;
        const int kx = r % nx; const int ky = (int)(r / nx);
        const int vy = ky * stride + ty, vx = kx * stride + tx;
        unsigned short v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0;
        if (vy < hv && vx < wv) {
            int iy, ix;
            view_to_image(xf, vy, vx, H, W, iy, ix);
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < C) v[c] = f2bf(img[((size_t)c * H + iy) * W + ix]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + i * 16);
        dst[0] = *reinterpret_cast<const uint4 *>(v);
        dst[1] = *reinterpret_cast<const uint4 *>(v + 8);
    }
}

// tiles f32 NCHW [ny*nx][K][th][tw] -> stitched f32 [K][hv][wv] with the reference's interior rule:
// pixel y belongs to the LAST window k with k*stride + (k ? overlap/2 : 0) <= y   (utils.py:683-712)
__global__ __launch_bounds__(256) void window_stitch_kernel(const float *__restrict__ tiles, int K, int th, int tw, int stride,
                                                            int half_ov, int ny, int nx, int hv, int wv, float *__restrict__ out) {
    const size_t total = (size_t)K * hv * wv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int x = r % wv; r /= wv;
        const int y = r % hv; const int k = (int)(r / hv);
        int ky = y < half_ov ? 0 : (y - half_ov) / stride; ky = ky > ny - 1 ? ny - 1 : ky;
        int kx = x < half_ov ? 0 : (x - half_ov) / stride; kx = kx > nx - 1 ? nx - 1 : kx;
        out[i] = tiles[(((size_t)(ky * nx + kx) * K + k) * th + (y - ky * stride)) * tw + (x - kx * stride)];
    }
}

inline int lin_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}


// ------------------------------------------------------------------------------------------------------
// HRNet fuse / residual sums (seg_hrnet_rev1.py:256-283, 76-92, 113-133, 528-533): out = [relu]( sum_j term_j ) where a
// term is a same-size bf16 NHWC tensor or the bilinear up-sampling (F.interpolate / F.upsample mode='bilinear',
// align_corners=False) of a lower-resolution one.  The output may be a channel slice of a wider tensor (the final
// torch.cat of the four branches).  8 channels per thread, fp32 arithmetic, one rounding at the end.
// ------------------------------------------------------------------------------------------------------
struct FuseTerm { const unsigned short *x; int Hs, Ws; const float *scale, *shift; int f16; };
struct FuseArgs { FuseTerm t[4]; int nterm; int N, H, W, C; int relu; unsigned short *out; int out_cstride, out_coff; };

__device__ __forceinline__ void bf8_to_f32(uint4 u, float *v) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}

// 8 channels of one term at one source pixel: bf16 / fp16 storage, optional per-channel affine (training mode: the raw
// convolution output with its BatchNorm scale / shift)
__device__ __forceinline__ void term8(const FuseTerm &T, const unsigned short *p, int c0, float *v) {
    if (T.f16 == 2) {                                // fp32 storage: `p` already counts floats (see term_ptr)
        const float4 *q = reinterpret_cast<const float4 *>(p);
        const float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        if (T.scale) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], T.scale[c0 + j], T.shift[c0 + j]);
        }
        return;
    }
    const uint4 u = *reinterpret_cast<const uint4 *>(p);
    if (T.f16) {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] & 0xffffu));
            v[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] >> 16));
        }
    } else bf8_to_f32(u, v);
    if (T.scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], T.scale[c0 + j], T.shift[c0 + j]);
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs A) {
    constexpr int ES = F32 ? 2 : 1;                  // element size in units of unsigned short (the pointer type of the structs)
    const int VPP = A.C / 8;
    const size_t total = (size_t)A.N * A.H * A.W * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % VPP);
        const size_t pix = i / VPP;
        const int x = (int)(pix % A.W), y = (int)((pix / A.W) % A.H), n = (int)(pix / ((size_t)A.W * A.H));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k = 0; k < A.nterm; ++k) {
            const FuseTerm &T = A.t[k];
            const unsigned short *base = T.x + ((size_t)n * T.Hs * T.Ws * A.C + slot * 8) * ES;
            float v[8];
            if (T.Hs == A.H && T.Ws == A.W) {
                term8(T, base + ((size_t)y * A.W + x) * A.C * ES, slot * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            } else {
                // PyTorch upsample_bilinear2d, align_corners=False: src = max(0, (dst + 0.5) * scale - 0.5), scale = in/out
                const float sy = (float)T.Hs / (float)A.H, sx = (float)T.Ws / (float)A.W;
                float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
                fy = fy < 0.f ? 0.f : fy;
                fx = fx < 0.f ? 0.f : fx;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < T.Hs - 1 ? 1 : 0), x1 = x0 + (x0 < T.Ws - 1 ? 1 : 0);
                const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
                float a[8], b[8], c[8], d[8];
                term8(T, base + ((size_t)y0 * T.Ws + x0) * A.C * ES, slot * 8, a);
                term8(T, base + ((size_t)y0 * T.Ws + x1) * A.C * ES, slot * 8, b);
                term8(T, base + ((size_t)y1 * T.Ws + x0) * A.C * ES, slot * 8, c);
                term8(T, base + ((size_t)y1 * T.Ws + x1) * A.C * ES, slot * 8, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += hy * (hx * a[j] + lx * b[j]) + ly * (hx * c[j] + lx * d[j]);
            }
        }
        if (F32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = A.relu ? fmaxf(acc[j], 0.f) : acc[j];
            float4 *o = reinterpret_cast<float4 *>(reinterpret_cast<float *>(A.out) + pix * A.out_cstride + A.out_coff + slot * 8);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            continue;
        }
        unsigned short oh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) oh[j] = f2bf(A.relu ? fmaxf(acc[j], 0.f) : acc[j]);
        uint4 ou;
        ou.x = oh[0] | ((unsigned)oh[1] << 16); ou.y = oh[2] | ((unsigned)oh[3] << 16);
        ou.z = oh[4] | ((unsigned)oh[5] << 16); ou.w = oh[6] | ((unsigned)oh[7] << 16);
        *reinterpret_cast<uint4 *>(A.out + pix * A.out_cstride + A.out_coff + slot * 8) = ou;
    }
}


// 16-bit form with the term count known to the compiler and every load of a pixel issued before the first conversion: the generic kernel
// above walks its terms one after the other (a run-time loop over the argument struct, four dependent taps per up-sampled term) and ran the
// fuse sums of HRNet's modules at 1.2-1.7 TB/s.  Same arithmetic, term by term in the same order: bit-identical results.
template <int NT>
__global__ __launch_bounds__(256) void fuse_sum16_kernel(FuseArgs A) {
    const unsigned VPP = (unsigned)A.C / 8u;
    const unsigned total = (unsigned)A.N * (unsigned)A.H * (unsigned)A.W * VPP;       // (the launcher checks that it fits 32 bits)
    const unsigned W_ = (unsigned)A.W, H_ = (unsigned)A.H;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned slot = i % VPP, pix = i / VPP;
        const unsigned x = pix % W_, yy = pix / W_, y = yy % H_, n = yy / H_;
        const int c0 = (int)slot * 8;
        uint4 raw[NT][4];
        float hy[NT], hx[NT], ly[NT], lx[NT];
        float4 sc[NT][2], sh[NT][2];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const FuseTerm &T = A.t[k];
            sc[k][0] = sc[k][1] = sh[k][0] = sh[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            raw[k][1] = raw[k][2] = raw[k][3] = make_uint4(0, 0, 0, 0);
            const unsigned short *base = T.x + (size_t)n * T.Hs * T.Ws * A.C + c0;
            if (T.Hs == A.H && T.Ws == A.W) {
                raw[k][0] = *reinterpret_cast<const uint4 *>(base + ((size_t)y * W_ + x) * A.C);
                hy[k] = hx[k] = 1.f; ly[k] = lx[k] = 0.f;
            } else {
                const float sy = (float)T.Hs / (float)A.H, sx = (float)T.Ws / (float)A.W;
                float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
                fy = fy < 0.f ? 0.f : fy;
                fx = fx < 0.f ? 0.f : fx;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < T.Hs - 1 ? 1 : 0), x1 = x0 + (x0 < T.Ws - 1 ? 1 : 0);
                ly[k] = fy - (float)y0; lx[k] = fx - (float)x0; hy[k] = 1.f - ly[k]; hx[k] = 1.f - lx[k];
                raw[k][0] = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * T.Ws + x0) * A.C);
                raw[k][1] = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * T.Ws + x1) * A.C);
                raw[k][2] = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * T.Ws + x0) * A.C);
                raw[k][3] = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * T.Ws + x1) * A.C);
            }
            if (T.scale) {
                const float4 *ps = reinterpret_cast<const float4 *>(T.scale + c0), *pq = reinterpret_cast<const float4 *>(T.shift + c0);
                sc[k][0] = ps[0]; sc[k][1] = ps[1]; sh[k][0] = pq[0]; sh[k][1] = pq[1];
            }
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const FuseTerm &T = A.t[k];
            const bool up = !(T.Hs == A.H && T.Ws == A.W);
            const float scv[8] = {sc[k][0].x, sc[k][0].y, sc[k][0].z, sc[k][0].w, sc[k][1].x, sc[k][1].y, sc[k][1].z, sc[k][1].w};
            const float shv[8] = {sh[k][0].x, sh[k][0].y, sh[k][0].z, sh[k][0].w, sh[k][1].x, sh[k][1].y, sh[k][1].z, sh[k][1].w};
            auto cvt = [&](const uint4 &u, float *v) {
                const unsigned w[4] = {u.x, u.y, u.z, u.w};
                if (T.f16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[q] & 0xffffu));
                        v[2 * q + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[q] >> 16));
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
                }
                if (T.scale) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], scv[j], shv[j]);
                }
            };
            if (!up) {
                float v[8];
                cvt(raw[k][0], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            } else {
                float a[8], b[8], c[8], d[8];
                cvt(raw[k][0], a); cvt(raw[k][1], b); cvt(raw[k][2], c); cvt(raw[k][3], d);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += hy[k] * (hx[k] * a[j] + lx[k] * b[j]) + ly[k] * (hx[k] * c[j] + lx[k] * d[j]);
            }
        }
        unsigned short oh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) oh[j] = f2bf(A.relu ? fmaxf(acc[j], 0.f) : acc[j]);
        uint4 ou;
        ou.x = oh[0] | ((unsigned)oh[1] << 16); ou.y = oh[2] | ((unsigned)oh[3] << 16);
        ou.z = oh[4] | ((unsigned)oh[5] << 16); ou.w = oh[6] | ((unsigned)oh[7] << 16);
        *reinterpret_cast<uint4 *>(A.out + (size_t)pix * A.out_cstride + A.out_coff + c0) = ou;
    }
}

// transpose of the bilinear up-sampling inside fuse_sum (gather form, deterministic): din[ys][xs] = sum over the output
// pixels (y, x) whose interpolation reads (ys, xs) of their weight x dout[y][x].  dout may be a channel slice.
template <bool F32>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const unsigned short *__restrict__ dout, int N, int H, int W, int C, int cstride,
                                                           int coff, int Hs, int Ws, unsigned short *__restrict__ din) {
    const int VPP = C / 8;
    const size_t total = (size_t)N * Hs * Ws * VPP;
    const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;
    const int ry = (H + Hs - 1) / Hs, rx = (W + Ws - 1) / Ws;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % VPP);
        const size_t pix = i / VPP;
        const int xs = (int)(pix % Ws), ys = (int)((pix / Ws) % Hs), n = (int)(pix / ((size_t)Ws * Hs));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const int ylo = (ys - 1) * ry < 0 ? 0 : (ys - 1) * ry, yhi = (ys + 2) * ry > H ? H : (ys + 2) * ry;
        const int xlo = (xs - 1) * rx < 0 ? 0 : (xs - 1) * rx, xhi = (xs + 2) * rx > W ? W : (xs + 2) * rx;
        for (int y = ylo; y < yhi; ++y) {
            float fy = ((float)y + 0.5f) * sy - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            const int y0 = (int)fy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
            const float ly = fy - (float)y0;
            const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x < xhi; ++x) {
                float fx = ((float)x + 0.5f) * sx - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
                const float lx = fx - (float)x0;
                const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
                if (wx == 0.f) continue;
                float v[8];
                if (F32) {
                    const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(dout) + (((size_t)n * H + y) * W + x) * cstride + coff + slot * 8);
                    const float4 a = q[0], b = q[1];
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
                    bf8_to_f32(*reinterpret_cast<const uint4 *>(dout + (((size_t)n * H + y) * W + x) * cstride + coff + slot * 8), v);
                }
                const float wgt = wy * wx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(wgt, v[j], acc[j]);
            }
        }
        if (F32) {
            float4 *o = reinterpret_cast<float4 *>(reinterpret_cast<float *>(din) + pix * C + slot * 8);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            continue;
        }
        unsigned short oh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) oh[j] = f2bf(acc[j]);
        uint4 ou;
        ou.x = oh[0] | ((unsigned)oh[1] << 16); ou.y = oh[2] | ((unsigned)oh[3] << 16);
        ou.z = oh[4] | ((unsigned)oh[5] << 16); ou.w = oh[6] | ((unsigned)oh[7] << 16);
        *reinterpret_cast<uint4 *>(din + pix * C + slot * 8) = ou;
    }
}

// space-to-depth gradient [N][H2][W2][(a, b, c)] -> NHWC [N][2*H2][2*W2][C] (the input gradient of a stride-2 convolution
// computed through the space-to-depth view)
template <typename VT>      // one 16-byte vector per thread: 8 16-bit or 4 fp32 channels
__global__ __launch_bounds__(256) void s2d_to_nhwc_kernel(const unsigned short *__restrict__ in_, int N, int H2, int W2, int C,
                                                          unsigned short *__restrict__ out_) {
    constexpr int EPV = 16 / sizeof(VT);
    const VT *in = reinterpret_cast<const VT *>(in_);
    VT *out = reinterpret_cast<VT *>(out_);
    const int VPP = C / EPV;
    const size_t total = (size_t)N * H2 * W2 * 4 * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % VPP);
        size_t r = i / VPP;
        const int ab = (int)(r & 3); r >>= 2;
        const int x2 = (int)(r % W2), y2 = (int)((r / W2) % H2), n = (int)(r / ((size_t)W2 * H2));
        const int a = ab >> 1, b = ab & 1;
        const uint4 v = *reinterpret_cast<const uint4 *>(in + (((size_t)n * H2 + y2) * W2 + x2) * 4 * C + (size_t)ab * C + slot * EPV);
        *reinterpret_cast<uint4 *>(out + (((size_t)n * 2 * H2 + 2 * y2 + a) * 2 * W2 + 2 * x2 + b) * C + slot * EPV) = v;
    }
}

// gradient of a fuse / residual sum: out = [mask > 0] * (sum of up to 6 gradient contributions, each possibly a channel slice)
struct GradSumArgs { const unsigned short *g[6]; int cstride[6], coff[6]; int nterm; const unsigned short *mask; size_t npix; int C; unsigned short *out; };

template <bool F32>
__global__ __launch_bounds__(256) void grad_sum_kernel(const GradSumArgs A) {
    const int VPP = A.C / 8;
    const size_t total = A.npix * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % VPP);
        const size_t pix = i / VPP;
        float acc[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        auto load8 = [&](const unsigned short *base, size_t e) {
            if (F32) {
                const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + e);
                const float4 a = p[0], b = p[1];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
                bf8_to_f32(*reinterpret_cast<const uint4 *>(base + e), v);
            }
        };
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k < A.nterm) {
                load8(A.g[k], pix * A.cstride[k] + A.coff[k] + slot * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        }
        if (A.mask) {
            load8(A.mask, pix * A.C + slot * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = v[j] > 0.f ? acc[j] : 0.f;
        }
        if (F32) {
            float4 *o = reinterpret_cast<float4 *>(reinterpret_cast<float *>(A.out) + pix * A.C + slot * 8);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
            unsigned short oh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) oh[j] = f2bf(acc[j]);
            uint4 ou;
            ou.x = oh[0] | ((unsigned)oh[1] << 16); ou.y = oh[2] | ((unsigned)oh[3] << 16);
            ou.z = oh[4] | ((unsigned)oh[5] << 16); ou.w = oh[6] | ((unsigned)oh[7] << 16);
            *reinterpret_cast<uint4 *>(A.out + pix * A.C + slot * 8) = ou;
        }
    }
}

}  // namespace

extern "C" int cdnet_input_pack(const float *x, int N, int C, int H, int W, void *out, void *stream) {
    CDNET_REQUIRE(x && out, "cdnet_input_pack: null pointer");
    CDNET_REQUIRE(N > 0 && C > 0 && C <= 16 && H > 0 && W > 0, "cdnet_input_pack: bad size (C=%d must be <= 16)", C);
    input_pack_kernel<<<lin_grid((size_t)N * H * W), 256, 0, (hipStream_t)stream>>>(x, N, C, H * W, (unsigned short *)out);
    return check_launch("cdnet_input_pack");
}

extern "C" int cdnet_input_pack_f32(const float *x, int N, int C, int H, int W, float *out, void *stream) {
    CDNET_REQUIRE(x && out, "cdnet_input_pack_f32: null pointer");
    CDNET_REQUIRE(N > 0 && C > 0 && C <= 16 && H > 0 && W > 0, "cdnet_input_pack_f32: bad size (C=%d must be <= 16)", C);
    input_pack_f32_kernel<<<lin_grid((size_t)N * H * W), 256, 0, (hipStream_t)stream>>>(x, N, C, H * W, out);
    return check_launch("cdnet_input_pack_f32");
}

extern "C" int cdnet_bn_fold_eval(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                                  const float *conv_bias, float eps, int C, float *scale, float *shift, void *stream) {
    CDNET_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "cdnet_bn_fold_eval: bad args");
    bn_fold_eval_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, conv_bias, eps, C,
                                                                      scale, shift);
    return check_launch("cdnet_bn_fold_eval");
}

extern "C" int cdnet_bn_finalize_train(const float *stats, int T, int C, float count, const float *gamma, const float *beta,
                                       const float *conv_bias, float eps, float momentum, float *running_mean,
                                       float *running_var, float *scale, float *shift, float *save_mean,
                                       float *save_invstd, void *stream) {
    CDNET_REQUIRE(stats && gamma && beta && scale && shift && T > 0 && C > 0 && count > 0, "cdnet_bn_finalize_train: bad args");
    bn_finalize_train_kernel<<<C, 256, 0, (hipStream_t)stream>>>(stats, T, C, count, gamma, beta, conv_bias, eps, momentum,
                                                                         running_mean, running_var, scale, shift, save_mean,
                                                                         save_invstd);
    return check_launch("cdnet_bn_finalize_train");
}

static HeadFeat mk_feat(const cdnet_head_feat &f) {
    HeadFeat h;
    h.raw = f.raw; h.res = f.res; h.scale = f.scale; h.shift = f.shift; h.relu = f.relu; h.f16 = f.f16;
    return h;
}

extern "C" int cdnet_dam_head_forward(const cdnet_head_feat *f1, const cdnet_head_feat *f2, const cdnet_head_feat *f3,
                                      const float *head_weights, int N, int H, int W, float *mask, float *point,
                                      float *direction, void *stream) {
    CDNET_REQUIRE(f1 && f2 && f3 && head_weights && mask && point && direction, "cdnet_dam_head_forward: null pointer");
    CDNET_REQUIRE(f1->raw && f2->raw && N > 0 && H > 0 && W > 0, "cdnet_dam_head_forward: bad args");
    static_assert(sizeof(HeadW) == CDNET_HEAD_WEIGHT_FLOATS * 4, "head weight block layout");
    const HeadFeat a = mk_feat(*f1), b = mk_feat(*f2), c = mk_feat(*f3);
    auto plain = [](const HeadFeat &f) { return f.f16 == 0 && !f.scale && !f.relu && !f.res; };
    auto train = [](const HeadFeat &f) { return f.f16 == 1 && f.scale && f.relu && f.res; };
    const int grid = lin_grid((size_t)N * H * W * 4);
    const HeadW *hw = reinterpret_cast<const HeadW *>(head_weights);
    // plain stored features (eval mode, the fused training forward): the matrix-core kernel
    const size_t total = (size_t)N * H * W;
    const int mgrid = (int)std::min<size_t>(2048, (total + 127) / 128);
    if (!c.raw) {
        // no third feature: `point` is an input (the point logit left by the producing convolution, cdnet_conv_args.dot_out)
        CDNET_REQUIRE(plain(a) && plain(b), "cdnet_dam_head_forward: f3->raw = NULL (point given) needs plain bf16 f1 / f2");
        dam_head_mfma_kernel<false, false><<<mgrid, 256, 0, (hipStream_t)stream>>>(a.raw, b.raw, nullptr, hw, total, H * W, mask, point, direction);
        return check_launch("cdnet_dam_head_forward");
    }
    if (plain(a) && plain(b) && plain(c)) {
        dam_head_mfma_kernel<true, false><<<mgrid, 256, 0, (hipStream_t)stream>>>(a.raw, b.raw, c.raw, hw, total, H * W, mask, point, direction);
        return check_launch("cdnet_dam_head_forward");
    }
    auto plain32 = [](const HeadFeat &f) { return f.f16 == 2 && !f.scale && !f.relu && !f.res; };
    if (plain32(a) && plain32(b) && plain32(c)) {
        const int g32 = (int)std::min<size_t>(4096, (total + 63) / 64);
        dam_head_mfma_kernel<true, true><<<g32, 256, 0, (hipStream_t)stream>>>(a.raw, b.raw, c.raw, hw, total, H * W, mask, point, direction);
        return check_launch("cdnet_dam_head_forward");
    }
    if (train(a) && train(b) && train(c)) dam_head_fwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a, b, c, hw, N, H * W, mask, point, direction);
    else dam_head_fwd_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(a, b, c, hw, N, H * W, mask, point, direction);
    return check_launch("cdnet_dam_head_forward");
}

extern "C" int cdnet_final_conv1x1(const cdnet_head_feat *f, const float *w, const float *b, int K, int N, int H, int W,
                                   float *out, void *stream) {
    CDNET_REQUIRE(f && f->raw && w && b && out, "cdnet_final_conv1x1: null pointer");
    CDNET_REQUIRE(K >= 1 && K <= 32 && N > 0 && H > 0 && W > 0, "cdnet_final_conv1x1: K=%d must be in [1,32]", K);
    final_conv1x1_kernel<<<lin_grid((size_t)N * H * W), 256, 0, (hipStream_t)stream>>>(mk_feat(*f), w, b, K, N, H * W, out);
    return check_launch("cdnet_final_conv1x1");
}

extern "C" int cdnet_window_pack(const float *img, int C, int H, int W, int view_xform, int tile_h, int tile_w, int stride, int ny,
                                 int nx, void *out_bf16_nhwc16, void *stream) {
    CDNET_REQUIRE(img && out_bf16_nhwc16, "cdnet_window_pack: null pointer");
    CDNET_REQUIRE(C > 0 && C <= 16 && H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && stride > 0 && ny > 0 && nx > 0 &&
                  view_xform >= 0 && view_xform < 8, "cdnet_window_pack: bad arguments");
    window_pack_kernel<<<lin_grid((size_t)ny * nx * tile_h * tile_w), 256, 0, (hipStream_t)stream>>>(
        img, C, H, W, view_xform, tile_h, tile_w, stride, ny, nx, (unsigned short *)out_bf16_nhwc16);
    return check_launch("cdnet_window_pack");
}

extern "C" int cdnet_window_pack_f32(const float *img, int C, int H, int W, int view_xform, int tile_h, int tile_w, int stride, int ny,
                                     int nx, float *out_f32_nhwc16, void *stream) {
    CDNET_REQUIRE(img && out_f32_nhwc16, "cdnet_window_pack_f32: null pointer");
    CDNET_REQUIRE(C > 0 && C <= 16 && H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && stride > 0 && ny > 0 && nx > 0 &&
                  view_xform >= 0 && view_xform < 8, "cdnet_window_pack_f32: bad arguments");
    window_pack_f32_kernel<<<lin_grid((size_t)ny * nx * tile_h * tile_w), 256, 0, (hipStream_t)stream>>>(
        img, C, H, W, view_xform, tile_h, tile_w, stride, ny, nx, out_f32_nhwc16);
    return check_launch("cdnet_window_pack_f32");
}

extern "C" int cdnet_window_stitch(const float *tiles, int K, int tile_h, int tile_w, int stride, int overlap, int ny, int nx, int Hv,
                                   int Wv, float *out, void *stream) {
    CDNET_REQUIRE(tiles && out, "cdnet_window_stitch: null pointer");
    CDNET_REQUIRE(K > 0 && tile_h > 0 && tile_w > 0 && stride > 0 && ny > 0 && nx > 0 && Hv > 0 && Wv > 0 && overlap >= 0,
                  "cdnet_window_stitch: bad arguments");
    CDNET_REQUIRE((ny - 1) * stride + tile_h >= Hv && (nx - 1) * stride + tile_w >= Wv, "cdnet_window_stitch: windows do not cover the view");
    window_stitch_kernel<<<lin_grid((size_t)K * Hv * Wv), 256, 0, (hipStream_t)stream>>>(tiles, K, tile_h, tile_w, stride, overlap / 2,
                                                                                         ny, nx, Hv, Wv, out);
    return check_launch("cdnet_window_stitch");
}


static int fuse_sum_impl(const cdnet_fuse_term *terms, int nterm, int N, int H, int W, int C, int relu, void *out, int out_cstride, int out_coff,
                         void *stream, bool f32) {
    CDNET_REQUIRE(terms && out && nterm >= 1 && nterm <= 4, "cdnet_fuse_sum: 1..4 terms");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "cdnet_fuse_sum: bad size (C %% 8)");
    FuseArgs A;
    for (int k = 0; k < 4; ++k) {
        if (k < nterm) {
            CDNET_REQUIRE(terms[k].x && terms[k].Hs > 0 && terms[k].Ws > 0 && terms[k].Hs <= H && terms[k].Ws <= W, "cdnet_fuse_sum: term %d", k);
            CDNET_REQUIRE((terms[k].f16 == 2) == f32, "cdnet_fuse_sum: term %d storage %d does not match the %s entry point", k, terms[k].f16,
                          f32 ? "fp32" : "16-bit");
            A.t[k].x = terms[k].x; A.t[k].Hs = terms[k].Hs; A.t[k].Ws = terms[k].Ws;
            A.t[k].scale = terms[k].scale; A.t[k].shift = terms[k].shift; A.t[k].f16 = terms[k].f16;
            CDNET_REQUIRE((terms[k].scale == nullptr) == (terms[k].shift == nullptr), "cdnet_fuse_sum: scale and shift come together");
        } else { A.t[k].x = nullptr; A.t[k].Hs = A.t[k].Ws = 0; A.t[k].scale = A.t[k].shift = nullptr; A.t[k].f16 = 0; }
    }
    A.nterm = nterm; A.N = N; A.H = H; A.W = W; A.C = C; A.relu = relu; A.out = reinterpret_cast<unsigned short *>(out);
    A.out_cstride = out_cstride ? out_cstride : C; A.out_coff = out_coff;
    CDNET_REQUIRE(A.out_cstride % 8 == 0 && out_coff % 8 == 0 && out_coff + C <= A.out_cstride, "cdnet_fuse_sum: output channel slice");
    const size_t total = (size_t)N * H * W * (C / 8);
    const int grid = lin_grid(total);
    if (f32) fuse_sum_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(A);
    else if (total < (1ull << 31)) {
        switch (nterm) {
            case 1: fuse_sum16_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(A); break;
            case 2: fuse_sum16_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(A); break;
            case 3: fuse_sum16_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(A); break;
            default: fuse_sum16_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(A); break;
        }
    } else fuse_sum_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(A);
    return check_launch("cdnet_fuse_sum");
}

extern "C" int cdnet_fuse_sum(const cdnet_fuse_term *terms, int nterm, int N, int H, int W, int C, int relu, uint16_t *out, int out_cstride,
                              int out_coff, void *stream) {
    return fuse_sum_impl(terms, nterm, N, H, W, C, relu, out, out_cstride, out_coff, stream, false);
}

extern "C" int cdnet_fuse_sum_f32(const cdnet_fuse_term *terms, int nterm, int N, int H, int W, int C, int relu, float *out, int out_cstride,
                                  int out_coff, void *stream) {
    return fuse_sum_impl(terms, nterm, N, H, W, C, relu, out, out_cstride, out_coff, stream, true);
}

static int upsample_bwd_impl(const void *dout, int N, int H, int W, int C, int dout_cstride, int dout_coff, int Hs, int Ws, void *din, void *stream,
                             bool f32) {
    CDNET_REQUIRE(dout && din && N > 0 && H >= Hs && W >= Ws && Hs > 0 && Ws > 0 && C % 8 == 0 && C >= 8, "cdnet_upsample_bilinear_backward: bad args");
    const int cs = dout_cstride ? dout_cstride : C;
    CDNET_REQUIRE(cs % 8 == 0 && dout_coff % 8 == 0 && dout_coff + C <= cs, "cdnet_upsample_bilinear_backward: channel slice");
    const unsigned short *d = reinterpret_cast<const unsigned short *>(dout);
    unsigned short *o = reinterpret_cast<unsigned short *>(din);
    if (f32) upsample_bwd_kernel<true><<<lin_grid((size_t)N * Hs * Ws * (C / 8)), 256, 0, (hipStream_t)stream>>>(d, N, H, W, C, cs, dout_coff, Hs, Ws, o);
    else upsample_bwd_kernel<false><<<lin_grid((size_t)N * Hs * Ws * (C / 8)), 256, 0, (hipStream_t)stream>>>(d, N, H, W, C, cs, dout_coff, Hs, Ws, o);
    return check_launch("cdnet_upsample_bilinear_backward");
}

extern "C" int cdnet_upsample_bilinear_backward(const uint16_t *dout, int N, int H, int W, int C, int dout_cstride, int dout_coff, int Hs, int Ws,
                                                uint16_t *din, void *stream) {
    return upsample_bwd_impl(dout, N, H, W, C, dout_cstride, dout_coff, Hs, Ws, din, stream, false);
}

extern "C" int cdnet_upsample_bilinear_backward_f32(const float *dout, int N, int H, int W, int C, int dout_cstride, int dout_coff, int Hs, int Ws,
                                                    float *din, void *stream) {
    return upsample_bwd_impl(dout, N, H, W, C, dout_cstride, dout_coff, Hs, Ws, din, stream, true);
}

extern "C" int cdnet_s2d_to_nhwc(const uint16_t *in, int N, int H2, int W2, int C, uint16_t *out, void *stream) {
    CDNET_REQUIRE(in && out && N > 0 && H2 > 0 && W2 > 0 && C % 8 == 0 && C >= 8, "cdnet_s2d_to_nhwc: bad args");
    s2d_to_nhwc_kernel<unsigned short><<<lin_grid((size_t)N * H2 * W2 * 4 * (C / 8)), 256, 0, (hipStream_t)stream>>>(in, N, H2, W2, C, out);
    return check_launch("cdnet_s2d_to_nhwc");
}

extern "C" int cdnet_s2d_to_nhwc_f32(const float *in, int N, int H2, int W2, int C, float *out, void *stream) {
    CDNET_REQUIRE(in && out && N > 0 && H2 > 0 && W2 > 0 && C % 4 == 0 && C >= 4, "cdnet_s2d_to_nhwc_f32: bad args");
    s2d_to_nhwc_kernel<float><<<lin_grid((size_t)N * H2 * W2 * 4 * (C / 4)), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const unsigned short *>(in), N, H2, W2, C, reinterpret_cast<unsigned short *>(out));
    return check_launch("cdnet_s2d_to_nhwc_f32");
}

static int grad_sum_impl(const cdnet_grad_term *terms, int nterm, const uint16_t *mask, long long npix, int C, uint16_t *out, void *stream, bool f32);
extern "C" int cdnet_grad_sum(const cdnet_grad_term *terms, int nterm, const uint16_t *mask, long long npix, int C, uint16_t *out, void *stream) {
    return grad_sum_impl(terms, nterm, mask, npix, C, out, stream, false);
}
extern "C" int cdnet_grad_sum_f32(const cdnet_grad_term *terms, int nterm, const float *mask, long long npix, int C, float *out, void *stream) {
    return grad_sum_impl(terms, nterm, reinterpret_cast<const uint16_t *>(mask), npix, C, reinterpret_cast<uint16_t *>(out), stream, true);
}
static int grad_sum_impl(const cdnet_grad_term *terms, int nterm, const uint16_t *mask, long long npix, int C, uint16_t *out, void *stream, bool f32) {
    CDNET_REQUIRE(terms && out && nterm >= 1 && nterm <= 6 && npix > 0 && C % 8 == 0 && C >= 8, "cdnet_grad_sum: bad args");
    GradSumArgs A;
    for (int k = 0; k < 6; ++k) { A.g[k] = nullptr; A.cstride[k] = C; A.coff[k] = 0; }
    for (int k = 0; k < nterm; ++k) {
        A.g[k] = terms[k].g;
        A.cstride[k] = terms[k].cstride ? terms[k].cstride : C;
        A.coff[k] = terms[k].coff;
        CDNET_REQUIRE(A.g[k] && A.cstride[k] % 8 == 0 && A.coff[k] % 8 == 0 && A.coff[k] + C <= A.cstride[k], "cdnet_grad_sum: channel slice");
    }
    A.nterm = nterm; A.mask = mask; A.npix = (size_t)npix; A.C = C; A.out = out;
    if (f32) grad_sum_kernel<true><<<lin_grid((size_t)npix * (C / 8)), 256, 0, (hipStream_t)stream>>>(A);
    else grad_sum_kernel<false><<<lin_grid((size_t)npix * (C / 8)), 256, 0, (hipStream_t)stream>>>(A);
    return check_launch("cdnet_grad_sum");
}
