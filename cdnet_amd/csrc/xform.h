// Packed-math forms of the fused input transform (producer BatchNorm scale/shift [+ residual] + ReLU on an fp16 raw
// tensor -> bf16 MFMA operand) shared by the forward/backward-data convolution and the backward-weight kernels.
// 20 VALU instructions per 8 channels: 8 v_cvt_f32_f16, 4 v_pk_fma_f32, 4 v_cvt_pk_bf16_f32, 4 v_pk_max_i16 (the ReLU is
// taken on the bf16 bit patterns: rounding keeps the sign, and a negative bf16 is a negative int16).
#pragma once
#include <hip/hip_runtime.h>

namespace cdnet {

typedef unsigned xf_u32x4 __attribute__((ext_vector_type(4)));
typedef float xf_f32x2 __attribute__((ext_vector_type(2)));
typedef short xf_s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xf_h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 xf_bf16x2 __attribute__((ext_vector_type(2)));

union XfWords {               // (element access through a union: subscripting a by-value ext-vector parameter in an
    xf_u32x4 u;               //  unrolled loop was seen to collapse to one element on ROCm 7.2)
    unsigned w[4];
};

template <bool RES>
__device__ __forceinline__ xf_u32x4 xf_bnrelu_f16(xf_u32x4 raw, xf_u32x4 res, const float *sc, const float *sh) {
    XfWords I, R, O;
    I.u = raw;
    R.u = res;
    const xf_s16x2 zero = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const xf_f32x2 x = __builtin_convertvector(__builtin_bit_cast(xf_h16x2, I.w[k]), xf_f32x2);
        const xf_f32x2 s = {sc[2 * k], sc[2 * k + 1]}, b = {sh[2 * k], sh[2 * k + 1]};
        xf_f32x2 t = __builtin_elementwise_fma(x, s, b);
        if (RES) t = t + __builtin_convertvector(__builtin_bit_cast(xf_h16x2, R.w[k]), xf_f32x2);
        const xf_bf16x2 r = __builtin_convertvector(t, xf_bf16x2);
        O.w[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, r), zero));
    }
    return O.u;
}

// element-wise maximum of two vectors of NON-NEGATIVE bf16 values (post-ReLU): their int16 order is their float order.
// Ties keep either (identical bit patterns).
__device__ __forceinline__ xf_u32x4 xf_max_nonneg_bf8(xf_u32x4 a, xf_u32x4 b) {
    XfWords A, B, O;
    A.u = a;
    B.u = b;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        O.w[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, A.w[k]), __builtin_bit_cast(xf_s16x2, B.w[k])));
    return O.u;
}



// eval-mode residual unit output: relu(raw + res), both fp16, no affine (the BatchNorm is folded into the producer) -> bf16
__device__ __forceinline__ xf_u32x4 xf_addrelu_f16(xf_u32x4 raw, xf_u32x4 res) {
    XfWords I, R, O;
    I.u = raw;
    R.u = res;
    const xf_s16x2 zero = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const xf_f32x2 t = __builtin_convertvector(__builtin_bit_cast(xf_h16x2, I.w[k]), xf_f32x2) +
                           __builtin_convertvector(__builtin_bit_cast(xf_h16x2, R.w[k]), xf_f32x2);
        const xf_bf16x2 r = __builtin_convertvector(t, xf_bf16x2);
        O.w[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, r), zero));
    }
    return O.u;
}

__device__ __forceinline__ void xf_addrelu_f16_to_f32(xf_u32x4 raw, xf_u32x4 res, float *out) {
    XfWords O;
    O.u = xf_addrelu_f16(raw, res);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out[2 * k] = __builtin_bit_cast(float, O.w[k] << 16);
        out[2 * k + 1] = __builtin_bit_cast(float, O.w[k] & 0xffff0000u);
    }
}

// the same transform, result widened to 8 floats (the values the MFMA kernels see: bf16-rounded)
template <bool RES>
__device__ __forceinline__ void xf_bnrelu_f16_to_f32(xf_u32x4 raw, xf_u32x4 res, const float *sc, const float *sh, float *out) {
    XfWords O;
    O.u = xf_bnrelu_f16<RES>(raw, res, sc, sh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out[2 * k] = __builtin_bit_cast(float, O.w[k] << 16);
        out[2 * k + 1] = __builtin_bit_cast(float, O.w[k] & 0xffff0000u);
    }
}

// sum over the 4 lanes of a quad with two DPP moves (no LDS permute)
__device__ __forceinline__ float xf_quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // lanes ^ 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // lanes ^ 2
    return v;
}


// 16-term dot product with packed fp32 FMAs (v_pk_fma_f32: two lanes of the sum per instruction)
__device__ __forceinline__ float xf_dot16(const float *__restrict__ w, const float *__restrict__ v) {
    xf_f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const xf_f32x2 a = {w[2 * c], w[2 * c + 1]}, b = {v[2 * c], v[2 * c + 1]};
        acc = __builtin_elementwise_fma(a, b, acc);
    }
    return acc[0] + acc[1];
}

// out[c] (+)= s * w[c] for 16 channels, two per instruction
__device__ __forceinline__ void xf_axpy16(float s, const float *__restrict__ w, float *__restrict__ out, bool accumulate) {
    const xf_f32x2 ss = {s, s};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const xf_f32x2 a = {w[2 * c], w[2 * c + 1]};
        xf_f32x2 o = {accumulate ? out[2 * c] : 0.f, accumulate ? out[2 * c + 1] : 0.f};
        o = __builtin_elementwise_fma(ss, a, o);
        out[2 * c] = o[0];
        out[2 * c + 1] = o[1];
    }
}

}  // namespace cdnet
