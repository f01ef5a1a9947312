// Staging helpers shared by the 16-bit convolution kernels (conv.hip, conv16ws.hip): storage-format loads, the generic fused source
// transform of 8 channels (producer BatchNorm scale / shift, residual operand, ReLU -> bf16 MFMA operand).
#pragma once
#include <hip/hip_runtime.h>
#include "xform.h"

namespace cdnet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

static __device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
static __device__ __forceinline__ float h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
static __device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
// storage formats: 0 = bf16 (activated tensors, gradients: MFMA operands), 1 = fp16 (raw pre-BatchNorm outputs and
// residual branches: the consumer's affine must see more than bf16's 8 significant bits when |mean| >> std)
static __device__ __forceinline__ float ld16(unsigned short u, bool f16) { return f16 ? h2f(u) : bf2f(u); }
static __device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

union V16 {
    uint4 u;
    unsigned short h[8];
    bf16x8 v;
};

struct ChanXf {           // per-thread channel transform for its 8 channels of the current chunk
    float sc[8], sh[8];
    bool on;
};

static __device__ __forceinline__ V16 xform8(V16 raw, const V16 *res, const ChanXf &t, bool relu, bool f16) {
    V16 o;
    if (f16 && !t.on && relu && res) {           // eval-mode residual unit: relu(raw + res), packed math
        o.u = __builtin_bit_cast(uint4, xf_addrelu_f16(__builtin_bit_cast(xf_u32x4, raw.u), __builtin_bit_cast(xf_u32x4, res->u)));
        return o;
    }
    if (f16 && t.on && relu) {                   // the training-mode combination: packed math (xform.h)
        const xf_u32x4 r = __builtin_bit_cast(xf_u32x4, raw.u);
        const xf_u32x4 v = res ? xf_bnrelu_f16<true>(r, __builtin_bit_cast(xf_u32x4, res->u), t.sc, t.sh)
                               : xf_bnrelu_f16<false>(r, r, t.sc, t.sh);
        o.u = __builtin_bit_cast(uint4, v);
        return o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = ld16(raw.h[j], f16);
        if (t.on) v = fmaf(v, t.sc[j], t.sh[j]);
        if (res) v += ld16(res->h[j], f16);
        if (relu) v = fmaxf(v, 0.f);
        o.h[j] = f2bf(v);
    }
    return o;
}


}  // namespace cdnet
