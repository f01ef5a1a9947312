// Box calibration (round 6): two micro-measurements that say what THIS MI355X box grants, so that rates measured on different boxes of
// the pool can be normalised (bench.py's `box` object).  Diagnostics of the runtime, nothing of the reference.
//   cdnet_box_copy : float4 grid-stride copy (the guide's HBM stream: 6.0-6.3 TB/s of the 8 TB/s peak on a good box)
//   cdnet_box_mfma : v_mfma_f32_32x32x16_bf16 loop, every operand re-read from LDS (ds_read_b128), random or caller-given bf16 data, one
//                    8-wave workgroup per CU; each workgroup stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) around its
//                    loop: the in-kernel clock the chip holds under matrix load = d(memtime) / d(memrealtime) x 100 MHz
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

__global__ __launch_bounds__(256) void box_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

// LDS: 64 KB of operand patterns; per iteration a wave reads two A and two B fragments (ds_read_b128 each) and issues four MFMAs
__global__ __launch_bounds__(512) void box_mfma_kernel(const unsigned *__restrict__ seed, float *sink, unsigned long long *stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = seed[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = smem + (wave & 3) * 1024 + lane * 16;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) alone: the stamps are back before the loop's LDS reads
    for (int i = 0; i < iters; ++i) {
        const unsigned char *p = base + (i & 7) * 4096;
        bf16x8 A0 = *reinterpret_cast<const bf16x8 *>(p), A1 = *reinterpret_cast<const bf16x8 *>(p + 8192);
        bf16x8 B0 = *reinterpret_cast<const bf16x8 *>(p + 16384), B1 = *reinterpret_cast<const bf16x8 *>(p + 24576);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && stamps) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace

extern "C" int cdnet_box_copy(const void *src, void *dst, size_t bytes, void *stream) {
    CDNET_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, "cdnet_box_copy: null pointer or size not a multiple of 16");
    CDNET_REQUIRE(((size_t)src & 15) == 0 && ((size_t)dst & 15) == 0, "cdnet_box_copy: 16-byte aligned buffers");
    box_copy_kernel<<<256 * 16, 256, 0, (hipStream_t)stream>>>((const float4 *)src, (float4 *)dst, bytes / 16);
    return cdnet::check_launch("cdnet_box_copy");
}

extern "C" int cdnet_box_mfma(const uint32_t *seed, float *sink, unsigned long long *stamps, int workgroups, int waves_per_wg, int iters,
                              void *stream) {
    CDNET_REQUIRE(seed && sink, "cdnet_box_mfma: null pointer");
    CDNET_REQUIRE(workgroups >= 1 && workgroups <= 4096 && waves_per_wg >= 1 && waves_per_wg <= 8 && iters >= 1, "cdnet_box_mfma: bad sizes");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(box_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess)
            return cdnet::check_launch("hipFuncSetAttribute(box_mfma)");
        attr = true;
    }
    box_mfma_kernel<<<workgroups, waves_per_wg * 64, 65536, (hipStream_t)stream>>>(seed, sink, stamps, iters);
    return cdnet::check_launch("cdnet_box_mfma");
}
