// Micro-benchmark: bf16 MFMA shapes under the chip's own clock management (MI355X_MICROARCH.md, DVFS give-back item 7): a consumer-like
// loop - every operand re-read from LDS by ds_read_b128, random bf16 data, 64 x 64 outputs per wave - with v_mfma_f32_32x32x16_bf16
// (2 x 2 blocks: 4 reads per 4 MFMAs) against v_mfma_f32_16x16x32_bf16 (4 x 4 blocks: 8 reads per 16 MFMAs): the same FLOP per LDS byte.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o tools/micro/mfma_shapes && tools/micro/mfma_shapes [waves_per_cu] [zero]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void loop(const unsigned *__restrict__ seed, float *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = seed[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = smem + (wave & 3) * 1024 + lane * 16;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
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
        if (s == 123.456f) sink[0] = s;
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
            const unsigned char *p = base + (i & 3) * 4096;
            bf16x8 A[4], B[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { A[k] = *reinterpret_cast<const bf16x8 *>(p + k * 4096); B[k] = *reinterpret_cast<const bf16x8 *>(p + 32768 + k * 4096); }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a], B[b], acc[a][b], 0, 0, 0);
        }
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
        if (s == 123.456f) sink[0] = s;
    }
}

int main(int argc, char **argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 4;
    const bool zero = argc > 2 && atoi(argv[2]);
    unsigned *seed; float *sink;
    hipMalloc(&seed, 65536); hipMalloc(&sink, 4);
    unsigned h[16384];
    srand(1);
    for (int i = 0; i < 16384; ++i) {
        // two random bf16 in [-1, 1): sign, exponent 120..126, random mantissa
        unsigned a = (rand() & 1) << 15 | (120 + rand() % 7) << 7 | (rand() & 127), b = (rand() & 1) << 15 | (120 + rand() % 7) << 7 | (rand() & 127);
        h[i] = zero ? 0u : (a | (b << 16));
    }
    hipMemcpy(seed, h, 65536, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(loop<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(loop<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 16}) {
            const int iters = shape == 32 ? 40000 : 10000;          // the same FLOPs: 4 x 32768 vs 16 x 16384 per iteration x 4
            const double flop = 256.0 * wpc * (double)iters * (shape == 32 ? 4 * 32768.0 : 16 * 16384.0);
            auto run = [&]() { if (shape == 32) loop<32><<<256, wpc * 64, 65536>>>(seed, sink, iters); else loop<16><<<256, wpc * 64, 65536>>>(seed, sink, iters); };
            auto t0 = std::chrono::steady_clock::now();
            int n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.5) { run(); hipDeviceSynchronize(); ++n; }   // settle
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s operands, %d waves/CU, v_mfma_f32_%s_bf16: %.3f ms per launch, %.0f TFLOP/s\n", zero ? "zero" : "random", wpc,
                   shape == 32 ? "32x32x16" : "16x16x32", ms / 20, flop / (ms / 20 * 1e-3) / 1e12);
        }
    return 0;
}
