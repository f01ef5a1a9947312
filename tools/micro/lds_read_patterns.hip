// Micro-benchmark: LDS-array cycles of ds_read_b128 for the A-fragment address patterns of the 16x16-pixel convolution tile
// (lane l: pixel m = l & 31 of a 32-pixel M block = two tile rows of 16, k-half l >> 5).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read_patterns.hip -o /tmp/lds_read_patterns && /tmp/lds_read_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ int pattern(int pat, int lane) {
    const int m = lane & 31, h = lane >> 5, row = m / 16, col = m % 16;
    switch (pat) {
    case 0: return lane * 16;                                                   // contiguous
    case 1: return (row * 18 + col) * 48 + h * 16;                              // padded 48-byte pixels, 18-pixel rows
    case 2: return (row * 18 + col) * 32 + ((h ^ (row & 1)) * 16);              // 32-byte pixels, k-half swizzled by row parity
    case 3: return row * 1024 + col * 48 + h * 16;                              // 48-byte pixels, 1024-byte row pitch
    case 4: return (row * 18 + col) * 32 + h * 16;                              // 32-byte pixels, no swizzle
    case 5: return (row * 18 + col) * 32 + ((h ^ (col & 1)) * 16);              // k-half swizzled by column parity
    case 6: return (row * 18 + col) * 32 + ((h ^ ((col >> 1) & 1)) * 16);
    case 7: return (row * 18 + col) * 32 + ((h ^ ((col >> 2) & 1)) * 16);
    case 8: return (row * 18 + col) * 32 + ((h ^ ((col >> 3) & 1)) * 16);
    case 9: return (row * 18 + col) * 80 + h * 16;                              // 80-byte pixels
    case 10: return (row * 18 + col) * 32 + ((h ^ ((col >> 2) & 1) ^ (row & 1)) * 16);
    case 11: return (row * 18 + col) * 32 + ((h ^ ((col >> 3) & 1) ^ (row & 1)) * 16);
    default: return 0;
    }
}

__global__ __launch_bounds__(1024) void bench(int pat, int iters, long long *out, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 1024) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)pattern(pat, lane);
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
        // eight reads in flight, the tap offsets of a 3x3 window (same pattern shifted by whole pixels / rows)
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:32\n ds_read_b128 %2, %8 offset:64\n ds_read_b128 %3, %8 offset:576\n"
                     "ds_read_b128 %4, %8 offset:608\n ds_read_b128 %5, %8 offset:640\n ds_read_b128 %6, %8 offset:1152\n ds_read_b128 %7, %8 offset:1184\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
        if (i == iters - 1) acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    long long *out;
    unsigned *sink;
    hipMalloc(&out, 16 * sizeof(long long));
    hipMalloc(&sink, 4);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(bench), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int pat = 0; pat < 12; ++pat) {
        for (int nw = 4; nw <= 16; nw *= 2) {
            bench<<<1, 64 * nw, 65536>>>(pat, iters, out, sink);
            hipDeviceSynchronize();
            bench<<<1, 64 * nw, 65536>>>(pat, iters, out, sink);
            hipDeviceSynchronize();
            long long h[16];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double t = 0;
            for (int w = 0; w < nw; ++w) t += (double)h[w];
            // LDS-array cycles per wave-instruction = wall cycles / (reads per wave) / (waves sharing the LDS) ... conflict-free = 4
            printf("pattern %2d, %2d waves: %.2f cycles per ds_read_b128 of the CU (conflict-free ideal 4)\n", pat, nw, t / nw / iters / 8 / nw);
        }
    }
    return 0;
}
