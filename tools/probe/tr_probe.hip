// Probe of ds_read_b64_tr_b16 semantics on gfx950 (run on the GPU box): which element does lane i receive?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ short lds[64 * 64];           // [row 64][col 64], value = row*64+col
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // group g: rows 8*g+q (q=0..3), columns 16*(g&1) + 4p..4p+3
  const int row = 8 * g + q, col = 16 * (g & 1) + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 64 + col));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int g = lane >> 4, i = lane & 15;
    printf("lane %2d:", lane);
    for (int j = 0; j < 4; ++j) {
      int v = h[lane * 4 + j];
      printf(" (r%2d,c%2d)", v / 64, v % 64);
      // hypothesis: lane i of group g receives column 16*(g&1)+i of rows 8g+0..3; element j = row 8g+j
      if (v / 64 != 8 * g + j || v % 64 != 16 * (g & 1) + i) ++bad;
    }
    printf("\n");
  }
  printf("hypothesis mismatches: %d\n", bad);
  return 0;
}
