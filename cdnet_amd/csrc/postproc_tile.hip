// The post-processing chain of a batch of independent tiles (one view each, in its own frame) in THREE launches (round 6; the per-step chain of
// postproc.hip takes ~17):
//   tile_maps_kernel : get_probmaps epilogue (test_dam.py:982-1015: softmax of the mask logits, gated soft-max arg-max of the direction logits)
//                      + generate_dd_map codes (getDirectionDiffMap.py:44-108) from an LDS window of direction classes whose one-pixel halo is
//                      RECOMPUTED from the logits - no stored class plane is read back - + partial (min, max) of the codes and the partial
//                      maximum of the point map (test_dam.py:530).  One workgroup = 16 rows x 256 columns.
//   tile_pred_kernel : boost + arg-max (test_dam.py:529-539) over the whole chip -> pred plane + foreground bit plane (ballots).
//   tile_chain_kernel: ONE 1024-thread workgroup per tile with the whole tile in LDS: fill holes,
//                      remove small objects, 8-connected labelling in raster order, disk dilation (test_dam.py:546-563).  Union-find over
//                      16-bit pixel indices (a tile has at most 65 536 pixels: 128 KB of labels + three bit planes in the CU's 160 KB),
//                      row runs from wave ballots, unions by compare-and-swap on the containing 32-bit word, per-component areas in a
//                      global scratch plane (one atomic per row run).  No grid-wide dependency is left: every phase boundary is a workgroup
//                      barrier.
// Same arithmetic, expression by expression, as probmaps_kernel / ddm_codes_kernel / boost_argmax_kernel / the cc_* kernels of postproc.hip:
// results are bit-identical to that chain (tests/test_gpu_tile_postproc.py) and to the CPU oracle.
#include "common.h"

using namespace cdnet;

namespace {

struct TileLut { int8_t v[17 * 17]; };

constexpr int MAPS_ROWS = 16;       // rows of a tile_maps_kernel workgroup
constexpr int MAPS_COLS = 256;      // columns (64 lanes x 4 pixels)

__device__ __forceinline__ void softmax3(float a0, float a1, float a2, float &p0, float &p1, float &p2) {
    float mx = fmaxf(a0, fmaxf(a1, a2));
    float e0 = expf(a0 - mx), e1 = expf(a1 - mx), e2 = expf(a2 - mx);
    float s = (e0 + e1) + e2;
    p0 = e0 / s; p1 = e1 / s; p2 = e2 / s;
}

// direction class of one pixel (test_dam.py:1011-1013): arg-max of softmax(direction logits) with class 0 scaled by the background probability
template <int C>
__device__ __forceinline__ int dir_class(const float *q_in, float p0) {
    float q[C];
    float dm = q_in[0];
#pragma unroll
    for (int c = 1; c < C; ++c) dm = fmaxf(dm, q_in[c]);
    float ds = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { q[c] = expf(q_in[c] - dm); ds += q[c]; }
    int best = 0;
    float bv = (q[0] / ds) * p0;
#pragma unroll
    for (int c = 1; c < C; ++c) { float v = q[c] / ds; if (v > bv) { bv = v; best = c; } }
    return best;
}

__device__ __forceinline__ int wmin(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wmax(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ float wmaxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// ------------------------------------------------------------------------------------------------------
// launch 1: probabilities / direction classes / DDM codes.  grid (ceil(W/256), ceil(H/16), B), block (64, 4)
// ------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void tile_maps_kernel(const float *__restrict__ ml, const float *__restrict__ dl, const float *__restrict__ point,
                                                        int H, int W, TileLut lut, int nbr, int extra_zero,
                                                        float *__restrict__ prob, uint8_t *__restrict__ dcm, uint8_t *__restrict__ code,
                                                        int32_t *__restrict__ part_mm, float *__restrict__ part_pmax) {
    constexpr int LW = MAPS_COLS + 8;                              // LDS row pitch: [0..3] left halo (col 3 used), 4..259 body, 260 right halo
    __shared__ __attribute__((aligned(16))) uint8_t s_d[(MAPS_ROWS + 2) * LW];
    __shared__ int8_t s_lut[17 * 17];
    __shared__ int s_red[8];
    __shared__ float s_pm[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < C * C; i += 256) s_lut[i] = lut.v[i];
    const int n = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float *m = ml + (size_t)n * 3 * plane;
    const float *d = dl + (size_t)n * C * plane;
    const int y0 = blockIdx.y * MAPS_ROWS, x0 = blockIdx.x * MAPS_COLS;
    const bool vec = (W & 3) == 0;

    // direction classes of the 18 x 258 window (outside the image: class 0, what generate_dd_map's zero padding gives)
    for (int t = tid; t < (MAPS_ROWS + 2) * 64; t += 256) {
        const int r = t >> 6, g = t & 63;
        const int y = y0 + r - 1, x = x0 + g * 4;
        uint8_t o[4] = {0, 0, 0, 0};
        if (y >= 0 && y < H && x < W) {
            const size_t p = (size_t)y * W + x;
            const bool own = r >= 1 && r <= MAPS_ROWS;
            if (vec) {
                const float4 a0 = *reinterpret_cast<const float4 *>(m + p), a1 = *reinterpret_cast<const float4 *>(m + plane + p),
                             a2 = *reinterpret_cast<const float4 *>(m + 2 * plane + p);
                float4 q[C];
#pragma unroll
                for (int c = 0; c < C; ++c) q[c] = *reinterpret_cast<const float4 *>(d + (size_t)c * plane + p);
                float p0[4], p1[4], p2[4];
                softmax3(a0.x, a1.x, a2.x, p0[0], p1[0], p2[0]);
                softmax3(a0.y, a1.y, a2.y, p0[1], p1[1], p2[1]);
                softmax3(a0.z, a1.z, a2.z, p0[2], p1[2], p2[2]);
                softmax3(a0.w, a1.w, a2.w, p0[3], p1[3], p2[3]);
                float qq[C];
#pragma unroll
                for (int c = 0; c < C; ++c) qq[c] = q[c].x;
                o[0] = (uint8_t)dir_class<C>(qq, p0[0]);
#pragma unroll
                for (int c = 0; c < C; ++c) qq[c] = q[c].y;
                o[1] = (uint8_t)dir_class<C>(qq, p0[1]);
#pragma unroll
                for (int c = 0; c < C; ++c) qq[c] = q[c].z;
                o[2] = (uint8_t)dir_class<C>(qq, p0[2]);
#pragma unroll
                for (int c = 0; c < C; ++c) qq[c] = q[c].w;
                o[3] = (uint8_t)dir_class<C>(qq, p0[3]);
                if (own) {
                    if (prob) {
                        float *pr = prob + (size_t)n * 3 * plane + p;
                        *reinterpret_cast<float4 *>(pr) = make_float4(p0[0], p0[1], p0[2], p0[3]);
                        *reinterpret_cast<float4 *>(pr + plane) = make_float4(p1[0], p1[1], p1[2], p1[3]);
                        *reinterpret_cast<float4 *>(pr + 2 * plane) = make_float4(p2[0], p2[1], p2[2], p2[3]);
                    }
                    if (dcm) *reinterpret_cast<uchar4 *>(dcm + (size_t)n * plane + p) = make_uchar4(o[0], o[1], o[2], o[3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (x + j >= W) break;
                    float p0, p1, p2, qq[C];
                    softmax3(m[p + j], m[plane + p + j], m[2 * plane + p + j], p0, p1, p2);
#pragma unroll
                    for (int c = 0; c < C; ++c) qq[c] = d[(size_t)c * plane + p + j];
                    o[j] = (uint8_t)dir_class<C>(qq, p0);
                    if (own) {
                        if (prob) { float *pr = prob + (size_t)n * 3 * plane + p + j; pr[0] = p0; pr[plane] = p1; pr[2 * plane] = p2; }
                        if (dcm) dcm[(size_t)n * plane + p + j] = o[j];
                    }
                }
            }
        }
        *reinterpret_cast<uchar4 *>(s_d + r * LW + 4 + g * 4) = make_uchar4(o[0], o[1], o[2], o[3]);
    }
    // the two halo columns
    for (int t = tid; t < (MAPS_ROWS + 2) * 2; t += 256) {
        const int r = t >> 1, side = t & 1;
        const int y = y0 + r - 1, x = side ? x0 + MAPS_COLS : x0 - 1;
        uint8_t o = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const size_t p = (size_t)y * W + x;
            float p0, p1, p2, qq[C];
            softmax3(m[p], m[plane + p], m[2 * plane + p], p0, p1, p2);
#pragma unroll
            for (int c = 0; c < C; ++c) qq[c] = d[(size_t)c * plane + p];
            o = (uint8_t)dir_class<C>(qq, p0);
        }
        s_d[r * LW + (side ? 4 + MAPS_COLS : 3)] = o;
    }
    __syncthreads();

    // DDM codes (ddm_codes_kernel's arithmetic) + partial min / max, partial point maximum
    int lmin = 0x7fffffff, lmax = -0x7fffffff;
    float pm = -INFINITY;
    const float *pt = point + (size_t)n * plane;
    for (int t = tid; t < MAPS_ROWS * 64; t += 256) {
        const int r = t >> 6, g = t & 63;
        const int y = y0 + r, x = x0 + g * 4;
        if (y >= H || x >= W) continue;
        const uint8_t *row = s_d + (r + 1) * LW + 4 + g * 4;
        uint8_t out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = row[j];
            int v = 0;
            if (c != 0) {
                int mn = extra_zero ? 0 : 2;
                const int8_t *lr = s_lut + c * C;
                if (nbr == 8) {
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (dy == 0 && dx == 0) continue;
                            int q = lr[row[dy * LW + j + dx]];
                            mn = q < mn ? q : mn;
                        }
                } else {
                    int q;
                    q = lr[row[-LW + j]]; mn = q < mn ? q : mn;
                    q = lr[row[LW + j]];  mn = q < mn ? q : mn;
                    q = lr[row[j - 1]];   mn = q < mn ? q : mn;
                    q = lr[row[j + 1]];   mn = q < mn ? q : mn;
                }
                v = 1 - mn;
            }
            out[j] = (uint8_t)v;
            if (x + j < W) { lmin = v < lmin ? v : lmin; lmax = v > lmax ? v : lmax; pm = fmaxf(pm, pt[(size_t)y * W + x + j]); }
        }
        uint8_t *dst = code + (size_t)n * plane + (size_t)y * W + x;
        if (vec) *reinterpret_cast<uchar4 *>(dst) = make_uchar4(out[0], out[1], out[2], out[3]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (x + j < W) dst[j] = out[j];
        }
    }
    lmin = wmin(lmin); lmax = wmax(lmax); pm = wmaxf(pm);
    if (threadIdx.x == 0) { s_red[threadIdx.y] = lmin; s_red[4 + threadIdx.y] = lmax; s_pm[threadIdx.y] = pm; }
    __syncthreads();
    if (tid == 0) {
        int a = s_red[0], b = s_red[4];
        float f = s_pm[0];
        for (int i = 1; i < 4; ++i) { a = s_red[i] < a ? s_red[i] : a; b = s_red[4 + i] > b ? s_red[4 + i] : b; f = fmaxf(f, s_pm[i]); }
        const int nb = gridDim.x * gridDim.y, bi = blockIdx.y * gridDim.x + blockIdx.x;
        part_mm[((size_t)n * nb + bi) * 2] = a;
        part_mm[((size_t)n * nb + bi) * 2 + 1] = b;
        part_pmax[(size_t)n * nb + bi] = f;
    }
}

// ------------------------------------------------------------------------------------------------------
// launch 2: boost + arg-max (boost_argmax_kernel's arithmetic for one view) over the whole chip: pred plane + the foreground bit plane
// (one 64-bit word per 64 consecutive pixels, from the wave's ballot).  grid (P / 1024, B), block 256; P a multiple of 64
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_pred_kernel(const float *__restrict__ ml, const float *__restrict__ point, const uint8_t *__restrict__ code,
                                                        const int32_t *__restrict__ part_mm, const float *__restrict__ part_pmax, int nb,
                                                        int H, int W, int32_t *__restrict__ minmax, uint8_t *__restrict__ pred,
                                                        unsigned long long *__restrict__ fgbits) {
    __shared__ int s_mm[2];
    __shared__ float s_pm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.y;
    const int P = H * W;
    if (tid < 64) {
        int a = 0x7fffffff, b = -0x7fffffff;
        float f = -INFINITY;
        for (int i = tid; i < nb; i += 64) {
            const int32_t *q = part_mm + ((size_t)n * nb + i) * 2;
            a = q[0] < a ? q[0] : a; b = q[1] > b ? q[1] : b; f = fmaxf(f, part_pmax[(size_t)n * nb + i]);
        }
        a = wmin(a); b = wmax(b); f = wmaxf(f);
        if (tid == 0) {
            s_mm[0] = a; s_mm[1] = b; s_pm = f;
            if (blockIdx.x == 0) { minmax[2 * n] = a; minmax[2 * n + 1] = b; }
        }
    }
    __syncthreads();
    const float mn = (float)s_mm[0], den = (float)(s_mm[1] - s_mm[0]);
    const float pmax = s_pm;
    const size_t base = (size_t)n * P;
    const float *m = ml + (size_t)n * 3 * P;
    const float *pt = point + base;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = blockIdx.x * 1024 + k * 256 + tid;
        if (p >= P) break;                                   // (P is a multiple of 64: whole waves leave together)
        const int y = p / W, x = p - y * W;
        bool in3 = pt[p] / pmax > 0.2f;
        if (y > 0) in3 |= pt[p - W] / pmax > 0.2f;
        if (y < H - 1) in3 |= pt[p + W] / pmax > 0.2f;
        if (x > 0) in3 |= pt[p - 1] / pmax > 0.2f;
        if (x < W - 1) in3 |= pt[p + 1] / pmax > 0.2f;
        const float val = ((float)code[base + p] - mn) / den;
        const double ddm = (double)val;                       // (the mean over ONE view)
        const double eb = 2.0 * (ddm - ddm * (double)(in3 ? 1 : 0));
        float p0, p1, p2;
        softmax3(m[p], m[P + p], m[2 * P + p], p0, p1, p2);
        p2 = (float)(((double)p2 + 0.5 * eb) * (1.0 + eb));
        int a = 0;
        float mx = p0;
        if (p1 > mx || (p1 != p1 && mx == mx)) { a = 1; mx = p1; }
        if (p2 > mx || (p2 != p2 && mx == mx)) { a = 2; mx = p2; }
        pred[base + p] = (uint8_t)a;
        const unsigned long long b = __ballot(a == 1);
        if (lane == 0) fgbits[(base + p) >> 6] = b;
    }
}

// ------------------------------------------------------------------------------------------------------
// launch 3: the connected-component chain of a tile inside ONE workgroup, the tile in LDS.
//   * bit planes as 64-bit words, one word = 64 consecutive pixels of a row (W is a multiple of 64); THREAD t OWNS SEGMENT t (at most 1024
//     segments): masks, run heads, the 4- and 8-neighbour tests are bit operations on the thread's own word and its neighbours' words;
//   * union-find over 16-bit pixel indices, defined at RUN HEADS only: the head of any pixel follows from its segment's word (no per-pixel
//     initialisation), a parent is always a head, a root the smallest head of its component (= the raster-first pixel); unions by
//     compare-and-swap on the containing 32-bit word;
//   * only two passes touch every pixel: the expansion of the run labels and the dilation.
// ------------------------------------------------------------------------------------------------------
typedef unsigned short u16;
typedef unsigned long long u64;

__device__ __forceinline__ u16 ldl(const u16 *L, int i) { return *reinterpret_cast<const volatile u16 *>(L + i); }

// root of head `a`, with path halving: a visited head is re-pointed at its grandparent.  The store races with the unions' compare-and-swap on
// purpose: whatever value survives in a non-root slot is a smaller head of the same (final) component - pointers only ever decrease along a
// path, so every walk ends at a root, and a union whose lowered pointer is overwritten has already moved on to linking the two roots
// themselves (tf_min16 told it the slot was no root).  The partition - and with it every output - does not depend on the interleaving.
__device__ __forceinline__ int tf_root(u16 *L, int a) {          // a: a run head
    int p = ldl(L, a);
    while (p != a) {
        const int g = ldl(L, p);
        if (g == p) return p;
        L[a] = (u16)g;
        a = g;
        p = ldl(L, a);
    }
    return a;
}

// read-only walk: the flatten passes store ROOTS into the slots they own, and a path-halving walker passing by could overwrite such a slot with
// a stale grandparent afterwards - while a flatten pass runs, nobody halves
__device__ __forceinline__ int tf_root_ro(const u16 *L, int a) {
    int p = ldl(L, a);
    while (p != a) { a = p; p = ldl(L, a); }
    return a;
}

// min-store of a 16-bit field by compare-and-swap on its 32-bit word; returns the field's value before
__device__ __forceinline__ int tf_min16(u16 *L, int idx, int val) {
    unsigned *w = reinterpret_cast<unsigned *>(L) + (idx >> 1);
    const int sh = (idx & 1) * 16;
    unsigned cur = *reinterpret_cast<volatile unsigned *>(w);
    for (;;) {
        const int f = (cur >> sh) & 0xffff;
        if (f <= val) return f;
        const unsigned nw = (cur & ~(0xffffu << sh)) | ((unsigned)val << sh);
        const unsigned old = atomicCAS(w, cur, nw);
        if (old == cur) return f;
        cur = old;
    }
}

__device__ __forceinline__ int t_run_start(u64 m, int lane) {
    const u64 zeros_below = ~m & ((1ull << lane) - 1ull);
    return zeros_below ? 64 - __clzll(zeros_below) : 0;
}
// the run of set bits of `m` that starts at bit `lane`
__device__ __forceinline__ u64 t_run_mask(u64 m, int lane) {
    const u64 z = ~(m >> lane);
    const int len = z ? __ffsll((long long)z) - 1 : 64 - lane;
    return (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
}
// heads of the survivor runs: a survivor run is a whole run of A
__device__ __forceinline__ u64 kheads_of(u64 k, u64 a) { return k & ~(k << 1) & (a & ~(a << 1)); }
__device__ __forceinline__ int t_head(const u64 *M, int p) { return (p & ~63) + t_run_start(M[p >> 6], p & 63); }

__device__ __forceinline__ void tf_union_heads(u16 *L, int a, int b) {          // a, b: run heads
    bool done;
    do {
        a = tf_root(L, a);
        b = tf_root(L, b);
        if (a < b) { int old = tf_min16(L, b, a); done = (old == b); b = old; }
        else if (b < a) { int old = tf_min16(L, a, b); done = (old == a); a = old; }
        else done = true;
    } while (!done);
}
__device__ __forceinline__ void tf_union(const u64 *M, u16 *L, int p, int q) { tf_union_heads(L, t_head(M, p), t_head(M, q)); }

// segment t of plane M: every run head is its own root
__device__ __forceinline__ void t_init(const u64 *M, u16 *L, int t, int nseg) {
    if (t >= nseg) return;
    const u64 m = M[t];
    u64 heads = m & ~(m << 1);
    while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = (u16)(t * 64 + l); }
}

// 4-connectivity: unions of segment t's runs with the runs of the row above (once per contiguous overlap - cc_merge_kernel's rule
// `N && !(W && NW)`) + the stitch to the segment on the left.  The heads on both sides come from the two words the thread holds anyway.
// (Joining the rows level by level - pairs, pairs of pairs ... - keeps every tree shallow but serialises eight rounds of unions, each a chain
// of ~10 dependent LDS operations: measured 74 k cycles against 45 k for all rows at once with path halving; two rounds - inside 16-row blocks,
// then the blocks - 46 k and the foreground pass 16 k instead of 10 k: the background's 35 k cycles of unions are not a depth problem.)
__device__ __forceinline__ void t_merge4(const u64 *M, u16 *L, int t, int nseg, int spr, int W) {
    if (t >= nseg) return;
    const u64 m = M[t];
    if (!m) return;
    const int xs = t % spr;
    const bool leftbit = xs > 0 && (M[t - 1] >> 63);
    if ((m & 1ull) && leftbit) tf_union(M, L, t * 64, t * 64 - 1);
    if (t < spr) return;
    const u64 u = M[t - spr];
    const u64 ov = m & u;
    const bool prev = leftbit && (M[t - spr - 1] >> 63);
    u64 starts = ov & ~((ov << 1) | (prev ? 1ull : 0ull));
    while (starts) {
        const int l = __ffsll((long long)starts) - 1; starts &= starts - 1;
        tf_union_heads(L, t * 64 + t_run_start(m, l), (t - spr) * 64 + t_run_start(u, l));
    }
}

#ifdef CDNET_TILE_STAMPS
// diagnostic build (tools/tile_stamps.py): s_memtime at every phase boundary of workgroup 0 .. 63
__device__ unsigned long long g_tile_stamps[64 * 32];
#define TSTAMP(k) do { if (tid == 0 && blockIdx.x < 64) g_tile_stamps[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif

// the per-component areas live in a global scratch plane, touched by device-scope atomics only (zero-store, add, load: all performed at the L2,
// no cache maintenance); the workgroup barriers between the three phases wait for the outstanding ones (vmcnt(0))
__device__ __forceinline__ int ld_area(const int *a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_area(int *a, int v) { __hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int R>      // R: the dilation radius as a compile-time constant (0, 1, 2), or -1: any radius up to 8
__global__ __launch_bounds__(1024) void tile_chain_kernel(const u64 *__restrict__ fgbits, int H, int W, int min_area, int radius_rt,
                                                          int *__restrict__ area_ws, uint8_t *__restrict__ fill, uint8_t *__restrict__ small,
                                                          int32_t *__restrict__ label, int32_t *__restrict__ final_, int32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int P = H * W;                                     // <= 65536, W % 64 == 0
    u16 *L = reinterpret_cast<u16 *>(smem);                 // [P]: parents at run heads; in the end every pixel's label
    u64 *BG = reinterpret_cast<u64 *>(smem + 131072);       // background; later: root bits
    u64 *AM = BG + 1024;                                     // filled mask A
    u64 *KM = AM + 1024;                                     // survivors (area >= min_area)
    unsigned *MK = reinterpret_cast<unsigned *>(KM);        // [2048] border-connected background roots (dead before KM is written)
    int *S = reinterpret_cast<int *>(KM + 1024);            // [1024 + 32] scan scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    TSTAMP(0);
    const size_t base = (size_t)n * P;
    const int nseg = P >> 6, spr = W >> 6;
    const int t = tid;                                       // this thread's segment
    int *area = area_ws + base;

    const u64 fg = t < nseg ? fgbits[(base >> 6) + t] : 0ull;
    if (t < nseg) BG[t] = ~fg;
    MK[tid] = 0u; MK[1024 + tid] = 0u;
    __syncthreads(); TSTAMP(1);

    // ---- fill holes (scipy.ndimage.binary_fill_holes, test_dam.py:546): 4-connected background components; those that reach the
    //      border stay background ---------------------------------------------------------------------------------------------------------
    t_init(BG, L, t, nseg);
    __syncthreads(); TSTAMP(2);
    t_merge4(BG, L, t, nseg, spr, W);
    __syncthreads(); TSTAMP(17);
    if (t < nseg) {                                          // (one more halving walk from every head: the paths the flatten reads get short)
        const u64 m = ~fg;
        u64 heads = m & ~(m << 1);
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; tf_root(L, t * 64 + l); }
    }
    __syncthreads(); TSTAMP(18);
    if (t < nseg) {
        const u64 m = ~fg;
        u64 heads = m & ~(m << 1);
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = (u16)tf_root_ro(L, t * 64 + l); }
    }
    __syncthreads(); TSTAMP(3);
    for (int i = tid; i < 2 * (W + H); i += 1024) {
        int y, x;
        if (i < W) { y = 0; x = i; }
        else if (i < 2 * W) { y = H - 1; x = i - W; }
        else if (i < 2 * W + H) { y = i - 2 * W; x = 0; }
        else { y = i - 2 * W - H; x = W - 1; }
        const int p = y * W + x;
        if ((BG[p >> 6] >> (p & 63)) & 1ull) { const int r = ldl(L, t_head(BG, p)); atomicOr(MK + (r >> 5), 1u << (r & 31)); }
    }
    __syncthreads(); TSTAMP(4);
    u64 am = fg;
    if (t < nseg) {
        const u64 m = ~fg;
        u64 heads = m & ~(m << 1);
        while (heads) {
            const int l = __ffsll((long long)heads) - 1; heads &= heads - 1;
            const int r = ldl(L, t * 64 + l);                // (flattened above)
            if (!((MK[r >> 5] >> (r & 31)) & 1u)) am |= t_run_mask(m, l);
        }
        AM[t] = am;
    }
    __syncthreads(); TSTAMP(5);

    // ---- 4-connected components of A with their areas (skimage remove_small_objects, test_dam.py:549) ------------------------------------
    if (t < nseg) {
        u64 heads = am & ~(am << 1);
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = (u16)(t * 64 + l); st_area(area + t * 64 + l, 0); }
    }
    __syncthreads(); TSTAMP(6);
    t_merge4(AM, L, t, nseg, spr, W);
    __syncthreads(); TSTAMP(7);
    if (t < nseg) {
        u64 heads = am & ~(am << 1);
        while (heads) {
            const int l = __ffsll((long long)heads) - 1; heads &= heads - 1;
            const int r = tf_root_ro(L, t * 64 + l);
            L[t * 64 + l] = (u16)r;
            atomicAdd(area + r, __popcll(t_run_mask(am, l)));
        }
    }
    __syncthreads(); TSTAMP(8);
    u64 km = 0ull;
    if (t < nseg) {
        u64 heads = am & ~(am << 1);
        while (heads) {
            const int l = __ffsll((long long)heads) - 1; heads &= heads - 1;
            if (ld_area(area + ldl(L, t * 64 + l)) >= min_area) km |= t_run_mask(am, l);
        }
        KM[t] = km;
    }
    __syncthreads(); TSTAMP(9);
    // ---- the diagonal unions among the survivors turn the 4-connected forest into the 8-connected one (skimage.measure.label):
    //      cc_diag_merge_kernel's rule - p kept, N not in A; NW kept and W not in A; NE kept and E not in A -----------------------------------
    if (t >= spr && t < nseg && km) {
        const int xs = t % spr;
        const u64 aup = AM[t - spr], kup = KM[t - spr];
        const u64 awest = (am << 1) | ((xs > 0 && (AM[t - 1] >> 63)) ? 1ull : 0ull);
        const u64 aeast = (am >> 1) | ((xs < spr - 1 && (AM[t + 1] & 1ull)) ? (1ull << 63) : 0ull);
        const u64 knw = (kup << 1) | ((xs > 0 && (KM[t - spr - 1] >> 63)) ? 1ull : 0ull);
        const u64 kne = (kup >> 1) | ((xs < spr - 1 && (KM[t - spr + 1] & 1ull)) ? (1ull << 63) : 0ull);
        u64 c = km & ~aup & ~awest & knw;
        while (c) { const int l = __ffsll((long long)c) - 1; c &= c - 1; tf_union(AM, L, t * 64 + l, t * 64 + l - W - 1); }
        c = km & ~aup & ~aeast & kne;
        while (c) { const int l = __ffsll((long long)c) - 1; c &= c - 1; tf_union(AM, L, t * 64 + l, t * 64 + l - W + 1); }
    }
    __syncthreads(); TSTAMP(10);
    const u64 kheads = km & ~(km << 1) & (am & ~(am << 1));         // (a survivor run is a whole run of A: its head is A's head)
    if (t < nseg) {
        u64 heads = kheads;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = (u16)tf_root_ro(L, t * 64 + l); }
    }
    __syncthreads(); TSTAMP(11);
    // roots in raster order = segment order, then bit order
    u64 roots = 0ull;
    {
        u64 heads = kheads;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; if (ldl(L, t * 64 + l) == t * 64 + l) roots |= 1ull << l; }
        const int c = __popcll(roots);
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) S[1024 + wave] = inc;
        __syncthreads(); TSTAMP(12);
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (i < wave) woff += S[1024 + i]; tot += S[1024 + i]; }
        if (tid == 0 && counts) counts[n] = tot;
        int rk = woff + inc - c;
        u64 r = roots;
        while (r) { const int l = __ffsll((long long)r) - 1; r &= r - 1; L[t * 64 + l] = (u16)(++rk); }      // (every reader of L[root] == root is done)
    }
    __syncthreads(); TSTAMP(13);
    if (t < nseg) {
        u64 heads = kheads & ~roots;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = ldl(L, ldl(L, t * 64 + l)); }
    }
    __syncthreads(); TSTAMP(14);
    // ---- run labels -> every pixel (the only other pass over the pixels is the dilation) ------------------------------------------------
    // (the CU's LDS pipe takes ~8 cycles per wave-level operation whatever its width: four consecutive pixels per lane - two word reads, four
    //  16-bit label reads and ONE 8-byte write per 256 pixels of a wave instead of four operations per 64.  A kept head reads its own slot and
    //  writes the same value back: the slots other lanes read are never changed by this pass.)
    for (int p = tid * 4; p < P; p += 4096) {
        const int sg = p >> 6, lb = p & 63;
        const u64 k = KM[sg], a = AM[sg];
        const u64 hw = a & ~(a << 1);
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = lb + j;
            const u64 x = hw & (~0ull >> (63 - l));
            const int hpos = 63 - __clzll((long long)(x | 1ull));            // (x == 0 only for pixels outside A: never kept)
            v[j] = L[sg * 64 + hpos];
        }
        u64 out = 0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((k >> (lb + j)) & 1ull) out |= (u64)(unsigned)v[j] << (16 * j);
        *reinterpret_cast<u64 *>(L + p) = out;
        if (fill) *reinterpret_cast<uchar4 *>(fill + base + p) = make_uchar4((a >> lb) & 1, (a >> (lb + 1)) & 1, (a >> (lb + 2)) & 1, (a >> (lb + 3)) & 1);
        if (small) *reinterpret_cast<uchar4 *>(small + base + p) = make_uchar4((k >> lb) & 1, (k >> (lb + 1)) & 1, (k >> (lb + 2)) & 1, (k >> (lb + 3)) & 1);
    }
    __syncthreads(); TSTAMP(15);
    // ---- labels out, dilation by disk(radius) (skimage.morphology.dilation, test_dam.py:563) ----------------------------------------------
    const int wsh = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
    if (R == 1 || R == 2) {
        // four pixels per lane: per row of the disk the twelve labels [x - 4, x + 8) as three 8-byte reads (label 0 outside the image: neutral)
        for (int p = tid * 4; p < P; p += 4096) {
            const int y = wsh >= 0 ? p >> wsh : p / W, x = p - y * W;
            int v[4];
            {
                const u64 c = *reinterpret_cast<const u64 *>(L + p);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (int)((c >> (16 * j)) & 0xffff);
                if (label) *reinterpret_cast<int4 *>(label + base + p) = make_int4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int dy = -R; dy <= R; ++dy) {
                const bool rowok = (unsigned)(y + dy) < (unsigned)H;
                const u16 *row = L + p + dy * W;
                const int ext = R * R - dy * dy >= 4 ? 2 : (R * R - dy * dy >= 1 ? 1 : 0);      // |dx| <= ext on this row
                u64 c0 = 0ull, c1 = 0ull, c2 = 0ull;
                if (rowok) {
                    c1 = *reinterpret_cast<const u64 *>(row);
                    if (ext > 0) {
                        if (x >= 4) c0 = *reinterpret_cast<const u64 *>(row - 4);
                        if (x + 4 < W) c2 = *reinterpret_cast<const u64 *>(row + 4);
                    }
                }
                int w[12];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w[j] = (int)((c0 >> (16 * j)) & 0xffff); w[4 + j] = (int)((c1 >> (16 * j)) & 0xffff); w[8 + j] = (int)((c2 >> (16 * j)) & 0xffff);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dx = -ext; dx <= ext; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const int q = w[4 + j + dx];
                        v[j] = q > v[j] ? q : v[j];
                    }
            }
            *reinterpret_cast<int4 *>(final_ + base + p) = make_int4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (int p = tid; p < P; p += 1024) {
            const int y = wsh >= 0 ? p >> wsh : p / W, x = p - y * W;
            int v = L[p];
            if (label) label[base + p] = v;
            const int radius = R == 0 ? 0 : radius_rt;
            for (int dy = -radius; dy <= radius; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -radius; dx <= radius; ++dx) {
                    if (dy * dy + dx * dx > radius * radius) continue;
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const int q = L[yy * W + xx];
                    v = q > v ? q : v;
                }
            }
            final_[base + p] = v;
        }
    }
    TSTAMP(16);
}

// ------------------------------------------------------------------------------------------------------
// 8-connected labelling of a binary tile in raster order (skimage.measure.label) in ONE launch: tile_chain_kernel's machinery without the
// hole filling and the area test - what the target generation (cdm.hip: label_instance = measure.label(new_label == 1),
// my_transforms_direction.py:785-790) needs per label image; replaces the seven launches of label8_raster for tiles.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tile_label8_kernel(const uint8_t *__restrict__ mask, int H, int W, int32_t *__restrict__ labels,
                                                           int32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int P = H * W;
    u16 *L = reinterpret_cast<u16 *>(smem);
    u64 *AM = reinterpret_cast<u64 *>(smem + 131072);
    int *S = reinterpret_cast<int *>(AM + 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    const size_t base = (size_t)n * P;
    const int nseg = P >> 6, spr = W >> 6;
    const int t = tid;
    // the mask's bit plane: sixteen pixels per lane and load (four loads per lane for a 256 x 256 tile instead of 64 dependent byte loads), four
    // neighbouring lanes OR their 16 bits into a segment's word
    for (int q = tid; q < (P >> 4); q += 1024) {
        const uint4 v = reinterpret_cast<const uint4 *>(mask + base)[q];
        const unsigned xs4[4] = {v.x, v.y, v.z, v.w};
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned x = xs4[i];
            const unsigned nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;       // bit 7 of every non-zero byte
            bits |= (((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u)) << (4 * i);
        }
        u64 w = (u64)bits << (16 * (q & 3));
        w |= __shfl_xor(w, 1);
        w |= __shfl_xor(w, 2);
        if ((q & 3) == 0) AM[q >> 2] = w;
    }
    __syncthreads();
    const u64 am = t < nseg ? AM[t] : 0ull;
    t_init(AM, L, t, nseg);
    __syncthreads();
    t_merge4(AM, L, t, nseg, spr, W);
    __syncthreads();
    if (t >= spr && t < nseg && am) {                        // the diagonal contacts (cc_merge_kernel<1, 8>: NW / NE only where N, W / E do not connect already)
        const int xs = t % spr;
        const u64 up = AM[t - spr];
        const u64 awest = (am << 1) | ((xs > 0 && (AM[t - 1] >> 63)) ? 1ull : 0ull);
        const u64 aeast = (am >> 1) | ((xs < spr - 1 && (AM[t + 1] & 1ull)) ? (1ull << 63) : 0ull);
        const u64 unw = (up << 1) | ((xs > 0 && (AM[t - spr - 1] >> 63)) ? 1ull : 0ull);
        const u64 une = (up >> 1) | ((xs < spr - 1 && (AM[t - spr + 1] & 1ull)) ? (1ull << 63) : 0ull);
        u64 c = am & ~up & ~awest & unw;
        while (c) { const int l = __ffsll((long long)c) - 1; c &= c - 1; tf_union(AM, L, t * 64 + l, t * 64 + l - W - 1); }
        c = am & ~up & ~aeast & une;
        while (c) { const int l = __ffsll((long long)c) - 1; c &= c - 1; tf_union(AM, L, t * 64 + l, t * 64 + l - W + 1); }
    }
    __syncthreads();
    const u64 hds = am & ~(am << 1);
    if (t < nseg) {
        u64 heads = hds;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = (u16)tf_root_ro(L, t * 64 + l); }
    }
    __syncthreads();
    u64 roots = 0ull;
    {
        u64 heads = hds;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; if (ldl(L, t * 64 + l) == t * 64 + l) roots |= 1ull << l; }
        const int c = __popcll(roots);
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) S[wave] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (i < wave) woff += S[i]; tot += S[i]; }
        if (tid == 0 && counts) counts[n] = tot;
        int rk = woff + inc - c;
        u64 r = roots;
        while (r) { const int l = __ffsll((long long)r) - 1; r &= r - 1; L[t * 64 + l] = (u16)(++rk); }
    }
    __syncthreads();
    if (t < nseg) {
        u64 heads = hds & ~roots;
        while (heads) { const int l = __ffsll((long long)heads) - 1; heads &= heads - 1; L[t * 64 + l] = ldl(L, ldl(L, t * 64 + l)); }
    }
    __syncthreads();
    for (int p = tid * 4; p < P; p += 4096) {
        const int sg = p >> 6, lb = p & 63;
        const u64 a = AM[sg];
        const u64 hw = a & ~(a << 1);
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = lb + j;
            const u64 x = hw & (~0ull >> (63 - l));
            const int hpos = 63 - __clzll((long long)(x | 1ull));
            v[j] = ((a >> l) & 1ull) ? (int)L[sg * 64 + hpos] : 0;
        }
        *reinterpret_cast<int4 *>(labels + base + p) = make_int4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

#ifdef CDNET_TILE_STAMPS
extern "C" int cdnet_debug_tile_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 64 * 32) == hipSuccess ? 0 : 3;
}
#endif

namespace cdnet {
// internal (cdm.hip through label8_raster): true when the shape is the tile kernel's (W a multiple of 64, at most 65 536 pixels) and the launch was queued
bool label8_tile(const uint8_t *mask, int N, int H, int W, int32_t *labels, int32_t *counts, hipStream_t st, int *rc) {
    if (N <= 0 || W % 64 != 0 || (long long)H * W > 65536 || (((size_t)labels) & 15) != 0 || (((size_t)mask) & 15) != 0) return false;
    constexpr int SMEM = 131072 + 8192 + 64 * 4;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(tile_label8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            *rc = check_launch("hipFuncSetAttribute(tile_label8)");
            return true;
        }
        attr = true;
    }
    tile_label8_kernel<<<N, 1024, SMEM, st>>>(mask, H, W, labels, counts);
    *rc = check_launch("tile_label8_kernel");
    return true;
}
}  // namespace cdnet

// workspace: code u8 [B*P] | area i32 [B*P] | fg bit plane u64 [B*P/64] | part_mm i32 [B*nb*2] | part_pmax f32 [B*nb]
static size_t tile_ws_layout(int B, int H, int W, size_t *oCode, size_t *oArea, size_t *oBits, size_t *oMM, size_t *oPM, int *nb_out) {
    const size_t P = (size_t)H * W;
    const int nb = cdiv(W, MAPS_COLS) * cdiv(H, MAPS_ROWS);
    size_t off = 0;
    *oCode = off; off = align_up(off + (size_t)B * P, 256);
    *oArea = off; off = align_up(off + (size_t)B * P * 4, 256);
    *oBits = off; off = align_up(off + (size_t)B * P / 8, 256);
    *oMM = off; off = align_up(off + (size_t)B * nb * 8, 256);
    *oPM = off; off = align_up(off + (size_t)B * nb * 4, 256);
    *nb_out = nb;
    return off;
}

static bool tile_shape_ok(int B, int C, int H, int W) {
    return B > 0 && H > 0 && W > 0 && (C == 5 || C == 9 || C == 17) && W % 64 == 0 && (long long)H * W <= 65536;
}

extern "C" size_t cdnet_tile_postproc_workspace_bytes(int B, int C, int H, int W) {
    if (!tile_shape_ok(B, C, H, W)) return 0;
    size_t a, b, c, d, e;
    int nb;
    return tile_ws_layout(B, H, W, &a, &b, &c, &d, &e, &nb);
}

extern "C" int cdnet_tile_postproc(const float *mask_logits, const float *dir_logits, const float *point, int B, int C, int H, int W,
                                   const int8_t *lut_host, int nbr, int extra_zero, int min_area, int radius, void *workspace,
                                   size_t workspace_bytes, float *prob, uint8_t *dcm, int32_t *minmax, uint8_t *pred, uint8_t *fill,
                                   uint8_t *small, int32_t *label, int32_t *final_, int32_t *counts, void *stream) {
    CDNET_REQUIRE(mask_logits && dir_logits && point && lut_host && workspace && minmax && pred && final_, "cdnet_tile_postproc: null pointer");
    CDNET_REQUIRE(tile_shape_ok(B, C, H, W), "cdnet_tile_postproc: B=%d C=%d H=%d W=%d: the fused tile chain takes 5 / 9 / 17 direction classes, W a "
                  "multiple of 64 and at most 65536 pixels per tile (cdnet_tile_postproc_workspace_bytes returns 0 for other shapes: take the per-step chain)",
                  B, C, H, W);
    CDNET_REQUIRE(nbr == 4 || nbr == 8, "cdnet_tile_postproc: nbr must be 4 or 8");
    CDNET_REQUIRE(radius >= 0 && radius <= 8, "cdnet_tile_postproc: radius %d not in [0,8]", radius);
    size_t oCode, oArea, oBits, oMM, oPM;
    int nb;
    const size_t need = tile_ws_layout(B, H, W, &oCode, &oArea, &oBits, &oMM, &oPM, &nb);
    if (workspace_bytes < need) {
        set_error("cdnet_tile_postproc: workspace %zu < %zu bytes", workspace_bytes, need);
        return CDNET_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint8_t *code = (uint8_t *)(ws + oCode);
    int *area = (int *)(ws + oArea);
    u64 *bits = (u64 *)(ws + oBits);
    int32_t *pmm = (int32_t *)(ws + oMM);
    float *ppm = (float *)(ws + oPM);
    TileLut lut;
    for (int i = 0; i < 17 * 17; ++i) lut.v[i] = i < C * C ? lut_host[i] : 0;
    const dim3 g1(cdiv(W, MAPS_COLS), cdiv(H, MAPS_ROWS), B), b1(64, 4);
    if (C == 9) tile_maps_kernel<9><<<g1, b1, 0, st>>>(mask_logits, dir_logits, point, H, W, lut, nbr, extra_zero, prob, dcm, code, pmm, ppm);
    else if (C == 5) tile_maps_kernel<5><<<g1, b1, 0, st>>>(mask_logits, dir_logits, point, H, W, lut, nbr, extra_zero, prob, dcm, code, pmm, ppm);
    else tile_maps_kernel<17><<<g1, b1, 0, st>>>(mask_logits, dir_logits, point, H, W, lut, nbr, extra_zero, prob, dcm, code, pmm, ppm);
    tile_pred_kernel<<<dim3(cdiv(H * W, 1024), B), 256, 0, st>>>(mask_logits, point, code, pmm, ppm, nb, H, W, minmax, pred, bits);
    constexpr int SMEM = 131072 + 3 * 8192 + (1024 + 64) * 4;
    static bool attr = false;
    if (!attr) {
        const void *ks[4] = {reinterpret_cast<const void *>(tile_chain_kernel<0>), reinterpret_cast<const void *>(tile_chain_kernel<1>),
                             reinterpret_cast<const void *>(tile_chain_kernel<2>), reinterpret_cast<const void *>(tile_chain_kernel<-1>)};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
                return check_launch("hipFuncSetAttribute(tile_chain)");
        attr = true;
    }
#define CDNET_TILE_CHAIN(R_) tile_chain_kernel<R_><<<B, 1024, SMEM, st>>>(bits, H, W, min_area, radius, area, fill, small, label, final_, counts)
    if (radius == 2) CDNET_TILE_CHAIN(2);
    else if (radius == 1) CDNET_TILE_CHAIN(1);
    else if (radius == 0) CDNET_TILE_CHAIN(0);
    else CDNET_TILE_CHAIN(-1);
#undef CDNET_TILE_CHAIN
    return check_launch("cdnet_tile_postproc");
}
