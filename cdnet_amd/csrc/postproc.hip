// Integer / byte post-processing kernels of the CDNet inference path for gfx950 (wave64).
//   DDM codes + normalise      <- data_prepare/getDirectionDiffMap.py:44-108
//   get_probmaps epilogue      <- test_dam.py:982-1015
//   TTA mean / boost / argmax  <- test_dam.py:445-450, 479-491, 529-539
//   CC chain                   <- test_dam.py:546-563
// All of this is HBM/latency-bound byte and index work: coalesced row-segment accesses, wave-level ballots for
// the row runs, union-find with agent-scope atomics for the label propagation.  Results are bit-exact against
// the CPU oracle (tests/test_gpu_postproc.py).
#include "common.h"

using namespace cdnet;

namespace {

// ------------------------------------------------------------------------------------------------------
// view transforms (test_dam.py:313-441): bit0 hflip, bit1 vflip (both in the view frame), bit2 rot90 ccw first.
// image pixel (y,x) of an HxW image -> offset inside the stored view plane.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int view_offset(int xf, int y, int x, int H, int W) {
    int hv, wv, yv, xv;
    if (xf & 4) { hv = W; wv = H; yv = W - 1 - x; xv = y; }
    else        { hv = H; wv = W; yv = y;         xv = x; }
    if (xf & 1) xv = wv - 1 - xv;
    if (xf & 2) yv = hv - 1 - yv;
    return yv * wv + xv;
}

struct ViewXf { int v[16]; };

__device__ __forceinline__ int ld_relaxed(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// ------------------------------------------------------------------------------------------------------
// DDM
// ------------------------------------------------------------------------------------------------------
// A direction-difference code is one of 0 .. 3 (1 - min of rounded cosines in {-1 .. 2}): the (min, max) of a map are gathered as four
// PRESENCE BYTES in the map's first minmax word - plain idempotent byte stores, no read-modify-write.  (Round 5 used atomicMin / atomicMax per
// workgroup: 16 000 atomics on 16 addresses per 8-view image serialise at the L2 - 186 us of a kernel that moves 16 MB; "only when it still
// moves the value" left the first wave of 2 048 resident workgroups, 69 us.)
__global__ void init_minmax_kernel(int32_t *minmax, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { minmax[2 * i] = 0; minmax[2 * i + 1] = 0; }
}

__global__ void finish_minmax_kernel(int32_t *minmax, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned f = (unsigned)minmax[2 * i];
    int mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if ((f >> (8 * v)) & 0xffu) { mn = v < mn ? v : mn; mx = v; }
    minmax[2 * i] = mn; minmax[2 * i + 1] = mx;
}

// block (64,4): 256 px x 4 rows, 4 consecutive pixels per thread.
// Round 6: the cosine table as PACKED ROWS - two bits per entry (round(cos) + 1 in {0, 1, 2}), one 64-bit word per centre class, ONE LDS read per
// pixel instead of eight byte reads with bank conflicts - and, for row pitches that are multiples of four, the 3 x 6 neighbourhood from three
// aligned 32-bit loads plus the neighbouring lanes' words (8 views of a 1000 x 1000 image: 97 -> ~15 us per four views).
struct DdmRows { unsigned long long r[17]; };

__global__ __launch_bounds__(256) void ddm_codes_kernel(const uint8_t *__restrict__ dcm, int H, int W, int classes,
                                                        DdmRows rows, int nbr, int extra_zero,
                                                        uint8_t *__restrict__ code, int32_t *minmax) {
    __shared__ unsigned long long s_rows[17];
    __shared__ int s_red[8];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid < 17) s_rows[tid] = rows.r[tid];
    __syncthreads();

    const int n = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const uint8_t *src = dcm + n * plane;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    unsigned pmask = 0;                                       // codes among this thread's pixels
    const bool vec = (W & 3) == 0 && (((size_t)src) & 3) == 0;
    uint8_t t[3][6];
    if (vec) {
        // three aligned words (zero outside the image); the byte left / right of them comes from the neighbouring lanes' words
        unsigned wv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            wv[r] = (yy >= 0 && yy < H && x0 < W) ? *reinterpret_cast<const unsigned *>(src + (size_t)yy * W + x0) : 0u;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            unsigned lw = __shfl_up(wv[r], 1), rw = __shfl_down(wv[r], 1);
            if (threadIdx.x == 0) lw = (yy >= 0 && yy < H && x0 >= 4 && x0 - 4 < W) ? *reinterpret_cast<const unsigned *>(src + (size_t)yy * W + x0 - 4) : 0u;
            if (threadIdx.x == 63) rw = (yy >= 0 && yy < H && x0 + 4 < W) ? *reinterpret_cast<const unsigned *>(src + (size_t)yy * W + x0 + 4) : 0u;
            t[r][0] = (uint8_t)(lw >> 24);
            t[r][1] = (uint8_t)wv[r]; t[r][2] = (uint8_t)(wv[r] >> 8); t[r][3] = (uint8_t)(wv[r] >> 16); t[r][4] = (uint8_t)(wv[r] >> 24);
            t[r][5] = (uint8_t)rw;
        }
    }
    if (y < H && x0 < W) {
        if (!vec) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int yy = y + r - 1;
                bool rowok = yy >= 0 && yy < H;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    int xx = x0 + c - 1;
                    t[r][c] = (rowok && xx >= 0 && xx < W) ? src[(size_t)yy * W + xx] : (uint8_t)0;
                }
            }
        }
        uint8_t out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = t[1][j + 1];
            int v = 0;
            if (c != 0) {
                int m = extra_zero ? 0 : 2;
                const unsigned long long row = s_rows[c];
                auto q_of = [&](int nb) { return (int)((row >> (2 * nb)) & 3ull) - 1; };
                if (nbr == 8) {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            if (r == 1 && d == 1) continue;
                            int q = q_of(t[r][j + d]);
                            m = q < m ? q : m;
                        }
                } else {
                    int q;
                    q = q_of(t[0][j + 1]); m = q < m ? q : m;
                    q = q_of(t[2][j + 1]); m = q < m ? q : m;
                    q = q_of(t[1][j]);     m = q < m ? q : m;
                    q = q_of(t[1][j + 2]); m = q < m ? q : m;
                }
                v = 1 - m;
            }
            out[j] = (uint8_t)v;
            if (x0 + j < W) pmask |= 1u << v;
        }
        uint8_t *dst = code + n * plane + (size_t)y * W + x0;
        if ((W & 3) == 0) {
            *reinterpret_cast<uchar4 *>(dst) = make_uchar4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < W) dst[j] = out[j];
        }
    }
    // which of the four codes occur in this workgroup's pixels -> their presence bytes (skipped when already set)
    unsigned present = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (__ballot((pmask >> v) & 1u)) present |= 1u << v;
    if (threadIdx.x == 0) s_red[threadIdx.y] = (int)present;
    __syncthreads();
    if (tid == 0) {
        present = (unsigned)(s_red[0] | s_red[1] | s_red[2] | s_red[3]);
        uint8_t *flags = reinterpret_cast<uint8_t *>(&minmax[2 * n]);
        const unsigned have = (unsigned)ld_relaxed(&minmax[2 * n]);
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (((present >> v) & 1u) && !((have >> (8 * v)) & 0xffu)) flags[v] = 1;
    }
}

__global__ __launch_bounds__(256) void ddm_normalize_kernel(const uint8_t *__restrict__ code,
                                                            const int32_t *__restrict__ minmax, int plane,
                                                            float *__restrict__ out) {
    const int n = blockIdx.y;
    const float mn = (float)minmax[2 * n], den = (float)(minmax[2 * n + 1] - minmax[2 * n]);
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x)
        out[base + i] = ((float)code[base + i] - mn) / den;
}

// ------------------------------------------------------------------------------------------------------
// get_probmaps epilogue
// ------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void probmaps_kernel(const float *__restrict__ ml, const float *__restrict__ dl,
                                                       int plane, float *__restrict__ prob,
                                                       uint8_t *__restrict__ dcm) {
    const int n = blockIdx.y;
    const float *m = ml + (size_t)n * 3 * plane;
    const float *d = dl + (size_t)n * C * plane;
    float *p = prob + (size_t)n * 3 * plane;
    uint8_t *o = dcm + (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        float a0 = m[i], a1 = m[plane + i], a2 = m[2 * plane + i];
        float mx = fmaxf(a0, fmaxf(a1, a2));
        float e0 = expf(a0 - mx), e1 = expf(a1 - mx), e2 = expf(a2 - mx);
        float s = (e0 + e1) + e2;
        float p0 = e0 / s;
        p[i] = p0; p[plane + i] = e1 / s; p[2 * plane + i] = e2 / s;
        float q[C];
        float dm = d[i];
        q[0] = dm;
#pragma unroll
        for (int c = 1; c < C; ++c) { q[c] = d[(size_t)c * plane + i]; dm = fmaxf(dm, q[c]); }
        float ds = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { q[c] = expf(q[c] - dm); ds += q[c]; }
        int best = 0;
        float bv = (q[0] / ds) * p0;
#pragma unroll
        for (int c = 1; c < C; ++c) { float v = q[c] / ds; if (v > bv) { bv = v; best = c; } }
        o[i] = (uint8_t)best;
    }
}

// ------------------------------------------------------------------------------------------------------
// TTA mean + boost + argmax
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// one view in its own frame (the tile pipeline): the "mean" over views IS the view - nothing to average or to store; what is left of
// tta_mean_kernel is the maximum of the point map (test_dam.py:530), four pixels per thread
__global__ __launch_bounds__(256) void point_max_kernel(const float *__restrict__ points, int plane, unsigned *pmax_key) {
    __shared__ float s_red[4];
    const int img = blockIdx.y;
    const float *pt = points + (size_t)img * plane;
    float m = -INFINITY;
    const int n4 = plane >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(pt)[i];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (plane & 3)) m = fmaxf(m, pt[(n4 << 2) + threadIdx.x]);
    m = wave_maxf(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        if (m != -INFINITY && f2key(m) > (unsigned)ld_relaxed(reinterpret_cast<const int *>(&pmax_key[img]))) atomicMax(&pmax_key[img], f2key(m));
    }
}

__global__ __launch_bounds__(256) void boost_argmax_kernel(const float *__restrict__ probs, const float *__restrict__ prob_mean,
                                                           const float *__restrict__ point_mean,
                                                           const uint8_t *__restrict__ codes, const int32_t *__restrict__ minmax,
                                                           int V, ViewXf xf, int H, int W, const unsigned *__restrict__ pmax_key,
                                                           uint8_t *__restrict__ ddm16, uint8_t *__restrict__ pred) {
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int plane = H * W, p = y * W + x;
    const float pmax = key2f(pmax_key[img]);
    const float *pt = point_mean + (size_t)img * plane;
    // inside3 = dilation((point/max > 0.2), disk(1))   (test_dam.py:530-531)
    bool in3 = pt[p] / pmax > 0.2f;
    if (y > 0) in3 |= pt[p - W] / pmax > 0.2f;
    if (y < H - 1) in3 |= pt[p + W] / pmax > 0.2f;
    if (x > 0) in3 |= pt[p - 1] / pmax > 0.2f;
    if (x < W - 1) in3 |= pt[p + 1] / pmax > 0.2f;
    // mean over views of the per-view normalised DDM (float32 values, float64 mean)   (:479-491)
    double s = 0.0;
    const uint8_t *cd = codes + (size_t)img * V * plane;
    const int32_t *mm = minmax + (size_t)img * V * 2;
    for (int v = 0; v < V; ++v) {
        int o = view_offset(xf.v[v], y, x, H, W);
        float mn = (float)mm[2 * v], den = (float)(mm[2 * v + 1] - mm[2 * v]);
        float val = ((float)cd[(size_t)v * plane + o] - mn) / den;
        s += (double)val;
    }
    const double ddm = s / (double)V;
    if (ddm16) {
        double t = ddm * 16.0;
        ddm16[(size_t)img * plane + p] = (t >= 0.0 && t <= 254.0 && t == (double)(int)t) ? (uint8_t)(int)t : (uint8_t)255;
    }
    // boost (:532-536) in float64, stored back to the float32 probability map, then argmax (:537)
    const double eb = 2.0 * (ddm - ddm * (double)(in3 ? 1 : 0));
    float p0, p1, p2;
    if (prob_mean) {
        const float *pm = prob_mean + (size_t)img * 3 * plane;
        p0 = pm[p]; p1 = pm[plane + p]; p2 = pm[2 * plane + p];
    } else {
        const float *pr = probs + (size_t)img * V * 3 * plane;
        int o = view_offset(xf.v[0], y, x, H, W);
        p0 = pr[o]; p1 = pr[plane + o]; p2 = pr[2 * plane + o];
        for (int v = 1; v < V; ++v) {
            o = view_offset(xf.v[v], y, x, H, W);
            const float *q = pr + (size_t)v * 3 * plane;
            p0 = p0 + q[o]; p1 = p1 + q[plane + o]; p2 = p2 + q[2 * plane + o];
        }
        const float fv = (float)V;
        p0 = p0 / fv; p1 = p1 / fv; p2 = p2 / fv;
    }
    p2 = (float)(((double)p2 + 0.5 * eb) * (1.0 + eb));
    int a = 0;
    float m = p0;
    if (p1 > m || (p1 != p1 && m == m)) { a = 1; m = p1; }
    if (p2 > m || (p2 != p2 && m == m)) { a = 2; m = p2; }
    pred[(size_t)img * plane + p] = (uint8_t)a;
}

// ------------------------------------------------------------------------------------------------------
// The same two steps for V views in their own frames (round 6): a 32 x 32-pixel image tile per workgroup, four pixels per thread.  A view
// that is rotated against the image (bit 2) lies transposed in memory - a lane per image column would touch a cache line per lane (the r05
// kernels: 79 + 63 us per 1000 x 1000 image, 1.6 TB/s of mostly wasted lines) - so its tile is read along ITS rows (lanes along the image's
// y) into an LDS tile and taken out transposed (pitch 33: conflict-free).  Arithmetic and summation order are tta_mean_kernel's /
// boost_argmax_kernel's: bit-identical results.
//   views_point_kernel: mean of the point maps + its maximum (test_dam.py:445-450, 530)
//   views_boost_kernel: mean of the probabilities, mean of the normalised direction-difference maps, boost, arg-max (test_dam.py:479-539)
// ------------------------------------------------------------------------------------------------------
constexpr int VT = 32;                 // tile edge

// value of pixel (y0 + ty + 8k, x0 + tx), k = 0..3, of one plane of view `xf`; `tile`: [VT][VT + 1] floats of LDS (rotated views only).
// Every thread of the workgroup calls it (barriers inside for a rotated view).
template <typename T>
__device__ __forceinline__ void view_tile_load(const T *__restrict__ plane, int xf, int H, int W, int y0, int x0, int tx, int ty, float *tile,
                                               float out[4]) {
    if (!(xf & 4)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = y0 + ty + 8 * k, x = x0 + tx;
            out[k] = (y < H && x < W) ? (float)plane[view_offset(xf, y, x, H, W)] : 0.f;
        }
        return;
    }
    __syncthreads();                                          // (the previous plane's readers are done with the tile)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + tx, x = x0 + ty + 8 * k;          // lanes along y: consecutive elements of the rotated view's row
        tile[(ty + 8 * k) * (VT + 1) + tx] = (y < H && x < W) ? (float)plane[view_offset(xf, y, x, H, W)] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = tile[tx * (VT + 1) + ty + 8 * k];      // pixel (y0 + ty + 8k, x0 + tx) sits at [x - x0][y - y0]
}

__global__ __launch_bounds__(256) void views_point_kernel(const float *__restrict__ points, int V, ViewXf xf, int H, int W,
                                                          float *__restrict__ point_mean, unsigned *pmax_key) {
    __shared__ float s_tile[VT * (VT + 1)];
    __shared__ float s_red[4];
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * VT, y0 = blockIdx.y * VT;
    const int plane = H * W;
    const float *pt = points + (size_t)img * V * plane;
    float sp[4], v[4];
    view_tile_load(pt, xf.v[0], H, W, y0, x0, tx, ty, s_tile, sp);
    for (int k = 1; k < V; ++k) {
        view_tile_load(pt + (size_t)k * plane, xf.v[k], H, W, y0, x0, tx, ty, s_tile, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) sp[j] = sp[j] + v[j];
    }
    const float fv = (float)V;
    float pm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + ty + 8 * j, x = x0 + tx;
        if (y < H && x < W) {
            const float m = sp[j] / fv;
            point_mean[(size_t)img * plane + (size_t)y * W + x] = m;
            pm = fmaxf(pm, m);
        }
    }
    pm = wave_maxf(pm);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        if (m != -INFINITY && f2key(m) > (unsigned)ld_relaxed(reinterpret_cast<const int *>(&pmax_key[img]))) atomicMax(&pmax_key[img], f2key(m));
    }
}

__global__ __launch_bounds__(256) void views_boost_kernel(const float *__restrict__ probs, const float *__restrict__ point_mean,
                                                          const uint8_t *__restrict__ codes, const int32_t *__restrict__ minmax, int V,
                                                          ViewXf xf, int H, int W, const unsigned *__restrict__ pmax_key,
                                                          float *__restrict__ prob_mean, uint8_t *__restrict__ ddm16, uint8_t *__restrict__ pred) {
    __shared__ float s_tile[VT * (VT + 1)];
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * VT, y0 = blockIdx.y * VT;
    const int plane = H * W;
    const float *pr = probs + (size_t)img * V * 3 * plane;
    const uint8_t *cd = codes + (size_t)img * V * plane;
    const int32_t *mm = minmax + (size_t)img * V * 2;
    float s0[4], s1[4], s2[4], v[4];
    double sd[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < V; ++k) {
        const float *q = pr + (size_t)k * 3 * plane;
        const int x_ = xf.v[k];
        if (k == 0) {
            view_tile_load(q, x_, H, W, y0, x0, tx, ty, s_tile, s0);
            view_tile_load(q + plane, x_, H, W, y0, x0, tx, ty, s_tile, s1);
            view_tile_load(q + 2 * plane, x_, H, W, y0, x0, tx, ty, s_tile, s2);
        } else {
            view_tile_load(q, x_, H, W, y0, x0, tx, ty, s_tile, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] = s0[j] + v[j];
            view_tile_load(q + plane, x_, H, W, y0, x0, tx, ty, s_tile, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) s1[j] = s1[j] + v[j];
            view_tile_load(q + 2 * plane, x_, H, W, y0, x0, tx, ty, s_tile, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) s2[j] = s2[j] + v[j];
        }
        view_tile_load(cd + (size_t)k * plane, x_, H, W, y0, x0, tx, ty, s_tile, v);       // (a code is an integer 0 .. 3: exact as a float)
        const float mn = (float)mm[2 * k], den = (float)(mm[2 * k + 1] - mm[2 * k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) sd[j] += (double)((v[j] - mn) / den);
    }
    const float pmax = key2f(pmax_key[img]);
    const float *pt = point_mean + (size_t)img * plane;
    const float fv = (float)V;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + ty + 8 * j, x = x0 + tx;
        if (y >= H || x >= W) continue;
        const int p = y * W + x;
        bool in3 = pt[p] / pmax > 0.2f;
        if (y > 0) in3 |= pt[p - W] / pmax > 0.2f;
        if (y < H - 1) in3 |= pt[p + W] / pmax > 0.2f;
        if (x > 0) in3 |= pt[p - 1] / pmax > 0.2f;
        if (x < W - 1) in3 |= pt[p + 1] / pmax > 0.2f;
        const double ddm = sd[j] / (double)V;
        if (ddm16) {
            double t = ddm * 16.0;
            ddm16[(size_t)img * plane + p] = (t >= 0.0 && t <= 254.0 && t == (double)(int)t) ? (uint8_t)(int)t : (uint8_t)255;
        }
        const double eb = 2.0 * (ddm - ddm * (double)(in3 ? 1 : 0));
        const float p0 = s0[j] / fv, p1 = s1[j] / fv;
        float p2 = s2[j] / fv;
        if (prob_mean) {
            float *dst = prob_mean + (size_t)img * 3 * plane;
            dst[p] = p0; dst[plane + p] = p1; dst[2 * plane + p] = p2;
        }
        p2 = (float)(((double)p2 + 0.5 * eb) * (1.0 + eb));
        int a = 0;
        float m = p0;
        if (p1 > m || (p1 != p1 && m == m)) { a = 1; m = p1; }
        if (p2 > m || (p2 != p2 && m == m)) { a = 2; m = p2; }
        pred[(size_t)img * plane + p] = (uint8_t)a;
    }
}

// ------------------------------------------------------------------------------------------------------
// Connected components: union-find over pixel indices, roots = smallest (raster-first) index of a component.
// One wave = 64 consecutive pixels of one row: the row runs come from a ballot, so only run heads talk to
// the forest.  Block (64,4); grid (ceil(W/64), ceil(H/4), N).  L: int32 per pixel, -1 = not in the mask.
// ------------------------------------------------------------------------------------------------------

__device__ __forceinline__ int uf_find(const int *L, int a) {
    int p = ld_relaxed(L + a);
    while (p != a) { a = p; p = ld_relaxed(L + a); }
    return a;
}

__device__ __forceinline__ void uf_union(int *L, int a, int b) {
    bool done;
    do {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) { int old = atomicMin(L + b, a); done = (old == b); b = old; }
        else if (b < a) { int old = atomicMin(L + a, b); done = (old == a); a = old; }
        else done = true;
    } while (!done);
}

// index of the first lane of the run of set bits that contains `lane`
__device__ __forceinline__ int run_start(unsigned long long m, int lane) {
    unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
    return zeros_below ? 64 - __clzll(zeros_below) : 0;
}
// number of set bits in the run starting at `lane` (lane is a run head)
__device__ __forceinline__ int run_length(unsigned long long m, int lane) {
    unsigned long long z = ~(m >> lane);          // first zero above
    return z ? __ffsll((long long)z) - 1 : 64 - lane;
}

// MODE 0: mask = (src != fgval)  [background of pred_inside, for fill-holes]; MODE 1: mask = (src != 0)
template <int MODE>
__device__ __forceinline__ bool in_mask(uint8_t v, int fgval) { return MODE == 0 ? (v != fgval) : (v != 0); }

template <int MODE>
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t *__restrict__ src, int fgval, int H, int W,
                                                      int *__restrict__ L) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const size_t base = (size_t)n * H * W;
    const bool valid = x < W && y < H;
    const bool fg = valid && in_mask<MODE>(src[base + (size_t)y * W + x], fgval);
    const unsigned long long b = __ballot(fg);
    if (valid) L[base + (size_t)y * W + x] = fg ? (y * W + blockIdx.x * 64 + run_start(b, threadIdx.x)) : -1;
}

// CONN: 4 or 8.  Unions between a row run and the runs of the row above, plus the stitch to the left segment.
template <int MODE, int CONN>
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t *__restrict__ src, int fgval, int H, int W,
                                                       int *L) {
    const int n = blockIdx.z;
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, x = x0 + lane, y = blockIdx.y * 4 + threadIdx.y;
    if (y >= H) return;                                   // whole wave exits together (y is wave-uniform)
    const size_t base = (size_t)n * H * W;
    const uint8_t *m = src + base;
    int *Ln = L + base;
    const bool valid = x < W;
    const bool fg = valid && in_mask<MODE>(m[(size_t)y * W + x], fgval);
    const bool up = valid && y > 0 && in_mask<MODE>(m[(size_t)(y - 1) * W + x], fgval);
    // edge pixels outside this 64-segment
    bool left_edge = false, upleft_edge = false, upright_edge = false;
    if (lane == 0 && x0 > 0) {
        left_edge = in_mask<MODE>(m[(size_t)y * W + x0 - 1], fgval);
        if (y > 0) upleft_edge = in_mask<MODE>(m[(size_t)(y - 1) * W + x0 - 1], fgval);
    }
    if (lane == 63 && x0 + 64 < W && y > 0) upright_edge = in_mask<MODE>(m[(size_t)(y - 1) * W + x0 + 64], fgval);
    const unsigned long long bf = __ballot(fg), bu = __ballot(up);
    if (!fg) return;
    const bool left = lane > 0 ? ((bf >> (lane - 1)) & 1ull) : left_edge;
    const bool a = lane > 0 ? ((bu >> (lane - 1)) & 1ull) : upleft_edge;      // NW
    const bool b = (bu >> lane) & 1ull;                                        // N
    const bool c = lane < 63 ? ((bu >> (lane + 1)) & 1ull) : upright_edge;    // NE
    const int p = y * W + x;
    if (lane == 0 && left_edge) uf_union(Ln, p, p - 1);
    if (CONN == 4) {
        if (b && !(left && a)) uf_union(Ln, p, p - W);
    } else {
        if (b) { if (!left) uf_union(Ln, p, p - W); }
        else {
            if (a && !left) uf_union(Ln, p, p - W - 1);
            if (c) uf_union(Ln, p, p - W + 1);
        }
    }
}

// L[p] <- root(p).  AREA: additionally area[root] += run length (one atomic per row run).
template <bool AREA>
__global__ __launch_bounds__(256) void cc_flatten_kernel(int H, int W, int *L, int *area) {
    const int n = blockIdx.z;
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + threadIdx.y;
    const size_t base = (size_t)n * H * W;
    int *Ln = L + base;
    const bool valid = x < W && y < H;
    int r = -1;
    if (valid) {
        int l = Ln[(size_t)y * W + x];
        if (l >= 0) { r = uf_find(Ln, l); }
    }
    const unsigned long long bf = __ballot(r >= 0);
    if (r >= 0) {
        Ln[(size_t)y * W + x] = r;
        if (AREA) {
            bool head = lane == 0 || !((bf >> (lane - 1)) & 1ull);
            if (head) atomicAdd(area + base + r, run_length(bf, lane));
        }
    }
}

// fill-holes: mark the roots of background components that touch the image border (L[root] = ~root < 0 ... -1 is
// "not background", so use -2-root).
__global__ __launch_bounds__(256) void fill_mark_border_kernel(int H, int W, int *L) {
    const int n = blockIdx.y;
    int *Ln = L + (size_t)n * H * W;
    const int per = 2 * W + 2 * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        int y, x;
        if (i < W) { y = 0; x = i; }
        else if (i < 2 * W) { y = H - 1; x = i - W; }
        else if (i < 2 * W + H) { y = i - 2 * W; x = 0; }
        else { y = i - 2 * W - H; x = W - 1; }
        int l = ld_relaxed(Ln + (size_t)y * W + x);
        if (l >= 0) {                     // background pixel whose root is not marked yet (as far as we saw)
            int rl = ld_relaxed(Ln + l);  // l is a root after flatten: rl == l, or already marked
            if (rl >= 0) __hip_atomic_store(Ln + l, -2 - l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// out = foreground OR background component not connected to the border
__global__ __launch_bounds__(256) void fill_output_kernel(const uint8_t *__restrict__ src, int fgval, int plane,
                                                          const int *__restrict__ L, uint8_t *__restrict__ out) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        int l = L[base + i];
        uint8_t o;
        if (src[base + i] == fgval) o = 1;
        else {
            bool outer;
            if (l == -1) outer = false;                       // cannot happen for background, keep defined
            else if (l < -1) outer = true;                    // a marked root itself
            else outer = L[base + l] < -1;
            o = outer ? 0 : 1;
        }
        out[base + i] = o;
    }
}

// remove-small + diagonal merge: components (4-conn) with area < min_area are dropped; the survivors get the
// extra NW/NE unions that turn the 4-connected forest into the 8-connected one (skimage.measure.label).
__global__ __launch_bounds__(256) void cc_diag_merge_kernel(const uint8_t *__restrict__ A, int H, int W, int min_area,
                                                            const int *__restrict__ area, int *L,
                                                            uint8_t *__restrict__ small) {   // small: required
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t base = (size_t)n * H * W;
    const uint8_t *m = A + base;
    int *Ln = L + base;
    const int p = y * W + x;
    const int l = ld_relaxed(Ln + p);            // flattened label (root of the 4-connected component) or -1
    // `area` is non-zero exactly at the roots of the 4-connected forest (cc_flatten<AREA>).  Entries of non-root
    // pixels are never touched by uf_union, so for them l is still that root; a root pixel's own entry may already
    // have been lowered by a concurrent diagonal union, hence the area[p] test instead of (l == p).
    bool keep = false;
    if (l >= 0) {
        const int r0 = area[base + p] > 0 ? p : l;
        keep = area[base + r0] >= min_area;
    }
    small[base + p] = keep ? 1 : 0;
    if (!keep) return;
    if (y == 0) return;
    const bool N = m[p - W] != 0;
    if (N) return;                               // NW / NE already 4-connected to p through N
    const bool Wp = x > 0 && m[p - 1] != 0;
    const bool Ep = x < W - 1 && m[p + 1] != 0;
    if (x > 0 && !Wp && m[p - W - 1] != 0) {
        int q = p - W - 1;
        int rq = area[base + q] > 0 ? q : ld_relaxed(Ln + q);
        if (area[base + rq] >= min_area) uf_union(Ln, p, q);
    }
    if (x < W - 1 && !Ep && m[p - W + 1] != 0) {
        int q = p - W + 1;
        int rq = area[base + q] > 0 ? q : ld_relaxed(Ln + q);
        if (area[base + rq] >= min_area) uf_union(Ln, p, q);
    }
}

// second flatten: survivors point at the root of their 8-connected component, everything else is -1
__global__ __launch_bounds__(256) void cc_flatten_kept_kernel(const uint8_t *__restrict__ keep, int plane, int *L) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    int *Ln = L + base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        if (keep[base + i]) Ln[i] = uf_find(Ln, i);
        else Ln[i] = -1;
    }
}

// raster-order numbering of the roots: chunk = 1024 consecutive pixels, 4 per thread
constexpr int CHUNK = 1024;

__device__ __forceinline__ int block_excl_scan(int v, int *total, int *s_w) {
    // 256 threads, returns exclusive prefix of v over the block in thread order
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < w) woff += s_w[i]; tot += s_w[i]; }
    *total = tot;
    return woff + inc - v;
}

__global__ __launch_bounds__(256) void cc_count_roots_kernel(const int *__restrict__ L, int plane, int nchunk,
                                                             int *__restrict__ chunk_cnt) {
    __shared__ int s_w[4];
    const int n = blockIdx.y, c = blockIdx.x;
    const int *Ln = L + (size_t)n * plane;
    int cnt = 0;
    const int p0 = c * CHUNK + threadIdx.x * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { int p = p0 + j; if (p < plane && Ln[p] == p) ++cnt; }
    int tot;
    block_excl_scan(cnt, &tot, s_w);
    if (threadIdx.x == 0) chunk_cnt[(size_t)n * nchunk + c] = tot;
}

// one block per image: exclusive scan of the chunk counts (in place) + instance count
__global__ __launch_bounds__(256) void cc_scan_chunks_kernel(int nchunk, int *chunk_cnt, int *__restrict__ counts) {
    __shared__ int s_w[4];
    __shared__ int s_carry;
    const int n = blockIdx.x;
    int *c = chunk_cnt + (size_t)n * nchunk;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nchunk; b0 += 256) {
        int i = b0 + threadIdx.x;
        int v = i < nchunk ? c[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot, s_w);
        int carry = s_carry;
        if (i < nchunk) c[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && counts) counts[n] = s_carry;
}

__global__ __launch_bounds__(256) void cc_rank_roots_kernel(const int *__restrict__ L, int plane, int nchunk,
                                                            const int *__restrict__ chunk_off, int *__restrict__ rank) {
    __shared__ int s_w[4];
    const int n = blockIdx.y, c = blockIdx.x;
    const int *Ln = L + (size_t)n * plane;
    int *rk = rank + (size_t)n * plane;
    bool f[4];
    int cnt = 0;
    const int p0 = c * CHUNK + threadIdx.x * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { int p = p0 + j; f[j] = p < plane && Ln[p] == p; cnt += f[j] ? 1 : 0; }
    int tot;
    int ex = block_excl_scan(cnt, &tot, s_w) + chunk_off[(size_t)n * nchunk + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (f[j]) { ++ex; rk[p0 + j] = ex; }
}

__global__ __launch_bounds__(256) void cc_relabel_kernel(const int *__restrict__ L, const int *__restrict__ rank,
                                                         int plane, int32_t *__restrict__ label) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        int l = L[base + i];
        label[base + i] = l >= 0 ? rank[base + l] : 0;
    }
}

// grey dilation by disk(r): max label over {dy^2+dx^2 <= r^2} (skimage.morphology.dilation, test_dam.py:563)
__global__ __launch_bounds__(256) void dilate_disk_kernel(const int32_t *__restrict__ label, int H, int W, int r,
                                                          int32_t *__restrict__ out) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int32_t *src = label + (size_t)n * H * W;
    int32_t v = src[(size_t)y * W + x];
    for (int dy = -r; dy <= r; ++dy) {
        int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            if (dy * dy + dx * dx > r * r) continue;
            int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            int32_t q = src[(size_t)yy * W + xx];
            v = q > v ? q : v;
        }
    }
    out[(size_t)n * H * W + (size_t)y * W + x] = v;
}

inline dim3 grid_rows(int N, int H, int W) { return dim3(cdiv(W, 64), cdiv(H, 4), N); }
inline int grid_lin(int plane) { int g = cdiv(plane, 256); return g > 2048 ? 2048 : (g < 1 ? 1 : g); }

}  // namespace

// internal (not part of the C ABI): 8-connected components of a binary mask with skimage.measure.label numbering.
// L, aux: int32 [N*H*W] scratch; chunk: int32 [N*ceil(H*W/1024)]; labels may alias L.
namespace cdnet {
int label8_raster(const uint8_t *mask, int N, int H, int W, int *L, int *aux, int *chunk, int32_t *labels, int32_t *counts,
                  hipStream_t st) {
    {
        int rc = 0;                                          // tiles: one launch with the tile in LDS (postproc_tile.hip)
        if (label8_tile(mask, N, H, W, labels, counts, st, &rc)) return rc;
    }
    const int plane = H * W, nchunk = cdiv(plane, CHUNK);
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const dim3 gl(grid_lin(plane), N);
    cc_init_kernel<1><<<gr, br, 0, st>>>(mask, 0, H, W, L);
    cc_merge_kernel<1, 8><<<gr, br, 0, st>>>(mask, 0, H, W, L);
    cc_flatten_kernel<false><<<gr, br, 0, st>>>(H, W, L, nullptr);
    cc_count_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L, plane, nchunk, chunk);
    cc_scan_chunks_kernel<<<N, 256, 0, st>>>(nchunk, chunk, counts);
    cc_rank_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L, plane, nchunk, chunk, aux);
    cc_relabel_kernel<<<gl, 256, 0, st>>>(L, aux, plane, labels);
    return check_launch("label8_raster");
}
}  // namespace cdnet

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" int cdnet_ddm_codes(const uint8_t *dcm, int N, int H, int W, int classes, const int8_t *lut_host, int nbr,
                               int extra_zero, uint8_t *code, int32_t *minmax, void *stream) {
    CDNET_REQUIRE(dcm && code && minmax && lut_host, "cdnet_ddm_codes: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0, "cdnet_ddm_codes: bad size N=%d H=%d W=%d", N, H, W);
    CDNET_REQUIRE(classes >= 2 && classes <= 17, "cdnet_ddm_codes: classes=%d not in [2,17]", classes);
    CDNET_REQUIRE(nbr == 4 || nbr == 8, "cdnet_ddm_codes: nbr must be 4 or 8");
    hipStream_t st = (hipStream_t)stream;
    DdmRows rows;
    for (int c = 0; c < 17; ++c) {
        rows.r[c] = 0ull;
        for (int k = 0; k < 17; ++k) {
            const int v = (c < classes && k < classes) ? lut_host[c * classes + k] : 0;
            CDNET_REQUIRE(v >= -1 && v <= 1, "cdnet_ddm_codes: table entry %d outside {-1, 0, 1} (rounded cosines)", v);
            rows.r[c] |= (unsigned long long)(v + 1) << (2 * k);
        }
    }
    init_minmax_kernel<<<cdiv(N, 256), 256, 0, st>>>(minmax, N);
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    ddm_codes_kernel<<<grid, dim3(64, 4), 0, st>>>(dcm, H, W, classes, rows, nbr, extra_zero, code, minmax);
    finish_minmax_kernel<<<cdiv(N, 256), 256, 0, st>>>(minmax, N);
    return check_launch("cdnet_ddm_codes");
}

extern "C" int cdnet_ddm_normalize(const uint8_t *code, const int32_t *minmax, int N, int H, int W, float *out,
                                   void *stream) {
    CDNET_REQUIRE(code && minmax && out, "cdnet_ddm_normalize: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0, "cdnet_ddm_normalize: bad size");
    ddm_normalize_kernel<<<dim3(grid_lin(H * W), N), 256, 0, (hipStream_t)stream>>>(code, minmax, H * W, out);
    return check_launch("cdnet_ddm_normalize");
}

extern "C" int cdnet_probmaps(const float *mask_logits, const float *dir_logits, int N, int C, int H, int W,
                              float *prob, uint8_t *dcm, void *stream) {
    CDNET_REQUIRE(mask_logits && dir_logits && prob && dcm, "cdnet_probmaps: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0, "cdnet_probmaps: bad size");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(grid_lin(H * W), N);
    if (C == 9) probmaps_kernel<9><<<grid, 256, 0, st>>>(mask_logits, dir_logits, H * W, prob, dcm);
    else if (C == 5) probmaps_kernel<5><<<grid, 256, 0, st>>>(mask_logits, dir_logits, H * W, prob, dcm);
    else if (C == 17) probmaps_kernel<17><<<grid, 256, 0, st>>>(mask_logits, dir_logits, H * W, prob, dcm);
    else CDNET_REQUIRE(false, "cdnet_probmaps: direction classes %d not in {5,9,17}", C);
    return check_launch("cdnet_probmaps");
}

extern "C" int cdnet_tta_boost_argmax(const float *probs, const float *points, const uint8_t *codes,
                                      const int32_t *minmax, int I, int V, const int *view_xform_host, int H, int W,
                                      float *prob_mean, float *point_mean, uint8_t *ddm16, uint8_t *pred,
                                      float *pmax_ws, void *stream) {
    CDNET_REQUIRE(probs && points && codes && minmax && pred && pmax_ws && view_xform_host, "cdnet_tta_boost_argmax: null pointer");
    // point_mean may be NULL for ONE view in its own frame (V == 1, view_xform 0): the mean over views is the view itself
    const bool single = V == 1 && view_xform_host[0] == 0 && !point_mean && !prob_mean;
    CDNET_REQUIRE(point_mean || single, "cdnet_tta_boost_argmax: point_mean is required unless V == 1, view_xform[0] == 0 and prob_mean == NULL");
    CDNET_REQUIRE(I > 0 && H > 0 && W > 0 && V >= 1 && V <= 16, "cdnet_tta_boost_argmax: bad size I=%d V=%d", I, V);
    hipStream_t st = (hipStream_t)stream;
    ViewXf xf;
    for (int v = 0; v < 16; ++v) {
        xf.v[v] = v < V ? view_xform_host[v] : 0;
        CDNET_REQUIRE(xf.v[v] >= 0 && xf.v[v] < 8, "cdnet_tta_boost_argmax: view_xform[%d]=%d", v, xf.v[v]);
    }
    if (hipMemsetAsync(pmax_ws, 0, sizeof(float) * I, st) != hipSuccess) return check_launch("memset pmax");
    dim3 grid = grid_rows(I, H, W);
    if (single && ((size_t)points & 15) == 0 && ((H * W) & 3) == 0) {
        // (x / 1.0f == x: the point map itself is the mean boost_argmax_kernel reads)
        int g = cdiv(H * W / 4, 256); if (g > 64) g = 64;
        point_max_kernel<<<dim3(g, I), 256, 0, st>>>(points, H * W, reinterpret_cast<unsigned *>(pmax_ws));
        point_mean = const_cast<float *>(points);
        boost_argmax_kernel<<<grid, dim3(64, 4), 0, st>>>(probs, prob_mean, point_mean, codes, minmax, V, xf, H, W,
                                                          reinterpret_cast<const unsigned *>(pmax_ws), ddm16, pred);
    } else {
        CDNET_REQUIRE(point_mean, "cdnet_tta_boost_argmax: point_mean is required for this shape (pixel count not a multiple of 4 or unaligned points)");
        const dim3 gt(cdiv(W, VT), cdiv(H, VT), I);
        views_point_kernel<<<gt, 256, 0, st>>>(points, V, xf, H, W, point_mean, reinterpret_cast<unsigned *>(pmax_ws));
        views_boost_kernel<<<gt, 256, 0, st>>>(probs, point_mean, codes, minmax, V, xf, H, W, reinterpret_cast<const unsigned *>(pmax_ws), prob_mean,
                                               ddm16, pred);
    }
    return check_launch("cdnet_tta_boost_argmax");
}

// workspace layout (per call): L i32[N*P] | aux i32[N*P] | A u8[N*P] | B u8[N*P] | chunk i32[N*nchunk]
static size_t cc_ws_layout(int N, int H, int W, size_t *oL, size_t *oAux, size_t *oA, size_t *oB, size_t *oC) {
    size_t P = (size_t)N * H * W, off = 0;
    *oL = off; off = align_up(off + P * 4, 256);
    *oAux = off; off = align_up(off + P * 4, 256);
    *oA = off; off = align_up(off + P, 256);
    *oB = off; off = align_up(off + P, 256);
    *oC = off; off = align_up(off + (size_t)N * cdiv(H * W, CHUNK) * 4, 256);
    return off;
}

extern "C" size_t cdnet_cc_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    size_t a, b, c, d, e;
    return cc_ws_layout(N, H, W, &a, &b, &c, &d, &e);
}

extern "C" int cdnet_cc_chain(const uint8_t *pred, int fg_value, int N, int H, int W, int min_area, int radius,
                              void *workspace, size_t workspace_bytes, uint8_t *fill, uint8_t *small, int32_t *label,
                              int32_t *final_, int32_t *counts, void *stream) {
    CDNET_REQUIRE(pred && final_ && workspace, "cdnet_cc_chain: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0, "cdnet_cc_chain: bad size N=%d H=%d W=%d", N, H, W);
    CDNET_REQUIRE((size_t)H * W < (1u << 30), "cdnet_cc_chain: image too large for 32-bit pixel indices");
    CDNET_REQUIRE(radius >= 0 && radius <= 8, "cdnet_cc_chain: radius %d not in [0,8]", radius);
    size_t oL, oAux, oA, oB, oC;
    size_t need = cc_ws_layout(N, H, W, &oL, &oAux, &oA, &oB, &oC);
    if (workspace_bytes < need) {
        set_error("cdnet_cc_chain: workspace %zu < %zu bytes", workspace_bytes, need);
        return CDNET_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *L = (int *)(ws + oL), *aux = (int *)(ws + oAux), *chunk = (int *)(ws + oC);
    uint8_t *A = fill ? fill : (uint8_t *)(ws + oA);
    uint8_t *B = small ? small : (uint8_t *)(ws + oB);
    const int plane = H * W, nchunk = cdiv(plane, CHUNK);
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const dim3 gl(grid_lin(plane), N);

    // 1. fill holes: 4-connected components of the background; those touching the border stay background
    cc_init_kernel<0><<<gr, br, 0, st>>>(pred, fg_value, H, W, L);
    cc_merge_kernel<0, 4><<<gr, br, 0, st>>>(pred, fg_value, H, W, L);
    cc_flatten_kernel<false><<<gr, br, 0, st>>>(H, W, L, nullptr);
    fill_mark_border_kernel<<<dim3(cdiv(2 * (H + W), 256), N), 256, 0, st>>>(H, W, L);
    fill_output_kernel<<<gl, 256, 0, st>>>(pred, fg_value, plane, L, A);
    // 2. remove small objects: 4-connected components of A with their areas
    if (hipMemsetAsync(aux, 0, (size_t)N * plane * 4, st) != hipSuccess) return check_launch("memset area");
    cc_init_kernel<1><<<gr, br, 0, st>>>(A, 0, H, W, L);
    cc_merge_kernel<1, 4><<<gr, br, 0, st>>>(A, 0, H, W, L);
    cc_flatten_kernel<true><<<gr, br, 0, st>>>(H, W, L, aux);
    // 3. drop small components, add the diagonal unions (8-connectivity) among the survivors, number in raster order
    cc_diag_merge_kernel<<<gr, br, 0, st>>>(A, H, W, min_area, aux, L, B);
    cc_flatten_kept_kernel<<<gl, 256, 0, st>>>(B, plane, L);
    cc_count_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L, plane, nchunk, chunk);
    cc_scan_chunks_kernel<<<N, 256, 0, st>>>(nchunk, chunk, counts);
    cc_rank_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L, plane, nchunk, chunk, aux);
    // 4. labels and the disk dilation
    int32_t *lab = label ? label : (int32_t *)L;      // in-place relabel is safe: rank lives in aux
    if (label) cc_relabel_kernel<<<gl, 256, 0, st>>>(L, aux, plane, lab);
    else {
        // L is overwritten element-wise with rank[L[i]]; reads of L[i] and writes of lab[i] touch the same element only
        cc_relabel_kernel<<<gl, 256, 0, st>>>(L, aux, plane, lab);
    }
    dilate_disk_kernel<<<gr, br, 0, st>>>(lab, H, W, radius, final_);
    return check_launch("cdnet_cc_chain");
}

// ======================================================================================================
// Watershed variant of the post-processing (postproc_other.py:15-99, ws branch :36-48)
//   dist   = per-instance Euclidean distance transform scaled to 0..255 (gen_inst_dst_map :16-27)
//   marker = label4(erode4(fill_holes(dist > 125))) with labels smaller than min_size dropped
//   out    = watershed(-dist as uint8, marker, mask = pred), labels smaller than min_size dropped
// Every step is integer / exactly-rounded fp64, so the result is bit-exact against the restatement in
// oracle/postproc_oracle.c; steps up to the marker are pinned against scipy itself (tests/test_oracle_watershed.py).
// The flood reproduces skimage's priority (value, then age); only its tie-break between marker pixels of equal value
// and age 0 is unpinned (skimage absent): here raster order.
// ======================================================================================================
namespace {

constexpr int WS_INF = 1 << 14;

// h[p] = horizontal distance from p to the nearest pixel of its row whose component differs (WS_INF if none)
__global__ __launch_bounds__(256) void ws_rowdist_kernel(const int *__restrict__ L, int H, int W, int *__restrict__ h) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int k = L[base + i];
        int best = WS_INF;
        if (k >= 0) {
            const int *row = L + base + (size_t)y * W;
            for (int d = 1; d < best; ++d) {
                const bool l = x - d >= 0, r = x + d < W;
                if (!l && !r) break;
                if ((l && row[x - d] != k) || (r && row[x + d] != k)) { best = d; break; }
            }
        }
        h[base + i] = best;
    }
}

// exact squared distance to the nearest pixel outside the own component: min over rows y' of dy^2 + (same ? h^2 : 0);
// per-component maximum by atomicMax on the component's root slot
__global__ __launch_bounds__(256) void ws_edt_kernel(const int *__restrict__ L, const int *__restrict__ h, int H, int W,
                                                     int *__restrict__ d2, int *__restrict__ maxd2) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int k = L[base + i];
        if (k < 0) { d2[base + i] = 0; continue; }
        long long best = (long long)h[base + i] * h[base + i];
        for (int dy = 1; (long long)dy * dy < best; ++dy) {
            bool any = false;
#pragma unroll
            for (int s = -1; s <= 1; s += 2) {
                const int yy = y + s * dy;
                if (yy < 0 || yy >= H) continue;
                any = true;
                const size_t q = base + (size_t)yy * W + x;
                const long long c = (long long)dy * dy + (L[q] == k ? (long long)h[q] * h[q] : 0);
                best = c < best ? c : best;
            }
            if (!any) break;
        }
        const int v = best > 0x3fffffff ? 0x3fffffff : (int)best;
        d2[base + i] = v;
        atomicMax(maxd2 + base + k, v);
    }
}

// canvas = uint8(255 * (sqrt(d2) / sqrt(max d2))) in fp64 (numpy: 255 * (nuc_dst / np.amax(nuc_dst)), astype uint8);
// marker mask = canvas > 125
__global__ __launch_bounds__(256) void ws_canvas_kernel(const int *__restrict__ L, const int *__restrict__ d2,
                                                        const int *__restrict__ maxd2, int plane, uint8_t *__restrict__ canvas,
                                                        uint8_t *__restrict__ mk) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int k = L[base + i];
        uint8_t c = 0;
        if (k >= 0) {
            const double d = sqrt((double)d2[base + i]), m = sqrt((double)maxd2[base + k]);
            c = (uint8_t)(255.0 * (d / m));
        }
        canvas[base + i] = c;
        mk[base + i] = c > 125 ? 1 : 0;
    }
}

// binary erosion by the 4-neighbour cross, outside = 0 (scipy binary_erosion defaults)
__global__ __launch_bounds__(256) void ws_erode4_kernel(const uint8_t *__restrict__ a, int H, int W, uint8_t *__restrict__ out) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const uint8_t *p = a + base;
        const bool v = p[i] && y > 0 && y < H - 1 && x > 0 && x < W - 1 && p[i - W] && p[i + W] && p[i - 1] && p[i + 1];
        out[base + i] = v ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void ws_hist_kernel(const int32_t *__restrict__ lab, int plane, int *__restrict__ area) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int l = lab[base + i];
        if (l > 0) atomicAdd(area + base + l, 1);          // labels <= plane / 2 < plane
    }
}

// remove_small_objects on a label image: labels with fewer than min_size pixels -> 0 (ids are kept); also records the
// last pixel (raster order) of every component of `comp` for the flood
__global__ __launch_bounds__(256) void ws_drop_small_kernel(int32_t *__restrict__ lab, const int *__restrict__ area, int plane,
                                                            int min_size) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int l = lab[base + i];
        if (l > 0 && area[base + l] < min_size) lab[base + i] = 0;
    }
}

__global__ __launch_bounds__(256) void ws_last_kernel(const int *__restrict__ L, int plane, int *__restrict__ last) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int k = L[base + i];
        if (k >= 0) atomicMax(last + base + k, i);
    }
}

// The flood.  4-connected components of the mask are independent, so one thread owns one component (its root is its
// raster-first pixel) and runs skimage's sequential algorithm on it: seeds = the component's marker pixels in raster
// order; pop the smallest (value, age); every unlabelled 4-neighbour inside the mask (order: up, left, right, down)
// takes the label and is pushed with its own value.  Values are uint8, so the priority queue is 256 FIFO buckets
// (linked through next[]): FIFO order inside a bucket IS age order.
__global__ __launch_bounds__(64) void ws_flood_kernel(const int *__restrict__ L, const int *__restrict__ last,
                                                      const uint8_t *__restrict__ canvas, const int32_t *__restrict__ marker,
                                                      int H, int W, int *__restrict__ next, int32_t *__restrict__ out) {
    const int n = blockIdx.y;
    const int plane = H * W;
    const size_t base = (size_t)n * plane;
    const int root = blockIdx.x * blockDim.x + threadIdx.x;
    if (root >= plane || L[base + root] != root) return;
    const int *Ln = L + base;
    const uint8_t *cv = canvas + base;
    int *nx = next + base;
    int32_t *o = out + base;
    int head[256], tail[256];
    for (int b = 0; b < 256; ++b) head[b] = -1;
    int lo = 256;
    const int end = last[base + root];
    for (int i = root; i <= end; ++i) {
        if (Ln[i] != root) continue;
        const int m = marker[base + i];
        if (m <= 0) continue;
        o[i] = m;
        const int v = (256 - cv[i]) & 255;                 // uint8 negation of the distance map (postproc_other.py:47)
        nx[i] = -1;
        if (head[v] < 0) head[v] = i; else nx[tail[v]] = i;
        tail[v] = i;
        lo = v < lo ? v : lo;
    }
    while (true) {
        while (lo < 256 && head[lo] < 0) ++lo;
        if (lo >= 256) break;
        const int p = head[lo];
        head[lo] = nx[p];
        const int lab = o[p];
        const int y = p / W, x = p - y * W;
        const int nb[4] = {y > 0 ? p - W : -1, x > 0 ? p - 1 : -1, x < W - 1 ? p + 1 : -1, y < H - 1 ? p + W : -1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = nb[j];
            if (q < 0 || Ln[q] != root || o[q] != 0) continue;     // outside the mask component / already labelled
            o[q] = lab;
            const int v = (256 - cv[q]) & 255;
            nx[q] = -1;
            if (head[v] < 0) head[v] = q; else nx[tail[v]] = q;
            tail[v] = q;
            lo = v < lo ? v : lo;
        }
    }
}

__global__ __launch_bounds__(256) void ws_binarize_kernel(const uint8_t *__restrict__ in, int plane, uint8_t *__restrict__ out) {
    const size_t base = (size_t)blockIdx.y * plane;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) out[base + i] = in[base + i] ? 1 : 0;
}

size_t ws_layout(int N, int H, int W, size_t o[10]) {
    const size_t P = (size_t)N * H * W;
    size_t off = 0;
    for (int k = 0; k < 6; ++k) { o[k] = off; off = align_up(off + P * 4, 256); }      // L1, L2, aux, h/area, d2/next, maxd2/last
    for (int k = 6; k < 9; ++k) { o[k] = off; off = align_up(off + P, 256); }          // canvas, m, m2
    o[9] = off; off = align_up(off + (size_t)N * cdiv(H * W, CHUNK) * 4, 256);          // chunk counters
    return off;
}

// 4-connected components of a u8 mask: roots in L (raster-first pixel of each component, -1 off-mask)
void label4_roots(const uint8_t *mask, int N, int H, int W, int *L, hipStream_t st) {
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    cc_init_kernel<1><<<gr, br, 0, st>>>(mask, 0, H, W, L);
    cc_merge_kernel<1, 4><<<gr, br, 0, st>>>(mask, 0, H, W, L);
    cc_flatten_kernel<false><<<gr, br, 0, st>>>(H, W, L, nullptr);
}

}  // namespace

extern "C" size_t cdnet_watershed_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    size_t o[10];
    return ws_layout(N, H, W, o);
}

extern "C" int cdnet_watershed_process(const uint8_t *pred, int N, int H, int W, int min_size, void *workspace, size_t workspace_bytes,
                                       uint8_t *dist_out, int32_t *marker_out, int32_t *labels, void *stream) {
    CDNET_REQUIRE(pred && labels && workspace, "cdnet_watershed_process: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0 && (size_t)H * W < (1u << 30) && H < WS_INF && W < WS_INF, "cdnet_watershed_process: bad size");
    size_t o[10];
    const size_t need = ws_layout(N, H, W, o);
    if (workspace_bytes < need) { set_error("cdnet_watershed_process: workspace %zu < %zu bytes", workspace_bytes, need); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *L1 = (int *)(ws + o[0]), *L2 = (int *)(ws + o[1]), *aux = (int *)(ws + o[2]), *hb = (int *)(ws + o[3]), *d2 = (int *)(ws + o[4]),
        *mx = (int *)(ws + o[5]), *chunk = (int *)(ws + o[9]);
    uint8_t *canvas = dist_out ? dist_out : (uint8_t *)(ws + o[6]), *m = (uint8_t *)(ws + o[7]), *m2 = (uint8_t *)(ws + o[8]);
    const int plane = H * W, nchunk = cdiv(plane, CHUNK);
    const size_t P = (size_t)N * plane;
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const dim3 gl(grid_lin(plane), N);
    int32_t *marker = marker_out ? marker_out : (int32_t *)L2;

    // 1. components of the prediction (measurements.label, 4-connected), exact EDT per component, scaled distance map
    label4_roots(pred, N, H, W, L1, st);
    if (hipMemsetAsync(mx, 0, P * 4, st) != hipSuccess) return check_launch("memset");
    ws_rowdist_kernel<<<gl, 256, 0, st>>>(L1, H, W, hb);
    ws_edt_kernel<<<gl, 256, 0, st>>>(L1, hb, H, W, d2, mx);
    ws_canvas_kernel<<<gl, 256, 0, st>>>(L1, d2, mx, plane, canvas, m);
    // 2. marker: fill holes, erode, label (raster numbering), drop small labels
    cc_init_kernel<0><<<gr, br, 0, st>>>(m, 1, H, W, L2);
    cc_merge_kernel<0, 4><<<gr, br, 0, st>>>(m, 1, H, W, L2);
    cc_flatten_kernel<false><<<gr, br, 0, st>>>(H, W, L2, nullptr);
    fill_mark_border_kernel<<<dim3(cdiv(2 * (H + W), 256), N), 256, 0, st>>>(H, W, L2);
    fill_output_kernel<<<gl, 256, 0, st>>>(m, 1, plane, L2, m2);
    ws_erode4_kernel<<<gl, 256, 0, st>>>(m2, H, W, m);
    label4_roots(m, N, H, W, L2, st);
    cc_count_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L2, plane, nchunk, chunk);
    cc_scan_chunks_kernel<<<N, 256, 0, st>>>(nchunk, chunk, nullptr);
    cc_rank_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L2, plane, nchunk, chunk, aux);
    cc_relabel_kernel<<<gl, 256, 0, st>>>(L2, aux, plane, marker);          // in place when marker aliases L2
    if (hipMemsetAsync(hb, 0, P * 4, st) != hipSuccess) return check_launch("memset");
    ws_hist_kernel<<<gl, 256, 0, st>>>(marker, plane, hb);
    ws_drop_small_kernel<<<gl, 256, 0, st>>>(marker, hb, plane, min_size);
    // 3. flood per component of the prediction, then drop small labels
    if (hipMemsetAsync(mx, 0xff, P * 4, st) != hipSuccess) return check_launch("memset");      // last = -1
    if (hipMemsetAsync(labels, 0, P * 4, st) != hipSuccess) return check_launch("memset");
    ws_last_kernel<<<gl, 256, 0, st>>>(L1, plane, mx);
    ws_flood_kernel<<<dim3(cdiv(plane, 64), N), 64, 0, st>>>(L1, mx, canvas, marker, H, W, d2, labels);
    if (hipMemsetAsync(hb, 0, P * 4, st) != hipSuccess) return check_launch("memset");
    ws_hist_kernel<<<gl, 256, 0, st>>>(labels, plane, hb);
    ws_drop_small_kernel<<<gl, 256, 0, st>>>(labels, hb, plane, min_size);
    return check_launch("cdnet_watershed_process");
}

// postproc_other.process, ws = False branch (postproc_other.py:49-52; forced for model_mode 'unet' / 'micronet', :35):
//   scipy.ndimage.binary_fill_holes -> measurements.label (4-connected, ids in raster order of each component's first pixel)
//   -> skimage remove_small_objects on the LABEL image (labels with fewer than min_size pixels become 0, ids are kept).
// Same workspace as the watershed variant.
extern "C" int cdnet_fill_label_process(const uint8_t *pred, int N, int H, int W, int min_size, void *workspace, size_t workspace_bytes,
                                        int32_t *labels, void *stream) {
    CDNET_REQUIRE(pred && labels && workspace, "cdnet_fill_label_process: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0 && (size_t)H * W < (1u << 30), "cdnet_fill_label_process: bad size");
    size_t o[10];
    const size_t need = ws_layout(N, H, W, o);
    if (workspace_bytes < need) { set_error("cdnet_fill_label_process: workspace %zu < %zu bytes", workspace_bytes, need); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *L2 = (int *)(ws + o[1]), *aux = (int *)(ws + o[2]), *hb = (int *)(ws + o[3]), *chunk = (int *)(ws + o[9]);
    uint8_t *m = (uint8_t *)(ws + o[7]), *m2 = (uint8_t *)(ws + o[8]);
    const int plane = H * W, nchunk = cdiv(plane, CHUNK);
    const size_t P = (size_t)N * plane;
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const dim3 gl(grid_lin(plane), N);
    ws_binarize_kernel<<<gl, 256, 0, st>>>(pred, plane, m);                           // non-zero -> 1
    // fill holes: 4-connected components of the background, those touching the border stay background
    cc_init_kernel<0><<<gr, br, 0, st>>>(m, 1, H, W, L2);
    cc_merge_kernel<0, 4><<<gr, br, 0, st>>>(m, 1, H, W, L2);
    cc_flatten_kernel<false><<<gr, br, 0, st>>>(H, W, L2, nullptr);
    fill_mark_border_kernel<<<dim3(cdiv(2 * (H + W), 256), N), 256, 0, st>>>(H, W, L2);
    fill_output_kernel<<<gl, 256, 0, st>>>(m, 1, plane, L2, m2);
    label4_roots(m2, N, H, W, L2, st);
    cc_count_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L2, plane, nchunk, chunk);
    cc_scan_chunks_kernel<<<N, 256, 0, st>>>(nchunk, chunk, nullptr);
    cc_rank_roots_kernel<<<dim3(nchunk, N), 256, 0, st>>>(L2, plane, nchunk, chunk, aux);
    cc_relabel_kernel<<<gl, 256, 0, st>>>(L2, aux, plane, labels);
    if (hipMemsetAsync(hb, 0, P * 4, st) != hipSuccess) return check_launch("memset");
    ws_hist_kernel<<<gl, 256, 0, st>>>(labels, plane, hb);
    ws_drop_small_kernel<<<gl, 256, 0, st>>>(labels, hb, plane, min_size);
    return check_launch("cdnet_fill_label_process");
}
