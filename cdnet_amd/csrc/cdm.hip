// Centripetal-direction-map / centre-point target generation on the GPU.
// Replaces the per-sample CPU transform `LabelEncoding.__call__` of the reference (my_transforms_direction.py:687-885,
// 3-class-PNG branch :763-781 + direction branch :785-871), its numba kernel `get_centerpoint2` (:650-685), the 11x11
// stencil of data_prepare/SegFix_offset_helper.py:97-132 and the 8-bin quantisation of DTOffsetHelper.align_angle
// (:311-341) - seconds per sample on the CPU, one pass of a few HBM-bound kernels here.
//
// The reference loops over nuclei and re-scans the whole image for each; here every pixel works for itself:
//   * its instance id (8-connected labelling of the eroded inside, grown back by the 4-neighbour cross),
//   * its centerness (8 rays x 30 bisection steps, double precision, python round-half-even),
//   * per instance: arg-max centerness with first-in-raster-order tie break, max distance to that centre,
//   * the stencil over the (once more dilated) nucleus of the LAST instance covering the pixel ("later instances overwrite").
// Integer outputs are bit-exact against the CPU oracle except where the float32 stencil sum lands on a 45-degree bin edge
// (the reference itself sums in torch's conv2d order there); tests allow <= 1e-3 of the pixels to differ.
#include "common.h"

using namespace cdnet;

namespace {

struct Rays { double s[8], c[8]; };      // sin/cos(2 pi k / 8) evaluated by the host libm, exactly as the reference's math.sin/cos

// float64 -> float16 with ONE rounding (numpy's astype(float16) from float64): float32 by round-to-odd, then RNE to half
__device__ __forceinline__ unsigned short d2h_bits(double a) {
    float f = __double2float_rz(a);
    unsigned u = __float_as_uint(f);
    if ((double)f != a) u |= 1u;
    return __builtin_bit_cast(unsigned short, (_Float16)__uint_as_float(u));
}

// 1. inside / boundary / 3-class label (:765-769, :781); m1 = eroded inside (new_label == 1)
__global__ __launch_bounds__(256) void cdm_prep_kernel(const uint8_t *__restrict__ in, int H, int W, uint8_t *__restrict__ label3,
                                                       uint8_t *__restrict__ m1) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const uint8_t *s = in + (size_t)n * H * W;
    auto ins = [&](int yy, int xx) { return s[(size_t)yy * W + xx] > 127; };
    const bool c = ins(y, x);
    bool dil = c, ero = c;
    if (y > 0) { bool v = ins(y - 1, x); dil |= v; ero &= v; }
    if (y < H - 1) { bool v = ins(y + 1, x); dil |= v; ero &= v; }
    if (x > 0) { bool v = ins(y, x - 1); dil |= v; ero &= v; }
    if (x < W - 1) { bool v = ins(y, x + 1); dil |= v; ero &= v; }
    const int nl = (dil && !ero) ? 2 : (c ? 1 : 0);
    const size_t o = (size_t)n * H * W + (size_t)y * W + x;
    label3[o] = nl == 2 ? 255 : (nl == 1 ? 127 : 0);
    m1[o] = nl == 1;
}

// 2. label_instance = dilation(measure.label(m1), disk(1)) (:773-774)
__global__ __launch_bounds__(256) void cdm_grow_kernel(const int32_t *__restrict__ lab, int H, int W, int32_t *__restrict__ inst, int maxid) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int32_t *s = lab + (size_t)n * H * W;
    int v = s[(size_t)y * W + x];
    if (y > 0) v = max(v, s[(size_t)(y - 1) * W + x]);
    if (y < H - 1) v = max(v, s[(size_t)(y + 1) * W + x]);
    if (x > 0) v = max(v, s[(size_t)y * W + x - 1]);
    if (x < W - 1) v = max(v, s[(size_t)y * W + x + 1]);
    inst[(size_t)n * H * W + (size_t)y * W + x] = v < maxid ? v : 0;      // ids beyond max_instances are dropped (the per-id tables end there)
}

// instance-label input (:752-760): inside = label > 0 (all of it dropped when the image has fewer than 5 foreground pixels: the
// reference's remove_small_objects(new_label, 5) sees ONE label), boundary where the cross neighbourhood's max and min ids differ
__global__ __launch_bounds__(256) void cdm_count_fg_kernel(const int32_t *__restrict__ in, int plane, int *__restrict__ fg) {
    const int n = blockIdx.y;
    int c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) c += in[(size_t)n * plane + i] > 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&fg[n], c);
}

__global__ __launch_bounds__(256) void cdm_prep_inst_kernel(const int32_t *__restrict__ in, const int *__restrict__ fg, int H, int W,
                                                            uint8_t *__restrict__ label3, uint8_t *__restrict__ m1, uint8_t *__restrict__ inside) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int32_t *s = in + (size_t)n * H * W;
    const int c = s[(size_t)y * W + x];
    int mx = c, mn = c;
    if (y > 0) { const int v = s[(size_t)(y - 1) * W + x]; mx = max(mx, v); mn = min(mn, v); }
    if (y < H - 1) { const int v = s[(size_t)(y + 1) * W + x]; mx = max(mx, v); mn = min(mn, v); }
    if (x > 0) { const int v = s[(size_t)y * W + x - 1]; mx = max(mx, v); mn = min(mn, v); }
    if (x < W - 1) { const int v = s[(size_t)y * W + x + 1]; mx = max(mx, v); mn = min(mn, v); }
    const bool in1 = c > 0 && fg[n] >= 5;
    const int nl = mx != mn ? 2 : (in1 ? 1 : 0);
    const size_t o = (size_t)n * H * W + (size_t)y * W + x;
    label3[o] = nl == 2 ? 255 : (nl == 1 ? 127 : 0);
    m1[o] = nl == 1 ? 255 : 0;
    inside[o] = in1 ? 255 : 0;
}

// per-instance maximum of non-negative doubles (their bit patterns order like the values): the lanes of a wave that hold the same instance
// id combine first, ONE atomic per (wave, id) - a nucleus spans a few hundred pixels and every one of them used to hit the same address
__device__ __forceinline__ void wave_max_to(unsigned long long *__restrict__ table, int k, unsigned long long v, bool active) {
    const int lane = (int)(threadIdx.x + threadIdx.y * blockDim.x) & 63;      // (blocks are (64, 4) or (256, 1): a wave is 64 consecutive threads)
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lk = __shfl(k, leader);
        const bool mine = active && k == lk;
        unsigned long long m = mine ? v : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = ((unsigned long long)(unsigned)__shfl_xor((int)(m >> 32), o) << 32) | (unsigned)__shfl_xor((int)(m & 0xffffffffu), o);
            m = other > m ? other : m;
        }
        if (lane == leader) atomicMax(&table[lk], m);
        todo &= ~__ballot(mine);
        active = active && !mine;
    }
}

// 3. get_centerpoint2 (:650-685): centerness of every instance pixel; per-instance maximum (double bits are monotone)
// 3. get_centerpoint2 (:650-685): centerness of every instance pixel; per-instance maximum (double bits are monotone).  A block takes 256
// consecutive pixels and compacts the ones that belong to an instance into its first waves (ballot + prefix counts through LDS): the long
// bisection loop runs on full waves instead of the ~30 % of lanes a row of pixels has inside nuclei, the other waves leave at once
__global__ __launch_bounds__(256) void cdm_centerness_kernel(const int32_t *__restrict__ inst, int H, int W, Rays R,
                                                             double *__restrict__ cness, unsigned long long *__restrict__ best, int maxid) {
    __shared__ int s_cnt[4], s_pix[256];
    const int n = blockIdx.y;
    const int plane = H * W;
    const int32_t *s = inst + (size_t)n * plane;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool member = i < plane && s[i] > 0;
    const unsigned long long m = __ballot(member);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += s_cnt[k];
    if (member) s_pix[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    if ((int)(threadIdx.x & ~63u) >= total) return;               // (whole waves beyond the block's list leave)
    const bool on = (int)threadIdx.x < total;
    const int p = on ? s_pix[threadIdx.x] : 0;
    const int y = p / W, x = p - y * W;
    const int id = on ? s[p] : 0;
    double c = 0;
    if (on) {
        // the eight rays' bisections step together: each ray's 30 steps are its own dependent chain (a gather per step), eight of them in flight
        double lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo[k] = 0; hi[k] = 1000; }
        for (int t = 0; t < 30; ++t) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double mid = (lo[k] + hi[k]) / 2;
                const double fy = y + R.s[k] * mid, fx = x + R.c[k] * mid;        // -ffp-contract=off: mul then add, like CPython
                const int ny = (int)rint(fy), nx = (int)rint(fx);                 // (|.| <= 1000 + the image size: exact in 32 bits)
                const bool inb = ny >= 0 && ny < H && nx >= 0 && nx < W;
                const int v = s[inb ? ny * W + nx : p];                           // (an unconditional load: the eight gathers of a step issue together)
                const bool in = inb && v == id;
                lo[k] = in ? mid : lo[k];
                hi[k] = in ? hi[k] : mid;
            }
        }
        double ma = 0, mi = 10000000;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ma = hi[k] > ma ? hi[k] : ma;
            mi = hi[k] < mi ? hi[k] : mi;
        }
        c = mi / ma;
        cness[(size_t)n * plane + p] = c;
    }
    wave_max_to(best + (size_t)n * maxid, id, (unsigned long long)__double_as_longlong(c), on);
}

// first pixel in raster order that attains the maximum (`if centerness > now`, :680)
__global__ __launch_bounds__(256) void cdm_argmax_kernel(const int32_t *__restrict__ inst, const double *__restrict__ cness,
                                                         const unsigned long long *__restrict__ best, int plane, int maxid,
                                                         int *__restrict__ center) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int id = inst[(size_t)n * plane + i];
        if (id <= 0) continue;
        if ((unsigned long long)__double_as_longlong(cness[(size_t)n * plane + i]) == best[(size_t)n * maxid + id])
            atomicMin(&center[(size_t)n * maxid + id], i);
    }
}

// candidate instances whose dilated nucleus (dilation(nucleus, disk(1)), :819) contains pixel (y,x): own id + 4 neighbours
__device__ __forceinline__ bool in_dilated(const int32_t *s, int H, int W, int y, int x, int k) {
    if (s[(size_t)y * W + x] == k) return true;
    if (y > 0 && s[(size_t)(y - 1) * W + x] == k) return true;
    if (y < H - 1 && s[(size_t)(y + 1) * W + x] == k) return true;
    if (x > 0 && s[(size_t)y * W + x - 1] == k) return true;
    if (x < W - 1 && s[(size_t)y * W + x + 1] == k) return true;
    return false;
}

// 4. int_pos.max() per instance (:822-824)
__global__ __launch_bounds__(256) void cdm_dmax_kernel(const int32_t *__restrict__ inst, int H, int W, const int *__restrict__ center,
                                                       int maxid, unsigned long long *__restrict__ dmax) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const bool inb = x < W && y < H;
    const int32_t *s = inst + (size_t)n * H * W;
    int ids[5] = {0, 0, 0, 0, 0};
    if (inb) {
        ids[0] = s[(size_t)y * W + x];
        ids[1] = y > 0 ? s[(size_t)(y - 1) * W + x] : 0;
        ids[2] = y < H - 1 ? s[(size_t)(y + 1) * W + x] : 0;
        ids[3] = x > 0 ? s[(size_t)y * W + x - 1] : 0;
        ids[4] = x < W - 1 ? s[(size_t)y * W + x + 1] : 0;
    }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int k = ids[a];
        bool use = k > 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) use = use && !(b < a && ids[b] == k);
        double d = 0;
        if (use) {
            const int c = center[(size_t)n * maxid + k];
            const int cy = c / W, cx = c % W;
            d = sqrt((double)(y - cy) * (y - cy) + (double)(x - cx) * (x - cx));
        }
        wave_max_to(dmax + (size_t)n * maxid, k, (unsigned long long)__double_as_longlong(d), use);
    }
}

// 4b. per pixel q: K = the LAST instance whose dilated nucleus covers q (the largest id in its cross neighbourhood) and the normalised
// distance value the stencil reads there, f(q; K) = (float)(1 - d(q, centre_K) / (dmax_K + 1e-7)) - computed ONCE per pixel instead of once
// per (pixel, tap): the 11x11 stencil of a pixel p reads f(q; K_p), and K_q = K_p for every tap except where nuclei touch
__global__ __launch_bounds__(256) void cdm_field_kernel(const int32_t *__restrict__ inst, int H, int W, const int *__restrict__ center,
                                                        const unsigned long long *__restrict__ dmax, int maxid, int32_t *__restrict__ Kp,
                                                        float *__restrict__ F0) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int32_t *s = inst + (size_t)n * H * W;
    int k = s[(size_t)y * W + x];
    if (y > 0) k = max(k, s[(size_t)(y - 1) * W + x]);
    if (y < H - 1) k = max(k, s[(size_t)(y + 1) * W + x]);
    if (x > 0) k = max(k, s[(size_t)y * W + x - 1]);
    if (x < W - 1) k = max(k, s[(size_t)y * W + x + 1]);
    float f = 0.f;
    if (k > 0) {
        const int c = center[(size_t)n * maxid + k];
        const int cy = c / W, cx = c % W;
        const double dm = __longlong_as_double((long long)dmax[(size_t)n * maxid + k]) + 0.0000001;
        const double d = sqrt((double)(y - cy) * (y - cy) + (double)(x - cx) * (x - cx));
        f = (float)((1 - d / dm) * 1.0);
    }
    const size_t o = (size_t)n * H * W + (size_t)y * W + x;
    Kp[o] = k;
    F0[o] = f;
}

// 5. the 11x11 stencil on (1 - d/(dmax+1e-7)) * nucleus of the last instance covering the pixel, then the angle bin
__global__ __launch_bounds__(256) void cdm_direction_kernel(const uint8_t *__restrict__ in, const int32_t *__restrict__ inst,
                                                            const int32_t *__restrict__ Kp, const float *__restrict__ F0, int H, int W,
                                                            const int *__restrict__ center, const unsigned long long *__restrict__ dmax,
                                                            int maxid, uint8_t *__restrict__ direction) {
    // the block's 4 x 64 pixels and their 5-pixel apron of (K, f) in LDS: a tap is two LDS reads and one compare
    constexpr int WR = 4 + 10, WC = 64 + 10;
    __shared__ int32_t sK[WR][WC + 1];
    __shared__ float sF[WR][WC + 1];
    __shared__ float sWy[11][11], sWx[11][11];                   // the stencil's float32 taps (float)(j / (i*i + j*j)), (float)(i / ...), once per block
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const size_t img = (size_t)n * H * W;
    if (tid < 121) {
        const int j = tid / 11 - 5, i = tid % 11 - 5;
        const double den = (double)(i * i + j * j);
        sWy[j + 5][i + 5] = (i == 0 && j == 0) ? 0.f : (float)(j / den);      // Sobel.kernel: float32 taps (:112-113)
        sWx[j + 5][i + 5] = (i == 0 && j == 0) ? 0.f : (float)(i / den);
    }
    for (int idx = tid; idx < WR * WC; idx += 256) {
        const int r = idx / WC, c = idx - r * WC;
        const int yy = y0 - 5 + r, xx = x0 - 5 + c;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        sK[r][c] = ok ? Kp[img + (size_t)yy * W + xx] : -1;      // (-1: below every instance id - a tap outside the image is no member)
        sF[r][c] = ok ? F0[img + (size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t o = img + (size_t)y * W + x;
    if (!(in[o] > 127)) { direction[o] = 0; return; }                          // new_label_inside == 0 -> background (:855-865)
    const int32_t *s = inst + img;
    const int k = sK[threadIdx.y + 5][threadIdx.x + 5];
    float gy = 0.f, gx = 0.f;
    if (k > 0) {
        const int c = center[(size_t)n * maxid + k];
        const int cy = c / W, cx = c % W;
        const double dm = __longlong_as_double((long long)dmax[(size_t)n * maxid + k]) + 0.0000001;
        double sy = 0, sx = 0;
        for (int j = -5; j <= 5; ++j) {
            const int yy = y + j;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int i = -5; i <= 5; ++i) {
                const int xx = x + i;
                if (xx < 0 || xx >= W || (i == 0 && j == 0)) continue;
                const int kq = sK[threadIdx.y + 5 + j][threadIdx.x + 5 + i];
                if (kq < k) continue;                               // every id around the tap is smaller: not in the dilated nucleus of k
                float f;
                if (kq == k) f = sF[threadIdx.y + 5 + j][threadIdx.x + 5 + i];
                else {
                    // a later instance also covers the tap (touching nuclei): membership and value for k itself
                    if (!in_dilated(s, H, W, yy, xx, k)) continue;
                    const double d = sqrt((double)(yy - cy) * (yy - cy) + (double)(xx - cx) * (xx - cx));
                    f = (float)((1 - d / dm) * 1.0);
                }
                sy += (double)sWy[j + 5][i + 5] * f;
                sx += (double)sWx[j + 5][i + 5] * f;
            }
        }
        gy = (float)sy; gx = (float)sx;
    }
    const float ang = atan2f(gy, gx) * (180.0f / 3.14159265358979323846f);      // np.degrees(np.arctan2(.)) on float32 (:848)
    int bin = 0;
    if (!(ang <= -157.5f || ang > 157.5f)) {
#pragma unroll
        for (int b = 1; b < 8; ++b) {
            const float mid = -180.f + 45.f * b;
            if (ang > mid - 22.5f && ang <= mid + 22.5f) bin = b;
        }
    }
    direction[o] = (uint8_t)(bin + 1);
}

// 6. gaussian_filter(label_point, sigma=2).astype(float16) (:842): separable, radius 8, reflect, float64, scipy's symmetric pairing
struct GaussK { double k[9]; };

__device__ __forceinline__ int reflect(int i, int n) {
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

__global__ void cdm_scatter_centers_kernel(const int *__restrict__ center, const int32_t *__restrict__ counts, int maxid, int plane,
                                           double *__restrict__ lp) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (k > counts[n] || k >= maxid) return;
    const int c = center[(size_t)n * maxid + k];
    if (c >= 0 && c < plane) lp[(size_t)n * plane + c] = 255.0;
}

template <int AXIS>
__global__ __launch_bounds__(256) void cdm_gauss_kernel(const double *__restrict__ src, int H, int W, GaussK G, double *__restrict__ dstd,
                                                        unsigned short *__restrict__ dsth) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const double *s = src + (size_t)n * H * W;
    double a = s[(size_t)y * W + x] * G.k[0];
#pragma unroll
    for (int i = 1; i <= 8; ++i) {
        double lo, hi;
        if (AXIS == 0) { lo = s[(size_t)reflect(y - i, H) * W + x]; hi = s[(size_t)reflect(y + i, H) * W + x]; }
        else { lo = s[(size_t)y * W + reflect(x - i, W)]; hi = s[(size_t)y * W + reflect(x + i, W)]; }
        a += (lo + hi) * G.k[i];
    }
    const size_t o = (size_t)n * H * W + (size_t)y * W + x;
    if (AXIS == 0) dstd[o] = a;
    else dsth[o] = d2h_bits(a);
}

__global__ void cdm_fill_counts_kernel(int32_t *counts, int N, int v) {
    if ((int)threadIdx.x < N) counts[threadIdx.x] = v;
}

__global__ void cdm_init_kernel(unsigned long long *best, unsigned long long *dmax, int *center, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { best[i] = 0ull; dmax[i] = 0ull; center[i] = 0x7fffffff; }
}

inline dim3 grid_rows(int N, int H, int W) { return dim3(cdiv(W, 64), cdiv(H, 4), N); }

}  // namespace

// the per-instance stage shared by both input kinds: centre search, distance normalisation, 11x11 stencil + angle bins, point map.
// `inside_u8` (> 127 = new_label_inside) masks the classes; `counts[n]` = largest instance id of image n
static int cdm_direction_stage(const uint8_t *label_ch0, const int32_t *inst, const int32_t *counts, int N, int H, int W, int maxid, const Rays &R,
                               const GaussK &G, double *cness, double *tmp, unsigned long long *best, unsigned long long *dmax, int *center,
                               uint8_t *direction, uint16_t *point_f16, int32_t *inst_out, int32_t *counts_out, hipStream_t st) {
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const int plane = H * W;
    const size_t nk = (size_t)N * maxid;
    cdm_init_kernel<<<(unsigned)((nk + 255) / 256), 256, 0, st>>>(best, dmax, center, nk);
    cdm_centerness_kernel<<<dim3(cdiv(plane, 256), N), 256, 0, st>>>(inst, H, W, R, cness, best, maxid);
    int g = cdiv(plane, 256); if (g > 1024) g = 1024;
    cdm_argmax_kernel<<<dim3(g, N), 256, 0, st>>>(inst, cness, best, plane, maxid, center);
    cdm_dmax_kernel<<<gr, br, 0, st>>>(inst, H, W, center, maxid, dmax);
    // (K | f planes: 4 + 4 bytes per pixel in the float64 scratch plane `tmp`, which the Gaussian below only needs afterwards)
    int32_t *Kp = reinterpret_cast<int32_t *>(tmp);
    float *F0 = reinterpret_cast<float *>(tmp) + (size_t)N * plane;
    cdm_field_kernel<<<gr, br, 0, st>>>(inst, H, W, center, dmax, maxid, Kp, F0);
    cdm_direction_kernel<<<gr, br, 0, st>>>(label_ch0, inst, Kp, F0, H, W, center, dmax, maxid, direction);
    // point map: impulses of 255 at the centres, separable Gaussian in float64
    if (hipMemsetAsync(cness, 0, (size_t)N * plane * 8, st) != hipSuccess) return check_launch("memset lp");
    cdm_scatter_centers_kernel<<<dim3(cdiv(maxid, 256), N), 256, 0, st>>>(center, counts, maxid, plane, cness);
    cdm_gauss_kernel<0><<<gr, br, 0, st>>>(cness, H, W, G, tmp, nullptr);
    cdm_gauss_kernel<1><<<gr, br, 0, st>>>(tmp, H, W, G, nullptr, point_f16);
    if (inst_out && hipMemcpyAsync(inst_out, inst, (size_t)N * plane * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return check_launch("copy inst");
    if (counts_out && hipMemcpyAsync(counts_out, counts, (size_t)N * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return check_launch("copy counts");
    return check_launch("cdnet_label_encoding");
}

// workspace layout
static size_t cdm_layout(int N, int H, int W, int maxid, size_t *o) {
    const size_t P = (size_t)N * H * W;
    size_t off = 0;
    o[0] = off; off = align_up(off + P, 256);                                   // m1 u8
    o[1] = off; off = align_up(off + P * 4, 256);                               // L
    o[2] = off; off = align_up(off + P * 4, 256);                               // aux
    o[3] = off; off = align_up(off + (size_t)N * cdiv(H * W, 1024) * 4, 256);   // chunk
    o[4] = off; off = align_up(off + P * 4, 256);                               // lab
    o[5] = off; off = align_up(off + P * 4, 256);                               // inst
    o[6] = off; off = align_up(off + P * 8, 256);                               // cness / lp (double)
    o[7] = off; off = align_up(off + P * 8, 256);                               // tmp (double)
    o[8] = off; off = align_up(off + (size_t)N * maxid * 8, 256);               // best
    o[9] = off; off = align_up(off + (size_t)N * maxid * 8, 256);               // dmax
    o[10] = off; off = align_up(off + (size_t)N * maxid * 4, 256);              // center
    o[11] = off; off = align_up(off + (size_t)N * 4, 256);                      // counts
    return off;
}

extern "C" size_t cdnet_label_encoding_workspace_bytes(int N, int H, int W, int max_instances) {
    if (N <= 0 || H <= 0 || W <= 0 || max_instances <= 0) return 0;
    size_t o[12];
    return cdm_layout(N, H, W, max_instances + 1, o);
}

extern "C" int cdnet_label_encoding(const uint8_t *label_ch0, int N, int H, int W, int max_instances, const double *rays_host,
                                    const double *gauss_host, void *workspace, size_t workspace_bytes, uint8_t *label3,
                                    uint16_t *point_f16, uint8_t *direction, int32_t *inst_out, int32_t *counts_out, void *stream) {
    CDNET_REQUIRE(label_ch0 && rays_host && gauss_host && workspace && label3 && point_f16 && direction, "cdnet_label_encoding: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0 && max_instances > 0, "cdnet_label_encoding: bad size");
    const int maxid = max_instances + 1;
    size_t o[12];
    const size_t need = cdm_layout(N, H, W, maxid, o);
    if (workspace_bytes < need) { set_error("cdnet_label_encoding: workspace %zu < %zu bytes", workspace_bytes, need); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint8_t *m1 = (uint8_t *)(ws + o[0]);
    int *L = (int *)(ws + o[1]), *aux = (int *)(ws + o[2]), *chunk = (int *)(ws + o[3]);
    int32_t *lab = (int32_t *)(ws + o[4]), *inst = (int32_t *)(ws + o[5]);
    double *cness = (double *)(ws + o[6]), *tmp = (double *)(ws + o[7]);
    unsigned long long *best = (unsigned long long *)(ws + o[8]), *dmax = (unsigned long long *)(ws + o[9]);
    int *center = (int *)(ws + o[10]);
    int32_t *counts = (int32_t *)(ws + o[11]);
    Rays R;
    for (int k = 0; k < 8; ++k) { R.s[k] = rays_host[2 * k]; R.c[k] = rays_host[2 * k + 1]; }
    GaussK G;
    for (int i = 0; i < 9; ++i) G.k[i] = gauss_host[i];
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    cdm_prep_kernel<<<gr, br, 0, st>>>(label_ch0, H, W, label3, m1);
    int rc = label8_raster(m1, N, H, W, L, aux, chunk, lab, counts, st);
    if (rc) return rc;
    cdm_grow_kernel<<<gr, br, 0, st>>>(lab, H, W, inst, maxid);
    return cdm_direction_stage(label_ch0, inst, counts, N, H, W, maxid, R, G, cness, tmp, best, dmax, center, direction, point_f16, inst_out, counts_out, st);
}


// Instance-label input of LabelEncoding (my_transforms_direction.py:752-760, `label_level_len > 2`): labels i32 [N][H][W] hold
// instance ids.  Boundary from the ids' cross max / min, instances through the watershed variant of postproc_other.process
// (min_size 5), grown by disk(1); then the common stage.  Workspace: cdnet_label_encoding_instances_workspace_bytes.
extern "C" size_t cdnet_label_encoding_instances_workspace_bytes(int N, int H, int W, int max_instances) {
    if (N <= 0 || H <= 0 || W <= 0 || max_instances <= 0) return 0;
    size_t o[12];
    return align_up(cdm_layout(N, H, W, max_instances + 1, o), 256) + align_up(cdnet_watershed_workspace_bytes(N, H, W), 256) +
           align_up((size_t)N * H * W, 256) + 256;
}

extern "C" int cdnet_label_encoding_instances(const int32_t *label_inst, int N, int H, int W, int max_instances, const double *rays_host,
                                              const double *gauss_host, void *workspace, size_t workspace_bytes, uint8_t *label3,
                                              uint16_t *point_f16, uint8_t *direction, int32_t *inst_out, int32_t *counts_out, void *stream) {
    CDNET_REQUIRE(label_inst && rays_host && gauss_host && workspace && label3 && point_f16 && direction, "cdnet_label_encoding_instances: null pointer");
    CDNET_REQUIRE(N > 0 && H > 0 && W > 0 && max_instances > 0, "cdnet_label_encoding_instances: bad size");
    const int maxid = max_instances + 1;
    size_t o[12];
    const size_t base = align_up(cdm_layout(N, H, W, maxid, o), 256), wsb = align_up(cdnet_watershed_workspace_bytes(N, H, W), 256);
    const size_t need = cdnet_label_encoding_instances_workspace_bytes(N, H, W, max_instances);
    if (workspace_bytes < need) { set_error("cdnet_label_encoding_instances: workspace %zu < %zu bytes", workspace_bytes, need); return CDNET_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint8_t *m1 = (uint8_t *)(ws + o[0]);
    int32_t *lab = (int32_t *)(ws + o[4]), *inst = (int32_t *)(ws + o[5]);
    double *cness = (double *)(ws + o[6]), *tmp = (double *)(ws + o[7]);
    unsigned long long *best = (unsigned long long *)(ws + o[8]), *dmax = (unsigned long long *)(ws + o[9]);
    int *center = (int *)(ws + o[10]);
    int32_t *counts = (int32_t *)(ws + o[11]);
    void *wsw = ws + base;
    uint8_t *inside = (uint8_t *)(ws + base + wsb);
    int *fg = (int *)(ws + base + wsb + align_up((size_t)N * H * W, 256));
    CDNET_REQUIRE(N <= 64, "cdnet_label_encoding_instances: at most 64 images per call");
    Rays R;
    for (int k = 0; k < 8; ++k) { R.s[k] = rays_host[2 * k]; R.c[k] = rays_host[2 * k + 1]; }
    GaussK G;
    for (int i = 0; i < 9; ++i) G.k[i] = gauss_host[i];
    const dim3 gr = grid_rows(N, H, W), br(64, 4);
    const int plane = H * W;
    if (hipMemsetAsync(fg, 0, 256, st) != hipSuccess) return check_launch("memset fg");
    int g = cdiv(plane, 256); if (g > 256) g = 256;
    cdm_count_fg_kernel<<<dim3(g, N), 256, 0, st>>>(label_inst, plane, fg);
    cdm_prep_inst_kernel<<<gr, br, 0, st>>>(label_inst, fg, H, W, label3, m1, inside);
    int rc = cdnet_watershed_process(m1, N, H, W, 5, wsw, wsb, nullptr, nullptr, lab, st);
    if (rc) return rc;
    cdm_grow_kernel<<<gr, br, 0, st>>>(lab, H, W, inst, maxid);
    // every id below maxid may exist (watershed marker ids are kept, not renumbered)
    cdm_fill_counts_kernel<<<1, 64, 0, st>>>(counts, N, maxid - 1);
    return cdm_direction_stage(inside, inst, counts, N, H, W, maxid, R, G, cness, tmp, best, dmax, center, direction, point_f16, inst_out, counts_out, st);
}
