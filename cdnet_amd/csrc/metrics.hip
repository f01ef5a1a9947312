// Instance-segmentation metrics (stats_utils.py: get_fast_aji :7-106, get_fast_pq :182-276, get_dice_1 :323-335,
// remap_label :361-392): the O(pixels) part - per-label areas and the sparse table of pairwise intersections between
// ground-truth and predicted instances - in one pass over the two label images; the O(pairs) arithmetic stays on the host
// in float64 with the reference's exact formulas (cdnet_amd/stats_utils.py).  Replaces the reference's
// O(instances_true x pixels) mask loops.
#include "common.h"

using namespace cdnet;

namespace {

// open-addressing table per image: key = (true_id << 16) | pred_id (ids < 65536), 0 = empty (id pairs with a zero id
// are never inserted)
__global__ __launch_bounds__(256) void pair_hist_kernel(const int32_t *__restrict__ t, const int32_t *__restrict__ p, int plane, int cap,
                                                        unsigned hmask, int *__restrict__ area_t, int *__restrict__ area_p,
                                                        unsigned *__restrict__ hkeys, int *__restrict__ hcnt, int *__restrict__ err) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * plane;
    int *at = area_t + (size_t)n * cap, *ap = area_p + (size_t)n * cap;
    unsigned *hk = hkeys + (size_t)n * (hmask + 1);
    int *hc = hcnt + (size_t)n * (hmask + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int a = t[base + i], b = p[base + i];
        if (a < 0 || b < 0 || a >= cap || b >= cap) { *err = 1; continue; }
        if (a > 0) atomicAdd(at + a, 1);
        if (b > 0) atomicAdd(ap + b, 1);
        if (a > 0 && b > 0) {
            const unsigned key = ((unsigned)a << 16) | (unsigned)b;
            unsigned h = (key * 2654435761u) & hmask;
            for (unsigned probe = 0; probe <= hmask; ++probe) {
                const unsigned cur = atomicCAS(hk + h, 0u, key);
                if (cur == 0u || cur == key) { atomicAdd(hc + h, 1); break; }
                h = (h + 1) & hmask;
                if (probe == hmask) *err = 2;          // table full
            }
        }
    }
}

// remap_label: ids -> 1..K in increasing id order (stats_utils.py:361-392, by_size = False)
__global__ __launch_bounds__(256) void label_present_kernel(const int32_t *__restrict__ lab, int plane, int cap, int *__restrict__ present,
                                                            int *__restrict__ err) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int a = lab[(size_t)n * plane + i];
        if (a < 0 || a >= cap) { *err = 1; continue; }
        if (a > 0) present[(size_t)n * cap + a] = 1;
    }
}

__global__ __launch_bounds__(256) void label_rank_kernel(int cap, int *__restrict__ present) {
    // one block per image: in-place exclusive scan + 1 for present ids (serial over 256-wide strips; cap is small)
    __shared__ int s[256];
    __shared__ int carry;
    int *pr = present + (size_t)blockIdx.x * cap;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < cap; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = (i < cap && i > 0) ? pr[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < cap) pr[i] = v ? carry + s[threadIdx.x] : 0;
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void label_apply_kernel(const int32_t *__restrict__ lab, int plane, int cap, const int *__restrict__ rank,
                                                          int32_t *__restrict__ out) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int a = lab[(size_t)n * plane + i];
        out[(size_t)n * plane + i] = (a > 0 && a < cap) ? rank[(size_t)n * cap + a] : 0;
    }
}

inline int glin(int plane) { int g = cdiv(plane, 256); return g > 1024 ? 1024 : (g < 1 ? 1 : g); }

}  // namespace

extern "C" int cdnet_label_pair_histogram(const int32_t *true_lab, const int32_t *pred_lab, int N, int plane, int cap, int hash_slots,
                                          int32_t *area_true, int32_t *area_pred, uint32_t *hash_keys, int32_t *hash_counts,
                                          int32_t *err_flag, void *stream) {
    CDNET_REQUIRE(true_lab && pred_lab && area_true && area_pred && hash_keys && hash_counts && err_flag, "cdnet_label_pair_histogram: null pointer");
    CDNET_REQUIRE(N > 0 && plane > 0 && cap > 1 && cap <= 65536, "cdnet_label_pair_histogram: label capacity %d not in (1, 65536]", cap);
    CDNET_REQUIRE(hash_slots >= 2 && (hash_slots & (hash_slots - 1)) == 0, "cdnet_label_pair_histogram: hash_slots must be a power of two");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(area_true, 0, (size_t)N * cap * 4, st) != hipSuccess || hipMemsetAsync(area_pred, 0, (size_t)N * cap * 4, st) != hipSuccess ||
        hipMemsetAsync(hash_keys, 0, (size_t)N * hash_slots * 4, st) != hipSuccess ||
        hipMemsetAsync(hash_counts, 0, (size_t)N * hash_slots * 4, st) != hipSuccess || hipMemsetAsync(err_flag, 0, 4, st) != hipSuccess)
        return check_launch("cdnet_label_pair_histogram(memset)");
    pair_hist_kernel<<<dim3(glin(plane), N), 256, 0, st>>>(true_lab, pred_lab, plane, cap, (unsigned)hash_slots - 1, area_true, area_pred,
                                                           hash_keys, hash_counts, err_flag);
    return check_launch("cdnet_label_pair_histogram");
}

extern "C" int cdnet_remap_label(const int32_t *lab, int N, int plane, int cap, int32_t *scratch, int32_t *out, int32_t *err_flag,
                                 void *stream) {
    CDNET_REQUIRE(lab && scratch && out && err_flag, "cdnet_remap_label: null pointer");
    CDNET_REQUIRE(N > 0 && plane > 0 && cap > 1, "cdnet_remap_label: bad size");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, (size_t)N * cap * 4, st) != hipSuccess || hipMemsetAsync(err_flag, 0, 4, st) != hipSuccess)
        return check_launch("cdnet_remap_label(memset)");
    label_present_kernel<<<dim3(glin(plane), N), 256, 0, st>>>(lab, plane, cap, scratch, err_flag);
    label_rank_kernel<<<N, 256, 0, st>>>(cap, scratch);
    label_apply_kernel<<<dim3(glin(plane), N), 256, 0, st>>>(lab, plane, cap, scratch, out);
    return check_launch("cdnet_remap_label");
}
