/*
 * ORACLE (test infrastructure, NOT product code): plain-C, single-threaded restatement of the reference's
 * CPU post-processing path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Parity is pinned against tests/golden/{ddm,postproc,probmaps}.npz, which were produced by running the
 * reference itself (tests/golden/make_golden.py).  The skimage calls of the reference were evaluated there
 * through scipy stand-ins ("skimage-semantics restated").
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---------------------------------------------------------------------------------------------------
 * generate_dd_map  (data_prepare/getDirectionDiffMap.py:44-108, circshift :14-42,
 *                   DTOffsetHelper.label_to_vector SegFix_offset_helper.py:246-261)
 * lut[a*classes+b] = round(cos(v_a, v_b)) as the reference computes it in float64 -> {-1,0,1}; built by the
 * caller (oracle/postproc.py: ddm_lut) from label_to_vector_mapping.  `extra_zero` = the reference's
 * never-written cos channels (17 classes: 16 channels, 8 filled -> min() always sees a 0).
 * code = 1 - min over neighbours (neighbour outside the image = zero vector = class 0 -> cos 0);
 * background pixels get cos 1 -> code 0.  out = (code-min)/(max-min) in float32 (NaN when constant).
 * nbr: 8 (9 or 17 classes) or 4 (5 classes).
 * ------------------------------------------------------------------------------------------------- */
void orc_ddm(const uint8_t *lab, int H, int W, int classes, const int8_t *lut, int nbr, int extra_zero,
             uint8_t *code, float *out)
{
    static const int dy8[8] = {1, 1, 1, 0, 0, -1, -1, -1};   /* feature1..9 without 5: (i+1,j+1),(i+1,j),(i+1,j-1),(i,j+1),(i,j-1),(i-1,j+1),(i-1,j),(i-1,j-1) */
    static const int dx8[8] = {1, 0, -1, 1, -1, 1, 0, -1};
    static const int dy4[4] = {1, 0, 0, -1};
    static const int dx4[4] = {0, 1, -1, 0};
    const int *dy = nbr == 8 ? dy8 : dy4, *dx = nbr == 8 ? dx8 : dx4;
    int cmin = 255, cmax = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int c = lab[y * W + x];
            int v = 0;
            if (c != 0) {
                int m = extra_zero ? 0 : 2;
                for (int k = 0; k < nbr; ++k) {
                    int yy = y + dy[k], xx = x + dx[k];
                    int q = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : lab[yy * W + xx];
                    int r = lut[c * classes + q];
                    if (r < m) m = r;
                }
                v = 1 - m;
            }
            code[y * W + x] = (uint8_t)v;
            if (v < cmin) cmin = v;
            if (v > cmax) cmax = v;
        }
    if (out) {
        float den = (float)(cmax - cmin);
        for (int i = 0; i < H * W; ++i) out[i] = ((float)code[i] - (float)cmin) / den;
    }
}

/* ---------------------------------------------------------------------------------------------------
 * morphology.dilation(binary u8, disk(1)) = max over the 4-neighbour cross (test_dam.py:531)
 * ------------------------------------------------------------------------------------------------- */
void orc_dilate_cross_u8(const uint8_t *in, int H, int W, uint8_t *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = in[y * W + x];
            if (y > 0 && in[(y - 1) * W + x] > v) v = in[(y - 1) * W + x];
            if (y < H - 1 && in[(y + 1) * W + x] > v) v = in[(y + 1) * W + x];
            if (x > 0 && in[y * W + x - 1] > v) v = in[y * W + x - 1];
            if (x < W - 1 && in[y * W + x + 1] > v) v = in[y * W + x + 1];
            out[y * W + x] = v;
        }
}

/* ---------------------------------------------------------------------------------------------------
 * TTA mean + DDM fuse + point-guided boost + argmax   (test_dam.py:445-450, 479-491, 529-539)
 * probs  f32 [V][3][H][W], points f32 [V][H][W], ddm f32 [V][H][W] (per-view normalised maps)
 * prob_mean/point_mean: f32 sequential sums in view order, then /V  (numpy float32 arithmetic)
 * ddm_mean: float64 mean (np.mean over a float64 array: pairwise == sequential for 8 exact k/2 values)
 * inside3 = dilate_cross(point_mean/max(point_mean) > 0.2f)
 * eb = 2*(ddm_mean - ddm_mean*inside3) (f64); P2 = f32((f64(P2) + 0.5*eb)*(1+eb)); pred = argmax (first max)
 * ------------------------------------------------------------------------------------------------- */
void orc_fuse_boost_argmax(const float *probs, const float *points, const float *ddm, int V, int H, int W,
                           float *prob_mean, float *point_mean, double *ddm_mean, uint8_t *inside3,
                           uint8_t *pred)
{
    int n = H * W;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < n; ++i) {
            float s = probs[(0 * 3 + c) * n + i];
            for (int v = 1; v < V; ++v) s = s + probs[(v * 3 + c) * n + i];
            prob_mean[c * n + i] = s / (float)V;
        }
    float pmax = -INFINITY;
    for (int i = 0; i < n; ++i) {
        float s = points[i];
        for (int v = 1; v < V; ++v) s = s + points[v * n + i];
        point_mean[i] = s / (float)V;
        if (point_mean[i] > pmax) pmax = point_mean[i];
    }
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int v = 0; v < V; ++v) s += (double)ddm[v * n + i];
        ddm_mean[i] = s / V;
    }
    uint8_t *t = (uint8_t *)malloc(n);
    for (int i = 0; i < n; ++i) t[i] = (point_mean[i] / pmax > 0.2f) ? 1 : 0;
    orc_dilate_cross_u8(t, H, W, inside3);
    free(t);
    for (int i = 0; i < n; ++i) {
        double d = ddm_mean[i];
        double eb = 2.0 * (d - d * (double)inside3[i]);
        float p2 = (float)(((double)prob_mean[2 * n + i] + 0.5 * eb) * (1.0 + eb));
        float p0 = prob_mean[i], p1 = prob_mean[n + i];
        int a = 0;
        float m = p0;
        /* np.argmax: first maximal element; NaN counts as maximal */
        if (p1 > m || (p1 != p1 && m == m)) { a = 1; m = p1; }
        if (p2 > m || (p2 != p2 && m == m)) { a = 2; m = p2; }
        pred[i] = (uint8_t)a;
    }
}

/* ---------------------------------------------------------------------------------------------------
 * flood-fill labelling helper (BFS, raster scan => ids in raster order of each component's first pixel)
 * conn8 != 0: full connectivity (skimage.measure.label default); else 4-connectivity.
 * Labels foreground of `mask` (non-zero).  Returns the component count.  area (optional): [count+1].
 * ------------------------------------------------------------------------------------------------- */
static int label_bfs(const uint8_t *mask, int H, int W, int conn8, int32_t *lab, int32_t *area)
{
    int n = H * W, cnt = 0;
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memset(lab, 0, sizeof(int32_t) * (size_t)n);
    for (int s = 0; s < n; ++s) {
        if (!mask[s] || lab[s]) continue;
        ++cnt;
        int head = 0, tail = 0, a = 0;
        q[tail++] = s;
        lab[s] = cnt;
        while (head < tail) {
            int p = q[head++];
            ++a;
            int y = p / W, x = p % W;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dy && !dx) continue;
                    if (!conn8 && dy && dx) continue;
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int r = yy * W + xx;
                    if (mask[r] && !lab[r]) { lab[r] = cnt; q[tail++] = r; }
                }
        }
        if (area) area[cnt] = a;
    }
    free(q);
    return cnt;
}

/* scipy.ndimage.binary_fill_holes (test_dam.py:546): background 4-connected to the image border stays
 * background, every other background pixel becomes foreground. */
void orc_fill_holes(const uint8_t *in, int H, int W, uint8_t *out)
{
    int n = H * W;
    uint8_t *bg = (uint8_t *)malloc(n);
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; ++i) bg[i] = in[i] ? 0 : 1;
    int cnt = label_bfs(bg, H, W, 0, lab, NULL);
    uint8_t *outer = (uint8_t *)calloc((size_t)cnt + 1, 1);
    for (int x = 0; x < W; ++x) { outer[lab[x]] = 1; outer[lab[(H - 1) * W + x]] = 1; }
    for (int y = 0; y < H; ++y) { outer[lab[y * W]] = 1; outer[lab[y * W + W - 1]] = 1; }
    for (int i = 0; i < n; ++i) out[i] = (in[i] || !outer[lab[i]]) ? 1 : 0;
    free(outer); free(lab); free(bg);
}

/* skimage.morphology.remove_small_objects(bool, min_size) (test_dam.py:548): connectivity=1 components with
 * fewer than min_size pixels are removed. */
void orc_remove_small(const uint8_t *in, int H, int W, int min_size, uint8_t *out)
{
    int n = H * W;
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *area = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    label_bfs(in, H, W, 0, lab, area);
    for (int i = 0; i < n; ++i) out[i] = (in[i] && area[lab[i]] >= min_size) ? 1 : 0;
    free(area); free(lab);
}

/* skimage.measure.label(uint8 binary) (test_dam.py:561): full connectivity, ids 1..N raster order. */
int orc_label8(const uint8_t *in, int H, int W, int32_t *lab)
{
    return label_bfs(in, H, W, 1, lab, NULL);
}

/* skimage.morphology.dilation(labels, disk(r)) (test_dam.py:563): max over {dy^2+dx^2 <= r^2}. */
void orc_dilate_disk_i32(const int32_t *in, int H, int W, int r, int32_t *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int32_t v = in[y * W + x];
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    if (dy * dy + dx * dx > r * r) continue;
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int32_t q = in[yy * W + xx];
                    if (q > v) v = q;
                }
            out[y * W + x] = v;
        }
}

/* the whole chain test_dam.py:546-563 */
int orc_cc_chain(const uint8_t *pred_inside, int H, int W, int min_area, int radius,
                 uint8_t *fill, uint8_t *small, int32_t *label, int32_t *final_)
{
    orc_fill_holes(pred_inside, H, W, fill);
    orc_remove_small(fill, H, W, min_area, small);
    int n = orc_label8(small, H, W, label);
    orc_dilate_disk_i32(label, H, W, radius, final_);
    return n;
}

/* ---------------------------------------------------------------------------------------------------
 * get_probmaps epilogue (test_dam.py:984, 1011-1013): softmax over 3 mask logits; softmax over C direction
 * logits with channel 0 multiplied by P(bg); argmax.  float32 arithmetic like torch CPU softmax
 * (max-subtracted exp, sum, divide).  exp rounding may differ by an ulp between libm and torch's vectorised
 * exp: the tests compare probabilities with a 1e-6 tolerance and class maps away from sub-1e-6 margins.
 * ------------------------------------------------------------------------------------------------- */
void orc_probmaps(const float *mask_logits, const float *dir_logits, int C, int H, int W,
                  float *prob, uint8_t *dcm)
{
    int n = H * W;
    for (int i = 0; i < n; ++i) {
        float m = mask_logits[i];
        for (int c = 1; c < 3; ++c) if (mask_logits[c * n + i] > m) m = mask_logits[c * n + i];
        float e[3], s = 0.f;
        for (int c = 0; c < 3; ++c) { e[c] = expf(mask_logits[c * n + i] - m); s += e[c]; }
        for (int c = 0; c < 3; ++c) prob[c * n + i] = e[c] / s;
        float dm = dir_logits[i];
        for (int c = 1; c < C; ++c) if (dir_logits[c * n + i] > dm) dm = dir_logits[c * n + i];
        float ds = 0.f, q[32];
        for (int c = 0; c < C; ++c) { q[c] = expf(dir_logits[c * n + i] - dm); ds += q[c]; }
        int a = 0;
        float best = (q[0] / ds) * prob[i];
        for (int c = 1; c < C; ++c) { float v = q[c] / ds; if (v > best) { best = v; a = c; } }
        dcm[i] = (uint8_t)a;
    }
}

/* ------------------------------------------------------------------------------------------------------
 * Watershed variant (postproc_other.py:15-99, ws branch :36-48).  ORACLE - test infrastructure only.
 *
 * orc_ws_dist: gen_inst_dst_map (:16-27) for a 4-connected label image: per instance the exact Euclidean distance to
 *   the nearest pixel outside the instance (scipy distance_transform_edt of the instance mask), scaled by 255 / max and
 *   truncated to uint8.  Brute force over growing square rings.
 * orc_watershed: skimage.segmentation.watershed(image, markers, mask=mask) with connectivity 1, non-compact, no
 *   watershed line, as published in skimage/segmentation/_watershed_cy.pyx (watershed_raveled): marker pixels are
 *   pushed in raster order; the smallest (value, age) is popped; each unlabelled neighbour inside the mask (raveled
 *   offsets -W, -1, +1, +W) takes the label and is pushed with its own image value and the next age.  skimage gives all
 *   marker pixels age 0 and leaves their mutual order to its heap; here they get ages 0, 1, 2, ... in raster order
 *   (PARITY UNPINNED for that tie-break: scikit-image is not available in this image).  A real binary heap on
 *   (value, age) - deliberately a different data structure from the HIP kernel's FIFO buckets.
 * ------------------------------------------------------------------------------------------------------ */
void orc_ws_dist(const int32_t *lab, int H, int W, uint8_t *canvas)
{
    int nlab = 0;
    for (int i = 0; i < H * W; ++i) if (lab[i] > nlab) nlab = lab[i];
    long long *d2 = (long long *)malloc(sizeof(long long) * (size_t)H * W);
    long long *mx = (long long *)calloc((size_t)nlab + 1, sizeof(long long));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int k = lab[y * W + x];
            long long best = -1;
            if (k > 0) {
                const int rmax = (H > W ? H : W);
                for (int r = 1; r <= rmax; ++r) {
                    if (best >= 0 && (long long)r * r >= best) break;
                    for (int dy = -r; dy <= r; ++dy) {
                        const int yy = y + dy;
                        if (yy < 0 || yy >= H) continue;
                        const int step = (dy == -r || dy == r) ? 1 : 2 * r;
                        for (int dx = -r; dx <= r; dx += step) {
                            const int xx = x + dx;
                            if (xx < 0 || xx >= W) continue;
                            if (lab[yy * W + xx] != k) {
                                const long long c = (long long)dy * dy + (long long)dx * dx;
                                if (best < 0 || c < best) best = c;
                            }
                        }
                    }
                }
                if (best < 0) best = 0;
                if (best > mx[k]) mx[k] = best;
            }
            d2[y * W + x] = best < 0 ? 0 : best;
        }
    for (int i = 0; i < H * W; ++i) {
        const int k = lab[i];
        canvas[i] = k > 0 ? (uint8_t)(255.0 * (sqrt((double)d2[i]) / sqrt((double)mx[k]))) : 0;
    }
    free(d2);
    free(mx);
}

typedef struct { int value; long long age; int index; } ws_elem;

static int ws_less(const ws_elem *a, const ws_elem *b) { return a->value != b->value ? a->value < b->value : a->age < b->age; }

void orc_watershed(const uint8_t *image, const int32_t *markers, const uint8_t *mask, int H, int W, int32_t *out)
{
    const int P = H * W;
    ws_elem *heap = (ws_elem *)malloc(sizeof(ws_elem) * (size_t)(P + 1));
    int n = 0;
    long long age = 0;
    for (int i = 0; i < P; ++i) out[i] = (markers[i] > 0 && mask[i]) ? markers[i] : 0;
#define WS_PUSH(V, I) do { ws_elem e_ = {(V), age++, (I)}; int c_ = n++; \
        while (c_ > 0) { int p_ = (c_ - 1) / 2; if (!ws_less(&e_, &heap[p_])) break; heap[c_] = heap[p_]; c_ = p_; } heap[c_] = e_; } while (0)
    for (int i = 0; i < P; ++i) if (out[i]) WS_PUSH(image[i], i);
    while (n > 0) {
        const ws_elem top = heap[0];
        const ws_elem lastv = heap[--n];
        int c = 0;
        while (1) {
            int l = 2 * c + 1, r = l + 1, m = -1;
            if (l < n) m = l;
            if (r < n && ws_less(&heap[r], &heap[l])) m = r;
            if (m < 0 || !ws_less(&heap[m], &lastv)) break;
            heap[c] = heap[m];
            c = m;
        }
        if (n > 0) heap[c] = lastv;
        const int p = top.index, y = p / W, x = p - y * W;
        const int nb[4] = {y > 0 ? p - W : -1, x > 0 ? p - 1 : -1, x < W - 1 ? p + 1 : -1, y < H - 1 ? p + W : -1};
        for (int j = 0; j < 4; ++j) {
            const int q = nb[j];
            if (q < 0 || !mask[q] || out[q]) continue;
            out[q] = out[p];
            WS_PUSH(image[q], q);
        }
    }
#undef WS_PUSH
    free(heap);
}
