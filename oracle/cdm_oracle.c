/*
 * ORACLE (test infrastructure, NOT product code): plain-C, single-threaded restatement of the reference's training
 * target generation `LabelEncoding.__call__` for the 3-class-PNG input branch with do_direction = 1
 * (my_transforms_direction.py:697-885; this branch :763-781 then :785-871) including `get_centerpoint2` (:650-685),
 * the 11x11 "Sobel" of data_prepare/SegFix_offset_helper.py:97-132 and the 8-bin quantisation of
 * DTOffsetHelper.align_angle (:311-341).  Pinned against tests/golden/cdm.npz (outputs of the reference itself, with
 * scipy stand-ins for the scikit-image calls: "skimage-semantics restated").
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* python round(): half to even (my_transforms_direction.py:672-673 under CPython / numba) */
static long py_round(double x) { return (long)nearbyint(x); }

/* get_centerpoint2 restricted to instance `id` of `inst` (the reference scans the whole image per instance) */
static void centerpoint(const int32_t *inst, int H, int W, int id, int y0, int y1, int x0, int x1, int *cy, int *cx)
{
    double P[8][2];
    for (int k = 0; k < 8; ++k) { P[k][0] = sin(2 * M_PI / 8 * k); P[k][1] = cos(2 * M_PI / 8 * k); }
    double now = -1;
    int bx = -1, by = -1;
    for (int i = y0; i <= y1; ++i)
        for (int j = x0; j <= x1; ++j) {
            if (inst[i * W + j] != id) continue;
            double ma = 0, mi = 10000000;
            for (int k = 0; k < 8; ++k) {
                double l = 0, r = 1000;
                for (int t = 0; t < 30; ++t) {
                    double mid = (l + r) / 2;
                    long nx = py_round(i + P[k][0] * mid), ny = py_round(j + P[k][1] * mid);
                    if (nx >= 0 && nx < H && ny >= 0 && ny < W && inst[nx * W + ny] == id) l = mid; else r = mid;
                }
                if (r > ma) ma = r;
                if (r < mi) mi = r;
            }
            double c = mi / ma;
            if (c > now) { now = c; by = i; bx = j; }
        }
    *cy = by; *cx = bx;
}

static int label8(const uint8_t *mask, int H, int W, int32_t *lab)
{
    int n = H * W, cnt = 0;
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memset(lab, 0, sizeof(int32_t) * (size_t)n);
    for (int s = 0; s < n; ++s) {
        if (!mask[s] || lab[s]) continue;
        int head = 0, tail = 0;
        q[tail++] = s; lab[s] = ++cnt;
        while (head < tail) {
            int p = q[head++], y = p / W, x = p % W;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if ((!dy && !dx) || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int r = yy * W + xx;
                    if (mask[r] && !lab[r]) { lab[r] = cnt; q[tail++] = r; }
                }
        }
    }
    free(q);
    return cnt;
}

/* optional tap for the tests: the float32 (row, column) gradient field the angles are taken from */
static float *g_dir_out = 0;
void orc_set_direction_field_out(float *p) { g_dir_out = p; }

/* The per-instance stage of LabelEncoding (:785-871) on a given (already dilated) instance map `inst` with ids in 1..cnt (ids
 * without pixels are skipped) and the pre-boundary inside mask: point map f32, direction classes u8, optional centres. */
static void direction_stage(const int32_t *inst, int cnt, const uint8_t *inside, int H, int W, float *point, uint8_t *direction,
                            int32_t *inst_out, int32_t *centers_out)
{
    const int n = H * W;
    /* bounding boxes of the (dilated) instances */
    int *bb = (int *)malloc(sizeof(int) * 4 * (cnt + 1));
    for (int k = 0; k <= cnt; ++k) { bb[4 * k] = H; bb[4 * k + 1] = -1; bb[4 * k + 2] = W; bb[4 * k + 3] = -1; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int k = inst[y * W + x];
            if (y < bb[4 * k]) bb[4 * k] = y;
            if (y > bb[4 * k + 1]) bb[4 * k + 1] = y;
            if (x < bb[4 * k + 2]) bb[4 * k + 2] = x;
            if (x > bb[4 * k + 3]) bb[4 * k + 3] = x;
        }
    float *dir = (float *)calloc((size_t)n * 2, sizeof(float));
    double *lp = (double *)calloc(n, sizeof(double));
    /* Sobel 11x11 (SegFix_offset_helper.py:102-132): ky[j][i] = j_/(i_^2+j_^2), kx = i_/(...), centre 0 */
    float ky[11][11], kx[11][11];
    for (int j = 0; j < 11; ++j)
        for (int i = 0; i < 11; ++i) {
            int j_ = j - 5, i_ = i - 5;
            if (!j_ && !i_) { ky[j][i] = kx[j][i] = 0.f; continue; }
            ky[j][i] = (float)(j_ / (double)(i_ * i_ + j_ * j_));
            kx[j][i] = (float)(i_ / (double)(i_ * i_ + j_ * j_));
        }
    uint8_t *nd = (uint8_t *)malloc(n);
    float *f = (float *)malloc(sizeof(float) * n);
    for (int k = 1; k <= cnt; ++k) {                                                      /* :800-835 */
        int y0 = bb[4 * k], y1 = bb[4 * k + 1], x0 = bb[4 * k + 2], x1 = bb[4 * k + 3];
        if (y1 < 0) continue;                                                             /* id vanished (cannot happen: dilation only grows) */
        int cy, cx;
        centerpoint(inst, H, W, k, y0, y1, x0, x1, &cy, &cx);                             /* :813 */
        lp[cy * W + cx] = 255.0;                                                          /* :816 */
        if (centers_out) { centers_out[2 * (k - 1)] = cy; centers_out[2 * (k - 1) + 1] = cx; }
        /* nucleus = dilation(nucleus, disk(1)) (:819); work inside the bbox grown by 1 (+5 for the stencil reads) */
        int Y0 = y0 - 1 < 0 ? 0 : y0 - 1, Y1 = y1 + 1 >= H ? H - 1 : y1 + 1;
        int X0 = x0 - 1 < 0 ? 0 : x0 - 1, X1 = x1 + 1 >= W ? W - 1 : x1 + 1;
        double dmax = 0;
        for (int y = Y0; y <= Y1; ++y)
            for (int x = X0; x <= X1; ++x) {
                int v = inst[y * W + x] == k;
                if (!v && y > 0) v = inst[(y - 1) * W + x] == k;
                if (!v && y < H - 1) v = inst[(y + 1) * W + x] == k;
                if (!v && x > 0) v = inst[y * W + x - 1] == k;
                if (!v && x < W - 1) v = inst[y * W + x + 1] == k;
                nd[y * W + x] = (uint8_t)v;
                if (v) {
                    double d = sqrt((double)(y - cy) * (y - cy) + (double)(x - cx) * (x - cx));   /* :822 EDT to one pixel */
                    if (d > dmax) dmax = d;
                }
            }
        for (int y = Y0; y <= Y1; ++y)
            for (int x = X0; x <= X1; ++x) {
                double d = sqrt((double)(y - cy) * (y - cy) + (double)(x - cx) * (x - cx));
                f[y * W + x] = nd[y * W + x] ? (float)((1 - d / (dmax + 0.0000001)) * 1.0) : 0.f;   /* :823-824, .float() :829 */
            }
        for (int y = Y0; y <= Y1; ++y)
            for (int x = X0; x <= X1; ++x) {
                if (!nd[y * W + x]) continue;                                             /* dir_i[nucleus==0] = 0 (:832) */
                double sy = 0, sx = 0;                                                    /* F.conv2d, padding 5 (:828-831) */
                for (int j = 0; j < 11; ++j) {
                    int yy = y + j - 5;
                    if (yy < Y0 || yy > Y1) continue;
                    for (int i = 0; i < 11; ++i) {
                        int xx = x + i - 5;
                        if (xx < X0 || xx > X1 || !nd[yy * W + xx]) continue;
                        sy += (double)ky[j][i] * f[yy * W + xx];
                        sx += (double)kx[j][i] * f[yy * W + xx];
                    }
                }
                dir[2 * (y * W + x)] = (float)sy;                                         /* dir_map[nucleus!=0] = 0; += dir_i (:833-834) */
                dir[2 * (y * W + x) + 1] = (float)sx;
            }
    }
    /* gaussian_filter(label_point, sigma=2) (scipy: radius 8, reflect), float64 (:842) */
    {
        double kk[17], s = 0;
        for (int i = -8; i <= 8; ++i) { kk[i + 8] = exp(-0.5 * i * i / 4.0); s += kk[i + 8]; }
        for (int i = 0; i < 17; ++i) kk[i] /= s;
        double *tmp = (double *)malloc(sizeof(double) * n);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                double a = lp[y * W + x] * kk[8];
                for (int i = 1; i <= 8; ++i) {
                    int ya = y - i, yb = y + i;
                    while (ya < 0 || ya >= H) ya = ya < 0 ? -ya - 1 : 2 * H - 1 - ya;
                    while (yb < 0 || yb >= H) yb = yb < 0 ? -yb - 1 : 2 * H - 1 - yb;
                    a += (lp[ya * W + x] + lp[yb * W + x]) * kk[8 + i];
                }
                tmp[y * W + x] = a;
            }
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                double a = tmp[y * W + x] * kk[8];
                for (int i = 1; i <= 8; ++i) {
                    int xa = x - i, xb = x + i;
                    while (xa < 0 || xa >= W) xa = xa < 0 ? -xa - 1 : 2 * W - 1 - xa;
                    while (xb < 0 || xb >= W) xb = xb < 0 ? -xb - 1 : 2 * W - 1 - xb;
                    a += (tmp[y * W + xa] + tmp[y * W + xb]) * kk[8 + i];
                }
                point[y * W + x] = (float)a;
            }
        free(tmp);
    }
    /* angle -> 8 bins -> +1, background 0 (:848-865, align_angle :324-339) */
    for (int i = 0; i < n; ++i) {
        if (!inside[i]) { direction[i] = 0; continue; }
        float ang = atan2f(dir[2 * i], dir[2 * i + 1]) * (180.0f / 3.14159265358979323846f);   /* np.degrees on float32 */
        int bin = 0;
        if (!(ang <= -157.5f || ang > 157.5f))
            for (int b = 1; b < 8; ++b) {
                float mid = -180.f + 45.f * b;
                if (ang > mid - 22.5f && ang <= mid + 22.5f) { bin = b; break; }
            }
        direction[i] = (uint8_t)(bin + 1);
    }
    if (inst_out) memcpy(inst_out, inst, sizeof(int32_t) * n);
    if (g_dir_out) memcpy(g_dir_out, dir, sizeof(float) * 2 * n);
    free(f); free(nd); free(lp); free(dir); free(bb);
}

/* LabelEncoding's direction branch for a caller-made instance map (the instance-label input branch, :752-760, hands in
 * dilation(postproc_other.process(...), disk(1))): inst i32 (ids need not be contiguous), inside u8 (new_label_inside). */
void orc_direction_from_instances(const int32_t *inst, const uint8_t *inside, int H, int W, float *point, uint8_t *direction,
                                  int32_t *centers_out)
{
    int cnt = 0;
    for (int i = 0; i < H * W; ++i) if (inst[i] > cnt) cnt = inst[i];
    direction_stage(inst, cnt, inside, H, W, point, direction, 0, centers_out);
}

/* in: channel 0 of the label PNG (u8).  out: label3 u8 {0,127,255}, point f32 (cast to f16 by the caller),
 * direction u8 0..8, and (optional) inst i32 = the dilated instance map, centers i32 [count][2].  Returns count. */
int orc_label_encoding(const uint8_t *in, int H, int W, uint8_t *label3, float *point, uint8_t *direction,
                       int32_t *inst_out, int32_t *centers_out)
{
    const int n = H * W;
    uint8_t *inside = (uint8_t *)malloc(n), *nl = (uint8_t *)malloc(n), *m1 = (uint8_t *)malloc(n);
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * n), *inst = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n; ++i) inside[i] = in[i] > 127.5 ? 1 : 0;                       /* :765-767 */
    /* boun = dilation(new_label) & ~erosion(new_label, disk(1)); both with the 4-neighbour cross, borders ignored (:768) */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int dil = inside[y * W + x], ero = inside[y * W + x];
            const int dy[4] = {-1, 1, 0, 0}, dx[4] = {0, 0, -1, 1};
            for (int k = 0; k < 4; ++k) {
                int yy = y + dy[k], xx = x + dx[k];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                if (inside[yy * W + xx]) dil = 1; else ero = 0;
            }
            nl[y * W + x] = (dil && !ero) ? 2 : inside[y * W + x];                       /* :769 */
        }
    for (int i = 0; i < n; ++i) { label3[i] = (uint8_t)(nl[i] / 2.0 * 255); m1[i] = nl[i] == 1; }   /* :781, :772 */
    int cnt = label8(m1, H, W, lab);                                                     /* :773 measure.label */
    /* label_instance = dilation(label_instance, disk(1)): max over the cross (:774) */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = lab[y * W + x];
            if (y > 0 && lab[(y - 1) * W + x] > v) v = lab[(y - 1) * W + x];
            if (y < H - 1 && lab[(y + 1) * W + x] > v) v = lab[(y + 1) * W + x];
            if (x > 0 && lab[y * W + x - 1] > v) v = lab[y * W + x - 1];
            if (x < W - 1 && lab[y * W + x + 1] > v) v = lab[y * W + x + 1];
            inst[y * W + x] = v;
        }
    direction_stage(inst, cnt, inside, H, W, point, direction, inst_out, centers_out);
    free(inst); free(lab); free(m1); free(nl); free(inside);
    return cnt;
}
