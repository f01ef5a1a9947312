/*
 * cdnet_hip.h - C ABI of libcdnet_hip.so: the MI355X (gfx950) implementation of CDNet's data-parallel hot path.
 *
 * The reference (honglianghe/CDNet) is pure Python: it has no FFI layer of its own.  Its "operator API" for
 * this path is a set of Python call signatures.  Each entry point below names the reference
 * call site(s) it replaces (paths relative to the reference repository root) - that is what a maintainer would
 * bind with ctypes (INTEGRATION.md shows the stubs; cdnet_amd/_lib.py is the binding the package itself uses).
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer unless its name ends in `_host`;
 *   - the caller owns all buffers (outputs and workspaces); nothing here allocates, frees or synchronises;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 = ok, otherwise a CDNET_E_* code; cdnet_last_error() gives a thread-local message;
 *   - images are dense row-major; batches are the leading dimension; activations are NHWC bf16 internally;
 *   - no global mutable state; re-entrant across streams.
 */
#ifndef CDNET_HIP_H
#define CDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDNET_OK            0
#define CDNET_E_ARG         1   /* bad argument (null pointer, size, unsupported configuration) */
#define CDNET_E_WORKSPACE   2   /* workspace too small */
#define CDNET_E_LAUNCH      3   /* HIP launch error */

#define CDNET_ABI_VERSION   1

int         cdnet_abi_version(void);
const char *cdnet_last_error(void);
/* static description: "gfx950;wave64;..." */
const char *cdnet_build_info(void);

/* ------------------------------------------------------------------------------------------------------
 * Direction-difference map.   Replaces data_prepare/getDirectionDiffMap.py:44-108 `generate_dd_map`
 * (+ `circshift` :14-42 and DTOffsetHelper.label_to_vector SegFix_offset_helper.py:246-261), which
 * test_dam.py:479-487 calls 8x per image.
 *
 * cdnet_ddm_codes: dcm u8 [N][H][W] (direction classes 0..classes-1) -> code u8 [N][H][W] in {0,1,2} and
 *   minmax i32 [N][2] = (min code, max code) per image.  `lut_host` int8 [classes*classes] holds
 *   round(cos(v_a, v_b)) for every class pair (host memory, copied into the launch); nbr = 8 (9/17 classes)
 *   or 4 (5 classes); extra_zero = 1 reproduces the reference's never-written cosine channels (17 classes).
 * cdnet_ddm_normalize: out f32 [N][H][W] = (code - min) / (max - min)  (NaN when a map is constant, as in
 *   the reference).
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_ddm_codes(const uint8_t *dcm, int N, int H, int W, int classes, const int8_t *lut_host, int nbr,
                    int extra_zero, uint8_t *code, int32_t *minmax, void *stream);
int cdnet_ddm_normalize(const uint8_t *code, const int32_t *minmax, int N, int H, int W, float *out,
                        void *stream);

/* ------------------------------------------------------------------------------------------------------
 * get_probmaps epilogue.   Replaces test_dam.py:982-1015: softmax over the 3 mask logits, softmax over the
 * direction logits with channel 0 multiplied by P(background), argmax -> direction class map.
 * mask_logits f32 [N][3][H][W], dir_logits f32 [N][C][H][W] -> prob f32 [N][3][H][W], dcm u8 [N][H][W].
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_probmaps(const float *mask_logits, const float *dir_logits, int N, int C, int H, int W,
                   float *prob, uint8_t *dcm, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * TTA mean + DDM fuse + point-guided boundary boost + argmax.   Replaces test_dam.py:445-450 (mean of the
 * un-flipped views), :479-491 (per-view DDM, mean), :529-539 (boost, argmax).
 *
 * Views are stored in their OWN frame; `view_xform[v]` (host, V ints, 0..7) tells how view v was made from the
 * image: bit0 = horizontal flip, bit1 = vertical flip, bit2 = rotated 90deg counter-clockwise first
 * (PIL rotate(90, expand=True), test_dam.py:372); the kernels read through the inverse map, which is what
 * np.flip / np.rot90(k=3) do in test_dam.py:356-441.  For rotated views the stored frame is [W][H].
 *   probs  f32 [I][V][3][h_v][w_v]   points f32 [I][V][h_v][w_v]
 *   codes  u8  [I][V][h_v][w_v]      minmax i32 [I][V][2]          (from cdnet_ddm_codes, N = I*V)
 * Outputs (image frame [H][W]): prob_mean f32 [I][3][H][W] (before the boost), point_mean f32 [I][H][W],
 *   ddm16 u8 [I][H][W] = 16 * mean_v DDM_v when every view has min=0,max in {1,2} (else the f64 path is used and
 *   ddm16 is 255), pred u8 [I][H][W] = argmax class.  Any of prob_mean / ddm16 may be NULL.
 *   pmax_ws: f32 workspace [I] (global max of point_mean).
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_tta_boost_argmax(const float *probs, const float *points, const uint8_t *codes, const int32_t *minmax,
                           int I, int V, const int *view_xform_host, int H, int W,
                           float *prob_mean, float *point_mean, uint8_t *ddm16, uint8_t *pred,
                           float *pmax_ws, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Connected-component chain.   Replaces test_dam.py:546-563:
 *   scipy.ndimage.binary_fill_holes -> skimage.morphology.remove_small_objects(min_area)
 *   -> skimage.measure.label (8-connectivity, ids in raster order) -> skimage.morphology.dilation(disk(radius)).
 * pred u8 [N][H][W]; a pixel is foreground when pred == fg_value (test_dam.py:538 `pred == 1`).
 * Outputs: final i32 [N][H][W] (required); fill u8, small u8, label i32 (stage outputs, each may be NULL);
 *   counts i32 [N] (number of instances, may be NULL).
 * workspace: cdnet_cc_workspace_bytes(N, H, W) bytes.
 * ---------------------------------------------------------------------------------------------------- */
size_t cdnet_cc_workspace_bytes(int N, int H, int W);
int cdnet_cc_chain(const uint8_t *pred, int fg_value, int N, int H, int W, int min_area, int radius,
                   void *workspace, size_t workspace_bytes,
                   uint8_t *fill, uint8_t *small, int32_t *label, int32_t *final_, int32_t *counts,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CDNET_HIP_H */
