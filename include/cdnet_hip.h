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

#define CDNET_ABI_VERSION   5   /* 2, 3 (round 4): cdnet_conv_args grew (taps1, pool_out; dot_w, dot_b, dot_out) - a caller built against an older
                                 version must not pass its struct.  4 (round 5): cdnet_spin added; cdnet_tta_boost_argmax accepts point_mean == NULL
                                 for one view in its own frame (no struct changed).  5 (round 6): cdnet_box_copy / cdnet_box_mfma (box calibration),
                                 cdnet_tile_postproc (the fused tile post-processing chain); no struct changed */

int         cdnet_abi_version(void);
/* sizeof() of an argument struct of this header by name ("cdnet_conv_args", ...), 0 for an unknown name: a binding that mirrors the structs
 * (ctypes, cgo, JNI) checks its own layout against the library's at load time. */
size_t      cdnet_abi_sizeof(const char *struct_name);
const char *cdnet_last_error(void);
/* static description: "gfx950;wave64;..." */
const char *cdnet_build_info(void);
/* One wavefront that spins for `microseconds` of the constant 100 MHz counter on `stream` and touches no memory (at most 1 s).  The host
 * side's stream probe times two of them on two streams (cdnet_amd/streams.py): streams that share a hardware queue run them one after the
 * other.  Replaces nothing of the reference (nn.DataParallel has no streams of its own, train.py:185); a diagnostic of the runtime. */
int         cdnet_spin(int microseconds, void *stream);
/* Box calibration (bench.py's `box` object: boxes of one pool differ by 10-25 % on memory-bound kernels and in the clock they hold under
 * matrix load; rates are reported beside what the box itself grants).  Diagnostics of the runtime; nothing of the reference.
 * cdnet_box_copy: dst[i] = src[i] over `bytes` (multiple of 16, both 16-byte aligned) with float4 accesses, 4096 workgroups grid-stride.
 * cdnet_box_mfma: `workgroups` x `waves_per_wg` waves each run `iters` iterations of four v_mfma_f32_32x32x16_bf16 whose operands are
 * re-read from LDS (64 KB of patterns copied from `seed`, u32 [16384] = bf16 pairs); 4 x 32768 flop per wave and iteration.  `stamps`
 * (NULL or u64 [2 * workgroups]): per workgroup the loop's duration in shader cycles (s_memtime) and in ticks of the constant 100 MHz counter
 * (s_memrealtime) - clock = cycles / ticks x 100 MHz. */
int         cdnet_box_copy(const void *src, void *dst, size_t bytes, void *stream);
int         cdnet_box_mfma(const uint32_t *seed, float *sink, unsigned long long *stamps, int workgroups, int waves_per_wg, int iters, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * The whole post-processing chain of a batch of independent tiles (one view each, in its own frame) in two launches (round 6).
 * Replaces, per tile, test_dam.py:982-1015 (get_probmaps epilogue), getDirectionDiffMap.py:44-108 (generate_dd_map), test_dam.py:529-539
 * (point-guided boost + arg-max) and test_dam.py:546-563 (fill holes, remove small objects, label, dilate) - the same steps as
 * cdnet_probmaps + cdnet_ddm_codes + cdnet_tta_boost_argmax (V = 1) + cdnet_cc_chain, with bit-identical results: launch 1 leaves the
 * direction-difference codes (its direction-class window lives in LDS, the halo recomputed from the logits), launch 2 runs everything else of
 * a tile inside ONE workgroup with the tile in LDS (16-bit union-find).
 *   mask_logits f32 [B,3,H,W], dir_logits f32 [B,C,H,W] (C = 5 / 9 / 17), point f32 [B,H,W]      (device, NCHW as Unet.forward returns them)
 *   lut_host: int8 [C*C] HOST memory as for cdnet_ddm_codes; nbr / extra_zero as there
 *   prob f32 [B,3,H,W] or NULL, dcm u8 [B,H,W] or NULL (stage outputs); minmax i32 [B,2] (min, max DDM code per tile: equal = the reference's
 *   assertion test_dam.py:535 would fail); pred u8 [B,H,W]; fill / small u8 [B,H,W] or NULL; label i32 [B,H,W] or NULL; final i32 [B,H,W];
 *   counts i32 [B] or NULL
 * Shapes: W a multiple of 64 and H * W <= 65536 (cdnet_tile_postproc_workspace_bytes returns 0 otherwise: use the per-step entries). */
size_t cdnet_tile_postproc_workspace_bytes(int B, int C, int H, int W);
int cdnet_tile_postproc(const float *mask_logits, const float *dir_logits, const float *point, int B, int C, int H, int W,
                        const int8_t *lut_host, int nbr, int extra_zero, int min_area, int radius, void *workspace, size_t workspace_bytes,
                        float *prob, uint8_t *dcm, int32_t *minmax, uint8_t *pred, uint8_t *fill, uint8_t *small, int32_t *label,
                        int32_t *final_, int32_t *counts, void *stream);

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
 *   ddm16 is 255), pred u8 [I][H][W] = argmax class.  Any of prob_mean / ddm16 may be NULL.  point_mean may be NULL only for ONE view in
 *   its own frame (V == 1, view_xform[0] == 0, prob_mean NULL, H*W a multiple of 4, points 16-byte aligned): the mean over views is the view
 *   itself, nothing is averaged or stored (the 256x256 tile pipeline).
 *   pmax_ws: f32 workspace [I] (global max of point_mean).
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_tta_boost_argmax(const float *probs, const float *points, const uint8_t *codes, const int32_t *minmax,
                           int I, int V, const int *view_xform_host, int H, int W,
                           float *prob_mean, float *point_mean, uint8_t *ddm16, uint8_t *pred,
                           float *pmax_ws, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Watershed variant of the instance post-processing.   Replaces postproc_other.py:15-99 `process(pred, model_mode,
 * min_size, ws=True)` for the non-'dcan' modes, steps :36-48:
 *   scipy.ndimage.measurements.label (4-connected) -> gen_inst_dst_map (:16-27: per-instance distance_transform_edt
 *   scaled to 0..255 uint8) -> marker = label(binary_erosion(binary_fill_holes(dist > 125))) with labels smaller than
 *   min_size removed -> skimage.segmentation.watershed(-dist [uint8 wrap], marker, mask=pred) -> remove small labels.
 * pred u8 [N][H][W] already thresholded (0 / non-zero = `pred > 0.5`, :33-34).
 * Outputs: labels i32 [N][H][W] (required, marker ids are kept like the reference does); dist u8 and marker i32 stage
 * outputs (each may be NULL).  workspace: cdnet_watershed_workspace_bytes(N, H, W) bytes.
 * ---------------------------------------------------------------------------------------------------- */
size_t cdnet_watershed_workspace_bytes(int N, int H, int W);
int cdnet_watershed_process(const uint8_t *pred, int N, int H, int W, int min_size, void *workspace, size_t workspace_bytes,
                            uint8_t *dist, int32_t *marker, int32_t *labels, void *stream);
/* The ws = False branch of the same function (postproc_other.py:49-52; forced for model_mode 'unet' / 'micronet', :35):
 *   scipy.ndimage.binary_fill_holes -> measurements.label (4-connected, ids in raster order) -> remove_small_objects on the label
 *   image (labels with fewer than min_size pixels become 0; the other ids are kept).  pred u8 [N][H][W] non-zero = foreground;
 *   labels i32 [N][H][W]; workspace: cdnet_watershed_workspace_bytes(N, H, W) bytes. */
int cdnet_fill_label_process(const uint8_t *pred, int N, int H, int W, int min_size, void *workspace, size_t workspace_bytes,
                             int32_t *labels, void *stream);

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


/* ------------------------------------------------------------------------------------------------------
 * Convolution stack (MFMA implicit GEMM, NHWC bf16, fp32 accumulate).   Replaces the cuDNN/MIOpen calls behind
 * nn.Conv2d / nn.ConvTranspose2d / nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d / F.pad / torch.cat in
 * models/dam/model_unet_rev1.py:86-170,244-287 and models/unet.py:8-50,90-106 (see DESIGN.md for the fusion map).
 *
 * A convolution reads up to two sources (virtual channel concat, torch.cat order).  Each source may carry the
 * PRODUCER's per-channel affine (BatchNorm as scale/shift), a residual tensor added before the ReLU, a ReLU, a 2x2
 * max-pool and an F.pad offset - all applied while the input tile is staged, never as separate passes.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cdnet_conv_src {
    const uint16_t *x;      /* 16-bit NHWC [N][Hs][Ws][C] (bf16, or fp16 when f16 = 1) */
    const uint16_t *res;    /* optional bf16 tensor of the same shape added before the ReLU (ResidualUnit) */
    const float *scale;     /* optional per-channel affine of the producer (BN): v*scale[c]+shift[c] */
    const float *shift;
    int C;                  /* channels of this source (multiple of CK) */
    int Hs, Ws;             /* stored spatial size */
    int pool;               /* 1: logical input = maxpool2x2(transformed source), size Hs/2 x Ws/2 */
    int relu;
    int off_y, off_x;       /* F.pad: logical (y,x) reads source (y-off_y, x-off_x); outside -> 0 */
    int f16;                /* storage of x / res: 0 = bf16, 1 = fp16 (raw pre-BatchNorm outputs, residual branches),
                               2 = fp32 (the pointers then address float tensors; fp32-precision path, cdnet_conv_args.f32) */
    int row_stride;         /* elements between rows (0 = dense: Ws*C); images are Hs*row_stride apart.  Lets the
                               space-to-depth view of a 2x-upsampled gradient be read without a copy. */
} cdnet_conv_src;

typedef struct cdnet_conv_args {
    cdnet_conv_src src[2];
    int nsrc;
    const uint16_t *w;      /* packed bf16 weights (cdnet_pack_conv_weights) */
    const float *bias;      /* [Cout] or NULL: added in the epilogue (before oscale/oshift) */
    const float *oscale;    /* epilogue affine (eval-mode BN fold) or NULL */
    const float *oshift;
    int orelu;
    uint16_t *out;          /* bf16 NHWC [N][H*ostride][W*ostride][out_cstride], channels [out_coff, out_coff+Cout) */
    int Cout, out_cstride, out_coff;
    float *stats;           /* NULL or f32 [N*npar*tiles][2][Cout]: per-tile channel sum / sum of squares of the
                               fp32 accumulators (training-mode BatchNorm statistics, bias excluded) */
    int N, H, W;            /* logical input size (= output size / ostride) */
    int taps;               /* 9: 3x3 pad 1; 1: 1x1; 4: sub-pixel 2x2 of ConvTranspose2d(k4,s2,p1) */
    int npar;               /* 1, or 4 sub-pixel parities (ConvTranspose2d stride 2) */
    int ostride;            /* 1, or 2 for the transposed convolutions */
    int nchunk;             /* total Cin chunks over both sources (two-source 3x3 launches of the 16-bit path: one more is allowed - a padding chunk
                               behind the second source whose packed weights are zeros, making the count even for conv_ws16_kernel's out-image form.  Valid on conv_ws16_kernel ONLY
                               (bounded buffer descriptors): ask cdnet_conv_ws_eligible == 2 first, any other kernel returns CDNET_E_ARG) */
    int tile, CK, BN;       /* kernel configuration: spatial tile (16 or 8), Cin chunk, Cout tile */
    int out_f16;            /* 1: store the output as fp16 instead of bf16 */
    int debug;              /* 0 in production.  Kernel-selection switches for the tests: 32 = never take the wave-specialised
                               persistent kernels (conv_ws_kernel; conv_ws32_kernel of the fp32 path), 64 = take them even for
                               small launches; bits 8..: at most (debug >> 8) persistent workgroups per output-channel tile (long
                               runs of tiles on small test shapes; conv_ws32_kernel); 16 (16-bit path): conv_ws16_kernel's quad-request form whatever the
                               launch's size (production: tensors beyond the Infinity Cache); other bits: ablations of tools/bench_conv.py */
    int ws;                 /* 0, or 2 (fp32 mode, conv_ws32_kernel only - ask cdnet_conv_ws_eligible): the launch also leaves the first
                               BatchNorm-backward pass of the layer its output feeds; eres / oscale / oshift / eres_scale / eres_shift /
                               stats then carry that layer's raw output, scale, shift, mean, invstd and the partial rows f32
                               [1024][2][Cout] (at most 4 x 256 workgroups write rows; see cdnet_bn_backward_finalize) */
    int f32;                /* 1: fp32 precision - x / res / eres / out are fp32 tensors (every source has f16 = 2), `w` is the split
                               pack (mode | CDNET_PACK_SPLIT), each product runs as three bf16 MFMAs over (hi, lo) operand pairs with
                               fp32 accumulation; CK = 16, no pooled sources (materialise them).  0: the 16-bit path. */
    /* Optional fused residual epilogue (ResidualUnit, model_unet_rev1.py:161-170: relu2(bn2(conv2(.)) + conv_1x1(x))): the
     * convolution result (bias added, rounded to fp16) is the residual r; eres = the other branch, a dense fp16 (eres_f16 = 1)
     * or bf16 tensor [N][H][W][Cout] with an optional per-channel affine (training: raw conv2 output x BatchNorm scale /
     * shift).  out = bf16( [relu]( (eres * eres_scale + eres_shift) + r ) ), bit-identical to cdnet_src_materialize over the
     * same pair.  The convolution's own epilogue affine (oscale / oshift: eval-mode BatchNorm fold) applies before the add, so
     * the identity-shortcut blocks of HRNet in eval mode (seg_hrnet_rev1.py:76-92, 113-133: relu(bn(conv(.)) + x), eres = x in
     * bf16) use it too.  NULL = off.  Needs ostride 1, out_coff 0, out_cstride = Cout, no stats, orelu = 0. */
    const uint16_t *eres;
    const float *eres_scale, *eres_shift;
    int eres_f16, eres_relu;
    /* taps of the chunks of the SECOND source: 0 = `taps` (the usual concatenation).  1 with taps = 9 (16-bit path, conv_ws16_kernel
     * only - ask cdnet_conv_ws_eligible): the second source contributes through a 1x1 convolution, its chunks carry the centre tap
     * alone.  `w` then holds, per output-channel tile, the nine-tap chunks of source 0 followed by the one-tap chunks of source 1
     * (two cdnet_pack_conv_weights packs laid end to end).  One launch computes relu(bn2(conv2(h)) + conv_1x1(x)) of a residual unit
     * in eval mode (model_unet_rev1.py:161-170) with the BatchNorm scale folded into conv2's weights and shift + bias in `oshift`. */
    int taps1;
    int pad_;
    /* Optional second output: nn.MaxPool2d(2, 2) of the (ReLU-activated, bf16) output, dense [N][H/2][W/2][Cout] - the 'M' layers of the
     * torchvision VGG16-BN encoder (model_unet_rev1.py:40-41) fused into the producing convolution's store path.  16-bit path:
     * conv_ws16_kernel's out-image form (cdnet_conv_ws_eligible answers 2 with the pointer set); fp32 mode (float tensors): conv_ws32_kernel
     * with plain sources, >= 4 chunks, BN = 64 (answer 1).  Otherwise leave it NULL and call cdnet_src_materialize.  Needs orelu = 1,
     * out_coff = 0, out_cstride = Cout. */
    uint16_t *pool_out;
    /* Optional 1x1 classifier over the (activated, bf16-rounded) output, fused into the store path: dot_out[n][y][x] = dot_b[0] +
     * sum_c dot_w[c] * out[n][y][x][c] as fp32 [N][H][W] - the DAM head's point logit (model_unet_rev1.py:252-253: point_conv over the
     * point feature, whose only other reader is nobody: with dot_out set `out` may be NULL and the 64-channel feature is then never
     * stored).  16-bit path, conv_ws16_kernel's out-image form only (cdnet_conv_ws_eligible answers 2 with the pointers set); needs
     * Cout <= BN (one output-channel tile), out_coff = 0; no pool_out beside it.  NULL = off. */
    const float *dot_w;     /* [Cout] */
    const float *dot_b;     /* [1], device memory */
    float *dot_out;
} cdnet_conv_args;

/* packed element count for a weight tensor; nchunk = Cin/CK over all sources */
size_t cdnet_conv_packed_weight_elems(int Cout, int nchunk, int taps, int CK, int BN, int npar);
/* fp32 master weights -> packed bf16.  mode 0: Conv2d [Cout][Cin][KH][KW] forward; mode 1: Conv2d backward-data
 * (Cout/Cin are the ROLES in the backward GEMM: Cout := original in_channels, Cin := original out_channels);
 * mode 2: ConvTranspose2d [Cin][Cout][4][4] k4 s2 p1 forward (4 parities); mode 3: ConvTranspose2d [Cin][Cout][2][2]
 * k2 s2 forward (4 parities); mode 7: backward-data of mode 6 (Cout := 4*C of the view, Cin := the conv's out_channels);
 * mode 4 / 5: backward-data of the k4 s2 p1 / k2 s2 transposed convolution, expressed as a
 * 3x3 / 1x1 convolution over the space-to-depth view of the output gradient (two sources = the two row parities, each
 * with 2*Cout_t channels ordered (column parity, channel)); Cout := transposed-conv in_channels, Cin := 4*out_channels.
 * mode 6: forward Conv2d [Cout][C][3][3] stride 2 pad 1 (HRNet transition / fuse layers, seg_hrnet_rev1.py:228-247,
 * 436-443) as a 3x3 convolution over the same space-to-depth view of its INPUT; Cin := 4*C. */
#define CDNET_PACK_SPLIT 16   /* OR into `mode`: fp32-precision pack = per Cin chunk the bf16(w) image followed by the
                                bf16(w - bf16(w)) image; twice cdnet_conv_packed_weight_elems elements */
int cdnet_pack_conv_weights(const float *w, void *packed, int Cout, int Cin, int KH, int KW, int CK, int BN, int mode,
                            void *stream);
/* mode 0 (optionally | CDNET_PACK_SPLIT) with every output channel's weights multiplied by cout_scale[cout] in fp32 before the rounding:
 * the eval-mode BatchNorm scale folded into the convolution (nn.BatchNorm2d in eval mode after nn.Conv2d, model_unet_rev1.py:40-41,
 * 112-114, 146-170), so that the epilogue is shift + ReLU only - what conv_ws16_kernel takes as the accumulators' initial value. */
int cdnet_pack_conv_weights_scaled(const float *w, const float *cout_scale, void *packed, int Cout, int Cin, int KH, int KW, int CK,
                                   int BN, int mode, void *stream);
/* The same packing for many tensors in one launch (the training step re-packs every layer's forward and backward-data
 * weights after each optimizer update - train_util_dam.py:308 optimizer.step()).  `table` is device scratch of
 * cdnet_pack_batch_table_bytes(n_jobs) bytes; upload = 1 copies the job table there (first call, or whenever the jobs
 * changed), upload = 0 re-runs the table uploaded before. */
typedef struct cdnet_pack_job {
    const float *w;
    void *packed;
    int Cout, Cin, KH, KW, CK, BN, mode, pad_;
} cdnet_pack_job;
size_t cdnet_pack_batch_table_bytes(int n_jobs);
int cdnet_pack_conv_weights_batch(const cdnet_pack_job *jobs, int n_jobs, void *table, size_t table_bytes, int upload,
                                  void *stream);
int cdnet_conv_forward(const cdnet_conv_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Model-path streaming kernels (not convolutions).
 * cdnet_input_pack: f32 NCHW [N][C<=16][H][W] (the ToTensor output, my_transforms_direction.py:945) -> bf16 NHWC
 *   [N][H][W][16] with zero-padded channels.
 * cdnet_bn_fold_eval: nn.BatchNorm2d in eval(): scale = g/sqrt(rv+eps), shift = b + (conv_bias - rm)*scale.
 * cdnet_bn_finalize_train: nn.BatchNorm2d in train(): reduces the per-tile statistics emitted by cdnet_conv_forward
 *   (stats f32 [T][2][C], `count` = N*H*W elements per channel) into scale/shift for the consumers, saves
 *   mean/invstd for backward and updates running_mean/var (momentum, unbiased variance); all reductions are
 *   deterministic (fixed tree, fp64).
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_input_pack(const float *x, int N, int C, int H, int W, void *out_bf16_nhwc16, void *stream);
/* the same layout change for the fp32-precision path: out f32 NHWC [N][H][W][16] */
int cdnet_input_pack_f32(const float *x, int N, int C, int H, int W, float *out_f32_nhwc16, void *stream);
int cdnet_bn_fold_eval(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                       const float *conv_bias, float eps, int C, float *scale, float *shift, void *stream);
int cdnet_bn_finalize_train(const float *stats, int T, int C, float count, const float *gamma, const float *beta,
                            const float *conv_bias, float eps, float momentum, float *running_mean,
                            float *running_var, float *scale, float *shift, float *save_mean, float *save_invstd,
                            void *stream);

/* A 64-channel head feature F = [relu](raw*scale + shift + res), recomputed on the fly from its stored pieces. */
typedef struct cdnet_head_feat {
    const uint16_t *raw;    /* 16-bit NHWC [N][H][W][64] */
    const uint16_t *res;    /* optional residual, same shape and format */
    const float *scale;     /* optional per-channel affine */
    const float *shift;
    int relu;
    int f16;                /* 0: raw/res are bf16, 1: fp16, 2: fp32 (pointers address float tensors; nothing is rounded to 16 bits) */
} cdnet_head_feat;

/* Direction-aware-mask head: replaces models/dam/model_unet_rev1.py:258-263 (point_conv, directionAtt,
 * direction_conv, maskAtt, mask_conv; revAttention :8-17).  head_weights: device f32 block of
 * CDNET_HEAD_WEIGHT_FLOATS = { point_conv.w[64], direction_conv.w[9][64], mask_conv.w[3][64], point_conv.b,
 * direction_conv.b[9], mask_conv.b[3], directionAtt.w, maskAtt.w[9] }.
 * Outputs f32 NCHW: mask [N][3][H][W], point [N][1][H][W], direction [N][9][H][W] (the tuple Unet.forward returns).
 * f3->raw = NULL (f1 / f2 plain bf16): `point` is an INPUT - it already holds point_conv(point feature) + bias, left there by the
 * convolution that produced the point feature (cdnet_conv_args.dot_out) - and only mask / direction are written. */
#define CDNET_HEAD_WEIGHT_FLOATS 855
int cdnet_dam_head_forward(const cdnet_head_feat *f1, const cdnet_head_feat *f2, const cdnet_head_feat *f3,
                           const float *head_weights, int N, int H, int W, float *mask, float *point,
                           float *direction, void *stream);
/* plain UNet classifier (models/unet.py:75,104 final_conv 64->K): f32 NCHW logits from a 64-channel feature */
int cdnet_final_conv1x1(const cdnet_head_feat *f, const float *w, const float *b, int K, int N, int H, int W,
                        float *out, void *stream);
/* bias gradient of a convolution that is not followed by BatchNorm (plain UNet's ConvTranspose2d, models/unet.py:30):
 * db[c] = sum over the npix pixels of grad_out (bf16 NHWC [npix][C]).  workspace: cdnet_bias_grad_workspace_floats(C). */
size_t cdnet_bias_grad_workspace_floats(int C);
int cdnet_bias_grad(const uint16_t *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db, void *stream);
int cdnet_bias_grad_f32(const float *grad_out, size_t npix, int C, float *workspace, size_t workspace_floats, float *db, void *stream);   /* fp32-precision path */
/* backward of that classifier (plain-UNet training, train_util.py:126-200 loss.backward()): dlogits f32 [N][K][H][W] ->
 * df bf16 NHWC [N][H][W][64] (gradient of the activated feature), dw f32 [K][64], db f32 [K].  K <= 20 (the 3-class mask and
 * 5- / 9- / 17-class direction classifiers of the ablation heads, models/dam/model_unet_MandD*.py).
 * workspace: cdnet_final_conv1x1_backward_workspace_floats() floats. */
size_t cdnet_final_conv1x1_backward_workspace_floats(void);
int cdnet_final_conv1x1_backward(const cdnet_head_feat *f, const float *w, const float *dlogits, int K, int N, int H, int W,
                                 uint16_t *df, float *workspace, size_t workspace_floats, float *dw, float *db, void *stream);

/* cdnet_src_materialize: a convolution source with its pending transform - BatchNorm scale / shift, residual add, ReLU,
 * nn.MaxPool2d(2, 2[, ceil_mode]) (model_unet_rev1.py:268-287 backbone 'M' layers, unet.py:19), F.pad offset - written out
 * as a plain bf16 NHWC tensor out[N][H][W][C]: bit-identical to what cdnet_conv_forward stages on the fly for the same
 * source (H, W = the logical size after the pool).  Consumers of a max-pooled training-mode activation read this copy. */
/* Non-zero when cdnet_conv_forward runs these arguments on a producer / consumer kernel: 2 = conv_ws16_kernel (16-bit path, launches
 * without statistics: the only one that takes cdnet_conv_args.taps1 = 1), 1 = conv_ws_kernel, or in fp32 mode conv_ws32_kernel - the
 * only one that takes cdnet_conv_args.ws = 2 (the BatchNorm-backward sums of the layer the output feeds, beside the stores).
 * Nothing is launched. */
int cdnet_conv_ws_eligible(const cdnet_conv_args *args);

int cdnet_src_materialize(const cdnet_conv_src *src, int N, int H, int W, uint16_t *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Backward-weight of a convolution (the dW half of loss.backward(), train_util_dam.py:307), one call per input
 * source of the layer.  `src` is the layer's forward source (its lazy transform is re-applied while staging),
 * grad_out the bf16 NHWC gradient w.r.t. the layer's raw output [N][H*ostride][W*ostride][Cout].
 * Work is split over `ksplit` slices of the 8x16-pixel tiles; partial fp32 slabs (cdnet_conv_wgrad_slab_floats)
 * are summed in a fixed order (bit-reproducible) and scattered into dw, the PyTorch-layout gradient:
 *   mode 0: Conv2d [Cout][Cin_real][KH][KW]; mode 2: ConvTranspose2d k4s2p1 [Cin_real][Cout][4][4];
 *   mode 3: ConvTranspose2d k2s2 [Cin_real][Cout][2][2].
 * src_coff: first channel of this source inside the weight's input-channel axis (torch.cat order); Csrc_real: real
 * channels of the source (3 for the zero-padded RGB stem).  ci_tiles in {1,2,4}: the workgroup owns
 * ci_tiles*32 input channels x (4/ci_tiles)*32 output channels.
 * ---------------------------------------------------------------------------------------------------- */
size_t cdnet_conv_wgrad_slab_floats(int C_src, int Cout, int taps, int npar, int ci_tiles, int ksplit);
int cdnet_conv_backward_weight(const cdnet_conv_src *src, int src_coff, int Csrc_real, int Cin_real,
                               const uint16_t *grad_out, int Cout, int N, int H, int W, int taps, int npar,
                               int ostride, int ci_tiles, int ksplit, float *slab, float *dw, int mode, void *stream);

/* Deferred split-K reduction: `mode | CDNET_WGRAD_DEFER_REDUCE` makes cdnet_conv_backward_weight stop after the slabs (which then
 * must stay untouched - one slab buffer per call); cdnet_wgrad_reduce_batch sums the slabs of any number of such calls and scatters
 * into their dw in ONE launch, bit-identical to the per-call reduction (same fixed order).  The table lives in device memory:
 * entries filled on the host by cdnet_wgrad_reduce_desc_fill (block0 = sum of the `blocks` of the entries before it), copied by
 * the caller; total_blocks = block0 + blocks of the last entry.  (The reference has no counterpart: autograd accumulates dW inside
 * cuDNN's backward-filter call, train_util_dam.py:307.) */
#define CDNET_WGRAD_DEFER_REDUCE 0x100
typedef struct cdnet_wgrad_reduce_desc {
    const float *slab;
    float *dw;
    int ksplit, npar, ci_blocks, co_blocks, taps, CI, CO, Csrc_real, Cin_real, src_coff, Cout, mode;
    int block0, blocks;
} cdnet_wgrad_reduce_desc;
int cdnet_wgrad_reduce_desc_fill(int C_src, int src_coff, int Csrc_real, int Cin_real, int Cout, int taps, int npar, int ci_tiles,
                                 int ksplit, const float *slab, float *dw, int mode, int block0, cdnet_wgrad_reduce_desc *out);
int cdnet_wgrad_reduce_batch(const cdnet_wgrad_reduce_desc *table_dev, int n, int total_blocks, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Training-only streaming kernels.  Together with cdnet_conv_forward (backward-data packs) and
 * cdnet_conv_backward_weight they replace loss.backward() / optimizer.step() of train_util_dam.py:303-308.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cdnet_grad_in {     /* gradient w.r.t. a tensor as delivered by ONE of its consumers */
    const uint16_t *g;             /* bf16 NHWC [N][Hg][Wg][C] */
    int Hg, Wg;
    int oy, ox;                    /* the consumer read the tensor through F.pad offsets (oy, ox) */
    int pooled;                    /* the consumer read maxpool2x2 of the tensor: route to the first maximum */
    int coff, cstride;             /* channel slice [coff, coff+C) of a tensor with cstride channels (concat consumers) */
    int pad_;
} cdnet_grad_in;

typedef struct cdnet_bn_bwd_args {
    const uint16_t *raw;           /* the layer's stored forward output [N][H][W][C] (fp16 when f16 = 1, else bf16) */
    const uint16_t *res;           /* residual that was added before the ReLU (same format) or NULL; with relu = 2: the stored
                                      post-ReLU OUTPUT of the unit (bf16) - the ReLU mask is read from it instead of being recomputed */
    const float *scale, *shift;    /* forward per-channel affine (BatchNorm as applied), NULL = identity */
    const float *mean, *invstd;    /* saved batch statistics; NULL = no BatchNorm (gradient passes through) */
    cdnet_grad_in gin[3];
    int ngin;
    int f16, relu;                 /* relu: 0 none, 1 relu(affine [+ res]), 2 mask = res > 0 (fused residual epilogue of cdnet_conv_forward) */
    int N, H, W, C;
} cdnet_bn_bwd_args;

/* BatchNorm(train) + residual + ReLU backward with the consumers' max-pool / pad routing and the summation of up to
 * three consumer gradients fused in:  dz = (sum_k g_k) * [activation > 0];  dgamma = sum dz*xhat;  dbeta = sum dz;
 * draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) (bf16); dz_out (optional, bf16) = dz for the residual
 * branch.  workspace: cdnet_bn_backward_workspace_floats(C) floats. */
size_t cdnet_bn_backward_workspace_floats(int C);
/* The two passes of cdnet_bn_backward as separate calls for the plain case (one same-size gradient source, BatchNorm + ReLU, no
 * residual branch).  cdnet_bn_backward_stats (16-bit tensors) = reduce + finalize; it also writes ktab f32 [7][C] =
 * scale | shift | mean | invstd | k1 | k2 | k3, the table cdnet_bn_backward_apply reads (the second pass alone: draw = k1 * (dz - k2 -
 * xhat * k3); 16-bit and fp32 tensors).  The trainer uses finalize + apply behind a backward-data launch that carried the sums
 * (cdnet_conv_args.ws = 2, fp32 mode). */
int cdnet_bn_backward_stats(const cdnet_bn_bwd_args *args, const float *gamma, float *dgamma, float *dbeta, float *workspace,
                            size_t workspace_floats, float *ktab, void *stream);
int cdnet_bn_backward_apply(const cdnet_bn_bwd_args *args, const float *ktab, uint16_t *draw, void *stream);
/* The finalize pass alone, over partial rows f32 [nb][2][C] produced by the backward-data launch that computed this layer's output
 * gradient: cdnet_conv_args.ws = 2 with eres = the layer's raw fp16 forward output [N][H][W][Cout], oscale / oshift / eres_scale /
 * eres_shift = its BatchNorm scale / shift / batch mean / invstd (f32 [Cout]), stats = the partial rows (4 per workgroup of the
 * producer / consumer kernel, at most 1024: zero-fill the buffer once and pass nb = 1024).  The first BatchNorm-backward pass then costs no pass of
 * its own over the two tensors.  Writes dgamma, dbeta and rows 4..6 (k1 | k2 | k3) of ktab - what cdnet_bn_backward_apply reads; rows 0..3 are not touched
 * (fp32 mode: eres = the raw fp32 output, conv_ws32_kernel). */
int cdnet_bn_backward_finalize(const cdnet_bn_bwd_args *args, const float *gamma, float *dgamma, float *dbeta, const float *partial,
                               int nb, float *ktab, void *stream);
int cdnet_bn_backward(const cdnet_bn_bwd_args *args, const float *gamma, float *dgamma, float *dbeta, float *workspace,
                      size_t workspace_floats, uint16_t *draw, uint16_t *dz_out, void *stream);

/* DAM head backward (model_unet_rev1.py:258-263): gradients of the three logit maps (f32 NCHW) -> gradients of the
 * three 64-channel features (bf16 NHWC) and of the head weights (f32, CDNET_HEAD_WEIGHT_FLOATS layout). */
size_t cdnet_dam_head_backward_workspace_floats(int N, int H, int W);
int cdnet_dam_head_backward(const cdnet_head_feat *f1, const cdnet_head_feat *f2, const cdnet_head_feat *f3,
                            const float *head_weights, const float *dmask, const float *dpoint, const float *ddir,
                            int N, int H, int W, uint16_t *df1, uint16_t *df2, uint16_t *df3, float *workspace,
                            size_t workspace_floats, float *dhead_weights, void *stream);

/* The five-term CDNet loss (train_util_dam.py:167-276; loss.py:131-260) and its gradient w.r.t. the logits.
 * label u8 {0,1,2}, dirlab u8 0..8, point target f16, weight map u8 (divided by 20 on the fly, :102).
 * quirk_sample0 = 1 reproduces train_util_dam.py:139 (direction one-hot masked by sample 0's foreground).
 * losses[11] = {total, direction CE, direction weighted dice, MSE, CE, dice, then the pixel-level metrics of
 * train_util_dam.py:279-293 (argmax direction class == 1 vs direction label == 1, utils.py:67-110), averaged over the batch:
 * accuracy, IoU, recall, precision, F1}.  dmask/dpoint/ddir may all be NULL.
 * Label content is validated on the device: a mask class > 2 or a direction class > 8 makes every entry of `losses` NaN
 * (indices are clamped, nothing is read or written out of bounds) - the reference's nn.NLLLoss raises on such targets.
 */
size_t cdnet_dam_loss_workspace_floats(int B, int P);
int cdnet_dam_loss(const float *mask, const float *point, const float *direction, const uint8_t *label,
                   const uint8_t *dirlab, const uint16_t *point_target_f16, const uint8_t *weight_u8, int B, int H, int W,
                   int quirk_sample0, float *workspace, size_t workspace_floats, float *losses, float *dmask,
                   float *dpoint, float *ddir, void *stream);
/* the same loss for direction maps of direction_classes = 5, 9 or 17 classes (options.py:45 "4 8 16" + background; the 4- and
 * 16-direction ablation models models/dam/model_unet_MandD4.py / model_unet_MandD16.py): direction / ddir f32 [B][classes][H][W],
 * dirlab u8 0..classes-1; loss.py:216-260 runs its cyclic neighbour terms over 1..classes-1 and averages over `classes`.
 * cdnet_dam_loss is this entry with 9 classes. */
size_t cdnet_dam_loss_classes_workspace_floats(int B, int P, int direction_classes);
int cdnet_dam_loss_classes(const float *mask, const float *point, const float *direction, const uint8_t *label,
                           const uint8_t *dirlab, const uint16_t *point_target_f16, const uint8_t *weight_u8, int B, int H, int W,
                           int direction_classes, int quirk_sample0, float *workspace, size_t workspace_floats, float *losses,
                           float *dmask, float *dpoint, float *ddir, void *stream);

/* validate() loss mix of train_util_dam.py:367-636 (default options): per-sample sums of ONE pass over the logits, combined on the
 * host by cdnet_amd.train_util_dam.validate.  sums f32 [B][CDNET_VAL_SUMS]:
 *   0..2 sum p_c [label==c], 3..5 sum p_c, 6..8 sum [label==c], 9 sum -log p_label (unweighted mask CE, :499-505);
 *   10..18 / 19..27 / 28..36 the same three sums for the direction probabilities with channel 0 multiplied by P(background)
 *   (:564-566) against the one-hot direction target - channel = dir_rank_host[class value] (the rank among the batch's unique
 *   values, :462-468; -1 = absent), zeroed off SAMPLE 0's foreground (:469); 37 sum w * -log q_dir (:553-559);
 *   38 sum (point - target / 255)^2 (:575-580); 39..41 tp, fp, fn of (argmax mask == 1) vs (label == 1) (:585-590).
 * workspace: cdnet_dam_val_sums_workspace_floats(B, H * W) floats. */
#define CDNET_VAL_SUMS 42
size_t cdnet_dam_val_sums_workspace_floats(int B, int P);
int cdnet_dam_val_sums(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                       const uint16_t *point_target_f16, const uint8_t *weight_u8, const int *dir_rank_host, int B, int H, int W,
                       float *workspace, size_t workspace_floats, float *sums, void *stream);
/* the same for direction_classes = ND in {5, 9, 17} (model_unet_MandD4 / MandD16): dir_rank_host holds ND entries and
 * sums f32 [B][15 + 3 ND]: 0..9 as above, the three direction blocks at 10, 10 + ND, 10 + 2 ND (ND wide each), the five scalars
 * {weighted direction CE, MSE, tp, fp, fn} at 10 + 3 ND.  cdnet_dam_val_sums is this entry with ND = 9. */
size_t cdnet_dam_val_sums_classes_workspace_floats(int B, int P, int direction_classes);
int cdnet_dam_val_sums_classes(const float *mask, const float *point, const float *dirn, const uint8_t *label, const uint8_t *dirlab,
                               const uint16_t *point_target_f16, const uint8_t *weight_u8, const int *dir_rank_host,
                               int direction_classes, int B, int H, int W, float *workspace, size_t workspace_floats, float *sums,
                               void *stream);

/* torch.optim.Adam step (utils.py:915-918: betas (0.9, 0.99), L2 weight decay added to the gradient) on flat fp32
 * buffers; `step` is 1-based; grad_scale multiplies the gradient first (1/world_size after an all-reduce). */
int cdnet_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Sliding-window inference.  Replaces utils.split_forward_dam (utils.py:658-726) and the construction of the eight
 * test-time-augmentation views (test_dam.py:313-385) without ever materialising a flipped / rotated / padded image.
 * cdnet_window_pack: img f32 [C<=16][H][W] -> ny*nx windows bf16 NHWC [ny*nx][tile_h][tile_w][16] of view `view_xform`
 *   (bit0 hflip, bit1 vflip, bit2 rot90 ccw first); window (ky,kx) starts at (ky*stride, kx*stride) of the view, pixels
 *   beyond the view are the zero padding of utils.py:665-676.
 * cdnet_window_stitch: window outputs f32 [ny*nx][K][tile_h][tile_w] -> out f32 [K][Hv][Wv], keeping for every pixel
 *   the last window whose interior (overlap/2 trimmed, except at the image border) contains it (utils.py:683-712).
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_window_pack(const float *img, int C, int H, int W, int view_xform, int tile_h, int tile_w, int stride, int ny,
                      int nx, void *out_bf16_nhwc16, void *stream);
/* fp32-precision path: the same windows as f32 NHWC [ny*nx][tile_h][tile_w][16] */
int cdnet_window_pack_f32(const float *img, int C, int H, int W, int view_xform, int tile_h, int tile_w, int stride, int ny,
                          int nx, float *out_f32_nhwc16, void *stream);
int cdnet_window_stitch(const float *tiles, int K, int tile_h, int tile_w, int stride, int overlap, int ny, int nx, int Hv,
                        int Wv, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Training-target generation.  Replaces my_transforms_direction.py:687-885 `LabelEncoding.__call__` (3-class-PNG input,
 * do_direction = 1) with get_centerpoint2 (:650-685), Sobel.kernel (SegFix_offset_helper.py:97-132) and
 * DTOffsetHelper.align_angle / angle_to_vector / vector_to_label (:311-341, 423-450, 486-506).
 *   label_ch0 u8 [N][H][W] (channel 0 of the label PNG; > 127 = inside)
 *   -> label3 u8 {0,127,255}, point f16 [N][H][W] (Gaussian sigma 2 of 255-impulses at the nucleus centres),
 *      direction u8 0..8 (centripetal class + 1, background 0); optional inst i32 (grown instance ids), counts i32 [N].
 * rays_host: 16 doubles {sin(2 pi k/8), cos(2 pi k/8)} k = 0..7 and gauss_host: 9 doubles (normalised half kernel,
 * centre first) are evaluated by the HOST math library so that they equal the reference's math.sin/cos and scipy weights.
 * max_instances: upper bound of nuclei per image (ids above it are dropped - check counts).
 * ---------------------------------------------------------------------------------------------------- */
size_t cdnet_label_encoding_workspace_bytes(int N, int H, int W, int max_instances);
int cdnet_label_encoding(const uint8_t *label_ch0, int N, int H, int W, int max_instances, const double *rays_host,
                         const double *gauss_host, void *workspace, size_t workspace_bytes, uint8_t *label3,
                         uint16_t *point_f16, uint8_t *direction, int32_t *inst_out, int32_t *counts_out, void *stream);
/* The instance-label input branch of the same transform (my_transforms_direction.py:752-760, `label_level_len > 2`; the
 * reference's default training data <label_dir>/train_ins): label_inst i32 [N][H][W] holds instance ids.  inside = id > 0 (dropped
 * when an image has fewer than 5 foreground pixels, :755), boundary = pixels whose 4-neighbourhood holds different ids (:759:
 * dilation(label) & ~erosion(label, disk(1)) on the integer ids), instances = dilation(postproc_other.process((new_label == 1) * 255,
 * 'modelName', min_size=5), disk(1)) - the watershed variant (cdnet_watershed_process) - then the same centre / direction / point
 * stage.  N <= 64.  Outputs as cdnet_label_encoding; counts = max_instances (watershed ids are not contiguous). */
size_t cdnet_label_encoding_instances_workspace_bytes(int N, int H, int W, int max_instances);
int cdnet_label_encoding_instances(const int32_t *label_inst, int N, int H, int W, int max_instances, const double *rays_host,
                                   const double *gauss_host, void *workspace, size_t workspace_bytes, uint8_t *label3,
                                   uint16_t *point_f16, uint8_t *direction, int32_t *inst, int32_t *counts, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * HRNet fuse / residual sums.  Replaces the y = y + x[j] / y = y + F.interpolate(..., mode='bilinear') / relu(y) chains of
 * seg_hrnet_rev1.py:256-283, the residual adds of BasicBlock / Bottleneck (:76-92, :113-133) and the F.upsample + torch.cat
 * of the four branches (:528-533).  out[N][H][W] = [relu](sum of 1..4 terms), every term a bf16 NHWC tensor with C
 * channels (optionally with a per-channel affine), either [H][W] or a lower resolution that is up-sampled bilinearly
 * (align_corners = False).  The output may be a
 * channel slice [out_coff, out_coff + C) of pixels out_cstride wide (0 = C).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cdnet_fuse_term {
    const uint16_t *x;      /* bf16 (or fp16 when f16 = 1) NHWC [N][Hs][Ws][C] */
    int Hs, Ws;
    const float *scale;     /* optional per-channel affine applied to the term (training mode: raw conv output x BatchNorm */
    const float *shift;     /* scale / shift), both NULL or both set */
    int f16, pad_;
} cdnet_fuse_term;
int cdnet_fuse_sum(const cdnet_fuse_term *terms, int nterm, int N, int H, int W, int C, int relu, uint16_t *out, int out_cstride,
                   int out_coff, void *stream);

/* backward pieces of the HRNet training step (loss.backward() through seg_hrnet_rev1.py:256-283, 436-443):
 * cdnet_upsample_bilinear_backward: transpose of the bilinear up-sampling done inside cdnet_fuse_sum: dout bf16 NHWC
 *   [N][H][W] (channel slice coff / cstride allowed) -> din bf16 [N][Hs][Ws][C];
 * cdnet_s2d_to_nhwc: gradient computed in the space-to-depth view [N][H2][W2][(a, b, c)] -> [N][2*H2][2*W2][C]
 *   (input gradient of a stride-2 convolution; weight pack mode 7 = backward-data of mode 6). */
int cdnet_upsample_bilinear_backward(const uint16_t *dout, int N, int H, int W, int C, int dout_cstride, int dout_coff, int Hs, int Ws,
                                     uint16_t *din, void *stream);
int cdnet_s2d_to_nhwc(const uint16_t *in, int N, int H2, int W2, int C, uint16_t *out, void *stream);
/* the three entries above for fp32 tensors (fp32 precision mode: every term has f16 = 2; same arithmetic, nothing is rounded to 16 bits) */
int cdnet_fuse_sum_f32(const cdnet_fuse_term *terms, int nterm, int N, int H, int W, int C, int relu, float *out, int out_cstride,
                       int out_coff, void *stream);
int cdnet_upsample_bilinear_backward_f32(const float *dout, int N, int H, int W, int C, int dout_cstride, int dout_coff, int Hs, int Ws,
                                         float *din, void *stream);
int cdnet_s2d_to_nhwc_f32(const float *in, int N, int H2, int W2, int C, float *out, void *stream);

/* cdnet_grad_sum: backward of the sums above (autograd's AddBackward + ReluBackward over seg_hrnet_rev1.py:76-92, 256-283 and the
 * torch.cat of :533): out[npix][C] = [mask > 0] * sum of 1..6 gradient contributions, each a bf16 tensor [npix][cstride] read at
 * channel slice [coff, coff + C) (cstride 0 = C).  mask = the stored (post-ReLU) forward output, or NULL for a sum without ReLU. */
typedef struct cdnet_grad_term {
    const uint16_t *g;
    int cstride, coff;
} cdnet_grad_term;
int cdnet_grad_sum(const cdnet_grad_term *terms, int nterm, const uint16_t *mask, long long npix, int C, uint16_t *out, void *stream);
/* the same over fp32 tensors (fp32 precision mode: `g` of every term, mask and out point at float data; strides in elements) */
int cdnet_grad_sum_f32(const cdnet_grad_term *terms, int nterm, const float *mask, long long npix, int C, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Instance metrics (stats_utils.py): one pass over a ground-truth and a predicted label image [N][plane] i32 gives the
 * per-label areas (area_*[N][cap], ids must be < cap <= 65536) and a sparse table of pairwise intersections (open
 * addressing, hash_slots a power of two >= 2 x expected pairs: hash_keys[N][slots] = (true_id << 16) | pred_id, 0 = empty;
 * hash_counts = pixels).  get_fast_aji :7-106, get_fast_pq :182-276 and get_dice_1 :323-335 are finished on the host from
 * these (cdnet_amd/stats_utils.py).  err_flag (1 int): 1 = id out of range, 2 = table full.
 * cdnet_remap_label = remap_label :361-392 (by_size = False); scratch i32 [N][cap].
 * ---------------------------------------------------------------------------------------------------- */
int cdnet_label_pair_histogram(const int32_t *true_lab, const int32_t *pred_lab, int N, int plane, int cap, int hash_slots,
                               int32_t *area_true, int32_t *area_pred, uint32_t *hash_keys, int32_t *hash_counts, int32_t *err_flag,
                               void *stream);
int cdnet_remap_label(const int32_t *lab, int N, int plane, int cap, int32_t *scratch, int32_t *out, int32_t *err_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CDNET_HIP_H */
