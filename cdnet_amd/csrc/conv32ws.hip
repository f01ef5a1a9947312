// Wave-specialised, persistent form of the fp32-precision 3x3 convolution (conv32.hip's arithmetic: fp32 NHWC tensors in HBM, every
// product a*b as three bf16 MFMAs over split operands, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulation) for the layers that carry
// the fp32 step: every 3x3 convolution with at least four 16-channel chunks on full 16x16 tiles - forward, backward-data and the
// space-to-depth backward of the transposed convolutions.
//
//   * one 8-wave workgroup per CU, persistent over an XCD-contiguous run of 16x16-pixel tiles, BN output channels per workgroup;
//   * waves 4..7 (movers) only move data in: the chunk's packed weights (hi image | lo image, 36 KB for 64 output channels) by LDS-DMA
//     into a two-slot ring - issued first in every interval, from inline assembly, so that they land while the rest of the interval
//     runs (a DMA the compiler knows about makes it guard every later LDS access of the wave with a wait for it); fp32 halo vectors of
//     a 16-channel chunk HBM -> registers (two chunks in flight, straight-line code with clamped addresses so that the loads are
//     counted, not drained) -> source transform (BatchNorm scale / shift, residual, ReLU) -> split into the hi / lo bf16 planes of a
//     two-slot LDS ring (32 B per pixel and plane, k-halves swizzled by the halo row's parity: conflict-free ds_read_b128);
//   * waves 0..3 (consumers) read fragments and issue MFMAs: wave w owns rows 4w..4w+3 of the tile (two 32-pixel M blocks) x all
//     BN/32 N blocks - 8 fragment reads per 12 MFMAs, each issued a whole tap (12 MFMAs) ahead of its use, one per MFMA gap: a
//     scheduling fence behind every MFMA pins that order (left alone, the scheduler sinks every read to just in front of its first use
//     and the matrix pipe waits out the LDS latency tap by tap).  Two accumulator sets: while one takes the current tile's MFMAs the
//     finished tile's epilogue rides in the gaps between them - channel statistics in the first chunk interval, and the tile itself
//     straight from the accumulators to HBM (bias / folded BatchNorm / ReLU in fp32; the 32 lanes of a half write one 128-byte line of
//     a pixel), a quarter of it per chunk interval, one 256-byte store every ~7 MFMAs.  No LDS out image, no second pass;
//   * one barrier per chunk interval (108 MFMAs per consumer wave); 120 KB of LDS.
// The MFMA sequence per accumulator (chunk, tap, lo*hi, hi*lo, hi*hi) is conv_f32_kernel's, so the outputs are bit-identical
// (tests/test_gpu_fp32_kernels.py compares the two and both against fp64); the per-tile statistics are summed in conv_ws_kernel's
// order (blocks (ni, mi), registers ascending, the two lane halves, then the four waves).
// Measured on the dominant layer (3x3 64->64 @256x256 x16, tools/bench_conv_ws32.py, in a loop of launches): 212-214 us against 255-267 us
// for conv_f32_kernel; consumers alone 156 us, movers alone 83 us, without the 268 MB of stores 170 us (whoever issues them - movers
// through an LDS out image, consumers as 16-byte or as 4-byte stores: +40 us each way; the matrix pipe runs at the clock the chip
// sustains under the combined load).
//
// Replaces the same reference lines as conv.hip / conv32.hip: models/dam/model_unet_rev1.py:86-170,244-287 (cuDNN fp32 convolutions).
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include <stdlib.h>

using namespace cdnet;


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// 8 fp32 values -> hi / lo bf16 vectors (conv32.hip: split8)
__device__ __forceinline__ void split8(const float *v, u32x4 &hi, u32x4 &lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        // two plain subtractions instead of the v_pk_add_f32 the compiler would make of them: a packed fp32 instruction of a mover wave
        // is expensive beside the consumers' MFMA stream (measured on wgrad_ws32_kernel, wgrad.hip: wg_split8)
        float d0, d1;
        asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x[0]), "v"(__builtin_bit_cast(float, hb << 16)));
        asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x[1]), "v"(__builtin_bit_cast(float, hb & 0xffff0000u)));
        const f32x2 df = {d0, d1};
        const bf16x2 l = __builtin_convertvector(df, bf16x2);
        hi[k] = hb;
        lo[k] = __builtin_bit_cast(unsigned, l);
    }
}

// One LDS-DMA piece (64 lanes x 16 bytes = 1 KB): global (uniform base + this lane's byte offset) -> LDS (uniform byte address + lane * 16).
// Inline assembly on purpose: the compiler does not count it, so (a) it puts no alias guard - s_waitcnt vmcnt(<everything up to the
// DMA>) - in front of this wave's later LDS accesses, which lets the weight DMA of an interval be issued FIRST and land while the halo
// chunk is transformed and written, and (b) the wave waits for it itself: conv_ws32_kernel's stream_sync counts the vector-memory
// operations issued after it.  M0 (the DMA's LDS base) is saved and restored around the piece.
__device__ __forceinline__ void glds_piece(const void *gbase, unsigned lane_bytes, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_bytes), "s"(gbase), "s"(lds_dst) : "memory");
}

template <int BN>
struct Ws32Lds {
    static constexpr int TH = 16, TW = 16, CK = 16, TAPS = 9;
    static constexpr int PSTR = 32;                               // bytes per pixel and plane (16 bf16)
    static constexpr int NPIX = (TH + 2) * (TW + 2);
    static constexpr int A_PLANE = NPIX * PSTR;
    static constexpr int A_SLOT = 2 * A_PLANE;                    // hi plane | lo plane
    static constexpr int B_PLANE = TAPS * CK * BN * 2;            // one image of a chunk's packed weights
    static constexpr int W_SLOT = 2 * B_PLANE;                    // hi image | lo image: the packed chunk as it lies in memory
    static constexpr int STATS_BYTES = 2 * 4 * 2 * BN * 4;        // double-buffered [wave][sum | sumsq][BN]
    static constexpr int RAW_Q = 4 * 32 * 32 * 4;                // BNS: a quarter of a tile's raw forward output (one 32 x 32 block per consumer wave)
    __host__ __device__ static int bytes(int ctot, bool bns = false) { return 2 * A_SLOT + 2 * W_SLOT + STATS_BYTES + 2 * ((ctot + 7) / 8 * 8) * 4 + (bns ? 2 * RAW_Q : 0); }
};

// XF: source transform decided by the launcher - 0 plain fp32, 1 x * scale + shift -> ReLU (training-mode BatchNorm source), 2 anything
// (run-time flags: optional scale / shift, residual operand, ReLU).  STATS: per-tile channel sums of the accumulators.
// The launcher guarantees: taps == 9, npar == 1, ostride == 1, H and W multiples of 16, nchunk >= 1 (>= 4 with BNS), no pooled source, no fused residual
// epilogue.
// BNS (cdnet_conv_args.ws == 2, backward-data launches): the output is the gradient w.r.t. the activated output of a BatchNorm + ReLU
// layer whose only reader this convolution was.  The first pass of that layer's BatchNorm backward - the channel sums of dz = dY * [act > 0]
// and of dz * xhat - rides in the consumers' deferred epilogue: the movers bring the layer's raw forward output (eres, fp32
// [N][H][W][Cout]) for the quarter of the finished tile that leaves in the NEXT interval into a 16 KB LDS buffer by DMA (two buffers;
// [wave][32 pixels][32 couts], the block (mi, ni) of every consumer wave), the consumers read it back in accumulator layout (ds_read_b32:
// 32 consecutive couts of a pixel) one element ahead of its use.  oscale | oshift | eres_scale | eres_shift = that layer's BatchNorm
// scale | shift | mean | invstd (f32 [Cout]); stats = partial rows f32 [4 * gridDim.x][2][Cout] for cdnet_bn_backward_finalize (one row
// per consumer wave, written once at the end of the run).  No bias / epilogue affine / ReLU in this mode; BN = 64, Cout % 64 == 0.
// NCS: chunks per tile known to the compiler - 0: four or more (the epilogue of a tile leaves over four intervals), 1: one, 2: two or three
// MIX (round 4, cdnet_conv_args.taps1 = 1): the chunks of the SECOND source carry one tap (the centre) instead of nine - a residual unit's 1x1
// branch as extra K steps of its second 3x3 convolution, eval mode (model_unet_rev1.py:161-170; conv16ws.hip has the 16-bit form).  The
// packed weights hold, per output-channel tile, the nine-tap chunks followed by the one-tap chunks (hi | lo images each).
// POOL (cdnet_conv_args.pool_out, eval mode): nn.MaxPool2d(2, 2) of the activated output beside the stores - the 'M' layers of the VGG16-BN
// encoder (model_unet_rev1.py:40-41).  A lane holds whole 2x2 windows: registers r, r + 1, r + 8, r + 9 (r = 0, 2, 4, 6) of a block are two
// neighbouring columns of its two tile rows - four pooled pixels per block and lane, one more store per four (conv16ws.hip has the 16-bit form).
template <int BN, int XF, bool STATS, bool BNS = false, int NCS = 0, bool MIX = false, bool POOL = false>
__global__ __launch_bounds__(512) void conv_ws32_kernel(ConvArgs A) {
    static_assert(!BNS || (BN == 64 && !STATS), "the BatchNorm-backward statistics epilogue serves 64-channel blocks of backward-data launches");
    static_assert(!MIX || (!STATS && !BNS && NCS == 0), "one-tap chunks: inference launches with at least four nine-tap chunks");
    static_assert(!POOL || (!STATS && !BNS && !MIX && NCS == 0), "fused max-pool: plain inference launches with at least four chunks");
    using L = Ws32Lds<BN>;
    constexpr int TH = 16, TW = 16, CK = 16, TAPS = 9, PSTR = L::PSTR, HW_ = TW + 2, NPIX = L::NPIX;
    constexpr int NPW = BN / 32, MPW = 2;
    constexpr int VPP = CK / 8, NA = (NPIX * VPP + 255) / 256;    // 8-channel vectors per halo pixel / per mover thread and chunk
    const int NCH = A.nchunk;
    const int n0c = A.src[0].C / CK;                              // chunks of the first source (MIX: the nine-tap ones)
    constexpr int W1_SLOT = 2 * CK * BN * 2;                      // MIX: hi | lo images of a one-tap chunk

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_a = smem;
    unsigned char *lds_w = smem + 2 * L::A_SLOT;
    float *s_stats = reinterpret_cast<float *>(lds_w + 2 * L::W_SLOT);
    float *s_xf = reinterpret_cast<float *>(lds_w + 2 * L::W_SLOT + L::STATS_BYTES);
    const int xfs_ = ((A.src[0].C + (A.nsrc > 1 ? A.src[1].C : 0)) + 7) / 8 * 8;
    unsigned char *lds_raw = lds_w + 2 * L::W_SLOT + L::STATS_BYTES + 2 * xfs_ * 4;      // BNS only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: tile rows, LDS bases and DMA pieces stay scalar
    const int c0n = A.src[0].C, ctot = c0n + (A.nsrc > 1 ? A.src[1].C : 0);
    const int xfs = (ctot + 7) / 8 * 8;
    const int cout_tile = blockIdx.y;
    const int cout0 = cout_tile * BN;

    // this workgroup's contiguous run of tiles; XCD k (workgroups k, k+8, ...) serves the k-th eighth of the tiles (conv_ws_kernel)
    const int tiles_x = A.W / TW, tiles_y = A.H / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int T = A.N * tiles_img;
    int t_lo, t_hi;
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        if ((G & 7) == 0) {
            const int xcd = b & 7, idx = b >> 3, nw = G >> 3;
            const long long x0 = (long long)T * xcd / 8, x1 = (long long)T * (xcd + 1) / 8;
            t_lo = (int)(x0 + (x1 - x0) * idx / nw);
            t_hi = (int)(x0 + (x1 - x0) * (idx + 1) / nw);
        } else {
            t_lo = (int)((long long)T * b / G);
            t_hi = (int)((long long)T * (b + 1) / G);
        }
    }
    const int ntl = t_hi - t_lo;
    const int S = ntl * NCH;                                      // chunk intervals of this workgroup
    if (S == 0) return;

    if (wave >= 4) {
        // ================================ movers ================================
        const int ptid = tid - 256, pw = wave - 4;
        const int slot = ptid % VPP;
        f32x4 pa[2][NA][2];                      // two chunks of halo vectors in flight (8 channels = two float4 each)
        unsigned eo[2][NA];                      // byte offsets of the requests (read again for a residual operand); bit 31 = zero fill
        int hyx[NA], doff[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pix = (ptid + i * 256) / VPP;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            hyx[i] = ptid + i * 256 < NPIX * VPP ? ((hy << 8) | hx) : 0x1f1f;      // (31 = a vector that never exists: bit 31 of the masks is always set)
            doff[i] = pix * PSTR + ((slot ^ (hy & 1)) * 16);
        }
        // Requests go through buffer descriptors (one per source, tensors below 2 GB): a vector outside the image / the source window
        // carries bit 31 in its byte offset, the range check returns zeros for it - no select, no 64-bit address arithmetic and, for plain
        // sources, no mask per vector (every vector instruction of a mover wave costs the consumers' MFMA stream issue time).  Validity of a
        // tile's halo rows / columns = two scalar bit masks per tile and source.
        const unsigned src_bytes0 = (unsigned)A.N * A.src[0].Hs * (A.src[0].row_stride ? A.src[0].row_stride : A.src[0].Ws * A.src[0].C) * 4u;
        const unsigned src_bytes1 = A.nsrc > 1 ? (unsigned)A.N * A.src[1].Hs * (A.src[1].row_stride ? A.src[1].row_stride : A.src[1].Ws * A.src[1].C) * 4u : 0u;
        const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 ? A.src[1].x : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].res ? A.src[0].res : A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 && A.src[1].res ? A.src[1].res : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
        };
        // bits [lo, hi) clear, everything else set (lo, hi clamped to [0, 31])
        auto bad_mask = [](int lo, int hi) -> unsigned {
            lo = lo < 0 ? 0 : (lo > 31 ? 31 : lo);
            hi = hi < lo ? lo : (hi > 31 ? 31 : hi);
            return ~(((1u << hi) - 1u) & ~((1u << lo) - 1u));
        };
        // issue cursor: chunk ik of the tile at (in_, iy0, ix0); saturates on the last chunk of the run
        int ik = 0, ic = 0, in_, iy0, ix0;
        {
            in_ = t_lo / tiles_img;
            const int r = t_lo - in_ * tiles_img, ty = r / tiles_x;
            iy0 = ty * TH; ix0 = (r - ty * tiles_x) * TW;
        }
        int o_n = in_, o_y0 = iy0, o_x0 = ix0;   // BNS: the tile the consumers are accumulating,
        int s_n = 0, s_y0 = 0, s_x0 = 0;         //      the finished tile whose epilogue rides in the current one
        bool s_ok = false;
        int kt = 0;                              //      chunk-in-tile index of the current interval
        int ck = 0;                              // commit cursor: chunk-in-tile index
        unsigned ge[NA];                         // byte offset of channel 0 of the tile's vectors in the current source, bit 31 = zero fill
        const int n0 = A.src[0].C / CK;
        auto chunk_src = [&](int k, int &si, int &cc0) {
            if (k < n0) { si = 0; cc0 = k * CK; } else { si = 1; cc0 = (k - n0) * CK; }
        };
        auto issue = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            int si, cc0;
            chunk_src(ik, si, cc0);
            const ConvSrc &s = A.src[si];
            if (ik == 0 || ik == n0) {
                const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
                // halo row r <-> y = iy0 - 1 + r, halo column c <-> x = ix0 - 1 + c; valid = inside the image and the source window
                const int ylo = s.off_y > 0 ? s.off_y : 0, yhi = A.H < s.off_y + s.Hs ? A.H : s.off_y + s.Hs;
                const int xlo = s.off_x > 0 ? s.off_x : 0, xhi = A.W < s.off_x + s.Ws ? A.W : s.off_x + s.Ws;
                const unsigned rowbad = bad_mask(ylo - (iy0 - 1), yhi - (iy0 - 1)), colbad = bad_mask(xlo - (ix0 - 1), xhi - (ix0 - 1));
                const unsigned img_b = (unsigned)((in_ * s.Hs + (iy0 - 1 - s.off_y)) * rs + (ix0 - 1 - s.off_x) * s.C) * 4u;
                const unsigned rs_b = (unsigned)rs * 4u, c_b = (unsigned)s.C * 4u;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const unsigned hy = (unsigned)hyx[i] >> 8, hx = (unsigned)hyx[i] & 0xffu;
                    const unsigned t = (rowbad >> hy) | (colbad >> hx);
                    ge[i] = ((img_b + hy * rs_b + hx * c_b + (unsigned)slot * 32u) & 0x7fffffffu) | (t << 31);
                }
            }
            const __amdgpu_buffer_rsrc_t rsx = si ? rsx1 : rsx0;
            const unsigned cc0_b = (unsigned)cc0 * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned voff = ge[i] + cc0_b;                   // (a zero-fill vector keeps bit 31: beyond every tensor the launcher admits)
                eo[R][i] = voff;
                pa[R][i][0] = bload(rsx, voff);
                pa[R][i][1] = bload(rsx, voff + 16u);
            }
            if (ic + 1 < S) {
                ++ic;
                if (++ik == NCH) {
                    ik = 0;
                    ix0 += TW;
                    if (ix0 >= A.W) { ix0 = 0; iy0 += TH; if (iy0 >= A.H) { iy0 = 0; ++in_; } }
                }
            }
        };
        // chunk c_ of the run (register set R = c_ & 1) -> halo slot c_ & 1
        auto commit = [&](auto rc, int c_) {
            constexpr int R = decltype(rc)::value;
            int si, cc0;
            chunk_src(ck, si, cc0);
            if (c_ + 1 < S) { if (++ck == NCH) ck = 0; }
            const ConvSrc &s = A.src[si];
            const float *xf = s_xf + (si ? c0n : 0) + cc0 + slot * 8;
            float sc[8], sh[8];
            const bool on = XF == 1 || (XF == 2 && s.scale != nullptr);
            if (XF != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = xf[j]; sh[j] = xf[xfs + j]; }
            }
            const bool relu = XF == 1 || (XF == 2 && s.relu != 0);
            f32x4 rr[NA][2];
            const bool has_res = XF == 2 && s.res != nullptr;
            if (has_res) {
                const __amdgpu_buffer_rsrc_t rsr = si ? rsr1 : rsr0;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    rr[i][0] = bload(rsr, eo[R][i]);
                    rr[i][1] = bload(rsr, eo[R][i] + 16u);
                }
            }
            unsigned char *dst0 = lds_a + R * L::A_SLOT;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float v[8] = {pa[R][i][0][0], pa[R][i][0][1], pa[R][i][0][2], pa[R][i][0][3], pa[R][i][1][0], pa[R][i][1][1], pa[R][i][1][2], pa[R][i][1][3]};
                u32x4 hi, lo;
                if (XF == 1) {
                    // BatchNorm + ReLU with the zero fill folded into the clamp: med3(v, 0, lim) = max(v, 0) for lim = +inf, 0 for lim = 0
                    const float lim = __builtin_bit_cast(float, (int)eo[R][i] < 0 ? 0u : 0x7f800000u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(fmaf(v[j], sc[j], sh[j]), 0.f, lim);
                    split8(v, hi, lo);
                } else {
                    if (XF != 0 && on) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
                    }
                    if (has_res) {
                        const float r[8] = {rr[i][0][0], rr[i][0][1], rr[i][0][2], rr[i][0][3], rr[i][1][0], rr[i][1][1], rr[i][1][2], rr[i][1][3]};
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += r[j];
                    }
                    if (XF != 0 && relu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    split8(v, hi, lo);
                    if (XF != 0) {                                       // (plain sources: the zero fill arrived as zeros)
                        const unsigned keep = (int)eo[R][i] < 0 ? 0u : 0xffffffffu;      // outside the image / source: zeros (after the transform)
                        hi &= keep;
                        lo &= keep;
                    }
                }
                if (i < NA - 1 || ptid + i * 256 < NPIX * VPP) {
                    *reinterpret_cast<u32x4 *>(dst0 + doff[i]) = hi;
                    *reinterpret_cast<u32x4 *>(dst0 + L::A_PLANE + doff[i]) = lo;
                }
            }
        };
        // weight chunk wk of the tile -> weight slot (run chunk & 1) by LDS-DMA: the packed chunk (hi image | lo image) is the LDS image
        int wk = 0;
        const unsigned lds_w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds_w;
        auto dma_w = [&](int wslot) {
            constexpr int NPC = L::W_SLOT / 1024;                 // 1 KB pieces of a chunk; the four mover waves take them in turn
            static_assert(L::W_SLOT % 1024 == 0, "whole wave-instructions");
            const bool one = MIX && wk >= n0c;
            const char *wsrc = reinterpret_cast<const char *>(A.w) +
                               (MIX ? (size_t)cout_tile * ((size_t)n0c * L::W_SLOT + (size_t)(NCH - n0c) * W1_SLOT) +
                                          (one ? (size_t)n0c * L::W_SLOT + (size_t)(wk - n0c) * W1_SLOT : (size_t)wk * L::W_SLOT)
                                    : ((size_t)cout_tile * NCH + wk) * L::W_SLOT);
            if (++wk == NCH) wk = 0;
            if (A.debug & 4) return;                              // ablation: no weight DMA
            if (one) {
                if (pw < W1_SLOT / 1024) glds_piece(wsrc + pw * 1024, (unsigned)lane * 16u, lds_w_addr + wslot * L::W_SLOT + pw * 1024);
                return;
            }
#pragma unroll
            for (int i = 0; i < (NPC + 3) / 4; ++i) {
                const int pc = i * 4 + pw;
                if (i * 4 + 3 < NPC || pc < NPC) glds_piece(wsrc + pc * 1024, (unsigned)lane * 16u, lds_w_addr + wslot * L::W_SLOT + pc * 1024);
            }
        };
        // BNS: quarter q = block (mi = q >> 1, ni = q & 1) of every consumer wave of the tile at (rn, ry0, rx0) -> raw buffer q & 1; this
        // wave brings the block of consumer wave pw: four 1 KB pieces of 8 pixels x 128 bytes
        const unsigned lds_raw_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds_raw;
        const unsigned raw_lane = (unsigned)((lane >> 3) * A.Cout * 4 + (lane & 7) * 16);
        auto dma_raw = [&](int q, int rn, int ry0, int rx0) {
            const int mi = q >> 1, ni = q & 1;
            const char *rb = reinterpret_cast<const char *>(A.eres) +
                             (((size_t)(rn * A.H + ry0 + pw * 4 + mi * 2) * A.W + rx0) * A.Cout + cout0 + ni * 32) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                glds_piece(rb + ((size_t)(i >> 1) * A.W + (i & 1) * 8) * A.Cout * 4, raw_lane, lds_raw_addr + (q & 1) * L::RAW_Q + pw * 4096 + i * 1024);
        };
        // the movers' barrier by hand (conv_ws_kernel): this wave's LDS writes done, everything older than its N youngest vector-memory
        // operations - the weight DMA of this interval - landed, then the barrier; the halo requests issued after the DMA stay in flight
        auto stream_sync = [](auto n_c) {
            constexpr int N = decltype(n_c)::value;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        using NHL = std::integral_constant<int, 2 * NA>;          // halo requests of one interval (one chunk: NA vectors of two loads)
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        issue(I0{});
        issue(I1{});
        for (int c = ptid; c < ctot; c += 256) {
            const ConvSrc &Sx = c < c0n ? A.src[0] : A.src[1];
            const int cc = c < c0n ? c : c - c0n;
            s_xf[c] = Sx.scale ? Sx.scale[cc] : 1.f;
            s_xf[xfs + c] = Sx.shift ? Sx.shift[cc] : 0.f;
        }
        __syncthreads();                                         // the scale / shift table
        dma_w(0);
        commit(I0{}, 0);
        issue(I0{});
        stream_sync(NHL{});                                      // chunk 0 staged, its weights landed
        // interval k: the consumers work on chunk k (slots k & 1).  Here: the weights of chunk k+1 leave for the other weight slot first
        // (DMA: they land while the rest of the interval runs), chunk k+1 is transformed, split and written into the other halo slot,
        // chunk k+3 is requested.  The explicit wait in front of the barrier leaves the 2 NA youngest vector-memory operations - those
        // halo requests - in flight; the DMA, which is older, has landed.
        auto interval = [&](auto rc, int k) {
            constexpr int R = decltype(rc)::value;               // register set / slot of chunk k + 1
            dma_w(R);
            if (BNS) {
                // the raw quarter the consumers need in the NEXT interval (all of this is older than the halo requests below: the wait
                // in front of the barrier covers it whether or not it was issued)
                if (kt == NCH - 1) dma_raw(0, o_n, o_y0, o_x0);
                else if (kt < 3 && s_ok) dma_raw(kt + 1, s_n, s_y0, s_x0);
            }
            commit(rc, k + 1);
            asm volatile("" ::: "memory");
            issue(rc);
            stream_sync(NHL{});
            if (BNS && ++kt == NCH) {
                kt = 0;
                s_n = o_n; s_y0 = o_y0; s_x0 = o_x0; s_ok = true;
                o_x0 += TW;
                if (o_x0 >= A.W) { o_x0 = 0; o_y0 += TH; if (o_y0 >= A.H) { o_y0 = 0; ++o_n; } }
            }
        };
        for (int k = 0; k < S; k += 2) {
            interval(I1{}, k);
            interval(I0{}, k + 1);                               // (S odd: one more interval than the consumers need; they add a barrier)
        }
        __syncthreads();                                         // (the consumers' statistics barrier of the last tile)
        return;
    }

    // ================================ consumers ================================
    __syncthreads();                                             // (the movers' table barrier)
    const int wm = wave;
    const int half = lane >> 5, l31 = lane & 31;
    int abase[MPW][2];                                           // [.][parity of the tap's row offset]
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int m = (wm * MPW + mi) * 32 + l31;
#pragma unroll
        for (int par = 0; par < 2; ++par) abase[mi][par] = ((m / TW) * HW_ + m % TW) * PSTR + ((half ^ ((m / TW + par) & 1)) * 16);
    }
    auto tpar = [](int t) { return (t / 3) & 1; };
    auto toff = [](int t) { return ((t / 3) * HW_ + (t % 3)) * PSTR; };
    const int bbase = half * BN * 16 + l31 * 16;
    f32x16 accA[MPW][NPW], accB[MPW][NPW];
    // The finished tile leaves straight from the accumulators, in the gaps between the next tile's MFMAs: register r of block (mi, ni)
    // is cout ni * 32 + l31 of pixel 4 half + (r & 3) + 8 (r >> 2) of the block's two tile rows - the 32 lanes of a half write one
    // 128-byte line of a pixel, a wave instruction two of them.  Everything but 4 half pixels + l31 couts is wave-uniform: a scalar base
    // per store, one per-lane offset register.
    const unsigned pix_b = (unsigned)A.out_cstride * 4u;
    const unsigned l_off = (unsigned)(4 * half) * pix_b + (unsigned)l31 * 4u;
    char *const out_b = reinterpret_cast<char *>(reinterpret_cast<float *>(A.out) + A.out_coff + cout0);
    int p_n, p_y0, p_x0;                                         // the finished tile (whose epilogue rides in the current one)
    {
        p_n = t_lo / tiles_img;
        const int r = t_lo - p_n * tiles_img, ty = r / tiles_x;
        p_y0 = ty * TH; p_x0 = (r - ty * tiles_x) * TW;
    }
    float e_osc[NPW], e_osh[NPW];
    bool e_ok[NPW];
#pragma unroll
    for (int ni = 0; ni < NPW; ++ni) {
        const int co = cout0 + ni * 32 + l31;
        const bool cok = co < A.Cout;
        e_ok[ni] = cok && !(A.debug & 8);                        // (8: ablation, nothing is stored)
        e_osc[ni] = (!BNS && A.oscale && cok) ? A.oscale[co] : 1.f;
        e_osh[ni] = BNS ? 0.f : fmaf((A.bias && cok) ? A.bias[co] : 0.f, e_osc[ni], (A.oshift && cok) ? A.oshift[co] : 0.f);
    }
    const bool orelu = A.orelu != 0;
    // BNS: the producer layer's BatchNorm scale | shift | mean | invstd of this lane's couts, its running sums, the raw value in flight
    float b_sc[NPW], b_sh[NPW], b_mu[NPW], b_is[NPW], b_s1[NPW], b_s2[NPW], b_x = 0.f;
    if (BNS) {
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) {
            const int co = cout0 + ni * 32 + l31;
            b_sc[ni] = A.oscale[co]; b_sh[ni] = A.oshift[co]; b_mu[ni] = A.eres_scale[co]; b_is[ni] = A.eres_shift[co];
            b_s1[ni] = 0.f; b_s2[ni] = 0.f;
            e_osc[ni] = 1.f; e_osh[ni] = 0.f;
        }
    }
    const unsigned char *raw_w = lds_raw + wave * 4096 + (4 * half) * 128 + l31 * 4;      // this lane inside its wave's raw block
    // raw value of element i (0..15) of the quarter in buffer `buf`: pixel (r & 3) + 8 (r >> 2) (+ 4 half) of the block, 128 bytes per pixel
    auto bns_read = [&](int buf, int r) { b_x = *reinterpret_cast<const float *>(raw_w + buf * L::RAW_Q + ((r & 3) + 8 * (r >> 2)) * 128); };
    auto bns_acc = [&](const f32x16 (&P)[MPW][NPW], int e, float x) {
        const int b_ = e / 16, r = e % 16, mi = b_ / NPW, ni = b_ % NPW;
        const float act = fmaf(x, b_sc[ni], b_sh[ni]);
        const float dz = act > 0.f ? P[mi][ni][r] : 0.f;          // (bn_bwd_flat32_kernel's arithmetic)
        b_s1[ni] += dz;
        b_s2[ni] = fmaf(dz, (x - b_mu[ni]) * b_is[ni], b_s2[ni]);
    };

    // epilogue units of a finished accumulator set, four registers each.  Statistics: blocks in the order (ni, mi).  Image: bias / scale /
    // shift / ReLU in fp32, one ds_write_b32 per value (the 32 lanes of a half write 32 consecutive couts of one pixel).
    float st_sum = 0.f, st_sq = 0.f;
    // statistics of element s of a finished set - blocks in the order (ni, mi), registers ascending; two VALU instructions
    auto stat_elem = [&](const f32x16 (&P)[MPW][NPW], int s_, int par) {
        const int ni = s_ / (MPW * 16), mi = (s_ / 16) % MPW, r = s_ % 16;
        if (mi == 0 && r == 0) { st_sum = 0.f; st_sq = 0.f; }
        const float v = P[mi][ni][r];
        st_sum += v;
        st_sq = fmaf(v, v, st_sq);
        if (mi == MPW - 1 && r == 15) {
            float *sp = s_stats + par * (4 * 2 * BN);
            const float a = st_sum + __shfl_xor(st_sum, 32), b2 = st_sq + __shfl_xor(st_sq, 32);
            if (half == 0) {
                sp[(wave * 2 + 0) * BN + ni * 32 + l31] = a;
                sp[(wave * 2 + 1) * BN + ni * 32 + l31] = b2;
            }
        }
    };
    // element e of a finished set leaves: register r of block (mi, ni) (blocks in the order (mi, ni)) = cout ni * 32 + l31 of the pixel in
    // tile row 4 wm + 2 mi + (r >> 3), column ((r >> 2) & 1) * 8 + (r & 3) + 4 half.  Scalar row base (computed once per tile), one
    // vector add for the column, bias / scale / shift / ReLU in fp32, one store: four instructions in one MFMA gap.
    unsigned s_col[8];                                           // column c of a 8-pixel group: c * pix_b (kernel constants, scalar)
#pragma unroll
    for (int c = 0; c < 8; ++c) s_col[c] = (unsigned)(c & 3) * pix_b + (unsigned)(c >> 2) * 8u * pix_b;
    char *row_base[4];                                           // rows 4 wm .. 4 wm + 3 of the finished tile, first column
    // POOL: rows 2 wm, 2 wm + 1 of the pooled tile (dense [N][H / 2][W / 2][Cout] fp32), the lane's offset inside a pooled row
    const unsigned ppix_b = (unsigned)A.Cout * 4u;
    const unsigned pl_off = (unsigned)(2 * half) * ppix_b + (unsigned)l31 * 4u;
    char *const pool_b = POOL ? reinterpret_cast<char *>(reinterpret_cast<float *>(A.pool_out) + cout0) : nullptr;
    char *prow_base[2] = {nullptr, nullptr};
    float pl_t0 = 0.f, pl_t1 = 0.f;
    auto set_row_bases = [&]() {
        char *t0 = out_b + ((size_t)(p_n * A.H + p_y0 + wm * 4) * A.W + p_x0) * pix_b;
#pragma unroll
        for (int r = 0; r < 4; ++r) row_base[r] = t0 + (size_t)r * A.W * pix_b;
        if (POOL) {
            char *q0 = pool_b + ((size_t)(p_n * (A.H >> 1) + ((p_y0 + wm * 4) >> 1)) * (A.W >> 1) + (p_x0 >> 1)) * ppix_b;
            prow_base[0] = q0;
            prow_base[1] = q0 + (size_t)(A.W >> 1) * ppix_b;
        }
    };
    auto img_elem = [&](const f32x16 (&P)[MPW][NPW], int e) {
        const int b_ = e / 16, r = e % 16, mi = b_ / NPW, ni = b_ % NPW;
        float v = fmaf(P[mi][ni][r], e_osc[ni], e_osh[ni]);
        if (orelu) v = fmaxf(v, 0.f);
        if (e_ok[ni]) {
            // (scalar base first, kept opaque: otherwise the 16 column + lane offsets - kernel constants - are hoisted into 16 vector registers)
            char *sb = row_base[mi * 2 + (r >> 3)] + (s_col[((r >> 2) & 1) * 4 + (r & 3)] + (unsigned)(ni * 128));
            asm volatile("" : "+s"(sb));
            // (global_, not flat_: the asm hid the pointer's origin; uniform base + 32-bit lane offset = the store's saddr form)
            __builtin_nontemporal_store(v, (__attribute__((address_space(1))) float *)((__attribute__((address_space(1))) char *)sb + l_off));
        }
    };
    // pooled unit u = (block, window k) of a finished set in three pieces (an MFMA gap takes ~5 instructions): the maxima of the window's
    // upper and lower pixel pair after the epilogue affine, then ReLU + one store.  Window k: registers 2k, 2k + 1, 2k + 8, 2k + 9 =
    // pooled row mi of this wave's two, pooled column ((2k >> 2) & 1) * 4 + ((2k & 3) >> 1) (+ 2 half: the lane's offset)
    auto pool_sub = [&](const f32x16 (&P)[MPW][NPW], int u, int sub) {
        const int b_ = u / 4, k = u % 4, mi = b_ / NPW, ni = b_ % NPW, r0 = 2 * k;
        if (sub == 0) pl_t0 = fmaxf(fmaf(P[mi][ni][r0], e_osc[ni], e_osh[ni]), fmaf(P[mi][ni][r0 + 1], e_osc[ni], e_osh[ni]));
        else if (sub == 1) pl_t1 = fmaxf(fmaf(P[mi][ni][r0 + 8], e_osc[ni], e_osh[ni]), fmaf(P[mi][ni][r0 + 9], e_osc[ni], e_osh[ni]));
        else {
            float v = fmaxf(pl_t0, pl_t1);
            if (orelu) v = fmaxf(v, 0.f);
            if (e_ok[ni]) {
                char *sb = prow_base[mi] + ((unsigned)(((r0 >> 2) & 1) * 4 + ((r0 & 3) >> 1)) * ppix_b + (unsigned)(ni * 128));
                asm volatile("" : "+s"(sb));
                *(__attribute__((address_space(1))) float *)((__attribute__((address_space(1))) char *)sb + pl_off) = v;
            }
        }
    };
    constexpr int NEL = MPW * NPW * 16;                          // elements of a set per lane; a quarter of them leaves per chunk interval

    int stats_tile = -1, stats_par = 0;
    auto flush_stats = [&]() {
        if (STATS && stats_tile >= 0 && tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            const float *sp = s_stats + stats_par * (4 * 2 * BN);
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += sp[(w4 * 2 + which) * BN + col];
            const int co = cout0 + col;
            if (co < A.Cout) A.stats[((size_t)stats_tile * 2 + which) * A.Cout + co] = v;
        }
        stats_tile = -1;
    };

    // one chunk interval on accumulator set C (halo / weight slots `sl`): 9 taps x MPW x NPW products of three MFMAs; the fragments of a
    // tap are requested one tap (12 MFMAs) ahead.  FIRST: the tile's first chunk starts from zero.  EPI 1: the statistics and the first
    // M block of the finished set P ride behind the MFMAs (every fourth gap one unit), EPI 2: its second M block.
    auto interval_nt = [&](auto nt_c, auto first_c, auto epi_c, f32x16 (&C)[MPW][NPW], const f32x16 (&P)[MPW][NPW], int par, int sl) {
        constexpr int NT = decltype(nt_c)::value;                 // taps of this chunk: 9, or 1 (MIX: the centre tap)
        constexpr int BPL = NT * CK * BN * 2;                     // one image of its packed weights
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int EPI = decltype(epi_c)::value;
        static_assert(NT == 9 || (EPI == 0 && !FIRST), "one-tap chunks come after the nine-tap ones");
        if (A.debug & 1) return;                                  // ablation (tools/bench_conv_ws32.py): no fragment reads, no MFMAs
        bf16x8 ah[2][MPW], al[2][MPW], bh[2][NPW], bl[2][NPW];
        const unsigned char *la = lds_a + sl * L::A_SLOT, *lw = lds_w + sl * L::W_SLOT;
        // fragment read number i of tap t, in the order tap t's MFMAs use them: al0 bh0 ah0 bl0 | bh1 bl1 | al1 ah1
        constexpr int NRD = 2 * MPW + 2 * NPW;
        auto request_one = [&](int t, int i) {
            const int s2 = t & 1;
            const int tq = NT == 9 ? t : 4;                       // (the centre tap's halo offset and row parity)
            const int ao = toff(tq), bo = bbase + t * 2 * BN * 16;
            if (i == 0) al[s2][0] = *reinterpret_cast<const bf16x8 *>(la + L::A_PLANE + abase[0][tpar(tq)] + ao);
            else if (i == 1) bh[s2][0] = *reinterpret_cast<const bf16x8 *>(lw + bo);
            else if (i == 2) ah[s2][0] = *reinterpret_cast<const bf16x8 *>(la + abase[0][tpar(tq)] + ao);
            else if (i == 3) bl[s2][0] = *reinterpret_cast<const bf16x8 *>(lw + BPL + bo);
            else if (NPW == 2 && i == 4) bh[s2][NPW - 1] = *reinterpret_cast<const bf16x8 *>(lw + bo + 512);
            else if (NPW == 2 && i == 5) bl[s2][NPW - 1] = *reinterpret_cast<const bf16x8 *>(lw + BPL + bo + 512);
            else if (i == NRD - 2) al[s2][1] = *reinterpret_cast<const bf16x8 *>(la + L::A_PLANE + abase[1][tpar(tq)] + ao);
            else if (i == NRD - 1) ah[s2][1] = *reinterpret_cast<const bf16x8 *>(la + abase[1][tpar(tq)] + ao);
        };
        // after MFMA number g of the interval (program order is issue order: a scheduling fence after every MFMA keeps the next tap's
        // fragment reads - one per MFMA gap, a whole tap ahead of their use - and the epilogue units where they are written; left to
        // itself the scheduler sinks every read to just in front of its first use and the matrix pipe waits out the LDS latency tap by tap)
        auto gap = [&](int t, int m, int g) {
            if (t + 1 < NT && m < NRD) request_one(t + 1, m);
            if (EPI != 0) {
                // interval EPI of the tile (1..4): a quarter of the finished set's elements leave, one store (4 instructions) every sixth
                // gap; the statistics ride in the first interval, two elements (4 instructions) in every third gap.  More than ~5 issued
                // instructions in one gap hold up the next MFMA (a whole 4-store unit with its addresses in one gap cost 17 % of the kernel).
                // Tiles of fewer than four chunks (the 16- / 32- / 48-channel layers: HBM-bound, the matrix pipe has the slack): EPI 5 = the
                // whole finished tile in ONE interval (statistics as in EPI 1, a store in each of the other two gaps of three), EPI 6 / 7 =
                // half of it each (a store in every third gap).
                constexpr int QEL = NEL / 4;
                if ((EPI == 1 || EPI == 5 || EPI == 6) && STATS && g % 3 == 1 && 2 * (g / 3) + 1 < NEL) { stat_elem(P, 2 * (g / 3), par); stat_elem(P, 2 * (g / 3) + 1, par); }
                if (EPI <= 4 && g % 6 == 3 && g / 6 < QEL) img_elem(P, (EPI - 1) * QEL + g / 6);
                // POOL: the quarter's block has four windows, three pieces each, in the gaps 5, 11, .. 71
                if (POOL && EPI <= 4 && g % 6 == 5 && g / 6 < 12) pool_sub(P, (EPI - 1) * 4 + (g / 6) / 3, (g / 6) % 3);
                if (EPI == 5 && g % 3 != 1 && 2 * (g / 3) + (g % 3 == 2 ? 1 : 0) < NEL) img_elem(P, 2 * (g / 3) + (g % 3 == 2 ? 1 : 0));
                if ((EPI == 6 || EPI == 7) && g % 3 == 0 && g / 3 < NEL / 2) img_elem(P, (EPI - 6) * (NEL / 2) + g / 3);
                if (BNS && EPI <= 4) {
                    // element i of the quarter (= block EPI - 1 of the finished set; raw buffer (EPI - 1) & 1): read in gap 1 (i = 0) or
                    // beside the sums of element i - 1, used six gaps later
                    if (g == 1) bns_read((EPI - 1) & 1, 0);
                    if (g % 6 == 0 && g >= 6 && g / 6 - 1 < QEL) {
                        const int i = g / 6 - 1;
                        const float x = b_x;
                        if (i + 1 < QEL) bns_read((EPI - 1) & 1, i + 1);
                        bns_acc(P, (EPI - 1) * QEL + i, x);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(NT != 9 || (3 * (NEL / 2) <= TAPS * MPW * NPW * 3 && 6 * (NEL / 4) <= TAPS * MPW * NPW * 3 && 2 * (TAPS * MPW * NPW) >= NEL),
                      "the deferred epilogue fits the MFMA gaps of an interval");
#pragma unroll
        for (int i = 0; i < NRD; ++i) request_one(0, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni) {
                    const int m0 = (mi * NPW + ni) * 3, g0 = t * MPW * NPW * 3 + m0;
                    // small terms first (conv_f32_kernel's order)
                    if (FIRST && t == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t & 1][mi], bh[t & 1][ni], z, 0, 0, 0);
                    } else {
                        C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t & 1][mi], bh[t & 1][ni], C[mi][ni], 0, 0, 0);
                    }
                    gap(t, m0, g0);
                    C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1][mi], bl[t & 1][ni], C[mi][ni], 0, 0, 0);
                    gap(t, m0 + 1, g0 + 1);
                    C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1][mi], bh[t & 1][ni], C[mi][ni], 0, 0, 0);
                    gap(t, m0 + 2, g0 + 2);
                }
        }
    };
    auto interval = [&](auto first_c, auto epi_c, f32x16 (&C)[MPW][NPW], const f32x16 (&P)[MPW][NPW], int par, int sl) {
        interval_nt(std::integral_constant<int, 9>{}, first_c, epi_c, C, P, par, sl);
    };
    using F_ = std::false_type;
    using T_ = std::true_type;
    using E0 = std::integral_constant<int, 0>;
    using E1 = std::integral_constant<int, 1>;
    using E2 = std::integral_constant<int, 2>;
    using E3 = std::integral_constant<int, 3>;
    using E4 = std::integral_constant<int, 4>;
    using E5 = std::integral_constant<int, 5>;
    using E6 = std::integral_constant<int, 6>;
    using E7 = std::integral_constant<int, 7>;
    // tile j (run chunks q .. q + NCH - 1) on set C; P = the finished tile j - 1 (at p_*): its statistics ride in chunk interval 0, a quarter
    // of its stores in each of the intervals 0 .. 3
    auto tile_step = [&](auto has_prev, f32x16 (&C)[MPW][NPW], const f32x16 (&P)[MPW][NPW], int j, int q) {
        constexpr bool HP = decltype(has_prev)::value;
        const int par = (j + 1) & 1;
        if (HP) set_row_bases();
        if constexpr (NCS != 0) {
            // one to three chunks per tile: the finished tile leaves in the first interval (one chunk) or the first two
            if constexpr (NCS == 1) { if (HP) interval(T_{}, E5{}, C, P, par, q & 1); else interval(T_{}, E0{}, C, P, par, q & 1); }
            else { if (HP) interval(T_{}, E6{}, C, P, par, q & 1); else interval(T_{}, E0{}, C, P, par, q & 1); }
            if (HP && STATS) { stats_tile = t_lo + j - 1; stats_par = par; }
            __syncthreads();
            flush_stats();
            if constexpr (NCS == 2) {
                if (HP) interval(F_{}, E7{}, C, P, par, (q + 1) & 1); else interval(F_{}, E0{}, C, P, par, (q + 1) & 1);
                __syncthreads();
                if (NCH == 3) {
                    interval(F_{}, E0{}, C, P, par, (q + 2) & 1);
                    __syncthreads();
                }
            }
        } else {
            if (HP) interval(T_{}, E1{}, C, P, par, q & 1); else interval(T_{}, E0{}, C, P, par, q & 1);
            if (HP && STATS) { stats_tile = t_lo + j - 1; stats_par = par; }
            __syncthreads();
            flush_stats();
            if (HP) interval(F_{}, E2{}, C, P, par, (q + 1) & 1); else interval(F_{}, E0{}, C, P, par, (q + 1) & 1);
            __syncthreads();
            if (HP) interval(F_{}, E3{}, C, P, par, (q + 2) & 1); else interval(F_{}, E0{}, C, P, par, (q + 2) & 1);
            __syncthreads();
            if (HP) interval(F_{}, E4{}, C, P, par, (q + 3) & 1); else interval(F_{}, E0{}, C, P, par, (q + 3) & 1);
            __syncthreads();
            for (int i = 4; i < NCH; ++i) {
                if (MIX && i >= n0c) interval_nt(std::integral_constant<int, 1>{}, F_{}, E0{}, C, P, par, (q + i) & 1);
                else interval(F_{}, E0{}, C, P, par, (q + i) & 1);
                __syncthreads();
            }
        }
        if (HP) {                                                // P has left: the finished tile is now the one just accumulated
            p_x0 += TW;
            if (p_x0 >= A.W) { p_x0 = 0; p_y0 += TH; if (p_y0 >= A.H) { p_y0 = 0; ++p_n; } }
        }
    };
    // the last tile of the run: nothing left to hide behind
    auto serial_epilogue = [&](const f32x16 (&P)[MPW][NPW], int j) {
        const int par = j & 1;
        if (STATS) {
#pragma unroll
            for (int e = 0; e < NEL; ++e) stat_elem(P, e, par);
            stats_tile = t_lo + j;
            stats_par = par;
        }
        set_row_bases();
#pragma unroll
        for (int e = 0; e < NEL; ++e) img_elem(P, e);
        if (POOL) {
#pragma unroll
            for (int u = 0; u < NEL / 4; ++u) { pool_sub(P, u, 0); pool_sub(P, u, 1); pool_sub(P, u, 2); }
        }
        if (BNS) {
            // the last tile of the run: its raw values straight from global memory (accumulator layout: 128-byte lines), then this wave's
            // partial row - the two lane halves hold the same couts
            const float *rw = reinterpret_cast<const float *>(A.eres) + ((size_t)(p_n * A.H + p_y0 + wm * 4) * A.W + p_x0 + 4 * half) * A.Cout + cout0 + l31;
#pragma unroll
            for (int b_ = 0; b_ < NEL / 16; ++b_) {               // (a block at a time: 16 values in flight)
                const int mi = b_ / NPW, ni = b_ % NPW;
                float xs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) xs[r] = rw[((size_t)(mi * 2 + (r >> 3)) * A.W + ((r >> 2) & 1) * 8 + (r & 3)) * A.Cout + ni * 32];
#pragma unroll
                for (int r = 0; r < 16; ++r) bns_acc(P, b_ * 16 + r, xs[r]);
            }
            float *row = A.stats + (size_t)(blockIdx.x * 4 + wave) * 2 * A.Cout + cout0 + l31;
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                const float a = b_s1[ni] + __shfl_xor(b_s1[ni], 32), b2 = b_s2[ni] + __shfl_xor(b_s2[ni], 32);
                if (half == 0) { row[ni * 32] = a; row[A.Cout + ni * 32] = b2; }
            }
        }
        __syncthreads();                                         // the statistics of the four waves are parked
        flush_stats();
    };

    __syncthreads();                                             // chunk 0 is staged
    tile_step(F_{}, accA, accB, 0, 0);
    int j = 1, q = NCH;
    for (; j + 1 < ntl; j += 2, q += 2 * NCH) {
        tile_step(T_{}, accB, accA, j, q);
        tile_step(T_{}, accA, accB, j + 1, q + NCH);
    }
    const bool tail = j < ntl;
    if (tail) tile_step(T_{}, accB, accA, j, q);
    if (S & 1) __syncthreads();                                  // the movers' loop runs whole pairs of intervals
    if (tail) serial_epilogue(accB, j); else serial_epilogue(accA, ntl - 1);
}

}  // namespace

namespace cdnet {

// eligibility + launch; returns -1 when the layer must take conv_f32_kernel
template <int BN>
static int try_launch_ws32(const ConvArgs &A, hipStream_t st, bool dry_run) {
    using L = Ws32Lds<BN>;
    int ctot = 0;
    for (int i = 0; i < A.nsrc; ++i) {
        if (A.src[i].pool) return -1;
        ctot += A.src[i].C;
        // the movers' requests: 31-bit byte offsets from the source's base (bit 31 marks a zero-fill vector)
        const long long rs = A.src[i].row_stride ? A.src[i].row_stride : (long long)A.src[i].Ws * A.src[i].C;
        if ((long long)A.N * A.src[i].Hs * rs * 4 >= (1LL << 31)) return -1;
    }
    const bool bns = A.ws == 2;
    const bool mix = A.taps1 == 1 && A.nsrc == 2;
    if (mix && (bns || A.stats || A.src[0].C < 64)) return -1;
    if (bns && (BN != 64 || A.Cout % 64 != 0 || !A.eres || !A.oscale || !A.oshift || !A.eres_scale || !A.eres_shift || !A.stats || A.bias ||
                A.orelu || A.out_cstride != A.Cout || A.out_coff)) return -1;
    const int smem = L::bytes(ctot, bns);
    if (smem > 160 * 1024) return -1;
    const int T = (A.W / 16) * (A.H / 16) * A.N;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return check_launch("hipGetDeviceProperties");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ctiles = cdiv(A.Cout, BN);
    const int Gmax = n_cu / ctiles > 0 ? n_cu / ctiles : 1;
    // a persistent workgroup pays a few microseconds of start-up and a serial epilogue for its last tile: worth it from a few tiles'
    // worth of chunk intervals per workgroup on
    if (!(A.debug & 64) && (long long)T * A.nchunk < 16LL * Gmax) return -1;
    bool all_plain = true, all_fast = true;
    for (int i = 0; i < A.nsrc; ++i) {
        const ConvSrc &s = A.src[i];
        all_plain = all_plain && !s.scale && !s.relu && !s.res;
        all_fast = all_fast && s.scale && s.shift && s.relu == 1 && !s.res;
    }
    int G = n_cu / ctiles;
    G = G > T ? T : G;
    if (bns && G > 256) G = 256;                                  // ws == 2 writes 4 * G partial rows into the caller's CDNET_BNS_PARTIAL_ROWS = 1024
    if (G >= 8) G &= ~7;
    if ((A.debug >> 8) > 0 && (A.debug >> 8) < G) G = A.debug >> 8;      // tests: few workgroups, long runs of tiles
    if (G < 1) G = 1;
    dim3 grid(G, ctiles, 1);
    auto launch_n = [&](auto xf_c, auto st_c, auto ncs_c) -> int {
        constexpr int XF = decltype(xf_c)::value;
        constexpr bool STATS = decltype(st_c)::value;
        constexpr int NCS = decltype(ncs_c)::value;
        auto kern = conv_ws32_kernel<BN, XF, STATS, false, NCS>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return check_launch("hipFuncSetAttribute(conv_ws32)");
            attr_done = true;
        }
        kern<<<grid, 512, smem, st>>>(A);
        return check_launch("conv_ws32_kernel");
    };
    auto launch = [&](auto xf_c, auto st_c) -> int {
        // tiles of one / two or three chunks (16- to 48-channel inputs): their own instantiations, and only for the source forms that
        // occur there (plain or BatchNorm + ReLU) with 64 output channels per workgroup
        constexpr int XF = decltype(xf_c)::value;
        if (A.nchunk >= 4) return launch_n(xf_c, st_c, std::integral_constant<int, 0>{});
        if constexpr (BN == 64 && XF != 2) {
            if (A.nchunk == 1) return launch_n(xf_c, st_c, std::integral_constant<int, 1>{});
            return launch_n(xf_c, st_c, std::integral_constant<int, 2>{});
        }
        return -1;
    };
    using X0 = std::integral_constant<int, 0>;
    using X1 = std::integral_constant<int, 1>;
    using X2 = std::integral_constant<int, 2>;
    const int xf = all_plain ? 0 : (all_fast ? 1 : 2);
    if (A.nchunk < 4 && (BN != 64 || xf == 2)) return -1;          // (no small-tile instantiation for these)
    const bool pool = A.pool_out != nullptr;
    if (pool && (BN != 64 || bns || mix || A.stats || A.nchunk < 4 || xf != 0 || !A.orelu || A.out_coff || A.out_cstride != A.Cout)) return -1;
    if (dry_run) return CDNET_OK;
    if (pool) {
        if constexpr (BN == 64) {
            auto kern = conv_ws32_kernel<64, 0, false, false, 0, false, true>;
            static bool attr_done = false;
            if (!attr_done) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return check_launch("hipFuncSetAttribute(conv_ws32 pool)");
                attr_done = true;
            }
            kern<<<grid, 512, smem, st>>>(A);
            return check_launch("conv_ws32_kernel(pool)");
        }
        return -1;
    }
    if (bns) {
        if constexpr (BN == 64) {
            auto launch_bns = [&](auto xf_c) -> int {
                constexpr int XF = decltype(xf_c)::value;
                auto kern = conv_ws32_kernel<64, XF, false, true>;
                static bool attr_done = false;
                if (!attr_done) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                        return check_launch("hipFuncSetAttribute(conv_ws32 bns)");
                    attr_done = true;
                }
                kern<<<grid, 512, smem, st>>>(A);
                return check_launch("conv_ws32_kernel(bns)");
            };
            return xf == 0 ? launch_bns(X0{}) : (xf == 1 ? launch_bns(X1{}) : launch_bns(X2{}));
        }
        return -1;
    }
    if (mix) {
        auto launch_mix = [&](auto xf_c) -> int {
            constexpr int XF = decltype(xf_c)::value;
            auto kern = conv_ws32_kernel<BN, XF, false, false, 0, true>;
            static bool attr_done = false;
            if (!attr_done) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return check_launch("hipFuncSetAttribute(conv_ws32 mix)");
                attr_done = true;
            }
            kern<<<grid, 512, smem, st>>>(A);
            return check_launch("conv_ws32_kernel(mix)");
        };
        return xf == 0 ? launch_mix(X0{}) : launch_mix(X2{});
    }
    if (A.stats) return xf == 0 ? launch(X0{}, std::true_type{}) : (xf == 1 ? launch(X1{}, std::true_type{}) : launch(X2{}, std::true_type{}));
    return xf == 0 ? launch(X0{}, std::false_type{}) : (xf == 1 ? launch(X1{}, std::false_type{}) : launch(X2{}, std::false_type{}));
}

// called by conv_forward_f32 first; -1 = not eligible (the caller falls back to conv_f32_kernel)
int conv_forward_f32_ws(const ConvArgs &A, hipStream_t st, bool dry_run) {
    if (A.debug & 32) return -1;
    if (A.taps != 9 || A.npar != 1 || A.ostride != 1 || A.tile != 16 || A.CK != 16 || (A.eres && A.ws != 2) || (A.ws && A.ws != 2)) return -1;
    if (A.taps1 != 0 && A.taps1 != 9 && !(A.taps1 == 1 && A.nsrc == 2)) return -1;
    if (A.H % 16 != 0 || A.W % 16 != 0 || A.nchunk < (A.ws == 2 ? 4 : 1)) return -1;
    if (A.BN == 64) return try_launch_ws32<64>(A, st, dry_run);
    if (A.BN == 32) return try_launch_ws32<32>(A, st, dry_run);
    return -1;
}

}  // namespace cdnet
