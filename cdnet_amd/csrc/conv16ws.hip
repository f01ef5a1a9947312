// Wave-specialised, persistent 3x3 convolution of the 16-bit path for every launch that carries no BatchNorm statistics: the eval-mode
// forward (inference tiles, sliding windows), every backward-data pass of training, the space-to-depth backward of the transposed
// convolutions.  Round 4: the recipe of conv_ws32_kernel (conv32ws.hip) for bf16 operands -
//
//   * OPERAND ROLES SWAPPED against conv_fwd_kernel / conv_ws_kernel: the packed weights are the MFMA's A operand (M = 32 output channels),
//     the halo pixels its B operand (N = 32 pixels).  An accumulator block then holds, per lane, ONE pixel x 16 output channels, and with
//     the weight rows read through the permutation sigma (bits 2 and 3 of the row swapped) registers 8s..8s+7 of a lane are 8 CONSECUTIVE
//     output channels: 16 bytes of the NHWC output per lane without a transpose.  The packed weights in memory are the ones every other
//     kernel reads (the permutation is in the fragment address);
//   * bias + folded BatchNorm shift enter as the accumulators' initial value (the first MFMA of a tile takes them as its C operand):
//     the epilogue is convert, ReLU on the packed 16-bit patterns, store - 10 instructions per 8 outputs, spread over the MFMA gaps of
//     the next tile (the eval-mode BatchNorm scale is folded into the packed weights by the host, runtime.ConvLayer.eval_pack);
//   * any number of 16-channel chunks from one up (the stem 16 -> 64, the decoder's 80 -> 16 and 160 -> 32, the 512-channel loops):
//     one barrier per TWO chunks of the workgroup's run of chunks, wherever tile boundaries fall; weights resident in LDS when the tile's
//     chunks fit (<= 82 KB), else streamed through a four-slot ring by LDS-DMA two chunks ahead (conv_ws_kernel's scheme);
//   * a second source may carry ONE tap per chunk instead of nine (cdnet_conv_args.taps1 = 1): the 1x1 branch of a residual unit then is
//     four more K steps of its second 3x3 convolution - relu(bn2(conv2(h)) + conv_1x1(x)) in one launch, eval mode
//     (model_unet_rev1.py:161-170), no stored conv2 output, no separate 1x1 pass;
//   * tiles of an even number (>= 4) of chunks - the heavy layers - leave through an LDS OUT IMAGE [pixel][channel] (16-byte writes of the
//     consumers, conflict-free 144-byte rows) that the MOVERS store one interval later as whole 128-byte lines (OUT): a store issued by a
//     consumer between MFMAs waits for the memory pipeline behind the movers' requests, and its 32-byte pieces of a line cost the write
//     path as much as whole lines.  There every store is the movers' (the last tile of a run goes through the image too), and they can
//     pool: nn.MaxPool2d(2, 2) of the block's two rows beside the stores (cdnet_conv_args.pool_out).  The other tile shapes (1 - 3
//     chunks, odd counts) store straight from the accumulators, a pixel block's lines in consecutive gaps;
//   * the streamed-weight launches (128+ input channels: matrix-bound) run v_mfma_f32_16x16x32_bf16 (K32): the chip holds a higher clock
//     on it (tools/micro/mfma_shapes.hip);
//   * waves 4..7 (movers) are conv_ws_kernel's: four halo chunks in flight in registers, requests through buffer descriptors (zero fill by
//     the range check), generic source transform, four-slot halo ring.
// Accumulation order per output (chunk, tap, k) is conv_fwd_kernel's in the 32x32x16 forms: without bias the outputs are bit-identical to
// it (tests/test_gpu_conv16ws.py); with bias the sum starts from the bias instead of ending with it (one rounding moved); K32 adds 32
// products per MFMA: one bf16 ulp.
//
// Replaces the same reference lines as conv.hip: models/dam/model_unet_rev1.py:86-170, 244-287.
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include "xform.h"
#include "stage16.h"
#include <stdlib.h>

using namespace cdnet;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#ifdef CDNET_WS_STAMPS
// debug build only (CDNET_HIPCC_FLAGS=-DCDNET_WS_STAMPS, tools/ws16_stamps.py): wall-clock stamps (100 MHz) of consumer wave 0 and mover wave 4
// of ONE workgroup over a window of barrier intervals in the middle of its run, parked in LDS (6 KB behind the kernel's own) and dumped at the
// end; the production build carries none of it
__device__ unsigned long long g_ws16_stamps[3 * 1024];      // consumer stamps from 0, mover stamps from 1024 (255 each + a zero; the third region served the storer waves of a removed experiment)
extern "C" __attribute__((visibility("default"))) int cdnet_debug_ws16_stamps(unsigned long long *dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws16_stamps), sizeof(g_ws16_stamps)) == hipSuccess ? 0 : 1;
}
#define W16_STAMP(id) do { if (stamp_on && stamp_iv >= stamp_i0 && sn < 255) { s_stamp[sn++] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(id); } } while (0)
#define W16_STAMP_NEXT() do { ++stamp_iv; } while (0)
#define W16_STAMP_BYTES (3 * 256 * 8)
#else
#define W16_STAMP(id) do { } while (0)
#define W16_STAMP_NEXT() do { } while (0)
#define W16_STAMP_BYTES 0
#endif

namespace {

template <int BN>
struct W16Lds {
    static constexpr int TH = 16, TW = 16, CK = 16;
    static constexpr int PSTR = 32;                               // bytes per halo pixel (16 bf16), k-halves swizzled by the halo row's parity
    static constexpr int NPIX = (TH + 2) * (TW + 2);
    static constexpr int A_IMG = NPIX * PSTR;                     // a ring slot: the halo image of a chunk, 10 368 B
    static constexpr int OROW = BN * 2 + 16;                      // out image: bytes per pixel (padded: conflict-free 16-byte writes of 8 consecutive pixels)
    static constexpr int OBLK = 32 * OROW;                        // a consumer wave's 32-pixel block
    static constexpr int OHALF = 4 * OBLK;                        // pixel block pi of the four consumer waves
    static constexpr int OUT_BYTES = 2 * OHALF;
    static constexpr int WCH9 = 9 * CK * BN * 2;                  // packed weights of a nine-tap chunk
    static constexpr int WCH1 = CK * BN * 2;                      // ... of a one-tap chunk
    static constexpr int RES_MAX = 82 * 1024;                     // resident weights up to this many bytes, else the ring
    __host__ __device__ static int wbytes(int n9, int n1) { return n9 * WCH9 + n1 * WCH1; }
    __host__ __device__ static int bytes(int ns, int wb, int ctot, bool out = false) { return ns * A_IMG + wb + 2 * ((ctot + 7) / 8 * 8) * 4 + (out ? OUT_BYTES : 0); }
};

// sigma: the weight row lane m of an A fragment reads (an involution: bits 2 and 3 swapped).  D row m = (r & 3) + 8 (r >> 2) + 4 half then
// holds output channel 16 (r >> 3) + 8 half + (r & 7) of the block: 8 consecutive channels per 8 registers, the two halves adjacent.
__device__ __forceinline__ int sigma32(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// XF: 0 every source plain bf16, 2 anything (run-time flags: fp16 storage, scale / shift, residual, ReLU)
// STREAM: the weight chunks stream through a four-slot ring (even chunk count); else all of a tile's chunks are resident
// MIX: the second source's chunks carry one tap (the centre) instead of nine
// NCS: chunks per tile known to the compiler - 1: one, 2: two or three, 0: four or more (how the finished tile's epilogue is spread)
// NS: halo ring slots (4).  PFD: halo chunks in flight per mover thread (4; 8 and 12 were measured: no gain - the launch is not latency-bound;
// so was a DMA form of the movers, buffer_load ... lds into a seven-slot ring: profiles/HISTORY.md)
// OUT: the finished tile leaves through an LDS out image and the MOVERS store it (register movers, an even chunk count >= 4): a store issued by
// a consumer wave between MFMAs waits for the memory pipeline behind the movers' requests, and the matrix pipe waits with it
// K32: the consumers of the out-image form run v_mfma_f32_16x16x32_bf16 - K = 32 = two (tap, 8-channel) pairs of a pair of chunks, 9 K steps per
// barrier interval.  Same FLOP per cycle and per LDS byte as 32x32x16, but the chip holds a higher clock on it under load (tools/micro/
// mfma_shapes.hip: 1.52 vs 1.36 PFLOP/s on random operands, 1.62 vs 1.58 on zeros).  Another summation order inside an MFMA: the outputs agree
// with conv_fwd_kernel to the last bf16 bit or two, not bit for bit.
// PAIR (round 5; every chunk of a pair from one source, i.e. even chunk counts per source): a halo request covers a PAIR of chunks - 32 channels =
// 64 contiguous bytes of a pixel = one whole request to the memory side - instead of one chunk's 32 bytes.  Measured at 64 tiles (1.07 GB, beyond
// every cache): with 32-byte pieces the L1 sent 1.26 GB of 64-byte read requests to the L2 for 0.68 GB of halo bytes and the L2 fetched 0.86 GB from
// HBM for the layer's 0.54 GB input (profiles/r05/ws16_pmc_64tiles.txt) - the launch was bound by its own over-fetch.  Register set R of a pair
// holds half of the pair's vectors (lane -> pixel v / 4, 16-byte segment v % 4: chunk (v % 4) / 2 of the pair, k-half v % 2); both sets are
// written into the pair's two ring slots in the same interval, as before.
// QUAD (round 5; resident weights, every source a multiple of 64 channels): BOTH pairs of a 128-byte piece of a pixel (four chunks = a whole line
// of the memory side) are requested in the same interval, every second interval, instead of one pair per interval.  With the pairs one interval
// (~2 us) apart the line fetched for the first was gone from the L2 for a third of the second requests at 64 tiles - 0.86 GB fetched for a
// 0.54 GB input (profiles/r05/dominant_conv_bf16_64tiles_pmc.json).  Three pair register sets: the even pairs use group 0, the odd pairs
// alternate between groups 1 and 2 (a quad is requested when its first pair's group has just been written to the ring and the odd group of the
// quad before last is free); the ring slot of a pair is its parity, whatever group it arrived in.
template <int BN, int XF, bool STREAM, bool MIX, int NCS, int NS, int PFD, bool OUT, bool K32 = false, bool PAIR = false, bool QUAD = false>
__global__ __launch_bounds__(512) void conv_ws16_kernel(ConvArgs A) {
    using L = W16Lds<BN>;
    constexpr int TH = 16, TW = 16, CK = 16, PSTR = L::PSTR, HW_ = TW + 2, NPIX = L::NPIX;
    constexpr int NCI = BN / 32, NPI = 2;                         // consumer wave: NCI blocks of 32 output channels x two blocks of 32 pixels
    constexpr int NMV = 256;                                      // mover threads
    constexpr int VPP = CK / 8, NA = (NPIX * VPP + NMV - 1) / NMV;
    constexpr int PF = QUAD ? 6 : PFD;
    constexpr int A_BYTES = L::A_IMG;                             // halo ring slot stride
    constexpr int IPG = QUAD ? 4 : PF / 2;                        // barrier intervals per iteration of the movers' loop
    static_assert(!(STREAM && MIX), "one-tap chunks only with resident weights");
    static_assert(!OUT || NCS == 0, "LDS out image: tiles of an even number (>= 4) of chunks");
    static_assert(!K32 || OUT, "the 16x16x32 consumers serve the out-image form");
    static_assert(NS == 4 && (QUAD || PF % 4 == 0), "a four-slot halo ring, register sets = slots mod 4");
    static_assert(!QUAD || (PAIR && !STREAM && OUT), "quad requests: the out-image form with resident weights and pair requests");
    static_assert(!PAIR || NCS == 0, "pair requests: tiles of an even number (>= 4) of chunks");
    const int NCH = A.nchunk;
    const int n0 = A.src[0].C / CK;                               // chunks of the first source (nine taps)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_a = smem;
    unsigned char *lds_w = smem + NS * A_BYTES;
    const int c0n = A.src[0].C, ctot = c0n + (A.nsrc > 1 ? A.src[1].C : 0);
    const int xfs = (ctot + 7) / 8 * 8;
    const int n1 = NCH - n0;
    const int wres_bytes = STREAM ? 4 * L::WCH9 : (MIX ? L::wbytes(n0, n1) : L::wbytes(NCH, 0));
    float *s_xf = reinterpret_cast<float *>(lds_w + wres_bytes);
    unsigned char *lds_o = lds_w + wres_bytes + 2 * xfs * 4;      // OUT only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int S = ntl * NCH;                                      // chunks of this workgroup's run
    if (S == 0) return;
    const int NI = (S + 1) >> 1;                                  // barrier intervals: two run chunks each
#ifdef CDNET_WS_STAMPS
    unsigned long long *s_stamp = reinterpret_cast<unsigned long long *>(lds_o + (OUT ? L::OUT_BYTES : 0)) + (wave >= 6 ? 512 : (wave >= 4 ? 256 : 0));
    const bool stamp_on = blockIdx.x == 17 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4);
    const int stamp_i0 = NI > 80 ? NI / 2 - 24 : 0;
    int sn = 0, stamp_iv = 0;
#endif

    if (wave >= 4) {
        // ================================ movers (conv_ws_kernel's, without the out path) ================================
        const int ptid = tid - 256, pw = wave - 4;
        constexpr int VPQ = PAIR ? 2 * VPP : VPP;                 // 16-byte vectors per halo pixel of a request group (PAIR: a pair of chunks)
        constexpr int NG = PAIR ? 2 : 1;                          // register-set groups that share one lane -> (pixel, segment) map
        const int slot = ptid % VPQ;                              // this thread's 16-byte segment (PAIR: chunk slot >> 1 of the pair, k-half slot & 1)
        const int khalf = slot & 1, c2 = slot >> 1;
        u32x4v pa[PF][NA];
        unsigned eo[XF != 0 ? PF : 1][NA];       // byte offsets of the requests (read again for a residual operand); bit 31 = zero fill
        int hyx[NG][NA], doff[NG][NA];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int v = ptid + (g * NA + i) * NMV;
                const int pix = v / VPQ;
                const int hy = pix / HW_, hx = pix - hy * HW_;
                hyx[g][i] = v < NPIX * VPQ ? ((hy << 8) | hx) : 0x1f1f;
                doff[g][i] = pix * PSTR + ((khalf ^ (hy & 1)) * 16);
            }
        const unsigned src_bytes0 = (unsigned)A.N * A.src[0].Hs * (A.src[0].row_stride ? A.src[0].row_stride : A.src[0].Ws * A.src[0].C) * 2u;
        const unsigned src_bytes1 = A.nsrc > 1 ? (unsigned)A.N * A.src[1].Hs * (A.src[1].row_stride ? A.src[1].row_stride : A.src[1].Ws * A.src[1].C) * 2u : 0u;
        const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 ? A.src[1].x : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.src[0].res ? A.src[0].res : A.src[0].x), 0, (int)src_bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A.nsrc > 1 && A.src[1].res ? A.src[1].res : A.src[0].x), 0, (int)src_bytes1, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff) -> u32x4v {
            return __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
        };
        auto bad_mask = [](int lo, int hi) -> unsigned {
            lo = lo < 0 ? 0 : (lo > 31 ? 31 : lo);
            hi = hi < lo ? lo : (hi > 31 ? 31 : hi);
            return ~(((1u << hi) - 1u) & ~((1u << lo) - 1u));
        };
        int ik = 0, ic = 0, in_, iy0, ix0;
        {
            in_ = t_lo / tiles_img;
            const int r = t_lo - in_ * tiles_img, ty = r / tiles_x;
            iy0 = ty * TH; ix0 = (r - ty * tiles_x) * TW;
        }
        int ck = 0;
        unsigned ge[NG][NA];
        auto chunk_src = [&](int k, int &si, int &cc0) {
            if (PAIR) k &= ~1;                                    // (the pair's first chunk: both register sets of a pair request from its base)
            if (k < n0) { si = 0; cc0 = k * CK; } else { si = 1; cc0 = (k - n0) * CK; }
        };
        auto issue = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            constexpr int G = PAIR ? (R & 1) : 0;                 // PAIR: which half of the pair's vectors this register set holds
            int si, cc0;
            chunk_src(ik, si, cc0);
            const ConvSrc &s = A.src[si];
            const int ikb = PAIR ? (ik & ~1) : ik;
            if (ikb == 0 || ikb == n0) {
                const int rs = s.row_stride ? s.row_stride : s.Ws * s.C;
                const int ylo = s.off_y > 0 ? s.off_y : 0, yhi = A.H < s.off_y + s.Hs ? A.H : s.off_y + s.Hs;
                const int xlo = s.off_x > 0 ? s.off_x : 0, xhi = A.W < s.off_x + s.Ws ? A.W : s.off_x + s.Ws;
                const unsigned rowbad = bad_mask(ylo - (iy0 - 1), yhi - (iy0 - 1)), colbad = bad_mask(xlo - (ix0 - 1), xhi - (ix0 - 1));
                const unsigned img_b = (unsigned)((in_ * s.Hs + (iy0 - 1 - s.off_y)) * rs + (ix0 - 1 - s.off_x) * s.C) * 2u;
                const unsigned rs_b = (unsigned)rs * 2u, c_b = (unsigned)s.C * 2u;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const unsigned hy = (unsigned)hyx[G][i] >> 8, hx = (unsigned)hyx[G][i] & 0xffu;
                    const unsigned t = (rowbad >> hy) | (colbad >> hx);
                    ge[G][i] = ((img_b + hy * rs_b + hx * c_b + (unsigned)slot * 16u) & 0x7fffffffu) | (t << 31);
                }
            }
            const __amdgpu_buffer_rsrc_t rsx = si ? rsx1 : rsx0;
            const unsigned cc0_b = (unsigned)cc0 * 2u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned voff = ge[G][i] + cc0_b;
                if (XF != 0) eo[R][i] = voff;
                if (!(A.debug & 2)) pa[R][i] = bload(rsx, voff);     // (2: ablation - no halo requests)
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
        // run chunk c_ (register set R = c_ % 4) -> ring slot c_ % 4 (QUAD: the register set is any of six, the slot pair is given: SB = 0 / 2)
        auto commit3 = [&](auto rc, int c_, auto sb_c) {
            constexpr int R = decltype(rc)::value;
            constexpr int SB = decltype(sb_c)::value;
            int si, cc0;
            chunk_src(ck, si, cc0);
            if (c_ + 1 < S) { if (++ck == NCH) ck = 0; }
            const ConvSrc &s = A.src[si];
            const float *xf = s_xf + (si ? c0n : 0) + cc0 + slot * 8;
            float sc[8], sh[8];
            if (XF != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = xf[j]; sh[j] = xf[xfs + j]; }
            }
            constexpr int G = PAIR ? (R & 1) : 0;
            // (PAIR: this thread's vectors belong to chunk c2 of the pair - ring slot (R & 2) + c2 - whichever of the pair's two sets R is)
            unsigned char *dst0 = lds_a + (SB >= 0 ? SB + c2 : (PAIR ? ((R & 2) + c2) : (R & 3))) * A_BYTES;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                u32x4v val;
                if (XF == 0) val = pa[R][i];
                else {
                    ChanXf t;
                    t.on = s.scale != nullptr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { t.sc[j] = sc[j]; t.sh[j] = sh[j]; }
                    V16 raw, rr;
                    raw.u = __builtin_bit_cast(uint4, pa[R][i]);
                    const bool relu = s.relu != 0, f16 = s.f16 != 0;
                    if (s.res) {
                        rr.u = __builtin_bit_cast(uint4, bload(si ? rsr1 : rsr0, eo[R][i]));
                        val = __builtin_bit_cast(u32x4v, xform8(raw, &rr, t, relu, f16).u);
                    } else if (!t.on && !relu && !f16) val = pa[R][i];
                    else val = __builtin_bit_cast(u32x4v, xform8(raw, nullptr, t, relu, f16).u);
                    const unsigned keep = (int)eo[R][i] < 0 ? 0u : 0xffffffffu;      // outside the image / source: zeros (after the transform)
                    val &= keep;
                }
                if (ptid + (G * NA + i) * NMV < NPIX * VPQ)
                    *reinterpret_cast<u32x4v *>(dst0 + doff[G][i]) = val;
            }
        };
        auto commit = [&](auto rc, int c_) __attribute__((always_inline)) { commit3(rc, c_, std::integral_constant<int, -1>{}); };
        // STREAM: weight chunk wk of the tile -> weight slot (run chunk & 3) by LDS-DMA (the packed chunk is the LDS image)
        int wk = 0, wq = 0;
        auto dma_w = [&]() {
            constexpr int NVEC = L::WCH9 / 16;
            static_assert(NVEC % 64 == 0, "whole wave-instructions");
            const unsigned short *wsrc = A.w + ((size_t)cout_tile * NCH + wk) * (L::WCH9 / 2);
            unsigned char *wdst = lds_w + (wq & 3) * L::WCH9;
#pragma unroll
            for (int i = 0; i < (NVEC / 64 + 3) / 4; ++i) {
                const int v0 = (i * 4 + pw) * 64;
                if (v0 < NVEC)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + (size_t)(v0 + lane) * 8),
                                                     (__attribute__((address_space(3))) void *)(wdst + v0 * 16), 16, 0, 0);
            }
            if (++wk == NCH) wk = 0;
            ++wq;
        };
        auto wait_vm = [](auto n_c) {
            constexpr int N = decltype(n_c)::value;
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
        };
        using NHL = std::integral_constant<int, 2 * NA>;          // halo requests of one interval (two chunks)
        // the movers' barrier of the streaming loop by hand (conv_ws_kernel): this wave's LDS writes done, its weight DMA landed (everything
        // older than the N youngest vector-memory operations), then the barrier; the halo requests issued after the DMA stay in flight
        auto stream_sync = [](auto n_c) {
            constexpr int N = decltype(n_c)::value;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // OUT: the half (pixel block pi of the four consumer waves) of a finished tile the consumers parked in the PREVIOUS interval leaves
        // here: wave pw stores the block of consumer wave pw - 8 lanes per pixel, one 128-byte line each (BN = 64)
        const int NIT = NCH >> 1;                                // intervals per tile (OUT: even chunk counts)
        int st_jt = 0, st_it = 0;                                // tile / interval-in-tile of the interval the consumers work on
        int h_n[2], h_y0[2], h_x0[2];                            // coordinates of tiles jt - 1, jt - 2
        int c_n = in_, c_y0 = iy0, c_x0 = ix0;                   // ... of tile jt
        h_n[0] = h_n[1] = c_n; h_y0[0] = h_y0[1] = c_y0; h_x0[0] = h_x0[1] = c_x0;
        float dot_w8[8], dot_b0 = 0.f;                           // cdnet_conv_args.dot_w of this lane's 8 channels (its 16-byte segment of a pixel)
#pragma unroll
        for (int j = 0; j < 8; ++j) dot_w8[j] = 0.f;
        if (OUT && !STREAM && A.dot_out) {
            const int c8 = cout0 + (lane % (BN / 8)) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot_w8[j] = c8 + j < A.Cout ? A.dot_w[c8 + j] : 0.f;
            dot_b0 = A.dot_b ? A.dot_b[0] : 0.f;
        }
        auto store_half = [&](int hf, int tn, int ty0, int tx0) {
            {
            const int bw = pw;                                   // the block of consumer wave pw leaves through mover wave pw
            constexpr int SPP = BN / 8;                          // 16-byte segments per pixel
            constexpr int PPR = 64 / SPP, NR = 32 / PPR;         // pixels per wave-instruction, instructions per block
            const int seg = lane % SPP, lp = lane / SPP;
            const unsigned char *blk = lds_o + hf * L::OHALF + bw * L::OBLK + seg * 16;
            u32x4v v[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = *reinterpret_cast<const u32x4v *>(blk + (r * PPR + lp) * L::OROW);
            const bool ok = cout0 + seg * 8 < A.Cout && !(A.debug & 8);
            if (A.out) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int px = r * PPR + lp;
                    unsigned short *dstp = A.out + (((size_t)tn * A.H + ty0 + bw * 4 + hf * 2 + (px >> 4)) * A.W + tx0 + (px & 15)) * A.out_cstride + A.out_coff + cout0 + seg * 8;
                    if (ok) *reinterpret_cast<u32x4v *>(dstp) = v[r];
                }
            }
            if (!STREAM && A.dot_out) {
                // the 1x1 classifier over the rounded output (cdnet_conv_args.dot_*): this lane's 8 channels of a pixel against its 8
                // weights, then the sum over the SPP lanes of the pixel (they are neighbours); the first of them stores the logit
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    XfWords m;
                    m.u = __builtin_bit_cast(xf_u32x4, v[r]);
                    float sd = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sd = fmaf(__uint_as_float(m.w[k] << 16), dot_w8[2 * k], sd);
                        sd = fmaf(__uint_as_float(m.w[k] & 0xffff0000u), dot_w8[2 * k + 1], sd);
                    }
#pragma unroll
                    for (int o = 1; o < SPP; o <<= 1) sd += __shfl_xor(sd, o);
                    const int px = r * PPR + lp;
                    if (seg == 0 && !(A.debug & 8))
                        A.dot_out[((size_t)tn * A.H + ty0 + bw * 4 + hf * 2 + (px >> 4)) * A.W + tx0 + (px & 15)] = sd + dot_b0;
                }
            }
            if (A.pool_out) {
                // nn.MaxPool2d(2, 2) of the block's two rows beside the stores (torchvision VGG 'M' layers after a ReLU: the values are
                // non-negative, their bf16 order is their int16 order): rows r and r + NR / 2 hold the same columns; the column partner of
                // a pixel sits SPP lanes away.  The lanes of even columns store the block's 8 pooled pixels.
                auto max4 = [](u32x4v a, u32x4v b) -> u32x4v {       // (xf_max_nonneg_bf8: element access through a union - xform.h)
                    return __builtin_bit_cast(u32x4v, xf_max_nonneg_bf8(__builtin_bit_cast(xf_u32x4, a), __builtin_bit_cast(xf_u32x4, b)));
                };
                const int Hp = A.H >> 1, Wp = A.W >> 1;
#pragma unroll
                for (int r = 0; r < NR / 2; ++r) {
                    XfWords m, o;
                    m.u = __builtin_bit_cast(xf_u32x4, max4(v[r], v[r + NR / 2]));
#pragma unroll
                    for (int k = 0; k < 4; ++k) o.w[k] = (unsigned)__shfl_xor((int)m.w[k], SPP);
                    m.u = xf_max_nonneg_bf8(m.u, o.u);
                    const int px = r * PPR + lp;                 // column of this lane's pixel inside the tile
                    unsigned short *dstp = A.pool_out + (((size_t)tn * Hp + ((ty0 + bw * 4 + hf * 2) >> 1)) * Wp + ((tx0 + px) >> 1)) * A.Cout + cout0 + seg * 8;
                    if (ok && !(lp & 1)) *reinterpret_cast<u32x4v *>(dstp) = __builtin_bit_cast(u32x4v, m.u);
                }
            }
            }
        };
        // called once per mover interval iv (the consumers' interval); stores what the consumers parked in interval iv - 1
        auto store_prev = [&](int iv) -> bool {
            bool stored = false;
            if (OUT) {
                const int pv = iv - 1;
                if (pv >= 0 && pv < NI) {
                    // interval pv = (tile pj, interval pit)
                    int pj = st_jt, pit = st_it - 1;
                    if (pit < 0) { pj -= 1; pit = NIT - 1; }
                    if (pj >= 1 && pit <= 1) {
                        const int k = st_jt - (pj - 1) - 1;           // tile pj - 1 = history slot (st_jt - 1 - (pj - 1)) ... 0: jt - 1, 1: jt - 2
                        store_half(pit, h_n[k], h_y0[k], h_x0[k]);
                        stored = true;
                    }
                }
                // advance to the next interval
                if (++st_it == NIT) {
                    st_it = 0; ++st_jt;
                    h_n[1] = h_n[0]; h_y0[1] = h_y0[0]; h_x0[1] = h_x0[0];
                    h_n[0] = c_n; h_y0[0] = c_y0; h_x0[0] = c_x0;
                    c_x0 += TW;
                    if (c_x0 >= A.W) { c_x0 = 0; c_y0 += TH; if (c_y0 >= A.H) { c_y0 = 0; ++c_n; } }
                }
            }
            return stored;
        };
        // lgkmcnt(0) + barrier, no wait for vector memory (the stores of the out image stay in flight)
        auto lds_sync = []() {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // register set of run chunk c = c mod PF (compile-time inside the unrolled loop body), ring slot = c mod 4
        auto for_sets = [&](auto f) {                            // f(integral_constant<int, k>) for k = 0 .. PF - 1
            [&]<int... K>(std::integer_sequence<int, K...>) { (f(std::integral_constant<int, K>{}), ...); }(std::make_integer_sequence<int, PF>{});
        };
        if constexpr (QUAD) {
            // ================================ movers, quad requests ================================
            using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>;       // group 0: the even pairs
            using R2 = std::integral_constant<int, 2>; using R3 = std::integral_constant<int, 3>;       // group 1 | the odd pairs,
            using R4 = std::integral_constant<int, 4>; using R5 = std::integral_constant<int, 5>;       // group 2 | alternating
            using S0 = std::integral_constant<int, 0>; using S2 = std::integral_constant<int, 2>;       // ring slots of an even / odd pair
            issue(R0{}); issue(R1{}); issue(R2{}); issue(R3{});                                         // quad 0: pairs 0, 1
            for (int c = ptid; c < ctot; c += NMV) {
                const ConvSrc &Sx = c < c0n ? A.src[0] : A.src[1];
                const int cc = c < c0n ? c : c - c0n;
                s_xf[c] = Sx.scale ? Sx.scale[cc] : 1.f;
                s_xf[xfs + c] = Sx.shift ? Sx.shift[cc] : 0.f;
            }
            __syncthreads();                                     // B0: table (+ resident weights)
            commit3(R0{}, 0, S0{}); commit3(R1{}, 1, S0{});        // pair 0
            issue(R0{}); issue(R1{}); issue(R4{}); issue(R5{});  // quad 1: pair 2 -> group 0, pair 3 -> group 2
            __syncthreads();                                     // B1: run chunks 0, 1 staged
            // interval i: the consumers work on pair i, pair i + 1 is staged here; a quad is requested in the odd intervals
            for (int i0 = 0; i0 < NI; i0 += 4) {
                W16_STAMP(1); commit3(R2{}, 2 * i0 + 2, S2{}); commit3(R3{}, 2 * i0 + 3, S2{}); W16_STAMP(3);      // pair i0 + 1 (group 1)
                store_prev(i0); W16_STAMP(4); lds_sync(); W16_STAMP(5); W16_STAMP_NEXT();
                W16_STAMP(1); commit3(R0{}, 2 * i0 + 4, S0{}); commit3(R1{}, 2 * i0 + 5, S0{});                    // pair i0 + 2 (group 0)
                issue(R0{}); issue(R1{}); issue(R2{}); issue(R3{}); W16_STAMP(3);                                // pairs i0 + 4, i0 + 5 -> groups 0, 1
                store_prev(i0 + 1); W16_STAMP(4); lds_sync(); W16_STAMP(5); W16_STAMP_NEXT();
                W16_STAMP(1); commit3(R4{}, 2 * i0 + 6, S2{}); commit3(R5{}, 2 * i0 + 7, S2{}); W16_STAMP(3);      // pair i0 + 3 (group 2)
                store_prev(i0 + 2); W16_STAMP(4); lds_sync(); W16_STAMP(5); W16_STAMP_NEXT();
                W16_STAMP(1); commit3(R0{}, 2 * i0 + 8, S0{}); commit3(R1{}, 2 * i0 + 9, S0{});                    // pair i0 + 4 (group 0)
                issue(R0{}); issue(R1{}); issue(R4{}); issue(R5{}); W16_STAMP(3);                                // pairs i0 + 6, i0 + 7 -> groups 0, 2
                store_prev(i0 + 3); W16_STAMP(4); lds_sync(); W16_STAMP(5); W16_STAMP_NEXT();
            }
            store_prev((NI + IPG - 1) / IPG * IPG);              // what the consumers parked in the very last interval
            lds_sync();                                          // E
            lds_sync();                                          // F
            const int tl = t_hi - 1, ln = tl / tiles_img, lr = tl - ln * tiles_img, lty = lr / tiles_x;
            store_half(0, ln, lty * TH, (lr - lty * tiles_x) * TW);
            store_half(1, ln, lty * TH, (lr - lty * tiles_x) * TW);
#ifdef CDNET_WS_STAMPS
            if (stamp_on) { for (int i = 0; i < sn; ++i) g_ws16_stamps[1024 + i] = s_stamp[i]; g_ws16_stamps[1024 + sn] = 0; }
#endif
            return;
        }
        for_sets([&](auto k) { issue(k); });
        for (int c = ptid; c < ctot; c += NMV) {
            const ConvSrc &Sx = c < c0n ? A.src[0] : A.src[1];
            const int cc = c < c0n ? c : c - c0n;
            s_xf[c] = Sx.scale ? Sx.scale[cc] : 1.f;
            s_xf[xfs + c] = Sx.shift ? Sx.shift[cc] : 0.f;
        }
        __syncthreads();                                         // B0: table (+ resident weights)
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        commit(I0{}, 0);
        commit(I1{}, 1);
        if (STREAM) { dma_w(); dma_w(); }
        issue(I0{});
        issue(I1{});
        if (STREAM) wait_vm(NHL{});
        __syncthreads();                                         // B1: run chunks 0, 1 staged
        // interval i: the consumers work on run chunks 2i, 2i+1; here the two chunks after them are staged (their register sets are
        // refilled with the chunks PF further on).  One loop iteration = PF / 2 intervals, so that every set index is a constant.
        for (int i0 = 0; i0 < NI; i0 += IPG) {
            [&]<int... M>(std::integer_sequence<int, M...>) {
                ([&] {
                    constexpr int RA = (2 * M + 2) % PF, RB = (2 * M + 3) % PF;
                    const int qa = 2 * (i0 + M) + 2;
                    using SA = std::integral_constant<int, RA>;
                    using SB = std::integral_constant<int, RB>;
                    if (!STREAM) {
                        W16_STAMP(1); commit(SA{}, qa); W16_STAMP(2); issue(SA{}); commit(SB{}, qa + 1); issue(SB{}); W16_STAMP(3);
                        store_prev(i0 + M);
                        W16_STAMP(4);
                        if (OUT) lds_sync(); else __syncthreads();
                        W16_STAMP(5); W16_STAMP_NEXT();
                    } else {
                        commit(SA{}, qa); commit(SB{}, qa + 1);
                        dma_w(); dma_w();
                        if (OUT) {
                            // (the stores sit between the weight DMA and the halo requests: the wait leaves them in flight too; the compiler
                            //  fences keep that issue order - the explicit vmcnt below counts on it)
                            asm volatile("" ::: "memory");
                            const bool stored = store_prev(i0 + M);
                            asm volatile("" ::: "memory");
                            issue(SA{}); issue(SB{});
                            // (the stores of a half - NRS of them, plus NRS / 2 pooled ones - are younger than the weight DMA and stay in flight)
                            constexpr int NRS = 32 / (64 / (BN / 8));
                            if (!stored) stream_sync(NHL{});
                            else if (A.pool_out) stream_sync(std::integral_constant<int, 2 * NA + NRS + NRS / 2>{});
                            else stream_sync(std::integral_constant<int, 2 * NA + NRS>{});
                        } else {
                            issue(SA{}); issue(SB{});
                            stream_sync(NHL{});
                        }
                    }
                }(), ...);
            }(std::make_integer_sequence<int, IPG>{});
        }
        if (OUT) {
            store_prev((NI + IPG - 1) / IPG * IPG);              // what the consumers parked in the very last interval
            lds_sync();                                          // E: this wave's reads of the out image are done - the consumers park the last tile
            lds_sync();                                          // F: ... both halves of it are in the image
            const int tl = t_hi - 1, ln = tl / tiles_img, lr = tl - ln * tiles_img, lty = lr / tiles_x;
            store_half(0, ln, lty * TH, (lr - lty * tiles_x) * TW);
            store_half(1, ln, lty * TH, (lr - lty * tiles_x) * TW);
        }
#ifdef CDNET_WS_STAMPS
        if (stamp_on) { for (int i = 0; i < sn; ++i) g_ws16_stamps[1024 + i] = s_stamp[i]; g_ws16_stamps[1024 + sn] = 0; }
#endif
        return;
    }

    // ================================ consumers ================================
    if (!STREAM) {
        // the resident weights of this output-channel tile: one contiguous block of the pack
        const u32x4v *src = reinterpret_cast<const u32x4v *>(reinterpret_cast<const unsigned char *>(A.w) + (size_t)cout_tile * wres_bytes);
        u32x4v *dst = reinterpret_cast<u32x4v *>(lds_w);
        const int nv = wres_bytes / 16;
        for (int v0 = 0; v0 < nv; v0 += 256 * 8) {
            u32x4v wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = src[v0 + tid + i * 256 < nv ? v0 + tid + i * 256 : 0];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v0 + tid + i * 256 < nv) dst[v0 + tid + i * 256] = wv[i];
        }
    }
    __syncthreads();                                             // B0
    if constexpr (K32) {
        // ================================ consumers, v_mfma_f32_16x16x32_bf16 ================================
        // A = weights (16 output channels x 32 k), B = pixels (32 k x 16 pixels of one tile row), D: lane (pixel n = lane & 15, kg = lane >> 4)
        // holds output channels 4 kg .. 4 kg + 3 of the block.  K step s of a chunk pair: this lane's 8 k = k-block kb = 4 s + kg of the 36
        // (chunk, tap, channel half) blocks - the fragment addresses differ between the four lane groups and are kept per step.
        typedef float f32x4a __attribute__((ext_vector_type(4)));
        constexpr int NCB = BN / 16, NPB = 4;                    // wave: NCB blocks of 16 output channels x the 4 tile rows 4 wm .. 4 wm + 3
        const int wm = wave, n16 = lane & 15, kg = lane >> 4;
        int PB9[9], WB9[9], PB1, WB1;                              // (pixel-fragment offsets for even tile rows; an odd row flips the k-half swizzle: ^ 16)
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int kb = 4 * st + kg, c2 = kb >= 18 ? 1 : 0, kbp = kb - 18 * c2, tap = kbp >> 1, ch = kbp & 1, tr = tap / 3, tc = tap - 3 * tr;
            PB9[st] = c2 * A_BYTES + ((wm * 4 + tr) * HW_ + n16 + tc) * PSTR + ((ch ^ (tr & 1)) * 16);
            WB9[st] = c2 * L::WCH9 + tap * 2 * BN * 16 + ch * BN * 16 + n16 * 16;
        }
        {
            const int c2 = kg >> 1, ch = kg & 1;                 // a pair of one-tap chunks: K = 2 x 16, the centre tap
            PB1 = c2 * A_BYTES + ((wm * 4 + 1) * HW_ + n16 + 1) * PSTR + ((ch ^ 1) * 16);
            WB1 = c2 * L::WCH1 + ch * BN * 16 + n16 * 16;
        }
        f32x4a accA[NCB][NPB], accB[NCB][NPB], binit[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cout0 + cb * 16 + 4 * kg + r;
                binit[cb][r] = co < A.Cout ? (A.bias ? A.bias[co] : 0.f) + (A.oshift ? A.oshift[co] : 0.f) : 0.f;
            }
        const xf_s16x2 lo_clamp = A.orelu ? xf_s16x2{0, 0} : xf_s16x2{(short)-32768, (short)-32768};
        unsigned char *const o_lane = lds_o + wave * L::OBLK + n16 * L::OROW + kg * 8;
        // epilogue of a finished set: unit u = (pb, cb) - 4 consecutive output channels of this lane's pixel in tile row pb - three
        // micro-operations: convert, ReLU clamp, one 8-byte write into the out image (half pb >> 1, pixel (pb & 1) * 16 + n)
        constexpr int NMO = NPB * NCB * 3;
        unsigned e0 = 0, e1 = 0;
        auto micro = [&](const f32x4a (&P)[NCB][NPB], int e) {
            const int u = e / 3, k = e % 3, pb = u / NCB, cb = u % NCB;
            if (k == 0) {
                const xf_f32x2 a = {P[cb][pb][0], P[cb][pb][1]}, b = {P[cb][pb][2], P[cb][pb][3]};
                e0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, xf_bf16x2));
                e1 = __builtin_bit_cast(unsigned, __builtin_convertvector(b, xf_bf16x2));
            } else if (k == 1) {
                e0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, e0), lo_clamp));
                e1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, e1), lo_clamp));
            } else {
                *reinterpret_cast<uint2 *>(o_lane + (pb >> 1) * L::OHALF + (pb & 1) * 16 * L::OROW + cb * 32) = make_uint2(e0, e1);
            }
        };
        // one barrier interval = a pair of run chunks on accumulator set C: NT = 9 -> nine K steps of NCB x 4 MFMAs, NT = 1 -> one.
        // The fragments of a step are requested one step ahead (two sets).  FIRST: the tile starts from the bias.  EP of 2 (HP): the interval
        // carries that half (tile rows 2 EP, 2 EP + 1 = out-image half EP) of the finished set's epilogue.
        auto pair_step = [&](auto nt_c, auto first_c, auto ep_c, f32x4a (&C)[NCB][NPB], const f32x4a (&P)[NCB][NPB], const unsigned char *la, const unsigned char *lw) {
            constexpr int NT = decltype(nt_c)::value;
            constexpr bool FIRST = decltype(first_c)::value;
            constexpr int EP = decltype(ep_c)::value;              // -1: no epilogue
            constexpr int NST = NT == 9 ? 9 : 1;
            if (A.debug & 1) return;
            // LDS-typed bases; resident weights: the slot bases of a tile's first pairs are compile-time constants and the compiler would keep
            // a fragment address per (pair, step) in registers (21 spilled) - opaque bases cost one add per read, as in the streamed form
            typedef const __attribute__((address_space(3))) unsigned char *lds_cp;
            lds_cp la3 = (lds_cp)la, lw3 = (lds_cp)lw;
            if constexpr (!STREAM) asm volatile("" : "+s"(la3), "+s"(lw3));
            // weight fragments of a whole step one step ahead (two sets), pixel fragments one tile row ahead (two registers sets of one)
            bf16x8 wa[2][NCB], pf[2];
            auto req_w = [&](int st) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    wa[st & 1][cb] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(lw3 + (NT == 9 ? WB9[st] : WB1) + cb * 256);
            };
            auto req_p = [&](int st, int pb) {
                const int o = (NT == 9 ? PB9[st] : PB1) ^ ((pb & 1) * 16);
                pf[(st * NPB + pb) & 1] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(la3 + o + pb * HW_ * PSTR);
            };
            constexpr int NG = NST * NCB * NPB;
            constexpr int E0 = EP >= 0 ? EP * NMO / 2 : 0, CNT = EP >= 0 ? NMO / 2 : 0, CNTD = CNT > 0 ? CNT : 1;
            req_w(0);
            req_p(0, 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) req_w(st + 1);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    if (pb + 1 < NPB) req_p(st, pb + 1); else if (st + 1 < NST) req_p(st + 1, 0);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const bf16x8 &pv = pf[(st * NPB + pb) & 1];
                        if (FIRST && st == 0) C[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[st & 1][cb], pv, binit[cb], 0, 0, 0);
                        else C[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[st & 1][cb], pv, C[cb][pb], 0, 0, 0);
                        if (CNT > 0) {
                            const int g = (st * NPB + pb) * NCB + cb;
                            bool any = false;
#pragma unroll
                            for (int i = 0; i < CNT; ++i)
                                if ((i * NG) / CNTD == g) {
                                    if (!any) __builtin_amdgcn_sched_barrier(0);
                                    any = true;
                                    micro(P, E0 + i);
                                }
                            if (any) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        };
        using F_ = std::false_type;
        using T_ = std::true_type;
        using N9 = std::integral_constant<int, 9>;
        using N1 = std::integral_constant<int, 1>;
        using EN = std::integral_constant<int, -1>;
        using E0_ = std::integral_constant<int, 0>;
        using E1_ = std::integral_constant<int, 1>;
        int q = 0, qs = 0;
        auto slot_a = [&]() -> const unsigned char * { return lds_a + qs * A_BYTES; };
        auto slot_w = [&](int k) -> const unsigned char * {
            if (STREAM) return lds_w + (q & 3) * L::WCH9;
            if (MIX && k >= n0) return lds_w + n0 * L::WCH9 + (k - n0) * L::WCH1;
            return lds_w + k * L::WCH9;
        };
        auto after_pair = [&]() {
            q += 2;
            qs = (qs + 2) & 3;
            __syncthreads();
        };
        auto tile_step = [&](auto has_prev, f32x4a (&C)[NCB][NPB], const f32x4a (&P)[NCB][NPB]) {
            constexpr bool HP = decltype(has_prev)::value;
            if (HP) pair_step(N9{}, T_{}, E0_{}, C, P, slot_a(), slot_w(0)); else pair_step(N9{}, T_{}, EN{}, C, P, slot_a(), slot_w(0));
            after_pair();
            if (HP) pair_step(N9{}, F_{}, E1_{}, C, P, slot_a(), slot_w(2)); else pair_step(N9{}, F_{}, EN{}, C, P, slot_a(), slot_w(2));
            after_pair();
            for (int k = 4; k < NCH; k += 2) {
                if (MIX && k >= n0) pair_step(N1{}, F_{}, EN{}, C, P, slot_a(), slot_w(k));
                else pair_step(N9{}, F_{}, EN{}, C, P, slot_a(), slot_w(k));
                after_pair();
            }
        };
        __syncthreads();                                         // B1: run chunks 0, 1 are staged
        tile_step(F_{}, accA, accB);
        int j = 1;
        for (; j + 1 < ntl; j += 2) {
            tile_step(T_{}, accB, accA);
            tile_step(T_{}, accA, accB);
        }
        const bool tail = j < ntl;
        if (tail) tile_step(T_{}, accB, accA);
        for (int i = NI; i % IPG != 0; ++i) __syncthreads();     // the movers' loop runs whole groups of IPG intervals
        __syncthreads();                                         // E: the movers' last reads of the out image are done
        if (tail) {
#pragma unroll
            for (int e = 0; e < NMO; ++e) micro(accB, e);
        } else {
#pragma unroll
            for (int e = 0; e < NMO; ++e) micro(accA, e);
        }
        __syncthreads();                                         // F: the last tile is in the image; the movers store it
        return;
    }
    const int wm = wave;
    const int half = lane >> 5, l31 = lane & 31;
    int pbase[NPI][2];                                           // pixel fragment (B operand) of block pi; [.][parity of the tap's row offset]
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) {
        const int m = (wm * NPI + pi) * 32 + l31;
#pragma unroll
        for (int par = 0; par < 2; ++par) pbase[pi][par] = ((m / TW) * HW_ + m % TW) * PSTR + ((half ^ ((m / TW + par) & 1)) * 16);
    }
    const int wbase = half * BN * 16 + sigma32(l31) * 16;         // weight fragment (A operand): packed column sigma(l31) of block ci (+ ci * 512)
    f32x16 accA[NCI][NPI], accB[NCI][NPI];
    // initial value of every accumulator block ci: bias + epilogue shift of this lane's 16 output channels
    f32x16 binit[NCI];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cout0 + ci * 32 + 16 * (r >> 3) + 8 * half + (r & 7);
            float b = 0.f;
            if (co < A.Cout) b = (A.bias ? A.bias[co] : 0.f) + (A.oshift ? A.oshift[co] : 0.f);
            binit[ci][r] = b;
        }
    const xf_s16x2 lo_clamp = A.orelu ? xf_s16x2{0, 0} : xf_s16x2{(short)-32768, (short)-32768};
    // the finished tile's stores: pixel (row 4 wm + 2 pi + (l31 >> 4), column l31 & 15) of the tile, channels 32 ci + 16 s + 8 half .. + 8
    const unsigned pix_b = (unsigned)A.out_cstride * 2u;
    const unsigned l_off = ((unsigned)(l31 >> 4) * (unsigned)A.W + (unsigned)(l31 & 15)) * pix_b + (unsigned)half * 16u;
    char *const out_b = reinterpret_cast<char *>(A.out + A.out_coff + cout0);
    int p_n, p_y0, p_x0;                                         // the finished tile (whose epilogue rides in the current one)
    {
        p_n = t_lo / tiles_img;
        const int r = t_lo - p_n * tiles_img, ty = r / tiles_x;
        p_y0 = ty * TH; p_x0 = (r - ty * tiles_x) * TW;
    }
    char *row_base[NPI];
    auto set_row_bases = [&]() {
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi)
            row_base[pi] = out_b + ((size_t)(p_n * A.H + p_y0 + wm * 4 + pi * 2) * A.W + p_x0) * pix_b;
    };
    // epilogue micro-operations of a finished set, one (pixel block, output-channel block) group at a time: for its two units s = 0, 1 - 8
    // consecutive output channels of this lane's pixel each - convert 2 x 2 and ReLU (three micro-operations per unit), then the group's two
    // stores one after the other (64 contiguous bytes per pixel); the groups of a pixel block follow each other, so the 128-byte lines of
    // its 32 pixels are completed while they are still in the L2 (with a tile's stores spread over the whole next tile, the partial lines of
    // a launch larger than the caches were evicted and written to HBM piecemeal: 398 us against 354 at 64 tiles)
    constexpr int UG = 2, NMO = NPI * NCI * UG * 4;
    unsigned dd[UG][4];
    auto cvt2 = [&](float a, float b) -> unsigned {              // (bf16 outputs only: the launcher sends fp16 outputs to the older kernels)
        const xf_f32x2 p = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(p, xf_bf16x2));
    };
    auto relu2 = [&](unsigned v) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(xf_s16x2, v), lo_clamp));
    };
    unsigned char *const o_lane = lds_o + wave * L::OBLK + l31 * L::OROW + half * 16;      // OUT: this lane's pixel in its wave's block
    auto micro = [&](const f32x16 (&P)[NCI][NPI], int e, bool direct) __attribute__((always_inline)) {
        const int gi = e / (4 * UG), j = e % (4 * UG);
        const int pi = gi / NCI, ci = gi % NCI;
        const f32x16 &v = P[ci][pi];
        if (j < 3 * UG) {
            const int s = j / 3, k = j % 3;
            if (k == 0) { dd[s][0] = cvt2(v[8 * s + 0], v[8 * s + 1]); dd[s][1] = cvt2(v[8 * s + 2], v[8 * s + 3]); }
            else if (k == 1) { dd[s][2] = cvt2(v[8 * s + 4], v[8 * s + 5]); dd[s][3] = cvt2(v[8 * s + 6], v[8 * s + 7]); }
            else { dd[s][0] = relu2(dd[s][0]); dd[s][1] = relu2(dd[s][1]); dd[s][2] = relu2(dd[s][2]); dd[s][3] = relu2(dd[s][3]); }
        } else {
            const int s = j - 3 * UG;
            if constexpr (OUT) {                                 // (every store of the out-image form is the movers': `direct` is never set there)
                const u32x4v val = {dd[s][0], dd[s][1], dd[s][2], dd[s][3]};
                *reinterpret_cast<u32x4v *>(o_lane + pi * L::OHALF + ci * 64 + s * 32) = val;
            } else if (cout0 + ci * 32 + s * 16 < A.Cout && !(A.debug & 8)) {
                char *sb = row_base[pi] + (ci * 64 + s * 32);
                asm volatile("" : "+s"(sb));                    // (scalar base kept opaque: uniform base + one per-lane offset register = the store's saddr form)
                const u32x4v val = {dd[s][0], dd[s][1], dd[s][2], dd[s][3]};
                *(__attribute__((address_space(1))) u32x4v *)((__attribute__((address_space(1))) char *)sb + l_off) = val;
            }
        }
    };

    // one chunk step on accumulator set C: NT taps x NCI x NPI MFMAs; the fragments of a tap are requested two taps ahead (three sets).
    // FIRST: the tile's first chunk starts from the bias.  EP of NP (NP > 0): this step carries that share of the finished set P's epilogue
    // micro-operations, one or two per MFMA gap, evenly spread.
    auto chunk_step = [&](auto nt_c, auto first_c, auto ep_c, auto np_c, f32x16 (&C)[NCI][NPI], const f32x16 (&P)[NCI][NPI],
                          const unsigned char *la, const unsigned char *lw) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int EP = decltype(ep_c)::value, NP = decltype(np_c)::value;
        if (A.debug & 1) return;                                  // ablation: no fragment reads, no MFMAs
        bf16x8 wf[3][NCI], pf[3][NPI];
        auto request = [&](int t) {
            const int tr = NT == 9 ? t / 3 : 1, tc = NT == 9 ? t % 3 : 1;
            const int po = (tr * HW_ + tc) * PSTR, par = tr & 1;
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) wf[t % 3][ci] = *reinterpret_cast<const bf16x8 *>(lw + wbase + t * 2 * BN * 16 + ci * 512);
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) pf[t % 3][pi] = *reinterpret_cast<const bf16x8 *>(la + pbase[pi][par] + po);
        };
        constexpr int NG = NT * NCI * NPI;                        // MFMA gaps of the step
        constexpr int NPD = NP > 0 ? NP : 1;
        constexpr int E0 = NP > 0 ? EP * NMO / NPD : 0, E1 = NP > 0 ? (EP + 1) * NMO / NPD : 0, CNT = E1 - E0, CNTD = CNT > 0 ? CNT : 1;
        request(0);
        if (NT > 1) request(1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 2 < NT) request(t + 2);
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) {
                    if (FIRST && t == 0) C[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t % 3][ci], pf[t % 3][pi], binit[ci], 0, 0, 0);
                    else C[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t % 3][ci], pf[t % 3][pi], C[ci][pi], 0, 0, 0);
                    if (CNT > 0) {
                        const int g = (t * NCI + ci) * NPI + pi;
                        // micro-operation i of this step rides in gap (i * NG) / CNT
                        bool any = false;
#pragma unroll
                        for (int i = 0; i < CNT; ++i)
                            if ((i * NG) / CNTD == g) {
                                if (!any) __builtin_amdgcn_sched_barrier(0);
                                any = true;
                                micro(P, E0 + i, false);
                            }
                        if (any) __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    };
    using F_ = std::false_type;
    using T_ = std::true_type;
    using N9 = std::integral_constant<int, 9>;
    using N1 = std::integral_constant<int, 1>;
    using Z0 = std::integral_constant<int, 0>;
    using Z1 = std::integral_constant<int, 1>;
    using Z2 = std::integral_constant<int, 2>;
    using Z3 = std::integral_constant<int, 3>;
    using Z4 = std::integral_constant<int, 4>;
    int q = 0, qs = 0;                                           // run chunk counter; its halo ring slot (q mod NS)
    auto slot_a = [&]() -> const unsigned char * { return lds_a + qs * A_BYTES; };
    auto slot_w = [&](int k) -> const unsigned char * {
        if (STREAM) return lds_w + (q & 3) * L::WCH9;
        if (MIX && k >= n0) return lds_w + n0 * L::WCH9 + (k - n0) * L::WCH1;
        return lds_w + k * L::WCH9;
    };
    auto after_chunk = [&]() {
        ++q;
        qs = qs + 1 == NS ? 0 : qs + 1;
        if ((q & 1) == 0) { W16_STAMP(11); __syncthreads(); W16_STAMP(12); W16_STAMP_NEXT(); } else W16_STAMP(13);
    };
    // tile j on set C; P = the finished tile j - 1: its epilogue rides in the first chunk steps
    auto tile_step = [&](auto has_prev, f32x16 (&C)[NCI][NPI], const f32x16 (&P)[NCI][NPI]) {
        constexpr bool HP = decltype(has_prev)::value;
        if (HP) set_row_bases();
        if constexpr (NCS == 1) {
            if (HP) chunk_step(N9{}, T_{}, Z0{}, Z1{}, C, P, slot_a(), slot_w(0)); else chunk_step(N9{}, T_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(0));
            after_chunk();
        } else if constexpr (NCS == 2) {
            if (HP) chunk_step(N9{}, T_{}, Z0{}, Z2{}, C, P, slot_a(), slot_w(0)); else chunk_step(N9{}, T_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(0));
            after_chunk();
            if (HP) chunk_step(N9{}, F_{}, Z1{}, Z2{}, C, P, slot_a(), slot_w(1)); else chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(1));
            after_chunk();
            if (NCH == 3) {
                chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(2));
                after_chunk();
            }
        } else {
            if (HP) chunk_step(N9{}, T_{}, Z0{}, Z4{}, C, P, slot_a(), slot_w(0)); else chunk_step(N9{}, T_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(0));
            after_chunk();
            if (HP) chunk_step(N9{}, F_{}, Z1{}, Z4{}, C, P, slot_a(), slot_w(1)); else chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(1));
            after_chunk();
            if (HP) chunk_step(N9{}, F_{}, Z2{}, Z4{}, C, P, slot_a(), slot_w(2)); else chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(2));
            after_chunk();
            if (HP) chunk_step(N9{}, F_{}, Z3{}, Z4{}, C, P, slot_a(), slot_w(3)); else chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(3));
            after_chunk();
            for (int k = 4; k < NCH; ++k) {
                if (MIX && k >= n0) chunk_step(N1{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(k));
                else chunk_step(N9{}, F_{}, Z0{}, Z0{}, C, P, slot_a(), slot_w(k));
                after_chunk();
            }
        }
        if (HP) {                                                // P has left: the finished tile is now the one just accumulated
            p_x0 += TW;
            if (p_x0 >= A.W) { p_x0 = 0; p_y0 += TH; if (p_y0 >= A.H) { p_y0 = 0; ++p_n; } }
        }
    };
    // the last tile of the run: nothing left to hide behind
    auto serial_epilogue = [&](const f32x16 (&P)[NCI][NPI]) {
        if (OUT) {
            // every store of this form is the movers': the last tile goes through the out image too (E: their last reads of it are done)
            __syncthreads();
#pragma unroll
            for (int e = 0; e < NMO; ++e) micro(P, e, false);
            __syncthreads();                                     // F
            return;
        }
        set_row_bases();
#pragma unroll
        for (int e = 0; e < NMO; ++e) micro(P, e, true);
    };

    __syncthreads();                                             // B1: run chunks 0, 1 are staged
    tile_step(F_{}, accA, accB);
    int j = 1;
    for (; j + 1 < ntl; j += 2) {
        tile_step(T_{}, accB, accA);
        tile_step(T_{}, accA, accB);
    }
    const bool tail = j < ntl;
    if (tail) tile_step(T_{}, accB, accA);
    if (q & 1) __syncthreads();                                  // the last, half-filled interval
    for (int i = NI; i % IPG != 0; ++i) __syncthreads();         // the movers' loop runs whole groups of IPG intervals
    if (tail) serial_epilogue(accB); else serial_epilogue(accA);
#ifdef CDNET_WS_STAMPS
    if (stamp_on) { for (int i = 0; i < sn; ++i) g_ws16_stamps[i] = s_stamp[i]; g_ws16_stamps[sn] = 0; }
#endif
}

}  // namespace

namespace cdnet {

// eligibility + launch; returns -1 when the launch must take the older kernels
template <int BN>
static int try_launch_ws16(const ConvArgs &A, hipStream_t st, bool dry_run) {
    using L = W16Lds<BN>;
    int ctot = 0;
    bool all_plain = true;
    for (int i = 0; i < A.nsrc; ++i) {
        const ConvSrc &s = A.src[i];
        if (s.pool || s.relu < 0 || s.relu > 2) return -1;
        ctot += s.C;
        const long long rs = s.row_stride ? s.row_stride : (long long)s.Ws * s.C;
        if ((long long)A.N * s.Hs * rs * 2 >= (1LL << 31)) return -1;      // the movers' requests: 31-bit byte offsets
        all_plain = all_plain && !s.scale && !s.relu && !s.res && !s.f16;
    }
    const bool mix = A.taps1 == 1 && A.nsrc == 2;
    if (A.taps1 != 0 && !mix && A.taps1 != 9) return -1;
    const int n0 = A.src[0].C / 16, nch = A.nchunk, n1 = nch - n0;
    if (mix && n0 < 4) return -1;                                 // (the epilogue rides in the first four nine-tap steps)
    const int wres = mix ? L::wbytes(n0, n1) : L::wbytes(nch, 0);
    const bool stream = wres > L::RES_MAX;
    if (stream && (mix || (nch & 1))) return -1;
    // halo ring: four slots under the register movers; under the DMA movers (plain sources) seven - five chunks ahead of the consumers -
    // or six beside the one-tap chunks' extra weights
    const int ns = 4;
    static const int out_env = getenv("CDNET_WS16_OUT") ? atoi(getenv("CDNET_WS16_OUT")) : 1;
    // (the out-image form requests its halo by chunk PAIRS: every pair from one source)
    const bool out = out_env && nch >= 4 && !(nch & 1) && !(n0 & 1) && L::bytes(ns, stream ? 4 * L::WCH9 : wres, ctot, true) + W16_STAMP_BYTES <= 160 * 1024;
    const int smem = L::bytes(ns, stream ? 4 * L::WCH9 : wres, ctot, out) + W16_STAMP_BYTES;
    if (smem > 160 * 1024) return -1;
    if (A.pool_out && (!out || !A.orelu || A.out_coff || A.out_cstride != A.Cout)) return -1;      // the fused 2x2 max-pool rides in the movers' store path
    if (A.dot_out && (!out || stream || A.pool_out || A.Cout > BN || A.out_coff || !A.dot_w)) return -1;      // ... and so does the fused 1x1 classifier
    if (!A.out && !A.dot_out) return -1;
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
    // a persistent workgroup pays a few microseconds of start-up and a serial epilogue for its last tile
    if (!(A.debug & 64) && ((long long)T * nch < 16LL * Gmax || (T < Gmax && nch < 16))) return -1;
    int G = n_cu / ctiles;
    G = G > T ? T : G;
    if (G >= 8) G &= ~7;
    if ((A.debug >> 8) > 0 && (A.debug >> 8) < G) G = A.debug >> 8;      // tests: few workgroups, long runs of tiles
    if (G < 1) G = 1;
    if (dry_run) return CDNET_OK;
    dim3 grid(G, ctiles, 1);
#ifndef CDNET_WS16_PAIR_DEFAULT
#define CDNET_WS16_PAIR_DEFAULT 1
#endif
    auto go = [&](auto xf_c, auto sm_c, auto mx_c, auto ncs_c, auto pf_c) -> int {
        constexpr int XF = decltype(xf_c)::value;
        constexpr bool PAIR_ = CDNET_WS16_PAIR_DEFAULT != 0;    // (A/B builds: tools/build_variant.sh nopair "-DCDNET_WS16_PAIR_DEFAULT=0" conv16ws.hip)
        constexpr bool OUTOK = decltype(ncs_c)::value == 0;
        if constexpr (OUTOK) {
            if (out) {
                constexpr bool STREAM_ = decltype(sm_c)::value;
                constexpr bool MIX_ = decltype(mx_c)::value;
                // the streamed-weight launches - the matrix-bound layers (128+ input channels) - take the 16x16x32 consumers:
                // 63 / 62 / 27 us against 72 / 70 / 30 us on 256 -> 256 @64², 512 -> 512 @32², 320 -> 64 @64².  The resident 64 -> 64 layer is
                // bound by power and does not gain (round 5, without spills and with pair requests: 74.0 vs 72.5 us at 16 tiles, 298 vs 297.5 us at
                // 64 - profiles/HISTORY.md); the one-tap form spilled.
                if constexpr (STREAM_) {
                    auto kern_k = conv_ws16_kernel<BN, XF, STREAM_, MIX_, 0, 4, decltype(pf_c)::value, true, true, PAIR_>;
                    static bool attr_k = false;
                    if (!attr_k) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                            return check_launch("hipFuncSetAttribute(conv_ws16 k32)");
                        attr_k = true;
                    }
                    kern_k<<<grid, 512, smem, st>>>(A);
                    return check_launch("conv_ws16_kernel(k32)");
                }
#ifndef CDNET_WS16_QUAD_DEFAULT
#define CDNET_WS16_QUAD_DEFAULT 1
#endif
                // resident weights, every source a multiple of 64 channels, tensors beyond the 256 MB Infinity Cache (the 64-tile inference
                // launches): whole 128-byte pieces of a pixel per request interval (QUAD) - HBM reads of the dominant layer 0.88 -> 0.69 GB per
                // 64 tiles (1.32 -> 1.14 x the algorithmic traffic), the launch time unchanged on a box bound by power (310 vs 302-309 us); at 16
                // tiles, where the L2 / Infinity Cache keep the line between its two pairs anyway, the pair form is 1-2 % faster and stays
                const long long io_bytes = 2LL * A.N * A.H * A.W * (ctot + A.Cout);
                if constexpr (!STREAM_ && PAIR_ && CDNET_WS16_QUAD_DEFAULT != 0) if (!(n0 & 3) && !(n1 & 3) && (io_bytes >= (512LL << 20) || (A.debug & 16))) {      // (debug bit 16: tests - the quad form on small launches)
                    auto kern_q = conv_ws16_kernel<BN, XF, STREAM_, MIX_, 0, 4, decltype(pf_c)::value, true, false, true, true>;
                    static bool attr_q = false;
                    if (!attr_q) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_q), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                            return check_launch("hipFuncSetAttribute(conv_ws16 quad)");
                        attr_q = true;
                    }
                    kern_q<<<grid, 512, smem, st>>>(A);
                    return check_launch("conv_ws16_kernel(quad)");
                }
                auto kern_o = conv_ws16_kernel<BN, XF, STREAM_, MIX_, 0, 4, decltype(pf_c)::value, true, false, PAIR_>;
                static bool attr_o = false;
                if (!attr_o) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_o), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                        return check_launch("hipFuncSetAttribute(conv_ws16 out)");
                    attr_o = true;
                }
                kern_o<<<grid, 512, smem, st>>>(A);
                return check_launch("conv_ws16_kernel(out)");
            }
        }
        constexpr bool STREAM = decltype(sm_c)::value;
        constexpr bool MIX = decltype(mx_c)::value;
        constexpr int NCS = decltype(ncs_c)::value;
        constexpr int PFD = decltype(pf_c)::value;
        constexpr int NS = 4;
        auto kern = conv_ws16_kernel<BN, XF, STREAM, MIX, NCS, NS, PFD, false>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return check_launch("hipFuncSetAttribute(conv_ws16)");
            attr_done = true;
        }
        kern<<<grid, 512, smem, st>>>(A);
        return check_launch("conv_ws16_kernel");
    };
    using X0 = std::integral_constant<int, 0>;
    using X2 = std::integral_constant<int, 2>;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using F_ = std::false_type;
    using T_ = std::true_type;
    auto by_xf = [&](auto sm_c, auto mx_c, auto ncs_c) -> int {
        if (!all_plain) return go(X2{}, sm_c, mx_c, ncs_c, std::integral_constant<int, 4>{});
        return go(X0{}, sm_c, mx_c, ncs_c, std::integral_constant<int, 4>{});
    };
    if (stream) return by_xf(T_{}, F_{}, C0{});                  // (an even chunk count >= 6)
    if (mix) return by_xf(F_{}, T_{}, C0{});
    if (nch == 1) return by_xf(F_{}, F_{}, C1{});
    if (nch <= 3) return by_xf(F_{}, F_{}, C2{});
    return by_xf(F_{}, F_{}, C0{});
}

// called by cdnet_conv_forward first (16-bit path); -1 = not eligible
int conv_forward_ws16(const ConvArgs &A, hipStream_t st, bool dry_run) {
    if ((A.debug & 32) || (A.debug & 128)) return -1;      // (128: tests - the older persistent kernel instead)
    if (A.f32 || A.taps != 9 || A.npar != 1 || A.ostride != 1 || A.tile != 16 || A.CK != 16 || A.eres || A.ws || A.stats || A.oscale || A.out_f16) return -1;
    if (A.H % 16 != 0 || A.W % 16 != 0 || A.nchunk < 1 || A.Cout % 16 != 0) return -1;
    if (A.BN == 64) return try_launch_ws16<64>(A, st, dry_run);
    if (A.BN == 32) return try_launch_ws16<32>(A, st, dry_run);
    return -1;
}

}  // namespace cdnet
