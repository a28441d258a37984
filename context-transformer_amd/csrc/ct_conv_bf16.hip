// libctdet: bf16 channels-last convolution on the bf16 MFMA path (BASELINE.json configs[4]: "RFBNet-512 bf16 MFMA
// convs + fp32 NMS").  Same arithmetic contract as ct_conv2d_fwd (models/RFB_Net_vgg.py:7-22 BasicConv and the plain
// Conv2d layers: y = act(conv(x) * scale + shift [* res_scale + res])), different storage: activations are
// [batch][h][w][channels] bf16, so the 8 consecutive input channels an MFMA lane needs at one filter tap are ONE
// 16-byte load, and the implicit-GEMM "im2col" gather degenerates to a row copy.  Accumulation and epilogue in fp32.
//
//   workgroup (256 threads, 4 waves) = 128 output pixels x 128 (or 64) output channels, or (512 threads, 8 waves) 256 x 256
//   for layers with >= 256 output channels whose 256-pixel tiles still give every CU a workgroup; a k-step = 64 (or 32)
//   input channels of one filter tap (for the 3-channel image, stored with 8 channels: 4 taps x 8 channels).  The A tile
//   [pixels][k] and the B tile [cout][k] go global -> LDS by DMA (buffer_load_dwordx4 ... lds; round 3 -- the VGPR-staged
//   form with ds_write_b128 remains for the image layer), XOR-swizzled 16-byte segments instead of padded rows, a ring of
//   two (three for 32-channel steps) buffers, one barrier per k-step, operand fragments of k-slice h+1 read while the
//   v_mfma_f32_32x32x16_bf16 of slice h run; wave tile 64 x 64 (128 x 64 in the big tile).  XCD-aware tile order: the cout
//   tiles of one pixel tile share an L2.
//   Epilogue: bf16 NHWC output goes through LDS once ([pixel][cout] fp32) so that a thread owns 8 consecutive
//   channels of a pixel -- one 16-byte store and one 16-byte residual load (2-byte accesses made the 1x1 layers
//   epilogue-bound: 270 -> 77 us for 512->512 @38x38); a channel slice of a wider buffer is torch.cat for free.
//   Multibox heads write fp32 into the flattened head buffers, which ARE channels-last
//   (models/RFB_Net_vgg.py:245-247).
//   Measured (bs 32, tools/bf16_probe.py): 512->512 @38x38 280-290 us = 750-780 TFLOP/s (30 % of the 2.5 PFLOP/s dense bf16
//   peak), 256->256 @75x75 802 TFLOP/s on the 256 x 256 tile (740 on 128 x 128).  What bounds it (ablations, same layer):
//   MFMAs + fragment reads alone 147 us (59 %); every KB a wave moves costs ~100 issue cycles whichever way it goes
//   (DMA piece, or buffer_load + ds_write_b128), and a 128 x 128 tile moves 8 KB per wave per 512 MFMA cycles.
#include "ct_common.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_set>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int KSMALL = 8;                      // cin <= 8 (the image): one k-step = 4 taps x 8 channels

struct Bf16Args {
    const void* in;
    const void* w;           // [taps][cin_pad / 8][cout_pad][8] bf16
    const float* scale;
    const float* shift;
    const float* lo;
    const void* res;         // bf16 NHWC
    void* out;               // bf16 NHWC (nseg == 0)
    unsigned in_bytes, w_bytes;
    int Cin, cin_pad, H, W, in_ctot, in_coff;
    int M, cout_pad;
    int KH, KW, stride, pad_h, pad_w, dil;
    int OW, OHW, Npix;
    int out_ctot, out_coff, res_ctot, res_coff;
    float res_scale;
    int relu, nseg;
    ct_out_segment seg[3];
    int tiles_m;
    int ksplit, steps_per_split;    // > 1: blockIdx.y owns k-steps [y*sps, (y+1)*sps) and writes raw sums to its slab
    float* ws;                      // [ksplit][Npix][M] fp32, reduced in order by conv_bf16_splitk_epilogue
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__device__ __forceinline__ unsigned short f2bf(float f)      // round to nearest even (torch's float -> bfloat16)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x7FFFFFu)) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// BN: output channels per workgroup (128: 2x2 waves of 64x64; 64: 4x1 waves of 32 px x 64 couts)
// BK: input channels (of one tap) per k-step / barrier;  SMALL: cin_pad == 8, a k-step is 4 taps x 8 channels
// BM x BN = output pixels x output channels of a workgroup of NT threads: 128 x {64, 128} with four waves, 256 x 256
// with eight (each wave 128 x 64: eight 32x32 accumulators; half the L2 -> LDS operand traffic per MFMA of 128 x 128,
// which at 64 flop per byte asks for ~12 TB/s at the 0.75 PFLOP/s that tile tops out at)
template <int BM, int BN, int BK, bool SMALL, int NT>
__global__ __launch_bounds__(NT) void conv_bf16_nhwc(const Bf16Args a)
{
    constexpr int SPR = BK / 8;                    // 16-byte segments per tile row
    // DMA (every variant but the 8-channel image layer): the tiles go global -> LDS directly (buffer_load_dwordx4 ...
    // lds, one KiB per wave instruction, lane i at M0 + 16 i), no VGPR staging and no ds_write_b128 -- ablation on
    // 512 -> 512 @38x38 bs 32: 300 us with the staging, 231 without the global loads, 153 without the LDS stores (MFMAs +
    // fragment reads alone: 147).  A DMA image is lane-contiguous, so rows are BK * 2 bytes without padding and the
    // 16-byte segments are XOR swizzled instead: segment s of row r sits at s ^ ((r / RL) % SPR), RL = rows per
    // 256-byte LDS line, which makes both the DMA write and the ds_read_b128 of a fragment (16 rows of one segment per
    // lane group) conflict-free.  The swizzle is applied on the GLOBAL side: the lane that owns physical slot (r, p)
    // fetches segment p ^ ((r / RL) % SPR).  The buffers form a ring of NBUF: step t lands in buffer t % NBUF while
    // steps t - NBUF + 1 .. t - 1 compute (BK = 32: three 16 KB buffers, two steps in flight, three workgroups per CU).
    constexpr bool DMA = !SMALL;
    constexpr int NBUF = DMA && BK == 32 ? 3 : 2;
    constexpr int ROWB = DMA ? BK * 2 : BK * 2 + 16;   // LDS row: padded by 16 bytes on the staged path (conflict-free 16-byte accesses)
    constexpr int RL = 256 / (BK * 2);                // DMA: rows per 256-byte line (SPR * RL == 16)
    constexpr int A_B = BM * ROWB, B_B = BN * ROWB;
    constexpr int NA = BM * SPR / NT, NB = BN * SPR / NT;
    constexpr int WAVES_M = BN == 64 ? 4 : 2;      // waves along the pixel dimension
    constexpr int WM = BM / WAVES_M, WN = BN / (NT / 64 / WAVES_M);
    constexpr int TM = WM / 32, TN = WN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; every XCD gets a contiguous chunk of the (cout tile
    // fastest) sequence, so the workgroups that share a pixel tile (same A rows) share an L2
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (wg % a.tiles_m) * BN;                  // cout tile
    const int n0 = (wg / a.tiles_m) * BM;                  // pixel tile

    // ---- staging role: 16-byte segment (tid + 256 q) of each tile
    int pix_n[NA], pix_h[NA], pix_w[NA];
    bool pix_ok[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int P = n0 + (tid + NT * q) / SPR;
        pix_ok[q] = P < a.Npix;
        const int Pc = pix_ok[q] ? P : 0;
        pix_n[q] = Pc / a.OHW;
        const int s = Pc - pix_n[q] * a.OHW;
        const int oh = s / a.OW;
        pix_h[q] = oh * a.stride - a.pad_h;
        pix_w[q] = (s - oh * a.OW) * a.stride - a.pad_w;
    }
    const int sseg = DMA ? (tid % SPR) ^ ((tid >> 4) % SPR) : tid % SPR;     // the segment this thread FETCHES (its LDS slot is tid % SPR)
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes), rw = make_rsrc(a.w, a.w_bytes);
    const int cgroups = a.cin_pad >> 3;                    // 8-channel groups per tap in the packed weights
    const int taps = a.KH * a.KW;
    const int csteps = SMALL ? 1 : a.cin_pad / BK;
    const int nsteps = SMALL ? (taps + SPR - 1) / SPR : taps * csteps;

    i32x4 ra[NA], rb[NB];
    // Loader state (round 3).  Steps are visited in increasing order, so the (tap, channel step) pair advances
    // incrementally: the per-lane part of every address (this lane's pixels at the current tap, its weight rows) lives in
    // VGPRs that change only when the tap does, the channel step is a wave-uniform scalar offset.  The first version
    // recomputed tap / pixel / bounds for every load (~120 VALU per 16 MFMAs; VALU never overlaps an MFMA on a SIMD,
    // profiles/r03_mfma_valu_overlap.txt).
    int ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_cs = 0;       // non-SMALL: tap index, its (kh, kw), channel step within the tap
    int va[NA], va_tail[NA], vb[NB];
    const bool has_tail = !SMALL && (a.Cin % BK) != 0;      // last channel step: segments at or past Cin read zeros
    auto tap_offsets = [&]() {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int ih = pix_h[q] + ld_kh * a.dil, iw = pix_w[q] + ld_kw * a.dil;
            const bool ok = pix_ok[q] && ld_tap < taps && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const int off = (((pix_n[q] * a.H + ih) * a.W + iw) * a.in_ctot + a.in_coff + sseg * 8) * 2;
            va[q] = ok ? off : kInvalidOff;
            va_tail[q] = ok && (csteps - 1) * BK + sseg * 8 < a.Cin ? off : kInvalidOff;
        }
    };
    if (!SMALL) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int co = m0 + (tid + NT * q) / SPR;
            vb[q] = co < a.cout_pad ? (sseg * a.cout_pad + co) * 16 : kInvalidOff;
        }
    }
    auto seek_step = [&](int step) {                        // position the loader on `step` (once, before the first load)
        ld_tap = step / csteps;
        ld_cs = step - ld_tap * csteps;
        ld_kh = ld_tap / a.KW;
        ld_kw = ld_tap - ld_kh * a.KW;
        tap_offsets();
    };
    int dma_buf = 0;                                       // DMA: the LDS buffer the next load_step fills
    auto load_step = [&](int step) {
        if (SMALL) {
            const int tap = step * SPR + sseg;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            const bool cok = 0 < a.Cin && tap < taps;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int ih = pix_h[q] + kh * a.dil, iw = pix_w[q] + kw * a.dil;
                const bool ok = pix_ok[q] && cok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                const int off = ok ? (((pix_n[q] * a.H + ih) * a.W + iw) * a.in_ctot + a.in_coff) * 2 : kInvalidOff;
                ra[q] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int co = m0 + (tid + NT * q) / SPR;
                const int woff = (co < a.cout_pad && tap < taps) ? ((tap * cgroups * a.cout_pad + co) * 8) * 2 : kInvalidOff;
                rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff, 0, 0);
            }
            return;
        }
        const int sa = ld_cs * BK * 2;                                      // bytes: channel step inside the pixel
        const int sb = (ld_tap * cgroups + ld_cs * SPR) * a.cout_pad * 16;   // bytes: (tap, channel group) plane
        const bool tail = has_tail && ld_cs == csteps - 1;
        if constexpr (DMA) {
            typedef __attribute__((address_space(3))) void* lds_ptr;
            unsigned char* Ad = lds + dma_buf * (A_B + B_B) + wave * 1024;
            unsigned char* Bd = Ad + A_B;
            (void)Ad; (void)Bd;
#if defined(__HIP_DEVICE_COMPILE__)         // the host pass of hipcc has no such builtin (and would drop the kernel's launch stub)
#pragma unroll
            for (int q = 0; q < NA; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(Ad + q * (NT * 16)), 16, tail ? va_tail[q] : va[q], sa, 0, 0);
#pragma unroll
            for (int q = 0; q < NB; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bd + q * (NT * 16)), 16, vb[q], sb, 0, 0);
#endif
        } else {
#pragma unroll
            for (int q = 0; q < NA; ++q) ra[q] = __builtin_amdgcn_raw_buffer_load_b128(rin, tail ? va_tail[q] : va[q], sa, 0);
#pragma unroll
            for (int q = 0; q < NB; ++q) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, vb[q], sb, 0);
        }
        // advance to the next step
        ++ld_cs;
        if (ld_cs == csteps) {
            ld_cs = 0;
            ++ld_tap;
            ++ld_kw;
            if (ld_kw == a.KW) { ld_kw = 0; ++ld_kh; }
            tap_offsets();
        }
    };
    auto store_step = [&](int buf) {
        unsigned char* A = lds + buf * (A_B + B_B);
        unsigned char* B = A + A_B;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            *reinterpret_cast<i32x4*>(A + ((tid + NT * q) / SPR) * ROWB + sseg * 16) = ra[q];
#pragma unroll
        for (int q = 0; q < NB; ++q)
            *reinterpret_cast<i32x4*>(B + ((tid + NT * q) / SPR) * ROWB + sseg * 16) = rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int s0 = blockIdx.y * a.steps_per_split;
    const int s1 = min(nsteps, s0 + a.steps_per_split);
    if (!SMALL) seek_step(s0);
    if constexpr (DMA) {
        // a wave waits for its own pieces of the NEXT step (vmcnt: everything but the pieces of the steps after it) before
        // the barrier that publishes that buffer
        constexpr int PF = NBUF - 1;                       // steps in flight
        constexpr int PIECES = NA + NB;                    // DMA instructions per step and wave
        int nload = s0;                                    // next step to load, its buffer
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (nload < s1) { load_step(nload); ++nload; dma_buf = dma_buf + 1 == NBUF ? 0 : dma_buf + 1; }
        if (PF == 2 && nload - s0 == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // bare barriers in this loop: __syncthreads() carries a workgroup fence, which the compiler emits as s_waitcnt vmcnt(0)
        // -- it drained the DMA ring at every k-step and made the counted wait above meaningless (rounds 3-4 ran that way)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int xa = (((wm0 + l31) / RL) % SPR) * 16, xb = (((wn0 + l31) / RL) % SPR) * 16;
        int buf = 0;
        for (int step = s0; step < s1; ++step) {
            const unsigned char* A = lds + buf * (A_B + B_B) + (wm0 + l31) * ROWB;
            const unsigned char* B = lds + buf * (A_B + B_B) + A_B + (wn0 + l31) * ROWB;
            i32x4 fa[2][TM], fb[2][TN];
            auto read_frag = [&](int h, int slot) {
                const int oa = (h * 32 + kg * 16) ^ xa, ob = (h * 32 + kg * 16) ^ xb;
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[slot][t] = *reinterpret_cast<const i32x4*>(A + t * 32 * ROWB + oa);
#pragma unroll
                for (int t = 0; t < TN; ++t) fb[slot][t] = *reinterpret_cast<const i32x4*>(B + t * 32 * ROWB + ob);
            };
            read_frag(0, 0);
            const bool more = nload < s1;                  // buffer dma_buf held step - 1: every wave is past that barrier
            if (more) { load_step(nload); ++nload; dma_buf = dma_buf + 1 == NBUF ? 0 : dma_buf + 1; }
#pragma unroll
            for (int h = 0; h < BK / 16; ++h) {
                const int cur = h & 1;
                if (h + 1 < BK / 16) read_frag(h + 1, cur ^ 1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][i]),
                                                                            __builtin_bit_cast(bf16x8, fb[cur][j]),
                                                                            acc[i][j], 0, 0, 0);
            }
            // the next step's pieces must have landed; with two steps in flight the newest step's may still fly
            if (PF == 2 && more && nload - step > 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        }
    } else {
        load_step(s0);
        store_step(0);
        if (s1 - s0 > 1) load_step(s0 + 1);
        __syncthreads();
        for (int step = s0; step < s1; ++step) {
            const int buf = (step - s0) & 1;
            const unsigned char* A = lds + buf * (A_B + B_B) + (wm0 + l31) * ROWB + kg * 16;
            const unsigned char* B = lds + buf * (A_B + B_B) + A_B + (wn0 + l31) * ROWB + kg * 16;
            // operand fragments of k-slice h+1 are read while the MFMAs of slice h run
            i32x4 fa[2][TM], fb[2][TN];
            auto read_frag = [&](int h, int slot) {
    #pragma unroll
                for (int t = 0; t < TM; ++t) fa[slot][t] = *reinterpret_cast<const i32x4*>(A + t * 32 * ROWB + h * 32);
    #pragma unroll
                for (int t = 0; t < TN; ++t) fb[slot][t] = *reinterpret_cast<const i32x4*>(B + t * 32 * ROWB + h * 32);
            };
            read_frag(0, 0);
            if (step + 1 < s1) store_step(buf ^ 1);
            if (step + 2 < s1) load_step(step + 2);
    #pragma unroll
            for (int h = 0; h < BK / 16; ++h) {
                const int cur = h & 1;
                if (h + 1 < BK / 16) read_frag(h + 1, cur ^ 1);
    #pragma unroll
                for (int i = 0; i < TM; ++i)
    #pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][i]),
                                                                            __builtin_bit_cast(bf16x8, fb[cur][j]),
                                                                            acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: acc[i][j][r] = pixel n0 + wm0 + 32 i + (r&3) + 8 (r>>2) + 4 kg, cout m0 + wn0 + 32 j + l31
    if (BM != 256 && a.ksplit > 1) {        // (the 256 x 256 tile is only launched unsplit, with the bf16 NHWC output)
        // split-K (maps too small to fill the chip): raw partial sums to this split's slab, no atomics;
        // conv_bf16_splitk_epilogue adds the slabs in order and applies the epilogue
        float* const slab = a.ws + (size_t)blockIdx.y * a.Npix * a.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = m0 + wn0 + 32 * j + l31;
            if (co >= a.M) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int P = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (P < a.Npix) slab[(size_t)P * a.M + co] = acc[i][j][r];
                }
        }
        return;
    }
    if (BM != 256 && a.nseg > 0) {
        // multibox heads: fp32, channels-last per segment -> a lane's cout is already the fast index
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = m0 + wn0 + 32 * j + l31;
            if (co >= a.M) continue;
            const float sc = a.scale[co], sh = a.shift[co];
            const float lo = a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int P = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (P >= a.Npix) continue;
                    float v = acc[i][j][r] * sc + sh;
                    v = v < lo ? lo : v;            // NaN propagates
                    const int n = P / a.OHW, s = P - n * a.OHW;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end)
                            a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                         (size_t)s * a.seg[g].pix_stride + (co - a.seg[g].co_begin)] = v;
                }
        }
        return;
    }
    // bf16 NHWC output: transpose the tile through LDS ([pixel][cout] fp32, operand buffers are free after the last
    // barrier) so that a thread owns 8 consecutive channels of a pixel: one 16-byte store (and one 16-byte residual
    // load) instead of eight 2-byte ones
    constexpr int SROW = BN * 4 + 16;                       // bytes per staged pixel row
    // pixel rows staged per pass: all of them, or 64 at a time where that keeps the workgroup's LDS at its operand ring
    // (BK = 32: 48 KB, three workgroups per CU)
    constexpr int RP = BM * SROW <= NBUF * (A_B + B_B) || !(DMA && BK == 32) ? BM : 64;
    static_assert(RP * SROW <= NBUF * (A_B + B_B) || RP == BM, "staging pass larger than the operand ring");
    float* const stage = reinterpret_cast<float*>(lds);
    constexpr int GPR = BN / 8;                             // 8-channel groups per pixel row
    const bool vec_ok = ((a.out_ctot | a.out_coff) & 7) == 0 && (!a.res || ((a.res_ctot | a.res_coff) & 7) == 0);
#pragma unroll
    for (int pass = 0; pass < BM / RP; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn0 + 32 * j + l31;
        const int co = m0 + col;
        const float sc = co < a.M ? a.scale[co] : 0.f, sh = co < a.M ? a.shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if ((wm0 + 32 * i) / RP != pass) continue;      // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg - pass * RP;
                stage[row * (SROW / 4) + col] = acc[i][j][r] * sc + sh;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < RP * GPR; e += NT) {
        const int row = e / GPR, g8 = e - row * GPR;
        const int P = n0 + pass * RP + row, co = m0 + g8 * 8;
        if (P >= a.Npix || co >= a.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(stage) + row * SROW + g8 * 32);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(stage) + row * SROW + g8 * 32 + 16);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const bool full = co + 8 <= a.M && vec_ok;
        if (a.res) {
            const unsigned short* rp = reinterpret_cast<const unsigned short*>(a.res) + (size_t)P * a.res_ctot + a.res_coff + co;
            if (full) {
                const i32x4 rr = *reinterpret_cast<const i32x4*>(rp);
                const unsigned rw4[4] = {(unsigned)rr.x, (unsigned)rr.y, (unsigned)rr.z, (unsigned)rr.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = v[2 * q] * a.res_scale + __uint_as_float(rw4[q] << 16);
                    v[2 * q + 1] = v[2 * q + 1] * a.res_scale + __uint_as_float(rw4[q] & 0xFFFF0000u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (co + q < a.M) v[q] = v[q] * a.res_scale + bf2f(rp[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float lo = a.lo ? (co + q < a.M ? a.lo[co + q] : 0.f) : (a.relu ? 0.f : -INFINITY);
            v[q] = v[q] < lo ? lo : v[q];
        }
        unsigned short* op = reinterpret_cast<unsigned short*>(a.out) + (size_t)P * a.out_ctot + a.out_coff + co;
        if (full) {
            i32x4 o;
            o.x = (int)((unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16));
            o.y = (int)((unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
            o.z = (int)((unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16));
            o.w = (int)((unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16));
            *reinterpret_cast<i32x4*>(op) = o;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (co + q < a.M) op[q] = f2bf(v[q]);
        }
    }
    }
}

// finishing kernel of a split-K bf16 convolution: slabs summed in split order, then the fused epilogue's arithmetic
__global__ __launch_bounds__(256) void conv_bf16_splitk_epilogue(const Bf16Args a)
{
    const int total = a.Npix * a.M;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int P = idx / a.M, co = idx - P * a.M;
        float sum = a.ws[idx];
        for (int k = 1; k < a.ksplit; ++k) sum += a.ws[(size_t)k * total + idx];
        float v = sum * a.scale[co] + a.shift[co];
        if (a.res)
            v = v * a.res_scale +
                bf2f(reinterpret_cast<const unsigned short*>(a.res)[(size_t)P * a.res_ctot + a.res_coff + co]);
        { const float fl = a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY); v = v < fl ? fl : v; }
        if (a.nseg == 0) {
            reinterpret_cast<unsigned short*>(a.out)[(size_t)P * a.out_ctot + a.out_coff + co] = f2bf(v);
        } else {
            const int n = P / a.OHW, s = P - n * a.OHW;
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end)
                    a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                 (size_t)s * a.seg[g].pix_stride + (co - a.seg[g].co_begin)] = v;
        }
    }
}

// weights: nparts tensors [cout_i][cin][kh][kw] fp32 -> [tap][cin_pad/8][cout_pad][8] bf16 (zero padded)
struct PackBf16Args {
    const float* w[6];
    int mbeg[7];
    int nparts, cin, cin_pad, cout_pad, taps;
    unsigned short* out;
};

__global__ void pack_bf16_kernel(const PackBf16Args p)
{
    const long total = (long)p.taps * p.cin_pad * p.cout_pad;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx & 7);
        long t = idx >> 3;
        const int co = (int)(t % p.cout_pad);
        t /= p.cout_pad;
        const int cg = (int)(t % (p.cin_pad >> 3));
        const int tap = (int)(t / (p.cin_pad >> 3));
        const int ci = cg * 8 + c8;
        float v = 0.f;
        if (ci < p.cin && co < p.mbeg[p.nparts]) {
            int part = 0;
            while (co >= p.mbeg[part + 1]) ++part;
            v = p.w[part][((size_t)(co - p.mbeg[part]) * p.cin + ci) * p.taps + tap];
        }
        p.out[idx] = f2bf(v);
    }
}

// NCHW fp32 -> NHWC bf16 (channels zero padded to c_pad) and back (for tests / the image input)
__global__ void nchw_f32_to_nhwc_bf16(const float* __restrict__ x, int batch, int C, int HW, int c_pad,
                                      unsigned short* __restrict__ y)
{
    const long total = (long)batch * HW * c_pad;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c_pad);
        const long p = idx / c_pad;
        const int n = (int)(p / HW), s = (int)(p - (long)n * HW);
        y[idx] = c < C ? f2bf(x[((size_t)n * C + c) * HW + s]) : (unsigned short)0;
    }
}

__global__ void nhwc_bf16_to_nchw_f32(const unsigned short* __restrict__ y, int batch, int C, int HW, int ctot,
                                      int coff, float* __restrict__ x)
{
    const long total = (long)batch * C * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int s = (int)(idx % HW);
        const long t = idx / HW;
        const int c = (int)(t % C), n = (int)(t / C);
        x[idx] = bf2f(y[((size_t)n * HW + s) * ctot + coff + c]);
    }
}

// MaxPool2d on NHWC bf16 (floor / ceil via the output size), -inf padding like nn.MaxPool2d
__global__ void maxpool_nhwc_bf16(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int batch,
                                  int C, int H, int W, int OH, int OW, int k, int stride, int pad)
{
    const long total = (long)batch * OH * OW * C;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long t = idx / C;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH), n = (int)(t / OH);
        float m = -INFINITY;
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    m = fmaxf(m, bf2f(x[(((size_t)n * H + ih) * W + iw) * C + c]));
            }
        y[idx] = f2bf(m);
    }
}

// the same pool, 8 channels (16 bytes) per thread
__global__ void maxpool_nhwc_bf16_v8(const i32x4* __restrict__ x, i32x4* __restrict__ y, int batch, int C8, int H,
                                     int W, int OH, int OW, int k, int stride, int pad)
{
    const long total = (long)batch * OH * OW * C8;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8);
        long t = idx / C8;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH), n = (int)(t / OH);
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -INFINITY;
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
                if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
                const i32x4 v = x[(((size_t)n * H + ih) * W + iw) * C8 + c];
                const unsigned w4[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    m[2 * q] = fmaxf(m[2 * q], __uint_as_float(w4[q] << 16));
                    m[2 * q + 1] = fmaxf(m[2 * q + 1], __uint_as_float(w4[q] & 0xFFFF0000u));
                }
            }
        i32x4 o;      // the maxima are bf16 values already: truncation is exact
        o.x = (int)((__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xFFFF0000u));
        o.y = (int)((__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xFFFF0000u));
        o.z = (int)((__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xFFFF0000u));
        o.w = (int)((__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xFFFF0000u));
        y[idx] = o;
    }
}

inline int grid1d(long total) { return (int)std::min<long>((total + 255) / 256, 256 * 32); }

}  // namespace

extern "C" int ct_conv_bf16_cin_pad(int cin) { return cin <= KSMALL ? KSMALL : (cin + 63) / 64 * 64; }
extern "C" int ct_conv_bf16_cout_pad(int cout) { return (cout + 7) / 8 * 8; }

extern "C" size_t ct_conv_bf16_packed_elems(int cin, int cout, int kh, int kw)
{
    return (size_t)kh * kw * ct_conv_bf16_cin_pad(cin) * ct_conv_bf16_cout_pad(cout);
}

extern "C" int ct_conv_pack_weights_bf16(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                                         void* wpacked, ct_stream_t stream)
{
    CT_REQUIRE(w && cout && wpacked && nparts >= 1 && nparts <= 6, "ct_conv_pack_weights_bf16: bad arguments");
    PackBf16Args p{};
    p.nparts = nparts;
    p.mbeg[0] = 0;
    for (int i = 0; i < nparts; ++i) {
        CT_REQUIRE(w[i] && cout[i] > 0, "ct_conv_pack_weights_bf16: null part");
        p.w[i] = w[i];
        p.mbeg[i + 1] = p.mbeg[i] + cout[i];
    }
    p.cin = cin; p.cin_pad = ct_conv_bf16_cin_pad(cin); p.cout_pad = ct_conv_bf16_cout_pad(p.mbeg[nparts]);
    p.taps = kh * kw;
    p.out = static_cast<unsigned short*>(wpacked);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid1d((long)p.taps * p.cin_pad * p.cout_pad)), dim3(256), 0,
                       ctdet::as_stream(stream), p);
    CT_LAUNCH_CHECK("pack_bf16_kernel");
    return CT_OK;
}

extern "C" int ct_nchw_f32_to_nhwc_bf16(const float* x, int batch, int channels, int hw, int c_pad, void* y,
                                        ct_stream_t stream)
{
    CT_REQUIRE(x && y && batch > 0 && channels > 0 && hw > 0 && c_pad >= channels, "ct_nchw_f32_to_nhwc_bf16: bad arguments");
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16, dim3(grid1d((long)batch * hw * c_pad)), dim3(256), 0,
                       ctdet::as_stream(stream), x, batch, channels, hw, c_pad, static_cast<unsigned short*>(y));
    CT_LAUNCH_CHECK("nchw_f32_to_nhwc_bf16");
    return CT_OK;
}

extern "C" int ct_nhwc_bf16_to_nchw_f32(const void* y, int batch, int channels, int hw, int ctot, int coff, float* x,
                                        ct_stream_t stream)
{
    CT_REQUIRE(x && y && batch > 0 && channels > 0 && hw > 0 && coff >= 0 && coff + channels <= ctot,
               "ct_nhwc_bf16_to_nchw_f32: bad arguments");
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32, dim3(grid1d((long)batch * hw * channels)), dim3(256), 0,
                       ctdet::as_stream(stream), static_cast<const unsigned short*>(y), batch, channels, hw, ctot,
                       coff, x);
    CT_LAUNCH_CHECK("nhwc_bf16_to_nchw_f32");
    return CT_OK;
}

extern "C" int ct_maxpool2d_nhwc_bf16(const void* x, void* y, int batch, int channels, int h, int w, int oh, int ow,
                                      int k, int stride, int pad, ct_stream_t stream)
{
    CT_REQUIRE(x && y && batch > 0 && channels > 0, "ct_maxpool2d_nhwc_bf16: bad arguments");
    if (channels % 8 == 0) {
        hipLaunchKernelGGL(maxpool_nhwc_bf16_v8, dim3(grid1d((long)batch * oh * ow * (channels / 8))), dim3(256), 0,
                           ctdet::as_stream(stream), static_cast<const i32x4*>(x), static_cast<i32x4*>(y), batch,
                           channels / 8, h, w, oh, ow, k, stride, pad);
        CT_LAUNCH_CHECK("maxpool_nhwc_bf16_v8");
        return CT_OK;
    }
    hipLaunchKernelGGL(maxpool_nhwc_bf16, dim3(grid1d((long)batch * oh * ow * channels)), dim3(256), 0,
                       ctdet::as_stream(stream), static_cast<const unsigned short*>(x),
                       static_cast<unsigned short*>(y), batch, channels, h, w, oh, ow, k, stride, pad);
    CT_LAUNCH_CHECK("maxpool_nhwc_bf16");
    return CT_OK;
}

extern "C" int ct_conv2d_bf16_fwd(const ct_conv_desc* d, ct_stream_t stream)
{
    CT_REQUIRE(d && d->in && d->wpacked && d->scale && d->shift, "ct_conv2d_bf16_fwd: null tensor");
    CT_REQUIRE(d->batch > 0 && d->cin > 0 && d->cout > 0 && !d->transposed, "ct_conv2d_bf16_fwd: bad shape");
    CT_REQUIRE(d->in_coff % 8 == 0 && d->in_ctot % 8 == 0 && d->cin % 8 == 0,
               "ct_conv2d_bf16_fwd: channel slices must be multiples of 8 (16-byte loads), got ctot %d coff %d cin %d",
               d->in_ctot, d->in_coff, d->cin);
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_bf16_fwd: input slice");
    const int eoh = (d->h + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int eow = (d->w + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    CT_REQUIRE(eoh == d->oh && eow == d->ow, "ct_conv2d_bf16_fwd: oh/ow mismatch");
    if (d->nseg == 0)
        CT_REQUIRE(d->out && d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "ct_conv2d_bf16_fwd: output slice");
    else
        CT_REQUIRE(d->nseg <= 3 && !d->res, "ct_conv2d_bf16_fwd: segments");
    const long long in_bytes = (long long)d->batch * d->h * d->w * d->in_ctot * 2;
    CT_REQUIRE(in_bytes < kMaxBufBytes, "ct_conv2d_bf16_fwd: input above 2 GiB");
    Bf16Args a{};
    a.in = d->in; a.w = d->wpacked; a.scale = d->scale; a.shift = d->shift; a.lo = d->lo; a.res = d->res;
    a.out = d->out;
    a.Cin = d->cin; a.cin_pad = ct_conv_bf16_cin_pad(d->cin); a.H = d->h; a.W = d->w;
    a.in_ctot = d->in_ctot; a.in_coff = d->in_coff;
    a.M = d->cout; a.cout_pad = ct_conv_bf16_cout_pad(d->cout);
    a.in_bytes = (unsigned)in_bytes;
    a.w_bytes = (unsigned)((size_t)d->kh * d->kw * a.cin_pad * a.cout_pad * 2);
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.dil = d->dil;
    a.OW = d->ow; a.OHW = d->oh * d->ow; a.Npix = d->batch * a.OHW;
    a.out_ctot = d->out_ctot; a.out_coff = d->out_coff; a.res_ctot = d->res_ctot; a.res_coff = d->res_coff;
    a.res_scale = d->res_scale; a.relu = d->relu; a.nseg = d->nseg;
    for (int g = 0; g < d->nseg; ++g) a.seg[g] = d->seg[g];
    int tiles_n = 0;
    hipStream_t st = ctdet::as_stream(stream);
    auto go = [&](auto kernel, int bm, int bn, int bk, int nt) {
        tiles_n = (a.Npix + bm - 1) / bm;
        a.tiles_m = (d->cout + bn - 1) / bn;
        const int taps = d->kh * d->kw;
        const int nsteps = a.cin_pad == KSMALL ? (taps + bk / 8 - 1) / (bk / 8) : taps * (a.cin_pad / bk);
        a.ksplit = 1;
        a.steps_per_split = nsteps;
        int want = d->ksplit;
        const int tiles = a.tiles_m * tiles_n;
        if (want < 0) want = tiles * 2 > 768 ? 1 : std::min(nsteps / 2, 768 / tiles);   // ~3 workgroups per CU
        if (bm == 256) want = 1;
        const long long slab = (long long)d->cout * a.Npix;
        if (d->ksplit_ws && slab > 0) want = (int)std::min<long long>(want, d->ksplit_ws_floats / slab);
        if (want > 1 && d->ksplit_ws && nsteps >= 2 && slab < 0x7FFFFFFFLL) {
            a.steps_per_split = (nsteps + want - 1) / want;
            a.ksplit = (nsteps + a.steps_per_split - 1) / a.steps_per_split;
            a.ws = d->ksplit_ws;
        }
        const bool dma = a.cin_pad != KSMALL;
        const size_t ring = (size_t)(dma && bk == 32 ? 3 : 2) * (bm + bn) * (dma ? bk * 2 : bk * 2 + 16);
        const size_t srow = (size_t)bn * 4 + 16;
        const size_t smem = std::max(ring, (bm * srow <= ring || !(dma && bk == 32) ? bm : 64) * srow);   // operand ring | output staging
        static std::mutex mu;
        static std::unordered_set<const void*> raised;
        if (smem > 64 * 1024) {
            std::lock_guard<std::mutex> lk(mu);
            if (!raised.count((const void*)kernel)) {
                if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                    return;
                raised.insert((const void*)kernel);
            }
        }
        hipLaunchKernelGGL(kernel, dim3(a.tiles_m * tiles_n, a.ksplit), dim3(nt), smem, st, a);
    };
    const int tn128 = (a.Npix + 127) / 128;
    const bool narrow = d->cout <= 64 || (long long)((d->cout + 127) / 128) * tn128 < 256;   // few tiles: finer ones
    // 256 x 256 (eight waves): layers with >= 256 output channels whose 256-pixel tiles still give every CU a workgroup
    const char* big_env = getenv("CTDET_BF16_BIG_MIN");     // read per call: the tests force / forbid the variant
    const int big_min = big_env ? atoi(big_env) : 200;
    const long long big_tiles = (long long)(a.cout_pad / 256) * ((a.Npix + 255) / 256);
    // one workgroup per CU: the tile count must fill its last round of 256 (362 tiles = 1.41 rounds lose to the 128-wide tiles)
    const bool big = a.cin_pad != KSMALL && d->cout >= 256 && a.cout_pad % 256 == 0 && d->nseg == 0 && big_tiles >= big_min &&
                     (big_env || big_tiles * 5 >= (big_tiles + 255) / 256 * 256 * 4);
    static const int wide_bk = getenv("CTDET_BF16_BK") ? atoi(getenv("CTDET_BF16_BK")) : 64;   // 32: the three-buffer ring (faster on conv6 / conv7 / 1x1, slower on the 3x3 trunk)
    if (a.cin_pad == KSMALL) {
        if (narrow) go(conv_bf16_nhwc<128, 64, 32, true, 256>, 128, 64, 32, 256);
        else go(conv_bf16_nhwc<128, 128, 32, true, 256>, 128, 128, 32, 256);
    } else if (big) {
        go(conv_bf16_nhwc<256, 256, 32, false, 512>, 256, 256, 32, 512);
    } else if (narrow && d->cin <= 64 && !(getenv("CTDET_BF16_SHORTK") && atoi(getenv("CTDET_BF16_SHORTK")) == 0)) {
        // short reductions (conv1_2: 9 k-steps): the workgroup is prologue / epilogue bound, 32-channel steps halve
        // its LDS footprint so that four instead of two of them share a CU
        go(conv_bf16_nhwc<128, 64, 32, false, 256>, 128, 64, 32, 256);
    } else {
        if (narrow) go(conv_bf16_nhwc<128, 64, 64, false, 256>, 128, 64, 64, 256);
        else if (wide_bk == 32) go(conv_bf16_nhwc<128, 128, 32, false, 256>, 128, 128, 32, 256);
        else go(conv_bf16_nhwc<128, 128, 64, false, 256>, 128, 128, 64, 256);
    }
    CT_LAUNCH_CHECK("conv_bf16_nhwc");
    if (a.ksplit > 1) {
        hipLaunchKernelGGL(conv_bf16_splitk_epilogue, dim3(std::min((a.Npix * a.M + 255) / 256, 2048)), dim3(256), 0, st, a);
        CT_LAUNCH_CHECK("conv_bf16_splitk_epilogue");
    }
    return CT_OK;
}
