// libctdet: fused implicit-GEMM convolution for the RFBNet-VGG stack on gfx950.
//
// GEMM view:  C[M = cout][N = batch*oh*ow] = W[M][K = cin*kh*kw] * im2col(X)[K][N]
// computed with v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//
// Data flow per workgroup (256 threads = 4 wave64, one per SIMD):
//   * A tile (weights, pre-packed k-major [k_pad][m_pad] so a tile row is contiguous in cout)
//     -> buffer_load_dwordx4 -> registers -> LDS As[buf][BK][BM]
//   * B tile (im2col gather from the NCHW activation): every thread owns ONE output pixel
//     for the whole kernel, so (n, oh, ow), the input base offset and the 3x3/1x3/... tap
//     validity mask are computed once; every k row of a k-step is wave-uniform, so channel
//     and tap offsets live in SGPRs and each gathered element costs one v_add + one
//     v_cndmask + one buffer_load_dword whose out-of-range offset returns 0 (zero padding
//     and ragged edges come from the buffer bounds check, no branches)
//     -> registers -> LDS Bs[buf][BK][BN] (pixel-contiguous: coalesced HBM reads, conflict
//     free ds_write_b32 / ds_read_b32)
//   * LDS double buffer, ONE barrier per k-step; global loads of step s+1 are issued before
//     the MFMAs of step s and written to the other buffer after them
//   * epilogue in registers: *scale[co] + shift[co] (bias or folded eval-BatchNorm),
//     optional (*res_scale + residual), ReLU, then either an NCHW store into a channel
//     slice of a wider buffer (concat fusion) or a channels-last scatter into the flattened
//     loc/conf/obj head buffers (permute+view+cat fusion).
//
// k ordering: k = ci*KH*KW + tap (the natural [cout][cin][kh][kw] order); a k-step covers
// CPB whole input channels so the row -> (channel, tap) split is a compile-time constant.
#include "ct_common.h"
#include "ct_f16x2.h"
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdlib>
#include <mutex>
#include <unordered_set>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kInvalidOff = 0x7FFFFFF0;          // >= num_records of every descriptor -> loads 0
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;  // descriptors stay below 2 GiB

struct ConvArgs {
    const float* in;
    const float* wpk;
    const float* scale;
    const float* shift;
    const float* res;
    const float* lo;
    float* out;
    unsigned in_bytes, w_bytes;
    int Cin, H, W, in_ctot, in_coff;
    int M, M_pad, nsteps;
    int stride, pad_h, pad_w, dil;
    int OW, OHW, Npix;
    int out_ctot, out_coff, res_ctot, res_coff;
    float res_scale;
    int relu;
    int nseg;
    ct_out_segment seg[3];
    int tiles_m, tiles_n;
    int transposed;
    int ksplit, steps_per_split;    // > 1: blockIdx.y owns k-steps [y*sps, (y+1)*sps) and writes its raw sums to ws
    float* ws;                      // [ksplit][M][Npix] partial sums, reduced in fixed order by conv_splitk_epilogue
    unsigned* out_amax;             // ct_conv_desc.out_absmax (per-image max |y| of what the launch stores), or null
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int KH, int KW, int CPB, int BM, int BN, int WAVES_M, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_igemm_f32(const ConvArgs a)
{
    constexpr int KHW = KH * KW;
    constexpr int BK = CPB * KHW;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = 256 / BN;             // B rows gathered per pass of the workgroup
    constexpr int NB = BK / RPP;              // gathered elements per thread per k-step
    constexpr int A_F4 = BK * BM / 4;         // float4s in one A tile
    constexpr int NA = (A_F4 + 255) / 256;
    static_assert(BK % 2 == 0 && BK % RPP == 0, "k-step must be even and divide into passes");
    static_assert(TM >= 1 && TN >= 1 && BN >= 64, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2*BK*BM + 2*BK*BN floats
    float* const As = smem;                   // [2][BK][BM]
    float* const Bs = smem + 2 * BK * BM;     // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31;
    const int hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WAVES_M) * WM;
    const int wn0 = (wave / WAVES_M) * WN;

    // XCD-aware tile order: workgroup b runs on XCD b%8; give every XCD a contiguous chunk of
    // the (cout-tile fastest) tile sequence so the workgroups sharing an im2col tile share an L2.
    int wg;
    {
        const int nwg = a.tiles_m * a.tiles_n;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (wg % a.tiles_m) * BM;
    const int n0 = (wg / a.tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpk, a.w_bytes);

    // ---- per-thread im2col constants (this thread's pixel never changes) ----
    const int bp = tid % BN;
    const int rowgrp = __builtin_amdgcn_readfirstlane(tid / BN);
    const int HW = a.H * a.W;
    // source pixel of tap (kh,kw) for this lane's output pixel, as an element offset, or -1:
    //   forward     (oh,ow) reads  ih = oh*stride - pad + kh*dil
    //   transposed  (data gradient, out pixel = input pixel (ih,iw) of the forward conv; the
    //               "input" here is dY):  oh = (ih + pad - kh*dil) / stride when divisible
    int img_base, oh_, ow_;
    bool pvalid;
    {
        const int P = n0 + bp;
        pvalid = P < a.Npix;
        const int Pc = pvalid ? P : 0;
        const int n = Pc / a.OHW;
        const int s = Pc - n * a.OHW;
        oh_ = s / a.OW;
        ow_ = s - oh_ * a.OW;
        img_base = (n * a.in_ctot + a.in_coff) * HW;
    }
    auto tap_offset = [&](int kh, int kw) -> int {
        int ih, iw;
        bool ok = pvalid;
        if (!a.transposed) {
            ih = oh_ * a.stride - a.pad_h + kh * a.dil;
            iw = ow_ * a.stride - a.pad_w + kw * a.dil;
        } else {
            const int th = oh_ + a.pad_h - kh * a.dil, tw = ow_ + a.pad_w - kw * a.dil;
            ih = th / a.stride;
            iw = tw / a.stride;
            ok = ok && th >= 0 && tw >= 0 && ih * a.stride == th && iw * a.stride == tw;
        }
        ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        return ok ? img_base + ih * a.W + iw : -1;
    };

    // ---- staging constants: everything per-lane is computed ONCE; inside the k loop the
    //      gather is buffer_load(voffset = lane constant, soffset = wave-uniform channel offset)
    int a_voff[NA];       // byte offset inside one k-step slab, or kInvalidOff
    int a_lds[NA];        // float index inside As[buf]
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int f = tid + 256 * j;
        if (f >= A_F4) f %= A_F4;         // surplus lanes of the last pass duplicate another element
                                          // (same data, same LDS slot): no exec masking needed
        const int arow = f / (BM / 4), ac4 = f % (BM / 4);
        const int col = m0 + ac4 * 4;
        a_voff[j] = (col < a.M_pad) ? (arow * a.M_pad + col) * 4 : kInvalidOff;
        a_lds[j] = arow * BM + ac4 * 4;
    }
    const int a_step_bytes = BK * a.M_pad * 4;

    int b_voff[NB];       // this lane's pixel at the tap of gathered row i (bytes), or kInvalidOff
    int b_chan[NB];       // wave-uniform: channel of row i inside the k-step
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = rowgrp + RPP * i;                 // wave-uniform
        const int c = r / KHW, tap = r - c * KHW;
        const int kh = tap / KW, kw = tap - kh * KW;
        const int off = tap_offset(kh, kw);
        b_voff[i] = off >= 0 ? off * 4 : kInvalidOff;
        b_chan[i] = c;
    }
    const int chan_bytes = HW * 4;

    i32x4 areg[NA];
    float breg[NB];

    // staging element e: e < NA -> A float4 j = e, else B dword i = e - NA
    auto load_elem = [&](int e, int step) {
        if (e < NA) {
            areg[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, a_voff[e], step * a_step_bytes, 0);
        } else {
            const int i = e - NA;
            // channels past Cin (last k-step only) read channel 0: their packed weights are zero
            const int ci = step * CPB + b_chan[i];
            const int soff = (ci < a.Cin ? ci : 0) * chan_bytes;
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, b_voff[i], soff, 0));
        }
    };
    auto store_elem = [&](int e, int buf) {
        if (e < NA) {
            *reinterpret_cast<i32x4*>(As + buf * (BK * BM) + a_lds[e]) = areg[e];
        } else {
            const int i = e - NA;
            Bs[buf * (BK * BN) + (rowgrp + RPP * i) * BN + bp] = breg[i];
        }
    };
    auto load_tile = [&](int step) {
#pragma unroll
        for (int e = 0; e < NA + NB; ++e) load_elem(e, step);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NA + NB; ++e) store_elem(e, buf);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NKP = BK / 2;
    constexpr int NST = NA + NB;                          // staging elements per lane per k-step
    constexpr int NSLOT = NKP - 1;                        // k-pairs that carry staging work
    constexpr int WPS = (NST + NSLOT - 1) / NSLOT;        // staging elements per k-pair slot

    // One k-step.  Software pipeline (registers hold tile step+1 on entry); per k-pair slot:
    //   ds_read the NEXT pair's operand fragments, ds_write a slice of tile step+1 into the other
    //   LDS buffer, re-issue the gather of the same slice for tile step+2 (a whole k-step to land),
    //   then this pair's MFMAs.  sched_barrier(0) keeps the compiler from regrouping the slots.
    const int s0 = blockIdx.y * a.steps_per_split;
    const int s1 = min(a.nsteps, s0 + a.steps_per_split);
    auto k_step = [&](int step, auto store_c, auto load_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
        const int buf = (step - s0) & 1;
        const float* Ab = As + buf * (BK * BM) + hsel * BM + wm0 + l31;
        const float* Bb = Bs + buf * (BK * BN) + hsel * BN + wn0 + l31;
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = Ab[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = Bb[j * 32];
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            const int cur = kp & 1;
            if (kp + 1 < NKP) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[cur ^ 1][i] = Ab[(2 * kp + 2) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[cur ^ 1][j] = Bb[(2 * kp + 2) * BN + j * 32];
            }
#pragma unroll
            for (int q = 0; q < WPS; ++q) {
                const int e = kp * WPS + q;
                if (e < NST) {
                    if (STORE) store_elem(e, buf ^ 1);
                    if (LOAD) load_elem(e, step + 2);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    load_tile(s0);
    store_tile(0);
    if (s1 - s0 > 1) load_tile(s0 + 1);
    __syncthreads();
    int step = s0;
    for (; step + 2 < s1; ++step) k_step(step, T_{}, T_{});
    if (step + 1 < s1) { k_step(step, T_{}, F_{}); ++step; }
    k_step(step, F_{}, F_{});

    if (a.ksplit > 1) {
        // split-K (small maps: too few tiles to fill the chip, k loop latency bound): every split stores its raw
        // partial sums in its own slab ws[split][co][pixel] (no atomics: the result does not depend on timing);
        // conv_splitk_epilogue adds the slabs in order and applies the epilogue
        float* const slab = a.ws + (size_t)blockIdx.y * a.M * a.Npix;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int P = n0 + wn0 + j * 32 + l31;
            if (P >= a.Npix) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                    if (co < a.M) slab[(size_t)co * a.Npix + P] = acc[i][j][r];
                }
        }
        return;
    }

    // ---- epilogue ----
    // per-cout epilogue vectors once per workgroup through LDS (the operand tiles are dead after the last barrier)
    // instead of a per-lane global gather for every output (round 3, measured on the bf16x3 twin of this kernel:
    // a third of a workgroup's fixed time)
    float* const ev = smem;                   // [3][BM]: scale, shift, floor
    for (int i = tid; i < BM; i += 256) {
        const int co = m0 + i;
        const bool in = co < a.M;
        ev[i] = in ? a.scale[co] : 0.f;
        ev[BM + i] = in ? a.shift[co] : 0.f;
        ev[2 * BM + i] = !in ? 0.f : a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY);
    }
    __syncthreads();
    const bool track = a.out_amax != nullptr;       // ct_conv_desc.out_absmax: per-image maxima of |v| of what the launch stores
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int P = n0 + wn0 + j * 32 + l31;
        const bool live = P < a.Npix;
        const int n = live ? P / a.OHW : -1;
        float amax_run = 0.f;
        if (live) {
            const int s = P - n * a.OHW;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                    const int co = m0 + cl;
                    if (co >= a.M) continue;
                    float v = acc[i][j][r] * ev[cl] + ev[BM + cl];
                    if (a.res)
                        v = v * a.res_scale +
                            a.res[((size_t)n * a.res_ctot + a.res_coff + co) * a.OHW + s];
                    { const float fl = ev[2 * BM + cl]; v = v < fl ? fl : v; }      // NaN propagates (torch.relu / no clamp)
                    if (track) ctdet::h2::track_absmax(amax_run, v);
                    if (a.nseg == 0) {
                        a.out[((size_t)n * a.out_ctot + a.out_coff + co) * a.OHW + s] = v;
                    } else {
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end)
                                a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                             (size_t)s * a.seg[g].pix_stride + (co - a.seg[g].co_begin)] = v;
                    }
                }
            }
        }
        if (track) ctdet::h2::flush_absmax(a.out_amax, n, amax_run);      // every lane arrives here
    }
}

// The 3-channel image layer (conv1_1: 3x3, 3 -> 64 at full resolution) on the vector ALU.  K = 27 is 14 MFMA steps of a
// GEMM whose B operand is gathered one element per lane and step; its time goes into that gather and the LDS round trip,
// not into the matrix pipe (measured 30 TFLOP/s, 335 us at 300x300 bs 32, against 737 MB of output = ~170 us of HBM
// writes).  Here one lane owns one output pixel: its 27 inputs stay in registers, the weights of eight output channels
// at a time arrive through the scalar cache (the packed [k][m_pad] layout makes them one s_load_dwordx8 per k) and feed
// v_pk_fma_f32 straight from SGPR pairs; 864 packed FMAs per pixel, stores coalesced along the image row.
template <int CIN, int PPT>
__global__ __launch_bounds__(256) void conv_valu3x3_f32(const ConvArgs a)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int K = CIN * 9;
    const int HW = a.H * a.W;
    float x[PPT][K];
    float* op[PPT];
    bool live[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {          // pixels P, P + 256, ...: every store stays coalesced along the image row
        const int P = (blockIdx.x * PPT + p) * 256 + threadIdx.x;
        live[p] = P < a.Npix;
        const int Pc = live[p] ? P : 0;
        const int n = Pc / a.OHW, s = Pc - n * a.OHW;
        const int oh = s / a.OW, ow = s - oh * a.OW;
        const float* ip = a.in + ((size_t)n * a.in_ctot + a.in_coff) * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * a.stride - a.pad_h + kh * a.dil;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * a.stride - a.pad_w + kw * a.dil;
                const bool ok = live[p] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                const int off = ok ? ih * a.W + iw : 0;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = ip[ci * HW + off];
                    x[p][ci * 9 + kh * 3 + kw] = ok ? v : 0.f;
                }
            }
        }
        op[p] = a.out + ((size_t)n * a.out_ctot + a.out_coff) * a.OHW + s;
    }
    // weights and epilogue vectors through the constant address space: wave-uniform, read-only for the whole launch,
    // so they load through the scalar cache into SGPRs (as plain global pointers the stores below could alias them and
    // the compiler keeps them on the vector memory path, 216 VGPRs of weights per eight output channels)
    typedef const __attribute__((address_space(4))) float cfloat;
    typedef const __attribute__((address_space(4))) f32x2 cf32x2;
    cfloat* const wk = (cfloat*)a.wpk;
    cfloat* const sc = (cfloat*)a.scale;
    cfloat* const sh = (cfloat*)a.shift;
    cfloat* const lo = (cfloat*)a.lo;
    float amax_run[PPT];                      // a.out_amax: the thread's maxima of |v| over what it stores, per pixel (= per image)
#pragma unroll
    for (int p = 0; p < PPT; ++p) amax_run[p] = 0.f;
    const bool track = a.out_amax != nullptr;
    for (int co0 = 0; co0 < a.M; co0 += 8) {
        f32x2 acc[PPT][4];
#pragma unroll
        for (int p = 0; p < PPT; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[p][j] = f32x2{0.f, 0.f};
        cfloat* wp = wk + co0;
        // groups of one filter row (3 k, 24 SGPRs of weights), double-buffered.  Scalar loads return out of order, so the
        // only wait is lgkmcnt(0): the first FMA of a group takes that wait, THEN the next group's loads are issued and
        // fly during the remaining FMAs.  (All 27 loads hoisted need 216 SGPRs: the compiler spills them to VGPR
        // lanes, 4 v_readlane per FMA.)
        f32x2 wc[3][4], wn[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) wc[i][j] = ((cf32x2*)(wp + (size_t)i * a.M_pad))[j];
#pragma unroll
        for (int g = 0; g < K / 3; ++g) {
            {
                const f32x2 xx = f32x2{x[0][3 * g], x[0][3 * g]};
                acc[0][0] = __builtin_elementwise_fma(xx, wc[0][0], acc[0][0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < K / 3) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wn[i][j] = ((cf32x2*)(wp + (size_t)(3 * g + 3 + i) * a.M_pad))[j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < PPT; ++p)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x2 xx = f32x2{x[p][3 * g + i], x[p][3 * g + i]};
#pragma unroll
                    for (int j = (p == 0 && i == 0 ? 1 : 0); j < 4; ++j)
                        acc[p][j] = __builtin_elementwise_fma(xx, wc[i][j], acc[p][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) wc[i][j] = wn[i][j];
        }
        // the sums are only used under `if (live)`: without this the compiler sinks most of the FMA chains into
        // that branch, keeps all 27 weight rows alive until there and spills them
#pragma unroll
        for (int p = 0; p < PPT; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[p][j]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int co = co0 + j;
            const float scv = sc[co], shv = sh[co];
            const float lov = a.lo ? lo[co] : (a.relu ? 0.f : -INFINITY);
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                float v = (j & 1 ? acc[p][j >> 1].y : acc[p][j >> 1].x) * scv + shv;
                v = v < lov ? lov : v;                  // NaN propagates
                if (live[p]) {
                    op[p][(size_t)co * a.OHW] = v;
                    if (track) ctdet::h2::track_absmax(amax_run[p], v);
                }
            }
        }
    }
    if (track) {                              // one atomic per image present in the wave (every lane arrives here)
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int P = (blockIdx.x * PPT + p) * 256 + threadIdx.x;
            ctdet::h2::flush_absmax(a.out_amax, live[p] ? P / a.OHW : -1, amax_run[p]);
        }
    }
}

// epilogue of a split-K convolution: sum of the slabs in split order, then the same arithmetic as the fused one
__global__ __launch_bounds__(256) void conv_splitk_epilogue(const ConvArgs a)
{
    const int total = a.M * a.Npix;
    const bool track = a.out_amax != nullptr;
    const int rounds = (total + gridDim.x * 256 - 1) / (gridDim.x * 256);       // the same trip count for every lane (flush below)
    for (int it = 0; it < rounds; ++it) {
        const int idx = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        float amax_run = 0.f;
        int img = -1;
        if (idx < total) {
            const int co = idx / a.Npix, P = idx - co * a.Npix;
            const int n = P / a.OHW, s = P - n * a.OHW;
            img = n;
            float sum = a.ws[idx];
            for (int k = 1; k < a.ksplit; ++k) sum += a.ws[(size_t)k * total + idx];
            float v = sum * a.scale[co] + a.shift[co];
            if (a.res) v = v * a.res_scale + a.res[((size_t)n * a.res_ctot + a.res_coff + co) * a.OHW + s];
            if (a.lo) { const float fl = a.lo[co]; v = v < fl ? fl : v; }      // NaN propagates
            else if (a.relu) v = v < 0.f ? 0.f : v;
            if (track) ctdet::h2::track_absmax(amax_run, v);
            if (a.nseg == 0) {
                a.out[((size_t)n * a.out_ctot + a.out_coff + co) * a.OHW + s] = v;
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end)
                        a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                     (size_t)s * a.seg[g].pix_stride + (co - a.seg[g].co_begin)] = v;
            }
        }
        if (track) ctdet::h2::flush_absmax(a.out_amax, img, amax_run);
    }
}

// --------------------------------------------------------------------------------------
struct PackArgs {
    const float* w[6];
    int mbeg[7];
    int nparts, K, K_pad, M_pad;
    int dgrad, cin, khw;      // dgrad: out[k = co*khw + tap][m = ci] = w[co][ci][tap]
    float* out;
};

__device__ __forceinline__ void pack_weights_body(const PackArgs& p, long first, long stride)
{
    const long total = (long)p.K_pad * p.M_pad;
    for (long idx = first; idx < total; idx += stride) {
        const int k = (int)(idx / p.M_pad), m = (int)(idx - (long)k * p.M_pad);
        float v = 0.f;
        if (!p.dgrad) {
            if (k < p.K) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (i < p.nparts && m >= p.mbeg[i] && m < p.mbeg[i + 1])
                        v = p.w[i][(size_t)(m - p.mbeg[i]) * p.K + k];
            }
        } else if (k < p.K && m < p.cin) {
            const int co = k / p.khw, tap = k - co * p.khw;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i < p.nparts && co >= p.mbeg[i] && co < p.mbeg[i + 1])
                    v = p.w[i][((size_t)(co - p.mbeg[i]) * p.cin + m) * p.khw + tap];
        }
        p.out[idx] = v;
    }
}

__global__ void pack_weights_kernel(const PackArgs p)
{
    pack_weights_body(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// all layers of a training step in one launch: blockIdx.y = recorded item (ct_pack_run)
__global__ void pack_weights_batched_kernel(const PackArgs* __restrict__ items)
{
    const PackArgs p = items[blockIdx.y];
    pack_weights_body(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

__global__ void fold_epilogue_kernel(const float* gamma, const float* beta, const float* mean,
                                     const float* var, float eps, const float* bias, int n,
                                     float* scale, float* shift)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (gamma) {
        const float sc = gamma[i] / sqrtf(var[i] + eps);
        scale[i] = sc;
        shift[i] = beta[i] - mean[i] * sc;
    } else {
        scale[i] = 1.f;
        shift[i] = bias ? bias[i] : 0.f;
    }
}

// --------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn, kmul;
    const char* name;
};
// kmul scales the channels-per-k-step of the geometry (longer k-steps = fewer barriers, more LDS)
const TileCfg kCfgs[] = {
    {128, 128, 1, "128x128"}, {64, 128, 1, "64x128"}, {128, 64, 1, "128x64"}, {64, 64, 1, "64x64"},
    {32, 128, 1, "32x128"},   {160, 128, 1, "160x128"}, {128, 128, 2, "128x128k2"}, {96, 128, 1, "96x128"},
    {0, 256, 0, "valu"},      // conv_valu3x3_f32: not an implicit-GEMM tile (3-channel 3x3 layers only)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kValuCfg = kNumCfgs - 1;

// channels per k-step for (geometry, BN) at kmul = 1
constexpr int cpb_for(int kh, int kw, int bn)
{
    return (kh == 3 && kw == 3) ? (bn == 128 ? 2 : 4)
         : (kh == 1 && kw == 1) ? 16
         : (kh * kw == 3)       ? (bn == 128 ? 6 : 4)
         : (kh == 4 && kw == 4) ? 1
                                : 0;
}

template <typename K>
hipError_t launch_one(K kernel, size_t smem, const ConvArgs& a, hipStream_t st)
{
    if (smem > 64 * 1024) {              // opt in to > 64 KiB of LDS once per kernel
        static std::mutex mu;
        static std::unordered_set<const void*> raised;
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        if (!raised.count(fn)) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            raised.insert(fn);
        }
    }
    hipLaunchKernelGGL(kernel, dim3(a.tiles_m * a.tiles_n, a.ksplit > 1 ? a.ksplit : 1), dim3(256), smem, st, a);
    return hipGetLastError();
}

template <int KH, int KW>
hipError_t launch_geo(int cfg, const ConvArgs& a, hipStream_t st)
{
    switch (cfg) {
#define CT_CASE(idx, BM, BN, WMV, KMUL)                                                       \
    case idx: {                                                                               \
        constexpr int CPB = cpb_for(KH, KW, BN) * KMUL;                                       \
        constexpr size_t SMEM = (size_t)2 * CPB * KH * KW * (BM + BN) * sizeof(float);        \
        /* 128x128 fits 128 registers: ask for 4 waves/SIMD (4 workgroups per CU) */         \
        constexpr int MINW = (BM == 128 && BN == 128 && KMUL == 1) ? 4 : 1;                   \
        return launch_one(conv_igemm_f32<KH, KW, CPB, BM, BN, WMV, MINW>, SMEM, a, st);       \
    }
        CT_CASE(0, 128, 128, 2, 1)
        CT_CASE(1, 64, 128, 2, 1)
        CT_CASE(2, 128, 64, 2, 1)
        CT_CASE(3, 64, 64, 2, 1)
        CT_CASE(4, 32, 128, 1, 1)
        CT_CASE(5, 160, 128, 1, 1)
        CT_CASE(6, 128, 128, 2, 2)
        CT_CASE(7, 96, 128, 1, 1)
#undef CT_CASE
        default:
            return hipErrorInvalidValue;
    }
}

int pick_config(int M, long N)
{
    // Heuristic default (the Python engine can autotune and pass an explicit config).
    const int pad64 = (M + 63) / 64 * 64, pad128 = (M + 127) / 128 * 128;
    const int bm = M <= 32 ? 32 : (pad64 < pad128 ? 64 : 128);
    if (bm == 32) return 4;
    const long tm = (M + bm - 1) / bm;
    const long blocks128 = tm * ((N + 127) / 128);
    const bool small = blocks128 < 2 * 256;
    if (bm == 128) return small ? 2 : 0;
    return small ? 3 : 1;
}

}  // namespace

extern "C" int ct_conv_kpad(int cin, int kh, int kw)
{
    // k_pad must be a whole number of k-steps for every tile config: lcm of their CPBs.
    if (cpb_for(kh, kw, 128) == 0) return -1;
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    int l = 1;
    for (int i = 0; i < kNumCfgs; ++i) {
        if (i == kValuCfg) continue;
        const int c = cpb_for(kh, kw, kCfgs[i].bn) * kCfgs[i].kmul;
        l = l / gcd(l, c) * c;
    }
    return (cin + l - 1) / l * l * kh * kw;
}

extern "C" int ct_conv_mpad(int cout) { return (cout + 31) / 32 * 32; }

extern "C" int ct_conv_num_configs(void) { return kNumCfgs; }

extern "C" const char* ct_conv_config_name(int i)
{
    return (i >= 0 && i < kNumCfgs) ? kCfgs[i].name : "?";
}

static int pack_impl(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                     float* wpacked, int m_pad, int k_pad, int dgrad, ct_stream_t stream)
{
    CT_REQUIRE(nparts >= 1 && nparts <= 6, "ct_conv_pack_weights: nparts=%d (1..6)", nparts);
    PackArgs p{};
    int mtot = 0;
    for (int i = 0; i < nparts; ++i) {
        p.w[i] = w[i];
        p.mbeg[i] = mtot;
        mtot += cout[i];
    }
    p.mbeg[nparts] = mtot;
    for (int i = nparts + 1; i < 7; ++i) p.mbeg[i] = mtot;
    p.nparts = nparts;
    p.dgrad = dgrad;
    p.cin = cin;
    p.khw = kh * kw;
    if (!dgrad) {
        CT_REQUIRE(mtot <= m_pad, "ct_conv_pack_weights: sum(cout)=%d > m_pad=%d", mtot, m_pad);
        p.K = cin * kh * kw;
    } else {
        CT_REQUIRE(cin <= m_pad, "ct_conv_pack_weights_dgrad: cin=%d > m_pad=%d", cin, m_pad);
        p.K = mtot * kh * kw;
    }
    CT_REQUIRE(k_pad >= p.K, "ct_conv_pack_weights: k_pad=%d < K=%d", k_pad, p.K);
    p.K_pad = k_pad;
    p.M_pad = m_pad;
    p.out = wpacked;
    if (ctdet::pack_recording()) {
        ctdet::pack_record(0, &p, sizeof(p));
        return CT_OK;
    }
    const long total = (long)k_pad * m_pad;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, ctdet::as_stream(stream), p);
    CT_LAUNCH_CHECK("pack_weights_kernel");
    return CT_OK;
}

size_t ctdet::pack_direct_item_bytes() { return sizeof(PackArgs); }

int ctdet::launch_pack_direct_batched(const void* items_dev, int n, hipStream_t st)
{
    if (n <= 0) return CT_OK;
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(96, n), dim3(256), 0, st, (const PackArgs*)items_dev);
    CT_LAUNCH_CHECK("pack_weights_batched_kernel");
    return CT_OK;
}

extern "C" int ct_conv_pack_weights(const float* const* w, const int* cout, int nparts, int cin,
                                    int kh, int kw, float* wpacked, int m_pad, int k_pad,
                                    ct_stream_t stream)
{
    return pack_impl(w, cout, nparts, cin, kh, kw, wpacked, m_pad, k_pad, 0, stream);
}

extern "C" int ct_conv_pack_weights_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                          int kh, int kw, float* wpacked, int m_pad, int k_pad,
                                          ct_stream_t stream)
{
    return pack_impl(w, cout, nparts, cin, kh, kw, wpacked, m_pad, k_pad, 1, stream);
}

extern "C" int ct_conv_fold_epilogue(const float* gamma, const float* beta, const float* mean,
                                     const float* var, float eps, const float* bias, int n,
                                     int offset, float* scale, float* shift, ct_stream_t stream)
{
    CT_REQUIRE(n > 0 && offset >= 0, "ct_conv_fold_epilogue: n=%d offset=%d", n, offset);
    CT_REQUIRE(!gamma || (beta && mean && var), "ct_conv_fold_epilogue: BN needs beta/mean/var");
    hipLaunchKernelGGL(fold_epilogue_kernel, dim3((n + 255) / 256), dim3(256), 0,
                       ctdet::as_stream(stream), gamma, beta, mean, var, eps, bias, n,
                       scale + offset, shift + offset);
    CT_LAUNCH_CHECK("fold_epilogue_kernel");
    return CT_OK;
}

extern "C" int ct_conv2d_fwd(const ct_conv_desc* d, ct_stream_t stream)
{
    CT_REQUIRE(d != nullptr, "ct_conv2d_fwd: null descriptor");
    CT_REQUIRE(d->in && d->wpacked && d->scale && d->shift, "ct_conv2d_fwd: null tensor");
    CT_REQUIRE(d->batch > 0 && d->cin > 0 && d->cout > 0 && d->h > 0 && d->w > 0,
               "ct_conv2d_fwd: bad shape");
    CT_REQUIRE(d->stride >= 1 && d->dil >= 1, "ct_conv2d_fwd: stride/dilation");
    if (!d->transposed) {
        const int eoh = (d->h + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int eow = (d->w + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        CT_REQUIRE(eoh == d->oh && eow == d->ow, "ct_conv2d_fwd: oh/ow %dx%d != expected %dx%d", d->oh,
                   d->ow, eoh, eow);
    } else {    // data gradient: (h,w) = spatial size of dY, (oh,ow) = spatial size of dX
        const int fh = (d->oh + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int fw = (d->ow + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        CT_REQUIRE(fh == d->h && fw == d->w, "ct_conv2d_fwd(transposed): dY %dx%d != forward output %dx%d of a %dx%d input",
                   d->h, d->w, fh, fw, d->oh, d->ow);
    }
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_fwd: input slice");
    CT_REQUIRE(d->m_pad >= d->cout && d->m_pad % 4 == 0, "ct_conv2d_fwd: m_pad");
    const int kpad = ct_conv_kpad(d->cin, d->kh, d->kw);
    if (kpad < 0)
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_fwd: %dx%d filters not built", d->kh, d->kw);
    CT_REQUIRE(d->k_pad == kpad, "ct_conv2d_fwd: k_pad=%d, expected %d", d->k_pad, kpad);
    CT_REQUIRE(d->nseg >= 0 && d->nseg <= 3, "ct_conv2d_fwd: nseg");
    if (d->nseg == 0) {
        CT_REQUIRE(d->out && d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot,
                   "ct_conv2d_fwd: output slice");
        CT_REQUIRE(!d->res || (d->res_coff >= 0 && d->res_coff + d->cout <= d->res_ctot),
                   "ct_conv2d_fwd: residual slice");
    } else {
        CT_REQUIRE(!d->res, "ct_conv2d_fwd: residual with segmented output");
        for (int g = 0; g < d->nseg; ++g) CT_REQUIRE(d->seg[g].ptr, "ct_conv2d_fwd: null segment");
    }
    CT_REQUIRE((long long)d->k_pad * d->m_pad * 4 < kMaxBufBytes, "ct_conv2d_fwd: weights too large");

    const long long img_in_bytes = (long long)d->in_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_in_bytes < kMaxBufBytes, "ct_conv2d_fwd: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / img_in_bytes);

    const bool valu_ok = d->kh == 3 && d->kw == 3 && d->cin == 3 && !d->transposed && d->nseg == 0 && !d->res &&
                         d->cout % 8 == 0 && d->batch <= max_chunk && (long long)d->batch * d->oh * d->ow < 0x7FFFFFFFLL;
    int cfg = d->config > 0 ? d->config - 1
            : valu_ok       ? kValuCfg          // the image layer: 1.6-1.75x the best implicit-GEMM tile at every batch
                            : pick_config(d->cout, (long)d->batch * d->oh * d->ow);
    CT_REQUIRE(cfg >= 0 && cfg < kNumCfgs, "ct_conv2d_fwd: config %d", d->config);
    if (cfg == kValuCfg) {
        if (!valu_ok)
            return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_fwd: config 'valu' is for 3x3 convolutions of 3 input channels "
                               "into a multiple of 8 output channels, NCHW output, no residual");
        ConvArgs a{};
        a.in = d->in; a.wpk = d->wpacked; a.scale = d->scale; a.shift = d->shift; a.lo = d->lo; a.out = d->out;
        a.Cin = d->cin; a.H = d->h; a.W = d->w; a.in_ctot = d->in_ctot; a.in_coff = d->in_coff;
        a.M = d->cout; a.M_pad = d->m_pad;
        a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.dil = d->dil;
        a.OW = d->ow; a.OHW = d->oh * d->ow; a.Npix = d->batch * a.OHW;
        a.out_ctot = d->out_ctot; a.out_coff = d->out_coff; a.relu = d->relu;
        a.out_amax = d->out_absmax;
        static const int ppt = getenv("CTDET_VALU_PPT") ? atoi(getenv("CTDET_VALU_PPT")) : 2;
        if (ppt == 1)
            hipLaunchKernelGGL((conv_valu3x3_f32<3, 1>), dim3((a.Npix + 255) / 256), dim3(256), 0, ctdet::as_stream(stream), a);
        else
            hipLaunchKernelGGL((conv_valu3x3_f32<3, 2>), dim3((a.Npix + 511) / 512), dim3(256), 0, ctdet::as_stream(stream), a);
        CT_LAUNCH_CHECK("conv_valu3x3_f32");
        return CT_OK;
    }
    const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
    const int cpb = cpb_for(d->kh, d->kw, bn) * kCfgs[cfg].kmul;

    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        ConvArgs a{};
        a.in = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.wpk = d->wpacked;
        a.scale = d->scale;
        a.shift = d->shift;
        a.lo = d->lo;
        a.OW = d->ow;
        a.OHW = d->oh * d->ow;
        a.res = d->res ? d->res + (size_t)b0 * d->res_ctot * a.OHW : nullptr;
        a.out = d->nseg == 0 ? d->out + (size_t)b0 * d->out_ctot * a.OHW : nullptr;
        a.in_bytes = (unsigned)(img_in_bytes * nb);
        a.w_bytes = (unsigned)((long long)d->k_pad * d->m_pad * 4);
        a.Cin = d->cin;
        a.H = d->h;
        a.W = d->w;
        a.in_ctot = d->in_ctot;
        a.in_coff = d->in_coff;
        a.M = d->cout;
        a.M_pad = d->m_pad;
        a.nsteps = (d->cin + cpb - 1) / cpb;
        a.stride = d->stride;
        a.pad_h = d->pad_h;
        a.pad_w = d->pad_w;
        a.dil = d->dil;
        a.transposed = d->transposed;
        a.Npix = nb * a.OHW;
        a.out_ctot = d->out_ctot;
        a.out_coff = d->out_coff;
        a.res_ctot = d->res_ctot;
        a.res_coff = d->res_coff;
        a.res_scale = d->res_scale;
        a.relu = d->relu;
        a.out_amax = (d->out_absmax && !d->transposed) ? d->out_absmax + (size_t)b0 * ctdet::h2::kLineWords : nullptr;
        a.nseg = d->nseg;
        for (int g = 0; g < d->nseg; ++g) {
            a.seg[g] = d->seg[g];
            a.seg[g].ptr += (size_t)b0 * d->seg[g].img_stride;
        }
        a.tiles_m = (d->cout + bm - 1) / bm;
        a.tiles_n = (a.Npix + bn - 1) / bn;
        a.ksplit = 1;
        a.steps_per_split = a.nsteps;
        int want = d->ksplit;
        if (want < 0) {     // auto: aim at ~3 workgroups per CU, at least two k-steps per split
            static const int target = getenv("CTDET_KSPLIT_TARGET") ? atoi(getenv("CTDET_KSPLIT_TARGET")) : 768;
            const int tiles = a.tiles_m * a.tiles_n;
            want = tiles * 2 > target ? 1 : std::min(a.nsteps / 2, target / tiles);
        }
        const long long slab = (long long)d->cout * a.Npix;
        if (d->ksplit_ws && slab > 0) want = (int)std::min<long long>(want, d->ksplit_ws_floats / slab);
        if (want > 1 && d->ksplit_ws && nb == d->batch && a.nsteps >= 2 && slab < 0x7FFFFFFFLL) {
            const int ks = std::min(want, a.nsteps);
            a.steps_per_split = (a.nsteps + ks - 1) / ks;
            a.ksplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
            a.ws = d->ksplit_ws;
        }
        hipError_t e;
        hipStream_t st = ctdet::as_stream(stream);
        if (d->kh == 3 && d->kw == 3) e = launch_geo<3, 3>(cfg, a, st);
        else if (d->kh == 1 && d->kw == 1) e = launch_geo<1, 1>(cfg, a, st);
        else if (d->kh == 1 && d->kw == 3) e = launch_geo<1, 3>(cfg, a, st);
        else if (d->kh == 3 && d->kw == 1) e = launch_geo<3, 1>(cfg, a, st);
        else if (d->kh == 4 && d->kw == 4) e = launch_geo<4, 4>(cfg, a, st);
        else return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_fwd: %dx%d filters not built", d->kh, d->kw);
        if (e != hipSuccess)
            return ctdet::fail(CT_ERR_HIP, "conv_igemm_f32 launch failed: %s", hipGetErrorString(e));
        if (a.ksplit > 1) {
            const int total = a.M * a.Npix;
            hipLaunchKernelGGL(conv_splitk_epilogue, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, st, a);
            CT_LAUNCH_CHECK("conv_splitk_epilogue");
        }
    }
    return CT_OK;
}
