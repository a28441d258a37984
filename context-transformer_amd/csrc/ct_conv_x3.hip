// libctdet: fp32 convolution on the bf16 matrix pipe -- "bf16x3": every fp32 operand is split EXACTLY into three
// bfloat16 pieces (x = hi + mid + lo, 3 x 8 significant bits, by truncation) and the product a.b is evaluated as the
// six piece products of weight 2^0 .. 2^-16 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped products are below 2^-24 |a.b|.  Same layers,
// descriptor and fused epilogue as ct_conv2d_fwd (models/RFB_Net_vgg.py:7-22 BasicConv, the plain Conv2d layers and the
// multibox heads :238-248): NCHW fp32 in, NCHW fp32 / channels-last head scatter out.
//
// Why: the fp32-input MFMA runs at the vector rate (157 TFLOP/s); the bf16 MFMA at 16x that, so six bf16 MFMAs cost
// 0.375 of one fp32 MFMA (MI355X_MICROARCH.md) -- for the layers that have no Winograd form (1x1, dilated, strided,
// 1x3 / 3x1: ~40 % of the conv time of a step).  Accuracy: bf16 x bf16 products are exact in fp32 and a k-group of 16
// is summed inside the MFMA before ONE rounding into the accumulator, so the accumulator sees 6 K / 16 roundings
// where the fp32 MFMA kernel (v_mfma_f32_32x32x2_f32) sees K / 2; measured per layer against fp64 in
// tests/test_gpu_x3.py (gate: no worse than ct_conv2d_fwd's error).
//
// GEMM view:  C[M = cout][N = batch*oh*ow] = W[M][K] * im2col(X)[K][N],  k ordered (channel group, tap, channel in group):
// a k-step = BK (16 or 32) input channels of ONE filter tap, so filter geometry is a runtime loop (no template per
// filter size) and a thread's gather of a k-step is 8 loads `buffer_load_dword voffset = its pixel at that tap,
// soffset = wave-uniform channel` (adjacent lanes = adjacent pixels: coalesced; out-of-range offset = 0 = padding).
//   workgroup 256 threads = 4 waves, tile BM x BN in {128x128, 64x128, 128x64, 64x64}, wave tile (BM/2) x (BN/2)
//   B side: thread (pixel p = tid % BN, k-octet) loads 8 channels, splits them in registers (4 VALU per element + 3
//     v_perm per pair) and writes three 16-byte rows into LDS [piece][k-octet][pixel][8 bf16]: linear in the lane
//     index, so every ds_write_b128 / ds_read_b128 is conflict-free without padding;
//   A side: the weights are split once per parameter version by ct_conv_pack_weights_x3 into
//     [k-step][piece][k-octet][m_pad][8 bf16]; a tile is three contiguous runs per octet, copied with 16-byte loads;
//   LDS double buffered, one barrier per k-step; the loads of step s+1 are issued before the MFMAs of step s.
// Small maps use the same deterministic slab split-K as ct_conv2d_fwd (desc->ksplit).
#include "ct_common.h"
#include "ct_f16x2.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <unordered_set>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;

struct X3Args {
    const float* in;
    const unsigned char* wx3;       // [step][piece 3][octet BK/8][m_pad][8 bf16]
    const float* scale;
    const float* shift;
    const float* res;
    const float* lo;
    float* out;
    unsigned in_bytes, w_bytes;
    int Cin, H, W, in_ctot, in_coff;
    int M, M_pad, cgroups, KH, KW;
    int stride, pad_h, pad_w, dil;
    int OW, OHW, Npix;
    int out_ctot, out_coff, res_ctot, res_coff;
    float res_scale;
    int relu;
    int nseg;
    ct_out_segment seg[3];
    int tiles_m, tiles_n;
    int ksplit, steps_per_split, nsteps;
    int transposed;                 // data gradient: `in` = dY, output pixel = input pixel of the forward convolution
    float* ws;
    // f16x2 form (H2 instantiations, ct_f16x2.h): activations are split as x 2^eX with eX from the producer's maximum of |input|
    // (ct_conv_desc.in_absmax), the weights arrive as w 2^eW (eW in the trailer of the split weights); the epilogue's per-channel
    // scale takes 2^-(eX + eW)
    const unsigned* in_amax;
    const int* eW;
    unsigned* out_amax;             // ct_conv_desc.out_absmax: max |y| of what the launch stores, or null (any form)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// x = hi + mid + lo exactly (fp32 has 24 significant bits, every piece keeps the next 8 by truncation); returns
// the three fp32 bit patterns whose upper halves are the bf16 pieces
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l)
{
    h = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    l = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, m));
}

// upper halves of (e0, e1) -> one dword [bf16 e0 | bf16 e1 << 16]
__device__ __forceinline__ int pack_hi(unsigned e0, unsigned e1)
{
    return (int)__builtin_amdgcn_perm(e1, e0, 0x07060302u);
}

// DUAL: the hi.hi products accumulate in one register block, the five small products (2^-8 .. 2^-16 of it) in a second
//   one, added at the end -- the large accumulator then sees K / 16 roundings instead of 6 K / 16 (measured: 2.5x less
//   error against fp64 than ct_conv2d_fwd's v_mfma_f32_32x32x2_f32, which sees K / 2).
// One k-step is ONE basic block: in source order a slice of the side work of the step (LDS write of tile s+1 from the
// staging registers, the split of one gathered element, the re-issue of the same registers' loads for tile s+2) follows
// every MFMA, and the compiler's scheduler keeps that interleaving (pinning the slots with sched_barrier measured
// 10-40 % SLOWER here, as cdna_hip_programming.md warns for 32-cycle MFMAs); first / last steps are peeled (STORE /
// LOAD flags) so no branch cuts the block.
// H2: the f16x2 operand form (csrc/ct_f16x2.h): two binary16 pieces per operand, the three products (lo, hi), (hi, lo), (hi, hi) on
//   v_mfma_f32_32x32x16_f16 -- half the MFMAs, two thirds of the LDS traffic and of the weight bytes, 2 instead of 5.5 vector
//   instructions per split activation; DUAL then keeps (hi, hi) apart from the two small products.
template <int BM, int BN, int BK, bool DUAL, bool H2>
__global__ __launch_bounds__(256, 2) void conv_x3_f32(const X3Args a)
{
    constexpr int NP = H2 ? 2 : 3;              // pieces per operand
    constexpr int NPROD = H2 ? 3 : 6;           // piece products per multiply-add
    constexpr int OCT = BK / 8;                 // k-octets per k-step
    constexpr int NH = BK / 16;                 // MFMA k-groups per k-step
    constexpr int WM = BM / 2, WN = BN / 2;     // wave tile (2 x 2 waves)
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_PIECE = OCT * BM * 16;      // bytes of one piece of the A tile
    constexpr int B_PIECE = OCT * BN * 16;
    constexpr int A_BYTES = NP * A_PIECE, B_BYTES = NP * B_PIECE;
    constexpr int NA = (NP * OCT * BM + 255) / 256;         // 16-byte rows of A per thread and k-step
    constexpr int OSTR = 256 / BN;                          // octet stride between a thread's gathers
    constexpr int GPT = OCT / OSTR;                         // k-octets gathered per thread and k-step
    static_assert(256 % BN == 0 && OCT % OSTR == 0 && GPT >= 1, "B staging: every thread gathers GPT whole octets");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];    // 2 x (A_BYTES + B_BYTES)

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave & 1) * WM, wn0 = (wave >> 1) * WN;

    int wg;
    {   // XCD-aware tile order (cout-tile fastest): workgroups sharing an im2col tile share an L2
        const int nwg = a.tiles_m * a.tiles_n;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (wg % a.tiles_m) * BM;
    const int n0 = (wg / a.tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wx3, a.w_bytes);

    // ---- this thread's output pixel (B gather) ----
    const int bp = tid % BN;
    const int oct0 = __builtin_amdgcn_readfirstlane(tid / BN);       // first k-octet of this thread; next: + OSTR
    const int HW = a.H * a.W;
    int img_base, ih0, iw0, bp_image;
    bool pvalid;
    {
        const int P = n0 + bp;
        pvalid = P < a.Npix;
        const int Pc = pvalid ? P : 0;
        const int n = Pc / a.OHW;
        bp_image = n;
        const int s = Pc - n * a.OHW;
        const int oh = s / a.OW, ow = s - oh * a.OW;
        // forward: (oh, ow) reads ih = oh*stride - pad + kh*dil.  transposed (this "output" pixel is the forward conv's
        // INPUT pixel, `in` is dY): dY row = (oh + pad - kh*dil) / stride when that divides
        ih0 = a.transposed ? oh + a.pad_h : oh * a.stride - a.pad_h;
        iw0 = a.transposed ? ow + a.pad_w : ow * a.stride - a.pad_w;
        img_base = (n * a.in_ctot + a.in_coff) * HW;
    }

    // ---- A staging constants ----
    int a_voff[NA], a_lds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int f = tid + 256 * j;
        if (f >= NP * OCT * BM) f %= (NP * OCT * BM);    // surplus lanes duplicate an element (same data, same slot)
        const int row = f % BM, po = f / BM;             // po = piece * OCT + octet
        const int col = m0 + row;
        a_voff[j] = col < a.M_pad ? (po * a.M_pad + col) * 16 : kInvalidOff;
        a_lds[j] = po * (BM * 16) + row * 16;
    }
    const int a_step_bytes = NP * OCT * a.M_pad * 16;
    // f16x2: this thread's gathered activations are split as x 2^eX with eX from the maximum of |input| of ITS pixel's image
    // (ct_f16x2.h: per image, so that an image's results do not depend on its batch mates)
    float vscale = 1.f;
    if constexpr (H2) vscale = __builtin_ldexpf(1.f, ctdet::h2::image_exponent(a.in_amax, bp_image, ctdet::h2::kGrowthNone));
    const int chan_bytes = HW * 4;

    i32x4 areg[NA];
    float breg[GPT][8];
    unsigned sh[GPT][8], sm[GPT][8], sl[GPT][8];      // split pieces of the tile being stored (bf16x3)
    int ph[GPT][4], pl[GPT][4];                       // ... packed pairs of hi / lo pieces (f16x2)

    // ---- loader state: (channel group, tap) of the NEXT tile to load, advanced tap-fastest so the BK channels of a
    // group stay in cache over the filter taps; everything wave-uniform lives in SGPRs
    const int s0 = blockIdx.y * a.steps_per_split;
    const int s1 = min(a.nsteps, s0 + a.steps_per_split);
    int ld_kh, ld_kw, ld_cbase, ld_asoff;
    {
        const int khw = a.KH * a.KW;
        const int cg = s0 / khw, tap = s0 - cg * khw;
        ld_kh = tap / a.KW;
        ld_kw = tap - ld_kh * a.KW;
        ld_cbase = cg * BK;
        ld_asoff = s0 * a_step_bytes;
    }
    int ld_voff = 0;
    const int tap_sgn = a.transposed ? -a.dil : a.dil, tap_sh = a.transposed ? a.stride - 1 : 0;
    auto begin_load = [&]() {                      // per-tile part of the gather address: this lane's pixel at the tap
        // forward: ih = ih0 + kh*dil.  transposed: th = ih0 - kh*dil must be >= 0 and divisible by the stride (1 or 2).
        // One branch-free form for both (the k-step has to stay ONE basic block): tap_sgn = +-dil, tap_sh = 0 or 1; a
        // negative th stays negative under the arithmetic shift and fails the unsigned range check.
        const int th = ih0 + ld_kh * tap_sgn, tw = iw0 + ld_kw * tap_sgn;
        const int ih = th >> tap_sh, iw = tw >> tap_sh;
        const bool ok = pvalid & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (((th | tw) & tap_sh) == 0);
        ld_voff = ok ? (img_base + ih * a.W + iw) * 4 : kInvalidOff;
    };
    auto end_load = [&]() {                        // advance (tap, channel group)
        ld_asoff += a_step_bytes;
        ++ld_kw;
        const int wrap_w = ld_kw == a.KW;               // scalar selects, no branches
        ld_kw = wrap_w ? 0 : ld_kw;
        ld_kh += wrap_w;
        const int wrap_h = ld_kh == a.KH;
        ld_kh = wrap_h ? 0 : ld_kh;
        ld_cbase += wrap_h ? BK : 0;
    };
    auto load_a = [&](int j) { areg[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, a_voff[j], ld_asoff, 0); };
    int ld_soff = 0;                               // running channel offset of the gather (bytes, wave-uniform)
    auto load_b = [&](int g, int e) {
        // Cin is a multiple of BK (checked by the launcher): every channel of a group exists
        if (e == 0) ld_soff = (ld_cbase + (oct0 + g * OSTR) * 8) * chan_bytes;
        breg[g][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, ld_voff, ld_soff, 0));
        ld_soff += chan_bytes;
    };
    auto store_a = [&](int j, int buf) {
        *reinterpret_cast<i32x4*>(lds + buf * (A_BYTES + B_BYTES) + a_lds[j]) = areg[j];
    };
    auto store_b = [&](int g, int piece, int buf) {
        i32x4 pk;
        if constexpr (H2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[q] = piece == 0 ? ph[g][q] : pl[g][q];
        } else {
            const unsigned(&v)[8] = piece == 0 ? sh[g] : piece == 1 ? sm[g] : sl[g];
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[q] = pack_hi(v[2 * q], v[2 * q + 1]);
        }
        *reinterpret_cast<i32x4*>(lds + buf * (A_BYTES + B_BYTES) + A_BYTES + piece * B_PIECE +
                                  ((oct0 + g * OSTR) * BN + bp) * 16) = pk;
    };
    // side work of a k-step as a list of items: A rows, then per gathered octet 8 x (split + reload) and NP x (pack + write);
    // f16x2 splits a PAIR (r, r + 1) at the even item, before either register is reloaded
    constexpr int IPG = 8 + NP;
    constexpr int NW = NA + GPT * IPG;
    auto side_item = [&](int w, int buf, auto store_c, auto load_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
        if (w < NA) {
            if (STORE) store_a(w, buf);
            if (LOAD) load_a(w);
        } else {
            const int g = (w - NA) / IPG, r = (w - NA) % IPG;
            if (r < 8) {
                if (STORE) {
                    if constexpr (H2) {
                        if ((r & 1) == 0) ctdet::h2::split2(breg[g][r] * vscale, breg[g][r + 1] * vscale, ph[g][r >> 1], pl[g][r >> 1]);
                    } else {
                        split3(breg[g][r], sh[g][r], sm[g][r], sl[g][r]);
                    }
                }
                if (LOAD) load_b(g, r);
            } else if (STORE) {
                store_b(g, r - 8, buf);
            }
        }
    };

    f32x16 acc[TM][TN];
    f32x16 acs[DUAL ? TM : 1][DUAL ? TN : 1];     // DUAL: sum of the five small products
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (DUAL) acs[i][j][r] = 0.f;
            }

    constexpr int NMF = NH * NPROD * TM * TN;                 // MFMAs per k-step and wave
    constexpr int WPS = (NW + NMF - 2) / (NMF - 1);           // side items per MFMA slot (the last slot stays free)
    using T_ = std::true_type;
    using F_ = std::false_type;

    auto k_step = [&](int buf, auto store_c, auto load_c) {
        constexpr bool LOAD = decltype(load_c)::value;
        const unsigned char* A = lds + buf * (A_BYTES + B_BYTES) + (wm0 + l31) * 16;
        const unsigned char* B = lds + buf * (A_BYTES + B_BYTES) + A_BYTES + (wn0 + l31) * 16;
        i32x4 fa[NH][NP][TM], fb[NH][NP][TN];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[h][p][i] = *reinterpret_cast<const i32x4*>(A + p * A_PIECE + (2 * h + hsel) * (BM * 16) + i * 512);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[h][p][j] = *reinterpret_cast<const i32x4*>(B + p * B_PIECE + (2 * h + hsel) * (BN * 16) + j * 512);
            }
        if (LOAD) begin_load();
        // smallest products first: bf16x3 (mid, mid), (lo, hi), (hi, lo), (mid, hi), (hi, mid), then (hi, hi); f16x2 (lo, hi), (hi, lo), (hi, hi)
        constexpr int PA[6] = {1, H2 ? 0 : 2, 0, 1, 0, 0}, PB[6] = {H2 ? 0 : 1, H2 ? 1 : 0, H2 ? 0 : 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NMF; ++s) {
            const int h = s / (NPROD * TM * TN), t = (s / (TM * TN)) % NPROD, i = (s / TN) % TM, j = s % TN;
            f32x16& dst = (DUAL && t < NPROD - 1) ? acs[i][j] : acc[i][j];
            if constexpr (H2)
                dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[h][PA[t]][i]),
                                                             __builtin_bit_cast(f16x8, fb[h][PB[t]][j]), dst, 0, 0, 0);
            else
                dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[h][PA[t]][i]),
                                                              __builtin_bit_cast(bf16x8, fb[h][PB[t]][j]), dst, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < WPS; ++q) {
                const int w = s * WPS + q;
                if (w < NW) side_item(w, buf ^ 1, store_c, load_c);
            }
        }
        if (LOAD) end_load();
        // A bare barrier behind the LDS counter only: __syncthreads() carries a workgroup fence that the compiler turns into
        // s_waitcnt vmcnt(0), i.e. a wait for the global loads this step has just issued for the tile two steps ahead (the
        // staging registers are private, only the LDS tiles are shared, and the compiler still waits for a register's own
        // load before its split).  Measured in round 5: no difference on any layer (the loads of a step land within it);
        // a second set of staging registers (loads three tiles ahead) was 10-25 % SLOWER on the layers that cannot fill the
        // chip -- their ~0.7 us per 12-MFMA step is not a memory round trip.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // prologue: tile s0 -> LDS buffer 0, tile s0 + 1 -> staging registers
    begin_load();
#pragma unroll
    for (int w = 0; w < NW; ++w) side_item(w, 0, F_{}, T_{});
    end_load();
#pragma unroll
    for (int w = 0; w < NW; ++w) side_item(w, 0, T_{}, F_{});
    if (s0 + 1 < s1) {
        begin_load();
#pragma unroll
        for (int w = 0; w < NW; ++w) side_item(w, 0, F_{}, T_{});
        end_load();
    }
    __syncthreads();
    int step = s0;
    for (; step + 2 < s1; ++step) k_step((step - s0) & 1, T_{}, T_{});
    if (step + 1 < s1) { k_step((step - s0) & 1, T_{}, F_{}); ++step; }
    k_step((step - s0) & 1, F_{}, F_{});
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acs[i][j][r];
    }

    if (a.ksplit > 1) {
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

    // ---- epilogue (the arithmetic of ct_conv2d_fwd) ----
    // per-cout epilogue vectors once per workgroup through LDS (the operand tiles are dead after the last barrier): a
    // per-lane global gather of scale / shift / floor for each of a lane's 64 outputs cost a third of the fixed time
    // of a workgroup (1x1 16->1024 @19x19 bs 32, one k-step: 39.9 -> 25.8 us)
    float* const ev = reinterpret_cast<float*>(lds);              // [3][BM]: scale, shift, floor
    for (int i = tid; i < BM; i += 256) {
        const int co = m0 + i;
        const bool in = co < a.M;
        ev[i] = in ? a.scale[co] : 0.f;
        ev[BM + i] = in ? a.shift[co] : 0.f;
        ev[2 * BM + i] = !in ? 0.f : a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY);
    }
    __syncthreads();
    const bool track = a.out_amax != nullptr;
    int eW = 0;
    if constexpr (H2) eW = *a.eW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int P = n0 + wn0 + j * 32 + l31;
        const bool live = P < a.Npix;
        const int n = live ? P / a.OHW : -1;
        float amax_run = 0.f;                       // a.out_amax: the thread's maximum of |v| over what it stores for pixel P's image
        if (live) {
            const int s = P - n * a.OHW;
            // f16x2: the sums carry 2^(eX[image] + eW); undone with an exact power of two in front of the per-channel scale
            float ymul = 1.f;
            if constexpr (H2) ymul = __builtin_ldexpf(1.f, -(eW + ctdet::h2::image_exponent(a.in_amax, n, ctdet::h2::kGrowthNone)));
            float* const orow = a.nseg == 0 ? a.out + ((size_t)n * a.out_ctot + a.out_coff) * a.OHW + s : nullptr;
            const float* const rrow = a.res ? a.res + ((size_t)n * a.res_ctot + a.res_coff) * a.OHW + s : nullptr;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // the residual values of the whole 32 x 32 block first, sixteen loads in flight: taken one by one between the
                // stores (which they may alias, so the compiler keeps the order) every load waited a full round trip behind the
                // previous store -- the RFB blocks' ConvLinear layers ran at 0.6 of the rate of the same GEMM without a shortcut
                float rv[16];
                if (rrow) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                        rv[r] = co < a.M ? rrow[(size_t)co * a.OHW] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;      // cout inside the tile
                    const int co = m0 + cl;
                    if (co >= a.M) continue;
                    float v = (H2 ? acc[i][j][r] * ymul : acc[i][j][r]) * ev[cl] + ev[BM + cl];
                    if (rrow) v = v * a.res_scale + rv[r];
                    { const float fl = ev[2 * BM + cl]; v = v < fl ? fl : v; }      // NaN propagates (torch.relu / no clamp)
                    if (track) ctdet::h2::track_absmax(amax_run, v);
                    if (a.nseg == 0) {
                        orow[(size_t)co * a.OHW] = v;
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
        // ct_conv_desc.out_absmax: every lane arrives here; one atomic per image present in the wave
        if (track) ctdet::h2::flush_absmax(a.out_amax, n, amax_run);
    }
}

__device__ __forceinline__ void x3_splitk_one(const X3Args& a, int idx, int total, bool track, float& amax_run, int& n_out);

// sum of the split-K slabs in split order, then the fused epilogue (f16x2 launches: the slabs hold the scaled sums)
__global__ __launch_bounds__(256) void conv_x3_splitk_epilogue(const X3Args a)
{
    const int total = a.M * a.Npix;
    const bool track = a.out_amax != nullptr;
    const int rounds = (total + gridDim.x * 256 - 1) / (gridDim.x * 256);       // the same trip count for every lane (flush below)
    for (int it = 0; it < rounds; ++it) {
        const int idx = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        float amax_run = 0.f;
        int n = -1;
        if (idx < total) x3_splitk_one(a, idx, total, track, amax_run, n);
        if (track) ctdet::h2::flush_absmax(a.out_amax, n, amax_run);
    }
}

__device__ __forceinline__ void x3_splitk_one(const X3Args& a, const int idx, const int total, const bool track, float& amax_run, int& n_out)
{
    {
        const int co = idx / a.Npix, P = idx - co * a.Npix;
        const int n = P / a.OHW, s = P - n * a.OHW;
        n_out = n;
        float ymul = 1.f;       // f16x2 launches: the slabs hold sums scaled by 2^(eX[image] + eW)
        if (a.eW) ymul = __builtin_ldexpf(1.f, -(*a.eW + ctdet::h2::image_exponent(a.in_amax, n, ctdet::h2::kGrowthNone)));
        float sum = a.ws[idx];
        for (int k = 1; k < a.ksplit; ++k) sum += a.ws[(size_t)k * total + idx];
        float v = (sum * ymul) * a.scale[co] + a.shift[co];
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
}

// ---- weight split: [cout][cin][kh][kw] fp32 -> [step = (cgroup, tap)][piece][octet][m_pad][8 bf16] ----
struct X3PackArgs {
    const float* w[6];
    int mbeg[7];
    int nparts, cin, khw, bk, m_pad, cgroups;
    int dgrad;                // rows m = forward INPUT channels, k-channels = forward OUTPUT channels (concatenated parts)
    unsigned short* out;
    // f16x2 layout: [step][piece 2][octet][m_pad][8 f16] of w 2^eW, eW from max |w| over the layer (trailer[0] = its bit pattern,
    // filled by x3h_wmax_kernel; the packing records trailer[1] = eW)
    unsigned* trailer;        // null: bf16x3
};

__device__ __forceinline__ void x3_pack_body(const X3PackArgs& p, long first, long stride)
{
    const int oct = p.bk / 8;
    const long rows = (long)p.cgroups * p.khw * oct * p.m_pad;            // 8-channel rows (all three pieces each)
    for (long idx = first; idx < rows; idx += stride) {
        const int m = (int)(idx % p.m_pad);
        long t = idx / p.m_pad;
        const int o = (int)(t % oct);
        t /= oct;
        const int tap = (int)(t % p.khw), cg = (int)(t / p.khw);
        const float* src = nullptr;
        int mm = 0;
        if (!p.dgrad) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i < p.nparts && m >= p.mbeg[i] && m < p.mbeg[i + 1]) { src = p.w[i]; mm = m - p.mbeg[i]; }
        }
        const long step = (long)cg * p.khw + tap;
        const int np = p.trailer ? 2 : 3;
        unsigned short* base = p.out + (((step * np) * oct + o) * (long)p.m_pad + m) * 8;
        const long piece_stride = (long)oct * p.m_pad * 8;
        const int eW = p.trailer ? ctdet::h2::exponent_for(p.trailer[0], ctdet::h2::kGrowthNone) : 0;
        // the eight channels of a row: pieces collected in registers, ONE 16-byte store per piece (24 two-byte stores before)
        unsigned q0[4] = {0u, 0u, 0u, 0u}, q1[4] = {0u, 0u, 0u, 0u}, q2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = cg * p.bk + o * 8 + e;
            float v = 0.f;
            if (!p.dgrad) {
                if (src && ci < p.cin) v = src[((size_t)mm * p.cin + ci) * p.khw + tap];
            } else if (m < p.cin) {             // ci runs over the concatenated forward output channels
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (i < p.nparts && ci >= p.mbeg[i] && ci < p.mbeg[i + 1])
                        v = p.w[i][((size_t)(ci - p.mbeg[i]) * p.cin + m) * p.khw + tap];
            }
            const int sh = 16 * (e & 1);
            if (p.trailer) {
                const float vs = __builtin_ldexpf(v, eW);
                const _Float16 hi = (_Float16)vs;
                const _Float16 lo = (_Float16)(vs - (float)hi);
                q0[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, hi) << sh;
                q1[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, lo) << sh;
                continue;
            }
            unsigned h, mid, l;
            split3(v, h, mid, l);
            q0[e >> 1] |= (h >> 16) << sh;
            q1[e >> 1] |= (mid >> 16) << sh;
            q2[e >> 1] |= (l >> 16) << sh;
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(base) = u32x4{q0[0], q0[1], q0[2], q0[3]};
        *reinterpret_cast<u32x4*>(base + piece_stride) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        if (!p.trailer) *reinterpret_cast<u32x4*>(base + 2 * piece_stride) = u32x4{q2[0], q2[1], q2[2], q2[3]};
    }
    if (p.trailer && first == 0) p.trailer[1] = (unsigned)ctdet::h2::exponent_for(p.trailer[0], ctdet::h2::kGrowthNone);
}

// max |w| over the parts of a layer -> trailer[0] (atomic max; the caller zeroes the trailer)
__global__ __launch_bounds__(256) void x3h_wmax_kernel(const X3PackArgs p)
{
    unsigned m = 0;
    for (int part = 0; part < p.nparts; ++part) {
        const long n = (long)(p.mbeg[part + 1] - p.mbeg[part]) * p.cin * p.khw;
        const float* w = p.w[part];
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
            const unsigned a = __builtin_bit_cast(unsigned, w[i]) & 0x7FFFFFFFu;
            m = a > m ? a : m;
        }
    }
    m = ctdet::h2::wave_max(m);
    if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(p.trailer, m);
}

__global__ void x3_pack_kernel(const X3PackArgs p)
{
    x3_pack_body(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// every bf16x3 weight split of a training step in ONE launch: blockIdx.y = item of a device-resident list
__global__ void x3_pack_batched_kernel(const X3PackArgs* __restrict__ items)
{
    const X3PackArgs p = items[blockIdx.y];
    x3_pack_body(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

struct X3Cfg {
    int bm, bn, bk, dual;
    const char* name;
    int h2;                   // 1: the f16x2 operand form (weights from ct_conv_pack_weights_x3h, needs ct_conv_desc.in_absmax)
};
// name: x3:<BM>x<BN>k<BK>[d]   d = dual accumulators -- the configurations the engine may select (accuracy gate);
// the single-accumulator forms are kept for the comparison in tests/test_gpu_x3.py and tools/x3_probe.py
const X3Cfg kX3[] = {
    {128, 128, 16, 1, "x3:128x128k16d"}, {64, 128, 16, 1, "x3:64x128k16d"}, {128, 64, 32, 1, "x3:128x64k32d"},
    {64, 64, 32, 1, "x3:64x64k32d"},     {128, 128, 16, 0, "x3:128x128k16"}, {64, 128, 16, 0, "x3:64x128k16"},
    // the same tiles on f16x2 ("h2:..."): the engine maps a table entry x3:<tile> to h2:<tile> where the runtime uses that form
    {128, 128, 16, 1, "h2:128x128k16d", 1}, {64, 128, 16, 1, "h2:64x128k16d", 1}, {128, 64, 32, 1, "h2:128x64k32d", 1},
    {64, 64, 32, 1, "h2:64x64k32d", 1},
};
constexpr int kNumX3 = sizeof(kX3) / sizeof(kX3[0]);

template <typename K>
hipError_t launch_x3(K kernel, size_t smem, const X3Args& a, hipStream_t st)
{
    if (smem > 64 * 1024) {
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

hipError_t launch_cfg(int cfg, const X3Args& a, hipStream_t st)
{
    switch (cfg) {
#define X3_CASE(idx, BM, BN, BK, DU, H2) \
    case idx: return launch_x3(conv_x3_f32<BM, BN, BK, DU, H2>, (size_t)2 * (H2 ? 2 : 3) * (BK / 8) * (BM + BN) * 16, a, st);
        X3_CASE(0, 128, 128, 16, true, false)
        X3_CASE(1, 64, 128, 16, true, false)
        X3_CASE(2, 128, 64, 32, true, false)
        X3_CASE(3, 64, 64, 32, true, false)
        X3_CASE(4, 128, 128, 16, false, false)
        X3_CASE(5, 64, 128, 16, false, false)
        X3_CASE(6, 128, 128, 16, true, true)
        X3_CASE(7, 64, 128, 16, true, true)
        X3_CASE(8, 128, 64, 32, true, true)
        X3_CASE(9, 64, 64, 32, true, true)
#undef X3_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

extern "C" int ct_conv_x3_num_configs(void) { return kNumX3; }

extern "C" const char* ct_conv_x3_config_name(int i) { return (i >= 0 && i < kNumX3) ? kX3[i].name : "?"; }

extern "C" int ct_conv_x3_config_bk(int i) { return (i >= 0 && i < kNumX3) ? kX3[i].bk : -1; }

extern "C" int ct_conv_x3_config_h2(int i) { return (i >= 0 && i < kNumX3) ? kX3[i].h2 : -1; }

constexpr int kX3hTrailerBytes = 256;

// bytes of the f16x2 split weights for k-steps of bk channels, trailer { max |w| bits, eW } included
extern "C" size_t ct_conv_x3h_packed_bytes(int cin, int cout, int kh, int kw, int bk)
{
    if (bk != 16 && bk != 32) return 0;
    const size_t cgroups = (size_t)(cin + bk - 1) / bk;
    return cgroups * kh * kw * 2 * (size_t)bk * ct_conv_mpad(cout) * 2 + kX3hTrailerBytes;
}

// bytes of the split weights for k-steps of bk (16 or 32) channels
extern "C" size_t ct_conv_x3_packed_bytes(int cin, int cout, int kh, int kw, int bk)
{
    if (bk != 16 && bk != 32) return 0;
    const size_t cgroups = (size_t)(cin + bk - 1) / bk;
    return cgroups * kh * kw * 3 * (size_t)bk * ct_conv_mpad(cout) * 2;
}

static int x3_pack_fill(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk, void* wx3,
                        int dgrad, X3PackArgs& p);

static int x3_pack_impl(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk, void* wx3,
                        int dgrad, ct_stream_t stream)
{
    X3PackArgs p{};
    if (int rc = x3_pack_fill(w, cout, nparts, cin, kh, kw, bk, wx3, dgrad, p)) return rc;
    const long rows = (long)p.cgroups * p.khw * (bk / 8) * p.m_pad;
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)std::min<long>((rows + 255) / 256, 4096)), dim3(256), 0,
                       ctdet::as_stream(stream), p);
    CT_LAUNCH_CHECK("x3_pack_kernel");
    return CT_OK;
}

static int x3_pack_fill(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk, void* wx3,
                        int dgrad, X3PackArgs& p)
{
    CT_REQUIRE(w && cout && wx3, "ct_conv_pack_weights_x3: null pointer");
    CT_REQUIRE(nparts >= 1 && nparts <= 6, "ct_conv_pack_weights_x3: nparts=%d (1..6)", nparts);
    CT_REQUIRE(bk == 16 || bk == 32, "ct_conv_pack_weights_x3: bk=%d (16 or 32)", bk);
    CT_REQUIRE(cin > 0 && kh > 0 && kw > 0, "ct_conv_pack_weights_x3: bad filter shape");
    p = X3PackArgs{};
    int mtot = 0;
    for (int i = 0; i < nparts; ++i) {
        CT_REQUIRE(w[i] && cout[i] > 0, "ct_conv_pack_weights_x3: part %d", i);
        p.w[i] = w[i];
        p.mbeg[i] = mtot;
        mtot += cout[i];
    }
    for (int i = nparts; i < 7; ++i) p.mbeg[i] = mtot;
    p.nparts = nparts;
    p.cin = cin;
    p.khw = kh * kw;
    p.bk = bk;
    p.dgrad = dgrad;
    // forward: rows = concatenated couts, k-channels = cin; data gradient: rows = cin, k-channels = concatenated couts
    p.m_pad = ct_conv_mpad(dgrad ? cin : mtot);
    p.cgroups = ((dgrad ? mtot : cin) + bk - 1) / bk;
    p.out = static_cast<unsigned short*>(wx3);
    return CT_OK;
}

// A training step re-splits the weights of every bf16x3 launch (forward and data gradient): ~80 small launches.  The
// arguments never change between steps, so the caller builds the list once -- ct_conv_x3_pack_item fills ONE item of
// ct_conv_x3_pack_item_bytes() bytes in HOST memory from the arguments of ct_conv_pack_weights_x3[_dgrad] --, copies
// it to the device and replays it with ct_conv_x3_pack_run: one launch.
extern "C" size_t ct_conv_x3_pack_item_bytes(void) { return sizeof(X3PackArgs); }

extern "C" int ct_conv_x3_pack_item(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk,
                                    void* wx3, int dgrad, void* host_item)
{
    CT_REQUIRE(host_item, "ct_conv_x3_pack_item: null item");
    X3PackArgs p{};
    if (int rc = x3_pack_fill(w, cout, nparts, cin, kh, kw, bk, wx3, dgrad, p)) return rc;
    memcpy(host_item, &p, sizeof(p));
    return CT_OK;
}

extern "C" int ct_conv_x3_pack_run(const void* items_dev, int n, ct_stream_t stream)
{
    CT_REQUIRE(items_dev || n == 0, "ct_conv_x3_pack_run: null list");
    if (n <= 0) return CT_OK;
    hipLaunchKernelGGL(x3_pack_batched_kernel, dim3(64, n), dim3(256), 0, ctdet::as_stream(stream),
                       static_cast<const X3PackArgs*>(items_dev));
    CT_LAUNCH_CHECK("x3_pack_batched_kernel");
    return CT_OK;
}

extern "C" int ct_conv_pack_weights_x3(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                                       int bk, void* wx3, ct_stream_t stream)
{
    return x3_pack_impl(w, cout, nparts, cin, kh, kw, bk, wx3, 0, stream);
}

// f16x2: max |w| first, then the split of w 2^eW (forward layout only; three launches, not recordable)
extern "C" int ct_conv_pack_weights_x3h(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                                        int bk, void* wx3, ct_stream_t stream)
{
    X3PackArgs p{};
    if (int rc = x3_pack_fill(w, cout, nparts, cin, kh, kw, bk, wx3, 0, p)) return rc;
    hipStream_t st = ctdet::as_stream(stream);
    int mtot = p.mbeg[nparts];
    p.trailer = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(wx3) + ct_conv_x3h_packed_bytes(cin, mtot, kh, kw, bk) -
                                            kX3hTrailerBytes);
    CT_HIP(hipMemsetAsync(p.trailer, 0, kX3hTrailerBytes, st));
    const long nw = (long)mtot * cin * kh * kw;
    hipLaunchKernelGGL(x3h_wmax_kernel, dim3((unsigned)std::min<long>((nw + 255) / 256, 1024)), dim3(256), 0, st, p);
    CT_LAUNCH_CHECK("x3h_wmax_kernel");
    const long rows = (long)p.cgroups * p.khw * (bk / 8) * p.m_pad;
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)std::min<long>((rows + 255) / 256, 4096)), dim3(256), 0, st, p);
    CT_LAUNCH_CHECK("x3_pack_kernel");
    return CT_OK;
}

extern "C" int ct_conv_pack_weights_x3_dgrad(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                                             int bk, void* wx3, ct_stream_t stream)
{
    return x3_pack_impl(w, cout, nparts, cin, kh, kw, bk, wx3, 1, stream);
}

extern "C" int ct_conv2d_x3_fwd(const ct_conv_desc* d, const void* wx3, int config, ct_stream_t stream)
{
    CT_REQUIRE(d != nullptr && wx3 != nullptr, "ct_conv2d_x3_fwd: null pointer");
    CT_REQUIRE(d->in && d->scale && d->shift, "ct_conv2d_x3_fwd: null tensor");
    CT_REQUIRE(config >= 0 && config < kNumX3, "ct_conv2d_x3_fwd: config %d (0..%d)", config, kNumX3 - 1);
    CT_REQUIRE(d->batch > 0 && d->cin > 0 && d->cout > 0 && d->h > 0 && d->w > 0, "ct_conv2d_x3_fwd: bad shape");
    CT_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->dil >= 1, "ct_conv2d_x3_fwd: filter geometry");
    if (!d->transposed) {
        const int eoh = (d->h + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int eow = (d->w + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        CT_REQUIRE(eoh == d->oh && eow == d->ow, "ct_conv2d_x3_fwd: oh/ow %dx%d != expected %dx%d", d->oh, d->ow, eoh, eow);
    } else {    // data gradient: (h,w) = spatial size of dY, (oh,ow) = spatial size of dX (ct_conv2d_fwd's contract)
        if (d->stride > 2)
            return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_x3_fwd(transposed): stride %d (1 or 2)", d->stride);
        const int fh = (d->oh + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int fw = (d->ow + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        CT_REQUIRE(fh == d->h && fw == d->w, "ct_conv2d_x3_fwd(transposed): dY %dx%d != forward output %dx%d of a %dx%d input",
                   d->h, d->w, fh, fw, d->oh, d->ow);
    }
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_x3_fwd: input slice");
    CT_REQUIRE(d->nseg >= 0 && d->nseg <= 3, "ct_conv2d_x3_fwd: nseg");
    if (d->nseg == 0) {
        CT_REQUIRE(d->out && d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "ct_conv2d_x3_fwd: output slice");
        CT_REQUIRE(!d->res || (d->res_coff >= 0 && d->res_coff + d->cout <= d->res_ctot), "ct_conv2d_x3_fwd: residual slice");
    } else {
        CT_REQUIRE(!d->res, "ct_conv2d_x3_fwd: residual with segmented output");
        for (int g = 0; g < d->nseg; ++g) CT_REQUIRE(d->seg[g].ptr, "ct_conv2d_x3_fwd: null segment");
    }
    const int bm = kX3[config].bm, bn = kX3[config].bn, bk = kX3[config].bk;
    const bool h2 = kX3[config].h2 != 0;
    CT_REQUIRE(!h2 || d->in_absmax, "ct_conv2d_x3_fwd: the f16x2 configurations need the maximum of |input| (ct_conv_desc.in_absmax: "
               "the producer's out_absmax slot, or ct_absmax_f32)");
    CT_REQUIRE(!h2 || !d->transposed, "ct_conv2d_x3_fwd: the f16x2 configurations are forward-only");
    if (d->cin % bk != 0)
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_x3_fwd: cin=%d is not a multiple of the k-step (%d channels)",
                           d->cin, bk);
    const int m_pad = ct_conv_mpad(d->cout);
    const size_t wbytes = h2 ? ct_conv_x3h_packed_bytes(d->cin, d->cout, d->kh, d->kw, bk) - kX3hTrailerBytes
                             : ct_conv_x3_packed_bytes(d->cin, d->cout, d->kh, d->kw, bk);
    CT_REQUIRE((long long)wbytes < kMaxBufBytes, "ct_conv2d_x3_fwd: weights too large");
    const long long img_in_bytes = (long long)d->in_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_in_bytes < kMaxBufBytes, "ct_conv2d_x3_fwd: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / img_in_bytes);
    hipStream_t st = ctdet::as_stream(stream);

    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        X3Args a{};
        a.in = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.wx3 = static_cast<const unsigned char*>(wx3);
        a.scale = d->scale;
        a.shift = d->shift;
        a.lo = d->lo;
        a.OW = d->ow;
        a.OHW = d->oh * d->ow;
        a.res = d->res ? d->res + (size_t)b0 * d->res_ctot * a.OHW : nullptr;
        a.out = d->nseg == 0 ? d->out + (size_t)b0 * d->out_ctot * a.OHW : nullptr;
        a.in_bytes = (unsigned)(img_in_bytes * nb);
        a.w_bytes = (unsigned)wbytes;
        a.Cin = d->cin;
        a.H = d->h;
        a.W = d->w;
        a.in_ctot = d->in_ctot;
        a.in_coff = d->in_coff;
        a.M = d->cout;
        a.M_pad = m_pad;
        a.cgroups = (d->cin + bk - 1) / bk;
        a.KH = d->kh;
        a.KW = d->kw;
        a.nsteps = a.cgroups * d->kh * d->kw;
        a.stride = d->stride;
        a.pad_h = d->pad_h;
        a.pad_w = d->pad_w;
        a.dil = d->dil;
        a.Npix = nb * a.OHW;
        a.out_ctot = d->out_ctot;
        a.out_coff = d->out_coff;
        a.res_ctot = d->res_ctot;
        a.res_coff = d->res_coff;
        a.res_scale = d->res_scale;
        a.relu = d->relu;
        a.transposed = d->transposed;
        a.in_amax = d->in_absmax ? d->in_absmax + (size_t)b0 * ctdet::h2::kLineWords : nullptr;
        a.eW = h2 ? reinterpret_cast<const int*>(static_cast<const unsigned char*>(wx3) + wbytes) + 1 : nullptr;
        a.out_amax = d->out_absmax ? d->out_absmax + (size_t)b0 * ctdet::h2::kLineWords : nullptr;
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
        if (want < 0) {     // auto: ~3 workgroups per CU, at least two k-steps per split
            const int tiles = a.tiles_m * a.tiles_n;
            want = tiles * 2 > 768 ? 1 : std::min(a.nsteps / 2, 768 / tiles);
        }
        const long long slab = (long long)d->cout * a.Npix;
        if (d->ksplit_ws && slab > 0) want = (int)std::min<long long>(want, d->ksplit_ws_floats / slab);
        if (want > 1 && d->ksplit_ws && nb == d->batch && a.nsteps >= 2 && slab < 0x7FFFFFFFLL) {
            const int ks = std::min(want, a.nsteps);
            a.steps_per_split = (a.nsteps + ks - 1) / ks;
            a.ksplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
            a.ws = d->ksplit_ws;
        }
        hipError_t e = launch_cfg(config, a, st);
        if (e != hipSuccess) return ctdet::fail(CT_ERR_HIP, "conv_x3_f32 launch failed: %s", hipGetErrorString(e));
        if (a.ksplit > 1) {
            const int total = a.M * a.Npix;
            hipLaunchKernelGGL(conv_x3_splitk_epilogue, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, st, a);
            CT_LAUNCH_CHECK("conv_x3_splitk_epilogue");
        }
    }
    return CT_OK;
}
