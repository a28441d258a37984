// libctdet: Winograd F(2x2,3x3) on the bf16 matrix pipe ("bf16x3") -- the 3x3 / stride 1 / dilation 1 / pad 1 layers
// of the RFBNet-VGG stack (models/RFB_Net_vgg.py:219-227 VGG trunk, :7-22 the 3x3 BasicConv layers, :238-248 the
// multibox heads) with the transform-domain GEMMs evaluated as the six bf16 piece products of ct_conv_x3.hip instead of
// on the fp32-input MFMA.  Same ct_conv_desc contract and fused epilogue as ct_conv2d_wino_fwd.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d; B^T, A^T have entries 0, +-1
//
// Why F(2x2) and not F(4x4) here: per (output, cin, cout) F(2x2,3x3) needs 4 multiplications; as six bf16 MFMA products
// at 16x the fp32 MFMA rate that is 4 * 6 / 16 = 1.5 fp32-MFMA multiplication times, against 2.25 for F(4x4,3x3) on the
// fp32 MFMA (ct_wino4.hip) and 3.4 for the direct bf16x3 kernel.  F(4x4,3x3) on bf16x3 would be 0.84, but its 36
// accumulator blocks fill the register file at 32 tiles x 64 couts, its split V does not fit LDS twice and a workgroup
// needs 221 KB of U per 16 channels; two F(4x4) / bf16x3 forms were built and measured in round 4 (git history: a fused
// four-wave kernel with the whole register file, 700 .. 800 us on 512 -> 512 @38x38 bs 32; a transform kernel + a
// VALU-free GEMM kernel, 85 + 780 us) and removed again.  Accuracy: the transforms are exact up to
// the usual fp32 adds, bf16 x bf16 products are exact in fp32, a k-group of 16 channels is summed inside the MFMA before
// ONE rounding, and the eight-wave form keeps the hi.hi products in their own accumulator (K / 16 roundings of the large
// sum; the fp32 MFMA sees K).
//
// One fused kernel, only the pre-transformed, pre-split weights U exist in HBM in the transform domain:
//   workgroup (512 threads, 8 waves) = 32 output tiles x 64 output channels, loops over 16-channel chunks
//     * every thread loads ONE 4x4 patch (tile = lane & 31, channel = 2 wave + lane / 32) with four 16-byte buffer
//       loads, applies B^T d B in registers and writes the 16 fp32 transform-domain values lane-linearly to LDS
//       V[point 16][channel 16][tile 32] (conflict-free ds_write_b32; three 32 KB buffers);
//     * wave w owns the transform points 2w, 2w+1 for all 64 couts x 32 tiles.  It is the ONLY reader of those
//       points, so the bf16x3 split happens after the LDS read: a lane reads its 8 channels of a point, splits them into
//       three pieces (4 VALU per value + 1.5 v_perm) and holds the three B fragments of v_mfma_f32_32x32x16_bf16;
//     * its A fragments (U) never pass through LDS: ct_conv_pack_weights_wino_x3 stores per (cout block, chunk, wave,
//       point, cout half, piece, lane) the 16 bytes that lane feeds to its MFMAs -- twelve coalesced 16-byte loads
//       per lane and chunk, issued one point (12 MFMAs) ahead;
//     * 24 MFMAs per wave and chunk, ONE barrier per chunk, V three chunks deep so that the fragments of a chunk's
//       first point are prepared behind the previous chunk's MFMAs;
//   after the channel loop the accumulators go through LDS once, each thread applies A^T M A for four (cout, tile)
//   pairs and the usual epilogue (scale / shift, residual, floor, fused 2x2 max-pool, NCHW or head scatter).
#include "ct_common.h"
#include "ct_wino_pack.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int CC = ctdet::kWinoX3CC;        // 16 channels per chunk = one MFMA k-group
constexpr int TB = 32;                      // tiles per workgroup
constexpr int KB = ctdet::kWinoKB;          // 64 output channels per workgroup
constexpr int PT_STRIDE = CC * TB;          // 512 floats per point: [channel 16][tile 32]
constexpr int V_FLOATS = 16 * PT_STRIDE;    // 8192 floats = 32 KB: V of one chunk
constexpr int U_CHUNK_BYTES = ctdet::kWinoX3ChunkBytes;    // [wave 8][x 2][cb 2][piece 3][lane 64][16 B] = 96 KB
constexpr int MS = 40, MXI = 64 * MS;       // output staging M[point 16][cout 64][tile 32 (+8 pad)]
constexpr int WX3_LDS_BYTES = 16 * MXI * 4; // 160 KB (the main loop uses 96 KB)

struct WinoX3Args {
    const float* in;
    const unsigned char* U;
    const float* scale;
    const float* shift;
    const float* res;
    const float* lo;
    float* out;
    unsigned in_bytes, out_bytes, res_bytes, u_bytes;
    int Cin, H, W, in_ctot, in_coff;
    int M, chunks, kblocks;
    int TY, TX, NT, tile_blocks;
    int out_ctot, out_coff, res_ctot, res_coff;
    float res_scale;
    int relu;
    float* pool_out;         // optional fused 2x2 / stride 2 max-pool of the activation (NCHW), else null
    int pool_ctot, pool_coff, pool_oh, pool_ow, write_full;
    int nseg;                // > 0: channels-last scatter into the flattened head buffers (ct_out_segment)
    ct_out_segment seg[3];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// x = hi + mid + lo exactly (3 x 8 significant bits by truncation); the upper halves of the three words are the pieces
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l)
{
    h = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    l = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, m));
}

__device__ __forceinline__ int pack_hi(unsigned e0, unsigned e1)      // [bf16 e0 | bf16 e1 << 16]
{
    return (int)__builtin_amdgcn_perm(e1, e0, 0x07060302u);
}

// Epilogue of one (cout, 2x2 output tile): y = the output-transformed sums [row 0: x, x+1 | row 1: x, x+1].
// scale / shift, residual, floor, the 2x2 pooling window, NCHW or head-scatter stores (the arithmetic of ct_wino.hip).
__device__ __forceinline__ void emit_tile(const WinoX3Args& a, const __amdgpu_buffer_rsrc_t rout,
                                          const __amdgpu_buffer_rsrc_t rres, const int n, const int ty, const int tx,
                                          const int co, const float (&y)[4])
{
    const int OH = a.H, OW = a.W;                  // pad 1, stride 1: same spatial size
    const int oy = 2 * ty, ox = 2 * tx;
    const bool two = ox + 1 < OW;
    const float sc = a.scale[co], sh = a.shift[co];
    const float lo = a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY);
    float pooled = -INFINITY;
#pragma unroll
    for (int q = 0; q < 2; ++q) {              // output row oy + q: two adjacent pixels, one 8-byte access
        const int yy = oy + q;
        if (yy >= OH) continue;
        float v0 = y[2 * q] * sc + sh, v1 = y[2 * q + 1] * sc + sh;
        if (a.res) {
            const unsigned ro = (unsigned)(((((size_t)n * a.res_ctot + a.res_coff + co) * OH + yy) * OW + ox) * 4);
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, ro, 0, 0));
            const float r1 = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rres, two ? ro + 4 : (unsigned)kInvalidOff, 0, 0));
            v0 = v0 * a.res_scale + r0;
            v1 = v1 * a.res_scale + r1;
        }
        v0 = v0 < lo ? lo : v0;                 // NaN propagates (fmaxf would turn it into the floor)
        v1 = v1 < lo ? lo : v1;
        pooled = fmaxf(pooled, two ? fmaxf(v0, v1) : v0);
        if (!a.write_full) continue;
        if (a.nseg > 0) {          // heads: permute(0,2,3,1) + view + cat of models/RFB_Net_vgg.py:239-248
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end) {
                    float* dst = a.seg[g].ptr + (size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                 (size_t)(yy * OW + ox) * a.seg[g].pix_stride + (co - a.seg[g].co_begin);
                    dst[0] = v0;
                    if (two) dst[a.seg[g].pix_stride] = v1;
                }
            continue;
        }
        const unsigned oo = (unsigned)(((((size_t)n * a.out_ctot + a.out_coff + co) * OH + yy) * OW + ox) * 4);
        if (two) {
            i32x2 pk;
            pk.x = __builtin_bit_cast(int, v0);
            pk.y = __builtin_bit_cast(int, v1);
            __builtin_amdgcn_raw_buffer_store_b64(pk, rout, oo, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), rout, oo, 0, 0);
        }
    }
    // the 2x2 output tile IS the pooling window of MaxPool2d(2, 2[, ceil_mode]) (models/RFB_Net_vgg.py:328-330)
    if (a.pool_out && ty < a.pool_oh && tx < a.pool_ow)
        a.pool_out[(((size_t)n * a.pool_ctot + a.pool_coff + co) * a.pool_oh + ty) * a.pool_ow + tx] = pooled;
}

// PIN: every MFMA slot (one MFMA + its slice of side work) is closed with a scheduling barrier, so the instruction
// stream is the source order below; without it the compiler is free to regroup the side work.
// Two accumulators per output block (DUAL): the hi.hi products in acc, the five small ones in acs.  A one-accumulator
// build of this kernel runs at the same speed (measured), so only the accurate one exists.
template <bool PIN>
__global__ __launch_bounds__(512) void wino_f2x2_3x3_x3(const WinoX3Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx -> (XCD-local sequence, cout block fastest): the cout blocks of a tile block run at the same time on the
    // same XCD and share its input patches through one L2 (ct_wino.hip)
    const int jx = blockIdx.x >> 3;
    const int kb = jx % a.kblocks;
    const int tblk = (jx / a.kblocks) * 8 + (blockIdx.x & 7);
    if (tblk >= a.tile_blocks) return;
    const int tb0 = tblk * TB;
    const int HW = a.H * a.W;

    // ---- patch-loader role: tile = l31, channel in chunk = 2 * wave + h.  One 16-byte buffer load per patch row
    // (4 pixels from x0 = 2tx-1, dword aligned); rows outside the image use the out-of-range offset (-> zeros), the left
    // padding column (tx == 0) is handled by loading from x = 0 and shifting the unpack, the right ones by masks.
    int voffr[4];
    bool lp, m2, m3;
    {
        const int T = tb0 + l31;
        const bool live = T < a.NT;
        const int n = T / (a.TY * a.TX);
        const int rem = T - n * (a.TY * a.TX);
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        lp = tx == 0;
        m2 = x0 + 2 < a.W;
        m3 = x0 + 3 < a.W;
        const long base = (((long)n * a.in_ctot + a.in_coff + h) * a.H + y0) * (long)a.W + x0 + (lp ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.H;
            voffr[i] = ok ? (int)((base + (long)i * a.W) * 4) : kInvalidOff;
        }
    }
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(a.U, a.u_bytes);
    const int last = a.chunks - 1;
    const int chan_pair_bytes = 2 * wave * HW * 4;

    auto load_patch = [&](int c, i32x4 (&r)[4]) {
        const int soff = c * (CC * HW * 4) + chan_pair_bytes;       // wave-uniform channel offset (bytes)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[i], soff, 0);
    };
    auto unpack_row = [&](const i32x4& r, float* d) {
        // reinterpret the WHOLE vector before taking components (bit_cast<float>(int_vector.y) picks component 0 here)
        const f32x4 q = __builtin_bit_cast(f32x4, r);
        const float vx = q.x, vy = q.y, vz = q.z, vw = q.w;
        d[0] = lp ? 0.f : vx;
        d[1] = lp ? vx : vy;
        d[2] = m2 ? (lp ? vy : vz) : 0.f;
        d[3] = m3 ? (lp ? vz : vw) : 0.f;
    };
    auto col_pass = [&](const float* d, float* t, int j) {            // t = B^T d, column j
        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    };
    // row i of V = t B, written to V[point 4i + j][channel][tile]: the address is lane-linear (wave * 64 + lane)
    float* const vw_base = lds + wave * 64 + lane;
    auto row_pass_store = [&](const float* t, int i, int buf) {
        float* vp = vw_base + buf * V_FLOATS + (i * 4) * PT_STRIDE;
        vp[0 * PT_STRIDE] = t[i * 4 + 0] - t[i * 4 + 2];
        vp[1 * PT_STRIDE] = t[i * 4 + 1] + t[i * 4 + 2];
        vp[2 * PT_STRIDE] = t[i * 4 + 2] - t[i * 4 + 1];
        vp[3 * PT_STRIDE] = t[i * 4 + 1] - t[i * 4 + 3];
    };
    auto transform_store = [&](const i32x4 (&r)[4], int buf) {        // prologue form: the whole patch at once
        float d[16], t[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) unpack_row(r[i], d + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) col_pass(d, t, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) row_pass_store(t, i, buf);
    };

    // ---- B fragments: lane (tile l31, k-group h) reads channels 8h .. 8h+7 of a point and splits them
    const float* const vr_base = lds + (8 * h) * TB + l31;
    auto read_raw = [&](int buf, int x, float (&raw)[8]) {
        const float* p = vr_base + buf * V_FLOATS + (2 * wave + x) * PT_STRIDE;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = p[e * TB];
    };
    auto split_pair = [&](const float (&raw)[8], int q, i32x4 (&fb)[3]) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(raw[2 * q], h0, m0, l0);
        split3(raw[2 * q + 1], h1, m1, l1);
        fb[0][q] = pack_hi(h0, h1);
        fb[1][q] = pack_hi(m0, m1);
        fb[2][q] = pack_hi(l0, l1);
    };

    // ---- A fragments straight from global memory in MFMA register order
    const int u_voff = wave * (12 * 1024) + lane * 16;
    const int u_kb = kb * a.chunks;
    auto load_u = [&](int c, int x, int j, i32x4 (&ua)[6]) {           // j = cb * 3 + piece
        const int soff = (u_kb + c) * U_CHUNK_BYTES + (x * 6 + j) * 1024;
        ua[j] = __builtin_amdgcn_raw_buffer_load_b128(rU, u_voff, soff, 0);
    };

    f32x16 acc[2][2];                              // [point][cout half]: the hi.hi products
    f32x16 acs[2][2];                              // the sum of the five small products
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[x][i][r] = 0.f;
                acs[x][i][r] = 0.f;
            }

    // smallest products first: (mid, mid), (lo, hi), (hi, lo), (mid, hi), (hi, mid), then (hi, hi)   [A piece, B piece]
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#define WX3_MFMA(X, S, UA, FB)                                                                                     \
    do {                                                                                                           \
        constexpr int t_ = (S) >> 1, cb_ = (S) & 1;                                                                \
        if (t_ < 5)                                                                                                \
            acs[X][cb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                 \
                __builtin_bit_cast(bf16x8, UA[cb_ * 3 + PA[t_]]), __builtin_bit_cast(bf16x8, FB[PB[t_]]),          \
                acs[X][cb_], 0, 0, 0);                                                                             \
        else                                                                                                       \
            acc[X][cb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                 \
                __builtin_bit_cast(bf16x8, UA[cb_ * 3 + PA[t_]]), __builtin_bit_cast(bf16x8, FB[PB[t_]]),          \
                acc[X][cb_], 0, 0, 0);                                                                             \
    } while (0)
#define WX3_PIN() do { if (PIN) __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: V(0), V(1) in LDS, patch(2) in registers, fragments of (chunk 0, point 0)
    i32x4 rw[4];
    i32x4 ua0[6], ua1[6];
    i32x4 fb0[3], fb1[3];
    {
        i32x4 r0[4], r1[4];
        load_patch(0, r0);
        load_patch(min(1, last), r1);
        load_patch(min(2, last), rw);
#pragma unroll
        for (int j = 0; j < 6; ++j) load_u(0, 0, j, ua0);
        transform_store(r0, 0);
        transform_store(r1, 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        float raw[8];
        read_raw(0, 0, raw);
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(raw, q, fb0);
    }

    int b0 = 0;                                    // buffer of chunk c; (c+1) -> b1, (c+2) -> b2
    for (int c = 0; c < a.chunks; ++c) {
        const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        const int cn = min(c + 1, last), cp3 = min(c + 3, last);
        float d[16], t[16];
        float raw[8];
        // ---- point 0 of chunk c.  Behind the MFMAs: fragments of point 1 (V(c)), its U, B^T d of patch(c+2),
        // the loads of patch(c+3)
        read_raw(b0, 1, raw);
        WX3_PIN(); WX3_MFMA(0, 0, ua0, fb0);  load_u(c, 1, 0, ua1);
        WX3_PIN(); WX3_MFMA(0, 1, ua0, fb0);  unpack_row(rw[0], d + 0); unpack_row(rw[1], d + 4); load_u(c, 1, 1, ua1);
        WX3_PIN(); WX3_MFMA(0, 2, ua0, fb0);  unpack_row(rw[2], d + 8); unpack_row(rw[3], d + 12); load_u(c, 1, 2, ua1);
        WX3_PIN(); WX3_MFMA(0, 3, ua0, fb0);  split_pair(raw, 0, fb1); load_u(c, 1, 3, ua1);
        WX3_PIN(); WX3_MFMA(0, 4, ua0, fb0);  split_pair(raw, 1, fb1); load_u(c, 1, 4, ua1);
        WX3_PIN(); WX3_MFMA(0, 5, ua0, fb0);  split_pair(raw, 2, fb1); load_u(c, 1, 5, ua1);
        WX3_PIN(); WX3_MFMA(0, 6, ua0, fb0);  split_pair(raw, 3, fb1);
        WX3_PIN(); WX3_MFMA(0, 7, ua0, fb0);  col_pass(d, t, 0); col_pass(d, t, 1);
        WX3_PIN(); WX3_MFMA(0, 8, ua0, fb0);  col_pass(d, t, 2); col_pass(d, t, 3);
        WX3_PIN(); WX3_MFMA(0, 9, ua0, fb0);
        {
            const int soff = cp3 * (CC * HW * 4) + chan_pair_bytes;
            rw[0] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[0], soff, 0);
            rw[1] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[1], soff, 0);
            WX3_PIN(); WX3_MFMA(0, 10, ua0, fb0);
            rw[2] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[2], soff, 0);
            rw[3] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[3], soff, 0);
        }
        WX3_PIN(); WX3_MFMA(0, 11, ua0, fb0);
        // ---- point 1 of chunk c.  Behind the MFMAs: fragments of (chunk c+1, point 0) from V(c+1), its U, the row
        // pass of patch(c+2) into V(c+2)
        read_raw(b1, 0, raw);
        WX3_PIN(); WX3_MFMA(1, 0, ua1, fb1);  load_u(cn, 0, 0, ua0);
        WX3_PIN(); WX3_MFMA(1, 1, ua1, fb1);  row_pass_store(t, 0, b2); load_u(cn, 0, 1, ua0);
        WX3_PIN(); WX3_MFMA(1, 2, ua1, fb1);  row_pass_store(t, 1, b2); load_u(cn, 0, 2, ua0);
        WX3_PIN(); WX3_MFMA(1, 3, ua1, fb1);  split_pair(raw, 0, fb0); load_u(cn, 0, 3, ua0);
        WX3_PIN(); WX3_MFMA(1, 4, ua1, fb1);  split_pair(raw, 1, fb0); load_u(cn, 0, 4, ua0);
        WX3_PIN(); WX3_MFMA(1, 5, ua1, fb1);  split_pair(raw, 2, fb0); load_u(cn, 0, 5, ua0);
        WX3_PIN(); WX3_MFMA(1, 6, ua1, fb1);  split_pair(raw, 3, fb0);
        WX3_PIN(); WX3_MFMA(1, 7, ua1, fb1);  row_pass_store(t, 2, b2);
        WX3_PIN(); WX3_MFMA(1, 8, ua1, fb1);  row_pass_store(t, 3, b2);
        WX3_PIN(); WX3_MFMA(1, 9, ua1, fb1);
        WX3_PIN(); WX3_MFMA(1, 10, ua1, fb1);
        WX3_PIN(); WX3_MFMA(1, 11, ua1, fb1);
        WX3_PIN();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        b0 = b1;
    }
#undef WX3_MFMA
#undef WX3_PIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][i][r] += acs[x][i][r];

    // ---- output transform through LDS  M[point][cout 64][tile 32], row stride 40 floats: the two half-waves of an
    // accumulator store (k and k+4) and of a transform read land on disjoint banks (ct_wino.hip)
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                lds[(2 * wave + x) * MXI + k * MS + l31] = acc[x][i][r];
            }
    __syncthreads();
    const int tl = tid & 31;
    const int T = tb0 + tl;
    const bool live = T < a.NT;
    const int n = T / (a.TY * a.TX);
    const int rem = T - n * (a.TY * a.TX);
    const int ty = rem / a.TX, tx = rem - ty * a.TX;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = 8 * wave + it + 4 * ((tid >> 5) & 1);
        const int co = kb * KB + k;
        float m[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = lds[e * MXI + k * MS + tl];
        if (!live || co >= a.M) continue;
        float u0[4], u1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
            u1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
        }
        float y[4] = {u0[0] + u0[1] + u0[2], u0[1] - u0[2] - u0[3], u1[0] + u1[1] + u1[2], u1[1] - u1[2] - u1[3]};
        emit_tile(a, rout, rres, n, ty, tx, co, y);
    }
}

bool winox3_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->cin % CC == 0 && d->nseg >= 0 && d->nseg <= 3 && (d->nseg == 0 || !d->res) && !d->transposed &&
           d->oh == d->h && d->ow == d->w;
}

}  // namespace

extern "C" int ct_conv_wino_x3_supported(const ct_conv_desc* d) { return d && winox3_ok(d) ? 1 : 0; }

extern "C" size_t ct_conv_wino_x3_packed_bytes(int cin, int cout)
{
    if (cin <= 0 || cout <= 0 || cin % CC) return 0;
    return (size_t)((cout + KB - 1) / KB) * (cin / CC) * U_CHUNK_BYTES;
}

extern "C" int ct_conv_pack_weights_wino_x3(const float* const* w, const int* cout, int nparts, int cin,
                                            void* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 0, 23, (float*)upacked, stream, "ct_conv_pack_weights_wino_x3");
}

extern "C" int ct_conv_pack_weights_wino_x3_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                                  void* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 1, 23, (float*)upacked, stream,
                                "ct_conv_pack_weights_wino_x3_dgrad");
}

// variant 1: two accumulators, eight waves (the only one left: the four-wave / two-workgroups-per-CU form with one accumulator,
// variant 2 of rounds 4-5, never won a layer in the pipeline and was removed in round 6)
static int launch_wino_x3(const ct_conv_desc* d, const void* upacked, int dual, float* pool_out, int pool_ctot,
                          int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream, const char* who)
{
    CT_REQUIRE(d && upacked, "%s: null pointer", who);
    CT_REQUIRE(d->in && (d->out || d->nseg > 0) && d->scale && d->shift, "%s: null tensor", who);
    if (!winox3_ok(d))
        return ctdet::fail(CT_ERR_UNSUPPORTED, "%s: needs 3x3 stride 1 dilation 1 pad 1, cin %% 16 == 0 "
                           "(got %dx%d s%d d%d p%d cin=%d nseg=%d)", who, d->kh, d->kw, d->stride, d->dil,
                           d->pad_h, d->cin, d->nseg);
    CT_REQUIRE(d->batch > 0 && d->cout > 0, "%s: bad shape", who);
    CT_REQUIRE(write_full || pool_out, "%s: nothing to write", who);
    if (pool_out) {
        CT_REQUIRE(pool_coff >= 0 && pool_coff + d->cout <= pool_ctot, "%s: pooled output slice", who);
        CT_REQUIRE((pool_oh == d->oh / 2 || pool_oh == (d->oh + 1) / 2) && (pool_ow == d->ow / 2 || pool_ow == (d->ow + 1) / 2),
                   "%s: pooled size %dx%d for a %dx%d map", who, pool_oh, pool_ow, d->oh, d->ow);
    }
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "%s: input slice", who);
    if (d->nseg == 0)
        CT_REQUIRE(d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "%s: output slice", who);
    else {
        CT_REQUIRE(!pool_out && write_full, "%s: pooling with segmented output", who);
        for (int g = 0; g < d->nseg; ++g) CT_REQUIRE(d->seg[g].ptr, "%s: null segment", who);
    }
    CT_REQUIRE(!d->res || (d->res_coff >= 0 && d->res_coff + d->cout <= d->res_ctot), "%s: residual slice", who);
    const long long img_in_bytes = (long long)d->in_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_in_bytes < kMaxBufBytes, "%s: one image exceeds 2 GiB", who);
    const long long img_out_bytes = d->nseg ? 4 : (long long)d->out_ctot * d->oh * d->ow * 4;
    const long long img_res_bytes = d->res ? (long long)d->res_ctot * d->oh * d->ow * 4 : 0;
    CT_REQUIRE(img_out_bytes < kMaxBufBytes && img_res_bytes < kMaxBufBytes, "%s: one image exceeds 2 GiB", who);
    const size_t u_bytes = ct_conv_wino_x3_packed_bytes(d->cin, d->cout);
    CT_REQUIRE(u_bytes < (size_t)kMaxBufBytes, "%s: packed weights exceed 2 GiB", who);
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_in_bytes, std::max(img_out_bytes, img_res_bytes)));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            const void* f8[] = {(const void*)wino_f2x2_3x3_x3<false>, (const void*)wino_f2x2_3x3_x3<true>};
            for (const void* f : f8)
                if (attr_err == hipSuccess)
                    attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WX3_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    const int OHW = d->oh * d->ow;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        WinoX3Args a{};
        a.in = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.U = (const unsigned char*)upacked;
        a.u_bytes = (unsigned)u_bytes;
        a.scale = d->scale; a.shift = d->shift; a.lo = d->lo;
        a.res = d->res ? d->res + (size_t)b0 * d->res_ctot * OHW : nullptr;
        a.out = d->nseg ? nullptr : d->out + (size_t)b0 * d->out_ctot * OHW;
        a.nseg = d->nseg;
        for (int g = 0; g < d->nseg; ++g) {
            a.seg[g] = d->seg[g];
            a.seg[g].ptr += (size_t)b0 * d->seg[g].img_stride;
        }
        a.in_bytes = (unsigned)(img_in_bytes * nb);
        a.out_bytes = (unsigned)(img_out_bytes * nb);
        a.res_bytes = (unsigned)(img_res_bytes * nb);
        a.Cin = d->cin; a.H = d->h; a.W = d->w; a.in_ctot = d->in_ctot; a.in_coff = d->in_coff;
        a.M = d->cout; a.chunks = d->cin / CC;
        a.TY = (d->oh + 1) / 2; a.TX = (d->ow + 1) / 2;
        a.NT = nb * a.TY * a.TX;
        a.tile_blocks = (a.NT + TB - 1) / TB;
        a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
        a.res_ctot = d->res_ctot; a.res_coff = d->res_coff; a.res_scale = d->res_scale;
        a.relu = d->relu;
        a.pool_out = pool_out ? pool_out + (size_t)b0 * pool_ctot * pool_oh * pool_ow : nullptr;
        a.pool_ctot = pool_ctot; a.pool_coff = pool_coff; a.pool_oh = pool_oh; a.pool_ow = pool_ow;
        a.write_full = write_full;
        a.kblocks = (d->cout + KB - 1) / KB;
        // 8 XCD-local sequences of (tile block group, cout block); sequences past the last tile block exit at once
        const int groups = (a.tile_blocks + 7) / 8;
        // every MFMA slot closed by a scheduling barrier (kernel template parameter PIN): measured +5 .. 8 %;
        // CTDET_WX3_PIN=0 selects the unpinned builds (A/B measurements)
        static const bool pin = [] { const char* e = getenv("CTDET_WX3_PIN"); return !e || e[0] != '0'; }();
        const dim3 grid(8 * groups * a.kblocks);
        if (pin) hipLaunchKernelGGL((wino_f2x2_3x3_x3<true>), grid, dim3(512), WX3_LDS_BYTES, st, a);
        else hipLaunchKernelGGL((wino_f2x2_3x3_x3<false>), grid, dim3(512), WX3_LDS_BYTES, st, a);
        CT_LAUNCH_CHECK("wino_f2x2_3x3_x3");
    }
    return CT_OK;
}

extern "C" int ct_conv2d_wino_x3_pool_fwd(const ct_conv_desc* d, const void* upacked, int dual, float* pool_out,
                                          int pool_ctot, int pool_coff, int pool_oh, int pool_ow, int write_full,
                                          ct_stream_t stream)
{
    if (dual != 1) return ctdet::fail(CT_ERR_INVALID, "ct_conv2d_wino_x3_fwd: variant %d (only 1 = two accumulators exists)", dual);
    return launch_wino_x3(d, upacked, dual, pool_out, pool_ctot, pool_coff, pool_oh, pool_ow, write_full, stream,
                          "ct_conv2d_wino_x3_fwd");
}

extern "C" int ct_conv2d_wino_x3_fwd(const ct_conv_desc* d, const void* upacked, int dual, ct_stream_t stream)
{
    return ct_conv2d_wino_x3_pool_fwd(d, upacked, dual, nullptr, 0, 0, 0, 0, 1, stream);
}
