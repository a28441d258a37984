// libctdet: Winograd F(4x4,3x3) convolution on the fp32 MFMA path -- the large-tile variant of ct_wino.hip for the
// 3x3 / stride 1 / dilation 1 / pad 1 layers of the RFBNet-VGG stack (models/RFB_Net_vgg.py:219-227).  Same
// ct_conv_desc contract and fused epilogue as ct_conv2d_wino_fwd; 36 multiplications per 16 outputs instead of
// F(2x2,3x3)'s 16 per 4: 4x fewer than the direct convolution, 1.78x fewer than F(2x2,3x3).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        per 4x4 output tile, 6x6 input patch d
//   B^T, G, A^T: Cook-Toom matrices of the interpolation points 0, +-3/4, +-3/2, inf (ct_wino4_points.h: half the
//   rms and a quarter of the worst-case rounding error of the textbook points 0, +-1, +-2, inf, which rounds 1-2 used;
//   measured in tests/test_gpu_wino.py against fp64).
//
// One fused kernel; only the pre-transformed weights U ever exist in HBM in the transform domain:
//   workgroup (512 threads, 8 waves) = 32 output tiles x 64 output channels, loops over 8-channel chunks
//     * wave w = (transform-point group xg = w>>1 of 9 points, cout half w&1): nine 32x32 accumulator blocks
//       (144 registers), 36 v_mfma_f32_32x32x2_f32 per chunk.  A fragments (U) are never in LDS: the pack kernel
//       stores per (cout block, chunk, wave, point, lane) the 4 floats that lane feeds to its MFMAs, one coalesced
//       16-byte load per point and chunk, re-issued into the same registers right after their last use (a whole
//       chunk of latency).  B fragments (V) are one conflict-free ds_read_b32 each;
//     * patch loader: a LANE PAIR owns one (tile, channel) patch.  Each lane loads three columns of the six rows
//       (six 12-byte buffer loads), runs the column pass B^T d on them, swaps three rows with its partner through
//       DPP (quad_perm 1,0,3,2: no LDS, no full 36-value patch in one lane's registers) and finishes the row pass
//       for three of the six transform rows: 18 ds_write_b32 per lane and chunk;
//     * V double-buffered in LDS (2 x 36 KB), one barrier per chunk;
//   after the channel loop the accumulators go through LDS in two passes of 32 couts (144 KB), each thread applies
//   A^T M A for two (cout, tile) pairs per pass and the usual epilogue (*scale + shift, residual, ReLU / per-channel
//   floor, optional fused 2x2 max-pool -- a 4x4 tile holds four pooling windows --, NCHW or head scatter).
#include "ct_common.h"
#include "ct_wino_pack.h"
#include "ct_wino4_points.h"
#include "ct_wino4_emit.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int CC = 8;                       // channels per chunk
constexpr int TB = 32;                      // tiles per workgroup
constexpr int KB = 64;                      // output channels per workgroup
constexpr int NXI = 36;                     // transform points
constexpr int XS = 256;                     // floats per point and chunk: [channel pair 4][h 2][tile 32]
constexpr int VSKEW = 16;                   // points 18..35 (written by the odd lane of a pair) start 16 banks later
constexpr int VBUF = NXI * XS + 32;         // 9248 floats = 36 KB: V of one chunk
constexpr int UCHUNK = ctdet::kWino4ChunkFloats;   // 18432 floats: U of one (cout block, chunk): [wave][point 9][lane][4]
constexpr int MXI = 32 * 32;                // output staging M[point][cout 32][tile 32]
constexpr int W4_LDS_BYTES = NXI * MXI * 4; // 144 KB (the main loop uses 72 KB)

struct Wino4Args {
    const float* in;
    const float* U;
    const float* scale;
    const float* shift;
    const float* res;
    const float* lo;
    float* out;
    unsigned in_bytes, out_bytes, res_bytes;
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
    // split over input channels (small maps: too few tiles to fill the chip): blockIdx.y owns chunks
    // [y * chunks_per_slice, ...) and stores its raw output-transformed sums in slab ws[y][cout][pixel];
    // wino4_slab_epilogue adds the slabs in slice order and applies the epilogue (no atomics: results do not depend
    // on timing)
    float* ws;
    int slices, chunks_per_slice, Npix;
    // stream-K (desc->ksplit == -2: the launch has the device to itself): a persistent grid of 8 x wgs_per_xcd
    // workgroups; the (work item, chunk) units of every XCD's item sequence are cut into equal contiguous ranges, so the
    // last round of a launch is as full as the others.  A range that starts or ends inside an item stores that item's
    // output-transformed partial sums in one of its two slab slots (sk_ws[(block * 2 + slot)][cout 64][tile 32][16]);
    // wino4_streamk_fixup adds an item's slabs in chunk order and applies the epilogue.
    int streamk;
    float* sk_ws;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

using ctdet::w4::bt6;      // x -> B^T x, m -> A^T m for the points 0, +-3/4, +-3/2, inf (ct_wino4_points.h)
using ctdet::w4::at4;

__device__ __forceinline__ float swap_pair(float x)       // value of the other lane of the pair (lane ^ 1)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

// Epilogue of one (cout, 4x4 output tile): y = the output-transformed sums.  Channel-split launches (a.ws) store them
// raw; everything else gets scale/shift, residual, floor, the four 2x2 pooling windows, NCHW or head-scatter stores.
__device__ __forceinline__ void emit_tile(const Wino4Args& a, const __amdgpu_buffer_rsrc_t rout,
                                          const __amdgpu_buffer_rsrc_t rres, const int n, const int ty, const int tx,
                                          const int co, const float (&y)[4][4])
{
    const int OH = a.H, OW = a.W;                      // pad 1, stride 1: same spatial size
    const int oy = 4 * ty, ox = 4 * tx;
    const bool c1 = ox + 1 < OW, c2 = ox + 2 < OW, c3 = ox + 3 < OW;
    if (a.ws) {
        float* slab = a.ws + ((size_t)blockIdx.y * a.M + co) * a.Npix + (size_t)n * OH * OW + ox;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = oy + i;
            if (yy >= OH) continue;
            float* row = slab + yy * OW;
            row[0] = y[i][0];
            if (c1) row[1] = y[i][1];
            if (c2) row[2] = y[i][2];
            if (c3) row[3] = y[i][3];
        }
        return;
    }
    ctdet::w4::emit_tile4(a, rout, rres, n, ty, tx, co, y);
}

// One segment of work: chunks [c_begin, c_end) of work item vb (virtual block index: XCD = vb & 7).  sk_slab == nullptr:
// the segment covers what its launch mode expects and ends in the fused epilogue (or the channel-split slab of a.ws);
// else the output-transformed partial sums go to sk_slab[cout 64][tile 32][16].
__device__ __forceinline__ void wino4_segment(const Wino4Args& a, const int vb, const int c_begin, const int c_end,
                                              float* const sk_slab, float* const lds)
{
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xg = wave >> 1, kbk = wave & 1;
    // blockIdx -> (XCD-local sequence, cout block fastest), as in ct_wino.hip: the cout blocks of a tile block run
    // together on one XCD and share its input patches through that XCD's L2
    const int jx = vb >> 3;
    const int kb = jx % a.kblocks;
    const int tblk = (jx / a.kblocks) * 8 + (vb & 7);
    if (tblk >= a.tile_blocks) return;
    const int tb0 = tblk * TB;
    const int HW = a.H * a.W;

    // ---- patch-loader role: tile = lane>>1, column half q = lane&1 (columns 3q..3q+2), channel-in-chunk = wave
    const int pt = lane >> 1;
    const bool q = lane & 1;
    int voffr[6];
    bool lp, m0, m1, m2;
    {
        const int T = tb0 + pt;
        const bool live = T < a.NT;
        const int n = T / (a.TY * a.TX);
        const int rem = T - n * (a.TY * a.TX);
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        const int y0 = 4 * ty - 1, xs = 4 * tx - 1 + (q ? 3 : 0);
        lp = xs < 0;                             // left padding column: load from x = 0 and shift the unpack by one
        m0 = !lp && xs < a.W;
        m1 = xs + 1 < a.W;
        m2 = xs + 2 < a.W;
        const long base = (((long)n * a.in_ctot + a.in_coff) * a.H + y0) * (long)a.W + xs + (lp ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.H;
            voffr[i] = ok ? (int)((base + (long)i * a.W) * 4) : kInvalidOff;
        }
    }
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);

    i32x3 raw[6];
    float t[6][3];
    auto load_patch = [&](int c) {
        const int soff = (c * CC + wave) * HW * 4;               // wave-uniform channel offset (bytes)
#pragma unroll
        for (int i = 0; i < 6; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b96(rin, voffr[i], soff, 0);
    };
    // column pass for local column c: t[.][c] = B^T d[.][c]
    auto col_pass = [&](int c) {
        float d[6], o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x3 v = __builtin_bit_cast(f32x3, raw[i]);
            if (c == 0) d[i] = m0 ? v.x : 0.f;
            else if (c == 1) d[i] = m1 ? (lp ? v.x : v.y) : 0.f;
            else d[i] = m2 ? (lp ? v.y : v.z) : 0.f;
        }
        bt6(d, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][c] = o[i];
    };
    // transform rows r (lanes q = 0) / r + 3 (lanes q = 1): the three columns of the other half come from the partner
    const int s_w = wave >> 1, h_w = wave & 1;
    float* const Vw = lds + (q ? 18 * XS + VSKEW : 0) + s_w * 64 + h_w * 32 + pt;
    auto row_pass = [&](int r, int buf) {
        float x[6], v[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float snd = q ? t[r][c] : t[r + 3][c];
            const float rcv = swap_pair(snd);
            x[c] = q ? rcv : t[r][c];
            x[3 + c] = q ? t[r + 3][c] : rcv;
        }
        bt6(x, v);
        float* vw = Vw + buf * VBUF + r * 6 * XS;
#pragma unroll
        for (int j = 0; j < 6; ++j) vw[j * XS] = v[j];
    };

    f32x16 acc[9];

    // ---- weights in registers: u[j] = the 4 A-fragment floats (channel pairs s = 0..3) of transform point 9 xg + j
    const f32x4* Ug = reinterpret_cast<const f32x4*>(a.U + (size_t)kb * a.chunks * UCHUNK) + wave * (9 * 64) + lane;
    f32x4 u[9];
    const int last = c_end - 1;
    const float* const Vr = lds + (9 * xg) * XS + (xg >= 2 ? VSKEW : 0) + lane;

    // prologue: the patches of chunks 0 AND 1 and the weights of chunk 0 leave together (one memory latency, not two
    // in a row; the accumulators are not live yet, so the second patch has registers to wait in)
    load_patch(c_begin);
    i32x3 raw1[6];
    {
        const int soff = (min(c_begin + 1, last) * CC + wave) * HW * 4;
#pragma unroll
        for (int i = 0; i < 6; ++i) raw1[i] = __builtin_amdgcn_raw_buffer_load_b96(rin, voffr[i], soff, 0);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) u[j] = Ug[(size_t)c_begin * (UCHUNK / 4) + j * 64];
    col_pass(0); col_pass(1); col_pass(2);
    row_pass(0, 0); row_pass(1, 0); row_pass(2, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) raw[i] = raw1[i];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
    // One chunk = 36 MFMA slots.  A wave's next MFMA cannot start before the partner wave's MFMA leaves the SIMD's matrix
    // pipe (64 cycles each), and an in-order wave issues nothing while it waits on an MFMA -- so every slot carries its
    // share of the side work BEHIND its MFMA, where it overlaps the partner's: a slice of the transform of patch(c+1)
    // (slots 0..14), one patch row load of chunk c+2 (slots 15..20), the next point's B fragments, the next chunk's U.
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const int cn = min(c + 1, last), cp = min(c + 2, last);
        const float* vr = Vr + buf * VBUF;
        const f32x4* un = Ug + (size_t)cn * (UCHUNK / 4);
        const int soff_p = (cp * CC + wave) * HW * 4;
        float* const vwn = Vw + (buf ^ 1) * VBUF;
        float b[2][4];
        float d[6], o[6], x[6], v[6];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[0][s] = vr[s * 64];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = 4 * j + s;
                const float ua = s == 0 ? u[j].x : s == 1 ? u[j].y : s == 2 ? u[j].z : u[j].w;
                __builtin_amdgcn_s_setprio(1);          // the MFMA wins the issue arbitration against the partner's VALU work
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, b[j & 1][s], acc[j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if (j + 1 < 9 && s < 2) {               // B fragments of the next point: two reads per slot
                    b[(j + 1) & 1][2 * s] = vr[(j + 1) * XS + (2 * s) * 64];
                    b[(j + 1) & 1][2 * s + 1] = vr[(j + 1) * XS + (2 * s + 1) * 64];
                }
                if (s == 3) u[j] = un[j * 64];
                if (m < 6) {                            // column passes: unpack + mask, then B^T
                    const int cc = m >> 1;
                    if ((m & 1) == 0) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const f32x3 q3 = __builtin_bit_cast(f32x3, raw[i]);
                            if (cc == 0) d[i] = m0 ? q3.x : 0.f;
                            else if (cc == 1) d[i] = m1 ? (lp ? q3.x : q3.y) : 0.f;
                            else d[i] = m2 ? (lp ? q3.y : q3.z) : 0.f;
                        }
                    } else {
                        bt6(d, o);
#pragma unroll
                        for (int i = 0; i < 6; ++i) t[i][cc] = o[i];
                    }
                } else if (m < 15) {                    // row passes: exchange, B^T, store
                    const int r = (m - 6) / 3, part = (m - 6) % 3;
                    if (part == 0) {
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const float snd = q ? t[r][cc] : t[r + 3][cc];
                            const float rcv = swap_pair(snd);
                            x[cc] = q ? rcv : t[r][cc];
                            x[3 + cc] = q ? t[r + 3][cc] : rcv;
                        }
                    } else if (part == 1) {
                        bt6(x, v);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 6; ++jj) vwn[(r * 6 + jj) * XS] = v[jj];
                    }
                } else if (m < 21) {
                    raw[m - 15] = __builtin_amdgcn_raw_buffer_load_b96(rin, voffr[m - 15], soff_p, 0);
                }
                W4_PIN();
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef W4_PIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- output transform: two passes of 32 couts through LDS  M[point][cout 32][tile 32]
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                // accumulator row r = 8 pass + rr holds cout kl = (r&3) + 8 (r>>2) + 4h of this wave's half
                const int kk = kbk * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                lds[(9 * xg + j) * MXI + kk * 32 + l31] = pass == 0 ? acc[j][rr] : acc[j][8 + rr];
            }
        __syncthreads();
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 512 * it;
            const int kk = idx >> 5, tl = idx & 31;
            const int co = kb * KB + (kk >> 4) * 32 + 16 * pass + (kk & 15);
            const int T = tb0 + tl;
            if (T >= a.NT || co >= a.M) continue;
            const int n = T / (a.TY * a.TX);
            const int rem = T - n * (a.TY * a.TX);
            const int ty = rem / a.TX, tx = rem - ty * a.TX;
            float z[4][6];
            const float* mp = lds + kk * 32 + tl;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float m[6], y[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = mp[(i * 6 + j) * MXI];
                at4(m, y);
#pragma unroll
                for (int i = 0; i < 4; ++i) z[i][j] = y[i];
            }
            float y[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) at4(z[i], y[i]);
            if (sk_slab) {
                f32x4* dst = reinterpret_cast<f32x4*>(sk_slab + ((size_t)((kk >> 4) * 32 + 16 * pass + (kk & 15)) * 32 + tl) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v4;
                    v4.x = y[i][0]; v4.y = y[i][1]; v4.z = y[i][2]; v4.w = y[i][3];
                    dst[i] = v4;
                }
                continue;
            }
            emit_tile(a, rout, rres, n, ty, tx, co, y);
        }
        __syncthreads();
    }
}

// Schedule of XCD x (= block & 7) in a stream-K launch.  Its work items jx = 0 .. Jx-1 (virtual block 8 jx + x) go to
// its W workgroups in full rounds first -- workgroup w takes items w, W + w, ... whole, in the same order as the plain
// grid, so the cout blocks of a tile block still run side by side on the XCD's L2 -- and only the last, partial round
// (items R W .. Jx-1) is cut by chunks: its (item, chunk) units are divided into W equal contiguous ranges.
struct SkSched { int Jx, R, U; };
__device__ __forceinline__ SkSched sk_sched(const Wino4Args& a, int x, int W)
{
    SkSched s;
    const int nx = a.tile_blocks > x ? (a.tile_blocks - x + 7) / 8 : 0;
    s.Jx = nx * a.kblocks;
    s.R = s.Jx / W;
    const int units = (s.Jx - s.R * W) * a.chunks;
    s.U = (units + W - 1) / W;
    return s;
}

__global__ __launch_bounds__(512) void wino_f4x4_3x3_f32(const Wino4Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!a.streamk) {
        const int c_begin = blockIdx.y * a.chunks_per_slice;
        wino4_segment(a, blockIdx.x, c_begin, min(a.chunks, c_begin + a.chunks_per_slice), nullptr, lds);
        return;
    }
    const int x = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const SkSched sc = sk_sched(a, x, W);
    for (int r = 0; r < sc.R; ++r) wino4_segment(a, (r * W + w) * 8 + x, 0, a.chunks, nullptr, lds);
    const int units = (sc.Jx - sc.R * W) * a.chunks;
    const int u_end = min((w + 1) * sc.U, units);
    bool first = true;
    for (int u = w * sc.U; u < u_end;) {
        const int t = u / a.chunks, c0 = u - t * a.chunks;
        const int c1 = min(a.chunks, c0 + (u_end - u));
        const bool whole = c0 == 0 && c1 == a.chunks;
        float* slab = whole ? nullptr : a.sk_ws + ((size_t)blockIdx.x * 2 + (first ? 0 : 1)) * (KB * TB * 16);
        wino4_segment(a, (sc.R * W + t) * 8 + x, c0, c1, slab, lds);
        u += c1 - c0;
        first = false;
    }
}

// Items of the last round that the stream-K schedule cut: sum of their slabs in chunk order, then the fused epilogue.
// One block per item of the plain grid; everything but the cut items returns at once.
__global__ __launch_bounds__(256) void wino4_streamk_fixup(const Wino4Args a, const int W)
{
    const int vb = blockIdx.x, x = vb & 7, jx = vb >> 3;
    const SkSched sc = sk_sched(a, x, W);
    const int t = jx - sc.R * W;                           // index in the last round
    if (jx >= sc.Jx || t < 0) return;
    const int u0 = t * a.chunks;
    const int w_first = u0 / sc.U, w_last = (u0 + a.chunks - 1) / sc.U;
    if (w_first == w_last) return;                         // one workgroup did the whole item, epilogue included
    const int kb = jx % a.kblocks, tblk = (jx / a.kblocks) * 8 + x;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
    for (int idx = threadIdx.x; idx < KB * TB; idx += 256) {
        const int col = idx >> 5, tl = idx & 31;
        const int co = kb * KB + col, T = tblk * TB + tl;
        if (T >= a.NT || co >= a.M) continue;
        float y[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[i][j] = 0.f;
        for (int w = w_first; w <= w_last; ++w) {
            // the item is workgroup w's first segment iff w's range starts inside it
            const int slot = (w * sc.U) / a.chunks == t ? 0 : 1;
            const f32x4* src = reinterpret_cast<const f32x4*>(
                a.sk_ws + (((size_t)(w * 8 + x) * 2 + slot) * KB * TB + idx) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v4 = src[i];
                y[i][0] += v4.x; y[i][1] += v4.y; y[i][2] += v4.z; y[i][3] += v4.w;
            }
        }
        const int n = T / (a.TY * a.TX);
        const int rem = T - n * (a.TY * a.TX);
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        emit_tile(a, rout, rres, n, ty, tx, co, y);
    }
}

// epilogue of a channel-split launch: sum of the slabs in slice order, then the same arithmetic as the fused one
__global__ __launch_bounds__(256) void wino4_slab_epilogue(const Wino4Args a)
{
    const int OHW = a.H * a.W;
    const long total = (long)a.M * a.Npix;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int co = (int)(idx / a.Npix), P = (int)(idx - (long)co * a.Npix);
        const int n = P / OHW, sp = P - n * OHW;
        float sum = a.ws[idx];
        for (int k = 1; k < a.slices; ++k) sum += a.ws[(size_t)k * total + idx];
        float v = sum * a.scale[co] + a.shift[co];
        if (a.res) v = v * a.res_scale + a.res[((size_t)n * a.res_ctot + a.res_coff + co) * OHW + sp];
        if (a.lo) { const float fl = a.lo[co]; v = v < fl ? fl : v; }      // NaN propagates
        else if (a.relu) v = v < 0.f ? 0.f : v;
        if (a.nseg == 0) {
            a.out[((size_t)n * a.out_ctot + a.out_coff + co) * OHW + sp] = v;
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end)
                    a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                 (size_t)sp * a.seg[g].pix_stride + (co - a.seg[g].co_begin)] = v;
        }
    }
}

bool wino4_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->cin % CC == 0 && d->nseg >= 0 && d->nseg <= 3 && (d->nseg == 0 || !d->res) && !d->transposed &&
           d->oh == d->h && d->ow == d->w;
}

}  // namespace

extern "C" int ct_conv_wino4_supported(const ct_conv_desc* d) { return d && wino4_ok(d) ? 1 : 0; }

extern "C" size_t ct_conv_wino4_packed_floats(int cin, int cout)
{
    if (cin <= 0 || cout <= 0 || cin % CC) return 0;
    return (size_t)((cout + KB - 1) / KB) * (cin / CC) * UCHUNK;
}

extern "C" int ct_conv_pack_weights_wino4(const float* const* w, const int* cout, int nparts, int cin,
                                          float* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 0, 4, upacked, stream, "ct_conv_pack_weights_wino4");
}

extern "C" int ct_conv_pack_weights_wino4_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                                float* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 1, 4, upacked, stream, "ct_conv_pack_weights_wino4_dgrad");
}

extern "C" int ct_conv2d_wino4_pool_fwd(const ct_conv_desc* d, const float* upacked, float* pool_out, int pool_ctot,
                                        int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream)
{
    CT_REQUIRE(d && upacked, "ct_conv2d_wino4_fwd: null pointer");
    CT_REQUIRE(d->in && (d->out || d->nseg > 0) && d->scale && d->shift, "ct_conv2d_wino4_fwd: null tensor");
    if (!wino4_ok(d))
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_wino4_fwd: needs 3x3 stride 1 dilation 1 pad 1, cin %% 8 == 0 "
                           "(got %dx%d s%d d%d p%d cin=%d nseg=%d)", d->kh, d->kw, d->stride, d->dil,
                           d->pad_h, d->cin, d->nseg);
    CT_REQUIRE(d->batch > 0 && d->cout > 0, "ct_conv2d_wino4_fwd: bad shape");
    CT_REQUIRE(write_full || pool_out, "ct_conv2d_wino4_pool_fwd: nothing to write");
    if (pool_out) {
        CT_REQUIRE(pool_coff >= 0 && pool_coff + d->cout <= pool_ctot, "ct_conv2d_wino4_pool_fwd: pooled output slice");
        CT_REQUIRE((pool_oh == d->oh / 2 || pool_oh == (d->oh + 1) / 2) && (pool_ow == d->ow / 2 || pool_ow == (d->ow + 1) / 2),
                   "ct_conv2d_wino4_pool_fwd: pooled size %dx%d for a %dx%d map", pool_oh, pool_ow, d->oh, d->ow);
    }
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_wino4_fwd: input slice");
    if (d->nseg == 0)
        CT_REQUIRE(d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "ct_conv2d_wino4_fwd: output slice");
    else {
        CT_REQUIRE(!pool_out && write_full, "ct_conv2d_wino4_fwd: pooling with segmented output");
        for (int g = 0; g < d->nseg; ++g) CT_REQUIRE(d->seg[g].ptr, "ct_conv2d_wino4_fwd: null segment");
    }
    CT_REQUIRE(!d->res || (d->res_coff >= 0 && d->res_coff + d->cout <= d->res_ctot), "ct_conv2d_wino4_fwd: residual slice");
    const long long img_in_bytes = (long long)d->in_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_in_bytes < kMaxBufBytes, "ct_conv2d_wino4_fwd: one image exceeds 2 GiB");
    const long long img_out_bytes = d->nseg ? 4 : (long long)d->out_ctot * d->oh * d->ow * 4;
    const long long img_res_bytes = d->res ? (long long)d->res_ctot * d->oh * d->ow * 4 : 0;
    CT_REQUIRE(img_out_bytes < kMaxBufBytes && img_res_bytes < kMaxBufBytes, "ct_conv2d_wino4_fwd: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_in_bytes, std::max(img_out_bytes, img_res_bytes)));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)wino_f4x4_3x3_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           W4_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    const int OHW = d->oh * d->ow;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        Wino4Args a{};
        a.in = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.U = upacked;
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
        a.TY = (d->oh + 3) / 4; a.TX = (d->ow + 3) / 4;
        a.NT = nb * a.TY * a.TX;
        a.tile_blocks = (a.NT + TB - 1) / TB;
        a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
        a.res_ctot = d->res_ctot; a.res_coff = d->res_coff; a.res_scale = d->res_scale;
        a.relu = d->relu;
        a.pool_out = pool_out ? pool_out + (size_t)b0 * pool_ctot * pool_oh * pool_ow : nullptr;
        a.pool_ctot = pool_ctot; a.pool_coff = pool_coff; a.pool_oh = pool_oh; a.pool_ow = pool_ow;
        a.write_full = write_full;
        a.kblocks = (d->cout + KB - 1) / KB;
        // split over input channels when the (tile block, cout block) grid cannot fill the chip (desc->ksplit: -1 auto,
        // > 1 forced, else off; needs the caller's slab workspace and an unchunked, unpooled launch)
        a.slices = 1;
        a.chunks_per_slice = a.chunks;
        a.Npix = nb * OHW;
        a.ws = nullptr;
        if (d->ksplit_ws && nb == d->batch && !pool_out && (d->ksplit == -1 || d->ksplit == -2 || d->ksplit > 1)) {
            static const int target = getenv("CTDET_W4_SPLIT_TARGET") ? atoi(getenv("CTDET_W4_SPLIT_TARGET")) : 256;
            const int wgs = a.tile_blocks * a.kblocks;
            int want = d->ksplit > 1 ? d->ksplit : target / std::max(wgs, 1);      // auto: at most one workgroup per CU in total
            const long long slab = (long long)d->cout * a.Npix;
            want = (int)std::min<long long>(std::min(want, a.chunks / 4), d->ksplit_ws_floats / std::max<long long>(slab, 1));
            if (want > 1) {
                a.chunks_per_slice = (a.chunks + want - 1) / want;
                a.slices = (a.chunks + a.chunks_per_slice - 1) / a.chunks_per_slice;
                a.ws = d->ksplit_ws;
            }
        }
        // 8 XCD-local sequences of (tile block group, cout block); sequences past the last tile block exit at once
        const int groups = (a.tile_blocks + 7) / 8;
        const int items = a.tile_blocks * a.kblocks;
        // stream-K: the caller says the launch runs alone (ksplit -2) and the plain grid would end in a ragged round
        constexpr int kCUs = 256, kWgsPerXcd = kCUs / 8;
        const long long sk_floats = (long long)kCUs * 2 * KB * TB * 16;
        a.streamk = 0;
        a.sk_ws = nullptr;
        if (d->ksplit == -2 && d->ksplit_ws && d->ksplit_ws_floats >= sk_floats && nb == d->batch && items > kCUs &&
            a.chunks >= 8) {
            const double rounds = (double)items / kCUs;
            const double frac = rounds - std::floor(rounds);      // fill of the last round
            const double max_frac = getenv("CTDET_W4_SK_FRAC") ? atof(getenv("CTDET_W4_SK_FRAC")) : 0.5;
            if (frac > 0.0 && frac <= max_frac) {
                a.streamk = 1;
                a.sk_ws = d->ksplit_ws;
                a.slices = 1;
                a.chunks_per_slice = a.chunks;
                a.ws = nullptr;
            }
        }
        if (a.streamk) {
            hipLaunchKernelGGL(wino_f4x4_3x3_f32, dim3(kCUs), dim3(512), W4_LDS_BYTES, st, a);
            CT_LAUNCH_CHECK("wino_f4x4_3x3_f32 (stream-K)");
            hipLaunchKernelGGL(wino4_streamk_fixup, dim3(8 * groups * a.kblocks), dim3(256), 0, st, a, kWgsPerXcd);
            CT_LAUNCH_CHECK("wino4_streamk_fixup");
            continue;
        }
        hipLaunchKernelGGL(wino_f4x4_3x3_f32, dim3(8 * groups * a.kblocks, a.slices), dim3(512), W4_LDS_BYTES, st, a);
        CT_LAUNCH_CHECK("wino_f4x4_3x3_f32");
        if (a.slices > 1) {
            const long total = (long)a.M * a.Npix;
            hipLaunchKernelGGL(wino4_slab_epilogue, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, st, a);
            CT_LAUNCH_CHECK("wino4_slab_epilogue");
        }
    }
    return CT_OK;
}

extern "C" int ct_conv2d_wino4_fwd(const ct_conv_desc* d, const float* upacked, ct_stream_t stream)
{
    return ct_conv2d_wino4_pool_fwd(d, upacked, nullptr, 0, 0, 0, 0, 1, stream);
}
