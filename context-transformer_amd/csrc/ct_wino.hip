// libctdet: Winograd F(2x2,3x3) convolution on the fp32 MFMA path for the 3x3 / stride 1 /
// dilation 1 / pad 1 layers of the RFBNet-VGG stack (80 % of its FLOPs: models/RFB_Net_vgg.py:219-227
// VGG trunk and the 3x3 BasicConv layers).  Same ct_conv_desc contract and fused epilogue as
// ct_conv2d_fwd; 2.25x fewer multiplications than the direct implicit GEMM:
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d
//
// One fused kernel, nothing transformed ever touches HBM except the pre-transformed weights U:
//   workgroup (512 threads, 8 waves) = 64 output tiles x 64 output channels, loops over 8-channel chunks
//     * every thread loads ONE 4x4 patch (tile, channel) with four 16-byte buffer loads (one per row; rows
//       outside the image = out-of-range offset -> zeros), applies B^T d B in registers and writes the 16
//       transform-domain values V[xi][c][tile] to LDS;
//     * wave w owns transform points xi = 2w, 2w+1 and runs, per xi, the [64 k] x [64 tiles] x [8 c]
//       GEMM as v_mfma_f32_32x32x2_f32.  Its A fragments (U) never pass through LDS: ct_conv_pack_weights_wino
//       stores, per (cout block, chunk, wave, lane), exactly the 16 floats that lane feeds to its MFMAs, so a
//       chunk is four coalesced 16-byte loads per lane into registers (double buffered).  B fragments (V) are
//       [c-pair][tile][2] in LDS: one conflict-free ds_read_b32 each.  Accumulators: 2 xi x 2x2 blocks x 16 = 128;
//     * V double-buffered in LDS (2 x 32 KB), ONE barrier per chunk, every vector-memory instruction issued
//       behind an MFMA (never back to back);
//   after the channel loop the accumulators go through LDS once (two passes of 32 tiles),
//   each thread applies A^T M A for a (channel, tile) pair and the usual epilogue
//   (*scale + shift, residual, ReLU / per-channel floor, optional fused 2x2 max-pool, NCHW or head scatter).
#include "ct_common.h"
#include "ct_wino_pack.h"
#include <algorithm>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int CC = 8;                       // channels per chunk
constexpr int TB = 64;                      // tiles per workgroup
constexpr int KB = 64;                      // output channels per workgroup
constexpr int XI_STRIDE = (CC / 2) * 64 * 2;          // 512 floats: [s][row 64][h 2]
constexpr int CHUNK_FLOATS = 16 * XI_STRIDE;          // 8192 floats = 32 KB (U or V of one chunk)
static_assert(CHUNK_FLOATS == ctdet::kWino2ChunkFloats && CC == ctdet::kWinoCC && KB == ctdet::kWinoKB, "pack layout");
constexpr int WINO_LDS_BYTES = 16 * 64 * 40 * 4;           // 160 KB: output staging M[16][64][40] (the main loop uses 64 KB)

struct WinoArgs {
    const float* in;
    const float* U;          // [kblocks][chunks][wave 8][piece 4][lane 64][4]
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
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(512) void wino_f2x2_3x3_f32(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx -> (XCD-local sequence, cout block fastest): the cout blocks of a tile block run at the same time on
    // the same XCD (block b is dispatched to XCD b % 8), so the input patches of the tile block come from HBM once
    // and are L2 hits for the other cout blocks.  (Walking all tile blocks per cout block re-read the whole input
    // cout/64 times: 4.2x the algorithmic traffic in the round-1 PMC passes.)
    const int jx = blockIdx.x >> 3;
    const int kb = jx % a.kblocks;
    const int tblk = (jx / a.kblocks) * 8 + (blockIdx.x & 7);
    if (tblk >= a.tile_blocks) return;
    const int tb0 = tblk * TB;
    const int HW = a.H * a.W;

    // ---- patch-loader role: tile = l31 + 32*(wave&1), channel-in-chunk = 2*(wave>>1) + h
    const int tile_l = l31 + 32 * (wave & 1);
    const int s_l = wave >> 1;
    // One 16-byte buffer load per patch row (4 consecutive pixels from x0 = 2tx-1; dword aligned).  Rows outside
    // the image use the out-of-range offset (-> zeros).  The left padding column (tx == 0, x0 = -1) is handled by
    // loading from x = 0 and shifting the unpack by one; the columns right of the image are masked after the load.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    int voffr[4];
    bool lp, m2, m3;
    {
        const int T = tb0 + tile_l;
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

    float* const Vlds = lds;                 // V double buffer only: the weights stay in registers
    auto load_patch = [&](int c, i32x4 (&r)[4]) {
        const int soff = (c * CC + 2 * s_l) * HW * 4;            // wave-uniform channel offset (bytes)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[i], soff, 0);
    };
    auto unpack = [&](const i32x4 (&r)[4], float (&d)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // Reinterpret the WHOLE vector before taking components: on this toolchain
            // bit_cast<float>(int_vector.y) compiles to component 0 for every lane of the vector.
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 q = __builtin_bit_cast(f32x4, r[i]);
            const float vx = q.x, vy = q.y, vz = q.z, vw = q.w;
            d[i * 4 + 0] = lp ? 0.f : vx;
            d[i * 4 + 1] = lp ? vx : vy;
            d[i * 4 + 2] = m2 ? (lp ? vy : vz) : 0.f;
            d[i * 4 + 3] = m3 ? (lp ? vz : vw) : 0.f;
        }
    };
    // V = B^T d B  ->  V[xi][s][tile][h]
    auto store_v = [&](int buf, const i32x4 (&r)[4]) {
        float* Vl = Vlds + buf * CHUNK_FLOATS;
        float d[16], t[16];
        unpack(r, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
            t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
        }
        float* vp = Vl + s_l * 128 + tile_l * 2 + h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vp[(i * 4 + 0) * XI_STRIDE] = t[i * 4 + 0] - t[i * 4 + 2];
            vp[(i * 4 + 1) * XI_STRIDE] = t[i * 4 + 1] + t[i * 4 + 2];
            vp[(i * 4 + 2) * XI_STRIDE] = t[i * 4 + 2] - t[i * 4 + 1];
            vp[(i * 4 + 3) * XI_STRIDE] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };

    f32x16 acc[2][2][2];       // [xi][k block][tile block]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.f;

#define WINO_MFMA(XI, II, JJ, av, bv)                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        acc[XI][II][JJ] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[XI][II][JJ], 0, 0, 0);       \
        __builtin_amdgcn_s_setprio(0);                                                                  \
    } while (0)
#define WINO_PIN() __builtin_amdgcn_sched_barrier(0)
    // ---- weights in registers.  The A fragments a wave needs for a chunk are 16 floats per lane, stored by
    // ct_conv_pack_weights_wino in exactly that order: four coalesced 16-byte loads per lane and chunk, no LDS
    // copy, no LDS reads for A.  Per iteration c:   MFMAs of chunk c  |  U(c+1) -> the other register set  |
    // transform patch(c+1) -> V buffer (c+1)&1  |  patch(c+2) -> registers  |  barrier
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4* Ug = reinterpret_cast<const f32x4*>(a.U + (size_t)kb * a.chunks * CHUNK_FLOATS) + wave * 256 + lane;
    auto load_u = [&](int c, f32x4 (&u)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = Ug[(size_t)c * (CHUNK_FLOATS / 4) + 64 * i];
    };
    struct FragB { float b0[4], b1[4]; };
    auto read_b = [&](int buf, int x, FragB& f) {
        const float* Vl = Vlds + buf * CHUNK_FLOATS + (2 * wave + x) * XI_STRIDE + l31 * 2 + h;
#pragma unroll
        for (int s = 0; s < 4; ++s) { f.b0[s] = Vl[s * 128]; f.b1[s] = Vl[s * 128 + 64]; }
    };
    // A fragments of (x, s): u[2x + (s>>1)] components ((s&1)*2, (s&1)*2 + 1)
#define WINO_A0(UU, XX, SS) ((SS) & 1 ? UU[2 * (XX) + ((SS) >> 1)].z : UU[2 * (XX) + ((SS) >> 1)].x)
#define WINO_A1(UU, XX, SS) ((SS) & 1 ? UU[2 * (XX) + ((SS) >> 1)].w : UU[2 * (XX) + ((SS) >> 1)].y)
#define WINO_4(UU, FF, XX, SS)                                        \
    WINO_MFMA(XX, 0, 0, WINO_A0(UU, XX, SS), FF.b0[SS]);              \
    WINO_MFMA(XX, 0, 1, WINO_A0(UU, XX, SS), FF.b1[SS]);              \
    WINO_MFMA(XX, 1, 0, WINO_A1(UU, XX, SS), FF.b0[SS]);              \
    WINO_MFMA(XX, 1, 1, WINO_A1(UU, XX, SS), FF.b1[SS])
    i32x4 rw[4];
    f32x4 uA[4], uB[4];
    const int last = a.chunks - 1;
    load_u(0, uA);
    load_patch(0, rw);
    store_v(0, rw);
    load_patch(min(1, last), rw);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto body = [&](int c, const f32x4 (&ucur)[4], f32x4 (&unxt)[4]) {
        const int buf = c & 1;
        const int cn = min(c + 1, last), cp = min(c + 2, last);
        const int soff = (cp * CC + 2 * s_l) * HW * 4;
        FragB f0, f1;
        read_b(buf, 0, f0);
        const f32x4* up = Ug + (size_t)cn * (CHUNK_FLOATS / 4);
        WINO_MFMA(0, 0, 0, WINO_A0(ucur, 0, 0), f0.b0[0]); unxt[0] = up[0]; WINO_PIN();
        WINO_MFMA(0, 0, 1, WINO_A0(ucur, 0, 0), f0.b1[0]); unxt[1] = up[64]; WINO_PIN();
        WINO_MFMA(0, 1, 0, WINO_A1(ucur, 0, 0), f0.b0[0]); unxt[2] = up[128]; WINO_PIN();
        WINO_MFMA(0, 1, 1, WINO_A1(ucur, 0, 0), f0.b1[0]); unxt[3] = up[192]; WINO_PIN();
        // x = 0, s = 1: the transform of patch(c+1) in four slices behind the MFMAs
        float d[16], t[16], v[16];
        unpack(rw, d);
        WINO_MFMA(0, 0, 0, WINO_A0(ucur, 0, 1), f0.b0[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
            t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
        }
        WINO_PIN();
        WINO_MFMA(0, 0, 1, WINO_A0(ucur, 0, 1), f0.b1[1]);
#pragma unroll
        for (int j = 2; j < 4; ++j) {
            t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
            t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
        }
        WINO_PIN();
        WINO_MFMA(0, 1, 0, WINO_A1(ucur, 0, 1), f0.b0[1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
            v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
            v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
            v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
        }
        WINO_PIN();
        WINO_MFMA(0, 1, 1, WINO_A1(ucur, 0, 1), f0.b1[1]);
#pragma unroll
        for (int i = 2; i < 4; ++i) {
            v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
            v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
            v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
            v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
        }
        WINO_PIN();
        read_b(buf, 1, f1);
        float* vp = Vlds + (buf ^ 1) * CHUNK_FLOATS + s_l * 128 + tile_l * 2 + h;
#define WINO_VST(e) vp[(e) * XI_STRIDE] = v[e]
#define WINO_LD(i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voffr[i], soff, 0)
        WINO_MFMA(0, 0, 0, WINO_A0(ucur, 0, 2), f0.b0[2]); WINO_VST(0); WINO_VST(1); WINO_VST(2); WINO_VST(3); WINO_PIN();
        WINO_MFMA(0, 0, 1, WINO_A0(ucur, 0, 2), f0.b1[2]); WINO_VST(4); WINO_VST(5); WINO_VST(6); WINO_VST(7); WINO_PIN();
        WINO_MFMA(0, 1, 0, WINO_A1(ucur, 0, 2), f0.b0[2]); WINO_VST(8); WINO_VST(9); WINO_VST(10); WINO_VST(11); WINO_PIN();
        WINO_MFMA(0, 1, 1, WINO_A1(ucur, 0, 2), f0.b1[2]); WINO_VST(12); WINO_VST(13); WINO_VST(14); WINO_VST(15); WINO_PIN();
        WINO_MFMA(0, 0, 0, WINO_A0(ucur, 0, 3), f0.b0[3]); WINO_LD(0); WINO_PIN();
        WINO_MFMA(0, 0, 1, WINO_A0(ucur, 0, 3), f0.b1[3]); WINO_PIN();
        WINO_MFMA(0, 1, 0, WINO_A1(ucur, 0, 3), f0.b0[3]); WINO_LD(1); WINO_PIN();
        WINO_MFMA(0, 1, 1, WINO_A1(ucur, 0, 3), f0.b1[3]); WINO_PIN();
        WINO_MFMA(1, 0, 0, WINO_A0(ucur, 1, 0), f1.b0[0]); WINO_LD(2); WINO_PIN();
        WINO_MFMA(1, 0, 1, WINO_A0(ucur, 1, 0), f1.b1[0]); WINO_PIN();
        WINO_MFMA(1, 1, 0, WINO_A1(ucur, 1, 0), f1.b0[0]); WINO_LD(3); WINO_PIN();
        WINO_MFMA(1, 1, 1, WINO_A1(ucur, 1, 0), f1.b1[0]); WINO_PIN();
        WINO_4(ucur, f1, 1, 1);
        WINO_4(ucur, f1, 1, 2);
        WINO_4(ucur, f1, 1, 3);
#undef WINO_VST
#undef WINO_LD
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    int c = 0;
    for (; c + 1 < a.chunks; c += 2) {          // straight-line pair of iterations: static wait counts
        body(c, uA, uB);
        body(c + 1, uB, uA);
    }
    if (c < a.chunks) body(c, uA, uB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- output transform: two passes of 32 tiles through LDS  M[xi][k 64][tile 32], row stride 40 floats:
    // the two half-waves of an accumulator store (k and k+4) and of a transform read land on disjoint banks
    constexpr int MS = 40, MXI = 64 * MS;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
    for (int tbk = 0; tbk < 2; ++tbk) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    lds[(2 * wave + x) * MXI + k * MS + l31] = tbk == 0 ? acc[x][i][0][r] : acc[x][i][1][r];
                }
        __syncthreads();
        const int tl = tid & 31;
        const int T = tb0 + tbk * 32 + tl;
        const bool live = T < a.NT;
        const int n = T / (a.TY * a.TX);
        const int rem = T - n * (a.TY * a.TX);
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        const int OH = a.H, OW = a.W;                  // pad 1, stride 1: same spatial size
        const int oy = 2 * ty, ox = 2 * tx;
        const bool two = ox + 1 < OW;
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
        __syncthreads();
    }
}

__global__ void wino_pack_kernel(const ctdet::WinoPackArgs p)
{
    ctdet::wino_pack_any(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// all Winograd layers of a training step in one launch: blockIdx.y = recorded item (ct_pack_run)
__global__ void wino_pack_batched_kernel(const ctdet::WinoPackArgs* __restrict__ items)
{
    // by reference: a local copy of the record would put its w[] / mbeg[] arrays (indexed by a run-time part number) into scratch
    ctdet::wino_pack_any(items[blockIdx.y], blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

bool wino_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->cin % CC == 0 && d->nseg >= 0 && d->nseg <= 3 && (d->nseg == 0 || !d->res) && !d->transposed &&
           d->oh == d->h && d->ow == d->w;
}

}  // namespace

extern "C" int ct_conv_wino_supported(const ct_conv_desc* d) { return d && wino_ok(d) ? 1 : 0; }

extern "C" size_t ct_conv_wino_packed_floats(int cin, int cout)
{
    if (cin <= 0 || cout <= 0 || cin % CC) return 0;
    return (size_t)((cout + KB - 1) / KB) * (cin / CC) * CHUNK_FLOATS;
}

int ctdet::pack_wino_any(const float* const* w, const int* cout, int nparts, int cin, int dgrad, int tile, float* upacked,
                         ct_stream_t stream, const char* who)
{
    CT_REQUIRE(w && cout && upacked && nparts >= 1 && nparts <= 6, "%s: bad argument", who);
    ctdet::WinoPackArgs p{};
    int tot = 0;
    for (int i = 0; i < nparts; ++i) {
        CT_REQUIRE(w[i] && cout[i] > 0, "%s: part %d", who, i);
        p.w[i] = w[i];
        p.mbeg[i] = tot;
        tot += cout[i];
    }
    p.mbeg[nparts] = tot;
    p.nparts = nparts;
    p.dgrad = dgrad;
    p.tile = tile;
    p.cin_fwd = cin;
    p.cin = dgrad ? tot : cin;          // input channels of THIS convolution
    p.cout = dgrad ? cin : tot;
    const int cc = tile == 23 || tile == 44 || tile == 46 ? ctdet::kWinoX3CC : CC;          // the bf16x3 layouts (16-channel k-groups)
    CT_REQUIRE(p.cin > 0 && p.cin % cc == 0, "%s: %d input channels, must be a multiple of %d", who, p.cin, cc);
    p.chunks = p.cin / cc;
    p.kblocks = tile == 44 ? (p.cout + ctdet::kWino4sBM - 1) / ctdet::kWino4sBM : (p.cout + KB - 1) / KB;
    p.U = upacked;
    if (ctdet::pack_recording()) {
        ctdet::pack_record(1, &p, sizeof(p));
        return CT_OK;
    }
    const long total = tile == 44 ? (long)p.kblocks * ctdet::kWino4sBM * (p.cin / 8) * 6 : tile == 46 ? (long)p.kblocks * KB * (p.cin / 8) * 6 :
                       (long)p.kblocks * p.chunks * (tile == 4 ? 512 : tile == 23 ? 2048 : ctdet::kWino2ChunkFloats);      // threads
    hipLaunchKernelGGL(wino_pack_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       ctdet::as_stream(stream), p);
    CT_LAUNCH_CHECK("wino_pack_kernel");
    return CT_OK;
}

size_t ctdet::pack_wino_item_bytes() { return sizeof(ctdet::WinoPackArgs); }

int ctdet::launch_pack_wino_batched(const void* items_dev, int n, hipStream_t st)
{
    if (n <= 0) return CT_OK;
    hipLaunchKernelGGL(wino_pack_batched_kernel, dim3(192, n), dim3(256), 0, st, (const ctdet::WinoPackArgs*)items_dev);
    CT_LAUNCH_CHECK("wino_pack_batched_kernel");
    return CT_OK;
}

extern "C" int ct_conv_pack_weights_wino(const float* const* w, const int* cout, int nparts, int cin,
                                         float* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 0, 2, upacked, stream, "ct_conv_pack_weights_wino");
}

extern "C" int ct_conv_pack_weights_wino_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                               float* upacked, ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 1, 2, upacked, stream, "ct_conv_pack_weights_wino_dgrad");
}

extern "C" int ct_conv2d_wino_pool_fwd(const ct_conv_desc* d, const float* upacked, float* pool_out, int pool_ctot,
                                       int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);

extern "C" int ct_conv2d_wino_fwd(const ct_conv_desc* d, const float* upacked, ct_stream_t stream)
{
    return ct_conv2d_wino_pool_fwd(d, upacked, nullptr, 0, 0, 0, 0, 1, stream);
}

extern "C" int ct_conv2d_wino_pool_fwd(const ct_conv_desc* d, const float* upacked, float* pool_out, int pool_ctot,
                                       int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream)
{
    CT_REQUIRE(d && upacked, "ct_conv2d_wino_fwd: null pointer");
    CT_REQUIRE(d->in && (d->out || d->nseg > 0) && d->scale && d->shift, "ct_conv2d_wino_fwd: null tensor");
    if (!wino_ok(d))
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_wino_fwd: needs 3x3 stride 1 dilation 1 pad 1, cin %% 8 == 0 "
                           "(got %dx%d s%d d%d p%d cin=%d nseg=%d)", d->kh, d->kw, d->stride, d->dil,
                           d->pad_h, d->cin, d->nseg);
    CT_REQUIRE(d->batch > 0 && d->cout > 0, "ct_conv2d_wino_fwd: bad shape");
    CT_REQUIRE(write_full || pool_out, "ct_conv2d_wino_pool_fwd: nothing to write");
    if (pool_out) {
        CT_REQUIRE(pool_coff >= 0 && pool_coff + d->cout <= pool_ctot, "ct_conv2d_wino_pool_fwd: pooled output slice");
        CT_REQUIRE((pool_oh == d->oh / 2 || pool_oh == (d->oh + 1) / 2) && (pool_ow == d->ow / 2 || pool_ow == (d->ow + 1) / 2),
                   "ct_conv2d_wino_pool_fwd: pooled size %dx%d for a %dx%d map", pool_oh, pool_ow, d->oh, d->ow);
    }
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_wino_fwd: input slice");
    if (d->nseg == 0)
        CT_REQUIRE(d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "ct_conv2d_wino_fwd: output slice");
    else {
        CT_REQUIRE(!pool_out && write_full, "ct_conv2d_wino_fwd: pooling with segmented output");
        for (int g = 0; g < d->nseg; ++g) CT_REQUIRE(d->seg[g].ptr, "ct_conv2d_wino_fwd: null segment");
    }
    CT_REQUIRE(!d->res || (d->res_coff >= 0 && d->res_coff + d->cout <= d->res_ctot), "ct_conv2d_wino_fwd: residual slice");
    const long long img_in_bytes = (long long)d->in_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_in_bytes < kMaxBufBytes, "ct_conv2d_wino_fwd: one image exceeds 2 GiB");
    const long long img_out_bytes = d->nseg ? 4 : (long long)d->out_ctot * d->oh * d->ow * 4;
    const long long img_res_bytes = d->res ? (long long)d->res_ctot * d->oh * d->ow * 4 : 0;
    CT_REQUIRE(img_out_bytes < kMaxBufBytes && img_res_bytes < kMaxBufBytes, "ct_conv2d_wino_fwd: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_in_bytes, std::max(img_out_bytes, img_res_bytes)));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)wino_f2x2_3x3_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           WINO_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    const int OHW = d->oh * d->ow;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        WinoArgs a{};
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
        hipLaunchKernelGGL(wino_f2x2_3x3_f32, dim3(8 * groups * a.kblocks), dim3(512), WINO_LDS_BYTES, st, a);
        CT_LAUNCH_CHECK("wino_f2x2_3x3_f32");
    }
    return CT_OK;
}
