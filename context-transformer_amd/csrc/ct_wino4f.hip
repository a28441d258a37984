// libctdet: Winograd F(4x4,3x3), FUSED, with the transform-domain products on the bf16 matrix pipe ("bf16x3") -- the narrow
// 3x3 / stride 1 / dilation 1 / pad 1 layers on the large maps of the RFBNet-VGG stack (models/RFB_Net_vgg.py:323-336:
// conv1_2, conv2_1, conv2_2, conv3_1; at 512 x 512 also conv3_2 / conv3_3; :219-227 the trunk they belong to).  Same
// ct_conv_desc contract, transforms (ct_wino4_points.h) and fused epilogue as ct_conv2d_wino4_fwd.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        per 4x4 output tile, 6x6 input patch d
//
// Where this kernel sits.  ct_wino4.hip (fused, fp32 MFMA) multiplies on the slowest matrix rate of the chip; ct_wino4s.hip
// (three kernels, bf16x3) needs V and M in HBM -- 13.5 + 9 bytes per output pixel and channel -- which does not pay below
// 256 channels on 150 x 150 / 300 x 300 maps (64 -> 64 @300x300 bs 32: V alone 2.5 GB).  Here everything stays on chip and the
// products run as the six bf16 piece products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; 0.375 of the fp32-MFMA time):
//   workgroup (512 threads, 8 waves) = 32 output tiles x ONE block of 64 output channels, 16-channel chunks
//     * phase T (all waves): every thread owns one (tile, channel) patch of the chunk.  The 6x6 patch arrives in LDS by DMA
//       (per row a 16-byte and two 4-byte buffer_load ... lds, issued a whole phase M earlier: no VGPR holds it while the MFMAs run);
//       B^T d B in registers, the 36 transform-domain values go lane-linearly to V[point 36][channel 16][tile 32] (fp32, 72 KB);
//     * phase M: wave w multiplies the points 4w .. 4w+3 for both cout halves and point 32 + w/2 for cout half w & 1
//       (nine 32x32 accumulator blocks = 144 registers, 54 v_mfma_f32_32x32x16_bf16 per chunk).  It is the only reader of
//       its full points, so the bf16x3 split happens after the LDS read (8 ds_read_b32 + 44 VALU per point); its A fragments
//       (U, pre-transformed and pre-split by ct_conv_pack_weights_wino4f) come straight from L2 in MFMA register order,
//       3 KB "units" = (point, cout half), two units ahead;
//     * V is single-buffered: barrier after T, barrier after M.  On a SIMD the vector ALU and the matrix pipe do not overlap
//       anyway ("law 1"), so separating the phases costs the barriers only and leaves the registers to
//       one phase at a time (144 accumulators + 36 patch / transform values OR 36 U + 24 V fragment registers);
//   after the channel loop the accumulators pass through LDS in two passes of 32 couts (144 KB), each thread applies
//   A^T M A for two (cout, tile) pairs per pass and the shared epilogue (ct_wino4_emit.h: scale / shift, residual, ReLU /
//   floor, the four 2x2 pooling windows of a tile, NCHW or head scatter).
// One accumulator per output, piece products smallest first: with cin <= 256 the large sum sees cin / 16 * 6 roundings
// (the fp32 MFMA: cin), measured in tests/test_gpu_wino.py::test_wino_rounding_error_vs_fp64.
#include "ct_common.h"
#include "ct_wino_pack.h"
#include "ct_wino4_points.h"
#include "ct_wino4_emit.h"
#include "ct_f16x2.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <mutex>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int CC = ctdet::kWinoX3CC;        // 16 channels per chunk = one MFMA k-group
constexpr int TB = 32;                      // tiles per workgroup
constexpr int KB = ctdet::kWinoKB;          // 64 output channels per workgroup
constexpr int NXI = 36;                     // transform points
constexpr int PT_STRIDE = CC * TB;          // 512 floats per point: V[channel 16][tile 32]
constexpr int V_BYTES = NXI * PT_STRIDE * 4;               // 72 KB
// One patch row (6 floats) per lane = one 16-byte DMA (columns 0..3) + two 4-byte ones (columns 4, 5).  A 12-byte DMA
// would need two per row, but the hardware places a lane's 12 bytes at lane x 16 (tools/ubench/lds_dma12.hip,
// profiles/r05_lds_dma12.txt): 2 KB per row instead of 1.5 KB, and V + stage would not fit the 160 KB.
constexpr int ROW_Q_BYTES = 64 * 16;        // [lane 64][16 B]
constexpr int ROW_E_BYTES = 64 * 4;         // [lane 64][4 B]
constexpr int ROW_BYTES = ROW_Q_BYTES + 2 * ROW_E_BYTES;   // 1536
constexpr int STAGE_WAVE_BYTES = 6 * ROW_BYTES;            // every lane's 6x6 patch: 9 KB per wave
constexpr int STAGE_BYTES = 8 * STAGE_WAVE_BYTES;          // 72 KB
constexpr int MXI = 32 * 32;                // output staging M[point][cout 32][tile 32]
constexpr int W4F_LDS_BYTES = NXI * MXI * 4;               // 144 KB = V + patch stage
static_assert(V_BYTES + STAGE_BYTES == W4F_LDS_BYTES, "LDS plan");
// U "units" = (transform point, cout half) of a wave: bf16x3 [piece 3][lane 64][16 B], f16x2 (H2, ct_f16x2.h) [piece 2][lane 64][16 B]
template <bool H2> struct ULayout {
    static constexpr int NP = H2 ? 2 : 3;                                   // pieces per value
    static constexpr int UNIT = H2 ? ctdet::kWino4fhUnitBytes : ctdet::kWino4fUnitBytes;
    static constexpr int WAVE = H2 ? ctdet::kWino4fhWaveBytes : ctdet::kWino4fWaveBytes;
    static constexpr int CHUNK = H2 ? ctdet::kWino4fhChunkBytes : ctdet::kWino4fChunkBytes;
};

struct Wino4fArgs {
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
    // f16x2 form (H2): V is split as V 2^eV with eV from the producer's maximum of |input| (ct_conv_desc.in_absmax), U arrives as
    // U 2^eU (eU in the trailer of the packed weights); the epilogue's per-channel scale takes 2^-(eU + eV)
    const unsigned* in_amax;
    const int* eU;
    unsigned* out_amax;      // any form: ct_conv_desc.out_absmax, max |y| of what the launch stores, or null
};

// The epilogue's view of the arguments (ct_wino4_emit.h is a template over any record with these members).  The persistent
// kernel re-reads them from the kernel-argument segment per item as plain words, which makes every pointer a GENERIC one
// -- flat loads and stores, and a flat store counts on vmcnt AND lgkmcnt, so the compiler drains all stores in front of
// the next LDS access (four vmcnt(0) per output quarter).  Global-address-space pointer types put them back on the global path.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) float gf32;
#else
typedef float gf32;
#endif
struct EpiArgs {
    const gf32* scale;
    const gf32* shift;
    const gf32* res;
    const gf32* lo;
    gf32* pool_out;
    int H, W, out_ctot, out_coff, res_ctot, res_coff;
    float res_scale;
    int relu, pool_ctot, pool_coff, pool_oh, pool_ow, write_full, nseg;
    ct_out_segment seg[3];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

using ctdet::w4::bt6;
using ctdet::w4::at4;

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

// Measurement build only (CTDET_EXTRA_FLAGS=-DCTDET_W4F_TRACE, tools/w4f_trace.py): wave 0 of every workgroup stamps the
// 100 MHz real-time counter (s_memrealtime: the shader clock counters of different XCDs are unrelated) at its phase boundaries into ct_wino4f_trace_buffer()[workgroup][8].
#ifdef CTDET_W4F_TRACE
__device__ unsigned long long* g_w4f_trace = nullptr;
#define W4F_STAMP(k)                                                                                   \
    do {                                                                                               \
        if (g_w4f_trace && tid == 0) g_w4f_trace[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define W4F_STAMP(k) do { } while (0)
#endif

// SEG: the launch scatters into the flattened head buffers (a.nseg > 0).  The plain instantiation drops that path and the
// 24 scalar registers of its three segment records, which a persistent loop would otherwise carry through its main loop.
// PLAIN: no residual, no per-channel floor, no head scatter (a.res == a.lo == nullptr, a.nseg == 0): what the narrow trunk layers
// this kernel exists for use (bias + ReLU, optionally the fused 2x2 max-pool) -- the epilogue then needs a third of the scalar
// registers and none of those branches.
template <bool SEG, bool PLAIN, bool H2>
__global__ __launch_bounds__(512) void wino_f4x4_3x3_x3(const Wino4fArgs a_in)
{
    const Wino4fArgs& a = a_in;
    typedef ULayout<H2> UL;
    constexpr int NP = UL::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* const lds = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = a.H * a.W;
    // Persistent grid: workgroup b walks the work items b, b + gridDim.x, ... (gridDim.x is a multiple of 8: an item keeps its XCD)
    // Item order = the plain grid's of ct_wino4.hip: item -> (XCD-local sequence, cout block fastest), so the cout blocks of a
    // tile block run at the same time on one XCD and share its input patches through that XCD's L2.  (Measured against the
    // opposite order -- cout block slowest, every workgroup of a round on the same 1.7 MB slice of U: 128 -> 128 @150x150 659 vs
    // 610-640 us; the input then comes from HBM once per cout block, 368 MB each at that shape.)
    const int nitems = 8 * ((a.tile_blocks + 7) / 8) * a.kblocks;
    auto item_tblk = [&](int vb) { return ((vb >> 3) / a.kblocks) * 8 + (vb & 7); };
    auto item_kb = [&](int vb) { return (vb >> 3) % a.kblocks; };
    auto next_valid = [&](int vb) {           // first item >= vb of this workgroup's sequence that has a tile block
        while (vb < nitems && item_tblk(vb) >= a.tile_blocks) vb += gridDim.x;
        return vb;
    };

    // ---- patch role: tile = l31, channel in chunk = 2 wave + h.  Rows outside the map use the out-of-range offset (the DMA
    // writes zeros for such lanes); a buffer load whose first byte lies before the row is dropped whole, so the left-edge
    // tiles load their 16-byte piece from x = 0 and shift (their column 0 is padding anyway: ct_wino4s.hip)
    int voffq[6];
    bool mc[6], lp;
    int d4;                                   // byte distance from the 16-byte piece to column 4
    int x0s;                                  // first patch column of this lane's tile (-1 for the left-edge tiles)
    // the addresses (integer divisions: computed where few registers are live) ...
    auto setup_addr = [&](int vb) {
        const int T = item_tblk(vb) * TB + l31;
        const bool live = T < a.NT;
        const int n = T / (a.TY * a.TX);
        const int rem = T - n * (a.TY * a.TX);
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        const int y0 = 4 * ty - 1;
        x0s = 4 * tx - 1;
        const bool left = tx == 0;
        d4 = left ? 12 : 16;
        const long base = (((long)n * a.in_ctot + a.in_coff + h) * a.H + y0) * (long)a.W + x0s + (left ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.H;
            voffq[i] = ok ? (int)((base + (long)i * a.W) * 4) : kInvalidOff;
        }
    };
    // ... and the column masks of the tile whose rows are being unpacked (they replace the previous item's only after its last
    // unpack)
    auto setup_masks = [&]() {
#pragma unroll
        for (int c = 0; c < 6; ++c) mc[c] = (unsigned)(x0s + c) < (unsigned)a.W;
        lp = x0s < 0;
    };
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(a.U, a.u_bytes);
    const int chunk_bytes = CC * HW * 4;
    const int chan_base = 2 * wave * HW * 4;
    const int last = a.chunks - 1;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    unsigned char* const stage = lds_raw + V_BYTES + wave * STAGE_WAVE_BYTES;     // this wave's patch rows: [row 6][1536 B]

    auto dma_patch = [&](int c) {
        const int soff = c * chunk_bytes + chan_base;
        (void)soff;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int v4 = voffq[i] + d4;          // an invalid row stays out of range: 0x7FFFFFF0 + 16 as an unsigned offset
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(stage + i * ROW_BYTES), 16, voffq[i], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(stage + i * ROW_BYTES + ROW_Q_BYTES), 4, v4, soff, 0, 0);
            // column 5 through the scalar offset: the instruction's immediate offset would move the LDS destination as well
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(stage + i * ROW_BYTES + ROW_Q_BYTES + ROW_E_BYTES), 4, v4, soff + 4, 0, 0);
        }
#endif
    };

    const unsigned char* const sq = stage + lane * 16;         // this lane's 16-byte piece of row 0
    const unsigned char* const se = stage + ROW_Q_BYTES + lane * 4;
    float* const vw = lds + wave * 64 + lane;                  // V[point][channel 2 wave + h][tile l31] = lane-linear
    const float* const vr = lds + (8 * h) * TB + l31;          // B fragment: channels 8h .. 8h+7 of tile l31

    // ---- A fragments: unit u of this wave = 3 KB [piece 3][lane 64][16 B]
    const int u_voff = wave * UL::WAVE + lane * 16;
    // f16x2: V is split as V 2^eV with eV PER IMAGE from the maximum the input's producer left for that image (ct_f16x2.h: an image's
    // results never depend on its batch mates); this lane's tile of the current item decides (transform and split roles: tile l31)
    float vscale = 1.f;
    float amax_run = 0.f;
    auto read_raw = [&](int xi, float (&raw)[8]) {
        const float* p = vr + xi * PT_STRIDE;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = p[e * TB];
    };
    auto split_all = [&](const float (&raw)[8], i32x4 (&fb)[NP]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (H2) {
                int hi, lo;
                ctdet::h2::split2(raw[2 * q] * vscale, raw[2 * q + 1] * vscale, hi, lo);
                fb[0][q] = hi;
                fb[1][q] = lo;
            } else {
                unsigned h0, m0, l0, h1, m1, l1;
                split3(raw[2 * q], h0, m0, l0);
                split3(raw[2 * q + 1], h1, m1, l1);
                fb[0][q] = pack_hi(h0, h1);
                fb[1][q] = pack_hi(m0, m1);
                fb[2][q] = pack_hi(l0, l1);
            }
        }
    };
    const int xi0 = 4 * wave, xi_half = 32 + (wave >> 1);

    // the piece products, smallest first   [A piece, B piece]: bf16x3 (mid, mid), (lo, hi), (hi, lo), (mid, hi), (hi, mid), (hi, hi);
    // f16x2 (lo, hi), (hi, lo), (hi, hi)
    constexpr int NPROD = H2 ? 3 : 6;
    constexpr int PA[6] = {H2 ? 1 : 1, H2 ? 0 : 2, 0, 1, 0, 0}, PB[6] = {H2 ? 0 : 1, H2 ? 1 : 0, H2 ? 0 : 2, 0, 1, 0};
#define W4F_UNIT(X, UA, FB)                                                                                          \
    do {                                                                                                             \
        _Pragma("unroll") for (int t_ = 0; t_ < NPROD; ++t_) {                                                       \
            if constexpr (H2)                                                                                        \
                acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, UA[PA[t_]]),               \
                                                                __builtin_bit_cast(f16x8, FB[PB[t_]]), acc[X], 0, 0, 0); \
            else                                                                                                     \
                acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, UA[PA[t_]]),             \
                                                                 __builtin_bit_cast(bf16x8, FB[PB[t_]]), acc[X], 0, 0, 0); \
        }                                                                                                            \
    } while (0)

    int vb = next_valid(blockIdx.x);
    if (vb >= nitems) return;
    setup_addr(vb);
    setup_masks();
    dma_patch(0);
    i32x4 ua0[NP], ua1[NP], ua2[NP];
    // f16x2: the maximum of |input| of the image this lane's tile (tile l31 of the item) belongs to.  For a follow-up item the load
    // leaves with the first U units, in front of the previous item's last stores (a wait for it must not wait for those stores)
    unsigned amax_bits = 0;
    bool amax_ahead = false;
    auto load_amax = [&](int tb_first) {
        const int Tl = tb_first + l31;
        const int nl = Tl < a.NT ? Tl / (a.TY * a.TX) : 0;
        return a.in_amax[(size_t)nl * ctdet::h2::kLineWords];
    };
    bool u_ahead = false;                     // ua0 / ua1 already hold (are receiving) units 0, 1 of this item's first chunk
    while (vb < nitems) {
        W4F_STAMP(0);
        const int kb = item_kb(vb);
        const int tb0 = item_tblk(vb) * TB;
        const int vb_next = next_valid(vb + gridDim.x);
        if constexpr (H2) {               // needed by the first phase M
            if (!amax_ahead) amax_bits = load_amax(tb0);
            vscale = __builtin_ldexpf(1.f, ctdet::h2::exponent_for(amax_bits, ctdet::h2::kGrowthBtB));
        }
        const int u_kb = kb * a.chunks;
        auto load_u_of = [&](int ukb, int c, int unit, i32x4 (&dst)[NP]) {
            const int soff = (ukb + c) * UL::CHUNK + unit * UL::UNIT;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) dst[pc] = __builtin_amdgcn_raw_buffer_load_b128(rU, u_voff, soff + pc * 1024, 0);
        };
        auto load_u = [&](int c, int unit, i32x4 (&dst)[NP]) { load_u_of(u_kb, c, unit, dst); };
        f32x16 acc[9];
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int c = 0; c < a.chunks; ++c) {
            // ================= phase T: patch(c), staged by DMA during the previous phase M (chunk 0: during the previous
            // item's output passes), -> V.  Chunk 0 of a follow-up item waits for nothing here: its rows were waited for before the
            // previous item's output passes and its first U units are under way since then (below) -- a vmcnt(0) at this
            // point would wait for the previous item's output STORES to drain (loads and stores share the counter).
            if (c > 0 || !u_ahead) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                load_u(c, 0, ua0);                   // the first two units of phase M: a whole phase T of latency
                load_u(c, 1, ua1);
            }
            const bool prefetch_next = c == last && vb_next < nitems;
            if (prefetch_next) {              // this item's addresses are dead: its last rows are in the stage
                // opaque copy: left to itself the compiler computes the next item's addresses at the TOP of this item (they only
                // depend on vb_next) and carries eight more registers through the whole channel loop
                int vbn = vb_next;
                asm volatile("" : "+s"(vbn));
                setup_addr(vbn);
            }
            {
                float t[6][6];
                float dd[6][6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(sq + i * ROW_BYTES);
                    float e4 = *reinterpret_cast<const float*>(se + i * ROW_BYTES);
                    float e5 = *reinterpret_cast<const float*>(se + i * ROW_BYTES + ROW_E_BYTES);
                    float qx = q.x, qy = q.y, qz = q.z, qw = q.w;
                    // unconditional reads, then selects: not exec-masked read / wait blocks
                    asm("" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qw), "+v"(e4), "+v"(e5));
                    dd[i][0] = (mc[0] && !lp) ? qx : 0.f;
                    dd[i][1] = mc[1] ? (lp ? qx : qy) : 0.f;
                    dd[i][2] = mc[2] ? (lp ? qy : qz) : 0.f;
                    dd[i][3] = mc[3] ? (lp ? qz : qw) : 0.f;
                    dd[i][4] = mc[4] ? e4 : 0.f;
                    dd[i][5] = mc[5] ? e5 : 0.f;
                }
                // the stage is free again once its reads have returned: the next rows -- chunk c + 1, or chunk 0 of this
                // workgroup's NEXT item -- are under way during the transform, the whole phase M and the output passes
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (c < last) {
                    dma_patch(c + 1);
                } else if (prefetch_next) {
                    setup_masks();
                    dma_patch(0);
                }
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    const float d[6] = {dd[0][cc], dd[1][cc], dd[2][cc], dd[3][cc], dd[4][cc], dd[5][cc]};
                    float o[6];
                    bt6(d, o);
#pragma unroll
                    for (int i = 0; i < 6; ++i) t[i][cc] = o[i];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    float v[6];
                    bt6(t[i], v);
                    float* vp = vw + (i * 6) * PT_STRIDE;
#pragma unroll
                    for (int j = 0; j < 6; ++j) vp[j * PT_STRIDE] = v[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c == 0) W4F_STAMP(1);
            // ================= phase M: nine units (point, cout half); U two units ahead, the next point's fragments split
            // behind the current point's MFMAs.  (The scheduler sinks some of the U loads to 3-5 MFMAs in front of their use to
            // shorten their live ranges; pinning them where they are written -- sched_barrier after every load -- needs 72 bytes of
            // scratch per lane at 256 registers and is slower.)
            {
                float raw[8];
                i32x4 fbA[NP], fbB[NP];
                read_raw(xi0, raw);
                split_all(raw, fbA);
                // point 0
                load_u(c, 2, ua2);
                read_raw(xi0 + 1, raw);
                W4F_UNIT(0, ua0, fbA);
                load_u(c, 3, ua0);
                split_all(raw, fbB);
                W4F_UNIT(1, ua1, fbA);
                // point 1
                load_u(c, 4, ua1);
                read_raw(xi0 + 2, raw);
                W4F_UNIT(2, ua2, fbB);
                load_u(c, 5, ua2);
                split_all(raw, fbA);
                W4F_UNIT(3, ua0, fbB);
                // point 2
                load_u(c, 6, ua0);
                read_raw(xi0 + 3, raw);
                W4F_UNIT(4, ua1, fbA);
                load_u(c, 7, ua1);
                split_all(raw, fbB);
                W4F_UNIT(5, ua2, fbA);
                // point 3
                load_u(c, 8, ua2);
                read_raw(xi_half, raw);
                W4F_UNIT(6, ua0, fbB);
                split_all(raw, fbA);
                W4F_UNIT(7, ua1, fbB);
                // the shared point 32 + wave / 2, cout half wave & 1
                W4F_UNIT(8, ua2, fbA);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        W4F_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next item's first rows (issued a phase M ago) are in the stage

        // ---- output transform: four passes of 16 couts through the V region  M[point][cout 16][tile 32] (72 KB; the stage
        // region is receiving the next item's first rows).  Quarter q = cout half q >> 1, accumulator registers 8 (q & 1) ..
        // + 7 of its blocks: register r of a block = cout (r & 3) + 8 (r >> 2) + 4 h of the half, tile l31.  Barriers are bare
        // s_barrier instructions: __syncthreads() would wait for the DMA in flight.
        // The epilogue's ~30 argument words are re-read from the kernel-argument segment per item, through a pointer the
        // compiler cannot see through: read once at kernel entry they would stay live across the main loop of every item of
        // this persistent workgroup and push its masks and descriptors out of the scalar registers.
        Wino4fArgs e;
#if defined(__HIP_DEVICE_COMPILE__)
        {
            typedef const __attribute__((address_space(4))) int* kernarg_words;
            kernarg_words kp = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            static_assert(sizeof(Wino4fArgs) % 4 == 0, "argument record in words");
            // the record is the kernel's FIRST AND ONLY explicit argument, at offset 0 of the kernel-argument segment (by-value
            // aggregates are placed at their natural alignment): a second parameter in front of it would silently shift these words
            static_assert(alignof(Wino4fArgs) <= 8, "argument record at offset 0 of the kernarg segment");
            int words[sizeof(Wino4fArgs) / 4];
#pragma unroll
            for (int i = 0; i < (int)(sizeof(Wino4fArgs) / 4); ++i) words[i] = kp[i];
            __builtin_memcpy(&e, words, sizeof(e));
        }
#else
        e = a_in;
#endif
        EpiArgs ep;
        ep.scale = (const gf32*)e.scale; ep.shift = (const gf32*)e.shift;
        ep.res = PLAIN ? nullptr : (const gf32*)e.res;
        ep.lo = PLAIN ? nullptr : (const gf32*)e.lo;
        ep.pool_out = (gf32*)e.pool_out;
        ep.H = e.H; ep.W = e.W; ep.out_ctot = e.out_ctot; ep.out_coff = e.out_coff;
        ep.res_ctot = e.res_ctot; ep.res_coff = e.res_coff; ep.res_scale = e.res_scale;
        ep.relu = e.relu; ep.pool_ctot = e.pool_ctot; ep.pool_coff = e.pool_coff; ep.pool_oh = e.pool_oh; ep.pool_ow = e.pool_ow;
        ep.write_full = e.write_full;
        ep.nseg = SEG ? e.nseg : 0;
#pragma unroll
        for (int g = 0; g < 3; ++g) ep.seg[g] = e.seg[g];
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(e.out, e.out_bytes);
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(e.res, e.res ? e.res_bytes : 0u);
        const int o_kk = tid >> 5, o_tl = tid & 31;
        const int o_T = tb0 + o_tl;
        const int o_n = o_T / (a.TY * a.TX);
        float ymul = 1.f;                 // f16x2: 2^-(eU + eV[image of this thread's tile]) into the per-channel scale
        if constexpr (H2) ymul = __builtin_ldexpf(1.f, -(*e.eU + ctdet::h2::image_exponent(e.in_amax, o_T < a.NT ? o_n : 0, ctdet::h2::kGrowthBtB)));
        const int o_rem = o_T - o_n * (a.TY * a.TX);
        const int o_ty = o_rem / a.TX, o_tx = o_rem - o_ty * a.TX;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    lds[(xi0 + p) * (16 * 32) + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[2 * p + (q >> 1)][8 * (q & 1) + r];
            if ((wave & 1) == (q >> 1)) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    lds[xi_half * (16 * 32) + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[8][8 * (q & 1) + r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!SEG && q == 3) {          // (the head-scatter instantiation has no registers to spare for it)
                // every accumulator is in LDS: the first U units of the next item leave BEFORE this quarter's stores, so that
                // waiting for them later does not wait for those stores
                u_ahead = vb_next < nitems;
                if (u_ahead) {
                    const int ukb_next = item_kb(vb_next) * a.chunks;
                    load_u_of(ukb_next, 0, 0, ua0);
                    load_u_of(ukb_next, 0, 1, ua1);
                    if constexpr (H2) amax_bits = load_amax(item_tblk(vb_next) * TB);
                }
                if constexpr (H2) amax_ahead = u_ahead;
            }
            {
                const int co = kb * KB + 16 * q + o_kk;
                if (o_T < a.NT && co < a.M) {
                    const int n = o_n, ty = o_ty, tx = o_tx;
                    float z[4][6];
                    const float* mp = lds + o_kk * 32 + o_tl;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        float m[6], y[4];
#pragma unroll
                        for (int i = 0; i < 6; ++i) m[i] = mp[(i * 6 + j) * (16 * 32)];
                        at4(m, y);
#pragma unroll
                        for (int i = 0; i < 4; ++i) z[i][j] = y[i];
                    }
                    float y[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) at4(z[i], y[i]);
                    ctdet::w4::emit_tile4(ep, rout, rres, n, ty, tx, co, y, ymul, H2 && e.out_amax != nullptr, amax_run);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (q == 1) W4F_STAMP(3);
        }
        W4F_STAMP(4);
        if constexpr (H2) {               // ct_conv_desc.out_absmax: this item's maxima, one atomic per image present in the wave
            if (e.out_amax) {
                ctdet::h2::flush_absmax(e.out_amax, o_T < a.NT ? o_n : -1, amax_run);
                amax_run = 0.f;
            }
        }
        vb = vb_next;
    }
#undef W4F_UNIT
}

bool wino4f_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->cin % CC == 0 && d->nseg >= 0 && d->nseg <= 3 && (d->nseg == 0 || !d->res) && !d->transposed &&
           d->oh == d->h && d->ow == d->w;
}

}  // namespace

extern "C" int ct_conv_wino4f_supported(const ct_conv_desc* d)
{
    // the kernel needs 144 KB of LDS per workgroup: not a geometry question, but a part without it cannot run any geometry
    static const bool lds_ok = [] {
        int dev = 0, bytes = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
            return true;                      // no device to ask (the CPU-side symbol checks): geometry decides
        return bytes >= W4F_LDS_BYTES;
    }();
    return d && lds_ok && wino4f_ok(d) ? 1 : 0;
}

#ifdef CTDET_W4F_TRACE
extern "C" int ct_wino4f_set_trace(unsigned long long* buf)      // device buffer of 8 words per workgroup, or null
{
    CT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_w4f_trace), &buf, sizeof(buf)));
    return CT_OK;
}
#endif

extern "C" size_t ct_conv_wino4f_packed_bytes(int cin, int cout)
{
    if (cin <= 0 || cout <= 0 || cin % CC) return 0;
    return (size_t)((cout + KB - 1) / KB) * (cin / CC) * ctdet::kWino4fChunkBytes;
}

extern "C" int ct_conv_pack_weights_wino4f(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                           ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 0, 46, (float*)upacked, stream, "ct_conv_pack_weights_wino4f");
}

extern "C" int ct_conv_pack_weights_wino4f_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                                 ct_stream_t stream)
{
    return ctdet::pack_wino_any(w, cout, nparts, cin, 1, 46, (float*)upacked, stream, "ct_conv_pack_weights_wino4f_dgrad");
}

extern "C" size_t ct_conv_wino4f_h2_packed_bytes(int cin, int cout)
{
    if (cin <= 0 || cout <= 0 || cin % CC) return 0;
    return ctdet::wino_h2_trailer_offset(cin, cout, 48) + ctdet::kWino4hTrailerBytes;
}

extern "C" int ct_conv_pack_weights_wino4f_h2(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                              ct_stream_t stream)
{
    return ctdet::pack_wino_h2(w, cout, nparts, cin, 0, 48, upacked, stream, "ct_conv_pack_weights_wino4f_h2");
}

extern "C" int ct_conv_pack_weights_wino4f_h2_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                                    ct_stream_t stream)
{
    return ctdet::pack_wino_h2(w, cout, nparts, cin, 1, 48, upacked, stream, "ct_conv_pack_weights_wino4f_h2_dgrad");
}

static int wino4f_launch(const ct_conv_desc* d, const void* upacked, int variant, float* pool_out, int pool_ctot,
                         int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);

extern "C" int ct_conv2d_wino4f_pool_fwd(const ct_conv_desc* d, const void* upacked, float* pool_out, int pool_ctot,
                                         int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream)
{
    return wino4f_launch(d, upacked, 1, pool_out, pool_ctot, pool_coff, pool_oh, pool_ow, write_full, stream);
}

extern "C" int ct_conv2d_wino4f_pool_fwd_v(const ct_conv_desc* d, const void* upacked, int variant, float* pool_out, int pool_ctot,
                                           int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream)
{
    return wino4f_launch(d, upacked, variant, pool_out, pool_ctot, pool_coff, pool_oh, pool_ow, write_full, stream);
}

static int wino4f_launch(const ct_conv_desc* d, const void* upacked, int variant, float* pool_out, int pool_ctot,
                         int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream)
{
    const char* who = "ct_conv2d_wino4f_fwd";
    CT_REQUIRE(variant == 1 || variant == 2, "%s: variant %d (1 = bf16x3, 2 = f16x2)", who, variant);
    const bool h2 = variant == 2;
    CT_REQUIRE(!h2 || (d && d->in_absmax), "%s: the f16x2 form needs the maximum of |input| (ct_conv_desc.in_absmax: the producer's "
               "out_absmax slot, or ct_absmax_f32)", who);
    CT_REQUIRE(d && upacked, "%s: null pointer", who);
    CT_REQUIRE(d->in && (d->out || d->nseg > 0) && d->scale && d->shift, "%s: null tensor", who);
    if (!wino4f_ok(d))
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
    const size_t u_bytes = h2 ? ctdet::wino_h2_trailer_offset(d->cin, d->cout, 48) : ct_conv_wino4f_packed_bytes(d->cin, d->cout);
    CT_REQUIRE(u_bytes < (size_t)kMaxBufBytes, "%s: packed weights exceed 2 GiB", who);
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_in_bytes, std::max(img_out_bytes, img_res_bytes)));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            const void* fs[] = {(const void*)wino_f4x4_3x3_x3<false, false, false>, (const void*)wino_f4x4_3x3_x3<false, true, false>,
                                (const void*)wino_f4x4_3x3_x3<true, false, false>, (const void*)wino_f4x4_3x3_x3<false, false, true>,
                                (const void*)wino_f4x4_3x3_x3<false, true, true>, (const void*)wino_f4x4_3x3_x3<true, false, true>};
            for (const void* f : fs)
                if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, W4F_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    const int OHW = d->oh * d->ow;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        Wino4fArgs a{};
        a.in = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.U = static_cast<const unsigned char*>(upacked);
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
        a.TY = (d->oh + 3) / 4; a.TX = (d->ow + 3) / 4;
        a.NT = nb * a.TY * a.TX;
        a.tile_blocks = (a.NT + TB - 1) / TB;
        a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
        a.res_ctot = d->res_ctot; a.res_coff = d->res_coff; a.res_scale = d->res_scale;
        a.relu = d->relu;
        a.pool_out = pool_out ? pool_out + (size_t)b0 * pool_ctot * pool_oh * pool_ow : nullptr;
        a.pool_ctot = pool_ctot; a.pool_coff = pool_coff; a.pool_oh = pool_oh; a.pool_ow = pool_ow;
        a.write_full = write_full;
        a.in_amax = d->in_absmax ? d->in_absmax + (size_t)b0 * ctdet::h2::kLineWords : nullptr;
        a.out_amax = d->out_absmax ? d->out_absmax + (size_t)b0 * ctdet::h2::kLineWords : nullptr;
        a.eU = h2 ? reinterpret_cast<const int*>(static_cast<const unsigned char*>(upacked) + u_bytes) + 1 : nullptr;
        a.kblocks = (d->cout + KB - 1) / KB;
        // 8 XCD-local sequences of (tile block group, cout block); sequences past the last tile block exit at once
        const int groups = (a.tile_blocks + 7) / 8;
        const int items = 8 * groups * a.kblocks;
        // persistent: one workgroup per CU (144 KB of LDS each) walks the items with a stride of the grid size
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                n = 256;
            return std::max(8, n / 8 * 8);
        }();
        CT_PROF(h2 ? "wino_f4x4_3x3_h2" : "wino_f4x4_3x3_x3", st);
        const dim3 grid(std::min(items, cus));
        if (h2) {
            if (a.nseg > 0) hipLaunchKernelGGL((wino_f4x4_3x3_x3<true, false, true>), grid, dim3(512), W4F_LDS_BYTES, st, a);
            else if (!a.res && !a.lo) hipLaunchKernelGGL((wino_f4x4_3x3_x3<false, true, true>), grid, dim3(512), W4F_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((wino_f4x4_3x3_x3<false, false, true>), grid, dim3(512), W4F_LDS_BYTES, st, a);
        } else {
            if (a.nseg > 0) hipLaunchKernelGGL((wino_f4x4_3x3_x3<true, false, false>), grid, dim3(512), W4F_LDS_BYTES, st, a);
            else if (!a.res && !a.lo) hipLaunchKernelGGL((wino_f4x4_3x3_x3<false, true, false>), grid, dim3(512), W4F_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((wino_f4x4_3x3_x3<false, false, false>), grid, dim3(512), W4F_LDS_BYTES, st, a);
        }
        CT_LAUNCH_CHECK("wino_f4x4_3x3_x3");
    }
    return CT_OK;
}

extern "C" int ct_conv2d_wino4f_fwd(const ct_conv_desc* d, const void* upacked, ct_stream_t stream)
{
    return ct_conv2d_wino4f_pool_fwd(d, upacked, nullptr, 0, 0, 0, 0, 1, stream);
}
