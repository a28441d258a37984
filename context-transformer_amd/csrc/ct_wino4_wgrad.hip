// libctdet: Winograd F(3x3, 4x4) weight gradient on the fp32 MFMA path -- the large-tile counterpart of
// ct_wino_wgrad.hip for the 3x3 / stride 1 / dilation 1 / pad 1 convolutions (what `losses.backward()`, train.py:228,
// makes autograd compute for the weights of models/RFB_Net_vgg.py:219-227 and the 3x3 BasicConv layers).  From the
// forward  Y = A^T [ (G g G^T) .* (B^T d B) ] A  (4x4 output tile, 6x6 input patch d; ct_wino4.hip) follows
//
//   dg[k][c] = G^T [ sum_tiles (A e A^T) .* (B^T d B) ] G      e = 4x4 tile of dZ[k], d = 6x6 patch of X[c]
//
// i.e. 36 independent [cout] x [cin] x [tiles] GEMMs: 4x fewer multiplications than the direct weight-gradient GEMM,
// 1.78x fewer than F(3x3, 2x2).
//
// Kernel: workgroup (256 threads, 4 waves) = 32 output channels x 32 input channels x all 36 transform points, walking
// its share of the tiles in chunks of 8; TWO workgroups per CU (72 KB of LDS and 256 registers per lane each):
//   * transform phase: thread (j = tid & 7, ch = tid >> 3) loads the 6x6 input patch of (tile j, input channel ch) and
//     the 4x4 dZ tile of (tile j, output channel ch) -- the loads were issued during the previous MFMA phase --,
//     applies B^T d B and A e A^T in registers and writes the 2 x 36 transform-domain values to LDS as
//     [xi][channel 32][tile parity 2][tile pair 4]: conflict-free scalar writes, and one ds_read_b128 hands an MFMA
//     lane its operand for all four k-steps of the chunk;
//   * MFMA phase: wave w owns transform points 9w .. 9w+8: per point a 32 x 32 x 8 GEMM = 4 v_mfma_f32_32x32x2_f32
//     fed by two ds_read_b128; 144 accumulator registers; the 16 global loads of the next chunk ride behind the MFMAs.
//   The phases of ONE workgroup do not overlap (single LDS buffer, two barriers per chunk); the CU overlaps the MFMA
//   phase of one workgroup with the transform phase of the other (measured: 512 workgroups 762 us, 256 workgroups
//   1027 us for 512->512 @38x38 bs 32; delaying the second workgroup of a CU by a transform phase changes nothing).
//   The 32 x 32 block writes 72 KB of operands to LDS per 144 MFMAs -- at 64 B/clk that path alone is half the MFMA
//   time, which is what keeps the matrix pipe at 0.45 here.  A wave-specialised form (one workgroup of 8 waves per CU:
//   four that only run MFMAs, four that only load and transform, double-buffered LDS, one barrier per chunk) was 45 %
//   SLOWER (1109 us): the transform waves run at half speed beside an MFMA wave on their SIMD.
// Tile ranges are split over blockIdx.y; partial sums meet in the workspace dU[36][cout][cin] through f32 atomics, and
// wino4_wgrad_finish applies G^T . G per (k, c) into the dense dw[cout][cin][3][3].
#include "ct_common.h"
#include "ct_wino4_points.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;
constexpr int TT = 8;                        // tiles per chunk
constexpr int CB = 32;                       // channels per block side
constexpr int XS = CB * 8;                   // floats per transform point and operand: [channel 32][parity 2][pair 4]
constexpr int OPF = 36 * XS;                 // one operand (V or E) of one chunk: 9216 floats = 36 KB
constexpr int W4W_LDS_BYTES = 2 * OPF * 4;   // V + E, single buffer: 72 KB

struct W4WArgs {
    const float* x;
    const float* dz;
    float* dU;               // [36][Cout][Cin]
    unsigned x_bytes, dz_bytes;
    int Cin, Cout, H, W, x_ctot, x_coff, dz_ctot, dz_coff;
    int TY, TX, NT, chunks, chunks_per_split, cblocks;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

using ctdet::w4::bt6;      // x -> B^T x, e -> A e, u -> G^T u for the points 0, +-3/4, +-3/2, inf (ct_wino4_points.h)
using ctdet::w4::a6;

__global__ __launch_bounds__(256, 2) void wino4_wgrad_f32(const W4WArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x / a.cblocks, cb = blockIdx.x - kb * a.cblocks;
    const int k0 = kb * CB, c0 = cb * CB;
    const int ch_first = blockIdx.y * a.chunks_per_split;
    const int ch_end = min(a.chunks, ch_first + a.chunks_per_split);
    const int nch = ch_end - ch_first;
    const int HW = a.H * a.W, TYX = a.TY * a.TX;

    // ---- loader role: tile j of the chunk, channel chn of the block (clamped: results past Cin / Cout are dropped)
    const int j = tid & 7, chn = tid >> 3;
    const int xcoff = (a.x_coff + min(c0 + chn, a.Cin - 1)) * HW;
    const int zcoff = (a.dz_coff + min(k0 + chn, a.Cout - 1)) * HW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rz = make_rsrc(a.dz, a.dz_bytes);
    float* const wpos = lds + chn * 8 + (j & 1) * 4 + (j >> 1);

    int voff[6], zoff[4];
    bool lp, m2, m3, m4, m5, z1, z2, z3;
    // addresses of chunk q (absolute index); past the end every offset is out of range -> zeros
    auto addr = [&](int q) {
        const int T = q * TT + j;
        const bool live = q < ch_end && T < a.NT;
        const int n = (unsigned)T / (unsigned)TYX;
        const int rem = T - n * TYX;
        const int ty = (unsigned)rem / (unsigned)a.TX, tx = rem - ty * a.TX;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        lp = tx == 0;                            // left padding column: load from x = 0, shift the unpack by one
        m2 = x0 + 2 < a.W; m3 = x0 + 3 < a.W; m4 = x0 + 4 < a.W; m5 = x0 + 5 < a.W;
        z1 = x0 + 2 < a.W; z2 = x0 + 3 < a.W; z3 = x0 + 4 < a.W;          // dZ columns 4tx + 1 .. 4tx + 3
        const int base = n * a.x_ctot * HW + xcoff + y0 * a.W + x0 + (lp ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.H;
            voff[i] = ok ? (base + i * a.W) * 4 : kInvalidOff;
        }
        const int zb = n * a.dz_ctot * HW + zcoff + (y0 + 1) * a.W + x0 + 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) zoff[i] = (live && y0 + 1 + i < a.H) ? (zb + i * a.W) * 4 : kInvalidOff;
    };
    i32x4 rv4[6], rz4[4];
    i32x2 rv2[6];

    // V = B^T d B of the loaded patch -> LDS
    auto transform_v = [&]() {
        float t[6][6];
        // an edge tile anywhere in the wave takes the masked unpack; interior chunks (most of a large map) skip it
        const bool edge = lp || !m5;
        if (__builtin_amdgcn_ballot_w64(edge) != 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 q4 = __builtin_bit_cast(f32x4, rv4[i]);
                const f32x2 q2 = __builtin_bit_cast(f32x2, rv2[i]);
                t[i][0] = lp ? 0.f : q4.x;
                t[i][1] = lp ? q4.x : q4.y;
                t[i][2] = m2 ? (lp ? q4.y : q4.z) : 0.f;
                t[i][3] = m3 ? (lp ? q4.z : q4.w) : 0.f;
                t[i][4] = m4 ? (lp ? q4.w : q2.x) : 0.f;
                t[i][5] = m5 ? (lp ? q2.x : q2.y) : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 q4 = __builtin_bit_cast(f32x4, rv4[i]);
                const f32x2 q2 = __builtin_bit_cast(f32x2, rv2[i]);
                t[i][0] = q4.x; t[i][1] = q4.y; t[i][2] = q4.z; t[i][3] = q4.w; t[i][4] = q2.x; t[i][5] = q2.y;
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {            // column pass: t[.][c] = B^T t[.][c]
            float d[6], o[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = t[i][c];
            bt6(d, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][c] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {            // row pass + store
            float o[6];
            bt6(t[i], o);
#pragma unroll
            for (int c = 0; c < 6; ++c) wpos[(i * 6 + c) * XS] = o[c];
        }
    };
    // E = A e A^T of the loaded dZ tile -> LDS
    auto transform_e = [&]() {
        float r[6][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {            // column pass: r[.][c] = A e[.][c]
            float e[4], o[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 q4 = __builtin_bit_cast(f32x4, rz4[i]);
                e[i] = c == 0 ? q4.x : c == 1 ? (z1 ? q4.y : 0.f) : c == 2 ? (z2 ? q4.z : 0.f) : (z3 ? q4.w : 0.f);
            }
            a6(e, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i][c] = o[i];
        }
        float* const epos = wpos + OPF;
#pragma unroll
        for (int i = 0; i < 6; ++i) {            // row pass + store
            float o[6];
            a6(r[i], o);
#pragma unroll
            for (int c = 0; c < 6; ++c) epos[(i * 6 + c) * XS] = o[c];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int x = 0; x < 9; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    addr(ch_first);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        rv4[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[i], 0, 0);
        rv2[i] = __builtin_amdgcn_raw_buffer_load_b64(rx, voff[i], 16, 0);      // soffset 16: columns 4, 5
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rz4[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, zoff[i], 0, 0);

    const float* const Vr = lds + (9 * wave) * XS + l31 * 8 + hi * 4;
    for (int c = 0; c < nch; ++c) {
        transform_v();
        transform_e();
        addr(ch_first + c + 1);
        __syncthreads();
        // ---- MFMA phase: D[cout 32][cin 32] += E[cout][tile pair] * V[cin][tile pair], 4 pairs per point
        f32x4 fa[2], fb[2];
        fb[0] = *reinterpret_cast<const f32x4*>(Vr);
        fa[0] = *reinterpret_cast<const f32x4*>(Vr + OPF);
#pragma unroll
        for (int x = 0; x < 9; ++x) {
            const int cur = x & 1;
            if (x + 1 < 9) {
                fb[cur ^ 1] = *reinterpret_cast<const f32x4*>(Vr + (x + 1) * XS);
                fa[cur ^ 1] = *reinterpret_cast<const f32x4*>(Vr + OPF + (x + 1) * XS);
            }
            __builtin_amdgcn_s_setprio(1);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc[x], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            // the next chunk's 16 loads, two per point behind the MFMAs
            if (x < 6) {
                rv4[x] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[x], 0, 0);
                rv2[x] = __builtin_amdgcn_raw_buffer_load_b64(rx, voff[x], 16, 0);
            } else if (x < 8) {
                rz4[2 * (x - 6)] = __builtin_amdgcn_raw_buffer_load_b128(rz, zoff[2 * (x - 6)], 0, 0);
                rz4[2 * (x - 6) + 1] = __builtin_amdgcn_raw_buffer_load_b128(rz, zoff[2 * (x - 6) + 1], 0, 0);
            }
            __builtin_amdgcn_s_setprio(1);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc[x], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    // ---- partial sums -> dU[xi][k][c]
#pragma unroll
    for (int x = 0; x < 9; ++x) {
        float* U = a.dU + (size_t)(9 * wave + x) * a.Cout * a.Cin;
        const int cc = c0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (k < a.Cout && cc < a.Cin) unsafeAtomicAdd(U + (size_t)k * a.Cin + cc, acc[x][r]);
        }
    }
}

// dw[k][c][3][3] = G^T dU[.][k][c] G,  G of ct_wino4_points.h
__global__ __launch_bounds__(256) void wino4_wgrad_finish(const float* __restrict__ dU, float* __restrict__ dw, int KC)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KC) return;
    auto gt = [](const float (&u)[6], float (&o)[3]) { ctdet::w4::gt3(u, o); };
    float w[3][6];                                       // G^T dU (rows), per transform column
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        float u[6], o[3];
#pragma unroll
        for (int m = 0; m < 6; ++m) u[m] = dU[(size_t)(m * 6 + n) * KC + i];
        gt(u, o);
#pragma unroll
        for (int m = 0; m < 3; ++m) w[m][n] = o[m];
    }
    float* out = dw + (size_t)i * 9;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float o[3];
        gt(w[m], o);
        out[m * 3 + 0] = o[0];
        out[m * 3 + 1] = o[1];
        out[m * 3 + 2] = o[2];
    }
}

bool w4w_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->oh == d->h && d->ow == d->w && !d->transposed && d->cin >= 1 && d->cout >= 1 &&
           (long long)d->in_ctot * d->h * d->w * 4 < kMaxBufBytes;
}

}  // namespace

extern "C" int ct_conv_wgrad_wino4_supported(const ct_conv_desc* d) { return d && w4w_ok(d) ? 1 : 0; }

extern "C" size_t ct_conv_wgrad_wino4_workspace_bytes(const ct_conv_desc* d)
{
    return d ? (size_t)36 * d->cout * d->cin * sizeof(float) : 0;
}

extern "C" int ct_conv2d_wgrad_wino4(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                                     void* workspace, ct_stream_t stream)
{
    CT_REQUIRE(d && dz && dw && workspace && d->in, "ct_conv2d_wgrad_wino4: null pointer");
    if (!w4w_ok(d))
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_wgrad_wino4: needs 3x3 stride 1 dilation 1 pad 1 "
                           "(got %dx%d s%d d%d p%d)", d->kh, d->kw, d->stride, d->dil, d->pad_h);
    CT_REQUIRE(d->batch > 0, "ct_conv2d_wgrad_wino4: bad shape");
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_wgrad_wino4: input slice");
    CT_REQUIRE(dz_coff >= 0 && dz_coff + d->cout <= dz_ctot, "ct_conv2d_wgrad_wino4: dz slice");
    const long long img_x = (long long)d->in_ctot * d->h * d->w * 4, img_z = (long long)dz_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_z < kMaxBufBytes, "ct_conv2d_wgrad_wino4: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_x, img_z));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)wino4_wgrad_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           W4W_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    float* dU = static_cast<float*>(workspace);
    const int KC = d->cout * d->cin;
    if (!ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(dU, 0, (size_t)36 * KC * 4, st));
    const int kblocks = (d->cout + CB - 1) / CB, cblocks = (d->cin + CB - 1) / CB;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        W4WArgs a{};
        a.x = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.dz = dz + (size_t)b0 * dz_ctot * d->h * d->w;
        a.dU = dU;
        a.x_bytes = (unsigned)(img_x * nb);
        a.dz_bytes = (unsigned)(img_z * nb);
        a.Cin = d->cin; a.Cout = d->cout; a.H = d->h; a.W = d->w;
        a.x_ctot = d->in_ctot; a.x_coff = d->in_coff; a.dz_ctot = dz_ctot; a.dz_coff = dz_coff;
        a.TY = (d->h + 3) / 4; a.TX = (d->w + 3) / 4;
        a.NT = nb * a.TY * a.TX;
        a.chunks = (a.NT + TT - 1) / TT;
        a.cblocks = cblocks;
        const int blocks = kblocks * cblocks;
        // two workgroups per CU, one round
        static const int wgs = getenv("CTDET_W4W_WGS") ? atoi(getenv("CTDET_W4W_WGS")) : 512;
        int splits = std::max(1, std::min(a.chunks, wgs / blocks));
        splits = std::min(splits, 65535);
        a.chunks_per_split = (a.chunks + splits - 1) / splits;
        splits = (a.chunks + a.chunks_per_split - 1) / a.chunks_per_split;
        hipLaunchKernelGGL(wino4_wgrad_f32, dim3(blocks, splits), dim3(256), W4W_LDS_BYTES, st, a);
        CT_LAUNCH_CHECK("wino4_wgrad_f32");
    }
    hipLaunchKernelGGL(wino4_wgrad_finish, dim3((KC + 255) / 256), dim3(256), 0, st, dU, dw, KC);
    CT_LAUNCH_CHECK("wino4_wgrad_finish");
    return CT_OK;
}
