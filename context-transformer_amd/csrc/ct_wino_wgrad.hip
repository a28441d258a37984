// libctdet: Winograd F(3x3, 2x2) weight gradient on the fp32 MFMA path for the 3x3 / stride 1 / dilation 1 /
// pad 1 convolutions of the RFBNet-VGG stack -- what `losses.backward()` (train.py:228) makes autograd compute
// for the weights of models/RFB_Net_vgg.py:219-227 (VGG trunk) and the 3x3 BasicConv layers.  From
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          (forward, per 2x2 output tile / 4x4 input patch d)
//
// the filter gradient is
//
//   dg[k][c] = G^T [ sum_tiles (A e A^T) .* (B^T d B) ] G      e = 2x2 tile of dZ[k], d = 4x4 patch of X[c]
//
// i.e. 16 independent [cout] x [cin] x [tiles] GEMMs instead of 9 [cout] x [cin] x [4 x tiles] ones: 2.25x fewer
// multiplications than the direct weight-gradient GEMM (conv_wgrad_f32 in ct_train.hip).
//
// Kernel: workgroup (512 threads, 8 waves) = 64 output channels x 64 input channels x all 16 transform points,
// walking its share of the tiles in chunks of 8:
//   * thread (j = tid & 7, ch = tid >> 3) loads the 4x4 input patch of (tile j, input channel ch) with four
//     16-byte buffer loads and the 2x2 dZ tile of (tile j, output channel ch) with two 8-byte loads, applies
//     B^T d B and A e A^T in registers and writes the 2 x 16 transform-domain values to LDS as
//     [xi][channel 64][tile parity 2][tile pair 4] -- conflict-free scalar writes, and one ds_read_b128 hands an
//     MFMA lane its operand for all four k-steps of the chunk;
//   * wave w owns xi = 2w, 2w+1: per chunk and xi a 64 x 64 x 8 GEMM = 16 v_mfma_f32_32x32x2_f32 fed by four
//     ds_read_b128; 128 accumulator registers per lane;
//   * LDS double-buffered (2 x 64 KB), ONE barrier per chunk; transforms, LDS writes and the loads of chunk c+2
//     are issued in slices behind the MFMAs of chunk c.
// Tile ranges are split over blockIdx.y; partial sums meet in the workspace dU[16][cout][cin] through f32 atomics,
// and wino_wgrad_finish applies G^T . G per (k, c) into the dense dw[cout][cin][3][3].
#include "ct_common.h"
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
constexpr int XS = 64 * 8;                   // floats per transform point: [channel 64][parity 2][pair 4]
constexpr int OPF = 16 * XS;                 // one operand (V or E) of one chunk: 8192 floats = 32 KB
constexpr int WW_LDS_BYTES = 4 * OPF * 4;    // 2 buffers x (V, E) = 128 KB

struct WWArgs {
    const float* x;
    const float* dz;
    float* dU;               // [16][Cout][Cin]
    unsigned x_bytes, dz_bytes;
    int Cin, Cout, H, W, x_ctot, x_coff, dz_ctot, dz_coff;
    int TY, TX, NT, chunks, chunks_per_split, cblocks;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(512) void wino_wgrad_f32(const WWArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x / a.cblocks, cb = blockIdx.x - kb * a.cblocks;
    const int k0 = kb * 64, c0 = cb * 64;
    const int ch_first = blockIdx.y * a.chunks_per_split;
    const int ch_end = min(a.chunks, ch_first + a.chunks_per_split);
    const int nch = ch_end - ch_first;
    const int HW = a.H * a.W, TYX = a.TY * a.TX;

    // ---- loader role: tile j of the chunk, channel ch of the block (clamped: results past Cin / Cout are dropped)
    const int j = tid & 7, chn = tid >> 3;
    const int xcoff = (a.x_coff + min(c0 + chn, a.Cin - 1)) * HW;
    const int zcoff = (a.dz_coff + min(k0 + chn, a.Cout - 1)) * HW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rz = make_rsrc(a.dz, a.dz_bytes);
    const int lds_pos = chn * 8 + (j & 1) * 4 + (j >> 1);

    int voff[4], zoff[2];
    bool lp, m2, m3, zm;
    // addresses of chunk q (absolute index); past the end every offset is out of range -> zeros
    auto addr = [&](int q) {
        const int T = q * TT + j;
        const bool live = q < ch_end && T < a.NT;
        const int n = (unsigned)T / (unsigned)TYX;
        const int rem = T - n * TYX;
        const int ty = (unsigned)rem / (unsigned)a.TX, tx = rem - ty * a.TX;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        lp = tx == 0;
        m2 = x0 + 2 < a.W;
        m3 = x0 + 3 < a.W;
        zm = x0 + 2 < a.W;
        const int base = n * a.x_ctot * HW + xcoff + y0 * a.W + x0 + (lp ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.H;
            voff[i] = ok ? (base + i * a.W) * 4 : kInvalidOff;
        }
        const int zb = n * a.dz_ctot * HW + zcoff + (y0 + 1) * a.W + x0 + 1;
        zoff[0] = live ? zb * 4 : kInvalidOff;
        zoff[1] = (live && y0 + 2 < a.H) ? (zb + a.W) * 4 : kInvalidOff;
    };
    i32x4 rw[4];
    i32x2 rq[2];
    auto load_all = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[i], 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) rq[i] = __builtin_amdgcn_raw_buffer_load_b64(rz, zoff[i], 0, 0);
    };
    auto unpack_rows = [&](int i0, float (&d)[16]) {
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            // whole-vector reinterpretation first (component-wise bit_cast of buffer-load results is miscompiled)
            const f32x4 q = __builtin_bit_cast(f32x4, rw[i]);
            const float vx = q.x, vy = q.y, vz = q.z, vw = q.w;
            d[i * 4 + 0] = lp ? 0.f : vx;
            d[i * 4 + 1] = lp ? vx : vy;
            d[i * 4 + 2] = m2 ? (lp ? vy : vz) : 0.f;
            d[i * 4 + 3] = m3 ? (lp ? vz : vw) : 0.f;
        }
    };
    auto bt_cols = [&](int j0, const float (&d)[16], float (&t)[16]) {       // t = B^T d, columns j0, j0+1
#pragma unroll
        for (int jj = j0; jj < j0 + 2; ++jj) {
            t[0 * 4 + jj] = d[0 * 4 + jj] - d[2 * 4 + jj];
            t[1 * 4 + jj] = d[1 * 4 + jj] + d[2 * 4 + jj];
            t[2 * 4 + jj] = d[2 * 4 + jj] - d[1 * 4 + jj];
            t[3 * 4 + jj] = d[1 * 4 + jj] - d[3 * 4 + jj];
        }
    };
    auto b_rows = [&](int i0, const float (&t)[16], float (&v)[16]) {        // v = t B, rows i0, i0+1
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
            v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
            v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
            v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };
    // r = A e : rows (e0, e0 + e1, e0 - e1, -e1), two columns each
    auto ae = [&](float (&r)[8]) {
        const f32x2 q0 = __builtin_bit_cast(f32x2, rq[0]);
        const f32x2 q1 = __builtin_bit_cast(f32x2, rq[1]);
        const float e00 = q0.x, e01 = zm ? q0.y : 0.f, e10 = q1.x, e11 = zm ? q1.y : 0.f;
        r[0] = e00;       r[1] = e01;
        r[2] = e00 + e10; r[3] = e01 + e11;
        r[4] = e00 - e10; r[5] = e01 - e11;
        r[6] = -e10;      r[7] = -e11;
    };
    auto aet = [&](const float (&r)[8], float (&e)[16]) {                    // e = r A^T
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i * 4 + 0] = r[2 * i];
            e[i * 4 + 1] = r[2 * i] + r[2 * i + 1];
            e[i * 4 + 2] = r[2 * i] - r[2 * i + 1];
            e[i * 4 + 3] = -r[2 * i + 1];
        }
    };

    f32x16 acc[2][2][2];       // [xi][cout block][cin block]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][i][jj][r] = 0.f;

    struct Frag { f32x4 a[2], b[2]; };
    auto read_frag = [&](int buf, int x, Frag& f) {
        const float* Vb = lds + buf * 2 * OPF + (2 * wave + x) * XS + l31 * 8 + hi * 4;
        const float* Eb = Vb + OPF;
        f.a[0] = *reinterpret_cast<const f32x4*>(Eb);
        f.a[1] = *reinterpret_cast<const f32x4*>(Eb + 256);
        f.b[0] = *reinterpret_cast<const f32x4*>(Vb);
        f.b[1] = *reinterpret_cast<const f32x4*>(Vb + 256);
    };

    // ---- prologue: chunk 0 -> buffer 0, raw data of chunk 1 -> registers
    {
        float d[16], t[16], v[16], r[8], e[16];
        addr(ch_first);
        load_all();
        unpack_rows(0, d); unpack_rows(2, d);
        bt_cols(0, d, t); bt_cols(2, d, t);
        b_rows(0, t, v); b_rows(2, t, v);
        ae(r); aet(r, e);
        float* vp = lds + lds_pos;
#pragma unroll
        for (int q = 0; q < 16; ++q) { vp[q * XS] = v[q]; vp[OPF + q * XS] = e[q]; }
        addr(ch_first + 1);
        load_all();
    }
    __syncthreads();

#define WW_MFMA(XI, II, JJ, F, SS) \
    acc[XI][II][JJ] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[II][SS], F.b[JJ][SS], acc[XI][II][JJ], 0, 0, 0)
#define WW_PIN() __builtin_amdgcn_sched_barrier(0)
#define WW_VST(q) vp[(q) * XS] = v[q]
#define WW_EST(q) vp[OPF + (q) * XS] = e[q]
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        Frag f0, f1;
        float d[16], t[16], v[16], r[8], e[16];
        float* vp = lds + (buf ^ 1) * 2 * OPF + lds_pos;
        read_frag(buf, 0, f0);
        WW_MFMA(0, 0, 0, f0, 0); unpack_rows(0, d); WW_PIN();
        WW_MFMA(0, 0, 1, f0, 0); unpack_rows(2, d); WW_PIN();
        WW_MFMA(0, 1, 0, f0, 0); bt_cols(0, d, t); WW_PIN();
        WW_MFMA(0, 1, 1, f0, 0); bt_cols(2, d, t); WW_PIN();
        WW_MFMA(0, 0, 0, f0, 1); b_rows(0, t, v); WW_PIN();
        WW_MFMA(0, 0, 1, f0, 1); b_rows(2, t, v); WW_PIN();
        WW_MFMA(0, 1, 0, f0, 1); WW_VST(0); WW_VST(1); WW_VST(2); WW_VST(3); WW_PIN();
        WW_MFMA(0, 1, 1, f0, 1); WW_VST(4); WW_VST(5); WW_VST(6); WW_VST(7); WW_PIN();
        read_frag(buf, 1, f1);
        WW_MFMA(0, 0, 0, f0, 2); WW_VST(8); WW_VST(9); WW_VST(10); WW_VST(11); WW_PIN();
        WW_MFMA(0, 0, 1, f0, 2); WW_VST(12); WW_VST(13); WW_VST(14); WW_VST(15); WW_PIN();
        WW_MFMA(0, 1, 0, f0, 2); ae(r); WW_PIN();
        WW_MFMA(0, 1, 1, f0, 2); aet(r, e); WW_PIN();
        WW_MFMA(0, 0, 0, f0, 3); WW_EST(0); WW_EST(1); WW_EST(2); WW_EST(3); WW_PIN();
        WW_MFMA(0, 0, 1, f0, 3); WW_EST(4); WW_EST(5); WW_EST(6); WW_EST(7); WW_PIN();
        WW_MFMA(0, 1, 0, f0, 3); WW_EST(8); WW_EST(9); WW_EST(10); WW_EST(11); WW_PIN();
        WW_MFMA(0, 1, 1, f0, 3); WW_EST(12); WW_EST(13); WW_EST(14); WW_EST(15); WW_PIN();
        WW_MFMA(1, 0, 0, f1, 0); addr(ch_first + c + 2); WW_PIN();
        WW_MFMA(1, 0, 1, f1, 0); WW_PIN();
        WW_MFMA(1, 1, 0, f1, 0); rw[0] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[0], 0, 0); WW_PIN();
        WW_MFMA(1, 1, 1, f1, 0); rw[1] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[1], 0, 0); WW_PIN();
        WW_MFMA(1, 0, 0, f1, 1); rw[2] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[2], 0, 0); WW_PIN();
        WW_MFMA(1, 0, 1, f1, 1); rw[3] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff[3], 0, 0); WW_PIN();
        WW_MFMA(1, 1, 0, f1, 1); rq[0] = __builtin_amdgcn_raw_buffer_load_b64(rz, zoff[0], 0, 0); WW_PIN();
        WW_MFMA(1, 1, 1, f1, 1); rq[1] = __builtin_amdgcn_raw_buffer_load_b64(rz, zoff[1], 0, 0); WW_PIN();
        WW_MFMA(1, 0, 0, f1, 2); WW_MFMA(1, 0, 1, f1, 2); WW_MFMA(1, 1, 0, f1, 2); WW_MFMA(1, 1, 1, f1, 2);
        WW_MFMA(1, 0, 0, f1, 3); WW_MFMA(1, 0, 1, f1, 3); WW_MFMA(1, 1, 0, f1, 3); WW_MFMA(1, 1, 1, f1, 3);
        __syncthreads();
    }
#undef WW_MFMA
#undef WW_PIN
#undef WW_VST
#undef WW_EST

    // ---- partial sums -> dU[xi][k][c]
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float* U = a.dU + (size_t)(2 * wave + x) * a.Cout * a.Cin;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int cc = c0 + 32 * jj + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + 32 * i + 8 * (r >> 2) + 4 * hi + (r & 3);
                    if (k < a.Cout && cc < a.Cin) unsafeAtomicAdd(U + (size_t)k * a.Cin + cc, acc[x][i][jj][r]);
                }
            }
    }
}

// dw[k][c][3][3] = G^T dU[.][k][c] G,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_wgrad_finish(const float* __restrict__ dU, float* __restrict__ dw,
                                                         int KC)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KC) return;
    float u[16], w[12];
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q] = dU[(size_t)q * KC + i];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const float s = 0.5f * (u[4 + n] + u[8 + n]), df = 0.5f * (u[4 + n] - u[8 + n]);
        w[0 * 4 + n] = u[n] + s;
        w[1 * 4 + n] = df;
        w[2 * 4 + n] = s + u[12 + n];
    }
    float* o = dw + (size_t)i * 9;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const float s = 0.5f * (w[m * 4 + 1] + w[m * 4 + 2]), df = 0.5f * (w[m * 4 + 1] - w[m * 4 + 2]);
        o[m * 3 + 0] = w[m * 4 + 0] + s;
        o[m * 3 + 1] = df;
        o[m * 3 + 2] = s + w[m * 4 + 3];
    }
}

bool ww_ok(const ct_conv_desc* d)
{
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->oh == d->h && d->ow == d->w && !d->transposed && d->cin >= 1 && d->cout >= 1 &&
           (long long)d->in_ctot * d->h * d->w * 4 < kMaxBufBytes;
}

} // namespace

extern "C" int ct_conv_wgrad_wino_supported(const ct_conv_desc* d) { return d && ww_ok(d) ? 1 : 0; }

extern "C" size_t ct_conv_wgrad_wino_workspace_bytes(const ct_conv_desc* d)
{
    return d ? (size_t)16 * d->cout * d->cin * sizeof(float) : 0;
}

extern "C" int ct_conv2d_wgrad_wino(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                                    void* workspace, ct_stream_t stream)
{
    CT_REQUIRE(d && dz && dw && workspace && d->in, "ct_conv2d_wgrad_wino: null pointer");
    if (!ww_ok(d))
        return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_wgrad_wino: needs 3x3 stride 1 dilation 1 pad 1 "
                           "(got %dx%d s%d d%d p%d)", d->kh, d->kw, d->stride, d->dil, d->pad_h);
    CT_REQUIRE(d->batch > 0, "ct_conv2d_wgrad_wino: bad shape");
    CT_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "ct_conv2d_wgrad_wino: input slice");
    CT_REQUIRE(dz_coff >= 0 && dz_coff + d->cout <= dz_ctot, "ct_conv2d_wgrad_wino: dz slice");
    const long long img_x = (long long)d->in_ctot * d->h * d->w * 4, img_z = (long long)dz_ctot * d->h * d->w * 4;
    CT_REQUIRE(img_z < kMaxBufBytes, "ct_conv2d_wgrad_wino: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, kMaxBufBytes / std::max(img_x, img_z));
    hipStream_t st = ctdet::as_stream(stream);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)wino_wgrad_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           WW_LDS_BYTES);
        });
        CT_HIP(attr_err);
    }
    float* dU = static_cast<float*>(workspace);
    const int KC = d->cout * d->cin;
    if (!ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(dU, 0, (size_t)16 * KC * 4, st));
    const int kblocks = (d->cout + 63) / 64, cblocks = (d->cin + 63) / 64;
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        const int nb = std::min(max_chunk, d->batch - b0);
        WWArgs a{};
        a.x = d->in + (size_t)b0 * d->in_ctot * d->h * d->w;
        a.dz = dz + (size_t)b0 * dz_ctot * d->h * d->w;
        a.dU = dU;
        a.x_bytes = (unsigned)(img_x * nb);
        a.dz_bytes = (unsigned)(img_z * nb);
        a.Cin = d->cin; a.Cout = d->cout; a.H = d->h; a.W = d->w;
        a.x_ctot = d->in_ctot; a.x_coff = d->in_coff; a.dz_ctot = dz_ctot; a.dz_coff = dz_coff;
        a.TY = (d->h + 1) / 2; a.TX = (d->w + 1) / 2;
        a.NT = nb * a.TY * a.TX;
        a.chunks = (a.NT + TT - 1) / TT;
        a.cblocks = cblocks;
        const int blocks = kblocks * cblocks;
        // one workgroup per CU (128 KB of LDS each), ONE round: every extra split pays the 64K-atomic epilogue again
        // (target 768 measured 7-25 % slower than 256 on the RFBNet shapes)
        static const int wgs = getenv("CTDET_WW_WGS") ? atoi(getenv("CTDET_WW_WGS")) : 256;
        int splits = std::max(1, std::min(a.chunks, wgs / blocks));
        splits = std::min(splits, 65535);
        a.chunks_per_split = (a.chunks + splits - 1) / splits;
        splits = (a.chunks + a.chunks_per_split - 1) / a.chunks_per_split;
        hipLaunchKernelGGL(wino_wgrad_f32, dim3(blocks, splits), dim3(512), WW_LDS_BYTES, st, a);
        CT_LAUNCH_CHECK("wino_wgrad_f32");
    }
    hipLaunchKernelGGL(wino_wgrad_finish, dim3((KC + 255) / 256), dim3(256), 0, st, dU, dw, KC);
    CT_LAUNCH_CHECK("wino_wgrad_finish");
    return CT_OK;
}
