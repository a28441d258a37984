// libctdet: training-side kernels of the RFBNet stack (what autograd + cuDNN do for the reference
// in train.py:222-229): weight gradient of a convolution, batch-statistics BatchNorm forward /
// backward, bias+ReLU backward, max-pool backward and the head-gradient gather.  The data gradient
// of a convolution is the `transposed` mode of ct_conv2d_fwd (ct_conv.hip).
#include "ct_common.h"
#include "ct_f16x2.h"
#include <algorithm>
#include <mutex>
#include <unordered_set>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kInvalidOff = 0x7FFFFFF0;
constexpr long long kMaxBufBytes = 0x7FFFFF00LL;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------
// weight gradient:  dW[co][ci][kh][kw] = sum_{n,oh,ow} dZ[n][co][oh][ow] * X[n][ci][ih][iw]
// GEMM  M = cout, N = cin*kh*kw, K = batch*oh*ow (pixels), fp32 MFMA 32x32x2, 128x128 (or 64x64) tile,
// split over pixel ranges across workgroups (fp32 atomicAdd into a zeroed dW).
// Lanes run along pixels (coalesced rows of dZ and of the shifted X), tiles are staged to LDS
// transposed ([pixel][row], stride 65) so the MFMA fragments read conflict-free.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dz;
    float* dw;
    unsigned x_bytes, dz_bytes;
    int Cin, H, W, x_ctot, x_coff;
    int Cout, OW, OHW, dz_ctot, dz_coff;
    int stride, pad_h, pad_w, dil;
    int Npix, Ncols;
    int tiles_m, tiles_n, pix_per_split;
};

// TAPMAJOR (needs cin % (64*TB) == 0): the GEMM columns are ordered tap-major (col = tap*cin + ci), so all rows of
// a column tile share ONE filter tap: the shifted-pixel offset and its bounds test are computed once per chunk and
// lane instead of once per gathered row -- the generic path spends ~12 VALU instructions per MFMA on that and starves
// the matrix pipe (measured 78 TFLOP/s, half of peak, independent of tile shape).
template <int KH, int KW, int TB, bool TAPMAJOR>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradArgs a)
{
    // workgroup tile (64*TB couts) x (64*TB columns), each wave TB x TB accumulator blocks of 32x32
    constexpr int KHW = KH * KW;
    constexpr int BM = 64 * TB, BN = 64 * TB, BKP = 64, LD = BM + 1, ROWS = 16 * TB;
    extern __shared__ float wg_lds[];
    float* As = wg_lds;
    float* Bs = wg_lds + BKP * LD;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave & 1) * 32 * TB, wn0 = (wave >> 1) * 32 * TB;
    const int tile = blockIdx.x;
    const int m0 = (tile % a.tiles_m) * BM;
    const int c0 = (tile / a.tiles_m) * BN;
    const int p_begin = blockIdx.y * a.pix_per_split;
    const int p_end = min(p_begin + a.pix_per_split, a.Npix);
    if (p_begin >= p_end) return;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz, a.dz_bytes);
    const int HW = a.H * a.W;

    // wave-uniform row descriptors: A rows = cout m0 + wave + 4j, B rows = column c0 + wave + 4j
    int a_soff[ROWS], b_soff[ROWS], b_dh[ROWS], b_dw[ROWS];
    const int tm_tap = TAPMAJOR ? c0 / a.Cin : 0, tm_ci0 = TAPMAJOR ? c0 - tm_tap * a.Cin : 0;
    const int tm_dh = (tm_tap / KW) * a.dil - a.pad_h, tm_dw = (tm_tap % KW) * a.dil - a.pad_w;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int m = m0 + wave + 4 * j;
        a_soff[j] = min(m, a.Cout - 1) * a.OHW * 4;      // rows past Cout repeat the last one: their results are never stored
        if (TAPMAJOR) {
            b_soff[j] = (tm_ci0 + wave + 4 * j) * HW * 4;
            b_dh[j] = tm_dh;
            b_dw[j] = tm_dw;
        } else {
            const int col = c0 + wave + 4 * j;
            const int ci = col / KHW, tap = col - ci * KHW;
            const int kh = tap / KW, kw = tap - kh * KW;
            b_soff[j] = min(ci, a.Cin - 1) * HW * 4;    // columns past Ncols: clamped, never stored
            b_dh[j] = kh * a.dil - a.pad_h;
            b_dw[j] = kw * a.dil - a.pad_w;
        }
    }

    float areg[ROWS], breg[ROWS];
    auto load_chunk = [&](int p0) {
        const int P = p0 + lane;
        const bool pv = P < p_end;
        const int Pc = pv ? P : 0;
        const int n = Pc / a.OHW;
        const int s = Pc - n * a.OHW;
        const int oh = s / a.OW, ow = s - oh * a.OW;
        const int zoff = pv ? ((n * a.dz_ctot + a.dz_coff) * a.OHW + s) * 4 : kInvalidOff;
        const int xbase = (n * a.x_ctot + a.x_coff) * HW;
        const int ih0 = oh * a.stride, iw0 = ow * a.stride;
        if (TAPMAJOR) {
            const int ih = ih0 + tm_dh, iw = iw0 + tm_dw;
            const bool ok = pv && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const int xoff = ok ? (xbase + ih * a.W + iw) * 4 : kInvalidOff;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                areg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zoff, a_soff[j], 0));
                breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff, b_soff[j], 0));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            areg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zoff, a_soff[j], 0));
            const int ih = ih0 + b_dh[j], iw = iw0 + b_dw[j];
            const bool ok = pv && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const int xoff = ok ? (xbase + ih * a.W + iw) * 4 : kInvalidOff;
            breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff, b_soff[j], 0));
        }
    };

    f32x16 acc[TB][TB], acc2[TB][TB];
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

    if (TAPMAJOR && TB == 1) {
        // Pipelined variant (same structure as the forward implicit-GEMM kernel): two LDS buffers, ONE barrier
        // per chunk; in k-pair slot e the wave stores element e of chunk c+1 (loaded during chunk c-1) into
        // the other buffer and re-issues the load of element e for chunk c+2, then runs the slot's MFMA.
        float* As2 = wg_lds + 2 * BKP * LD;
        float* Bs2 = As2 + BKP * LD;
        int zoff, xoff;
        auto setup = [&](int p0) {
            const int P = p0 + lane;
            const bool pv = P < p_end;
            const int Pc = pv ? P : 0;
            const int n = Pc / a.OHW;
            const int sp = Pc - n * a.OHW;
            const int oh = sp / a.OW, ow = sp - oh * a.OW;
            zoff = pv ? ((n * a.dz_ctot + a.dz_coff) * a.OHW + sp) * 4 : kInvalidOff;
            const int ih = oh * a.stride + tm_dh, iw = ow * a.stride + tm_dw;
            const bool ok = pv && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            xoff = ok ? ((n * a.x_ctot + a.x_coff) * HW + ih * a.W + iw) * 4 : kInvalidOff;
        };
        auto load_e = [&](int e) {
            if (e < ROWS) areg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zoff, a_soff[e], 0));
            else breg[e - ROWS] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff, b_soff[e - ROWS], 0));
        };
        auto store_e = [&](int e, float* Ad, float* Bd) {
            if (e < ROWS) Ad[lane * LD + wave + 4 * e] = areg[e];
            else Bd[lane * LD + wave + 4 * (e - ROWS)] = breg[e - ROWS];
        };
        setup(p_begin);
#pragma unroll
        for (int e = 0; e < 2 * ROWS; ++e) load_e(e);
#pragma unroll
        for (int e = 0; e < 2 * ROWS; ++e) store_e(e, As, Bs);
        setup(p_begin + BKP);                    // past the end: every offset invalid -> zeros
#pragma unroll
        for (int e = 0; e < 2 * ROWS; ++e) load_e(e);
        __syncthreads();
        int cidx = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += BKP, ++cidx) {
            const float* Ac = (cidx & 1) ? As2 : As;
            const float* Bc = (cidx & 1) ? Bs2 : Bs;
            float* An = (cidx & 1) ? As : As2;
            float* Bn = (cidx & 1) ? Bs : Bs2;
            const float* Ab = Ac + hsel * LD + wm0 + l31;
            const float* Bb = Bc + hsel * LD + wn0 + l31;
            // registers hold chunk c+1; the loads issued below fetch chunk c+2
            setup(p0 + 2 * BKP);
            float af[2], bf[2];
            af[0] = Ab[0];
            bf[0] = Bb[0];
#pragma unroll
            for (int sidx = 0; sidx < BKP / 2; ++sidx) {
                const int cur = sidx & 1;
                if (sidx + 1 < BKP / 2) {
                    af[cur ^ 1] = Ab[(2 * sidx + 2) * LD];
                    bf[cur ^ 1] = Bb[(2 * sidx + 2) * LD];
                }
                store_e(sidx, An, Bn);
                load_e(sidx);
                if (sidx & 1) acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc2[0][0], 0, 0, 0);
                else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
    load_chunk(p_begin);
        for (int p0 = p_begin; p0 < p_end; p0 += BKP) {
    #pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                As[lane * LD + wave + 4 * j] = areg[j];
                Bs[lane * LD + wave + 4 * j] = breg[j];
            }
            __syncthreads();
            if (p0 + BKP < p_end) load_chunk(p0 + BKP);
            const float* Ab = As + hsel * LD + wm0 + l31;
            const float* Bb = Bs + hsel * LD + wn0 + l31;
    #pragma unroll 8
            for (int s = 0; s < BKP / 2; ++s) {
                float af[TB], bf[TB];
    #pragma unroll
                for (int i = 0; i < TB; ++i) {
                    af[i] = Ab[(2 * s) * LD + 32 * i];
                    bf[i] = Bb[(2 * s) * LD + 32 * i];
                }
    #pragma unroll
                for (int i = 0; i < TB; ++i)
    #pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        if (s & 1) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc2[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int j = 0; j < TB; ++j) {
        int col = c0 + wn0 + 32 * j + l31;
        if (col >= a.Ncols) continue;
        if (TAPMAJOR) col = (tm_ci0 + wn0 + 32 * j + l31) * KHW + tm_tap;      // back to dW's [ci][tap] order
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (m < a.Cout) unsafeAtomicAdd(a.dw + (size_t)m * a.Ncols + col, acc[i][j][r] + acc2[i][j][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm2d training statistics over (batch, h, w) of a channel slice of an NCHW buffer
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int batch, int ctot,
                                                       int coff, int HW, float* __restrict__ mean,
                                                       float* __restrict__ var, float momentum, float unbias,
                                                       float* __restrict__ rmean, float* __restrict__ rvar)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int n = 0; n < batch; ++n) {
        const float* p = z + ((size_t)n * ctot + coff + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) {
            const double v = p[i];
            s += v;
            ss += v * v;
        }
    }
    s = block_sum(s, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        const double cnt = (double)batch * HW;
        const double m = s / cnt;
        const float mf = (float)m, vf = (float)fmax(ss / cnt - m * m, 0.0);   // biased variance (normalisation)
        mean[c] = mf;
        var[c] = vf;
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (vf * unbias);
        }
    }
}

// sliced variants (grid = channels x slices): block sums in double meet in `scratch` ([2][C] doubles, zeroed by the
// host entry) through f64 atomics; a one-block-per-channel reduction leaves most of the chip idle on the RFB maps
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const float* __restrict__ z, int batch, int ctot,
                                                            int coff, int C, int HW, int per_slice,
                                                            double* __restrict__ scratch)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const long e0 = (long)blockIdx.y * per_slice, e1 = min(e0 + per_slice, (long)batch * HW);
    double s = 0.0, ss = 0.0;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const int n = (int)(e / HW), i = (int)(e - (long)n * HW);
        const double v = z[((size_t)n * ctot + coff + c) * HW + i];
        s += v;
        ss += v * v;
    }
    s = block_sum(s, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&scratch[c], s);
        unsafeAtomicAdd(&scratch[C + c], ss);
    }
}

__global__ void bn_stats_final_kernel(const double* __restrict__ scratch, int C, double cnt,
                                      float* __restrict__ mean, float* __restrict__ var, float momentum,
                                      float unbias, float* __restrict__ rmean, float* __restrict__ rvar)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = scratch[c] / cnt;
    const float mf = (float)m, vf = (float)fmax(scratch[C + c] / cnt - m * m, 0.0);
    mean[c] = mf;
    var[c] = vf;
    if (rmean) {        // running statistics of nn.BatchNorm2d (momentum, unbiased variance), same launch
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (vf * unbias);
    }
}

struct BnApplyArgs {
    const float* z;          // conv output, channel slice [z_coff, z_coff+C) of z_ctot
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    const float* lo;         // per-channel clamp (0 / -inf) or null
    const float* res;        // residual slice or null
    float* y;
    int batch, C, HW, y_ctot, y_coff, res_ctot, res_coff, z_ctot, z_coff;
    float eps, rscale;
    int relu;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a)
{
    const long total = (long)a.batch * a.C * a.HW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.HW);
        const unsigned t = idx / a.HW;
        const int c = (int)(t % a.C);
        const int n = (int)(t / a.C);
        const float inv = 1.f / sqrtf(a.var[c] + a.eps);
        float v = (a.z[((size_t)n * a.z_ctot + a.z_coff + c) * a.HW + i] - a.mean[c]) * inv * a.gamma[c] + a.beta[c];
        if (a.res) v = v * a.rscale + a.res[((size_t)n * a.res_ctot + a.res_coff + c) * a.HW + i];
        if (a.lo) v = fmaxf(v, a.lo[c]);
        else if (a.relu) v = fmaxf(v, 0.f);
        a.y[((size_t)n * a.y_ctot + a.y_coff + c) * a.HW + i] = v;
    }
}

struct BnBwdArgs {
    const float* dy;         // slice of the consumer-side gradient buffer
    const float* y;          // slice of the forward output (ReLU mask) or null
    const float* z;          // conv output slice (z_ctot / z_coff); dz uses the same slicing
    const float* mean;
    const float* var;
    const float* gamma;
    const float* lo;
    float* dz;               // dense
    float* dgamma;
    float* dbeta;
    float* dres;             // where the residual branch's gradient goes (slice) or null
    int batch, C, HW, dy_ctot, dy_coff, y_ctot, y_coff, dres_ctot, dres_coff, dres_accumulate, z_ctot, z_coff;
    float eps, rscale;
    int relu;
    int frozen;              // 1: mean/var are constants (nn.BatchNorm2d in eval mode): no statistics terms in dz
};

__device__ __forceinline__ float masked_dy(const BnBwdArgs& a, int n, int c, int i)
{
    float g = a.dy[((size_t)n * a.dy_ctot + a.dy_coff + c) * a.HW + i];
    const bool act = a.lo ? (a.lo[c] == 0.f) : (a.relu != 0);
    if (act && a.y[((size_t)n * a.y_ctot + a.y_coff + c) * a.HW + i] <= 0.f) g = 0.f;
    return g;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const float inv = 1.f / sqrtf(a.var[c] + a.eps), mu = a.mean[c];
    double sb = 0.0, sg = 0.0;
    for (int n = 0; n < a.batch; ++n) {
        const float* zp = a.z + ((size_t)n * a.z_ctot + a.z_coff + c) * a.HW;
        for (int i = threadIdx.x; i < a.HW; i += 256) {
            const float g = masked_dy(a, n, c, i) * a.rscale;
            sb += g;
            sg += (double)g * ((zp[i] - mu) * inv);
        }
    }
    sb = block_sum(sb, red);
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) {
        a.dbeta[c] = (float)sb;
        a.dgamma[c] = (float)sg;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_part_kernel(const BnBwdArgs a, int per_slice,
                                                                 double* __restrict__ scratch)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const float inv = 1.f / sqrtf(a.var[c] + a.eps), mu = a.mean[c];
    const long e0 = (long)blockIdx.y * per_slice, e1 = min(e0 + per_slice, (long)a.batch * a.HW);
    double sb = 0.0, sg = 0.0;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const int n = (int)(e / a.HW), i = (int)(e - (long)n * a.HW);
        const float g = masked_dy(a, n, c, i) * a.rscale;
        sb += g;
        sg += (double)g * ((a.z[((size_t)n * a.z_ctot + a.z_coff + c) * a.HW + i] - mu) * inv);
    }
    sb = block_sum(sb, red);
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&scratch[c], sb);
        unsafeAtomicAdd(&scratch[a.C + c], sg);
    }
}

__global__ void bn_bwd_reduce_final_kernel(const double* __restrict__ scratch, int C, float* __restrict__ dbeta,
                                           float* __restrict__ dgamma)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)scratch[c];
    dgamma[c] = (float)scratch[C + c];
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a)
{
    const long total = (long)a.batch * a.C * a.HW;
    const float cnt = (float)a.batch * a.HW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.HW);
        const unsigned t = idx / a.HW;
        const int c = (int)(t % a.C);
        const int n = (int)(t / a.C);
        const float inv = 1.f / sqrtf(a.var[c] + a.eps);
        const float gm = masked_dy(a, n, c, i);
        if (a.dres) {
            float* d = a.dres + ((size_t)n * a.dres_ctot + a.dres_coff + c) * a.HW + i;
            *d = a.dres_accumulate ? *d + gm : gm;
        }
        const float g = gm * a.rscale;
        const size_t zi = ((size_t)n * a.z_ctot + a.z_coff + c) * a.HW + i;
        const float xh = (a.z[zi] - a.mean[c]) * inv;
        a.dz[zi] = a.frozen ? a.gamma[c] * inv * g
                            : a.gamma[c] * inv * (g - a.dbeta[c] / cnt - xh * a.dgamma[c] / cnt);
    }
}

// y = act(conv + bias):  dz = dy * (y > 0 if relu),  dbias = sum dz   (VGG convs, heads)
// grid (channel, slice): a slice is a contiguous range of the channel's batch*HW elements, so a 64-channel
// 300x300 layer still fills the chip; per-slice sums meet in dbias through one float atomic per block.
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ dy, int dy_ctot,
                                                           int dy_coff, const float* __restrict__ y,
                                                           int y_ctot, int y_coff, int relu, int batch,
                                                           int C, int HW, float* __restrict__ dz,
                                                           int dz_ctot, int dz_coff,
                                                           float* __restrict__ dbias, int per_slice,
                                                           unsigned* __restrict__ amax)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const long e0 = (long)blockIdx.y * per_slice, e1 = min(e0 + per_slice, (long)batch * HW);
    double sb = 0.0;
    const bool vec = (HW & 3) == 0 && (per_slice & 3) == 0;
    int n = (int)(e0 / HW);
    int i0 = (int)(e0 - (long)n * HW);
    for (long e = e0; e < e1;) {
        const int len = (int)min((long)(HW - i0), e1 - e);
        const float* g = dy + ((size_t)n * dy_ctot + dy_coff + c) * HW + i0;
        const float* yy = y ? y + ((size_t)n * y_ctot + y_coff + c) * HW + i0 : nullptr;
        float* o = dz + ((size_t)n * dz_ctot + dz_coff + c) * HW + i0;
        float run = 0.f;                 // max |dz| of what this thread stores for image n (ct_f16x2.h: the f16x2 data gradients)
        if (vec) {
            for (int i = threadIdx.x * 4; i < len; i += 1024) {
                float4 v = *reinterpret_cast<const float4*>(g + i);
                if (relu) {
                    const float4 t = *reinterpret_cast<const float4*>(yy + i);
                    v.x = t.x <= 0.f ? 0.f : v.x;
                    v.y = t.y <= 0.f ? 0.f : v.y;
                    v.z = t.z <= 0.f ? 0.f : v.z;
                    v.w = t.w <= 0.f ? 0.f : v.w;
                }
                *reinterpret_cast<float4*>(o + i) = v;
                sb += (double)((v.x + v.y) + (v.z + v.w));
                ctdet::h2::track_absmax(run, v.x);
                ctdet::h2::track_absmax(run, v.y);
                ctdet::h2::track_absmax(run, v.z);
                ctdet::h2::track_absmax(run, v.w);
            }
        } else {
            for (int i = threadIdx.x; i < len; i += 256) {
                float v = g[i];
                if (relu && yy[i] <= 0.f) v = 0.f;
                o[i] = v;
                sb += v;
                ctdet::h2::track_absmax(run, v);
            }
        }
        if (amax) ctdet::h2::flush_absmax(amax, n, run);         // uniform over the block: every lane is in the same image
        e += len;
        i0 = 0;
        ++n;
    }
    sb = block_sum(sb, red);
    if (threadIdx.x == 0 && dbias) unsafeAtomicAdd(&dbias[c], (float)sb);
}

// max-pool backward (gather form): every input element collects dy of the windows whose FIRST
// maximum (row-major scan, torch's tie rule) it is.
__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dx, long planes, int H,
                                                            int W, int OH, int OW, int k, int stride,
                                                            int pad, int accumulate)
{
    const long total = planes * H * W;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const int w = (int)(idx % W);
        const unsigned t = idx / W;
        const int h = (int)(t % H);
        const long pl = t / H;
        const float* xp = x + pl * (long)H * W;
        const float* gp = dy + pl * (long)OH * OW;
        float g = 0.f;
        const int oh_lo = max(0, (h + pad - k + stride) / stride), oh_hi = min(OH - 1, (h + pad) / stride);
        const int ow_lo = max(0, (w + pad - k + stride) / stride), ow_hi = min(OW - 1, (w + pad) / stride);
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int h0 = oh * stride - pad, w0 = ow * stride - pad;
                float m = -INFINITY;
                int mh = -1, mw = -1;
                for (int hh = max(h0, 0); hh < min(h0 + k, H); ++hh)
                    for (int ww = max(w0, 0); ww < min(w0 + k, W); ++ww) {
                        const float v = xp[(long)hh * W + ww];
                        if (v > m) { m = v; mh = hh; mw = ww; }
                    }
                if (mh == h && mw == w) g += gp[(long)oh * OW + ow];
            }
        dx[idx] = accumulate ? dx[idx] + g : g;
    }
}

// The same for planes that fit in LDS (pool5: 3x3 / stride 1 on 19x19 or 32x32 -- 81 global reads per element in the kernel
// above, 492 us of the bs-32 training step): workgroup = one plane; the plane goes to LDS once, every window's first maximum
// is found once (its flat index), every element then sums dy over the <= k*k windows that name it.  Same tie rule, same sum
// order (windows in row-major order) as the kernel above.
constexpr int kPoolPlaneMax = 4096;
__global__ __launch_bounds__(256) void maxpool2d_bwd_plane_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, int H, int W, int OH, int OW,
                                                                  int k, int stride, int pad, int accumulate)
{
    __shared__ float xs[kPoolPlaneMax];
    __shared__ int am[kPoolPlaneMax];
    const long pl = blockIdx.x;
    const float* xp = x + pl * (long)H * W;
    const float* gp = dy + pl * (long)OH * OW;
    float* dp = dx + pl * (long)H * W;
    for (int i = threadIdx.x; i < H * W; i += 256) xs[i] = xp[i];
    __syncthreads();
    for (int o = threadIdx.x; o < OH * OW; o += 256) {
        const int oh = o / OW, ow = o - oh * OW;
        const int h0 = oh * stride - pad, w0 = ow * stride - pad;
        float m = -INFINITY;
        int mi = -1;
        for (int hh = max(h0, 0); hh < min(h0 + k, H); ++hh)
            for (int ww = max(w0, 0); ww < min(w0 + k, W); ++ww) {
                const float v = xs[hh * W + ww];
                if (v > m) { m = v; mi = hh * W + ww; }
            }
        am[o] = mi;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int h = i / W, w = i - h * W;
        float g = 0.f;
        const int oh_lo = max(0, (h + pad - k + stride) / stride), oh_hi = min(OH - 1, (h + pad) / stride);
        const int ow_lo = max(0, (w + pad - k + stride) / stride), ow_hi = min(OW - 1, (w + pad) / stride);
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow)
                if (am[oh * OW + ow] == i) g += gp[oh * OW + ow];
        dp[i] = accumulate ? dp[i] + g : g;
    }
}

// MaxPool2d(2, 2) (floor or ceil mode): the windows do not overlap, so one thread owns one window -- reads its
// (up to) 2x2 inputs, finds the first maximum in row-major order (strict >, as the gather kernel above) and
// writes all four gradients; input rows/columns no window covers (floor mode, odd extent) get zeros.
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ dy,
                                                             float* __restrict__ dx, int H, int W, int OH, int OW,
                                                             int accumulate)
{
    const int WH = (H + 1) >> 1, WW = (W + 1) >> 1;
    const float* xp = x + (size_t)blockIdx.x * H * W;
    float* dp = dx + (size_t)blockIdx.x * H * W;
    const float* gp = dy + (size_t)blockIdx.x * OH * OW;
    for (unsigned idx = threadIdx.x; idx < (unsigned)(WH * WW); idx += 256) {
        const int wh = idx / (unsigned)WW, ww = idx - wh * WW;
        const int h0 = 2 * wh, w0 = 2 * ww;
        const bool h1 = h0 + 1 < H, w1 = w0 + 1 < W;
        const float g = (wh < OH && ww < OW) ? gp[wh * OW + ww] : 0.f;
        const int o = h0 * W + w0;
        const float v00 = xp[o];
        const float v01 = w1 ? xp[o + 1] : -INFINITY;
        const float v10 = h1 ? xp[o + W] : -INFINITY;
        const float v11 = (h1 && w1) ? xp[o + W + 1] : -INFINITY;
        int sel = -1;
        float m = -INFINITY;
        if (v00 > m) { m = v00; sel = 0; }
        if (v01 > m) { m = v01; sel = 1; }
        if (v10 > m) { m = v10; sel = 2; }
        if (v11 > m) { m = v11; sel = 3; }
        const float g0 = sel == 0 ? g : 0.f, g1 = sel == 1 ? g : 0.f, g2 = sel == 2 ? g : 0.f, g3 = sel == 3 ? g : 0.f;
        if (accumulate) {
            dp[o] += g0;
            if (w1) dp[o + 1] += g1;
            if (h1) dp[o + W] += g2;
            if (h1 && w1) dp[o + W + 1] += g3;
        } else {
            dp[o] = g0;
            if (w1) dp[o + 1] = g1;
            if (h1) dp[o + W] = g2;
            if (h1 && w1) dp[o + W + 1] = g3;
        }
    }
}

// MaxPool2d(2, 2) backward FUSED with the bias + ReLU backward of the convolution that feeds the pool (conv1_2 / conv2_2 /
// conv3_3 of the VGG trunk: the pool is the only reader of their output): dz = (y > 0) * [this element is the first maximum of its
// window] * dy_pool, dbias[c] = sum dz, max |dz| per image -- the pool's input gradient (as large as the activation) is never
// written and read back, and y is read once instead of twice: 2.2 GB less per step on 64 x 300 x 300 x 32.  One block per (n, c) plane.
__global__ __launch_bounds__(256) void maxpool2x2_bias_relu_bwd_kernel(const float* __restrict__ y, int y_ctot, int y_coff,
                                                                       const float* __restrict__ dy, int C, int H, int W,
                                                                       int OH, int OW, float* __restrict__ dz, int dz_ctot,
                                                                       int dz_coff, float* __restrict__ dbias,
                                                                       unsigned* __restrict__ amax)
{
    __shared__ double red[4];
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const int WH = (H + 1) >> 1, WW = (W + 1) >> 1;
    const float* xp = y + ((size_t)n * y_ctot + y_coff + c) * H * W;
    float* dp = dz + ((size_t)n * dz_ctot + dz_coff + c) * H * W;
    const float* gp = dy + (size_t)blockIdx.x * OH * OW;
    double sb = 0.0;
    float run = 0.f;
    for (unsigned idx = threadIdx.x; idx < (unsigned)(WH * WW); idx += 256) {
        const int wh = idx / (unsigned)WW, ww = idx - wh * WW;
        const int h0 = 2 * wh, w0 = 2 * ww;
        const bool h1 = h0 + 1 < H, w1 = w0 + 1 < W;
        const float g = (wh < OH && ww < OW) ? gp[wh * OW + ww] : 0.f;
        const int o = h0 * W + w0;
        const float v00 = xp[o];
        const float v01 = w1 ? xp[o + 1] : -INFINITY;
        const float v10 = h1 ? xp[o + W] : -INFINITY;
        const float v11 = (h1 && w1) ? xp[o + W + 1] : -INFINITY;
        int sel = -1;
        float m = -INFINITY;
        if (v00 > m) { m = v00; sel = 0; }
        if (v01 > m) { m = v01; sel = 1; }
        if (v10 > m) { m = v10; sel = 2; }
        if (v11 > m) { m = v11; sel = 3; }
        const float gm = m <= 0.f ? 0.f : g;               // ReLU: the selected element passes its gradient only where y > 0
        dp[o] = sel == 0 ? gm : 0.f;
        if (w1) dp[o + 1] = sel == 1 ? gm : 0.f;
        if (h1) dp[o + W] = sel == 2 ? gm : 0.f;
        if (h1 && w1) dp[o + W + 1] = sel == 3 ? gm : 0.f;
        sb += (double)gm;
        ctdet::h2::track_absmax(run, gm);
    }
    if (amax) ctdet::h2::flush_absmax(amax, n, run);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0 && dbias) unsafeAtomicAdd(&dbias[c], (float)sb);
}

// gradient of the channels-last head scatter: dz[n][co][pix] = dflat[n*img_stride + base + pix*ps + (co-co0)]
struct HeadGatherArgs {
    ct_out_segment seg[3];
    int nseg;
    float* dz;
    int batch, C, HW;
};

__global__ __launch_bounds__(256) void head_grad_gather_kernel(const HeadGatherArgs a)
{
    const long total = (long)a.batch * a.C * a.HW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.HW);
        const unsigned t = idx / a.HW;
        const int c = (int)(t % a.C);
        const int n = (int)(t / a.C);
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            if (g < a.nseg && c >= a.seg[g].co_begin && c < a.seg[g].co_end)
                v = a.seg[g].ptr[(size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                 (size_t)i * a.seg[g].pix_stride + (c - a.seg[g].co_begin)];
        a.dz[idx] = v;
    }
}

inline int grid_for(long total) { return (int)std::min<long>((total + 255) / 256, 256 * 16); }

}  // namespace

static int wgrad_impl(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw, bool zero,
                      ct_stream_t stream);

extern "C" int ct_conv2d_wgrad(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff,
                               float* dw, ct_stream_t stream)
{
    CT_REQUIRE(d && d->in && dz && dw, "ct_conv2d_wgrad: null pointer");
    CT_REQUIRE(d->batch > 0 && d->cin > 0 && d->cout > 0, "ct_conv2d_wgrad: bad shape");
    // buffers above 2 GiB (32-bit buffer offsets): batch chunks accumulate into the same dw
    const long long img_x = (long long)d->in_ctot * d->h * d->w * 4, img_z = (long long)dz_ctot * d->oh * d->ow * 4;
    CT_REQUIRE(img_x < kMaxBufBytes && img_z < kMaxBufBytes, "ct_conv2d_wgrad: one image exceeds 2 GiB");
    const int max_chunk = (int)std::max<long long>(1, (kMaxBufBytes - 1) / std::max(img_x, img_z));
    if (d->batch <= max_chunk) return wgrad_impl(d, dz, dz_ctot, dz_coff, dw, true, stream);
    for (int b0 = 0; b0 < d->batch; b0 += max_chunk) {
        ct_conv_desc sub = *d;
        sub.batch = std::min(max_chunk, d->batch - b0);
        sub.in = d->in + (size_t)b0 * (img_x / 4);
        const int rc = wgrad_impl(&sub, dz + (size_t)b0 * (img_z / 4), dz_ctot, dz_coff, dw, b0 == 0, stream);
        if (rc != CT_OK) return rc;
    }
    return CT_OK;
}

static int wgrad_impl(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw, bool zero,
                      ct_stream_t stream)
{
    CT_REQUIRE(d && d->in && dz && dw, "ct_conv2d_wgrad: null pointer");
    CT_REQUIRE(d->batch > 0 && d->cin > 0 && d->cout > 0, "ct_conv2d_wgrad: bad shape");
    CT_REQUIRE(dz_coff >= 0 && dz_coff + d->cout <= dz_ctot, "ct_conv2d_wgrad: dZ slice");
    const int eoh = (d->h + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int eow = (d->w + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    CT_REQUIRE(eoh == d->oh && eow == d->ow, "ct_conv2d_wgrad: oh/ow mismatch");
    const long long x_bytes = (long long)d->batch * d->in_ctot * d->h * d->w * 4;
    const long long z_bytes = (long long)d->batch * dz_ctot * d->oh * d->ow * 4;
    CT_REQUIRE(x_bytes < kMaxBufBytes && z_bytes < kMaxBufBytes,
               "ct_conv2d_wgrad: buffers above 2 GiB are not supported yet (split the batch)");
    WgradArgs a{};
    a.x = d->in; a.dz = dz; a.dw = dw;
    a.x_bytes = (unsigned)x_bytes; a.dz_bytes = (unsigned)z_bytes;
    a.Cin = d->cin; a.H = d->h; a.W = d->w; a.x_ctot = d->in_ctot; a.x_coff = d->in_coff;
    a.Cout = d->cout; a.OW = d->ow; a.OHW = d->oh * d->ow; a.dz_ctot = dz_ctot; a.dz_coff = dz_coff;
    a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.dil = d->dil;
    a.Npix = d->batch * a.OHW;
    a.Ncols = d->cin * d->kh * d->kw;
    // the 128x128 variant (TB = 2) measured slower on the RFBNet shapes (fewer pixel splits in flight, the
    // 64 gathers per thread arrive in one burst); kept for experiments behind CTDET_WGRAD_TB=2
    static const int tb_env = getenv("CTDET_WGRAD_TB") ? atoi(getenv("CTDET_WGRAD_TB")) : 1;
    const int tb = (tb_env == 2 && d->cout >= 96 && a.Ncols >= 96) ? 2 : 1;
    const int bt = 64 * tb;
    a.tiles_m = (d->cout + bt - 1) / bt;
    a.tiles_n = (a.Ncols + bt - 1) / bt;
    const int tiles = a.tiles_m * a.tiles_n;
    // Pixel splits: the chip holds `slots` workgroups at a time (2 per CU with the double-buffered LDS); take the
    // smallest split count whose last round is at least 92 % full -- one workgroup too many costs a whole round,
    // and every extra split pays the atomic epilogue again.
    static const int slots = getenv("CTDET_WGRAD_WGS") ? atoi(getenv("CTDET_WGRAD_WGS")) : 512;
    const int smax = std::max(1, std::min((a.Npix + 255) / 256, (4 * slots) / tiles));
    int splits = 1;
    double best = 0.0;
    for (int sp = 1; sp <= smax; ++sp) {
        const int wg = tiles * sp, rounds = (wg + slots - 1) / slots;
        const double eff = (double)wg / ((double)rounds * slots);
        if (eff > best + 1e-9) { best = eff; splits = sp; }
        if (eff >= 0.92) break;
    }
    a.pix_per_split = ((a.Npix + splits - 1) / splits + 63) / 64 * 64;
    splits = (a.Npix + a.pix_per_split - 1) / a.pix_per_split;
    hipStream_t st = ctdet::as_stream(stream);
    if (zero && !ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(dw, 0, (size_t)d->cout * a.Ncols * 4, st));
    const bool tapmajor = d->cin % bt == 0 && !(getenv("CTDET_WGRAD_GENERIC"));
    const bool tapmajor_for_smem = tapmajor;
    const dim3 grid(tiles, splits), block(256);
    const bool pipelined = tapmajor_for_smem && tb == 1;
    const size_t smem = (pipelined ? 4 : 2) * 64 * (size_t)(bt + 1) * 4;
    hipError_t le = hipSuccess;
    auto go = [&](auto kernel) {
        if (smem > 64 * 1024) {
            static std::mutex mu;
            static std::unordered_set<const void*> done;
            std::lock_guard<std::mutex> lk(mu);
            if (!done.count((const void*)kernel)) {
                le = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                done.insert((const void*)kernel);
            }
        }
        if (le == hipSuccess) hipLaunchKernelGGL(kernel, grid, block, smem, st, a);
    };
#define CT_WGRAD_GO(KH, KW)                                                    \
    do {                                                                       \
        if (tb == 2 && tapmajor) go(conv_wgrad_f32<KH, KW, 2, true>);          \
        else if (tb == 2) go(conv_wgrad_f32<KH, KW, 2, false>);                \
        else if (tapmajor) go(conv_wgrad_f32<KH, KW, 1, true>);                \
        else go(conv_wgrad_f32<KH, KW, 1, false>);                             \
    } while (0)
    if (d->kh == 3 && d->kw == 3) CT_WGRAD_GO(3, 3);
    else if (d->kh == 1 && d->kw == 1) CT_WGRAD_GO(1, 1);
    else if (d->kh == 1 && d->kw == 3) CT_WGRAD_GO(1, 3);
    else if (d->kh == 3 && d->kw == 1) CT_WGRAD_GO(3, 1);
    else if (d->kh == 4 && d->kw == 4) CT_WGRAD_GO(4, 4);
    else return ctdet::fail(CT_ERR_UNSUPPORTED, "ct_conv2d_wgrad: %dx%d filters not built", d->kh, d->kw);
#undef CT_WGRAD_GO
    CT_HIP(le);
    CT_LAUNCH_CHECK("conv_wgrad_f32");
    return CT_OK;
}

extern "C" int ct_bn_train_stats(const float* z, int batch, int ctot, int coff, int channels, int hw,
                                 float* mean, float* var, float momentum, float* running_mean,
                                 float* running_var, void* scratch, ct_stream_t stream)
{
    CT_REQUIRE(z && mean && var && batch > 0 && channels > 0 && hw > 0, "ct_bn_train_stats: bad arguments");
    hipStream_t st = ctdet::as_stream(stream);
    const long per_channel = (long)batch * hw;
    const int slices = (int)std::max<long>(1, std::min<long>((1024 + channels - 1) / channels, per_channel / 2048));
    const bool run = running_mean && running_var;
    float* const rm = run ? running_mean : nullptr;
    float* const rv = run ? running_var : nullptr;
    const float cntf = (float)batch * hw;
    const float unbias = cntf > 1.f ? cntf / (cntf - 1.f) : 1.f;
    if (scratch && slices > 1) {
        const int per_slice = (int)((per_channel + slices - 1) / slices);
        if (!ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(scratch, 0, (size_t)2 * channels * sizeof(double), st));
        hipLaunchKernelGGL(bn_stats_part_kernel, dim3(channels, slices), dim3(256), 0, st, z, batch, ctot, coff,
                           channels, hw, per_slice, (double*)scratch);
        CT_LAUNCH_CHECK("bn_stats_part_kernel");
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3((channels + 255) / 256), dim3(256), 0, st,
                           (const double*)scratch, channels, (double)per_channel, mean, var, momentum, unbias, rm, rv);
        CT_LAUNCH_CHECK("bn_stats_final_kernel");
    } else {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(channels), dim3(256), 0, st, z, batch, ctot, coff, hw, mean, var,
                           momentum, unbias, rm, rv);
        CT_LAUNCH_CHECK("bn_stats_kernel");
    }
    return CT_OK;
}

extern "C" int ct_bn_train_apply(const float* z, const float* mean, const float* var, const float* gamma,
                                 const float* beta, float eps, int relu, const float* lo, const float* res,
                                 int res_ctot, int res_coff, float res_scale, float* y, int y_ctot,
                                 int y_coff, int z_ctot, int z_coff, int batch, int channels, int hw,
                                 ct_stream_t stream)
{
    CT_REQUIRE(z && mean && var && gamma && beta && y, "ct_bn_train_apply: null pointer");
    BnApplyArgs a{};
    a.z = z; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.lo = lo; a.res = res; a.y = y;
    a.batch = batch; a.C = channels; a.HW = hw; a.y_ctot = y_ctot; a.y_coff = y_coff;
    a.res_ctot = res_ctot; a.res_coff = res_coff; a.eps = eps; a.rscale = res_scale; a.relu = relu;
    a.z_ctot = z_ctot; a.z_coff = z_coff;
    CT_REQUIRE((long)batch * channels * hw < 0xFFFFFFFFL, "ct_bn_train_apply: more than 2^32 elements");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for((long)batch * channels * hw)), dim3(256), 0,
                       ctdet::as_stream(stream), a);
    CT_LAUNCH_CHECK("bn_apply_kernel");
    return CT_OK;
}

static int bn_backward_impl(int frozen, const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot,
                            int y_coff, const float* z, const float* mean, const float* var,
                            const float* gamma, float eps, int relu, const float* lo,
                            float res_scale, float* dres, int dres_ctot, int dres_coff,
                            int dres_accumulate, float* dz, float* dgamma, float* dbeta,
                            int z_ctot, int z_coff, int batch, int channels, int hw,
                            void* scratch, ct_stream_t stream)
{
    CT_REQUIRE(dy && z && mean && var && gamma && dz && dgamma && dbeta, "ct_bn_train_backward: null pointer");
    CT_REQUIRE(!(relu || lo) || y, "ct_bn_train_backward: ReLU mask needs the forward output");
    BnBwdArgs a{};
    a.dy = dy; a.y = y; a.z = z; a.mean = mean; a.var = var; a.gamma = gamma; a.lo = lo;
    a.dz = dz; a.dgamma = dgamma; a.dbeta = dbeta; a.dres = dres;
    a.batch = batch; a.C = channels; a.HW = hw; a.dy_ctot = dy_ctot; a.dy_coff = dy_coff;
    a.y_ctot = y_ctot; a.y_coff = y_coff; a.dres_ctot = dres_ctot; a.dres_coff = dres_coff;
    a.dres_accumulate = dres_accumulate; a.eps = eps; a.rscale = res_scale; a.relu = relu;
    a.z_ctot = z_ctot; a.z_coff = z_coff;
    a.frozen = frozen;
    hipStream_t st = ctdet::as_stream(stream);
    const long per_channel = (long)batch * hw;
    const int slices = (int)std::max<long>(1, std::min<long>((1024 + channels - 1) / channels, per_channel / 2048));
    if (scratch && slices > 1) {
        const int per_slice = (int)((per_channel + slices - 1) / slices);
        if (!ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(scratch, 0, (size_t)2 * channels * sizeof(double), st));
        hipLaunchKernelGGL(bn_bwd_reduce_part_kernel, dim3(channels, slices), dim3(256), 0, st, a, per_slice,
                           (double*)scratch);
        CT_LAUNCH_CHECK("bn_bwd_reduce_part_kernel");
        hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3((channels + 255) / 256), dim3(256), 0, st,
                           (const double*)scratch, channels, dbeta, dgamma);
        CT_LAUNCH_CHECK("bn_bwd_reduce_final_kernel");
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(channels), dim3(256), 0, st, a);
        CT_LAUNCH_CHECK("bn_bwd_reduce_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((long)batch * channels * hw)), dim3(256), 0, st, a);
    CT_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return CT_OK;
}

extern "C" int ct_bn_train_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot,
                                    int y_coff, const float* z, const float* mean, const float* var,
                                    const float* gamma, float eps, int relu, const float* lo,
                                    float res_scale, float* dres, int dres_ctot, int dres_coff,
                                    int dres_accumulate, float* dz, float* dgamma, float* dbeta,
                                    int z_ctot, int z_coff, int batch, int channels, int hw,
                                    void* scratch, ct_stream_t stream)
{
    return bn_backward_impl(0, dy, dy_ctot, dy_coff, y, y_ctot, y_coff, z, mean, var, gamma, eps, relu, lo, res_scale,
                            dres, dres_ctot, dres_coff, dres_accumulate, dz, dgamma, dbeta, z_ctot, z_coff, batch,
                            channels, hw, scratch, stream);
}

extern "C" int ct_bn_eval_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot,
                                   int y_coff, const float* z, const float* running_mean, const float* running_var,
                                   const float* gamma, float eps, int relu, const float* lo,
                                   float res_scale, float* dres, int dres_ctot, int dres_coff,
                                   int dres_accumulate, float* dz, float* dgamma, float* dbeta,
                                   int z_ctot, int z_coff, int batch, int channels, int hw,
                                   void* scratch, ct_stream_t stream)
{
    return bn_backward_impl(1, dy, dy_ctot, dy_coff, y, y_ctot, y_coff, z, running_mean, running_var, gamma, eps, relu,
                            lo, res_scale, dres, dres_ctot, dres_coff, dres_accumulate, dz, dgamma, dbeta, z_ctot,
                            z_coff, batch, channels, hw, scratch, stream);
}

extern "C" int ct_bias_act_backward_amax(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot,
                                         int y_coff, int relu, int batch, int channels, int hw, float* dz,
                                         int dz_ctot, int dz_coff, float* dbias, unsigned* dz_absmax, ct_stream_t stream)
{
    CT_REQUIRE(dy && dz && (!relu || y), "ct_bias_act_backward: null pointer");
    CT_REQUIRE(batch > 0 && channels > 0 && hw > 0, "ct_bias_act_backward: bad shape");
    hipStream_t st = ctdet::as_stream(stream);
    const long per_channel = (long)batch * hw;
    int slices = (int)std::max<long>(1, std::min<long>((2048 + channels - 1) / channels, (per_channel + 4095) / 4096));
    slices = std::min(slices, 65535);
    int per_slice = (int)((per_channel + slices - 1) / slices);
    per_slice = (per_slice + 3) / 4 * 4;
    slices = (int)((per_channel + per_slice - 1) / per_slice);
    if (dbias && !ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(dbias, 0, (size_t)channels * 4, st));
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(channels, slices), dim3(256), 0, st, dy, dy_ctot, dy_coff, y,
                       y_ctot, y_coff, relu, batch, channels, hw, dz, dz_ctot, dz_coff, dbias, per_slice, dz_absmax);
    CT_LAUNCH_CHECK("bias_act_bwd_kernel");
    return CT_OK;
}

extern "C" int ct_bias_act_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot,
                                    int y_coff, int relu, int batch, int channels, int hw, float* dz,
                                    int dz_ctot, int dz_coff, float* dbias, ct_stream_t stream)
{
    return ct_bias_act_backward_amax(dy, dy_ctot, dy_coff, y, y_ctot, y_coff, relu, batch, channels, hw, dz, dz_ctot, dz_coff, dbias,
                                     nullptr, stream);
}

extern "C" int ct_maxpool2d_bwd(const float* x, const float* dy, float* dx, long planes, int h, int w,
                                int oh, int ow, int k, int stride, int pad, int accumulate,
                                ct_stream_t stream)
{
    CT_REQUIRE(x && dy && dx && planes > 0, "ct_maxpool2d_bwd: bad arguments");
    CT_REQUIRE(planes * h * w < 0xFFFFFFFFL, "ct_maxpool2d_bwd: more than 2^32 elements");
    if (k == 2 && stride == 2 && pad == 0 && oh <= (h + 1) / 2 && ow <= (w + 1) / 2 && planes <= 0x7FFFFFFFL) {
        hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3((unsigned)planes), dim3(256), 0, ctdet::as_stream(stream), x,
                           dy, dx, h, w, oh, ow, accumulate);
        CT_LAUNCH_CHECK("maxpool2x2_bwd_kernel");
        return CT_OK;
    }
    if (h * w <= kPoolPlaneMax && oh * ow <= kPoolPlaneMax && planes <= 0x7FFFFFFFL) {
        hipLaunchKernelGGL(maxpool2d_bwd_plane_kernel, dim3((unsigned)planes), dim3(256), 0, ctdet::as_stream(stream), x, dy, dx,
                           h, w, oh, ow, k, stride, pad, accumulate);
        CT_LAUNCH_CHECK("maxpool2d_bwd_plane_kernel");
        return CT_OK;
    }
    hipLaunchKernelGGL(maxpool2d_bwd_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0,
                       ctdet::as_stream(stream), x, dy, dx, planes, h, w, oh, ow, k, stride, pad, accumulate);
    CT_LAUNCH_CHECK("maxpool2d_bwd_kernel");
    return CT_OK;
}

extern "C" int ct_maxpool2x2_bias_relu_bwd(const float* y, int y_ctot, int y_coff, const float* dy, int batch, int channels, int h,
                                           int w, int oh, int ow, float* dz, int dz_ctot, int dz_coff, float* dbias,
                                           unsigned* dz_absmax, ct_stream_t stream)
{
    CT_REQUIRE(y && dy && dz && batch > 0 && channels > 0 && h > 0 && w > 0, "ct_maxpool2x2_bias_relu_bwd: bad arguments");
    CT_REQUIRE(oh <= (h + 1) / 2 && ow <= (w + 1) / 2 && oh >= h / 2 && ow >= w / 2, "ct_maxpool2x2_bias_relu_bwd: %dx%d is not a 2x2 / stride 2 "
               "pooling of %dx%d", oh, ow, h, w);
    CT_REQUIRE(y_coff >= 0 && y_coff + channels <= y_ctot && dz_coff >= 0 && dz_coff + channels <= dz_ctot, "ct_maxpool2x2_bias_relu_bwd: channel slice");
    CT_REQUIRE((long)batch * channels <= 0x7FFFFFFFL, "ct_maxpool2x2_bias_relu_bwd: too many planes");
    hipStream_t st = ctdet::as_stream(stream);
    if (dbias && !ctdet::scratch_prezeroed()) CT_HIP(hipMemsetAsync(dbias, 0, (size_t)channels * 4, st));
    hipLaunchKernelGGL(maxpool2x2_bias_relu_bwd_kernel, dim3((unsigned)(batch * channels)), dim3(256), 0, st, y, y_ctot, y_coff, dy,
                       channels, h, w, oh, ow, dz, dz_ctot, dz_coff, dbias, dz_absmax);
    CT_LAUNCH_CHECK("maxpool2x2_bias_relu_bwd_kernel");
    return CT_OK;
}

extern "C" int ct_head_grad_gather(const ct_out_segment* segs, int nseg, int batch, int channels, int hw,
                                   float* dz, ct_stream_t stream)
{
    CT_REQUIRE(segs && dz && nseg >= 1 && nseg <= 3, "ct_head_grad_gather: bad arguments");
    HeadGatherArgs a{};
    for (int g = 0; g < nseg; ++g) a.seg[g] = segs[g];
    a.nseg = nseg; a.dz = dz; a.batch = batch; a.C = channels; a.HW = hw;
    hipLaunchKernelGGL(head_grad_gather_kernel, dim3(grid_for((long)batch * channels * hw)), dim3(256), 0,
                       ctdet::as_stream(stream), a);
    CT_LAUNCH_CHECK("head_grad_gather_kernel");
    return CT_OK;
}
