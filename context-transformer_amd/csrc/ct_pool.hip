// libctdet: max-pooling kernels (HBM-bound, one output element per thread, coalesced along w
// for NCHW planes and along channels for the channels-last head buffers).
#include "ct_common.h"
#include <algorithm>

namespace {

// IDX = int when planes*OH*OW < 2^31 (always, for this network): 64-bit div/mod per element made the kernel
// instruction-bound at ~0.5 TB/s
template <typename IDX>
__global__ __launch_bounds__(256) void maxpool2d_nchw(const float* __restrict__ in,
                                                      float* __restrict__ out, long planes, int H,
                                                      int W, int OH, int OW, int k, int stride, int pad)
{
    const IDX total = (IDX)(planes * OH * OW);
    for (IDX idx = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x; idx < total;
         idx += (IDX)gridDim.x * (IDX)blockDim.x) {
        const int ow = (int)(idx % OW);
        const IDX t = idx / OW;
        const int oh = (int)(t % OH);
        const IDX pl = t / OH;
        const int h0 = oh * stride - pad, w0 = ow * stride - pad;
        const int h1 = min(h0 + k, H), w1 = min(w0 + k, W);
        const float* p = in + (long)pl * H * W;
        float m = -INFINITY;
        for (int h = max(h0, 0); h < h1; ++h)
            for (int w = max(w0, 0); w < w1; ++w) m = fmaxf(m, p[h * W + w]);
        out[idx] = m;
    }
}

// MaxPool2d(3, 1, 1) on small planes (pool5 of models/RFB_Net_vgg.py:225: 512 x 19x19 / 32x32 per image): the generic
// kernel above spends ~40 instructions per output on index arithmetic and reads every input nine times through the L1
// (75 us for 47 MB at bs 32).  Here a workgroup walks a run of planes: plane -> LDS with coalesced loads, every thread
// computes the outputs whose (row, column, window mask) it worked out ONCE before the loop, coalesced stores.
constexpr int P3_MAX_HW = 4096;            // 16 KB of LDS per buffer, two buffers
constexpr int P3_EPT = P3_MAX_HW / 256;    // elements per thread

__global__ __launch_bounds__(256) void maxpool3x3s1_planes(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                           int H, int W, int planes_per_wg)
{
    __shared__ float tile[2][P3_MAX_HW];
    const int HW = H * W, tid = threadIdx.x;
    int off[P3_EPT];
    unsigned mask[P3_EPT];                   // bit 3 r + c: neighbour (r - 1, c - 1) of the window exists
    const int n_e = (HW + 255) / 256;
#pragma unroll
    for (int j = 0; j < P3_EPT; ++j) {
        const int e = tid + 256 * j;
        off[j] = e;
        mask[j] = 0;
        if (j < n_e && e < HW) {
            const int h = e / W, w = e - h * W;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if ((unsigned)(h + r - 1) < (unsigned)H && (unsigned)(w + c - 1) < (unsigned)W) mask[j] |= 1u << (3 * r + c);
        }
    }
    const int p0 = blockIdx.x * planes_per_wg, p1 = min(planes, p0 + planes_per_wg);
    if (p0 >= p1) return;
    auto load = [&](int pl, int b) {
        const float* src = in + (size_t)pl * HW;
#pragma unroll
        for (int j = 0; j < P3_EPT; ++j)
            if (j < n_e && off[j] < HW) tile[b][off[j]] = src[off[j]];
    };
    load(p0, 0);
    __syncthreads();
    for (int pl = p0; pl < p1; ++pl) {
        const int b = (pl - p0) & 1;
        if (pl + 1 < p1) load(pl + 1, b ^ 1);           // the next plane arrives while this one is reduced
        float* dst = out + (size_t)pl * HW;
#pragma unroll
        for (int j = 0; j < P3_EPT; ++j) {
            if (j >= n_e || off[j] >= HW) continue;
            const float* c0 = &tile[b][off[j]];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (mask[j] >> (3 * r + c) & 1u) m = fmaxf(m, c0[(r - 1) * W + (c - 1)]);
            dst[off[j]] = m;
        }
        __syncthreads();
    }
}

// channels-last: in [n][h*w][ch] -> out [n][oh*ow][ch], kernel = stride = k, ceil_mode
__global__ __launch_bounds__(256) void ctx_pool_nhwc(const float* __restrict__ in, long long in_img,
                                                     float* __restrict__ out, long long out_img,
                                                     int batch, int H, int W, int OH, int OW, int ch, int k)
{
    const long total = (long)batch * OH * OW * ch;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % ch);
        long t = idx / ch;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        const int h1 = min(oh * k + k, H), w1 = min(ow * k + k, W);
        const float* p = in + (long long)n * in_img;
        float m = -INFINITY;
        for (int h = oh * k; h < h1; ++h)
            for (int w = ow * k; w < w1; ++w) m = fmaxf(m, p[((long)h * W + w) * ch + c]);
        out[(long long)n * out_img + ((long)oh * OW + ow) * ch + c] = m;
    }
}

inline int grid_for(long total) { return (int)std::min<long>((total + 255) / 256, 256 * 16); }

}  // namespace

extern "C" int ct_maxpool2d_fwd(const float* in, float* out, long planes, int h, int w, int oh, int ow,
                                int k, int stride, int pad, ct_stream_t stream)
{
    CT_REQUIRE(in && out && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "ct_maxpool2d_fwd: bad shape");
    CT_REQUIRE(k >= 1 && stride >= 1 && pad >= 0 && pad < k, "ct_maxpool2d_fwd: k=%d stride=%d pad=%d", k, stride, pad);
    CT_REQUIRE((oh - 1) * stride - pad < h && (ow - 1) * stride - pad < w,
               "ct_maxpool2d_fwd: last window starts outside the input");
    if (k == 3 && stride == 1 && pad == 1 && oh == h && ow == w && h * w <= P3_MAX_HW && planes < 0x7FFFFFFFL) {
        // about four workgroups per CU, at least two planes each (the second plane's load hides behind the first's reduction)
        const int ppw = (int)std::max<long>(2, (planes + 1023) / 1024);
        hipLaunchKernelGGL(maxpool3x3s1_planes, dim3((int)((planes + ppw - 1) / ppw)), dim3(256), 0, ctdet::as_stream(stream), in, out,
                           (int)planes, h, w, ppw);
        CT_LAUNCH_CHECK("maxpool3x3s1_planes");
        return CT_OK;
    }
    if (planes * oh * ow < 0x7FFFFFFFL)
        hipLaunchKernelGGL(maxpool2d_nchw<int>, dim3(grid_for(planes * oh * ow)), dim3(256), 0,
                           ctdet::as_stream(stream), in, out, planes, h, w, oh, ow, k, stride, pad);
    else
        hipLaunchKernelGGL(maxpool2d_nchw<long>, dim3(grid_for(planes * oh * ow)), dim3(256), 0,
                           ctdet::as_stream(stream), in, out, planes, h, w, oh, ow, k, stride, pad);
    CT_LAUNCH_CHECK("maxpool2d_nchw");
    return CT_OK;
}

extern "C" int ct_ctx_pool_fwd(const float* in, long long in_img_stride, float* out,
                               long long out_img_stride, int batch, int h, int w, int ch, int k,
                               ct_stream_t stream)
{
    CT_REQUIRE(in && out && batch > 0 && h > 0 && w > 0 && ch > 0 && k >= 1, "ct_ctx_pool_fwd: bad shape");
    const int oh = (h + k - 1) / k, ow = (w + k - 1) / k;
    hipLaunchKernelGGL(ctx_pool_nhwc, dim3(grid_for((long)batch * oh * ow * ch)), dim3(256), 0,
                       ctdet::as_stream(stream), in, in_img_stride, out, out_img_stride, batch, h, w,
                       oh, ow, ch, k);
    CT_LAUNCH_CHECK("ctx_pool_nhwc");
    return CT_OK;
}
