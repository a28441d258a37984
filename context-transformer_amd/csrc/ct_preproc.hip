// libctdet: test-time input transform (data/data_augment.py:224-266 BaseTransform):
//   cv2.resize(img, (S,S), INTER_LINEAR) on uint8 HxWx3  ->  float32  ->  -= means  ->  CHW.
// One launch per batch of variable-size images packed back to back in one byte buffer.
// HBM-bound byte work: a thread produces one output pixel (4 source pixels x 3 bytes in,
// three coalesced fp32 plane stores out).  Arithmetic is OpenCV's published 8-bit bilinear
// path (imgproc/resize.cpp: 11-bit fixed-point coefficients, HResizeLinear then
// VResizeLinear with the (b*(s>>4))>>16, +2, >>2 rounding), so results are whole numbers
// exactly as the uint8 image cv2 returns.
#include "ct_common.h"

namespace {

struct Tap {
    int s0, s1;
    int c0, c1;
};

// coefficient pair of one output coordinate.  zero_frac_at_edges: OpenCV clears the fraction
// at the horizontal borders; vertically it only clamps the row indices.
__device__ inline Tap make_tap(int d, double scale, int n, bool zero_frac_at_edges)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    if (zero_frac_at_edges) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n - 1) { f = 0.f; s = n - 1; }
        t.s0 = s;
        t.s1 = min(s + 1, n - 1);
    } else {
        t.s0 = min(max(s, 0), n - 1);
        t.s1 = min(max(s + 1, 0), n - 1);
    }
    t.c0 = __float2int_rn((1.f - f) * 2048.f);
    t.c1 = __float2int_rn(f * 2048.f);
    return t;
}

__global__ __launch_bounds__(256) void resize_sub_chw(const unsigned char* __restrict__ src,
                                                      const long long* __restrict__ offs,
                                                      const int* __restrict__ hw, float* __restrict__ out,
                                                      int S, float m0, float m1, float m2)
{
    const int n = blockIdx.y;
    const int H = hw[2 * n], W = hw[2 * n + 1];
    const unsigned char* img = src + offs[n];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= S * S) return;
    const int dy = pix / S, dx = pix - dy * S;
    const Tap tx = make_tap(dx, (double)W / S, W, true);
    const Tap ty = make_tap(dy, (double)H / S, H, false);
    const unsigned char* r0 = img + (long)ty.s0 * W * 3;
    const unsigned char* r1 = img + (long)ty.s1 * W * 3;
    const float mean[3] = {m0, m1, m2};
    float* o = out + (long)n * 3 * S * S + pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[tx.s0 * 3 + c] * tx.c0 + r0[tx.s1 * 3 + c] * tx.c1;
        const int h1 = r1[tx.s0 * 3 + c] * tx.c0 + r1[tx.s1 * 3 + c] * tx.c1;
        int v = (((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        o[(long)c * S * S] = (float)v - mean[c];
    }
}

}  // namespace

extern "C" int ct_preproc_resize(const unsigned char* src, const long long* offsets, const int* hw,
                                 int batch, int size, const float* means3, float* out, ct_stream_t stream)
{
    CT_REQUIRE(src && offsets && hw && out && means3, "ct_preproc_resize: null pointer");
    CT_REQUIRE(batch > 0 && batch <= 65535 && size > 0 && size <= 4096, "ct_preproc_resize: batch=%d size=%d",
               batch, size);
    dim3 grid((size * size + 255) / 256, batch);
    hipLaunchKernelGGL(resize_sub_chw, grid, dim3(256), 0, ctdet::as_stream(stream), src, offsets, hw, out, size,
                       means3[0], means3[1], means3[2]);
    CT_LAUNCH_CHECK("resize_sub_chw");
    return 0;
}
