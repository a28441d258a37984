// libctdet: host-side NMS entry points of the reference's `--cpu` path
// (utils/nms/cpu_nms.pyx:17-68 `cpu_nms`, :70-163 `cpu_soft_nms`).  These ARE the reference's
// CPU API (utils/nms_wrapper.py:27-30 `force_cpu`), not a fallback for the device kernels.
// Built with -ffp-contract=off so fp32 expressions round exactly like the Cython/C original.
#include "ct_common.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

extern "C" int ct_cpu_nms(const float* dets, int n, float thresh, int ge, int* keep_out, int* num_out)
{
    CT_REQUIRE(num_out && (n == 0 || (dets && keep_out)) && n >= 0, "ct_cpu_nms: bad arguments");
    *num_out = 0;
    if (n == 0) return CT_OK;
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    // descending score, lower index first on ties (the build's defined tie order)
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return dets[(size_t)a * 5 + 4] > dets[(size_t)b * 5 + 4]; });
    std::vector<float> area(n);
    for (int i = 0; i < n; ++i) {
        const float* b = dets + (size_t)i * 5;
        area[i] = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    }
    std::vector<unsigned char> dead(n, 0);
    int nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        const int i = order[_i];
        if (dead[i]) continue;
        keep_out[nk++] = i;
        const float* a = dets + (size_t)i * 5;
        const float ix1 = a[0], iy1 = a[1], ix2 = a[2], iy2 = a[3], iarea = area[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            const int j = order[_j];
            if (dead[j]) continue;
            const float* b = dets + (size_t)j * 5;
            const float xx1 = ix1 >= b[0] ? ix1 : b[0];
            const float yy1 = iy1 >= b[1] ? iy1 : b[1];
            const float xx2 = ix2 <= b[2] ? ix2 : b[2];
            const float yy2 = iy2 <= b[3] ? iy2 : b[3];
            const float w0 = xx2 - xx1 + 1.0f, h0 = yy2 - yy1 + 1.0f;
            const float w = 0.0f >= w0 ? 0.0f : w0;
            const float h = 0.0f >= h0 ? 0.0f : h0;
            const float inter = w * h;
            const float ovr = inter / (iarea + area[j] - inter);
            if (ge ? (ovr >= thresh) : (ovr > thresh)) dead[j] = 1;
        }
    }
    *num_out = nk;
    return CT_OK;
}

extern "C" int ct_cpu_soft_nms(float* boxes, int n, float sigma, float Nt, float threshold,
                               unsigned method, int* n_out)
{
    CT_REQUIRE(n_out && (n == 0 || boxes) && n >= 0, "ct_cpu_soft_nms: bad arguments");
    int N = n;
    auto B = [&](int r, int c) -> float& { return boxes[(size_t)r * 5 + c]; };
    for (int i = 0; i < N; ++i) {
        float maxscore = B(i, 4);
        int maxpos = i;
        float t[5];
        for (int c = 0; c < 5; ++c) t[c] = B(i, c);
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < B(pos, 4)) {
                maxscore = B(pos, 4);
                maxpos = pos;
            }
        for (int c = 0; c < 5; ++c) B(i, c) = B(maxpos, c);
        for (int c = 0; c < 5; ++c) B(maxpos, c) = t[c];
        const float tx1 = B(i, 0), ty1 = B(i, 1), tx2 = B(i, 2), ty2 = B(i, 3);
        int pos = i + 1;
        while (pos < N) {
            const float x1 = B(pos, 0), y1 = B(pos, 1), x2 = B(pos, 2), y2 = B(pos, 3);
            const float area = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);
            const float iw = std::min(tx2, x2) - std::max(tx1, x1) + 1.0f;
            if (iw > 0) {
                const float ih = std::min(ty2, y2) - std::max(ty1, y1) + 1.0f;
                if (ih > 0) {
                    const float ua = (tx2 - tx1 + 1.0f) * (ty2 - ty1 + 1.0f) + area - iw * ih;
                    const float ov = iw * ih / ua;
                    float weight;
                    if (method == 1) weight = ov > Nt ? 1.0f - ov : 1.0f;
                    else if (method == 2) weight = (float)std::exp(-((double)ov * (double)ov) / (double)sigma);
                    else weight = ov > Nt ? 0.0f : 1.0f;
                    B(pos, 4) = weight * B(pos, 4);
                    if (B(pos, 4) < threshold) {
                        for (int c = 0; c < 5; ++c) B(pos, c) = B(N - 1, c);
                        --N;
                        --pos;
                    }
                }
            }
            ++pos;
        }
    }
    *n_out = N;
    return CT_OK;
}
