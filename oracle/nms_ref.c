/* Oracle (TEST INFRASTRUCTURE, not product): scalar C greedy NMS.
 *
 * Restates utils/nms/cpu_nms.pyx:17-68 (cpu_nms, suppress ovr >= thresh) and
 * utils/nms/py_cpu_nms.py:10-38 / utils/nms/nms_kernel.cu:24-32,71 (suppress ovr > thresh)
 * of the reference, on boxes already sorted by descending score (the `_nms` contract,
 * utils/nms/nms_kernel.cu:91-144).  fp32 expression order as the reference:
 *   area = (x2-x1+1)*(y2-y1+1);  w = max(0, xx2-xx1+1);  ovr = inter/(area_i+area_j-inter)
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, IEEE division).
 * Used by tests for full-size cases and by bench.py's cpu_baseline leg only.
 */
#include <stdlib.h>

static inline float fmaxf_(float a, float b) { return a >= b ? a : b; }
static inline float fminf_(float a, float b) { return a <= b ? a : b; }

int oracle_nms_sorted(const float* dets, int n, float thresh, int ge, int* keep)
{
    if (n <= 0) return 0;
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    for (int i = 0; i < n; ++i) {
        const float* b = dets + 5 * (size_t)i;
        area[i] = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    }
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        const float* a = dets + 5 * (size_t)i;
        const float ax1 = a[0], ay1 = a[1], ax2 = a[2], ay2 = a[3], aa = area[i];
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float* b = dets + 5 * (size_t)j;
            float xx1 = fmaxf_(ax1, b[0]);
            float yy1 = fmaxf_(ay1, b[1]);
            float xx2 = fminf_(ax2, b[2]);
            float yy2 = fminf_(ay2, b[3]);
            float w = fmaxf_(0.0f, xx2 - xx1 + 1.0f);
            float h = fmaxf_(0.0f, yy2 - yy1 + 1.0f);
            float inter = w * h;
            float ovr = inter / (aa + area[j] - inter);
            if (ge ? (ovr >= thresh) : (ovr > thresh)) dead[j] = 1;
        }
    }
    free(area);
    free(dead);
    return nk;
}
