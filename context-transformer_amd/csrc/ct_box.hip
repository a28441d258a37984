// libctdet: prior-box decode / encode, Detect score fusion, last-dim softmax, pairwise IoU and
// batched prior<->ground-truth matching.  All HBM-bound elementwise / small-reduction kernels.
//
// This translation unit is compiled with -ffp-contract=off: every fp32 expression keeps the
// reference's rounding sequence (no FMA contraction), so integer results derived from these
// values (arg-max indices, match labels) are reproducible against the CPU path.
#include "ct_common.h"
#include <cstdint>
#include <mutex>
#include <algorithm>

#pragma clang fp contract(off)

namespace {

inline int grid_for(long total, int block = 256)
{
    return (int)std::min<long>((total + block - 1) / block, 256 * 16);
}

// utils/box_utils.py:184-202.  x2y2 is formed from the already rounded x1y1 (:200-201).
__device__ __forceinline__ float4 decode_one(const float4 l, const float4 p, float v0, float v1)
{
    const float cx = p.x + l.x * v0 * p.z;
    const float cy = p.y + l.y * v0 * p.w;
    const float w = p.z * expf(l.z * v1);
    const float h = p.w * expf(l.w * v1);
    float4 o;
    o.x = cx - w / 2.f;
    o.y = cy - h / 2.f;
    o.z = w + o.x;
    o.w = h + o.y;
    return o;
}

__global__ __launch_bounds__(256) void decode_kernel(const float4* __restrict__ loc,
                                                     const float4* __restrict__ priors, int batch,
                                                     int P, float v0, float v1,
                                                     const float* __restrict__ scale4, int per_image,
                                                     float4* __restrict__ boxes)
{
    const long total = (long)batch * P;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int p = (int)(idx % P);
        float4 o = decode_one(loc[idx], priors[p], v0, v1);
        if (scale4) {
            const float* s = scale4 + (per_image ? 4 * (idx / P) : 0);
            o.x *= s[0]; o.y *= s[1]; o.z *= s[2]; o.w *= s[3];
        }
        boxes[idx] = o;
    }
}

// utils/box_utils.py:135-156
__global__ __launch_bounds__(256) void encode_kernel(const float4* __restrict__ matched,
                                                     const float4* __restrict__ priors, int P,
                                                     float v0, float v1, float4* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 m = matched[i], p = priors[i];
    float4 o;
    o.x = ((m.x + m.z) / 2.f - p.x) / (v0 * p.z);
    o.y = ((m.y + m.w) / 2.f - p.y) / (v0 * p.w);
    o.z = logf((m.z - m.x) / p.z) / v1;
    o.w = logf((m.w - m.y) / p.w) / v1;
    out[i] = o;
}

// layers/functions/detection.py:44-53 (+ models/RFB_Net_vgg.py:282-284 when SOFTMAX)
// One workgroup = 256 consecutive (image, prior) rows.  The class rows (C floats in, C+1 out) are contiguous
// in memory, so the block copies them through LDS with fully coalesced loads / stores (odd row stride in LDS ->
// conflict-free per-row access); each thread then does its row's arithmetic
// in exactly the reference's order (max, sum of exp, exp / sum, * obj1).
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void detect_kernel(const float4* __restrict__ loc,
                                                     const float* __restrict__ conf,
                                                     const float2* __restrict__ obj,
                                                     const float4* __restrict__ priors, int batch,
                                                     int P, int C, float v0, float v1,
                                                     const float* __restrict__ scale4, int per_image,
                                                     float4* __restrict__ boxes, float* __restrict__ scores)
{
    extern __shared__ float tile[];                    // [256][LDC], row = [obj0, class 0 .. C-1]
    const int LDC = (C + 1) | 1;                       // odd row stride: conflict-free per-row access
    const long total = (long)batch * P;
    const long row0 = (long)blockIdx.x * 256;
    const int rows = (int)min((long)256, total - row0);
    const float* cin = conf + row0 * C;
    for (int e = threadIdx.x; e < rows * C; e += 256) {
        const int r = e / C, k = e - r * C;
        tile[r * LDC + 1 + k] = cin[e];
    }
    __syncthreads();
    const int r = threadIdx.x;
    if (r < rows) {
        const long idx = row0 + r;
        const int p = (int)(idx % P);
        float4 bx = decode_one(loc[idx], priors[p], v0, v1);
        if (scale4) {
            const float* sc = scale4 + (per_image ? 4 * (idx / P) : 0);
            bx.x *= sc[0]; bx.y *= sc[1]; bx.z *= sc[2]; bx.w *= sc[3];
        }
        boxes[idx] = bx;
        float2 o = obj[idx];
        float* c = tile + r * LDC + 1;
        if (SOFTMAX) {
            const float om = fmaxf(o.x, o.y);
            const float e0 = expf(o.x - om), e1 = expf(o.y - om);
            const float os = e0 + e1;
            o.x = e0 / os;
            o.y = e1 / os;
            float m = -INFINITY;
            for (int k = 0; k < C; ++k) m = fmaxf(m, c[k]);
            float sum = 0.f;
            for (int k = 0; k < C; ++k) {
                const float ex = expf(c[k] - m);
                c[k] = ex;
                sum += ex;
            }
            for (int k = 0; k < C; ++k) c[k] = o.y * (c[k] / sum);
        } else {
            for (int k = 0; k < C; ++k) c[k] = o.y * c[k];
        }
        c[-1] = o.x;
    }
    __syncthreads();
    float* sout = scores + row0 * (C + 1);
    for (int e = threadIdx.x; e < rows * (C + 1); e += 256) {
        const int rr = e / (C + 1), k = e - rr * (C + 1);
        sout[e] = tile[rr * LDC + k];
    }
}

// The same for C <= CMAX classes with the row in registers (round 6).  detect_kernel above spends most of its time on the
// integer divisions of its strided LDS copies (e / C per element, twice): here the 256 rows of a workgroup are ONE contiguous,
// 16-byte aligned block of 256 C floats in and 256 (C + 1) floats out, copied linearly with 16-byte accesses; a thread reads its
// row from the linear tile (16-byte reads where C % 4 == 0: conflict-free for the C = 20 / 60 of the reference's settings),
// keeps it in registers across the barrier that lets the output tile take the same LDS, and does the row's arithmetic in the
// reference's order -- the same expressions as detect_kernel, hence the same bits.
template <bool SOFTMAX, int CMAX>
__global__ __launch_bounds__(256) void detect_rows_kernel(const float4* __restrict__ loc, const float* __restrict__ conf,
                                                          const float2* __restrict__ obj, const float4* __restrict__ priors,
                                                          int batch, int P, int C, float v0, float v1,
                                                          const float* __restrict__ scale4, int per_image,
                                                          float4* __restrict__ boxes, float* __restrict__ scores)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];           // 256 (C + 1) floats: input rows, then output rows
    const long total = (long)batch * P;
    const long row0 = (long)blockIdx.x * 256;
    const int rows = (int)min((long)256, total - row0);
    {
        const float* cin = conf + row0 * C;                // 16-byte aligned: row0 is a multiple of 256
        const int n = rows * C, n4 = n >> 2;
        for (int e = threadIdx.x; e < n4; e += 256)
            reinterpret_cast<float4*>(tile)[e] = reinterpret_cast<const float4*>(cin)[e];
        for (int e = 4 * n4 + threadIdx.x; e < n; e += 256) tile[e] = cin[e];
    }
    __syncthreads();
    const int r = threadIdx.x;
    float c[CMAX];
    if (r < rows) {
        const float* src = tile + r * C;
        if ((C & 3) == 0) {
#pragma unroll
            for (int k4 = 0; k4 < CMAX / 4; ++k4)
                if (4 * k4 < C) {
                    const float4 f = reinterpret_cast<const float4*>(src)[k4];
                    c[4 * k4] = f.x; c[4 * k4 + 1] = f.y; c[4 * k4 + 2] = f.z; c[4 * k4 + 3] = f.w;
                }
        } else {
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) c[k] = src[k];
        }
    }
    __syncthreads();                                       // every input row is in registers: the tile becomes the output tile
    if (r < rows) {
        const long idx = row0 + r;
        const int p = (int)(idx % P);
        float4 bx = decode_one(loc[idx], priors[p], v0, v1);
        if (scale4) {
            const float* sc = scale4 + (per_image ? 4 * (idx / P) : 0);
            bx.x *= sc[0]; bx.y *= sc[1]; bx.z *= sc[2]; bx.w *= sc[3];
        }
        boxes[idx] = bx;
        float2 o = obj[idx];
        if (SOFTMAX) {
            const float om = fmaxf(o.x, o.y);
            const float e0 = expf(o.x - om), e1 = expf(o.y - om);
            const float os = e0 + e1;
            o.x = e0 / os;
            o.y = e1 / os;
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) m = fmaxf(m, c[k]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) {
                    const float ex = expf(c[k] - m);
                    c[k] = ex;
                    sum += ex;
                }
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) c[k] = o.y * (c[k] / sum);
        } else {
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) c[k] = o.y * c[k];
        }
        float* dst = tile + r * (C + 1);
        dst[0] = o.x;
#pragma unroll
        for (int k = 0; k < CMAX; ++k)
            if (k < C) dst[1 + k] = c[k];
    }
    __syncthreads();
    {
        float* sout = scores + row0 * (C + 1);             // 16-byte aligned
        const int n = rows * (C + 1), n4 = n >> 2;
        for (int e = threadIdx.x; e < n4; e += 256)
            reinterpret_cast<float4*>(sout)[e] = reinterpret_cast<const float4*>(tile)[e];
        for (int e = 4 * n4 + threadIdx.x; e < n; e += 256) sout[e] = tile[e];
    }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, long rows, int cols)
{
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float* x = in + r * cols;
        float* y = out + r * cols;
        float m = -INFINITY;
        for (int k = 0; k < cols; ++k) m = fmaxf(m, x[k]);
        float sum = 0.f;
        for (int k = 0; k < cols; ++k) sum += expf(x[k] - m);
        for (int k = 0; k < cols; ++k) y[k] = expf(x[k] - m) / sum;
    }
}

// utils/box_utils.py:29-68: inter / (area_a + area_b - inter), no +1
__device__ __forceinline__ float iou_plain(const float4 a, const float4 b)
{
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h;
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    return inter / (area_a + area_b - inter);
}

// utils/box_utils.py:5-14
__device__ __forceinline__ float4 point_form(const float4 p)
{
    return make_float4(p.x - p.z / 2.f, p.y - p.w / 2.f, p.x + p.z / 2.f, p.y + p.w / 2.f);
}

__global__ __launch_bounds__(256) void jaccard_kernel(const float4* __restrict__ a, int na,
                                                      const float4* __restrict__ b, int nb,
                                                      int b_center, float* __restrict__ out)
{
    const long total = (long)na * nb;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % nb), i = (int)(idx / nb);
        const float4 bb = b_center ? point_form(b[j]) : b[j];
        out[idx] = iou_plain(a[i], bb);
    }
}

// ---- match (utils/box_utils.py:83-132), two passes -----------------------------------
constexpr int kMaxGT = 256;

// pass 1: one thread per (image, prior): best GT per prior; per-GT best prior through a
// packed 64-bit atomicMax (iou bits << 32 | ~prior) so ties resolve to the LOWEST prior index,
// as torch.max(dim) does on CPU.
__global__ __launch_bounds__(256) void match_pass1(const float* __restrict__ truths,
                                                   const int* __restrict__ gt_off,
                                                   const float4* __restrict__ priors, int P,
                                                   float* __restrict__ best_ov, int* __restrict__ best_idx,
                                                   unsigned long long* __restrict__ gt_best, int max_gt)
{
    __shared__ float4 gtb[kMaxGT];
    __shared__ unsigned long long gbest[kMaxGT];
    const int b = blockIdx.y;
    const int g0 = gt_off[b], G = min(gt_off[b + 1] - g0, max_gt);    // the workspace holds max_gt slots per image
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 pf = point_form(priors[p < P ? p : 0]);
    float bo = -INFINITY;
    int bi = 0;
    // ground-truth boxes in chunks of kMaxGT through LDS: no limit on the number of boxes per image
    for (int c0 = 0; c0 < G || c0 == 0; c0 += kMaxGT) {
        const int n = min(G - c0, kMaxGT);
        for (int g = threadIdx.x; g < n; g += blockDim.x) {
            const float* t = truths + (size_t)(g0 + c0 + g) * 6;
            gtb[g] = make_float4(t[0], t[1], t[2], t[3]);
            gbest[g] = 0ull;
        }
        __syncthreads();
        if (p < P) {
            for (int g = 0; g < n; ++g) {
                const float ov = iou_plain(gtb[g], pf);
                if (ov > bo) { bo = ov; bi = c0 + g; }          // first maximal GT wins
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(ov) << 32) | (unsigned)(~(unsigned)p);
                atomicMax(&gbest[g], key);
            }
        }
        __syncthreads();
        for (int g = threadIdx.x; g < n; g += blockDim.x)
            atomicMax(&gt_best[(size_t)b * max_gt + c0 + g], gbest[g]);
        __syncthreads();
    }
    if (p < P) {
        best_ov[(size_t)b * P + p] = bo;
        best_idx[(size_t)b * P + p] = bi;
    }
}

// pass 2: force-match (later GT wins), labels, encode, outputs
__global__ __launch_bounds__(256) void match_pass2(const float* __restrict__ truths,
                                                   const int* __restrict__ gt_off,
                                                   const float4* __restrict__ priors, int P,
                                                   const float* __restrict__ best_ov,
                                                   const int* __restrict__ best_idx,
                                                   const unsigned long long* __restrict__ gt_best,
                                                   int max_gt, float threshold, float v0, float v1,
                                                   float4* __restrict__ loc_t, float2* __restrict__ conf_t,
                                                   uint8_t* __restrict__ obj_t, float* __restrict__ overlap)
{
    __shared__ int bprior[kMaxGT];
    const int b = blockIdx.y;
    const int g0 = gt_off[b], G = min(gt_off[b + 1] - g0, max_gt);    // the workspace holds max_gt slots per image
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t o = (size_t)b * P + (p < P ? p : 0);
    float ov = best_ov[o];
    int gi = best_idx[o];
    for (int c0 = 0; c0 < G; c0 += kMaxGT) {            // chunks of kMaxGT boxes, in order: the later GT wins
        const int n = min(G - c0, kMaxGT);
        __syncthreads();
        for (int g = threadIdx.x; g < n; g += blockDim.x)
            bprior[g] = (int)(~(unsigned)(gt_best[(size_t)b * max_gt + c0 + g] & 0xFFFFFFFFull));
        __syncthreads();
        for (int g = 0; g < n; ++g)
            if (bprior[g] == p) { ov = 2.f; gi = c0 + g; }
    }
    if (p >= P) return;
    if (overlap) overlap[o] = best_ov[o];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    float label = 0.f, weight = 1.f;
    if (G > 0) {
        const float* t = truths + (size_t)(g0 + gi) * 6;
        label = t[4];
        weight = t[5];
        if (ov < threshold) { label = 0.f; weight = 1.f; }
        const float4 pr = priors[p];
        out.x = ((t[0] + t[2]) / 2.f - pr.x) / (v0 * pr.z);
        out.y = ((t[1] + t[3]) / 2.f - pr.y) / (v0 * pr.w);
        out.z = logf((t[2] - t[0]) / pr.z) / v1;
        out.w = logf((t[3] - t[1]) / pr.w) / v1;
    }
    loc_t[o] = out;
    conf_t[o] = make_float2(label, weight);
    obj_t[o] = label != 0.f ? 1 : 0;
}

}  // namespace

extern "C" int ct_decode(const float* loc, const float* priors, int batch, int num_priors, float var0,
                         float var1, const float* scale4, int scale_per_image, float* boxes,
                         ct_stream_t stream)
{
    CT_REQUIRE(loc && priors && boxes && batch > 0 && num_priors > 0, "ct_decode: bad arguments");
    hipLaunchKernelGGL(decode_kernel, dim3(grid_for((long)batch * num_priors)), dim3(256), 0,
                       ctdet::as_stream(stream), (const float4*)loc, (const float4*)priors, batch,
                       num_priors, var0, var1, scale4, scale_per_image, (float4*)boxes);
    CT_LAUNCH_CHECK("decode_kernel");
    return CT_OK;
}

extern "C" int ct_encode(const float* matched, const float* priors, int num_priors, float var0,
                         float var1, float* out, ct_stream_t stream)
{
    CT_REQUIRE(matched && priors && out && num_priors > 0, "ct_encode: bad arguments");
    hipLaunchKernelGGL(encode_kernel, dim3((num_priors + 255) / 256), dim3(256), 0,
                       ctdet::as_stream(stream), (const float4*)matched, (const float4*)priors,
                       num_priors, var0, var1, (float4*)out);
    CT_LAUNCH_CHECK("encode_kernel");
    return CT_OK;
}

extern "C" int ct_detect_fused(const float* loc, const float* conf, const float* obj,
                               const float* priors, int batch, int num_priors, int num_fg, float var0,
                               float var1, int apply_softmax, const float* scale4, int scale_per_image,
                               float* boxes, float* scores, ct_stream_t stream)
{
    CT_REQUIRE(loc && conf && obj && priors && boxes && scores, "ct_detect_fused: null tensor");
    CT_REQUIRE(batch > 0 && num_priors > 0 && num_fg > 0, "ct_detect_fused: bad shape");
    CT_REQUIRE(num_fg <= 1024, "ct_detect_fused: num_fg=%d (<= 1024)", num_fg);
    const long nblk = ((long)batch * num_priors + 255) / 256;
    CT_REQUIRE(nblk < 0x7FFFFFFFL, "ct_detect_fused: too many rows");
    const dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = ctdet::as_stream(stream);
    // up to 64 classes (every setting of the reference: 15 / 20 / 60): the row-in-registers kernel; conf and scores must be 16-byte
    // aligned for its linear copies (torch allocations are), else -- or with more classes -- the strided-copy kernel below
    const bool aligned = (reinterpret_cast<uintptr_t>(conf) & 15) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
    if (num_fg <= 64 && aligned) {
        const size_t lds = (size_t)256 * (num_fg + 1) * sizeof(float);
        static std::once_flag once_rows;
        static hipError_t er = hipSuccess;
        std::call_once(once_rows, [] {
            const void* fs[] = {(const void*)detect_rows_kernel<true, 32>, (const void*)detect_rows_kernel<false, 32>,
                                (const void*)detect_rows_kernel<true, 64>, (const void*)detect_rows_kernel<false, 64>};
            for (const void* f : fs)
                if (er == hipSuccess) er = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 65 * 4);
        });
        CT_HIP(er);
        CT_PROF("detect_kernel", st);
#define CT_DETECT_ROWS(SM, CM)                                                                                              \
        hipLaunchKernelGGL((detect_rows_kernel<SM, CM>), grid, block, lds, st, (const float4*)loc, conf, (const float2*)obj,     \
                           (const float4*)priors, batch, num_priors, num_fg, var0, var1, scale4, scale_per_image, (float4*)boxes, scores)
        if (num_fg <= 32) { if (apply_softmax) CT_DETECT_ROWS(true, 32); else CT_DETECT_ROWS(false, 32); }
        else { if (apply_softmax) CT_DETECT_ROWS(true, 64); else CT_DETECT_ROWS(false, 64); }
#undef CT_DETECT_ROWS
        CT_LAUNCH_CHECK("detect_rows_kernel");
        return CT_OK;
    }
    const size_t smem = (size_t)256 * (num_fg + 2) * sizeof(float);
    if (smem > 64 * 1024) {
        static std::once_flag once;
        static hipError_t e1 = hipSuccess, e2 = hipSuccess;
        std::call_once(once, [] {
            e1 = hipFuncSetAttribute((const void*)detect_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            e2 = hipFuncSetAttribute((const void*)detect_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        CT_HIP(e1);
        CT_HIP(e2);
        CT_REQUIRE(smem <= 160 * 1024, "ct_detect_fused: num_fg=%d needs %zu bytes of LDS", num_fg, smem);
    }
    if (apply_softmax)
        { CT_PROF("detect_kernel", st); hipLaunchKernelGGL(detect_kernel<true>, grid, block, smem, st, (const float4*)loc, conf,
                           (const float2*)obj, (const float4*)priors, batch, num_priors, num_fg, var0,
                           var1, scale4, scale_per_image, (float4*)boxes, scores); }
    else
        { CT_PROF("detect_kernel", st); hipLaunchKernelGGL(detect_kernel<false>, grid, block, smem, st, (const float4*)loc, conf,
                           (const float2*)obj, (const float4*)priors, batch, num_priors, num_fg, var0,
                           var1, scale4, scale_per_image, (float4*)boxes, scores); }
    CT_LAUNCH_CHECK("detect_kernel");
    return CT_OK;
}

extern "C" int ct_softmax_lastdim(const float* in, float* out, long rows, int cols, ct_stream_t stream)
{
    CT_REQUIRE(in && out && rows > 0 && cols > 0, "ct_softmax_lastdim: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for(rows)), dim3(256), 0,
                       ctdet::as_stream(stream), in, out, rows, cols);
    CT_LAUNCH_CHECK("softmax_rows_kernel");
    return CT_OK;
}

extern "C" int ct_jaccard(const float* a, int na, const float* b, int nb, int b_center_form,
                          float* out, ct_stream_t stream)
{
    CT_REQUIRE(a && b && out && na > 0 && nb > 0, "ct_jaccard: bad arguments");
    hipLaunchKernelGGL(jaccard_kernel, dim3(grid_for((long)na * nb)), dim3(256), 0,
                       ctdet::as_stream(stream), (const float4*)a, na, (const float4*)b, nb,
                       b_center_form, out);
    CT_LAUNCH_CHECK("jaccard_kernel");
    return CT_OK;
}

extern "C" size_t ct_match_workspace_bytes(int batch, int num_priors, int max_gt)
{
    const size_t bp = (size_t)batch * num_priors;
    return ctdet::align_up(bp * 4, 256) + ctdet::align_up(bp * 4, 256) +
           ctdet::align_up((size_t)batch * std::max(max_gt, 1) * 8, 256);
}

extern "C" int ct_match_batched(const float* truths, const int* gt_off, int batch, int max_gt,
                                const float* priors, int num_priors, float threshold, float var0,
                                float var1, float* loc_t, float* conf_t, uint8_t* obj_t, float* overlap,
                                void* workspace, size_t workspace_bytes, ct_stream_t stream)
{
    CT_REQUIRE(truths && gt_off && priors && loc_t && conf_t && obj_t && workspace, "ct_match_batched: null");
    CT_REQUIRE(batch > 0 && num_priors > 0, "ct_match_batched: bad shape");
    CT_REQUIRE(max_gt >= 1, "ct_match_batched: max_gt=%d", max_gt);
    if (workspace_bytes < ct_match_workspace_bytes(batch, num_priors, max_gt))
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_match_batched: workspace %zu < %zu", workspace_bytes,
                           ct_match_workspace_bytes(batch, num_priors, max_gt));
    const size_t bp = (size_t)batch * num_priors;
    char* ws = (char*)workspace;
    float* best_ov = (float*)ws;
    ws += ctdet::align_up(bp * 4, 256);
    int* best_idx = (int*)ws;
    ws += ctdet::align_up(bp * 4, 256);
    unsigned long long* gt_best = (unsigned long long*)ws;
    hipStream_t st = ctdet::as_stream(stream);
    CT_HIP(hipMemsetAsync(gt_best, 0, (size_t)batch * max_gt * 8, st));
    const dim3 grid((num_priors + 255) / 256, batch), block(256);
    hipLaunchKernelGGL(match_pass1, grid, block, 0, st, truths, gt_off, (const float4*)priors,
                       num_priors, best_ov, best_idx, gt_best, max_gt);
    CT_LAUNCH_CHECK("match_pass1");
    hipLaunchKernelGGL(match_pass2, grid, block, 0, st, truths, gt_off, (const float4*)priors,
                       num_priors, best_ov, best_idx, gt_best, max_gt, threshold, var0, var1,
                       (float4*)loc_t, (float2*)conf_t, obj_t, overlap);
    CT_LAUNCH_CHECK("match_pass2");
    return CT_OK;
}
