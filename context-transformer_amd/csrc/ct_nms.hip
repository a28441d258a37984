// libctdet: greedy NMS with the reference's +1 pixel convention, bit-exact keep lists.
//
// Reference: utils/nms/nms_kernel.cu:24-144 builds an N x N/64 bit matrix of "IoU > thresh"
// on the device, copies it to the host and reduces it sequentially there.  Only rows of boxes
// that end up KEPT are ever used by that reduction, so this kernel never computes the others:
//
//   one workgroup (4 wave64) per segment = one (image, class) problem, boxes sorted by score;
//   the kept boxes live in an LDS list (x1,y1,x2,y2 + area, broadcast ds_reads);
//   candidates are taken 256 at a time, one per lane:
//     phase A  every lane tests its candidate against the kept list so far (uniform loop,
//              ~N*K/2 IoUs in total instead of N^2/2);
//     phase B  the four 64-candidate sub-chunks are resolved in order by their own wave with
//              a ballot loop (lowest surviving lane = next kept box, broadcast by readlane,
//              later lanes test against it), the other waves then test against the few boxes
//              that sub-chunk appended.
//   The result is the greedy keep list of the reference for ANY evaluation order, because
//   IoU(i, j) uses the reference's exact fp32 expression:
//       w = max(min(ax2,bx2) - max(ax1,bx1) + 1, 0); inter = w*h;
//       ovr = inter / (Sa + Sb - inter)           (IEEE division, no FMA contraction)
//   The division is only executed when a guarded reciprocal estimate is within 1e-5 of the
//   threshold (a few ulp), which never changes the decision.
//
// Compiled with -ffp-contract=off.
#include "ct_common.h"
#include <algorithm>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int kKeptLds = 2048;   // kept boxes held in LDS; further ones are re-read from HBM/L2
constexpr int kThreads = 256;

struct Thr {
    float t, lo, hi;
};

__host__ inline Thr make_thr(float t)
{
    Thr r;
    r.t = t;
    if (t > 1e-30f && t < 1e30f) {
        r.lo = t * (1.f - 1e-5f);
        r.hi = t * (1.f + 1e-5f);
    } else {             // degenerate thresholds: always take the exact path
        r.lo = -INFINITY;
        r.hi = INFINITY;
    }
    return r;
}

// v_max_f32 / v_min_f32 without the canonicalising self-max the compiler puts in front of
// fmaxf/fminf on loaded values (inputs are finite; result for finite inputs is identical).
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Branch-free estimate of "the reference suppresses b given kept box a" (a = higher score):
//   returns 1 = certainly yes, 0 = certainly no, 2 = within 1e-5 of the threshold (decide exactly)
// PLAIN: no +1 and union = (Sb - inter) + Sa, the expression of utils/box_utils.py:288-299.
template <bool PLAIN>
__device__ __forceinline__ int overlap_class(const float4 a, const float sa, const float bx1,
                                             const float by1, const float bx2, const float by2,
                                             const float sb, const Thr th, float& inter, float& uni)
{
    const float left = vmax(a.x, bx1), right = vmin(a.z, bx2);
    const float top = vmax(a.y, by1), bottom = vmin(a.w, by2);
    const float w = PLAIN ? vmax(right - left, 0.f) : vmax(right - left + 1.f, 0.f);
    const float h = PLAIN ? vmax(bottom - top, 0.f) : vmax(bottom - top + 1.f, 0.f);
    inter = w * h;
    uni = PLAIN ? (sb - inter) + sa : sa + sb - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    return q > th.hi ? 1 : (q < th.lo ? 0 : 2);
}

// reciprocal-estimate of inter/union only (inter, uni returned for the exact path)
template <bool PLAIN>
__device__ __forceinline__ float ratio(const float4 a, const float sa, const float bx1, const float by1,
                                       const float bx2, const float by2, const float sb, float& inter,
                                       float& uni)
{
    const float left = vmax(a.x, bx1), right = vmin(a.z, bx2);
    const float top = vmax(a.y, by1), bottom = vmin(a.w, by2);
    const float w = PLAIN ? vmax(right - left, 0.f) : vmax(right - left + 1.f, 0.f);
    const float h = PLAIN ? vmax(bottom - top, 0.f) : vmax(bottom - top + 1.f, 0.f);
    inter = w * h;
    uni = PLAIN ? (sb - inter) + sa : sa + sb - inter;
    return inter * __builtin_amdgcn_rcpf(uni);
}

template <bool GE>
__device__ __forceinline__ bool exact_rule(float inter, float uni, const Thr th)
{
    const float ovr = inter / uni;                 // IEEE quotient, only for borderline pairs
    return GE ? (ovr >= th.t) : (ovr > th.t);
}

// value of lane `i` (wave-uniform i) in every lane
__device__ __forceinline__ float bcast(float v, int i)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}

template <bool GE, bool PLAIN>
__global__ __launch_bounds__(kThreads) void nms_segments_kernel(const float* __restrict__ dets,
                                                                const int* __restrict__ seg_off,
                                                                const int* __restrict__ seg_len,
                                                                const int seg_stride,
                                                                const Thr th, int* __restrict__ keep,
                                                                int* __restrict__ keep_count)
{
    __shared__ float4 kbox[kKeptLds];
    __shared__ __attribute__((aligned(16))) float karea[kKeptLds];
    __shared__ int s_kept[kThreads / 64];   // one slot per sub-chunk (no reuse within 3 barriers)

    const int seg = blockIdx.x;
    // segments are either CSR (seg_off) or fixed-stride slots with a length array
    const int base = seg_len ? seg * seg_stride : seg_off[seg];
    const int n = seg_len ? seg_len[seg] : seg_off[seg + 1] - base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* d = dets + (size_t)base * 5;
    int* kp = keep + base;

    int kept = 0;   // uniform across the workgroup at chunk boundaries
    for (int c0 = 0; c0 < n; c0 += kThreads) {
        const int j = c0 + tid;
        const bool valid = j < n;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        if (valid) {
            const float* r = d + (size_t)j * 5;
            x1 = r[0]; y1 = r[1]; x2 = r[2]; y2 = r[3];
        }
        const float sj = PLAIN ? (x2 - x1) * (y2 - y1) : (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
        bool alive = valid;

        // test this lane's candidate against kept boxes [i0, i1): LDS part 4 at a time with all
        // eight ds_reads in flight together, the (rare) part beyond the LDS window from L2
        auto test_range = [&](int i0, int i1) {
            const int lds_end = min(i1, kKeptLds);
            int i = i0;
            for (; i < lds_end && (i & 3); ++i) {
                float inter, uni;
                const int c = overlap_class<PLAIN>(kbox[i], karea[i], x1, y1, x2, y2, sj, th, inter, uni);
                if (c == 1 || (c == 2 && exact_rule<GE>(inter, uni, th))) alive = false;
            }
            for (; i + 4 <= lds_end; i += 4) {
                if (!__any(alive)) return;
                const float4 b0 = kbox[i], b1 = kbox[i + 1], b2 = kbox[i + 2], b3 = kbox[i + 3];
                const float4 ar = *reinterpret_cast<const float4*>(&karea[i]);
                float n0, u0, n1, u1, n2, u2, n3, u3;
                const float q0 = ratio<PLAIN>(b0, ar.x, x1, y1, x2, y2, sj, n0, u0);
                const float q1 = ratio<PLAIN>(b1, ar.y, x1, y1, x2, y2, sj, n1, u1);
                const float q2 = ratio<PLAIN>(b2, ar.z, x1, y1, x2, y2, sj, n2, u2);
                const float q3 = ratio<PLAIN>(b3, ar.w, x1, y1, x2, y2, sj, n3, u3);
                const bool yes = (q0 > th.hi) | (q1 > th.hi) | (q2 > th.hi) | (q3 > th.hi);
                const bool no = (q0 < th.lo) & (q1 < th.lo) & (q2 < th.lo) & (q3 < th.lo);
                if (yes) alive = false;
                if (__any(!yes && !no)) {                  // some pair within 1e-5 of the threshold
                    const bool m0 = !(q0 > th.hi) && !(q0 < th.lo), m1 = !(q1 > th.hi) && !(q1 < th.lo);
                    const bool m2 = !(q2 > th.hi) && !(q2 < th.lo), m3 = !(q3 > th.hi) && !(q3 < th.lo);
                    if ((m0 && exact_rule<GE>(n0, u0, th)) || (m1 && exact_rule<GE>(n1, u1, th)) ||
                        (m2 && exact_rule<GE>(n2, u2, th)) || (m3 && exact_rule<GE>(n3, u3, th)))
                        alive = false;
                }
            }
            for (; i < lds_end; ++i) {
                float inter, uni;
                const int c = overlap_class<PLAIN>(kbox[i], karea[i], x1, y1, x2, y2, sj, th, inter, uni);
                if (c == 1 || (c == 2 && exact_rule<GE>(inter, uni, th))) alive = false;
            }
            for (i = max(i, kKeptLds); i < i1; ++i) {
                const float* r = d + (size_t)kp[i] * 5;
                const float4 b = make_float4(r[0], r[1], r[2], r[3]);
                const float s = PLAIN ? (b.z - b.x) * (b.w - b.y) : (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
                float inter, uni;
                const int c = overlap_class<PLAIN>(b, s, x1, y1, x2, y2, sj, th, inter, uni);
                if (c == 1 || (c == 2 && exact_rule<GE>(inter, uni, th))) alive = false;
            }
        };

        // ---- phase A: against everything kept before this chunk ----
        test_range(0, kept);

        // ---- phase B: resolve the four 64-candidate sub-chunks in order ----
        int kstart = kept;
#pragma unroll 1
        for (int w = 0; w < kThreads / 64; ++w) {
            if (c0 + w * 64 >= n) break;                    // uniform
            if (wave == w) {
                int k = kstart;
                unsigned long long m = __ballot(alive);
                while (m) {
                    const int i = __builtin_ctzll(m);       // highest-scoring survivor
                    const float4 bb = make_float4(bcast(x1, i), bcast(y1, i), bcast(x2, i), bcast(y2, i));
                    const float bs = bcast(sj, i);
                    if (lane == i) {
                        if (k < kKeptLds) {
                            kbox[k] = make_float4(x1, y1, x2, y2);
                            karea[k] = sj;
                        }
                        kp[k] = j;
                    }
                    ++k;
                    float inter, uni;
                    const int c = overlap_class<PLAIN>(bb, bs, x1, y1, x2, y2, sj, th, inter, uni);
                    if (lane > i && (c == 1 || (c == 2 && exact_rule<GE>(inter, uni, th)))) alive = false;
                    m = __ballot(alive) & ~((2ull << i) - 1ull);
                }
                if (lane == 0) s_kept[w] = k;
            }
            __syncthreads();
            const int know = s_kept[w];
            if (wave > w) test_range(kstart, know);
            kstart = know;
        }
        kept = kstart;
    }
    if (tid == 0) keep_count[seg] = kept;
}

int launch_nms(const float* dets, const int* seg_off, const int* seg_len, int seg_stride, int nseg,
               float thresh, int ge, int* keep, int* keep_count, hipStream_t st)
{
    const Thr th = make_thr(thresh);
    const dim3 grid(nseg), block(kThreads);
#define CT_NMS_LAUNCH(GE, PL)                                                                  \
    hipLaunchKernelGGL((nms_segments_kernel<GE, PL>), grid, block, 0, st, dets, seg_off, seg_len, \
                       seg_stride, th, keep, keep_count)
    CT_PROF("nms_segments_kernel", st);
    switch (ge & 3) {            // bit0: >=, bit1: plain IoU
        case 0: CT_NMS_LAUNCH(false, false); break;
        case 1: CT_NMS_LAUNCH(true, false); break;
        case 2: CT_NMS_LAUNCH(false, true); break;
        default: CT_NMS_LAUNCH(true, true); break;
    }
#undef CT_NMS_LAUNCH
    CT_LAUNCH_CHECK("nms_segments_kernel");
    return CT_OK;
}

}  // namespace

namespace ctdet {
// fixed-stride segments (slot s = rows [s*stride, s*stride + seg_len[s])); used by ct_post.hip
int nms_launch_strided(const float* dets, const int* seg_len, int seg_stride, int nseg, float thresh,
                       int ge, int* keep, int* keep_count, hipStream_t st)
{
    return launch_nms(dets, nullptr, seg_len, seg_stride, nseg, thresh, ge, keep, keep_count, st);
}
}  // namespace ctdet

extern "C" size_t ct_nms_batched_workspace_bytes(int, int) { return 256; }

extern "C" int ct_nms_batched_dev(const float* dets, const int* seg_off, int num_segments,
                                  int max_seg_len, float thresh, int ge, int* keep, int* keep_count,
                                  void*, size_t, ct_stream_t stream)
{
    CT_REQUIRE(dets && seg_off && keep && keep_count, "ct_nms_batched_dev: null pointer");
    CT_REQUIRE(num_segments > 0 && max_seg_len >= 0, "ct_nms_batched_dev: bad sizes");
    return launch_nms(dets, seg_off, nullptr, 0, num_segments, thresh, ge, keep, keep_count,
                      ctdet::as_stream(stream));
}

extern "C" int ct_nms_sorted_host_mode(int* keep_out, int* num_out, const float* boxes_host,
                                       int boxes_num, int boxes_dim, float thresh, int ge, int device_id)
{
    CT_REQUIRE(keep_out && num_out && (boxes_host || boxes_num == 0), "ct_nms_sorted_host: null pointer");
    CT_REQUIRE(boxes_num >= 0 && boxes_dim >= 4, "ct_nms_sorted_host: boxes_num=%d boxes_dim=%d", boxes_num, boxes_dim);
    *num_out = 0;
    if (boxes_num == 0) return CT_OK;
    int prev = -1;
    CT_HIP(hipGetDevice(&prev));
    if (prev != device_id) CT_HIP(hipSetDevice(device_id));
    // One device buffer per thread and device, grown on demand and kept between calls (test.py:152 calls this 20 times
    // per image; a hipMalloc / hipFree pair per call cost more than the kernel), one H2D and one D2H copy per call:
    //   [meta 256 B: seg_off[2], count][keep n ints][rows n x (x1,y1,x2,y2,score)]
    struct Scratch { int device = -1; char* ptr = nullptr; size_t bytes = 0; };
    static thread_local Scratch scratch;
    const size_t keep_bytes = ctdet::align_up((size_t)boxes_num * 4, 256);
    const size_t det_bytes = ctdet::align_up((size_t)boxes_num * 5 * 4, 256);
    const size_t need = 256 + keep_bytes + det_bytes;
    int rc = CT_OK;
    hipError_t e = hipSuccess;
    if (scratch.device != device_id || scratch.bytes < need) {
        if (scratch.ptr) {
            int cur = device_id;
            if (scratch.device != device_id) (void)hipSetDevice(scratch.device);
            (void)hipFree(scratch.ptr);
            if (scratch.device != cur) (void)hipSetDevice(cur);
            scratch = Scratch{};
        }
        const size_t want = std::max(need, (size_t)1 << 20);
        e = hipMalloc((void**)&scratch.ptr, want);
        if (e == hipSuccess) { scratch.device = device_id; scratch.bytes = want; }
        else scratch = Scratch{};
    }
    if (e != hipSuccess) {
        rc = ctdet::fail(CT_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    } else {
        std::vector<char> host(need);
        int* meta = reinterpret_cast<int*>(host.data());
        meta[0] = 0; meta[1] = boxes_num; meta[2] = 0;
        float* rows = reinterpret_cast<float*>(host.data() + 256 + keep_bytes);
        for (int i = 0; i < boxes_num; ++i) {       // rows -> [x1,y1,x2,y2,score] (the kernel does not read the score)
            const float* r = boxes_host + (size_t)i * boxes_dim;
            float* o = rows + (size_t)i * 5;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
            o[4] = boxes_dim > 4 ? r[4] : 0.f;
        }
        char* dev = scratch.ptr;
        int* d_meta = reinterpret_cast<int*>(dev);
        int* d_keep = reinterpret_cast<int*>(dev + 256);
        float* d_dets = reinterpret_cast<float*>(dev + 256 + keep_bytes);
        e = hipMemcpy(dev, host.data(), need, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = launch_nms(d_dets, d_meta, nullptr, 0, 1, thresh, ge, d_keep, d_meta + 2, nullptr);
            if (rc == CT_OK) {
                e = hipMemcpy(host.data(), dev, 256 + (size_t)boxes_num * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) {
                    const int cnt = meta[2];
                    if (cnt > 0) memcpy(keep_out, host.data() + 256, (size_t)cnt * 4);
                    *num_out = cnt;
                }
            }
        }
        if (e != hipSuccess) rc = ctdet::fail(CT_ERR_HIP, "ct_nms_sorted_host: %s", hipGetErrorString(e));
    }
    if (prev != device_id) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int ct_nms_sorted_host(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
                                  int boxes_dim, float nms_overlap_thresh, int device_id)
{
    return ct_nms_sorted_host_mode(keep_out, num_out, boxes_host, boxes_num, boxes_dim,
                                   nms_overlap_thresh, 0, device_id);
}
