// libctdet: batched form of the reference's per-image / per-class post-processing loop
// (test.py:136-161): score threshold -> descending-score order -> NMS -> per-image top-k.
//
//   select_sort_kernel  one workgroup per (image, class): collects priors with score > thresh
//                       as 64-bit keys (~score_bits << 32 | prior_index), sorts them ascending
//                       (= descending score, lower index first on ties -- the build's defined
//                       tie order) with a bitonic network that runs in LDS for strides < 4096
//                       and through L2 for the few larger strides, then writes the sorted
//                       [x1,y1,x2,y2,score] rows.  With the top-k rule on it sorts only the best >= 1024
//                       candidates of a class (partial sort, see the kernel); prefix_cut_kernel flags the
//                       segments whose cut lies further down and a second launch sorts those in full.
//   nms_segments_kernel (ct_nms.hip) on the fixed-stride segments.
//   topk_kernel         one workgroup per image: if more than max_per_image boxes survive,
//                       8-bit radix select of the k-th largest score; because every segment is
//                       in descending order the survivors `score >= thresh_k` are a prefix.
//   gather_kernel       copies the surviving rows to the caller's [B,T,cap,5] buffer.
// Nothing synchronises with the host; counts / overflow flag stay on the device.
#include "ct_common.h"
#include <algorithm>

#pragma clang fp contract(off)

namespace ctdet {
int nms_launch_strided(const float* dets, const int* seg_len, int seg_stride, int nseg, float thresh,
                       int ge, int* keep, int* keep_count, hipStream_t st);
}

namespace {

typedef unsigned long long u64;
constexpr int kSortThreads = 512;
constexpr int kSelectBatch = 8;      // strided score loads in flight per thread in the select loop
// keys a workgroup sorts in LDS (template parameter LK of select_sort_kernel): 4096 = 32 KB; 8192 = 64 KB for RFBNet-512's 32 756
// priors, whose per-class candidate counts (~8 000 with random weights) otherwise take the bitonic network's long strides
// through L2 (470 us of the 512 x 512 bs-32 step)

__device__ __forceinline__ void cmpswap(u64& a, u64& b, bool asc)
{
    if ((a > b) == asc) {
        const u64 t = a;
        a = b;
        b = t;
    }
}

// all sub-stages j = jstart .. 1 of bitonic stage k on one LDS chunk whose first key has
// global index gbase
__device__ __forceinline__ void lds_substages(u64* sk, int nloc, int k, int jstart, int gbase)
{
    for (int j = jstart; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < nloc / 2; t += kSortThreads) {
            const int i = 2 * t - (t & (j - 1));
            const bool asc = ((gbase + i) & k) == 0;
            u64 a = sk[i], b = sk[i + j];
            cmpswap(a, b, asc);
            sk[i] = a;
            sk[i + j] = b;
        }
        __syncthreads();
    }
}

// Partial sort (the top-k rule's bounding pass only ever looks at a prefix): with max_per_image > 0 a workgroup sorts only
// its best m >= min(n, kPartialMin) candidates -- every candidate at or above a cut score taken from an LDS histogram of the
// score bits (11 bits: exponent + 3 mantissa bits; 11 more inside the crossing bin if that bin alone overflows the LDS sort
// buffer).  seg_sorted[seg] = m rows are written, seg_count[seg] = n stays the true candidate count, the unsorted keys
// stay in keys_ws.  prefix_cut_kernel flags a segment whose cut falls behind its sorted prefix (redo[seg] = 1); the SAME
// kernel launched again with `redo` sorts exactly those in full.  Bit-exact either way: which rows exist never changes
// their order or values.  (Round 5: the full 4096 / 8192 / 16384-key bitonic networks were 122 us of the 300 and 470 us
// of the 512 pipeline with random weights; VERDICT r04 task 7.)
constexpr int kPartialMin = 1024;
constexpr int kHistBins = 2048;

// inclusive scan over the workgroup (kSortThreads threads); s_w: 8 ints of scratch
__device__ __forceinline__ int block_incl_scan(int v, int* s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    __syncthreads();
    return x + base;
}

// bins in DESCENDING order (d = 0 is the highest bin): the first d whose cumulative count reaches `need`.
// -> s_out[0] = bin, s_out[1] = cumulative count through that bin, s_out[2] = count above it.  Every bin count summed must
// be >= need.
__device__ __forceinline__ void hist_crossing(const int* s_hist, int need, int* s_w, int* s_out)
{
    constexpr int PER = kHistBins / kSortThreads;
    int c[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        c[u] = s_hist[kHistBins - 1 - (PER * (int)threadIdx.x + u)];
        sum += c[u];
    }
    const int incl = block_incl_scan(sum, s_w);
    int run = incl - sum;
    if (run < need && incl >= need) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (run < need && run + c[u] >= need) {
                s_out[0] = kHistBins - 1 - (PER * (int)threadIdx.x + u);
                s_out[1] = run + c[u];
                s_out[2] = run;
            }
            run += c[u];
        }
    }
    __syncthreads();
}

template <int kLdsKeys, bool PARTIAL>
__global__ __launch_bounds__(kSortThreads) void select_sort_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, int batch, int P, int T, float thresh,
    int npow2_cap, u64* __restrict__ keys_ws, float* __restrict__ dets_sorted,
    int* __restrict__ sorted_idx, int* __restrict__ seg_count, int* __restrict__ seg_sorted, const int* __restrict__ redo)
{
    extern __shared__ __attribute__((aligned(16))) u64 sk_dyn[];      // kLdsKeys keys, [histogram,] scalars
    u64* const sk = sk_dyn;
    int* const s_hist = reinterpret_cast<int*>(sk_dyn + kLdsKeys);
    int* const s_sc = s_hist + (PARTIAL ? kHistBins : 0);             // [0] counter [1..3] crossing [4] counter 2; [8..15] scan
    int& s_cnt = s_sc[0];
    // blockIdx -> (image, class) so that all classes of an image run on ONE XCD (block q is dispatched to XCD q % 8):
    // a class column of scores[b][P][T+1] is a strided read that touches every cache line of the image's score
    // array, so the T workgroups of an image share those lines through one L2 instead of fetching the array once
    // per XCD (round 2 PMC: 432 MB fetched per launch against 81 MB algorithmic)
    const int b = ((int)(blockIdx.x >> 3) / T) * 8 + (int)(blockIdx.x & 7);
    const int cls = 1 + (int)(blockIdx.x >> 3) % T;
    if (b >= batch) return;
    const int seg = b * T + (cls - 1);
    if (redo && !redo[seg]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    u64* keys = keys_ws + (size_t)seg * npow2_cap;
    if (tid == 0) { s_cnt = 0; s_sc[4] = 0; }
    if (PARTIAL)
        for (int i = tid; i < kHistBins; i += kSortThreads) s_hist[i] = 0;
    __syncthreads();

    // ---- select (wave-aggregated append; order is irrelevant, the keys are unique) ----
    const float* sc = scores + (size_t)b * P * (T + 1) + cls;
    // kSelectBatch independent loads per thread in flight before the first ballot
    for (int p0 = 0; p0 < P; p0 += kSortThreads * kSelectBatch) {
        float v[kSelectBatch];
#pragma unroll
        for (int u = 0; u < kSelectBatch; ++u) {
            const int p = p0 + u * kSortThreads + tid;
            v[u] = p < P ? sc[(size_t)p * (T + 1)] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < kSelectBatch; ++u) {
            const int p = p0 + u * kSortThreads + tid;
            const bool pass = v[u] > thresh;                   // padding lanes: -inf never passes
            const u64 m = __ballot(pass);
            int basepos = 0;
            if (lane == 0 && m) basepos = atomicAdd(&s_cnt, __popcll(m));
            basepos = __shfl(basepos, 0);
            if (pass) {
                const int pos = basepos + __popcll(m & ((1ull << lane) - 1ull));
                keys[pos] = ((u64)(~__float_as_uint(v[u])) << 32) | (unsigned)p;
                if (PARTIAL) atomicAdd(&s_hist[__float_as_uint(v[u]) >> 20], 1);
            }
        }
    }
    __syncthreads();
    const int n = s_cnt;
    if (tid == 0) seg_count[seg] = n;
    if (n == 0) {
        if (tid == 0) seg_sorted[seg] = 0;
        return;
    }

    const float4* bx = reinterpret_cast<const float4*>(boxes) + (size_t)b * P;
    float* out = dets_sorted + (size_t)seg * P * 5;
    int* oi = sorted_idx + (size_t)seg * P;
    auto write_row = [&](int i, u64 key) {
        const int p = (int)(key & 0xFFFFFFFFull);
        const float v = __uint_as_float(~(unsigned)(key >> 32));
        const float4 q = bx[p];
        float* r = out + (size_t)i * 5;
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; r[4] = v;
        oi[i] = p;
    };

    if (PARTIAL) {
        // ---- the cut: every candidate whose score bits are >= cutbits is sorted ----
        unsigned cutbits = 0;                       // n <= kPartialMin: everything
        int m = n;
        bool partial_ok = n <= kLdsKeys;
        if (n > kPartialMin) {
            __threadfence_block();
            hist_crossing(s_hist, kPartialMin, s_sc + 8, s_sc + 1);
            const int bin = s_sc[1], above = s_sc[3];
            m = s_sc[2];
            cutbits = (unsigned)bin << 20;
            partial_ok = true;
            __syncthreads();
            if (m > kLdsKeys) {
                // the crossing bin alone is too large: 11 more bits inside it
                for (int i = tid; i < kHistBins; i += kSortThreads) s_hist[i] = 0;
                __syncthreads();
                for (int i = tid; i < n; i += kSortThreads) {
                    const unsigned sb = ~(unsigned)(keys[i] >> 32);
                    if ((int)(sb >> 20) == bin) atomicAdd(&s_hist[(sb >> 9) & (kHistBins - 1)], 1);
                }
                __syncthreads();
                hist_crossing(s_hist, kPartialMin - above, s_sc + 8, s_sc + 1);
                cutbits = ((unsigned)bin << 20) | ((unsigned)s_sc[1] << 9);
                m = above + s_sc[2];
                partial_ok = m <= kLdsKeys;
                __syncthreads();
            }
        }
        if (partial_ok) {
            if (m < n) {
                for (int i0 = 0; i0 < n; i0 += kSortThreads) {
                    const int i = i0 + tid;
                    u64 key = 0;
                    bool pass = false;
                    if (i < n) {
                        key = keys[i];
                        pass = ~(unsigned)(key >> 32) >= cutbits;
                    }
                    const u64 bm = __ballot(pass);
                    int basepos = 0;
                    if (lane == 0 && bm) basepos = atomicAdd(&s_sc[4], __popcll(bm));
                    basepos = __shfl(basepos, 0);
                    if (pass) sk[basepos + __popcll(bm & ((1ull << lane) - 1ull))] = key;
                }
            } else {
                for (int i = tid; i < n; i += kSortThreads) sk[i] = keys[i];
            }
            int N = 1;
            while (N < m) N <<= 1;
            for (int i = m + tid; i < N; i += kSortThreads) sk[i] = ~0ull;
            __syncthreads();
            for (int k = 2; k <= N; k <<= 1) lds_substages(sk, N, k, k >> 1, 0);
            for (int i = tid; i < m; i += kSortThreads) write_row(i, sk[i]);
            if (tid == 0) seg_sorted[seg] = m;
            return;
        }
        // (too many equal score bits around the cut: the full sort below)
    }

    int N = 1;
    while (N < n) N <<= 1;
    for (int i = n + tid; i < N; i += kSortThreads) keys[i] = ~0ull;
    __syncthreads();

    // ---- bitonic sort of keys[0..N) ----
    if (N <= kLdsKeys) {
        for (int i = tid; i < N; i += kSortThreads) sk[i] = keys[i];
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1) lds_substages(sk, N, k, k >> 1, 0);
        for (int i = tid; i < N; i += kSortThreads) keys[i] = sk[i];
    } else {
        for (int c = 0; c < N; c += kLdsKeys) {            // sort every chunk
            for (int i = tid; i < kLdsKeys; i += kSortThreads) sk[i] = keys[c + i];
            __syncthreads();
            for (int k = 2; k <= kLdsKeys; k <<= 1) lds_substages(sk, kLdsKeys, k, k >> 1, c);
            for (int i = tid; i < kLdsKeys; i += kSortThreads) keys[c + i] = sk[i];
            __syncthreads();
        }
        for (int k = 2 * kLdsKeys; k <= N; k <<= 1) {
            for (int j = k >> 1; j >= kLdsKeys; j >>= 1) {  // long strides through L2
                for (int t = tid; t < N / 2; t += kSortThreads) {
                    const int i = 2 * t - (t & (j - 1));
                    u64 a = keys[i], bb = keys[i + j];
                    cmpswap(a, bb, (i & k) == 0);
                    keys[i] = a;
                    keys[i + j] = bb;
                }
                __syncthreads();
            }
            for (int c = 0; c < N; c += kLdsKeys) {
                for (int i = tid; i < kLdsKeys; i += kSortThreads) sk[i] = keys[c + i];
                __syncthreads();
                lds_substages(sk, kLdsKeys, k, kLdsKeys >> 1, c);
                for (int i = tid; i < kLdsKeys; i += kSortThreads) keys[c + i] = sk[i];
                __syncthreads();
            }
        }
    }
    __syncthreads();

    // ---- sorted rows ----
    for (int i = tid; i < n; i += kSortThreads) write_row(i, keys[i]);
    if (tid == 0) seg_sorted[seg] = n;
}

// k-th largest kept score of image b (as uint bits; all scores are positive floats), or 0 when
// at most k boxes are kept.  256 threads; hist/s_* are workgroup scratch.
constexpr int kStageCap = 8192;      // kept scores of one image staged in LDS for the radix select (32 KB)
constexpr int kStageMaxT = 128;

__device__ unsigned kth_largest_kept(const float* __restrict__ dets_sorted, const int* __restrict__ keep,
                                     const int* __restrict__ keep_count, int P, int T, int b, int k,
                                     int* hist, int* s_scalars)
{
    __shared__ unsigned s_scores[kStageCap];
    __shared__ int s_off[kStageMaxT + 1];
    __shared__ int s_suf[256];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int t = 0;
        for (int c = 0; c < T; ++c) {
            if (c <= kStageMaxT) s_off[c] = t;
            t += keep_count[b * T + c];
        }
        if (T <= kStageMaxT) s_off[T] = t;
        s_scalars[0] = t;
    }
    __syncthreads();
    const int total = s_scalars[0];
    __syncthreads();
    if (k <= 0 || total <= k) return 0u;
    // The four radix passes each walk every kept score through two dependent global loads (index, then score);
    // when the image's kept scores fit in LDS they are fetched once, all classes in parallel, and the passes
    // run out of LDS.
    const bool staged = total <= kStageCap && T <= kStageMaxT;
    if (staged) {
        // four elements per thread and trip: the two dependent loads (kept index, then its score) of all four are
        // in flight together
        for (int e0 = tid; e0 < total; e0 += 4 * 256) {
            int idx[4];
            size_t row[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + 256 * u, total - 1);
                int lo = 0, hi = T;                   // class of element e: last c with s_off[c] <= e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_off[mid] <= e) lo = mid; else hi = mid;
                }
                row[u] = (size_t)(b * T + lo) * P;
                idx[u] = keep[row[u] + (e - s_off[lo])];
            }
            unsigned sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sc[u] = __float_as_uint(dets_sorted[(row[u] + idx[u]) * 5 + 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + 256 * u < total) s_scores[e0 + 256 * u] = sc[u];
        }
        __syncthreads();
    }
    unsigned prefix = 0, pmask = 0;
    int krem = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        if (staged) {
            for (int e = tid; e < total; e += 256) {
                const unsigned u = s_scores[e];
                if ((u & pmask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
            }
        } else {
            for (int c = 0; c < T; ++c) {
                const int seg = b * T + c, kc = keep_count[seg];
                const float* d = dets_sorted + (size_t)seg * P * 5;
                const int* kp = keep + (size_t)seg * P;
                for (int i = tid; i < kc; i += 256) {
                    const unsigned u = __float_as_uint(d[(size_t)kp[i] * 5 + 4]);
                    if ((u & pmask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
                }
            }
        }
        __syncthreads();
        // bin of the krem-th largest: the highest x whose suffix count sum_{i >= x} hist[i] reaches krem.  A parallel
        // suffix scan -- one thread walking 256 LDS bins serially cost ~10 us per radix pass.
        {
            const int v = hist[tid];
            s_suf[tid] = v;
            __syncthreads();
            for (int off = 1; off < 256; off <<= 1) {
                const int add = (tid + off < 256) ? s_suf[tid + off] : 0;
                __syncthreads();
                s_suf[tid] += add;
                __syncthreads();
            }
            const int incl = s_suf[tid], excl = incl - v;
            if (excl < krem && incl >= krem) {
                s_scalars[1] = tid;
                s_scalars[2] = krem - excl;
            }
        }
        __syncthreads();
        prefix |= (unsigned)s_scalars[1] << shift;
        pmask |= 255u << shift;
        krem = s_scalars[2];
        __syncthreads();
    }
    return prefix;
}

// Exact early cut for the top-k rule.  NMS is prefix-consistent (whether a box is kept depends
// only on higher-scoring boxes), so after NMS of every class's first `prefix_len` candidates the
// k-th largest kept score t_est of the image is a LOWER bound of the final threshold: no candidate
// below t_est can appear in the output, and the kept set above it is determined by the candidates
// above it alone.  This kernel shortens every segment to its candidates with score >= t_est.
// first candidate of a descending segment with score bits < thr: 256-ary search by the whole block (256 threads), two or
// three dependent loads instead of the 14 of a per-thread bisection over 11 620 candidates
__device__ __forceinline__ int cut_search(const float* __restrict__ d, int hi, unsigned thr)
{
    int lo = 0;
    while (hi - lo > 256) {
        const int step = (hi - lo + 255) / 256;
        const int p = lo + (int)threadIdx.x * step;
        const int cnt = __syncthreads_count(p < hi && __float_as_uint(d[(size_t)p * 5 + 4]) >= thr);
        if (cnt == 0) { hi = lo; break; }
        const int nlo = lo + (cnt - 1) * step + 1;
        hi = min(hi, lo + cnt * step);
        lo = nlo;
    }
    const int p = lo + (int)threadIdx.x;
    const int cnt = __syncthreads_count(p < hi && __float_as_uint(d[(size_t)p * 5 + 4]) >= thr);
    return lo + cnt;
}

// seg_sorted: rows select_sort_kernel wrote (a partial sort's prefix); a segment whose every sorted row passes while
// unsorted candidates remain is flagged in redo[] (its cut lies further down): it is sorted in full and cut again
// (recut_kernel) with the image's threshold kept in thr_img[].
__global__ __launch_bounds__(256) void prefix_cut_kernel(const float* __restrict__ dets_sorted,
                                                         const int* __restrict__ keep,
                                                         const int* __restrict__ keep_count,
                                                         const int* __restrict__ seg_count,
                                                         const int* __restrict__ seg_sorted, int P, int T,
                                                         int max_per_image, int* __restrict__ seg_len_out,
                                                         int* __restrict__ redo, unsigned* __restrict__ thr_img)
{
    __shared__ int hist[256];
    __shared__ int s_scalars[4];
    const int b = blockIdx.x, c = blockIdx.y;
    const unsigned thr = kth_largest_kept(dets_sorted, keep, keep_count, P, T, b, max_per_image, hist, s_scalars);
    const int seg = b * T + c;
    const int m = seg_sorted[seg];
    const int len = cut_search(dets_sorted + (size_t)seg * P * 5, m, thr);
    if (threadIdx.x == 0) {
        seg_len_out[seg] = len;
        redo[seg] = (len == m && m < seg_count[seg]) ? 1 : 0;
        if (c == 0) thr_img[b] = thr;
    }
}

__global__ __launch_bounds__(256) void recut_kernel(const float* __restrict__ dets_sorted, const int* __restrict__ seg_sorted,
                                                    const int* __restrict__ redo, const unsigned* __restrict__ thr_img,
                                                    int P, int T, int* __restrict__ seg_len_out)
{
    const int seg = blockIdx.x;
    if (!redo[seg]) return;
    const int len = cut_search(dets_sorted + (size_t)seg * P * 5, seg_sorted[seg], thr_img[seg / T]);
    if (threadIdx.x == 0) seg_len_out[seg] = len;
}

__global__ void clamp_len_kernel(const int* __restrict__ in, int n, int cap, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = min(in[i], cap);
}

// per image: the `>= k-th largest score` rule of test.py:155-161
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ dets_sorted,
                                                   const int* __restrict__ keep,
                                                   const int* __restrict__ keep_count, int P, int T,
                                                   int max_per_image, int out_cap,
                                                   int* __restrict__ out_count, int* __restrict__ overflow)
{
    __shared__ int hist[256];
    __shared__ int s_scalars[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned thr_bits = kth_largest_kept(dets_sorted, keep, keep_count, P, T, b, max_per_image, hist, s_scalars);
    // survivors of every class are a prefix of its descending kept list
    for (int c = tid; c < T; c += 256) {
        const int seg = b * T + c, kc = keep_count[seg];
        const float* d = dets_sorted + (size_t)seg * P * 5;
        const int* kp = keep + (size_t)seg * P;
        int lo = 0, hi = kc;               // first index with score < thr
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (__float_as_uint(d[(size_t)kp[mid] * 5 + 4]) >= thr_bits) lo = mid + 1;
            else hi = mid;
        }
        if (lo > out_cap) {
            atomicExch(overflow, 1);
            lo = out_cap;
        }
        out_count[seg] = lo;
    }
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ dets_sorted,
                                                     const int* __restrict__ sorted_idx,
                                                     const int* __restrict__ keep,
                                                     const int* __restrict__ out_count, int P,
                                                     int out_cap, float* __restrict__ out_dets,
                                                     int* __restrict__ out_index)
{
    const int seg = blockIdx.x;
    const int n = out_count[seg];
    const float* d = dets_sorted + (size_t)seg * P * 5;
    const int* kp = keep + (size_t)seg * P;
    for (int e = threadIdx.x; e < n * 5; e += blockDim.x) {
        const int i = e / 5, c = e - i * 5;
        out_dets[((size_t)seg * out_cap + i) * 5 + c] = d[(size_t)kp[i] * 5 + c];
    }
    if (out_index)
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            out_index[(size_t)seg * out_cap + i] = sorted_idx[(size_t)seg * P + kp[i]];
}

int next_pow2(int n)
{
    int v = 1;
    while (v < n) v <<= 1;
    return v;
}

struct PostWs {
    u64* keys;
    float* dets_sorted;
    int* sorted_idx;
    int* keep;
    int* seg_count;
    int* keep_count;
    int* seg_len;      // working segment lengths (prefix pass / cut pass)
    int* seg_sorted;   // rows written per segment (partial sort: a prefix of the candidates)
    int* redo;         // segments whose cut fell behind their sorted prefix
    unsigned* thr_img; // per image: the bounding pass's threshold
    size_t total;
};

PostWs carve(char* base, int batch, int P, int T)
{
    const size_t S = (size_t)batch * T;
    const int np2 = std::max(next_pow2(P), 2);
    PostWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += ctdet::align_up(bytes, 256);
        return p;
    };
    w.keys = (u64*)take(S * np2 * 8);
    w.dets_sorted = (float*)take(S * P * 5 * 4);
    w.sorted_idx = (int*)take(S * P * 4);
    w.keep = (int*)take(S * P * 4);
    w.seg_count = (int*)take(S * 4);
    w.keep_count = (int*)take(S * 4);
    w.seg_len = (int*)take(S * 4);
    w.seg_sorted = (int*)take(S * 4);
    w.redo = (int*)take(S * 4);
    w.thr_img = (unsigned*)take((size_t)batch * 4);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t ct_postprocess_workspace_bytes(int batch, int num_priors, int num_fg)
{
    return carve(nullptr, batch, num_priors, num_fg).total;
}

extern "C" int ct_postprocess_batched(const float* boxes, const float* scores, int batch, int num_priors,
                                      int num_fg, float conf_thresh, float nms_thresh, int ge,
                                      int max_per_image, int out_cap, float* out_dets, int* out_count,
                                      int* out_index, int* overflow, void* workspace,
                                      size_t workspace_bytes, ct_stream_t stream)
{
    CT_REQUIRE(boxes && scores && out_dets && out_count && overflow && workspace, "ct_postprocess_batched: null pointer");
    CT_REQUIRE(batch > 0 && num_priors > 0 && num_fg > 0 && out_cap > 0, "ct_postprocess_batched: bad sizes");
    CT_REQUIRE(conf_thresh >= 0.f, "ct_postprocess_batched: conf_thresh must be >= 0 (scores are compared as unsigned bit patterns)");
    const size_t need = ct_postprocess_workspace_bytes(batch, num_priors, num_fg);
    if (workspace_bytes < need)
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_postprocess_batched: workspace %zu < %zu", workspace_bytes, need);
    PostWs w = carve((char*)workspace, batch, num_priors, num_fg);
    const int S = batch * num_fg;
    const int np2 = std::max(next_pow2(num_priors), 2);
    hipStream_t st = ctdet::as_stream(stream);
    CT_HIP(hipMemsetAsync(overflow, 0, sizeof(int), st));
    constexpr int kPrefix = 256;          // candidates per class in the bounding pass
    static_assert(kPrefix <= kPartialMin, "the bounding pass reads only sorted rows");
    const bool topk_rule = max_per_image > 0 && num_priors > kPrefix;
    constexpr size_t kLdsPartial = 4096 * 8 + kHistBins * 4 + 64, kLds4k = 4096 * 8 + 64, kLds8k = 8192 * 8 + 64;
    {
        // per instantiation, and the 8 192-key one (64 KB + 64 B of dynamic LDS) only for the launches that need it: a part whose
        // LDS cannot hold it still runs every network with at most 16 384 priors
        static hipError_t attr_err = [] {
            hipError_t e = hipFuncSetAttribute((const void*)select_sort_kernel<4096, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPartial);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)select_sort_kernel<4096, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds4k);
            return e;
        }();
        CT_HIP(attr_err);
        if (num_priors > 16384) {
            static hipError_t attr8k = hipFuncSetAttribute((const void*)select_sort_kernel<8192, false>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLds8k);
            CT_HIP(attr8k);
        }
    }
    const dim3 sort_grid(8 * ((batch + 7) / 8) * num_fg);
    auto full_sort = [&](const int* redo) {        // every candidate of every segment (or of the flagged ones) in order
        if (num_priors > 16384)
            hipLaunchKernelGGL((select_sort_kernel<8192, false>), sort_grid, dim3(kSortThreads), kLds8k, st, boxes, scores, batch, num_priors,
                               num_fg, conf_thresh, np2, w.keys, w.dets_sorted, w.sorted_idx, w.seg_count, w.seg_sorted, redo);
        else
            hipLaunchKernelGGL((select_sort_kernel<4096, false>), sort_grid, dim3(kSortThreads), kLds4k, st, boxes, scores, batch, num_priors,
                               num_fg, conf_thresh, np2, w.keys, w.dets_sorted, w.sorted_idx, w.seg_count, w.seg_sorted, redo);
    };
    {
        CT_PROF("select_sort_kernel", st);
        if (topk_rule)
            hipLaunchKernelGGL((select_sort_kernel<4096, true>), sort_grid, dim3(kSortThreads), kLdsPartial, st, boxes, scores, batch, num_priors,
                               num_fg, conf_thresh, np2, w.keys, w.dets_sorted, w.sorted_idx, w.seg_count, w.seg_sorted, (const int*)nullptr);
        else
            full_sort(nullptr);
    }
    CT_LAUNCH_CHECK("select_sort_kernel");
    int rc;
    if (topk_rule) {
        // pass 1: NMS of every class's best kPrefix candidates -> exact lower bound of the top-k
        // threshold -> cut every segment there (see prefix_cut_kernel); pass 2: NMS of what is left
        { CT_PROF("clamp_len_kernel", st); hipLaunchKernelGGL(clamp_len_kernel, dim3((S + 255) / 256), dim3(256), 0, st, w.seg_sorted, S, kPrefix, w.seg_len); }
        CT_LAUNCH_CHECK("clamp_len_kernel");
        rc = ctdet::nms_launch_strided(w.dets_sorted, w.seg_len, num_priors, S, nms_thresh, ge, w.keep, w.keep_count, st);
        if (rc != CT_OK) return rc;
        { CT_PROF("prefix_cut_kernel", st); hipLaunchKernelGGL(prefix_cut_kernel, dim3(batch, num_fg), dim3(256), 0, st, w.dets_sorted, w.keep, w.keep_count,
                           w.seg_count, w.seg_sorted, num_priors, num_fg, max_per_image, w.seg_len, w.redo, w.thr_img); }
        CT_LAUNCH_CHECK("prefix_cut_kernel");
        {   // segments whose cut lies behind their sorted prefix (none with a trained detector's few candidates per class;
            // with random weights only when the bounding pass keeps fewer than max_per_image boxes): workgroups of the others exit
            CT_PROF("select_sort_redo", st);
            full_sort(w.redo);
            hipLaunchKernelGGL(recut_kernel, dim3(S), dim3(256), 0, st, w.dets_sorted, w.seg_sorted, w.redo, w.thr_img, num_priors, num_fg, w.seg_len);
        }
        CT_LAUNCH_CHECK("select_sort_redo");
        rc = ctdet::nms_launch_strided(w.dets_sorted, w.seg_len, num_priors, S, nms_thresh, ge, w.keep, w.keep_count, st);
    } else {
        rc = ctdet::nms_launch_strided(w.dets_sorted, w.seg_count, num_priors, S, nms_thresh, ge, w.keep, w.keep_count, st);
    }
    if (rc != CT_OK) return rc;
    { CT_PROF("topk_kernel", st); hipLaunchKernelGGL(topk_kernel, dim3(batch), dim3(256), 0, st, w.dets_sorted, w.keep, w.keep_count,
                       num_priors, num_fg, max_per_image, out_cap, out_count, overflow); }
    CT_LAUNCH_CHECK("topk_kernel");
    { CT_PROF("gather_kernel", st); hipLaunchKernelGGL(gather_kernel, dim3(S), dim3(256), 0, st, w.dets_sorted, w.sorted_idx, w.keep,
                       out_count, num_priors, out_cap, out_dets, out_index); }
    CT_LAUNCH_CHECK("gather_kernel");
    return CT_OK;
}
