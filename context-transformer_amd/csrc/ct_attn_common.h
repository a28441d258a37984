// Shared pieces of the Context-Transformer kernels (forward: ct_attn.hip, backward: ct_attn_bwd.hip).
#pragma once
#include "ct_common.h"
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int DP = 64;      // padded feature dim
constexpr int QW = 32;      // queries per wave
constexpr int QB = 128;     // queries per workgroup
constexpr int KT = 32;      // keys per tile

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// y = Linear(x) + x for 64 rows per block, written to any of four layouts (null = skip):
//   o_sw    [B][rows_pad][2][32]  "swizzled" rows: o_sw[r][h][s] = y[r][2s+h]  (MFMA B-operand order)
//   o_t     [B][64][rows_pad]     transposed (feature-major, rows contiguous)
//   o_rows  [B][rows_pad][64]     natural rows, zero padded
//   o_plain [B][rows_valid][ostride] first d columns only (fc_base half of the 'incre' output)
__global__ __launch_bounds__(256) void ctx_project_kernel(const float* __restrict__ x, int rows_valid,
                                                          int rows_pad, int d,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ o_sw, float* __restrict__ o_t,
                                                          float* __restrict__ o_rows,
                                                          float* __restrict__ o_plain, int ostride)
{
    __shared__ float Wt[DP * DP];      // Wt[i][o]
    __shared__ float Xs[64 * DP];      // 64 rows
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < DP * DP; e += 256) {
        const int i = e / DP, o = e % DP;
        Wt[e] = (i < d && o < d) ? W[o * d + i] : 0.f;
    }
    for (int e = tid; e < 64 * DP; e += 256) {
        const int r = e / DP, i = e % DP;
        const int row = r0 + r;
        Xs[e] = (row < rows_valid && i < d) ? x[((size_t)b * rows_valid + row) * d + i] : 0.f;
    }
    __syncthreads();
    const int o = tid & 63, rg = tid >> 6;
    const float bo = (o < d) ? bias[o] : 0.f;
    for (int r = rg; r < 64; r += 4) {
        const int row = r0 + r;
        if (row >= rows_pad) break;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < DP; ++i) acc += Xs[r * DP + i] * Wt[i * DP + o];
        float y = (row < rows_valid && o < d) ? acc + bo + Xs[r * DP + o] : 0.f;
        if (o_sw) o_sw[((size_t)b * rows_pad + row) * DP + (o & 1) * 32 + (o >> 1)] = y;
        if (o_t) o_t[((size_t)b * DP + o) * rows_pad + row] = y;
        if (o_rows) o_rows[((size_t)b * rows_pad + row) * DP + o] = y;
        if (o_plain && row < rows_valid && o < d) o_plain[((size_t)b * rows_valid + row) * ostride + o] = y;
    }
}

}  // namespace
