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

// ---- bf16x3 operand fragments (csrc/ct_conv_x3.hip has the arithmetic) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int XT_BYTES = 3 * 8 * KT * 16;          // one 32-row tile of a [rows][64] operand as bf16x3 fragments: 12 KB
constexpr int XQ_BYTES = 3 * 8 * 16;               // one row in the register-operand layout: 384 B

// x = hi + mid + lo exactly (three bfloat16 pieces by truncation); returns the fp32 bit patterns whose upper halves
// are the pieces
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l)
{
    h = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    l = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, m));
}
__device__ __forceinline__ int pack_hi(unsigned e0, unsigned e1)
{
    return (int)__builtin_amdgcn_perm(e1, e0, 0x07060302u);
}

// element (row, feature o) of a [rows_pad][64] operand, split and stored in one of three fragment orders:
//   mode 0  rows as the register (B) operand of a contraction over features:  [b][row][piece 3][octet 8][8]
//   mode 1  32-row tiles as the LDS (A) operand of a contraction over features: [b][tile][piece][octet 8][row 32][8]
//   mode 2  32-row tiles as the LDS (A) operand of a contraction over ROWS:    [b][tile][piece][kg 2][h 2][feature 64][8];
//           slot j of (kg, h) = the row that accumulator register 8 kg + j of lane half h holds (acc_row), so a matrix
//           held in accumulator registers goes to the MFMA as the other operand as it sits
__device__ __forceinline__ void x3_emit(float y, unsigned short* __restrict__ out, int mode, int b, int row, int rows_pad,
                                        int o)
{
    unsigned ph, pm, pl;
    split3(y, ph, pm, pl);
    size_t base, pstride;        // in bf16 elements
    if (mode == 0) {
        base = ((size_t)b * rows_pad + row) * (XQ_BYTES / 2) + (o >> 3) * 8 + (o & 7);
        pstride = 8 * 8;
    } else {
        const int tile = row / KT, kl = row % KT;
        const size_t tb = ((size_t)b * (rows_pad / KT) + tile) * (XT_BYTES / 2);
        if (mode == 1) {
            base = tb + ((size_t)(o >> 3) * KT + kl) * 8 + (o & 7);
        } else {
            const int h = (kl >> 2) & 1, rr = (kl & 3) + 4 * (kl >> 3);     // kl = acc_row(rr, h)
            base = tb + ((size_t)((rr >> 3) * 2 + h) * DP + o) * 8 + (rr & 7);
        }
        pstride = 8 * KT * 8;
    }
    out[base] = (unsigned short)(ph >> 16);
    out[base + pstride] = (unsigned short)(pm >> 16);
    out[base + 2 * pstride] = (unsigned short)(pl >> 16);
}

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
    __shared__ float Wt[DP * DP];                                  // Wt[i][o]
    __shared__ __attribute__((aligned(16))) float Xt[DP * 64];     // Xt[i][row]: four rows of a feature are one 16-byte broadcast
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
        Xt[i * 64 + r] = (row < rows_valid && i < d) ? x[((size_t)b * rows_valid + row) * d + i] : 0.f;
    }
    __syncthreads();
    // thread = output feature o of rows 16 rg .. 16 rg + 15, four rows at a time (one weight read and one 16-byte row read
    // per four FMAs); every output sums its 64 products in the order i = 0 .. 63
    const int o = tid & 63, rg = tid >> 6;
    const float bo = (o < d) ? bias[o] : 0.f;
    for (int rq = 0; rq < 4; ++rq) {
        const int rl = 16 * rg + 4 * rq;
        if (r0 + rl >= rows_pad) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int i = 0; i < DP; ++i) {
            const float w = Wt[i * DP + o];
            const float4 xv = *reinterpret_cast<const float4*>(&Xt[i * 64 + rl]);
            acc[0] += xv.x * w; acc[1] += xv.y * w; acc[2] += xv.z * w; acc[3] += xv.w * w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = r0 + rl + k;
            if (row >= rows_pad) break;
            const float y = (row < rows_valid && o < d) ? acc[k] + bo + Xt[o * 64 + rl + k] : 0.f;
            if (o_sw) o_sw[((size_t)b * rows_pad + row) * DP + (o & 1) * 32 + (o >> 1)] = y;
            if (o_t) o_t[((size_t)b * DP + o) * rows_pad + row] = y;
            if (o_rows) o_rows[((size_t)b * rows_pad + row) * DP + o] = y;
            if (o_plain && row < rows_valid && o < d) o_plain[((size_t)b * rows_valid + row) * ostride + o] = y;
        }
    }
}

}  // namespace
