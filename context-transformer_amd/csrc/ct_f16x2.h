// libctdet: the "f16x2" operand form shared by the kernels that run fp32 convolutions on the f16 matrix pipe (round 6).
// Internal header.
//
// An fp32 value x is carried as TWO binary16 pieces,  x * 2^e = hi + lo,  hi = rne16(x 2^e),  lo = rne16(x 2^e - hi):
// 11 + 1 + 11 significant bits (the sign of lo is the twelfth), |x 2^e - hi - lo| <= 2^-23 |x 2^e| -- and a product a.b as the
// THREE piece products hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 (each exact in fp32: 22-bit significands), fp32
// accumulation; the dropped lo.lo is <= 2^-24 |a b|.  Against the bf16x3 form (three bfloat16 pieces by truncation, six
// products, dropped terms <= 2^-23 |a b|) that is HALF the matrix instructions, two thirds of the operand bytes and 2
// instead of 5.5 vector instructions per split value, at the same error (tools/ubench/f16x2_probe.hip on the MI355X,
// profiles/r06_f16x2_probe.txt: 32 x 32 x 512 GEMM against fp64, rms of the output range, one / two accumulators:
// bf16x3 8.0e-8 / 3.3e-8, f16x2 6.1e-8 / 3.8e-8, a sequential fp32 FMA chain 7.6e-8).
//
// What binary16 does not have is bfloat16's range (largest value 65504, smallest normal 6.1e-5), so every operand tensor is
// scaled by a power of two chosen from its maximum: e = the largest exponent for which no hi piece can exceed 2^15.  The
// maximum is the tensor's own -- a bound known before the values are (input transform: ||B^T||_inf^2 max|x|; weight transform:
// ||G||_inf^2 max|g|) -- so nothing overflows whatever the data, and since the matrix pipe honours SUBNORMAL binary16 inputs
// (probe, question 1) a value 2^-14 below the scaled maximum still has an exact hi piece and an lo piece that is only
// coarser in absolute terms (quantum 2^-24 of the scaled unit = 2^-39 of the maximum): the GEMM error is unchanged with a
// scale 1024 x too small (probe, question 4).  The consumer multiplies the fp32 sums by 2^-(eA + eB) -- exact.
// An Inf anywhere in an image makes its maximum Inf: the exponent is then 0; a NaN or Inf propagates through the pieces (binary16 has
// both) like through any fp32 kernel, whatever the exponent.
#pragma once
#include <hip/hip_runtime.h>

namespace ctdet {
namespace h2 {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// (x0, x1) -> packed hi pieces, packed lo pieces (element 0 in the low half-word): v_cvt_pk_f16_f32, two v_fma_mix_f32
// (x - hi with the binary16 operand read in place), v_cvt_pk_f16_f32.  Identical to the host's rne split
// (tools/ubench/f16x2_probe.hip, question 3: 0 of 2^20 values differ).
__device__ __forceinline__ void split2(float x0, float x1, int& hi, int& lo)
{
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, f16x2_t));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    hi = (int)h;
    lo = __builtin_bit_cast(int, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}

// Exponent e such that |v| 2^e < 2^15 for every |v| <= 2^growth_log2 * max, given the bit pattern of max >= 0
// (growth_log2 = ceil(log2) of the transform's gain).  0 for an all-zero, NaN or Inf tensor; clamped to +-100.
__host__ __device__ __forceinline__ int exponent_for(unsigned max_bits, int growth_log2)
{
    if (max_bits == 0u || max_bits >= 0x7F800000u) return 0;
    const int ex = (int)(max_bits >> 23) - 127;              // max < 2^(ex + 1)   (a subnormal maximum: ex = -127, clamped below)
    const int e = 15 - (ex + 1 + growth_log2);
    return e < -100 ? -100 : e > 100 ? 100 : e;
}
constexpr int kGrowthBtB = 6;      // ||B^T||_inf^2 = 5.6875^2 = 32.35 < 2^6 (interpolation points 0, +-3/4, +-3/2, inf: ct_wino4_points.h)
constexpr int kGrowthGG = 1;       // ||G||_inf^2 = 1.2197^2 = 1.4877 < 2^1
constexpr int kGrowthNone = 0;     // operands used as they are (direct convolutions)

// max over a wave of a non-negative bit pattern (unsigned order = float order for non-negative floats; NaN sorts above Inf)
__device__ __forceinline__ unsigned wave_max(unsigned m)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, d, 64);
        m = o > m ? o : m;
    }
    return m;
}

// Maxima of |activation| travel PER IMAGE (ct_conv_desc.in_absmax / out_absmax): `batch` lines of CT_ABSMAX_LINE_BYTES, the bit
// pattern of image n's maximum in the first word of line n.  Per image and not per batch so that what a kernel computes for an
// image never depends on the other images of its batch (the subnormal lo pieces make the split depend on the exponent in the
// last bits; the harness tests require detections that are bit-identical whatever the batch composition).
constexpr int kLineWords = 32;

__device__ __forceinline__ int image_exponent(const unsigned* __restrict__ lines, int n, int growth_log2)
{
    return exponent_for(lines[(size_t)n * kLineWords], growth_log2);
}

// running maximum of |v| in a thread: one v_max_f32 with the |.| source modifier per value.  A NaN is skipped (IEEE maxNum): it does
// not need the maximum -- it turns every piece it is split into, and every sum it enters, into NaN whatever the exponent is; an Inf
// becomes the maximum (exponent 0).
__device__ __forceinline__ void track_absmax(float& run, float v)
{
    run = __builtin_fmaxf(run, __builtin_fabsf(v));
}

// The producer side: every lane of a wave brings (image n, maximum of |v| over what it stored for that image; n < 0 = nothing);
// one atomic max per distinct image in the wave (one or two on the large maps).  Must be called by all lanes of the wave.
__device__ __forceinline__ void flush_absmax(unsigned* __restrict__ lines, int n, float run)
{
    const unsigned m = __builtin_bit_cast(unsigned, run) & 0x7FFFFFFFu;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(n >= 0 && m != 0u);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int n0 = __shfl(n, leader, 64);
        const bool mine = n == n0 && ((todo >> lane) & 1ull);
        const unsigned mm = wave_max(mine ? m : 0u);
        if (lane == leader) atomicMax(lines + (size_t)n0 * kLineWords, mm);
        todo &= ~__ballot(mine);
    }
}

}  // namespace h2
}  // namespace ctdet
