// Micro-benchmark behind DESIGN.md section 4: how much independent VALU work hides behind an MFMA on gfx950 --
// v_mfma_f32_32x32x2_f32 (fp32 inputs, 64 cycles) vs v_mfma_f32_32x32x16_bf16 (32 cycles), 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV>          // KIND 0: fp32 MFMA, 1: bf16 MFMA; NV independent v_fma per MFMA
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 0.001f + i;
    const float a = seed * 0.5f, b = seed * 0.25f;
    i32x4 pa = {(int)threadIdx.x, 1, 2, 3}, pb = {4, 5, 6, (int)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV>
double run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;      // 256 CUs x (1 or 2) workgroups of 4 waves: 1 or 2 waves per SIMD
        printf("%d wave(s) per SIMD, %d MFMAs per wave: ms (cycles per MFMA slot per SIMD at 2.1 GHz)\n", wps, iters * 4);
#define ROW(K, NV) { double ms = run<K, NV>(out, blocks, iters); \
        printf("  %s + %2d v_fma per MFMA: %8.3f ms  (%.1f cycles per MFMA of ONE wave)\n", K ? "bf16 32x32x16" : "fp32 32x32x2 ", NV, ms, ms * 1e-3 * 2.1e9 / (iters * 4.0)); }
        ROW(0, 0) ROW(0, 2) ROW(0, 4) ROW(0, 8) ROW(0, 12)
        ROW(1, 0) ROW(1, 2) ROW(1, 4) ROW(1, 6) ROW(1, 8) ROW(1, 12)
    }
    return 0;
}
