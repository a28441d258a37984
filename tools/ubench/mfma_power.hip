// Micro-benchmark behind DESIGN.md section 4 ("law 3"): what the matrix pipe of THIS MI355X sustains when the operands
// are real data.  A loop of independent v_mfma_f32_32x32x16_bf16 (or v_mfma_f32_32x32x2_f32) with no memory traffic at all,
// 256 CUs x 2 workgroups of 4 waves, operands either all zero or gaussian.  Zero operands leave the multiplier arrays
// idle; with real data the chip runs into its power limit and lowers the clock, so the dense "peak" of the data sheet
// (2.5 PFLOP/s bf16 at 2.4 GHz) is the zero-data number.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND>                  // 0: fp32 MFMA 32x32x2, 1: bf16 MFMA 32x32x16
__global__ __launch_bounds__(256) void k(const i32x4* __restrict__ ops, float* out, int iters)
{
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // four operand pairs per lane, loaded once
    i32x4 pa[4], pb[4];
    for (int j = 0; j < 4; ++j) {
        pa[j] = ops[(threadIdx.x * 8 + j) % 2048];
        pb[j] = ops[(threadIdx.x * 8 + 4 + j) % 2048];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, pa[j].x), __builtin_bit_cast(float, pb[j].x), acc[j], 0, 0, 0);
            else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[j]), __builtin_bit_cast(bf16x8, pb[j]), acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
double run(const i32x4* ops, float* out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(512), dim3(256), 0, 0, ops, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(512), dim3(256), 0, 0, ops, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    i32x4* ops;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&ops, 2048 * sizeof(i32x4));
    std::vector<unsigned> h(2048 * 4);
    srand(1);
    auto gauss = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return std::sqrt(-2 * std::log(u)) * std::cos(6.2831853 * v); };
    for (int kind = 0; kind < 2; ++kind)
        for (int data = 0; data < 2; ++data) {
            for (auto& w : h) {
                if (!data) { w = 0; continue; }
                if (kind == 0) { float f = (float)(gauss() * 0.05); w = *reinterpret_cast<unsigned*>(&f); }
                else {
                    float f0 = (float)gauss(), f1 = (float)gauss();
                    w = (*reinterpret_cast<unsigned*>(&f0) >> 16) | (*reinterpret_cast<unsigned*>(&f1) & 0xFFFF0000u);
                }
            }
            hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            for (int iters : {20000, 200000}) {       // ~15 ms and ~150 ms of back-to-back MFMAs
                const double ms = kind ? run<1>(ops, out, iters) : run<0>(ops, out, iters);
                const double flop = 512.0 * 4 * iters * 4.0 * (kind ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
                printf("%s, %s operands, %6d x 4 MFMAs per wave: %8.2f ms = %7.1f TFLOP/s = %.0f MHz-equivalent at full issue rate\n",
                       kind ? "bf16 32x32x16" : "fp32 32x32x2 ", data ? "gaussian" : "zero    ", iters, ms, flop / ms / 1e9,
                       (iters * 4.0 * 2 * (kind ? 32 : 64)) / (ms * 1e-3) / 1e6);
            }
        }
    return 0;
}
