// What a 12-MFMA k-step costs a workgroup that has its CU to itself (the regime of the 5x5 .. 1x1 layers and of everything at
// bs 4: ~0.7 us per step measured in conv_x3_f32<64,64,32>), piece by piece, and the shader clock in that regime:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/small_step tools/ubench/small_step.hip && /tmp/small_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 12 dependent MFMAs (2 accumulators: 10 + 2) per step; 1: + bare barrier; 2: + 12 ds_read_b128 feeding them after the
// barrier; 3: + 6 ds_write_b128 before the barrier; 4: + 11 global loads issued per step, consumed the step after
template <int MODE>
__global__ __launch_bounds__(256) void k(const i32x4* g, float* out, unsigned long long* clk, int steps)
{
    __shared__ i32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = g[i & 1023];
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    i32x4 f[12];
    for (int i = 0; i < 12; ++i) f[i] = lds[(tid + 64 * i) & 4095];
    i32x4 pre[11];
    const i32x4* gp = g + (blockIdx.x * 256 + tid) % 1024;
    for (int i = 0; i < 11; ++i) pre[i] = gp[i * 1024];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 2)
            for (int i = 0; i < 12; ++i) f[i] = lds[(tid + 64 * i + s) & 4095];
        for (int i = 0; i < 12; ++i) {
            f32x16& d = i < 10 ? a0 : a1;
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, f[(i + 1) % 12]), d, 0, 0, 0);
        }
        if (MODE >= 4) {
            for (int i = 0; i < 6; ++i) lds[(tid + 256 * i + 7 * s) & 4095] = pre[i] + pre[(i + 5) % 11];
            for (int i = 0; i < 11; ++i) pre[i] = gp[((i + s) & 63) * 1024];
        } else if (MODE >= 3) {
            for (int i = 0; i < 6; ++i) lds[(tid + 256 * i + 7 * s) & 4095] = f[i];
        }
        if (MODE >= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += a0[r] + a1[r];
    for (int i = 0; i < 11; ++i) sum += (float)pre[i].x;
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same work software-pipelined over the barrier: the fragments of step s + 1 are read (from a third LDS buffer) while the
// MFMAs of step s run, the LDS writes and the global loads of later tiles sit between the MFMAs.
__global__ __launch_bounds__(256) void kp(const i32x4* g, float* out, unsigned long long* clk, int steps)
{
    __shared__ i32x4 lds[3 * 1280];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 1280; i += 256) lds[i] = g[i & 1023];
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    i32x4 fA[12], fB[12];
    for (int i = 0; i < 12; ++i) fA[i] = lds[(tid + 64 * i) % 1280];
    i32x4 pre[11];
    const i32x4* gp = g + (blockIdx.x * 256 + tid) % 1024;
    for (int i = 0; i < 11; ++i) pre[i] = gp[i * 1024];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto step = [&](int s, i32x4 (&cur)[12], i32x4 (&nxt)[12]) {
        const int rb = ((s + 1) % 3) * 1280, wb = ((s + 2) % 3) * 1280;
        for (int i = 0; i < 12; ++i) nxt[i] = lds[rb + (tid + 64 * i + s) % 1280];
        for (int i = 0; i < 12; ++i) {
            f32x16& d = i < 10 ? a0 : a1;
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[i]), __builtin_bit_cast(bf16x8, cur[(i + 1) % 12]), d, 0, 0, 0);
            if (i < 6) lds[wb + (tid + 256 * i + 7 * s) % 1280] = pre[i] + pre[(i + 5) % 11];
            if (i >= 1) pre[i - 1] = gp[((i - 1 + s) & 63) * 1024];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int s = 0; s < steps; s += 2) {
        step(s, fA, fB);
        step(s + 1, fB, fA);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += a0[r] + a1[r];
    for (int i = 0; i < 11; ++i) sum += (float)pre[i].x;
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// Eight waves per workgroup, the work of a step divided between the two waves of a SIMD: six MFMAs, six fragment reads, three
// LDS writes and six global loads each (the two k-groups of a 32-channel step on different waves).
__global__ __launch_bounds__(512) void k8(const i32x4* g, float* out, unsigned long long* clk, int steps)
{
    __shared__ i32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) lds[i] = g[i & 1023];
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    i32x4 f[6];
    for (int i = 0; i < 6; ++i) f[i] = lds[(tid + 64 * i) & 4095];
    i32x4 pre[6];
    const i32x4* gp = g + (blockIdx.x * 512 + tid) % 1024;
    for (int i = 0; i < 6; ++i) pre[i] = gp[i * 1024];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < 6; ++i) f[i] = lds[(tid + 64 * i + s) & 4095];
        for (int i = 0; i < 6; ++i) {
            f32x16& d = i < 5 ? a0 : a1;
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, f[(i + 1) % 6]), d, 0, 0, 0);
        }
        for (int i = 0; i < 3; ++i) lds[(tid + 512 * i + 7 * s) & 4095] = pre[i] + pre[(i + 3) % 6];
        for (int i = 0; i < 6; ++i) pre[i] = gp[((i + s) & 63) * 1024];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += a0[r] + a1[r];
    for (int i = 0; i < 6; ++i) sum += (float)pre[i].x;
    out[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const i32x4* g, float* out, unsigned long long* clk, int wgs, const char* what)
{
    const int steps = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, g, out, clk, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, g, out, clk, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s %4d WGs: %.3f us per step, %.0f shader ticks per step, shader clock %.2f GHz (memtime / memrealtime at 100 MHz)\n", what, wgs,
           ms * 1e3 / steps, (double)h[0] / steps, (double)h[0] / ((double)h[1] * 10.0));
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    i32x4* g; float* out; unsigned long long* clk;
    hipMalloc(&g, 64 * 1024 * 16 * 2); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 16);
    hipMemset(g, 0x3c, 64 * 1024 * 16 * 2);
    for (int wgs : {26, 256, 512}) {
        run<0>(g, out, clk, wgs, "12 MFMAs (chains of 10 + 2)");
        run<1>(g, out, clk, wgs, "+ bare barrier");
        run<2>(g, out, clk, wgs, "+ 12 ds_read_b128 behind the barrier");
        run<3>(g, out, clk, wgs, "+ 6 ds_write_b128 in front of it");
        run<4>(g, out, clk, wgs, "+ 11 global loads per step, consumed one step later");
        {
            const int steps = 4000;
            hipLaunchKernelGGL(kp, dim3(wgs), dim3(256), 0, 0, g, out, clk, steps);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kp, dim3(wgs), dim3(256), 0, 0, g, out, clk, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2];
            hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-58s %4d WGs: %.3f us per step, %.0f shader ticks per step\n", "the same, software-pipelined over the barrier (3 buffers)", wgs,
                   ms * 1e3 / steps, (double)h[0] / steps);
        }
        {
            const int steps = 4000;
            hipLaunchKernelGGL(k8, dim3(wgs), dim3(512), 0, 0, g, out, clk, steps);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k8, dim3(wgs), dim3(512), 0, 0, g, out, clk, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = -1.f; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2] = {0, 0};
            hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-58s %4d WGs: %.3f us per step, %.0f shader ticks per step\n", "the same work on eight waves (two per SIMD, half each)", wgs,
                   ms * 1e3 / steps, (double)h[0] / steps);
        }
    }
    return 0;
}
