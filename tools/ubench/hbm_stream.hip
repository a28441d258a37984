// HBM streaming rates on one MI355X for the access shapes of the Winograd transform kernels (csrc/ct_wino4s.hip):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream tools/ubench/hbm_stream.hip && /tmp/hbm_stream
// W  contiguous: every wave stores 1 KB pieces (16 B per lane) one after the other
// WP plane-scattered: a workgroup's pieces go to 108 planes `plane` bytes apart (wino4s_in: 36 points x 3 pieces), 1 KB per wave
//    and plane, neighbouring workgroups next to each other inside a plane
// R  contiguous 16-byte loads;  RW  copy (R + W bytes counted)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(i32x4* dst, size_t n16, int val)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = i32x4{val, val, val, val};
}
__global__ __launch_bounds__(256) void k_read(const i32x4* src, size_t n16, int* sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    i32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) *sink = 1;
}
__global__ __launch_bounds__(256) void k_copy(const i32x4* src, i32x4* dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// planes x (items x 4 KB): workgroup w handles items w, w + grid, ...; per item 4 waves x 1 KB into each plane
__global__ __launch_bounds__(256) void k_write_planes(i32x4* dst, int planes, size_t plane16, int items, int val)
{
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        i32x4* p = dst + (size_t)it * 256 + threadIdx.x;
        for (int pl = 0; pl < planes; ++pl) p[(size_t)pl * plane16] = i32x4{val, val, val, val};
    }
}
// the same bytes, item-major: an item's planes x 4 KB are contiguous (what a [tile block][point] layout would write)
__global__ __launch_bounds__(256) void k_write_items(i32x4* dst, int planes, int items, int val)
{
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        i32x4* p = dst + (size_t)it * planes * 256 + threadIdx.x;
        for (int pl = 0; pl < planes; ++pl) p[(size_t)pl * 256] = i32x4{val, val, val, val};
    }
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int* sink; CK(hipMalloc(&sink, 4));
    const size_t cap = (size_t)1 << 30;
    i32x4 *a, *b;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 2, cap));
    auto timeit = [&](auto&& launch, int reps) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / reps * 1e-3;
    };
    for (size_t mb : {64, 160, 354, 1024}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        for (int wgs : {512, 2048, 8192}) {
            const double tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, 0, a, n16, 3); }, 10);
            const double tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, a, n16, sink); }, 10);
            const double tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, a, b, n16 / 2 > 0 ? n16 : n16); }, 10);
            printf("%5zu MB, %5d workgroups:  W %.2f TB/s (%.1f us)   R %.2f TB/s   RW copy %.2f TB/s (R + W bytes)\n", mb, wgs,
                   bytes / tw / 1e12, tw * 1e6, bytes / tr / 1e12, 2.0 * bytes / tc / 1e12);
        }
    }
    // wino4s_in of 512 -> 512 @38x38 bs 32: 108 planes of 3.28 MB (25 tile blocks x 32 chunks x 4 KB), 800 items of 4 KB per plane
    for (int planes : {108}) {
        for (int items : {800, 2000}) {      // 108 x 2000 x 4 KB = 885 MB of the 1 GB buffer
            const size_t plane16 = (size_t)items * 256;
            for (int wgs : {400, 512, 1024}) {
                const double bytes = (double)planes * items * 4096;
                const double tp = timeit([&] { hipLaunchKernelGGL(k_write_planes, dim3(wgs), dim3(256), 0, 0, a, planes, plane16, items, 5); }, 10);
                const double ti = timeit([&] { hipLaunchKernelGGL(k_write_items, dim3(wgs), dim3(256), 0, 0, a, planes, items, 5); }, 10);
                printf("%d planes x %d items x 4 KB = %.0f MB, %4d workgroups:  plane-scattered W %.2f TB/s (%.1f us)   item-major W %.2f TB/s (%.1f us)\n",
                       planes, items, bytes / 1e6, wgs, bytes / tp / 1e12, tp * 1e6, bytes / ti / 1e12, ti * 1e6);
            }
        }
    }
    return 0;
}
