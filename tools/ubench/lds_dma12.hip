// Probe behind csrc/ct_wino4f.hip: where does `buffer_load_dwordx3 ... lds` (12 bytes per lane) put each lane's data, and
// what happens to the LDS bytes of a lane whose buffer offset is out of range?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma12.hip -o /tmp/lds_dma12 && /tmp/lds_dma12
// Prints, per lane, the LDS dword index at which its three source dwords landed, and the content of the slot of the
// out-of-range lanes (pre-filled with a marker) after the DMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ __launch_bounds__(64) void probe(const int* src, unsigned bytes, int* dump)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int* l = reinterpret_cast<int*>(lds);
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) l[i] = -7;                     // marker
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(src), 0, bytes, 0x00020000);
    const int voff = (lane % 5 == 3) ? 0x7FFFFFF0 : lane * 12;        // every fifth lane out of range
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + 256), 12, voff, 0, 0, 0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) dump[i] = l[i];
}

int main()
{
    std::vector<int> h(64 * 3);
    for (int i = 0; i < 64 * 3; ++i) h[i] = 1000 + i;                   // lane l holds 1000 + 3l .. + 2
    int *src, *dump;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dump, 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, src, (unsigned)(h.size() * 4), dump);
    std::vector<int> d(512);
    hipMemcpy(d.data(), dump, 512 * 4, hipMemcpyDeviceToHost);
    printf("LDS dwords 64 .. (destination base = byte 256 = dword 64), -7 = untouched marker:\n");
    for (int i = 56; i < 64 + 64 * 4 + 8; ++i) printf("%s%5d", (i - 64) % 12 == 0 ? "\n  " : " ", d[i]);
    printf("\n");
    int stride = 0;
    for (int i = 64; i < 512; ++i)
        if (d[i] == 1000 + 3) { stride = (i - 64) * 4; break; }        // first dword of lane 1
    printf("lane stride in LDS: %d bytes\n", stride);
    int oob_written = 0, oob_zero = 0;
    for (int lane = 3; lane < 64; lane += 5)
        for (int k = 0; k < 3; ++k) {
            const int v = d[64 + (lane * (stride ? stride : 12)) / 4 + k];
            oob_written += v != -7;
            oob_zero += v == 0;
        }
    printf("out-of-range lanes: %d of 39 dwords overwritten, %d of them with 0\n", oob_written, oob_zero);
    return 0;
}
