// The reference's own native call, unchanged: gpu_nms.pyx hands `_nms` (declared in utils/nms/gpu_nms.hpp:1-2 with
// C++ linkage) a score-sorted host array.  libctdet exports that very symbol, so this file needs no ctdet header:
//
//   g++ examples/cpp_caller.cpp -Lcontext-transformer_amd/lib -lctdet -Wl,-rpath,$PWD/context-transformer_amd/lib -o cpp_caller
//   ./cpp_caller
//
// Runs the known-answer case of SURVEY 8c on device 0; exit code 0 = boxes 0 and 2 kept.
#include <cstdio>

void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id);

int main()
{
    const float dets[4][5] = {{10, 10, 60, 60, .9f}, {12, 12, 62, 62, .8f}, {100, 100, 150, 150, .7f}, {10, 10, 60, 110, .6f}};
    int keep[4] = {-1, -1, -1, -1}, n = -1;
    _nms(keep, &n, &dets[0][0], 4, 5, 0.45f, 0);
    std::printf("_nms keeps %d:", n);
    for (int i = 0; i < n; ++i) std::printf(" %d", keep[i]);
    std::printf("\n");
    return n == 2 && keep[0] == 0 && keep[1] == 2 ? 0 : 1;
}
