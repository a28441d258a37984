/* A plain C99 caller of libctdet: what a maintainer of the reference links instead of its nvcc-built `_nms`
 * (utils/nms/gpu_nms.hpp:1-2) -- no Python, no torch, just include/ctdet.h.
 *
 *   gcc -std=c99 -Iinclude examples/c_caller.c -Lcontext-transformer_amd/lib -lctdet \
 *       -Wl,-rpath,$PWD/context-transformer_amd/lib -o c_caller && ./c_caller [gpu]
 *
 * Without an argument only host entry points run (ABI version, the reference's `--cpu` NMS); with `gpu` the
 * `_nms`-contract entry point ct_nms_sorted_host runs on device 0.  Exit code 0 = all answers as expected. */
#include <stdio.h>
#include <string.h>
#include "ctdet.h"

int main(int argc, char** argv)
{
    /* the known-answer case of SURVEY 8c: py_cpu_nms(..., 0.45) keeps boxes 0 and 2 */
    const float dets[4][5] = {{10, 10, 60, 60, .9f}, {12, 12, 62, 62, .8f}, {100, 100, 150, 150, .7f}, {10, 10, 60, 110, .6f}};
    int keep[4], n = 0;
    if (ct_abi_version() != 1) return 1;
    if (ct_cpu_nms(&dets[0][0], 4, 0.45f, 1, keep, &n) != CT_OK) {
        fprintf(stderr, "ct_cpu_nms: %s\n", ct_last_error_string());
        return 2;
    }
    printf("cpu_nms keeps %d:", n);
    for (int i = 0; i < n; ++i) printf(" %d", keep[i]);
    printf("\n");
    if (!(n == 2 && keep[0] == 0 && keep[1] == 2)) return 3;
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        /* rows are already in descending score order: the `_nms` precondition */
        n = 0;
        if (ct_nms_sorted_host(keep, &n, &dets[0][0], 4, 5, 0.45f, 0) != CT_OK) {
            fprintf(stderr, "ct_nms_sorted_host: %s\n", ct_last_error_string());
            return 4;
        }
        printf("_nms contract keeps %d:", n);
        for (int i = 0; i < n; ++i) printf(" %d", keep[i]);
        printf("\n");
        if (!(n == 2 && keep[0] == 0 && keep[1] == 2)) return 5;
    }
    return 0;
}
