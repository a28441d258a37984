// Internal helpers shared by the libctdet translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "ctdet.h"

namespace ctdet {

// Thread-local text of the last error (returned by ct_last_error_string()).
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(ct_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define CT_HIP(expr)                                                                    \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return ::ctdet::fail(CT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                 hipGetErrorString(_e), __FILE__, __LINE__);            \
    } while (0)

#define CT_LAUNCH_CHECK(name)                                                           \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess)                                                           \
            return ::ctdet::fail(CT_ERR_HIP, "launch of %s failed: %s", name,           \
                                 hipGetErrorString(_e));                                \
    } while (0)

#define CT_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) return ::ctdet::fail(CT_ERR_INVALID, __VA_ARGS__);                 \
    } while (0)

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace ctdet
