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

// Per-launch HIP-event timing for the measurement tools (ct_profile_enable / ct_profile_collect): a ProfScope
// around a launch records a start and a stop event on the launch stream when profiling is on, and costs one
// relaxed load when it is off.
bool prof_enabled();
void prof_start(const char* name, hipStream_t st, int* slot);
void prof_stop(hipStream_t st, int slot);
struct ProfScope {
    hipStream_t st;
    int slot = -1;
    ProfScope(const char* name, hipStream_t s) : st(s) { if (prof_enabled()) prof_start(name, s, &slot); }
    ~ProfScope() { if (slot >= 0) prof_stop(st, slot); }
};
#define CT_PROF_CAT2(a, b) a##b
#define CT_PROF_CAT(a, b) CT_PROF_CAT2(a, b)
#define CT_PROF(name, stream) ::ctdet::ProfScope CT_PROF_CAT(_ct_prof_scope_, __LINE__)(name, stream)

// Recording of weight-packing launches (ct_pack_record_begin / _end / ct_pack_run): while a recording is open on the
// calling thread the pack entry points append their kernel arguments here instead of launching; the recorded table
// is replayed each training step by two batched launches.  kind 0: direct layouts (ct_conv.hip PackArgs), 1: Winograd.
bool pack_recording();
// ct_scratch_prezeroed(1): the caller zeroes every accumulation buffer of the training kernels itself (one memset
// per pass over an arena) and the library skips its ~150 small per-launch memsets
bool scratch_prezeroed();
void set_scratch_prezeroed(bool on);
void pack_record(int kind, const void* args, size_t bytes);
int launch_pack_direct_batched(const void* items_dev, int n, hipStream_t st);
int launch_pack_wino_batched(const void* items_dev, int n, hipStream_t st);
size_t pack_direct_item_bytes();
size_t pack_wino_item_bytes();

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace ctdet
