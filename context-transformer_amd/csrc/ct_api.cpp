// libctdet: error reporting, ABI version and device query.
#include "ct_common.h"
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace ctdet {

char* error_buffer()
{
    static thread_local char buf[512] = "no error";
    return buf;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-launch event timing (measurement only) ----
namespace {
struct ProfRec { const char* name; hipEvent_t e0, e1; bool stopped; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
}  // namespace

bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed); }

void prof_start(const char* name, hipStream_t st, int* slot)
{
    ProfRec r{name, nullptr, nullptr, false};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    if (hipEventRecord(r.e0, st) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    *slot = (int)g_prof.size() - 1;
}

void prof_stop(hipStream_t st, int slot)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof.size() && hipEventRecord(g_prof[slot].e1, st) == hipSuccess) g_prof[slot].stopped = true;
}

}  // namespace ctdet

extern "C" int ct_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(ctdet::g_prof_mu);
    for (auto& r : ctdet::g_prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    ctdet::g_prof.clear();
    ctdet::g_prof_on.store(on != 0);
    return CT_OK;
}

extern "C" int ct_profile_collect(ct_profile_record* out, int max_records, int* num_records)
{
    CT_REQUIRE(num_records && (out || max_records == 0), "ct_profile_collect: null pointer");
    std::lock_guard<std::mutex> lk(ctdet::g_prof_mu);
    int n = 0;
    for (auto& r : ctdet::g_prof) {
        if (!r.stopped) continue;
        if (n < max_records) {
            CT_HIP(hipEventSynchronize(r.e1));
            float ms = 0.f;
            CT_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            out[n].name = r.name;
            out[n].ms = ms;
        }
        ++n;
    }
    *num_records = n;
    return CT_OK;
}

extern "C" int ct_abi_version(void) { return CTDET_ABI_VERSION; }

extern "C" const char* ct_last_error_string(void) { return ctdet::error_buffer(); }

extern "C" int ct_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len)
{
    hipDeviceProp_t prop;
    CT_HIP(hipGetDeviceProperties(&prop, device));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return CT_OK;
}

// The reference's only native symbol, with its own (C++) linkage and argument list: utils/nms/gpu_nms.hpp:1-2,
// defined by utils/nms/nms_kernel.cu:91-144.  A build of the reference's gpu_nms.pyx links against libctdet
// unchanged (the mangled name is _Z4_nmsPiS_PKfiifi).  Like the original it returns nothing; failures are
// printed to stderr (the original prints the CUDA error and carries on, nms_kernel.cu:12-19) and num_out is 0.
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id)
{
    if (ct_nms_sorted_host(keep_out, num_out, boxes_host, boxes_num, boxes_dim, nms_overlap_thresh, device_id) != CT_OK) {
        fprintf(stderr, "_nms (libctdet): %s\n", ct_last_error_string());
        if (num_out) *num_out = 0;
    }
}
