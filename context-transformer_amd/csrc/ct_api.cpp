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
    if (hipEventCreate(&r.e0) != hipSuccess) return;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
    if (hipEventRecord(r.e0, st) != hipSuccess) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        return;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    *slot = (int)g_prof.size() - 1;
}

void prof_stop(hipStream_t st, int slot)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof.size() && hipEventRecord(g_prof[slot].e1, st) == hipSuccess) g_prof[slot].stopped = true;
}

// ---- recorded weight packing (training steps) ----
namespace {
thread_local bool t_pack_rec = false;
thread_local std::vector<unsigned char> t_pack_items[2];
}  // namespace

bool pack_recording() { return t_pack_rec; }

namespace {
thread_local bool t_prezeroed = false;
}
bool scratch_prezeroed() { return t_prezeroed; }
void set_scratch_prezeroed(bool on) { t_prezeroed = on; }

void pack_record(int kind, const void* args, size_t bytes)
{
    const unsigned char* b = static_cast<const unsigned char*>(args);
    t_pack_items[kind & 1].insert(t_pack_items[kind & 1].end(), b, b + bytes);
}

}  // namespace ctdet

extern "C" int ct_scratch_prezeroed(int on)
{
    ctdet::set_scratch_prezeroed(on != 0);
    return CT_OK;
}

extern "C" int ct_pack_record_begin(void)
{
    ctdet::t_pack_items[0].clear();
    ctdet::t_pack_items[1].clear();
    ctdet::t_pack_rec = true;
    return CT_OK;
}

extern "C" size_t ct_pack_record_bytes(void)
{
    return ctdet::align_up(ctdet::t_pack_items[0].size(), 256) + ctdet::align_up(ctdet::t_pack_items[1].size(), 256) + 256;
}

extern "C" int ct_pack_record_end(void* table_dev, size_t table_bytes, int* num_direct, int* num_wino, ct_stream_t stream)
{
    ctdet::t_pack_rec = false;
    CT_REQUIRE(table_dev && num_direct && num_wino, "ct_pack_record_end: null pointer");
    if (table_bytes < ct_pack_record_bytes())
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_pack_record_end: table %zu < %zu bytes", table_bytes, ct_pack_record_bytes());
    hipStream_t st = ctdet::as_stream(stream);
    const size_t n0 = ctdet::t_pack_items[0].size(), n1 = ctdet::t_pack_items[1].size();
    unsigned char* t = static_cast<unsigned char*>(table_dev);
    if (n0) CT_HIP(hipMemcpyAsync(t, ctdet::t_pack_items[0].data(), n0, hipMemcpyHostToDevice, st));
    if (n1) CT_HIP(hipMemcpyAsync(t + ctdet::align_up(n0, 256), ctdet::t_pack_items[1].data(), n1, hipMemcpyHostToDevice, st));
    CT_HIP(hipStreamSynchronize(st));                   // the host vectors may be reused after this returns
    *num_direct = (int)(n0 / ctdet::pack_direct_item_bytes());
    *num_wino = (int)(n1 / ctdet::pack_wino_item_bytes());
    return CT_OK;
}

extern "C" int ct_pack_run(const void* table_dev, int num_direct, int num_wino, ct_stream_t stream)
{
    CT_REQUIRE(table_dev && num_direct >= 0 && num_wino >= 0, "ct_pack_run: bad arguments");
    hipStream_t st = ctdet::as_stream(stream);
    const unsigned char* t = static_cast<const unsigned char*>(table_dev);
    int rc = ctdet::launch_pack_direct_batched(t, num_direct, st);
    if (rc != CT_OK) return rc;
    return ctdet::launch_pack_wino_batched(t + ctdet::align_up((size_t)num_direct * ctdet::pack_direct_item_bytes(), 256),
                                           num_wino, st);
}

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
