// libctdet: error reporting, ABI version and device query.
#include "ct_common.h"
#include <cstring>

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

}  // namespace ctdet

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
