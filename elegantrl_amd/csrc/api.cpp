// Host-only part of liberl_hip.so: ABI version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include "erl_common.h"

static thread_local char g_err[512] = "";

void erl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int erl_abi_version(void) { return ERL_ABI_VERSION; }

extern "C" const char *erl_last_error_string(void) { return g_err; }

extern "C" int erl_device_info(int *num_cu, int *lds_bytes_per_block)
{
    int dev = 0;
    hipDeviceProp_t prop;
    int rc = erl_hip_status(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    rc = erl_hip_status(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc) return rc;
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.maxSharedMemoryPerMultiProcessor;
    return ERL_OK;
}
