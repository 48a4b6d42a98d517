// capi.cpp -- error plumbing of the C-ABI (include/racformer_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include <hip/hip_runtime.h>
#include "../../include/racformer_hip.h"

static thread_local char g_err[512] = "";

void rac_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rac_last_error(void) { return g_err; }
extern "C" int rac_abi_version(void) { return RAC_ABI_VERSION; }

// one bit per (device, kernel slot): set by the first caller, whoever it is
static std::atomic<unsigned long long> g_attr_done[64];

bool rac_first_use_on_device(int id)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || id < 0 || id >= 64)
        return true;                       // unknown device: set the attribute again, it is idempotent
    const unsigned long long bit = 1ull << id;
    return (g_attr_done[dev].fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}
