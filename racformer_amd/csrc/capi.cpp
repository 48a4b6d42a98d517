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

// Dynamic-LDS limits are function attributes per device.  One slot per (device, kernel): the attribute is set under a lock by
// the first caller and the slot is marked only AFTER hipFuncSetAttribute has succeeded -- a second thread on the same device either
// sees the mark (the attribute is in place) or waits on the lock, and a failed set is reported and retried by the next call (the
// round-3 form marked the slot before the attribute was set and dropped its return code).
#include <mutex>
static std::atomic<unsigned long long> g_attr_done[64];
static std::mutex g_attr_lock;

int rac_set_dynamic_lds_once(int id, const void *func, int bytes)
{
    int dev = 0;
    const bool tracked = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && id >= 0 && id < 64;
    const unsigned long long bit = 1ull << (id & 63);
    if (tracked && (g_attr_done[dev].load(std::memory_order_acquire) & bit))
        return 0;
    std::lock_guard<std::mutex> hold(g_attr_lock);
    if (tracked && (g_attr_done[dev].load(std::memory_order_acquire) & bit))
        return 0;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        rac_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) for kernel slot %d: %s", bytes, id, hipGetErrorString(e));
        return (int)e;
    }
    if (tracked)
        g_attr_done[dev].fetch_or(bit, std::memory_order_release);
    return 0;
}
