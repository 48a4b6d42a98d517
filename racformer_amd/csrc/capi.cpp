// capi.cpp -- error plumbing of the C-ABI (include/racformer_hip.h).
#include <stdarg.h>
#include <stdio.h>
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
