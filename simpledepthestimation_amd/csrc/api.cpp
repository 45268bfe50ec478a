// Error string + version for libsde_hip.so (see include/sde_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "sde_hip.h"

static thread_local char g_err[512] = "";

void sde_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sde_last_error(void) { return g_err; }
extern "C" int sde_version(void) { return 1; }
