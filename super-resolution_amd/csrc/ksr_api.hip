// Error channel and version for libksr.
#include "ksr_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ksr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ksr_last_error(void) { return g_err; }

extern "C" int ksr_version(void) { return 100; }
