// abi.cpp -- error reporting + version of libptmi355.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ptmi355.h"

static thread_local char g_err[512] = "";

void ptmi_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* ptmi_last_error(void) { return g_err; }
int ptmi_abi_version(void) { return 1; }
}
