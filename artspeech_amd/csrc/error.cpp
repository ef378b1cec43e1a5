// Error reporting and identification of libartspeech_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include "artspeech_hip.h"

static thread_local char g_err[512] = "";

void as_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* as_last_error(void) { return g_err; }
extern "C" const char* as_version(void) { return "artspeech_hip 0.1.0"; }
extern "C" const char* as_arch(void) { return "gfx950"; }
