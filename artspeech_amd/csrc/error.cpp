// Error reporting and identification of libartspeech_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

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

// ---- matrix arithmetic mode (include/artspeech_hip.h: as_set_matrix_arith).  -1 = not decided yet (environment).
static std::atomic<int> g_arith{-1};

int as_matrix_arith() {
    int m = g_arith.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("ARTSPEECH_MATRIX_ARITH");
        m = (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) ? 0 : 1;
        g_arith.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" void as_set_matrix_arith(int32_t mode) { g_arith.store(mode ? 1 : 0, std::memory_order_relaxed); }
extern "C" int32_t as_get_matrix_arith(void) { return as_matrix_arith(); }

// ---- see gemm_internal.h: the event the next knowing kernel launch of this thread binds to its completion
static thread_local void* g_stop_event = nullptr;
void as_stop_event_set_raw(void* ev) { g_stop_event = ev; }
void* as_stop_event_take_raw() {
    void* e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}
