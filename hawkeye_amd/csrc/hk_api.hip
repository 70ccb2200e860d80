// Library identification, per-thread last-error text.
#include <stdarg.h>
#include <string.h>
#include "hk_common.h"

static thread_local char g_err[512] = "";

void hk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int hk_version(void) { return 100; }

extern "C" const char* hk_arch(void) { return "gfx950"; }

extern "C" size_t hk_last_error(char* buf, size_t cap) {
    const size_t n = strlen(g_err);
    if (buf && cap) {
        const size_t c = n < cap - 1 ? n : cap - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return n;
}
