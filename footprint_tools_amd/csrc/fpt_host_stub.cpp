// fpt_host_stub.cpp -- stands in for fpt_capi.cpp in the HOST-ONLY sanitizer build (`make asan`:
// fpt_bam.cpp + fpt_text.cpp with -fsanitize=address,undefined, no HIP): the error channel of the
// C ABI and nothing else.  Never part of libfpt_hip.so.
#include <cstdarg>
#include <cstdio>

#include "../../include/fpt.h"

static thread_local char g_err[1024];

int fpt_internal_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" __attribute__((visibility("default"))) const char *fpt_last_error(void) { return g_err; }
