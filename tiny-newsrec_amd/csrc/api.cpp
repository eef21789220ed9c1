// error plumbing + version for libtnr_hip.so
#include <stdarg.h>
#include <stdio.h>

#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void tnr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tnr_last_error(void) { return g_err; }
extern "C" int tnr_version(void) { return 1; }

static TnrGemmOpts g_gemm_opts = {3, 8, 60, 1, 0, 0, 1, 2, 1, 0, nullptr, 0};
TnrGemmOpts* tnr_gemm_opts() { return &g_gemm_opts; }

extern "C" int tnr_gemm_set_option(const char* key, int value) {
    if (!key) { tnr_set_error("tnr_gemm_set_option: null key"); return TNR_EINVAL; }
    TnrGemmOpts& o = g_gemm_opts;
    if (!strcmp(key, "ver")) o.ver = value;
    else if (!strcmp(key, "gm")) o.gm = value;
    else if (!strcmp(key, "fine_pct")) o.fine_pct = value;
    else if (!strcmp(key, "allow_fine")) o.allow_fine = value;
    else if (!strcmp(key, "bm")) o.bm = value;
    else if (!strcmp(key, "nt")) o.nt = value;
    else if (!strcmp(key, "pp")) o.pp = value;
    else if (!strcmp(key, "tnpp")) o.tnpp = value;
    else if (!strcmp(key, "mix")) o.mix = value;
    else if (!strcmp(key, "cus")) o.cus = value;
    else { tnr_set_error("tnr_gemm_set_option: unknown key %s", key); return TNR_EINVAL; }
    return TNR_OK;
}

extern "C" int tnr_gemm_clock_stamps(void* buf, int64_t n_pairs) {
    if (n_pairs < 0 || (buf && ((uintptr_t)buf % 8) != 0)) { tnr_set_error("tnr_gemm_clock_stamps: bad buffer"); return TNR_EINVAL; }
    g_gemm_opts.clock_buf = n_pairs > 0 ? buf : nullptr;
    g_gemm_opts.clock_n = buf ? (int)(n_pairs > (1 << 20) ? (1 << 20) : n_pairs) : 0;
    return TNR_OK;
}
