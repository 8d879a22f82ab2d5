// msq_host.h -- host-side element-format table (formats.py:65-129 _get_format_params
// plus the posit<n,es> extension).  No GPU needed.
#pragma once
#include <math.h>
#include <string.h>
#include "../../include/msq.h"

#include <atomic>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
namespace msq_host {

struct FmtInfo { int kind, ebits, mbits, emax; float max_norm, min_norm; };

inline bool format_info(int fmt, FmtInfo* o) {
    int e = 0, m = 0, ex = 0; bool e4m3 = false;
    if (fmt & 0x100) {                                   // MSQ_FMT_POSIT(n, es)
        const int n = (fmt & 0xFF) >> 2, es = fmt & 3;
        if (n < 3 || n > 16) return false;
        o->kind = 1; o->ebits = es; o->mbits = n; o->emax = 1;
        o->max_norm = (float)ldexp(1.0, (1 << es) * (n - 2));
        o->min_norm = (float)ldexp(1.0, -(1 << es) * (n - 2));
        return true;
    }
    switch (fmt) {
        case MSQ_FMT_INT8: e = 0; m = 8; ex = 0; break;
        case MSQ_FMT_INT4: e = 0; m = 4; ex = 0; break;
        case MSQ_FMT_INT2: e = 0; m = 2; ex = 0; break;
        case MSQ_FMT_FP8_E5M2: e = 5; m = 4; ex = 15; break;
        case MSQ_FMT_FP8_E4M3: e = 4; m = 5; ex = 8; e4m3 = true; break;
        case MSQ_FMT_FP6_E3M2: e = 3; m = 4; ex = 4; break;
        case MSQ_FMT_FP6_E2M3: e = 2; m = 5; ex = 2; break;
        case MSQ_FMT_FP4_E2M1: e = 2; m = 3; ex = 2; break;
        case MSQ_FMT_FP16: e = 5; m = 12; ex = 15; break;
        case MSQ_FMT_BF16: e = 8; m = 9; ex = 127; break;
        default: return false;
    }
    o->kind = 0; o->ebits = e; o->mbits = m; o->emax = ex;
    const double mx = e4m3 ? ldexp(1.0, ex) * 1.75
                           : ldexp(1.0, ex) * (double)((1 << (m - 1)) - 1) / ldexp(1.0, m - 2);
    o->max_norm = (float)mx;
    o->min_norm = (e == 0) ? 0.0f : (float)ldexp(1.0, 2 - (1 << (e - 1)));
    return true;
}

inline int format_id(const char* name) {
    if (!name) return MSQ_ERR_BAD_ARG;
    char s[32]; size_t n = strlen(name);
    if (n >= sizeof(s)) return MSQ_ERR_BAD_ARG;
    for (size_t i = 0; i <= n; ++i) s[i] = (name[i] >= 'A' && name[i] <= 'Z') ? name[i] + 32 : name[i];
    if (!strcmp(s, "int8")) return MSQ_FMT_INT8;
    if (!strcmp(s, "int4")) return MSQ_FMT_INT4;
    if (!strcmp(s, "int2")) return MSQ_FMT_INT2;
    if (!strcmp(s, "fp8_e5m2")) return MSQ_FMT_FP8_E5M2;
    if (!strcmp(s, "fp8_e4m3")) return MSQ_FMT_FP8_E4M3;
    if (!strcmp(s, "fp6_e3m2")) return MSQ_FMT_FP6_E3M2;
    if (!strcmp(s, "fp6_e2m3")) return MSQ_FMT_FP6_E2M3;
    if (!strcmp(s, "fp4") || !strcmp(s, "fp4_e2m1")) return MSQ_FMT_FP4_E2M1;
    if (!strcmp(s, "float16") || !strcmp(s, "fp16")) return MSQ_FMT_FP16;
    if (!strcmp(s, "bfloat16") || !strcmp(s, "bf16")) return MSQ_FMT_BF16;
    if (!strncmp(s, "posit", 5)) {
        int pn = 0, es = 0; const char* p = s + 5;
        if (*p < '0' || *p > '9') return MSQ_ERR_BAD_ARG;
        while (*p >= '0' && *p <= '9') pn = pn * 10 + (*p++ - '0');
        if (strncmp(p, "_es", 3)) return MSQ_ERR_BAD_ARG;
        p += 3;
        if (*p < '0' || *p > '9') return MSQ_ERR_BAD_ARG;
        while (*p >= '0' && *p <= '9') es = es * 10 + (*p++ - '0');
        if (*p || pn < 3 || pn > 16 || es > 3) return MSQ_ERR_BAD_ARG;
        return MSQ_FMT_POSIT(pn, es);
    }
    return MSQ_ERR_BAD_ARG;
}


// A tuning switch: an atomic that msq_set_tuning() can set (thread-safe, wins), else the environment variable of the same name read per call
// (single-threaded tests and A / B scripts flip it inside one process).  value(): the override, or atoi(env), or `unset`.
struct TuneKey {
    const char* name; std::atomic<int> v{INT_MIN};
    explicit TuneKey(const char* n) : name(n) {}
    int value(int unset) const { const int t = v.load(std::memory_order_relaxed); if (t != INT_MIN) return t; const char* e = getenv(name); return e ? atoi(e) : unset; }
    bool is_set() const { return v.load(std::memory_order_relaxed) != INT_MIN || getenv(name) != nullptr; }
    bool set_if(const char* key, int value) { if (strcmp(key, name)) return false; v.store(value, std::memory_order_relaxed); return true; }
};
}  // namespace msq_host
