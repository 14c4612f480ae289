// Thread-local error message + ABI version for libvince_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/vince_hip.h"

static thread_local char g_err[512] = "";

void vince_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// VINCE_KNOBS="name=value,name=value" (common.h): the product library's cross-check switches
#include <string.h>
long vince_knob(const char* name, long dflt) {
    const char* s = getenv("VINCE_KNOBS");
    if (!s) return dflt;
    const size_t n = strlen(name);
    while (*s) {
        while (*s == ',' || *s == ' ') ++s;
        if (strncmp(s, name, n) == 0 && s[n] == '=') return strtol(s + n + 1, nullptr, 10);
        while (*s && *s != ',') ++s;
    }
    return dflt;
}
#ifdef VINCE_MEASURE
#include <ctype.h>
long vince_measure_knob(const char* name, long dflt) {   // measurement build: VINCE_<NAME> in the environment
    char env[96] = "VINCE_";
    size_t i = 6;
    for (const char* c = name; *c && i + 1 < sizeof(env); ++c) env[i++] = (char)toupper((unsigned char)*c);
    env[i] = 0;
    const char* v = getenv(env);
    return v ? strtol(v, nullptr, 10) : dflt;
}
#endif

extern "C" const char* vince_last_error(void) { return g_err; }
extern "C" int vince_abi_version(void) { return VINCE_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------------------------
// Per-kernel event timing (bench.py's roofline leg): while enabled, the instrumented launchers bracket every launch
// with a hipEvent pair on the launch stream; vince_profile_collect() synchronises and sums duration / work per tag.
#include <hip/hip_runtime.h>
#include <vector>

namespace {
struct ProfRec { int tag; double work; hipEvent_t a, b; int dims[6]; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace

bool vince_profile_enabled() { return g_prof_on; }

void vince_profile_set_dims(void* token, int a, int b, int c, int d, int e, int f) {
    size_t idx = (size_t)(uintptr_t)token - 1;
    int* p = g_prof[idx].dims;
    p[0] = a; p[1] = b; p[2] = c; p[3] = d; p[4] = e; p[5] = f;
}

void vince_profile_set_tag(void* token, int tag) { g_prof[(size_t)(uintptr_t)token - 1].tag = tag; }

void vince_profile_begin_launch(int tag, double work, void* stream, void** token) {
    ProfRec r;
    r.tag = tag; r.work = work;
    for (int i = 0; i < 6; ++i) r.dims[i] = 0;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, (hipStream_t)stream);
    g_prof.push_back(r);
    *token = (void*)(uintptr_t)g_prof.size();
}

void vince_profile_end_launch(void* token, void* stream) {
    size_t idx = (size_t)(uintptr_t)token - 1;
    hipEventRecord(g_prof[idx].b, (hipStream_t)stream);
}

#include <atomic>
static std::atomic<long long> g_launches{0};
void vince_note_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t vince_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }

static int g_side_streams = 2;
int vince_side_stream_budget() { return g_side_streams; }

extern "C" int vince_set_side_streams(int32_t n) {
    g_side_streams = n < 0 ? 0 : (n > 2 ? 2 : n);
    return 0;
}

extern "C" int vince_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

extern "C" int vince_profile_collect(int32_t ntags, double* ms, double* work, int64_t* count) {
    for (int i = 0; i < ntags; ++i) { ms[i] = 0; work[i] = 0; count[i] = 0; }
    const char* dump = getenv("VINCE_PROFILE_DUMP");
    FILE* f = dump ? fopen(dump, "w") : nullptr;
    if (f) fprintf(f, "tag,M,Co,K,taps,stride,flags,us,tflops\n");
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float t = 0;
        hipEventElapsedTime(&t, r.a, r.b);
        if (f) fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.2f,%.1f\n", r.tag, r.dims[0], r.dims[1], r.dims[2], r.dims[3], r.dims[4],
                       r.dims[5], t * 1000.0, r.work / (t * 1e-3) / 1e12);
        if (r.tag >= 0 && r.tag < ntags) { ms[r.tag] += t; work[r.tag] += r.work; count[r.tag] += 1; }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    if (f) fclose(f);
    g_prof.clear();
    return 0;
}
