// Optional in-process kernel timing with HIP events (used by bench.py for the roofline figure): when enabled, the
// GEMM / attention host entries bracket each launch with a pair of events recorded on the launch stream.
#include "common.h"
#include <vector>

struct ProfRec { hipEvent_t a, b; int cat; double work; };
static std::vector<ProfRec> g_recs;
static int g_prof_on = 0;

extern "C" int ecamp_prof_enable(int on) {
    g_prof_on = on;
    return 0;
}
int ecamp_prof_active() { return g_prof_on; }
void ecamp_prof_begin(int cat, double work, hipStream_t s) {
    ProfRec r;
    r.cat = cat; r.work = work;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void ecamp_prof_end(hipStream_t s) { hipEventRecord(g_recs.back().b, s); }

// Sums elapsed ms, work (FLOPs) and launch count of category `cat`; synchronises; clears when cat < 0.
extern "C" int ecamp_prof_collect(int cat, double* total_ms, double* total_work, int64_t* count) {
    double ms = 0, w = 0;
    int64_t n = 0;
    for (auto& r : g_recs) {
        if (cat >= 0 && r.cat != cat) continue;
        if (cat < 0) { hipEventDestroy(r.a); hipEventDestroy(r.b); continue; }
        hipEventSynchronize(r.b);
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        ms += t; w += r.work; n++;
    }
    if (cat < 0) g_recs.clear();
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = w;
    if (count) *count = n;
    return 0;
}
