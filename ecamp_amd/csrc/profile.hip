// Optional in-process kernel timing with HIP events (used by bench.py for the roofline figure and by `--profile`): when
// enabled, the GEMM / attention host entries bracket each launch with a pair of events recorded on the launch stream.
// The number of live events is BOUNDED: pairs come from a pool of at most PROF_CAP; when the pool is exhausted the oldest
// half of the open records is folded into per-category running totals (their kernels finished long ago -- the fold only waits for
// the youngest of them) and their events go back to the free list.  A whole epoch under `--profile` therefore holds <= 2*PROF_CAP
// events however many steps it runs.
#include "common.h"
#include <deque>
#include <map>
#include <string>
#include <vector>

namespace {
constexpr size_t PROF_CAP = 4096;   // event pairs alive at any time (one training step records ~1 700)
constexpr int PROF_NCAT = 8;
struct ProfRec { hipEvent_t a, b; int cat; double work; char tag[40]; };
struct Totals { double ms = 0.0, work = 0.0; int64_t n = 0; };
std::deque<ProfRec> g_open;                          // recorded, not yet folded (in record order)
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
Totals g_tot[PROF_NCAT];
std::map<std::string, Totals> g_by_tag;              // per (form, epilogue, shape) totals of the tagged records (ecamp_prof_dump)
int g_prof_on = 0;
size_t g_created = 0;

void fold(size_t count) {
    for (size_t i = 0; i < count && !g_open.empty(); ++i) {
        ProfRec r = g_open.front();
        g_open.pop_front();
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cat >= 0 && r.cat < PROF_NCAT) {
            g_tot[r.cat].ms += t; g_tot[r.cat].work += r.work; g_tot[r.cat].n += 1;
            if (r.tag[0]) { Totals& x = g_by_tag[r.tag]; x.ms += t; x.work += r.work; x.n += 1; }
        }
        g_free.emplace_back(r.a, r.b);
    }
}
}  // namespace

extern "C" int ecamp_prof_enable(int on) {
    g_prof_on = on;
    return 0;
}
int ecamp_prof_active() { return g_prof_on; }
void ecamp_prof_begin(int cat, double work, hipStream_t s, const char* tag) {
    if (g_free.empty() && g_created >= PROF_CAP) fold(g_open.size() / 2 + 1);
    ProfRec r;
    r.cat = cat; r.work = work;
    r.tag[0] = 0;
    if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); r.tag[sizeof(r.tag) - 1] = 0; }
    if (!g_free.empty()) {
        r.a = g_free.back().first; r.b = g_free.back().second;
        g_free.pop_back();
    } else {
        hipEventCreate(&r.a);
        hipEventCreate(&r.b);
        ++g_created;
    }
    hipEventRecord(r.a, s);
    g_open.push_back(r);
}
void ecamp_prof_end(hipStream_t s) { hipEventRecord(g_open.back().b, s); }

// Sums elapsed ms, work (FLOPs) and launch count of category `cat` (waits for every open record first); cat < 0 clears the totals
// and releases the event pool.
extern "C" int ecamp_prof_collect(int cat, double* total_ms, double* total_work, int64_t* count) {
    fold(g_open.size());
    Totals t;
    if (cat >= 0 && cat < PROF_NCAT) t = g_tot[cat];
    if (cat < 0) {
        for (auto& e : g_free) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        g_free.clear();
        g_created = 0;
        for (auto& x : g_tot) x = Totals();
        g_by_tag.clear();
    }
    if (total_ms) *total_ms = t.ms;
    if (total_work) *total_work = t.work;
    if (count) *count = t.n;
    return 0;
}
// development aid (tools/gemm_in_step.py): the per-tag totals of the GEMM launches recorded since the last clear, one line per tag --
// "<tag> <launches> <total ms> <total FLOP>" -- into buf (NUL-terminated, truncated at cap); returns the bytes the full text needs
extern "C" int64_t ecamp_prof_dump(char* buf, int64_t cap) {
    fold(g_open.size());
    std::string out;
    char line[160];
    for (const auto& kv : g_by_tag) {
        snprintf(line, sizeof line, "%s %lld %.6f %.6e\n", kv.first.c_str(), (long long)kv.second.n, kv.second.ms, kv.second.work);
        out += line;
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}
// development aid: event pairs currently allocated (tests assert the bound)
extern "C" int64_t ecamp_prof_live_events(void) { return (int64_t)g_created; }
