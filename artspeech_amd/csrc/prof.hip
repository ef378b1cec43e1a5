// Optional in-library kernel timing: when enabled, every named phase of the composite entry points
// is bracketed by HIP events ON THE STREAM THE KERNELS ARE LAUNCHED ON; as_profile_report() waits for
// the stream's events and returns "name count total_ms" lines.  Disabled (the default) it costs one
// predictable branch per phase.  Used by bench.py for the roofline object; not for the timed `value`.
#include <map>
#include <string>
#include <vector>

#include "as_common.h"

namespace {
struct Rec { const char* name; hipEvent_t a, b; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;

hipEvent_t get_event() {
    if (g_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_next++];
}
}  // namespace

AsProfScope::AsProfScope(const char* name, hipStream_t st) : name_(name), st_(st), a_(nullptr), b_(nullptr) {
    if (!g_on) return;
    a_ = get_event();
    b_ = get_event();
    if (a_) (void)hipEventRecord((hipEvent_t)a_, st);
}
AsProfScope::~AsProfScope() {
    if (!a_ || !b_) return;
    (void)hipEventRecord((hipEvent_t)b_, st_);
    g_recs.push_back({name_, (hipEvent_t)a_, (hipEvent_t)b_});
}

extern "C" void as_profile_enable(int32_t on) { g_on = on != 0; }
bool as_profile_active() { return g_on; }
extern "C" void as_profile_reset(void) {
    g_recs.clear();
    g_next = 0;
}
extern "C" int32_t as_profile_report(char* buf, int32_t buflen) {
    std::map<std::string, std::pair<int, double>> acc;
    std::vector<std::string> order;
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = acc.find(r.name);
        if (it == acc.end()) {
            order.push_back(r.name);
            acc[r.name] = {1, ms};
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
    }
    std::string s;
    char line[256];
    for (const std::string& n : order) {
        snprintf(line, sizeof line, "%s %d %.6f\n", n.c_str(), acc[n].first, acc[n].second);
        s += line;
    }
    if (buf && buflen > 0) {
        snprintf(buf, (size_t)buflen, "%s", s.c_str());
    }
    return (int32_t)s.size();
}
