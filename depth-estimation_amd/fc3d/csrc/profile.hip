// Optional per-kernel timing with HIP events recorded on the launching stream.
// Off by default: a launch then costs one predictable branch.  When enabled every
// kernel launch of the library is bracketed by an event pair; fc3d_profile_report()
// synchronises those events and aggregates elapsed time per kernel name.
#include "fc3d_common.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace fc3d {

struct ProfRecord {
    const char* name;
    hipEvent_t a, b;
};

static bool g_on = false;
static std::mutex g_mu;
static std::vector<ProfRecord> g_used;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;

bool prof_enabled() { return g_on; }

void prof_before(const char* name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRecord r;
    r.name = name;
    if (!g_free.empty()) {
        r.a = g_free.back().first;
        r.b = g_free.back().second;
        g_free.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    }
    (void)hipEventRecord(r.a, s);
    g_used.push_back(r);
}

void prof_after(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_used.empty()) (void)hipEventRecord(g_used.back().b, s);
}

}  // namespace fc3d

using namespace fc3d;

extern "C" int fc3d_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return FC3D_OK;
}

extern "C" int fc3d_profile_report(char* buf, size_t len) {
    if (!buf || len == 0) return FC3D_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    std::map<std::string, std::pair<long, double>> agg;
    std::vector<std::string> order;
    for (auto& r : g_used) {
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            order.push_back(r.name);
            agg[r.name] = {1, (double)ms};
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
        g_free.push_back({r.a, r.b});
    }
    g_used.clear();
    size_t off = 0;
    buf[0] = 0;
    for (auto& n : order) {
        int w = snprintf(buf + off, len - off, "%s %ld %.6f\n", n.c_str(), agg[n].first, agg[n].second);
        if (w < 0 || (size_t)w >= len - off) return FC3D_ERANGE;
        off += (size_t)w;
    }
    return FC3D_OK;
}
