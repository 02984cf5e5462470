// ctmi_profile_begin / ctmi_profile_end: see prof.h and include/ctmi355.h.
#include "common.h"
#include "prof.h"
#include <mutex>
#include <vector>

std::atomic<int> g_ctmi_prof_on{0};
namespace {
struct Bracket { hipEvent_t a, b; int cls; bool closed; };
std::mutex g_mu;
std::vector<Bracket> g_br;           // brackets of the current session (events are created once and reused by the next session)
size_t g_used = 0;
constexpr size_t MAX_BRACKETS = 1 << 16;
}

int ctmi_prof_open(int cls, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used >= MAX_BRACKETS) return -1;
    if (g_used == g_br.size()) {
        Bracket B{};
        if (hipEventCreate(&B.a) != hipSuccess || hipEventCreate(&B.b) != hipSuccess) return -1;
        g_br.push_back(B);
    }
    Bracket& B = g_br[g_used];
    B.cls = cls; B.closed = false;
    if (hipEventRecord(B.a, st) != hipSuccess) return -1;
    return (int)g_used++;
}
void ctmi_prof_close(int slot, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < 0 || (size_t)slot >= g_used) return;
    if (hipEventRecord(g_br[slot].b, st) == hipSuccess) g_br[slot].closed = true;
}

extern "C" int ctmi_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_used = 0;
    g_ctmi_prof_on.store(1, std::memory_order_relaxed);
    return CTMI_OK;
}
extern "C" int ctmi_profile_end(float* ms, int* launches) {
    CTMI_REQUIRE(ms != nullptr && launches != nullptr, "profile_end: null output");
    g_ctmi_prof_on.store(0, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_mu);
    for (int c = 0; c < CTMI_PROF_NCLASS; ++c) { ms[c] = 0.f; launches[c] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        Bracket& B = g_br[i];
        if (!B.closed || B.cls < 0 || B.cls >= CTMI_PROF_NCLASS) continue;
        if (hipEventSynchronize(B.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, B.a, B.b) == hipSuccess) { ms[B.cls] += t; launches[B.cls] += 1; }
    }
    g_used = 0;
    return CTMI_OK;
}
