// Per-class device-time brackets (HIP events on the launch stream) around everything the library launches while profiling is on
// (ctmi_profile_begin / ctmi_profile_end, include/ctmi355.h).  Off: one relaxed atomic load per entry point.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "../../include/ctmi355.h"

extern std::atomic<int> g_ctmi_prof_on;
int ctmi_prof_open(int cls, hipStream_t st);            // records the start event, returns the bracket index (-1: pool exhausted)
void ctmi_prof_close(int slot, hipStream_t st);

struct ProfScope {
    int slot = -1;
    hipStream_t st;
    ProfScope(int cls, hipStream_t s) : st(s) { if (g_ctmi_prof_on.load(std::memory_order_relaxed)) slot = ctmi_prof_open(cls, s); }
    ~ProfScope() { if (slot >= 0) ctmi_prof_close(slot, st); }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};
