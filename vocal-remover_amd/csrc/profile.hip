// Launch profiler behind VR_LAUNCH (vr_common.h): HIP events around every kernel launch of a profiled step, on the stream the
// kernel is launched on; per-kernel-name totals with the algorithmic FLOPs / bytes noted by the executor and the launch wrappers.
// Measurement infrastructure for bench.py's `roofline.classes` -- nothing here runs unless vr_profile_begin was called.
#include <cxxabi.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "vr_common.h"

namespace vr {

struct LaunchRec {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const void* fn = nullptr;
    const char* label = nullptr;
    double flops = 0, bytes = 0;
    bool noted = false;
    std::string tag;
};

struct LaunchProfiler {
    // g_launch_prof is thread-local: only the thread that opened the profile can clear its own pointer.  A handle closed from ANOTHER
    // thread marks the profiler dead and leaks it (a few KB) instead of freeing memory the opening thread's VR_LAUNCH would still touch.
    std::thread::id owner = std::this_thread::get_id();
    std::atomic<bool> dead{false};
    std::vector<LaunchRec> recs;
    bool pend = false, pend_strong = false;
    double pend_flops = 0, pend_bytes = 0;
    std::string pend_tag;
};

thread_local LaunchProfiler* g_launch_prof = nullptr;

void prof_note(double flops, double bytes, bool strong, const char* tag) {
    LaunchProfiler* p = g_launch_prof;
    if (!p) return;
    if (p->pend && p->pend_strong && !strong) return;          // the executor's figures win over a wrapper's
    p->pend = true; p->pend_strong = strong; p->pend_flops = flops; p->pend_bytes = bytes;
    p->pend_tag = tag ? tag : "";
}

void prof_note_update(double bytes, const char* tag) {
    LaunchProfiler* p = g_launch_prof;
    if (!p || !p->pend) return;
    p->pend_bytes = bytes;
    if (tag) p->pend_tag = tag;
}

void prof_note_clear() {
    if (g_launch_prof) g_launch_prof->pend = false;
}

void prof_before(const void* fn, const char* label, hipStream_t st) {
    LaunchProfiler* p = g_launch_prof;
    if (p->dead.load(std::memory_order_acquire)) { g_launch_prof = nullptr; return; }     // its handle was closed from another thread
    LaunchRec r;
    r.fn = fn; r.label = label;
    if (p->pend) { r.flops = p->pend_flops; r.bytes = p->pend_bytes; r.noted = true; r.tag = p->pend_tag; p->pend = false; }
    VR_HIP(hipEventCreate(&r.e0));
    VR_HIP(hipEventCreate(&r.e1));
    VR_HIP(hipEventRecord(r.e0, st));
    p->recs.push_back(r);
}

void prof_after(hipStream_t st) {
    LaunchProfiler* p = g_launch_prof;
    if (!p || p->recs.empty()) return;
    VR_HIP(hipEventRecord(p->recs.back().e1, st));
}

bool prof_owned_by_this_thread(const LaunchProfiler* p) { return p && p->owner == std::this_thread::get_id(); }
void prof_abandon(LaunchProfiler* p) { if (p) p->dead.store(true, std::memory_order_release); }

void prof_memset_async(void* ptr, int value, size_t bytes, hipStream_t st) {
    if (g_launch_prof) {
        prof_note(0.0, (double)bytes, false, nullptr);
        prof_before(nullptr, "memset", st);
    }
    VR_HIP(hipMemsetAsync(ptr, value, bytes, st));
    if (g_launch_prof) prof_after(st);
}

LaunchProfiler* prof_create() { return new LaunchProfiler(); }

void prof_destroy(LaunchProfiler* p) {
    if (!p) return;
    for (LaunchRec& r : p->recs) { if (r.e0) hipEventDestroy(r.e0); if (r.e1) hipEventDestroy(r.e1); }
    delete p;
}

static std::string kernel_name(const LaunchRec& r) {
    if (!r.fn) return r.label ? r.label : "?";
    const char* mangled = hipKernelNameRefByPtr(r.fn, nullptr);
    if (!mangled) return r.label ? r.label : "?";
    int status = 0;
    char* d = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    std::string s = (status == 0 && d) ? d : mangled;
    std::free(d);
    const size_t paren = s.find('(');                           // drop the parameter list
    if (paren != std::string::npos) s.resize(paren);
    if (s.rfind("void ", 0) == 0) s.erase(0, 5);
    return s;
}

// Aggregate the recorded launches (all events must have completed: the caller synchronised the device).
//   conv_*: the launches that carry a STRONG note with FLOPs (the executor's convolutions)
//   report: one line per kernel name  "name\tcalls\tms\tflops\tbytes\tnoted_calls\n"  (FLOPs / bytes summed over the noted launches)
void prof_collect(LaunchProfiler* p, double* conv_ms, double* conv_flops, double* conv_bytes, int* conv_launches, std::string* report,
                  bool dump) {
    struct Agg { int calls = 0, noted = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> by_name;
    double cm = 0, cf = 0, cb = 0;
    int cn = 0;
    for (LaunchRec& r : p->recs) {
        float ms = 0.f;
        VR_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        const std::string name = kernel_name(r);
        Agg& a = by_name[name];
        a.calls++; a.ms += ms;
        if (r.noted) { a.noted++; a.flops += r.flops; a.bytes += r.bytes; }
        if (r.noted && r.flops > 0 && !r.tag.empty()) { cm += ms; cf += r.flops; cb += r.bytes; ++cn; }
        if (dump) fprintf(stderr, "[vr-prof] %-58s %-44s %9.1f us %8.2f GFLOP %8.2f MB\n", name.c_str(), r.tag.c_str(), ms * 1e3,
                          r.flops * 1e-9, r.bytes * 1e-6);
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
        r.e0 = r.e1 = nullptr;
    }
    p->recs.clear();
    if (conv_ms) *conv_ms = cm;
    if (conv_flops) *conv_flops = cf;
    if (conv_bytes) *conv_bytes = cb;
    if (conv_launches) *conv_launches = cn;
    if (report) {
        std::ostringstream os;
        os.precision(9);
        for (auto& kv : by_name)
            os << kv.first << '\t' << kv.second.calls << '\t' << kv.second.ms << '\t' << kv.second.flops << '\t' << kv.second.bytes << '\t'
               << kv.second.noted << '\n';
        *report = os.str();
    }
}

}  // namespace vr
