// Host-side runtime pieces shared by the two hot paths: error string, device buffers, per-site
// HIP-event profiler.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

// the C ABI's opaque event (include/arp_hip.h)
struct arp_event {
    hipEvent_t e;
};

namespace arp {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        if (p) {
            ARP_HIP_OK(hipFree(p));
            p = nullptr;
            bytes = 0;
        }
        ARP_HIP_OK(hipMalloc(&p, need));
        bytes = need;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Brackets every launch of a call site with two HIP events on the launch stream; durations are
// summed per site name when read.  Off by default (the event packets cost a few microseconds of
// stream time per launch).
struct Profiler {
    bool on = false;
    struct Span { int site; hipEvent_t a, b; };
    std::vector<std::string> names;
    std::map<std::string, int> index;
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    std::vector<double> ms;
    std::vector<long long> calls;

    int site_id(const char* name) {
        auto it = index.find(name);
        if (it != index.end()) return it->second;
        const int id = (int)names.size();
        names.push_back(name);
        index[name] = id;
        ms.push_back(0.0);
        calls.push_back(0);
        return id;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    hipEvent_t begin(hipStream_t s) {
        hipEvent_t e = get_event();
        (void)hipEventRecord(e, s);
        return e;
    }
    void end(const char* name, hipEvent_t a, hipStream_t s) {
        hipEvent_t b = get_event();
        (void)hipEventRecord(b, s);
        spans.push_back(Span{site_id(name), a, b});
    }
    // synchronises; folds the recorded spans into ms / calls
    void collect() {
        for (auto& sp : spans) {
            (void)hipEventSynchronize(sp.b);
            float t = 0.f;
            (void)hipEventElapsedTime(&t, sp.a, sp.b);
            ms[sp.site] += t;
            calls[sp.site] += 1;
            pool.push_back(sp.a);
            pool.push_back(sp.b);
        }
        spans.clear();
    }
    void reset() {
        collect();
        for (auto& v : ms) v = 0.0;
        for (auto& v : calls) v = 0;
    }
    std::string json() {
        collect();
        std::string s = "{";
        for (size_t i = 0; i < names.size(); ++i) {
            if (i) s += ", ";
            s += "\"" + names[i] + "\": {\"ms\": " + std::to_string(ms[i]) + ", \"calls\": " + std::to_string(calls[i]) + "}";
        }
        return s + "}";
    }
    void destroy() {
        collect();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

// RAII-ish scope used as:  { ProfScope ps(prof, stream, "vit.fc1"); launch...; }
struct ProfScope {
    Profiler& p;
    hipStream_t s;
    const char* name;
    hipEvent_t a = nullptr;
    ProfScope(Profiler& p_, hipStream_t s_, const char* n) : p(p_), s(s_), name(n) {
        if (p.on) a = p.begin(s);
    }
    ~ProfScope() {
        if (p.on) p.end(name, a, s);
    }
};

// Pillow-exact resample table for one axis (precompute_coeffs + normalize_coeffs_8bpc of
// Pillow's Resample.c, bicubic a = -0.5).
struct ResampleTable {
    std::vector<int> xmin, cnt, w;  // w: [out][ksize]
    int ksize = 0, kmax = 0;
};
void build_bicubic_table(int in_size, int out_size, ResampleTable& t);

}  // namespace arp
