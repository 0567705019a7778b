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

// One synchronous host -> device copy on the NULL stream before the process creates its first stream (round 6, profiles/r6_n1_flow.txt / r6_n1_flow3.txt).
// Measured on one box, same binary, same kernels: the encoder-inside policy step with two encoder part streams + the encode-ahead stream runs 11.7 ms per
// step when the handle's hipStreamCreate is the process's FIRST HIP work after device selection, and 10.05 ms when any hipMemcpy(H2D) came before it -- an encoder
// created first (its weight uploads), the bench's parity gate, or just this copy; a hipMalloc / hipFree alone does not do it, GPU_MAX_HW_QUEUES 4 / 8 / 16 does
// not explain it.  The runtime sets up the null stream's hardware queue (and its copy machinery) at that first copy; which hardware queue the handle's own streams
// then land on relative to it is what differs.  Every *_create calls this once per device before it creates a stream, so the fast arrangement is the only one.
inline int prime_runtime(int device) {
    static bool done[64] = {false};
    if (device < 0 || device >= 64 || done[device]) return 0;
    void* d = nullptr;
    const size_t n = 16u << 20;
    std::vector<char> h(n, 0);
    ARP_HIP_OK(hipMalloc(&d, n));
    const hipError_t e = hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    (void)hipDeviceSynchronize();
    (void)hipFree(d);
    if (e != hipSuccess) return fail(std::string("prime_runtime: hipMemcpy failed: ") + hipGetErrorString(e));
    done[device] = true;
    return 0;
}

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
