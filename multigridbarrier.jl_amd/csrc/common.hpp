// common.hpp -- error handling, device buffers and stage timers shared by the library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgbhip {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct InvalidArgument : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define MGB_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            char _buf[512];                                                                  \
            snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                    \
            throw ::mgbhip::HipError(_buf);                                                  \
        }                                                                                    \
    } while (0)

#define MGB_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) throw ::mgbhip::InvalidArgument(msg);       \
    } while (0)

// Owning device buffer (hipMalloc / hipFree); never allocated inside a timed loop.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        if (count == n && p) return;
        release();
        if (count) MGB_HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
    void upload(const T* h, size_t count, hipStream_t st) {
        ensure(count);
        if (count) MGB_HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
    }
    void upload(const std::vector<T>& h, hipStream_t st) { upload(h.data(), h.size(), st); }
    void download(T* h, size_t count, hipStream_t st) const {
        if (count) MGB_HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, st));
    }
    void zero(hipStream_t st, size_t count = (size_t)-1) {
        if (count == (size_t)-1) count = n;
        if (count) MGB_HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), st));
    }
};

// Small pinned host block for scalar read-backs (a pageable destination sends every 8-byte
// copy through the runtime's staging path).
struct PinnedBuf {
    double* d = nullptr;        // 16 doubles
    int32_t* i = nullptr;       // 16 ints
    double* dev = nullptr;      // the same 16 doubles as the device sees them (finishing kernels store their results here)
    PinnedBuf() {
        void* p = nullptr;
        if (hipHostMalloc(&p, 16 * sizeof(double) + 16 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
            throw HipError("hipHostMalloc failed for the scalar read-back block");
        d = static_cast<double*>(p);
        i = reinterpret_cast<int32_t*>(d + 16);
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || dp == nullptr)
            throw HipError("hipHostGetDevicePointer failed for the scalar read-back block");
        dev = static_cast<double*>(dp);
    }
    ~PinnedBuf() { if (d) (void)hipHostFree(d); }
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
};

// hipEvent-based per-stage device timers on the handle's stream (bench.py's roofline
// numbers come from these; torch.cuda.Event would only see torch's own stream).
struct StageTimers {
    struct Rec { double ms = 0; int64_t launches = 0; };
    bool enabled = false;
    hipStream_t stream = nullptr;
    std::map<std::string, Rec> recs;
    struct Pending { std::string name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;

    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        MGB_HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    std::vector<size_t> open;      // indices into `pending` of the scopes not yet closed (they nest)
    void begin(const char* name) {
        if (!enabled) return;
        Pending p{name, get(), get()};
        MGB_HIP_CHECK(hipEventRecord(p.a, stream));
        pending.push_back(p);
        open.push_back(pending.size() - 1);
    }
    void end() {
        if (!enabled || open.empty()) return;
        MGB_HIP_CHECK(hipEventRecord(pending[open.back()].b, stream));
        open.pop_back();
        if (open.empty() && pending.size() > 4096) collect();
    }
    void collect() {
        if (!open.empty()) return;         // a scope is still running: its end event does not exist yet
        for (auto& p : pending) {
            MGB_HIP_CHECK(hipEventSynchronize(p.b));
            float ms = 0;
            MGB_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
            auto& r = recs[p.name];
            r.ms += ms;
            r.launches += 1;
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void reset(bool en) {
        collect();
        recs.clear();
        enabled = en;
    }
    ~StageTimers() {
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct StageScope {
    StageTimers& t;
    StageScope(StageTimers& tt, const char* name) : t(tt) { t.begin(name); }
    ~StageScope() {
        try { t.end(); } catch (...) {}
    }
};

}  // namespace mgbhip
