// Version / error plumbing of the C ABI (include/kvpress_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "kvp_common.h"

namespace {
thread_local char g_err[512] = "";
}

void kvp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Tuning knobs (KVP_* environment variables, used for A/B runs on hardware): every variable is read from the environment
// ONCE -- at its first use -- and then served from this table, so the launch path never calls getenv (a data race in glibc
// if another host thread calls setenv concurrently) and the launch geometry cannot change silently in the middle of a run.
// kvp_tuning_reload() drops the table: the next use of every knob re-reads the environment (tests, lab scripts).
#include <atomic>
#include <mutex>
#include <vector>
namespace {
struct Knob { const char* name; bool set; int value; };
std::mutex g_knob_mu;
std::vector<Knob> g_knobs;
}  // namespace
int kvp_env_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    for (const Knob& k : g_knobs)
        if (k.name == name || strcmp(k.name, name) == 0) return k.set ? k.value : dflt;
    const char* s = getenv(name);
    const Knob k{name, s && *s, (s && *s) ? atoi(s) : 0};
    g_knobs.push_back(k);
    return k.set ? k.value : dflt;
}
extern "C" int kvp_tuning_reload(void) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    g_knobs.clear();
    return KVP_OK;
}

// ---- asynchronous failure reports (kvp_common.h) ----------------------------------------------------------------------------
// One 64-byte host-pinned, device-mapped, coherent allocation per process, made at the first launch that may report through it
// (the only allocation this library ever makes on a launch path, once).  Kernels store a non-zero code with system scope; the
// host polls the word with plain loads -- no synchronisation, no stream traffic.
namespace {
std::mutex g_async_mu;
volatile uint32_t* g_async_host = nullptr;
uint32_t* g_async_dev = nullptr;
bool g_async_tried = false;
}  // namespace
uint32_t* kvp_async_flag() {
    std::lock_guard<std::mutex> lk(g_async_mu);
    if (!g_async_tried) {
        g_async_tried = true;
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && h) {
            memset(h, 0, 64);
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                g_async_host = static_cast<volatile uint32_t*>(h);
                g_async_dev = static_cast<uint32_t*>(d);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();   // a failed pinned allocation must not surface as some later launch's error
    }
    return g_async_dev;
}
uint32_t kvp_async_next_seq() {
    static std::atomic<uint32_t> seq{0};
    uint32_t v;
    do v = (seq.fetch_add(1, std::memory_order_relaxed) + 1) & 0xFFFFFFu; while (v == 0);
    return v;
}
int kvp_async_check(const char* who) {
    volatile uint32_t* f = g_async_host;   // (set once, under the mutex, before any kernel could have been handed the address)
    if (!f) return KVP_OK;
    if (*f == 0) return KVP_OK;
    // consume the report atomically: two host threads polling at once must not both see it, and a store that lands between a plain
    // read and a plain clear must not be lost
    const uint32_t word = __atomic_exchange_n(const_cast<uint32_t*>(f), 0u, __ATOMIC_ACQ_REL);
    if (word == 0) return KVP_OK;
    {
        // a late store of a launch whose failure was already reported (its other workgroups' stores came first): not a new event
        static std::atomic<uint32_t> last_reported{0};
        const uint32_t seq = word >> 8;
        if (seq != 0 && last_reported.exchange(seq, std::memory_order_acq_rel) == seq) return KVP_OK;
    }
    const uint32_t code = word & 0xFFu;
    kvp_set_error("%s: an EARLIER cluster select of this process (any thread) gave up at its barrier %u -- its 32 workgroups per row never "
                  "became co-resident within KVP_TC_TIMEOUT_US, or its 'clean' workspace still carried an earlier failure.  That call's "
                  "indices are -1 (gathered rows: NaN); zero-fill every workspace used with KVP_TOPK_WS_CLEAN since",
                  who, (unsigned)code);
    return KVP_EASYNC;
}
extern "C" int kvp_async_error_check(void) { return kvp_async_check("kvp_async_error_check"); }

extern "C" int kvp_version(void) { return KVP_VERSION; }
extern "C" const char* kvp_last_error(void) { return g_err; }

// ---- opt-in per-kernel timing with HIP events on the launch stream ------------------------------
// Not part of the reference boundary: a measurement aid for bench.py (roofline.achieved must be
// measured with HIP events on the stream the kernel is launched on).  Off by default; when on,
// every launch is bracketed by two hipEventRecord calls (which perturbs back-to-back timing, so
// the timed benchmark steps run with profiling off).
#include <string>
#include <vector>
namespace {
struct ProfRec { std::string name; hipEvent_t e0, e1; };
thread_local bool g_prof_on = false;
thread_local std::vector<ProfRec> g_prof;
void prof_clear() {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
}
}  // namespace

bool kvp_prof_enabled() { return g_prof_on; }
void kvp_prof_begin(const char* name, hipStream_t stream) {
    ProfRec r;
    r.name = name;
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, stream);
    g_prof.push_back(r);
}
void kvp_prof_end(hipStream_t stream) {
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().e1, stream);
}

extern "C" int kvp_prof_enable(int on) {
    prof_clear();
    g_prof_on = on != 0;
    return KVP_OK;
}
extern "C" int kvp_prof_count(void) { return (int)g_prof.size(); }
extern "C" int kvp_prof_get(int i, const char** name, float* ms) {
    if (i < 0 || i >= (int)g_prof.size()) { kvp_set_error("kvp_prof_get: index %d out of range", i); return KVP_EINVAL; }
    if (hipEventSynchronize(g_prof[i].e1) != hipSuccess || hipEventElapsedTime(ms, g_prof[i].e0, g_prof[i].e1) != hipSuccess) {
        kvp_set_error("kvp_prof_get: event query failed");
        return KVP_EHIP;
    }
    *name = g_prof[i].name.c_str();
    return KVP_OK;
}

// ---- shader-clock probe (measurement aid) ------------------------------------------------------------------------
// One wave spins for ~`spin_us` microseconds and reports shader-clock ticks (s_memtime) per 100 MHz real-time tick
// (s_memrealtime): the core clock in MHz at that point of the stream.  The clock governor moves slowly compared with a
// kernel, so a probe enqueued right behind a kernel shows the clock that kernel ran at.
namespace {
__global__ void clock_probe_kernel(float* out, uint32_t spin_ticks) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) *out = (float)((double)(c1 - c0) / (double)(r1 - r0) * 100.0);
}
}  // namespace
extern "C" int kvp_clock_probe(float* mhz_out, int spin_us, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(mhz_out && spin_us >= 1 && spin_us <= 100000, "clock_probe: bad arguments");
    clock_probe_kernel<<<1, 64, 0, stream>>>(mhz_out, (uint32_t)spin_us * 100u);
    KVP_CHECK_LAUNCH("clock_probe");
    return KVP_OK;
}

// ---- CU occupier (test aid) -------------------------------------------------------------------------------------------------
namespace {
__global__ void occupy_kernel(uint32_t spin_ticks) {
    extern __shared__ unsigned char occ_lds[];
    if (threadIdx.x == 0) occ_lds[0] = 1;   // the allocation is what matters
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < spin_ticks) __builtin_amdgcn_s_sleep(16);
}
}  // namespace
extern "C" int kvp_occupy_cus(int blocks, int threads, int lds_bytes, int spin_us, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(blocks >= 1 && blocks <= 65535 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && lds_bytes >= 0 &&
                      lds_bytes <= 160 * 1024 && spin_us >= 1 && spin_us <= 200000,
                  "occupy_cus: bad arguments");
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        kvp_set_error("occupy_cus: cannot raise the dynamic LDS limit to %d", lds_bytes);
        return KVP_EHIP;
    }
    occupy_kernel<<<blocks, threads, (size_t)std::max(lds_bytes, 16), stream>>>((uint32_t)spin_us * 100u);
    KVP_CHECK_LAUNCH("occupy_cus");
    return KVP_OK;
}
