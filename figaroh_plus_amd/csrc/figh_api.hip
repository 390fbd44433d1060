// Runtime plumbing of libfigh.so: device selection, memory, the library stream, per-kernel event timing
// and the kinematic-tree handle.  No compute lives here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "figh_internal.h"

namespace figh {

static thread_local std::string g_error;
static hipStream_t g_stream = nullptr;
static bool g_ready = false;
static int g_profile = 0;  // 0 off, 1 the dominant kernels only (cheap enough for a timed region), 2 every launch

struct ProfileAcc {
    int launches = 0;
    double ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static std::map<std::string, ProfileAcc> g_prof;
static std::vector<hipEvent_t> g_event_pool;  // events are recycled: creating one costs far more than recording it

static hipEvent_t acquire_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static void *g_ws[40] = {nullptr};
static size_t g_ws_bytes[40] = {0};

void set_error(const std::string &msg) { g_error = msg; }

hipStream_t stream() { return g_stream; }

static bool g_blocking_wait = false;
static double g_null_pivot_sq = 0.0;

double null_pivot_sq() { return g_null_pivot_sq; }

int ensure_device() {
    if (g_ready) return FIGH_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(std::string("no HIP device available (") + hipGetErrorString(e) +
                  "); libfigh has no CPU path");
        return FIGH_ERR_NO_DEVICE;
    }
    if (g_blocking_wait) {
        // figh_host_wait_mode(1) without figh_device_set: the flag must precede the creation of the device's context.  A
        // process that already has one (another library initialised HIP first) keeps its mode -- say so instead of
        // silently spinning
        const hipError_t fe = hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            std::fprintf(stderr, "libfigh: interrupt-driven host waits were requested but the device context already "
                                 "exists (%s); the host keeps spinning\n", hipGetErrorString(fe));
        }
    }
    FIGH_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_ready = true;
    return FIGH_OK;
}

void *workspace(size_t bytes, int slot) {
    if (bytes <= g_ws_bytes[slot]) return g_ws[slot];
    if (g_ws[slot]) {
        (void)hipStreamSynchronize(g_stream);
        (void)hipFree(g_ws[slot]);
        g_ws[slot] = nullptr;
        g_ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&g_ws[slot], want) != hipSuccess) {
        set_error("workspace allocation failed");
        return nullptr;
    }
    g_ws_bytes[slot] = want;
    return g_ws[slot];
}

static hipStream_t g_side_stream = nullptr;
static hipEvent_t g_fork_event = nullptr, g_join_event = nullptr;

SideStream::SideStream() {
    if (!g_side_stream) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return;
        if (hipStreamCreateWithPriority(&g_side_stream, hipStreamNonBlocking, greatest) != hipSuccess ||
            hipEventCreateWithFlags(&g_fork_event, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g_join_event, hipEventDisableTiming) != hipSuccess) {
            g_side_stream = nullptr;
            (void)hipGetLastError();
            return;
        }
    }
    main_ = g_stream;
    if (hipEventRecord(g_fork_event, main_) != hipSuccess || hipStreamWaitEvent(g_side_stream, g_fork_event, 0) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    g_stream = g_side_stream;
    ok_ = open_ = true;
}

hipEvent_t SideStream::finish() {
    if (!open_) return nullptr;
    open_ = false;
    const hipError_t e = hipEventRecord(g_join_event, g_side_stream);
    g_stream = main_;
    if (e != hipSuccess) {  // (cannot order the streams by an event: wait for the side work here)
        (void)hipGetLastError();
        (void)hipStreamSynchronize(g_side_stream);
        return nullptr;
    }
    return g_join_event;
}

SideStream::~SideStream() {
    if (open_) stream_wait(finish());
}

void stream_wait(hipEvent_t ev) {
    if (ev && hipStreamWaitEvent(g_stream, ev, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(g_side_stream);
    }
}

static LaunchEvents g_launch_events;
static bool g_launch_events_taken = false;

ProfileScope::ProfileScope(const char *name, bool at_launch) : name_(name), at_launch_(at_launch) {
    if (!g_profile) return;
    if (g_profile == 1 && std::strcmp(name, "regressor_chain") != 0 && std::strcmp(name, "regressor_tree") != 0 &&
        std::strcmp(name, "tsqr") != 0 && std::strcmp(name, "fused_chain_tsqr") != 0)
        return;
    e0_ = acquire_event();
    e1_ = acquire_event();
    if (!e0_ || !e1_) return;
    if (at_launch_) {
        g_launch_events.start = e0_;
        g_launch_events.stop = e1_;
        g_launch_events_taken = false;
    } else {
        (void)hipEventRecord(e0_, g_stream);
    }
}

ProfileScope::~ProfileScope() {
    if (!e0_ || !e1_) return;
    if (at_launch_) {
        const bool used = g_launch_events_taken && g_launch_events.start == nullptr;
        g_launch_events = LaunchEvents();
        if (!used) {  // nothing was launched through FIGH_LAUNCH_TIMED inside the scope
            g_event_pool.push_back(e0_);
            g_event_pool.push_back(e1_);
            return;
        }
    } else {
        (void)hipEventRecord(e1_, g_stream);
    }
    g_prof[name_].pending.emplace_back(e0_, e1_);
}

LaunchEvents take_launch_events() {
    const LaunchEvents ev = g_launch_events;
    if (ev.start) g_launch_events_taken = true;
    g_launch_events = LaunchEvents();
    return ev;
}

static void drain_profile() {
    for (auto &kv : g_prof) {
        for (auto &p : kv.second.pending) {
            (void)hipEventSynchronize(p.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.launches += 1;
            }
            g_event_pool.push_back(p.first);
            g_event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

}  // namespace figh

using namespace figh;

extern "C" {

int figh_version(void) { return FIGH_ABI_VERSION; }

const char *figh_last_error(void) { return g_error.c_str(); }

int figh_device_count(int *count) {
    FIGH_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return FIGH_OK;
}

int figh_device_set(int device) {
    FIGH_REQUIRE(!g_ready, "figh_device_set must be called before any other device call");
    FIGH_HIP(hipSetDevice(device));
    return ensure_device();  // (applies the host wait mode before the context is created)
}

int figh_device_pci_bus_id(int device, char *out, int out_len) {
    FIGH_REQUIRE(out && out_len >= 16, "figh_device_pci_bus_id: buffer of at least 16 bytes");
    hipError_t e = hipDeviceGetPCIBusId(out, out_len, device);  // creates no context
    if (e != hipSuccess) {
        set_error(std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e));
        return FIGH_ERR_NO_DEVICE;
    }
    return FIGH_OK;
}

int figh_tsqr_null_pivot_tol(double tol) {
    FIGH_REQUIRE(tol >= 0.0 && tol < 1.0, "figh_tsqr_null_pivot_tol: 0 <= tol < 1");
    g_null_pivot_sq = tol * tol;
    return FIGH_OK;
}

int figh_host_wait_mode(int blocking) {
    FIGH_REQUIRE(!g_ready, "figh_host_wait_mode must be called before any device call");
    g_blocking_wait = blocking != 0;
    return FIGH_OK;
}

int figh_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
    if (int rc = ensure_device()) return rc;
    int dev = 0;
    FIGH_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FIGH_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) {
        std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return FIGH_OK;
}

#ifdef FIGH_ABLATION
static std::vector<void *> g_vmm_bases;  // FIGH_MALLOC_VMM probe allocations (left mapped)
#endif

int figh_malloc(void **d_ptr, size_t bytes) {
    FIGH_REQUIRE(d_ptr, "d_ptr is NULL");
    if (int rc = ensure_device()) return rc;
    hipError_t e = hipErrorUnknown;
#ifdef FIGH_ABLATION
    // FIGH_MALLOC_CONTIG: large buffers from physically contiguous memory (tools/k1_alloc_probe.py: does K1's dependence on
    // the placement of W go away?)
    static const bool contig = getenv("FIGH_MALLOC_CONTIG") != nullptr;
    if (contig && bytes >= (64u << 20)) {
        e = hipExtMallocWithFlags(d_ptr, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            std::fprintf(stderr, "libfigh: contiguous allocation of %zu bytes failed (%s), plain hipMalloc\n", bytes, hipGetErrorString(e));
        }
    }
#endif
#ifdef FIGH_ABLATION
    // FIGH_MALLOC_VMM=<chunk MB>[,<order>]: large buffers as one virtual range backed by separately created physical chunks
    // (order 0: in sequence, 1: reversed) -- does scattering the pages make K1 fast every time?
    // (never freed: probe only)
    if (e != hipSuccess && bytes >= (64u << 20)) {
        if (const char *env = getenv("FIGH_MALLOC_VMM")) {
            int chunk_mb = 64, order = 0;
            sscanf(env, "%d,%d", &chunk_mb, &order);
            int dev = 0;
            (void)hipGetDevice(&dev);
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = dev;
            size_t gran = 0;
            hipError_t ve = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
            size_t chunk = (size_t)chunk_mb << 20;
            if (ve == hipSuccess && gran) chunk = (chunk + gran - 1) / gran * gran;
            const size_t nchunk = (bytes + chunk - 1) / chunk;
            void *base = nullptr;
            if (ve == hipSuccess) ve = hipMemAddressReserve(&base, nchunk * chunk, 0, nullptr, 0);
            std::vector<hipMemGenericAllocationHandle_t> hs(nchunk);
            for (size_t k = 0; ve == hipSuccess && k < nchunk; ++k) ve = hipMemCreate(&hs[k], chunk, &prop, 0);
            for (size_t k = 0; ve == hipSuccess && k < nchunk; ++k) {
                const size_t src = order == 1 ? nchunk - 1 - k : k;
                ve = hipMemMap(static_cast<char *>(base) + k * chunk, chunk, 0, hs[src], 0);
            }
            if (ve == hipSuccess) {
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                ve = hipMemSetAccess(base, nchunk * chunk, &acc, 1);
            }
            if (ve == hipSuccess) {
                *d_ptr = base;
                e = hipSuccess;
                g_vmm_bases.push_back(base);
            } else {
                (void)hipGetLastError();
                std::fprintf(stderr, "libfigh: VMM allocation failed (%s), plain hipMalloc\n", hipGetErrorString(ve));
            }
        }
    }
#endif
    if (e != hipSuccess) e = hipMalloc(d_ptr, bytes ? bytes : 8);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return FIGH_ERR_ALLOC;
    }
    return FIGH_OK;
}

int figh_free(void *d_ptr) {
    if (!d_ptr) return FIGH_OK;
    if (int rc = ensure_device()) return rc;
    FIGH_HIP(hipStreamSynchronize(g_stream));
#ifdef FIGH_ABLATION
    for (void *b : g_vmm_bases)
        if (b == d_ptr) return FIGH_OK;
#endif
    FIGH_HIP(hipFree(d_ptr));
    return FIGH_OK;
}

// Small transfers (index lists, column norms, the n x n triangle) go through a pinned staging buffer: a pageable
// hipMemcpyAsync is staged by the runtime anyway and costs 20-30 us per call, three to five times per pass.
static void *g_pinned = nullptr;
static const size_t kPinnedBytes = 1 << 20;

static void *pinned_staging() {
    if (!g_pinned && hipHostMalloc(&g_pinned, kPinnedBytes, hipHostMallocDefault) != hipSuccess) g_pinned = nullptr;
    return g_pinned;
}

static bool is_pinned(const void *p, size_t bytes);

// Bulk transfers between PAGEABLE host memory and HBM (the drop-in boundary: the reference's functions take and return NumPy
// arrays, src/figaroh/tools/regressor.py:20-194 -- a 4 GB W for UR10 at 1e6 samples): chunks of 32 MB go through two page-locked
// staging buffers, the DMA of chunk k + 1 overlapping the host-side copy of chunk k, which eight threads share -- they also
// take the page faults of a freshly allocated destination in parallel.  Measured on the GPU box (tools/microbench/d2h_probe*.hip,
// profiles/r06_d2h_probe.txt): one hipMemcpyAsync into fresh pageable memory 12 - 24 GB/s, this path into a fresh
// huge-page-backed buffer 48 GB/s (device.py allocates such buffers), into a page-locked buffer (figh_host_alloc) the plain
// DMA reaches 57 GB/s.
namespace {
constexpr size_t kBulkChunk = size_t(32) << 20;
constexpr size_t kBulkMin = size_t(24) << 20;
constexpr int kBulkThreads = 8;
char *g_bulk_stage[2] = {nullptr, nullptr};
hipEvent_t g_bulk_ev[2];

bool bulk_ready() {
    if (g_bulk_stage[0]) return true;
    void *a = nullptr, *b = nullptr;
    if (hipHostMalloc(&a, kBulkChunk, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipHostMalloc(&b, kBulkChunk, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(a); return false; }
    if (hipEventCreateWithFlags(&g_bulk_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_bulk_ev[1], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); (void)hipHostFree(a); (void)hipHostFree(b);
        return false;
    }
    g_bulk_stage[0] = static_cast<char *>(a);
    g_bulk_stage[1] = static_cast<char *>(b);
    return true;
}

// host <-> device in staged chunks; to_host: d -> h, else h -> d.  Returns a hipError_t.
hipError_t bulk_copy(char *h, char *d, const size_t bytes, const bool to_host) {
    const long nch = (long)((bytes + kBulkChunk - 1) / kBulkChunk);
    const int nt = (int)std::max(1u, std::min<unsigned>(kBulkThreads, std::thread::hardware_concurrency()));
    std::atomic<long> go{to_host ? 0 : 2};  // to_host: chunks < go sit in staging; else: chunks < go may be filled
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nch]);  // per chunk: threads that have copied their slice
    for (long k = 0; k < nch; ++k) done[k].store(0, std::memory_order_relaxed);
    std::atomic<bool> abort{false};
    auto span = [&](long k, size_t &lo, size_t &n) { lo = (size_t)k * kBulkChunk; n = std::min(kBulkChunk, bytes - lo); };
    auto worker = [&](int t) {
        for (long k = 0; k < nch; ++k) {
            while (go.load(std::memory_order_acquire) <= k) {
                if (abort.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            size_t lo, n;
            span(k, lo, n);
            const size_t per = ((n / nt) + 4095) & ~size_t(4095);
            const size_t a = std::min(n, per * (size_t)t), b = std::min(n, a + per);
            if (b > a) {
                if (to_host) std::memcpy(h + lo + a, g_bulk_stage[k & 1] + a, b - a);
                else std::memcpy(g_bulk_stage[k & 1] + a, h + lo + a, b - a);
            }
            done[k].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 0; t < nt; ++t) pool.emplace_back(worker, t);
    } catch (...) {  // (thread limit of the container: the caller takes the one-piece copy instead)
        abort.store(true);
        for (auto &t : pool) t.join();
        return hipErrorNotReady;
    }
    // (per CHUNK: a global count of thread-chunks would let a fast thread's next chunk stand in for a slow thread's current one)
    auto wait_chunk = [&](long k) { while (done[k].load(std::memory_order_acquire) < nt) std::this_thread::yield(); };
    hipError_t err = hipSuccess;
    auto fail = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) { err = e; abort.store(true); } return e != hipSuccess; };
    size_t lo, n;
    if (to_host) {
        span(0, lo, n);
        if (!fail(hipMemcpyAsync(g_bulk_stage[0], d, n, hipMemcpyDeviceToHost, g_stream))) fail(hipEventRecord(g_bulk_ev[0], g_stream));
        for (long k = 0; k < nch && err == hipSuccess; ++k) {
            if (k + 1 < nch) {
                if (k >= 1) wait_chunk(k - 1);  // chunk k - 1 has left the buffer chunk k + 1 lands in
                span(k + 1, lo, n);
                if (fail(hipMemcpyAsync(g_bulk_stage[(k + 1) & 1], d + lo, n, hipMemcpyDeviceToHost, g_stream))) break;
                if (fail(hipEventRecord(g_bulk_ev[(k + 1) & 1], g_stream))) break;
            }
            if (fail(hipEventSynchronize(g_bulk_ev[k & 1]))) break;
            go.store(k + 1, std::memory_order_release);
        }
    } else {
        for (long k = 0; k < nch && err == hipSuccess; ++k) {
            wait_chunk(k);  // chunk k is in its staging buffer
            span(k, lo, n);
            if (fail(hipMemcpyAsync(d + lo, g_bulk_stage[k & 1], n, hipMemcpyHostToDevice, g_stream))) break;
            if (fail(hipEventRecord(g_bulk_ev[k & 1], g_stream))) break;
            if (k >= 1) {
                if (fail(hipEventSynchronize(g_bulk_ev[(k - 1) & 1]))) break;  // chunk k - 1 has left its buffer: chunk k + 1 may be filled
                go.store(k + 2, std::memory_order_release);
            }
        }
    }
    if (err != hipSuccess) abort.store(true);
    for (auto &t : pool) t.join();
    const hipError_t e2 = hipStreamSynchronize(g_stream);
    return err != hipSuccess ? err : e2;
}
}  // namespace

int figh_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes) {
    if (int rc = ensure_device()) return rc;
    if (bytes >= kBulkMin && !is_pinned(h_src, bytes) && bulk_ready()) {
        FIGH_HIP(hipStreamSynchronize(g_stream));  // (the staging buffers may still feed an earlier copy)
        const hipError_t e = bulk_copy(const_cast<char *>(static_cast<const char *>(h_src)), static_cast<char *>(d_dst), bytes, false);
        if (e != hipErrorNotReady) {  // (hipErrorNotReady: the copy threads could not be started -- nothing was copied)
            FIGH_HIP(e);
            return FIGH_OK;
        }
    }
    void *stage = bytes <= kPinnedBytes ? pinned_staging() : nullptr;
    if (stage) {
        FIGH_HIP(hipStreamSynchronize(g_stream));  // the staging buffer may still feed an earlier copy
        std::memcpy(stage, h_src, bytes);
        FIGH_HIP(hipMemcpyAsync(d_dst, stage, bytes, hipMemcpyHostToDevice, g_stream));
        return FIGH_OK;  // stream-ordered: later kernels on the library stream see the data
    }
    FIGH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g_stream));
    FIGH_HIP(hipStreamSynchronize(g_stream));
    return FIGH_OK;
}

// page-locked buffers handed out by figh_host_alloc: copies into them need no staging
static std::vector<std::pair<char *, size_t>> g_host_bufs;

int figh_host_alloc(void **h_ptr, size_t bytes) {
    FIGH_REQUIRE(h_ptr, "h_ptr is NULL");
    if (int rc = ensure_device()) return rc;
    if (hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc failed");
        return FIGH_ERR_ALLOC;
    }
    g_host_bufs.emplace_back(static_cast<char *>(*h_ptr), bytes ? bytes : 8);
    return FIGH_OK;
}

int figh_host_free(void *h_ptr) {
    if (!h_ptr) return FIGH_OK;
    for (size_t k = 0; k < g_host_bufs.size(); ++k)
        if (g_host_bufs[k].first == h_ptr) {
            g_host_bufs.erase(g_host_bufs.begin() + k);
            FIGH_HIP(hipStreamSynchronize(g_stream));
            FIGH_HIP(hipHostFree(h_ptr));
            return FIGH_OK;
        }
    set_error("figh_host_free: not a figh_host_alloc buffer");
    return FIGH_ERR_INVALID;
}

static bool is_pinned(const void *p, size_t bytes) {
    const char *c = static_cast<const char *>(p);
    for (const auto &b : g_host_bufs)
        if (c >= b.first && c + bytes <= b.first + b.second) return true;
    return false;
}

int figh_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes) {
    if (int rc = ensure_device()) return rc;
    void *stage = (bytes <= kPinnedBytes && !is_pinned(h_dst, bytes)) ? pinned_staging() : nullptr;
    if (stage) {
        FIGH_HIP(hipMemcpyAsync(stage, d_src, bytes, hipMemcpyDeviceToHost, g_stream));
        FIGH_HIP(hipStreamSynchronize(g_stream));
        std::memcpy(h_dst, stage, bytes);
        return FIGH_OK;
    }
    if (bytes >= kBulkMin && !is_pinned(h_dst, bytes) && bulk_ready()) {
        const hipError_t e = bulk_copy(static_cast<char *>(h_dst), const_cast<char *>(static_cast<const char *>(d_src)), bytes, true);
        if (e != hipErrorNotReady) {
            FIGH_HIP(e);
            return FIGH_OK;
        }
    }
    FIGH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g_stream));
    FIGH_HIP(hipStreamSynchronize(g_stream));
    return FIGH_OK;
}

int figh_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes) {
    if (int rc = ensure_device()) return rc;
    FIGH_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g_stream));
    return FIGH_OK;
}

int figh_memset(void *d_dst, int value, size_t bytes) {
    if (int rc = ensure_device()) return rc;
    FIGH_HIP(hipMemsetAsync(d_dst, value, bytes, g_stream));
    return FIGH_OK;
}

int figh_synchronize(void) {
    if (int rc = ensure_device()) return rc;
    FIGH_HIP(hipStreamSynchronize(g_stream));
    return FIGH_OK;
}

int figh_profile_enable(int level) {
    g_profile = level < 0 ? 0 : (level > 2 ? 2 : level);
    return FIGH_OK;
}

int figh_profile_reset(void) {
    if (g_ready) {
        (void)hipStreamSynchronize(g_stream);
        drain_profile();
    }
    g_prof.clear();
    return FIGH_OK;
}

int figh_profile_get(const char *name, int *launches, double *total_ms) {
    FIGH_REQUIRE(name, "name is NULL");
    if (g_ready) {
        FIGH_HIP(hipStreamSynchronize(g_stream));
        drain_profile();
    }
    auto it = g_prof.find(name);
    if (launches) *launches = it == g_prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == g_prof.end() ? 0.0 : it->second.ms;
    return FIGH_OK;
}

int figh_model_create(int njoints, const int32_t *parents, const int32_t *jtype, const double *axis,
                      const double *placement, const int32_t *idx_q, const int32_t *idx_v, const double *gravity,
                      const int32_t *body_mask, figh_model_t *out) {
    FIGH_REQUIRE(out, "out is NULL");
    FIGH_REQUIRE(njoints >= 2 && njoints <= kMaxJoints, "njoints must be in [2, 64]");
    FIGH_REQUIRE(parents && jtype && axis && placement && idx_q && idx_v && gravity && body_mask, "NULL array");
    if (int rc = ensure_device()) return rc;
    auto *m = new figh_model_s();
    DevModel &h = m->host;
    std::memset(&h, 0, sizeof(h));
    h.njoints = njoints;
    h.nlinks = njoints - 1;
    bool chain = true;
    int nq = 0, nv = 0;
    for (int i = 0; i < njoints; ++i) {
        h.parents[i] = parents[i];
        h.jtype[i] = jtype[i];
        h.idx_q[i] = idx_q[i];
        h.idx_v[i] = idx_v[i];
        h.body_mask[i] = body_mask[i] != 0;
        for (int k = 0; k < 3; ++k) h.axis[i][k] = axis[3 * i + k];
        for (int k = 0; k < 12; ++k) h.placement[i][k] = placement[12 * i + k];
        if (i == 0) continue;
        if (parents[i] < 0 || parents[i] >= i) {
            delete m;
            set_error("parents[i] must satisfy 0 <= parents[i] < i (depth-first numbering)");
            return FIGH_ERR_INVALID;
        }
        h.depth[i] = h.depth[parents[i]] + 1;
        if (h.depth[i] > m->max_depth) m->max_depth = h.depth[i];
        int jq, jv;
        switch (jtype[i]) {
            case FIGH_JT_REVOLUTE: jq = 1; jv = 1; break;
            case FIGH_JT_PRISMATIC: jq = 1; jv = 1; break;
            case FIGH_JT_CONTINUOUS: jq = 2; jv = 1; break;
            case FIGH_JT_FREEFLYER: jq = 7; jv = 6; break;
            default:
                delete m;
                set_error("unsupported joint type");
                return FIGH_ERR_INVALID;
        }
        if (idx_q[i] != nq || idx_v[i] != nv) {
            delete m;
            set_error("idx_q / idx_v must be cumulative in joint order");
            return FIGH_ERR_INVALID;
        }
        nq += jq;
        nv += jv;
        if (jtype[i] != FIGH_JT_REVOLUTE || parents[i] != i - 1) chain = false;
    }
    for (int k = 0; k < 3; ++k) h.gravity[k] = gravity[k];
    h.nq = nq;
    h.nv = nv;
    m->is_chain = chain && h.nlinks <= 8;
    if (hipMalloc(&m->dev, sizeof(DevModel)) != hipSuccess) {
        delete m;
        set_error("hipMalloc(model) failed");
        return FIGH_ERR_ALLOC;
    }
    FIGH_HIP(hipMemcpy(m->dev, &h, sizeof(DevModel), hipMemcpyHostToDevice));
    *out = m;
    return FIGH_OK;
}

int figh_model_destroy(figh_model_t model) {
    if (!model) return FIGH_OK;
    if (g_ready) (void)hipStreamSynchronize(g_stream);
    forget_tapes(model);
    if (model->dev) (void)hipFree(model->dev);
    delete model;
    return FIGH_OK;
}

int figh_model_set_active_rows(figh_model_t model, const int32_t *h_rows, int n) {
    FIGH_REQUIRE(model, "model is NULL");
    unsigned long long mask = ~0ull;
    if (h_rows && n > 0) {
        // (the store decision is one bit per row block: a model with more than 64 dofs cannot restrict them -- rows >= 64
        // would alias rows mod 64 in the tape builder)
        FIGH_REQUIRE(model->host.nv <= 64, "active row blocks: at most 64 dofs");
        mask = 0ull;
        for (int k = 0; k < n; ++k) {
            FIGH_REQUIRE(h_rows[k] >= 0 && h_rows[k] < model->host.nv && h_rows[k] < 64, "active row block out of range");
            mask |= 1ull << h_rows[k];
        }
    }
    if (mask != model->active_rows) {
        if (g_ready) (void)hipStreamSynchronize(g_stream);
        forget_tapes(model);  // (the tapes carry the per-row store decision)
        model->active_rows = mask;
    }
    return FIGH_OK;
}

int figh_regressor_shape(figh_model_t model, int mode, int flags, int *rows_per_sample, int *ncols) {
    FIGH_REQUIRE(model, "model is NULL");
    FIGH_REQUIRE(mode == FIGH_MODE_JOINT_TORQUE || mode == FIGH_MODE_EXT_WRENCH, "bad mode");
    const DevModel &h = model->host;
    if (mode == FIGH_MODE_JOINT_TORQUE) {
        // the reference's joint-torque branch sizes W as (N*nv, 14*nv) and copies a (nv x 10*(njoints-1))
        // block into its first 10*nv columns (regressor.py:46,52): only consistent when njoints-1 == nv
        FIGH_REQUIRE(h.nlinks == h.nv, "joint-torque mode needs njoints-1 == nv (one dof per joint)");
    }
    if (flags & FIGH_FLAG_TX40) {
        FIGH_REQUIRE(mode == FIGH_MODE_JOINT_TORQUE && h.nv == 6, "TX40 coupling needs a 6-dof joint-torque model");
    }
    if (rows_per_sample) *rows_per_sample = mode == FIGH_MODE_JOINT_TORQUE ? h.nv : 6;
    if (ncols) *ncols = 14 * h.nlinks + ((flags & FIGH_FLAG_TX40) ? 3 : 0);
    return FIGH_OK;
}

}  // extern "C"

#ifdef FIGH_ABLATION
// Ablation build only (libfigh_ab.so, tools/overlap_probe.py): switch the library stream between two streams, so that two
// entry points can be put in flight side by side, and wait for the whole device.
extern "C" int figh_ab_stream_select(int k) {
    static hipStream_t streams[2] = {nullptr, nullptr};
    if (int rc = figh::ensure_device()) return rc;
    if (!streams[0]) streams[0] = g_stream;
    if (k == 1 && !streams[1]) FIGH_HIP(hipStreamCreateWithFlags(&streams[1], hipStreamNonBlocking));
    g_stream = streams[k ? 1 : 0];
    return FIGH_OK;
}
extern "C" int figh_ab_device_sync(void) {
    FIGH_HIP(hipDeviceSynchronize());
    return FIGH_OK;
}
#endif
