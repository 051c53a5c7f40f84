// xr_engine.hip -- engine context, HBM block pool, error reporting, kernel timing, raw HBM helpers.
#include <cstdarg>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <cstring>
#include <thread>
#include <condition_variable>
#include <functional>

#include "xr_internal.h"

namespace xr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

std::recursive_mutex &engine_mutex() {
    static std::recursive_mutex m;
    return m;
}
std::shared_mutex &engine_rw() {
    static std::shared_mutex m;
    return m;
}

// ---- exclusive / shared use of the engine ------------------------------------------------------------------------
static thread_local int t_exclusive_depth = 0;
static thread_local Lane *t_lane = nullptr;
static constexpr int N_LANES = 4;
static Lane g_lanes[N_LANES];
static std::mutex g_pool_mutex; // the block pool and the kernel-timer tables are shared by all threads
static std::mutex g_main_turn;  // shared-scope calls that run on the main stream (no lane) take turns

Lane *current_lane() { return t_lane; }
bool exclusive_held() { return t_exclusive_depth > 0; }

static thread_local hipStream_t t_stream_override = nullptr;
hipStream_t stream_override() { return t_stream_override; }
StreamOverride::StreamOverride(hipStream_t s) : prev(t_stream_override) { t_stream_override = s; }
StreamOverride::~StreamOverride() { t_stream_override = prev; }

// ---- run-time options (xr_internal.h: enum Option) --------------------------------------------------------------------------
namespace {
struct OptionDef {
    const char *name;
    int64_t def;
};
const OptionDef k_option_defs[OPT_COUNT] = {
    {"overlap_fused", 1}, {"queue_margin", 0}, {"clip_quad", 1}, {"dust", 1}, {"no_side", 0}, {"side_fork", 1}, {"debug", 0},
    {"host_stamps", 0}, {"apply_plan", 1}, {"apply_contract", 0}, {"plan_merge", -1}, {"plan_dbg", 0}, {"apply_chunk_bytes", 0},
    {"outer_apply", 0}, {"edge_big", 0}, {"edge_stage", 0}, {"edge_queue", 0}, {"edge_sort", 1}, {"mail_poll", 1},
    {"points_defer", 1}, {"ingest_device", 0}, {"stats_sample", 1}, {"force_query_sort", 0}, {"early_apply", 1}, {"star_flag", 1},
};
std::atomic<int64_t> g_options[OPT_COUNT];
std::once_flag g_options_once;
// words some options accepted as environment values before they were numbers
int64_t option_word(const char *v) {
    static const struct { const char *word; int64_t value; } words[] = {{"free", 1}, {"csr", 2}, {"device", 1}};
    for (const auto &w : words)
        if (!strcmp(v, w.word)) return w.value;
    char *end = nullptr;
    const long long n = strtoll(v, &end, 10);
    return end && end != v ? (int64_t)n : 0;
}
void options_init() {
    std::call_once(g_options_once, [] {
        for (int i = 0; i < OPT_COUNT; i++) {
            char env[64] = "XR_";
            size_t k = 3;
            for (const char *c = k_option_defs[i].name; *c && k + 1 < sizeof(env); c++) env[k++] = (char)toupper((unsigned char)*c);
            env[k] = 0;
            const char *v = getenv(env); // (the library's only look at the environment besides XUGRID_AMD_LIB on the Python side)
            g_options[i].store(v ? option_word(v) : k_option_defs[i].def, std::memory_order_relaxed);
        }
    });
}
int option_index(const char *name) {
    for (int i = 0; i < OPT_COUNT; i++)
        if (!strcmp(name, k_option_defs[i].name)) return i;
    return -1;
}
} // namespace
int64_t option(Option o) {
    options_init();
    return g_options[o].load(std::memory_order_relaxed);
}

static thread_local bool t_mark_open = false; // (side_mark() .. SideScope(at_mark): see side_mark)
ExclusiveScope::ExclusiveScope() {
    if (t_exclusive_depth++ == 0) {
        t_mark_open = false; // (a call that failed between a mark and its scope leaves nothing behind)
        engine_mutex().lock();
        engine_rw().lock();
    }
}
ExclusiveScope::~ExclusiveScope() {
    if (--t_exclusive_depth == 0) {
        engine_rw().unlock();
        engine_mutex().unlock();
    }
}

static void lane_release_blocks(Lane *lane);

SharedScope::SharedScope() {
    if (t_exclusive_depth > 0 || t_lane) return; // nested inside another entry point: that one's context is used
    t_mark_open = false;
    engine_rw().lock_shared();
    leased = true;
    Engine &e = engine();
    // Bound to the caller's stream (xr_set_stream) or asynchronous mode (xr_set_async): everything stays on the ONE main
    // stream (stream-ordered pool) and there are no lanes.  The main stream's state -- on_side, the fork / join / aux events,
    // the deferred pool list -- is not atomic, so calls from several host threads take turns here instead of running
    // concurrently: a second thread waits until the first one's call has been ENQUEUED (not finished: the stream orders them).
    if (e.stream != e.own_stream || e.own_async) {
        g_main_turn.lock();
        serial = true;
        return;
    }
    // a free lane, else wait for "this thread's" one
    const size_t h = std::hash<std::thread::id>()(std::this_thread::get_id());
    Lane *lane = nullptr;
    for (int i = 0; i < N_LANES && !lane; i++) {
        Lane &c = g_lanes[(h + i) % N_LANES];
        if (c.busy.try_lock()) lane = &c;
    }
    if (!lane) {
        lane = &g_lanes[h % N_LANES];
        lane->busy.lock();
    }
    if (!lane->stream) {
        if (hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) {
            // no lane, no concurrency: running on the engine's own stream under the SHARED lock would race with the
            // other lanes' pool blocks -- fail the call instead
            (void)hipGetLastError();
            lane->stream = nullptr;
            lane->busy.unlock();
            engine_rw().unlock_shared();
            leased = false;
            set_error("could not create a HIP stream for a concurrent apply");
            throw Failure{XR_ERR_HIP};
        }
    }
    t_lane = lane;
}
SharedScope::~SharedScope() {
    if (!leased) return;
    if (serial) g_main_turn.unlock();
    if (t_lane) {
        (void)hipStreamSynchronize(t_lane->stream); // the call is over: nothing of it is in flight
        if (t_lane->side_active) { // (a call that failed between fork and join)
            (void)hipStreamSynchronize(t_lane->side);
            t_lane->side_active = false;
            t_lane->on_side = false;
        }
        lane_release_blocks(t_lane);
        Lane *lane = t_lane;
        t_lane = nullptr;
        lane->busy.unlock();
    }
    engine_rw().unlock_shared();
}

static Engine g_engine;

void engine_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); xugrid_amd has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        throw Failure{XR_ERR_NO_DEVICE};
    }
    XR_REQUIRE(device >= 0 && device < count, XR_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    if (g_engine.device == device && g_engine.own_stream) return;
    XR_REQUIRE(g_engine.device < 0, XR_ERR_INVALID,
               "engine already bound to device %d; one process per GPU", g_engine.device);
    XR_HIP(hipSetDevice(device));
    XR_HIP(hipStreamCreateWithFlags(&g_engine.stream, hipStreamNonBlocking));
    g_engine.own_stream = g_engine.stream;
    hipDeviceProp_t prop;
    XR_HIP(hipGetDeviceProperties(&prop, device));
    g_engine.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    XR_HIP(hipHostMalloc(&g_engine.pinned, 4096, hipHostMallocDefault));
    void *mail = nullptr;
    XR_HIP(hipHostMalloc(&mail, 4096, hipHostMallocCoherent));
    g_engine.mailbox = static_cast<volatile int32_t *>(mail);
    XR_HIP(hipEventCreateWithFlags(&g_engine.mail_event, hipEventDisableTiming | hipEventReleaseToSystem));
    {
        // the side stream carries short latency-bound kernels next to a long one on the main stream: highest priority
        int lo = 0, hi = 0;
        XR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // (lowest / normal / highest measured in round 5 on config 3: 1.81-1.86 / 1.83-1.88 / 1.89-1.91 ms under one script's clock, but
        // the big faces' chain of xr_overlap needs the highest)
        XR_HIP(hipStreamCreateWithPriority(&g_engine.side, hipStreamNonBlocking, hi));
        XR_HIP(hipStreamCreateWithPriority(&g_engine.side2, hipStreamNonBlocking, hi));
    }
    // fork / join between two streams of the one device: a device-scope release is all the waiting stream needs (the default,
    // a system-scope release, writes the caches back to host visibility at every record: ~7 us between two dependent kernels
    // on the main stream, rocprofv3 timeline of round 4).
    const unsigned ev_flags = hipEventDisableTiming | hipEventReleaseToDevice;
    XR_HIP(hipEventCreateWithFlags(&g_engine.fork_event, ev_flags));
    XR_HIP(hipEventCreateWithFlags(&g_engine.join_event, ev_flags));
    XR_HIP(hipEventCreateWithFlags(&g_engine.aux_event, ev_flags));
    XR_HIP(hipEventCreateWithFlags(&g_engine.fork2_event, ev_flags));
    XR_HIP(hipEventCreateWithFlags(&g_engine.join2_event, ev_flags));
    g_engine.device = device;
}

Engine &engine() {
    if (g_engine.device < 0) engine_init(0);
    return g_engine;
}

// ---------------------------------------------------------------------------------------------
// block pool: power-of-two-ish size classes (1/4-octave), free lists per class
// ---------------------------------------------------------------------------------------------
static std::map<void *, size_t> g_live;                  // ptr -> class size
static std::multimap<size_t, void *> g_free;             // class size -> ptr
// blocks freed by a thread on a lane: reused only by that lane (stream order) until the lane is drained
static std::map<Lane *, std::multimap<size_t, void *>> g_lane_free;

static size_t size_class(size_t bytes) {
    size_t b = bytes < 4096 ? 4096 : bytes;
    size_t p = 4096;
    while (p < b) p <<= 1;
    // quarter-octave steps to limit internal waste to 25 %
    size_t q = p >> 1;
    for (int i = 1; i <= 3; i++) {
        size_t c = q + (q >> 2) * i;
        if (c >= b) return c;
    }
    return p;
}

// Blocks are handed out again in stream order, which is only safe with ONE stream.  From the moment a side stream is
// forked until the host has seen both streams drained, freed blocks are parked instead of being reused: a block freed
// by code on one stream could otherwise be given to the other while a kernel of the first still touches it.
static bool g_side_active = false;
static std::vector<std::pair<size_t, void *>> g_deferred;
static std::map<Lane *, std::vector<std::pair<size_t, void *>>> g_lane_deferred; // the same per lane, until its side_join()

void *pool_alloc(size_t bytes) {
    engine();
    size_t c = size_class(bytes);
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    void *p = nullptr;
    if (t_lane) {
        auto &mine = g_lane_free[t_lane];
        auto lt = mine.find(c);
        if (lt != mine.end()) {
            p = lt->second;
            mine.erase(lt);
            g_live[p] = c;
            return p;
        }
    }
    // (the common list only holds blocks nothing in flight touches: exclusive calls end synchronised, lanes hand their
    // blocks over when drained)
    auto it = g_free.find(c);
    if (it != g_free.end()) {
        p = it->second;
        g_free.erase(it);
    } else {
        hipError_t e = hipMalloc(&p, c);
        if (e != hipSuccess) {
            // give cached blocks back and retry once.  The lanes' own lists and the parked (deferred) blocks may still
            // be touched by kernels in flight: wait for the device first, then every cached block is really free
            // (lists are only changed under g_pool_mutex, which is held here).
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            for (auto &kv : g_free) (void)hipFree(kv.second);
            g_free.clear();
            for (auto &lane : g_lane_free) {
                for (auto &kv : lane.second) (void)hipFree(kv.second);
                lane.second.clear();
            }
            for (auto &kv : g_deferred) (void)hipFree(kv.second);
            g_deferred.clear();
            for (auto &lane : g_lane_deferred) {
                for (auto &kv : lane.second) (void)hipFree(kv.second);
                lane.second.clear();
            }
            XR_HIP(hipMalloc(&p, c));
        }
    }
    g_live[p] = c;
    return p;
}


void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    if (t_lane && t_lane->side_active) g_lane_deferred[t_lane].emplace_back(it->second, p);
    else if (t_lane) g_lane_free[t_lane].emplace(it->second, p);
    else if (g_side_active) g_deferred.emplace_back(it->second, p);
    else g_free.emplace(it->second, p);
    g_live.erase(it);
}

static void lane_release_blocks(Lane *lane) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto &mine = g_lane_free[lane];
    for (auto &kv : g_lane_deferred[lane]) mine.emplace(kv.first, kv.second); // (the lane, side stream included, is drained)
    g_lane_deferred[lane].clear();
    for (auto &kv : mine) g_free.emplace(kv.first, kv.second);
    mine.clear();
}

// called after the main stream has been synchronised with the host
static void pool_release_deferred(bool may_block = true) {
    if (!g_side_active || t_lane) return;
    if (may_block) {
        (void)hipStreamSynchronize(g_engine.side);
        (void)hipStreamSynchronize(g_engine.side2);
    } else if (hipStreamQuery(g_engine.side) != hipSuccess || hipStreamQuery(g_engine.side2) != hipSuccess) { // (still busy: the blocks stay parked until a later host sync)
        (void)hipGetLastError();
        return;
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &kv : g_deferred) g_free.emplace(kv.first, kv.second);
    g_deferred.clear();
    g_side_active = false;
}

// ---------------------------------------------------------------------------------------------
// "zero at rest" scratch words (see xr_internal.h)
// ---------------------------------------------------------------------------------------------
static int32_t *g_zero_buf[ZERO_SLOTS] = {nullptr, nullptr};
static size_t g_zero_cap[ZERO_SLOTS] = {0, 0};
static bool g_zero_dirty[ZERO_SLOTS] = {false, false};

int32_t *zero_scratch(int slot, size_t n_words) {
    if (!exclusive_held() || stream_override() || engine().on_side || n_words > ZERO_SCRATCH_MAX_WORDS) return nullptr;
    if (n_words > g_zero_cap[slot]) {
        if (g_zero_buf[slot]) pool_free(g_zero_buf[slot]);
        g_zero_buf[slot] = nullptr;
        g_zero_cap[slot] = 0;
        const size_t cap = n_words < 1024 ? 1024 : n_words + n_words / 4; // (meshes of about one size keep their buffer)
        g_zero_buf[slot] = static_cast<int32_t *>(pool_alloc(cap * sizeof(int32_t)));
        g_zero_cap[slot] = cap;
        g_zero_dirty[slot] = true;
    }
    if (g_zero_dirty[slot]) XR_HIP(hipMemsetAsync(g_zero_buf[slot], 0, g_zero_cap[slot] * sizeof(int32_t), engine().stream));
    g_zero_dirty[slot] = true; // until the caller has enqueued the kernel that restores the zeros
    return g_zero_buf[slot];
}

void zero_scratch_done(int slot) { g_zero_dirty[slot] = false; }

static void zero_scratch_drop() {
    for (int s = 0; s < ZERO_SLOTS; s++) {
        if (g_zero_buf[s]) pool_free(g_zero_buf[s]);
        g_zero_buf[s] = nullptr;
        g_zero_cap[s] = 0;
        g_zero_dirty[s] = false;
    }
}

void pool_trim() {
    if (g_engine.stream) (void)hipStreamSynchronize(g_engine.stream);
    pool_release_deferred();
    zero_scratch_drop();
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &kv : g_free) (void)hipFree(kv.second);
    g_free.clear();
}

static constexpr size_t STAGE_BYTES = (size_t)64 << 20;
static constexpr size_t BIG_COPY_BYTES = (size_t)1 << 20; // copies of this size and more go through the pinned staging buffers
static constexpr size_t SMALL_COPY_BYTES = BIG_COPY_BYTES;

static bool mail_poll_enabled();
static inline void cpu_relax();

// ---- small copies without a stream synchronisation (round 5) -----------------------------------------------------------
// A read-back of a few bytes to a few hundred KB used to be hipMemcpyAsync + hipStreamSynchronize: 35-55 us of host
// latency each, with the device idle behind it (the Voronoi pre-step of the barycentric path makes eight of them, a small
// regridder construction is little else).  On the engine's own main stream they now go the mailbox's way: a kernel copies
// the bytes into a coherent pinned page and the LAST of its blocks stores a sequence number behind them, which the host
// polls.  Uploads copy the host bytes into a pinned ring slot and enqueue the DMA without waiting for it (the slot is
// reused after its event).  Anything else -- lanes, a caller's stream, the side stream, XR_MAIL_POLL=0 -- keeps the
// synchronous path.
static constexpr size_t FAST_D2H_MAX = (size_t)1 << 20;  // bytes of the pinned read-back page
static constexpr size_t FAST_H2D_SLOT = (size_t)1 << 18, FAST_H2D_SLOTS = 16; // (256 KB: the boundary lists of a Voronoi pre-step -- a pageable copy goes through a blit KERNEL, which waits for wave slots beside a kernel that fills the device)
struct FastCopy {
    char *page = nullptr;          // coherent pinned: [0, FAST_D2H_MAX) data, then one sequence word
    int32_t *done = nullptr;       // device word: blocks of the copy kernel that have finished
    int32_t seq = 0;
    char *ring = nullptr;          // pinned upload slots
    hipEvent_t ring_ev[FAST_H2D_SLOTS] = {};
    size_t ring_next = 0;
};
static FastCopy g_fast;

__global__ void __launch_bounds__(256)
k_copy_to_host(const uint32_t *__restrict__ src, uint32_t *dst, size_t n_words, const uint8_t *__restrict__ tail_src, uint8_t *tail_dst,
               int n_tail, int32_t *done, volatile int32_t *seq_word, int32_t seq) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(done, 1) == (int)gridDim.x - 1) { // the last block: every other block's bytes are out
            *done = 0;
            __threadfence_system();
            *seq_word = seq;
        }
    }
}

static bool fast_copy_allowed() {
    Engine &e = engine();
    return mail_poll_enabled() && !t_lane && !t_stream_override && !e.on_side && e.stream == e.own_stream;
}
static void fast_copy_init() {
    if (g_fast.page) return;
    void *p = nullptr;
    XR_HIP(hipHostMalloc(&p, FAST_D2H_MAX + 64, hipHostMallocCoherent));
    memset(p, 0, FAST_D2H_MAX + 64);
    g_fast.page = static_cast<char *>(p);
    XR_HIP(hipMalloc(reinterpret_cast<void **>(&g_fast.done), 64));
    XR_HIP(hipMemset(g_fast.done, 0, 64));
    XR_HIP(hipHostMalloc(&p, FAST_H2D_SLOT * FAST_H2D_SLOTS, hipHostMallocDefault));
    g_fast.ring = static_cast<char *>(p);
}

void assert_no_open_mark(const char *what);
void h2d(void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    assert_no_open_mark("h2d");
    if (bytes >= BIG_COPY_BYTES) {
        h2d_big(dst, src, bytes);
        return;
    }
    if (bytes <= FAST_H2D_SLOT && fast_copy_allowed()) {
        fast_copy_init();
        const size_t slot = g_fast.ring_next++ % FAST_H2D_SLOTS;
        if (g_fast.ring_ev[slot]) XR_HIP(hipEventSynchronize(g_fast.ring_ev[slot])); // (sixteen uploads ago: long done)
        else XR_HIP(hipEventCreateWithFlags(&g_fast.ring_ev[slot], hipEventDisableTiming));
        char *stage = g_fast.ring + slot * FAST_H2D_SLOT;
        memcpy(stage, src, bytes);
        XR_HIP(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, engine().stream));
        XR_HIP(hipEventRecord(g_fast.ring_ev[slot], engine().stream));
        engine().main_busy = true;
        return; // (the caller's buffer is free again; the copy is ordered in front of everything enqueued after it)
    }
    XR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, launch_stream()));
    XR_HIP(hipStreamSynchronize(launch_stream()));
}

static void d2h_impl(void *dst, const void *src, size_t bytes, const std::function<void()> *behind);
void d2h(void *dst, const void *src, size_t bytes) { d2h_impl(dst, src, bytes, nullptr); }
void d2h(void *dst, const void *src, size_t bytes, const std::function<void()> &behind) { d2h_impl(dst, src, bytes, &behind); }
static void d2h_impl(void *dst, const void *src, size_t bytes, const std::function<void()> *behind) {
    if (!bytes) {
        if (behind) (*behind)();
        return;
    }
    if (bytes <= FAST_D2H_MAX && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && fast_copy_allowed()) {
        fast_copy_init();
        Engine &e = engine();
        const size_t n_words = bytes / 4;
        const int n_tail = (int)(bytes & 3);
        g_fast.seq = g_fast.seq >= (1 << 30) ? 1 : g_fast.seq + 1;
        volatile int32_t *seq_word = reinterpret_cast<volatile int32_t *>(g_fast.page + FAST_D2H_MAX);
        const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n_words + 255) / 256, 1), 256);
        hipLaunchKernelGGL(k_copy_to_host, dim3(grid), dim3(256), 0, e.stream, static_cast<const uint32_t *>(src),
                           reinterpret_cast<uint32_t *>(g_fast.page), n_words, static_cast<const uint8_t *>(src) + 4 * n_words,
                           reinterpret_cast<uint8_t *>(g_fast.page) + 4 * n_words, n_tail, g_fast.done, seq_word, g_fast.seq);
        XR_HIP(hipGetLastError());
        if (behind) (*behind)(); // (enqueued behind the copy: it runs while the host waits for the words and works on them)
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = true;
        for (uint64_t spins = 0; *seq_word != g_fast.seq; spins++) {
            cpu_relax();
            if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
                XR_HIP(hipStreamSynchronize(e.stream)); // (slow or failed kernel: the error, if any, surfaces here)
                seen = *seq_word == g_fast.seq;
                break;
            }
        }
        XR_REQUIRE(seen, XR_ERR_HIP, "read-back sequence word missing after synchronisation");
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(dst, g_fast.page, bytes);
        if (behind) return; // (work was enqueued behind the copy: the stream is not drained)
        // (everything in front of the copy kernel has executed: the stream is drained as far as pool blocks are concerned)
        e.main_busy = false;
        pool_release_deferred(/*may_block=*/false);
        return;
    }
    if (behind) { // (no mailbox here -- a caller's stream, a lane --: the plain order, the work first)
        (*behind)();
        behind = nullptr;
    }
    if (bytes <= 4096) {
        // scalar read-backs go through the pinned page: no pageable staging, one sync
        XR_HIP(hipMemcpyAsync(engine().pinned, src, bytes, hipMemcpyDeviceToHost, engine().stream));
        XR_HIP(hipStreamSynchronize(engine().stream));
        memcpy(dst, engine().pinned, bytes);
        return;
    }
    if (bytes >= BIG_COPY_BYTES) {
        d2h_big(dst, src, bytes);
        return;
    }
    XR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, launch_stream()));
    XR_HIP(hipStreamSynchronize(launch_stream()));
}

void stream_sync() {
    XR_HIP(hipStreamSynchronize(launch_stream()));
    if (!t_lane && !t_stream_override && !g_engine.on_side) g_engine.main_busy = false;
    pool_release_deferred();
}
void release_point() {
    Engine &e = engine();
    if (e.own_async && e.stream == e.own_stream && !t_lane && !t_stream_override) return;
    stream_sync();
}
void dev_call_done() {
    Engine &e = engine();
    if (e.async_dev) return;
    // xr_set_async(1): calls on the engine's own main stream return with their kernels in flight (blocks they freed are
    // reused in stream order on that one stream; a thread that takes a lane waits for the main stream first)
    if (e.own_async && !t_lane && !t_stream_override && e.stream == e.own_stream) {
        e.main_busy = true;
        return;
    }
    stream_sync();
}

// ---------------------------------------------------------------------------------------------
// staged copies of large pageable host arrays
// ---------------------------------------------------------------------------------------------
static constexpr int STAGE_THREADS = 8;
static Lane g_main_staging; // staging buffers of the engine's own path

// -> the staging buffers of the calling thread's lane (allocated on first use: 2 x 64 MiB of pinned host memory)
static Lane &stage_init() {
    Lane &l = t_lane ? *t_lane : g_main_staging;
    if (l.stage[0]) return l;
    for (int i = 0; i < 2; i++) {
        void *p = nullptr;
        XR_HIP(hipHostMalloc(&p, STAGE_BYTES, hipHostMallocDefault));
        l.stage[i] = static_cast<char *>(p);
        XR_HIP(hipEventCreateWithFlags(&l.stage_ev[i], hipEventDisableTiming));
    }
    return l;
}

// A small persistent pool of host threads for the staging copies (spawning 7 threads per copy costs ~0.1 ms -- as much as
// copying 8 MB).  parallel_ranges(n, fn) runs fn(begin, end) on STAGE_THREADS disjoint ranges of [0, n); the caller takes
// one range itself.  One job at a time (callers hold the engine lock or a lane; a mutex serialises the rest).
namespace {
struct HostPool {
    std::mutex job_mutex;                // one job at a time
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::function<void(size_t, size_t)> fn;
    size_t n = 0, per = 0;
    uint64_t generation = 0;
    int pending = 0;
    bool started = false;
    std::thread workers[STAGE_THREADS - 1];

    void start() {
        if (started) return;
        started = true;
        for (int t = 0; t < STAGE_THREADS - 1; t++) {
            workers[t] = std::thread([this, t] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lock(m);
                    cv_work.wait(lock, [&] { return generation != seen; });
                    seen = generation;
                    const size_t b = (size_t)(t + 1) * per, e = std::min(n, b + per);
                    auto f = fn;
                    lock.unlock();
                    if (b < e) {
                        JobFlag flag;
                        f(b, e);
                    }
                    lock.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
            workers[t].detach(); // (the pool lives as long as the process)
        }
    }
    // set on EVERY thread while it runs a slice of a job (the caller's own slice and the workers' slices alike)
    static bool &in_job() {
        static thread_local bool inside = false;
        return inside;
    }
    struct JobFlag {
        bool &f;
        JobFlag() : f(in_job()) { f = true; }
        ~JobFlag() { f = false; }
    };
    void run(size_t total, size_t align, const std::function<void(size_t, size_t)> &f) {
        if (total == 0) return;
        // not re-entrant: a slice that called back into the pool would wait for the job mutex its own job holds -- from the
        // caller's thread as well as from a worker's.  A nested call runs inline on the thread it came from.
        if (in_job()) {
            f(0, total);
            return;
        }
        std::lock_guard<std::mutex> job(job_mutex);
        JobFlag flag;
        size_t p = (total + STAGE_THREADS - 1) / STAGE_THREADS;
        p = (p + align - 1) / align * align;
        if (total < (size_t)1 << 16) { // small: not worth a wake-up
            f(0, total);
            return;
        }
        start();
        {
            std::lock_guard<std::mutex> lock(m);
            fn = f;
            n = total;
            per = p;
            pending = STAGE_THREADS - 1;
            generation++;
        }
        cv_work.notify_all();
        f(0, std::min(p, total));
        std::unique_lock<std::mutex> lock(m);
        cv_done.wait(lock, [&] { return pending == 0; });
    }
};
HostPool &host_pool() {
    static HostPool *pool = new HostPool(); // (never destroyed: its threads may outlive static destruction order)
    return *pool;
}
} // namespace

void parallel_ranges(size_t n, size_t align, const std::function<void(size_t, size_t)> &fn) { host_pool().run(n, align, fn); }

static void parallel_memcpy(char *dst, const char *src, size_t n) {
    parallel_ranges(n, 64, [=](size_t b, size_t e) { memcpy(dst + b, src + b, e - b); });
}

// Host array -> device through the pinned staging buffers in pieces (4 or 16 MiB OF DEVICE DATA), double-buffered: the
// pool threads fill piece i + 1 (fill(dst_pinned, first_byte, n_bytes): a copy, or a narrowing conversion of the caller's
// array) while the DMA engine moves piece i.  Returns without waiting for the last DMA: the caller's array has been
// consumed, everything later on the stream is ordered behind the copies.
static inline size_t up_piece(size_t bytes) { // small pieces start the pipeline early; big copies amortise the per-piece cost
    return bytes >= ((size_t)128 << 20) ? ((size_t)16 << 20) : ((size_t)4 << 20);
}
void h2d_staged(void *dst, size_t bytes, const std::function<void(char *, size_t, size_t)> &fill) {
    if (!bytes) return;
    assert_no_open_mark("h2d_staged");
    Lane &sl = stage_init();
    hipStream_t st = launch_stream();
    // the two 64 MiB staging buffers as a ring of pieces
    const size_t UP_PIECE = up_piece(bytes);
    const size_t per_buf = STAGE_BYTES / UP_PIECE;
    size_t off = 0;
    for (size_t k = 0; off < bytes; k++) {
        const int buf = (int)((k / per_buf) & 1);
        const size_t slot = k % per_buf;
        if (slot == 0) XR_HIP(hipEventSynchronize(sl.stage_ev[buf])); // the DMAs that last read this buffer are done
        const size_t c = std::min(UP_PIECE, bytes - off);
        char *pinned = sl.stage[buf] + slot * UP_PIECE;
        fill(pinned, off, c);
        XR_HIP(hipMemcpyAsync(static_cast<char *>(dst) + off, pinned, c, hipMemcpyHostToDevice, st));
        off += c;
        if (slot == per_buf - 1 || off >= bytes) XR_HIP(hipEventRecord(sl.stage_ev[buf], st));
    }
}

void h2d_big(void *dst, const void *src, size_t bytes) {
    if (bytes < SMALL_COPY_BYTES) {
        XR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, launch_stream()));
        XR_HIP(hipStreamSynchronize(launch_stream()));
        return;
    }
    // pieces through the pinned staging buffers, filled by the host thread pool while the previous piece is in flight
    const char *bytes_src = static_cast<const char *>(src);
    h2d_staged(dst, bytes, [=](char *pinned, size_t off, size_t n) { parallel_memcpy(pinned, bytes_src + off, n); });
    XR_HIP(hipStreamSynchronize(launch_stream()));
}

// Device -> pageable host array: DMA into the pinned staging buffers in pieces, each copied out by the
// host thread pool while the next pieces are on their way (D2H_DEPTH pieces in flight, one event per piece).
static constexpr int D2H_DEPTH = 8;
void d2h_big(void *dst, const void *src, size_t bytes) {
    if (bytes < SMALL_COPY_BYTES) {
        XR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, launch_stream()));
        XR_HIP(hipStreamSynchronize(launch_stream()));
        return;
    }
    Lane &sl = stage_init();
    static thread_local hipEvent_t piece_ev[D2H_DEPTH] = {};
    if (!piece_ev[0])
        for (int i = 0; i < D2H_DEPTH; i++) XR_HIP(hipEventCreateWithFlags(&piece_ev[i], hipEventDisableTiming));
    hipStream_t st = launch_stream();
    // The staging buffers may still be read by un-awaited uploads of h2d_staged.  On the same stream the DMAs below are
    // ordered behind them; should the two calls ever run on different streams (a side scope or a stream override around
    // one of them) only the events say so: the stream waits for both before the first piece is written.
    for (int i = 0; i < 2; i++)
        if (sl.stage_ev[i]) XR_HIP(hipStreamWaitEvent(st, sl.stage_ev[i], 0));
    const size_t UP_PIECE = up_piece(bytes), per_buf = STAGE_BYTES / UP_PIECE; // (16 or 4 pieces per buffer: 8 slots fit)
    const size_t n_piece = (bytes + UP_PIECE - 1) / UP_PIECE;
    auto piece_bytes = [&](size_t k) { return std::min(UP_PIECE, bytes - k * UP_PIECE); };
    auto pinned_of = [&](size_t k) {
        const size_t slot = k % D2H_DEPTH;
        return sl.stage[slot / per_buf] + (slot % per_buf) * UP_PIECE;
    };
    auto issue = [&](size_t k) {
        XR_HIP(hipMemcpyAsync(pinned_of(k), static_cast<const char *>(src) + k * UP_PIECE, piece_bytes(k), hipMemcpyDeviceToHost, st));
        XR_HIP(hipEventRecord(piece_ev[k % D2H_DEPTH], st));
    };
    for (size_t k = 0; k < n_piece && k < (size_t)D2H_DEPTH; k++) issue(k);
    for (size_t k = 0; k < n_piece; k++) {
        XR_HIP(hipEventSynchronize(piece_ev[k % D2H_DEPTH]));
        parallel_memcpy(static_cast<char *>(dst) + k * UP_PIECE, pinned_of(k), piece_bytes(k));
        if (k + D2H_DEPTH < n_piece) issue(k + D2H_DEPTH); // (its slot has just been copied out)
    }
    XR_HIP(hipStreamSynchronize(st));
}

void mailbox_wait() {
    XR_HIP(hipEventRecord(engine().mail_event, engine().stream));
    XR_HIP(hipEventSynchronize(engine().mail_event));
    pool_release_deferred();
}

static bool mail_poll_enabled() { return option(OPT_MAIL_POLL) != 0; }

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

int32_t mailbox_next_seq() {
    Engine &e = engine();
    e.mail_seq = e.mail_seq >= (1 << 30) ? 1 : e.mail_seq + 1;
    return e.mail_seq;
}

void mailbox_wait_seq(int32_t seq) {
    Engine &e = engine();
    if (!mail_poll_enabled() || e.stream != e.own_stream) { // (a caller's stream: its events order things, keep the event path)
        mailbox_wait();
        XR_REQUIRE(e.mailbox[MAIL_SEQ_SLOT] == seq, XR_ERR_HIP, "mailbox sequence word missing after synchronisation");
        return;
    }
    const volatile int32_t *word = e.mailbox + MAIL_SEQ_SLOT;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *word != seq; spins++) {
        cpu_relax();
        if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            XR_HIP(hipStreamSynchronize(e.stream)); // (slow or failed kernel: the error, if any, surfaces here)
            XR_REQUIRE(*word == seq, XR_ERR_HIP, "mailbox sequence word missing after synchronisation");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    pool_release_deferred(/*may_block=*/false); // (the publisher ran behind the join: the side stream is normally drained)
}

bool poll_pinned_f64(const volatile double *word, double expected) {
    if (!mail_poll_enabled()) return false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *word != expected; spins++) {
        cpu_relax();
        if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return false;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

// ---------------------------------------------------------------------------------------------
// host stamps (XR_HOST_STAMPS=1)
// ---------------------------------------------------------------------------------------------
namespace {
struct HostStamps {
    bool on = option(OPT_HOST_STAMPS) != 0;
    std::chrono::steady_clock::time_point last[16];
    bool seen[16] = {};
    double sum_us[16][16] = {}; // [from][to] accumulated interval from the latest stamp of `from` to a stamp of `to`
    long n[16][16] = {};
    int prev = -1;
    ~HostStamps() {
        if (!on) return;
        for (int a = 0; a < 16; a++)
            for (int b = 0; b < 16; b++)
                if (n[a][b] > 20) fprintf(stderr, "[host stamps] %d -> %d: %.2f us (n = %ld)\n", a, b, sum_us[a][b] / n[a][b], n[a][b]);
    }
};
HostStamps g_stamps;
} // namespace
void host_stamp(int point) {
    if (!g_stamps.on) return;
    const auto now = std::chrono::steady_clock::now();
    if (g_stamps.prev >= 0) {
        const int a = g_stamps.prev;
        g_stamps.sum_us[a][point] += std::chrono::duration<double, std::micro>(now - g_stamps.last[a]).count();
        g_stamps.n[a][point]++;
    }
    g_stamps.last[point] = now;
    g_stamps.prev = point;
}

// ---------------------------------------------------------------------------------------------
// kernel timing
// ---------------------------------------------------------------------------------------------
struct ProfRec {
    int64_t launches = 0;
    double ms = 0.0;
};
struct Pending {
    const char *name;
    hipEvent_t e0, e1;
};
static std::map<std::string, ProfRec> g_prof;
static std::vector<std::string> g_prof_order;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t get_event() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    XR_HIP(hipEventCreate(&e));
    return e;
}

ProfScope::ProfScope(const char *n) : name(n) {
    if (!g_engine.prof) return;
    e0 = get_event();
    e1 = get_event();
    (void)hipEventRecord(e0, launch_stream());
    on_side = g_engine.on_side;
    lane_stream = t_stream_override ? t_stream_override : (t_lane ? t_lane->stream : nullptr);
}

ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, lane_stream ? lane_stream : (on_side ? g_engine.side : g_engine.stream));
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pending.push_back({name, e0, e1});
}

// XR_NO_SIDE=1 (measurement hook): the "side" work runs in line on the main stream -- every kernel alone on the device
static bool side_disabled() { return option(OPT_NO_SIDE) != 0; }

static bool lane_side_ready(Lane *l) {
    if (l->side) return true;
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return false; // (no side stream: in line)
    }
    l->side = st;
    l->fork_event = a;
    l->join_event = b;
    return true;
}
// Invariant of the host -> device copies (h2d, h2d_staged): they are ordered on the MAIN stream only.  Work started with
// SideScope(at_mark = true) depends on what the main stream held at the last side_mark(), so an upload issued between the two
// would not be visible to it: t_mark_open is set by side_mark(), cleared by the SideScope that consumes the mark, and every
// upload checks it.  (The one at_mark user is the early apply of xr_overlap_apply_dev, which uploads nothing.)
void assert_no_open_mark(const char *what) {
    XR_REQUIRE(!t_mark_open, XR_ERR_INVALID, "internal: %s between side_mark() and SideScope(at_mark): the side work would not see it", what);
}
bool side_mark() {
    if (side_disabled()) return false;
    if (Lane *l = current_lane()) {
        if (!lane_side_ready(l)) return false;
        XR_HIP(hipEventRecord(l->fork_event, l->stream));
        t_mark_open = true;
        return true;
    }
    if (t_exclusive_depth == 0) return false;
    XR_HIP(hipEventRecord(engine().fork_event, engine().stream));
    t_mark_open = true;
    return true;
}
SideScope::SideScope(bool at_mark) {
    if (at_mark) t_mark_open = false;
    if (side_disabled()) return;
    if (Lane *l = current_lane()) {
        if (!lane_side_ready(l)) return;
        if (!at_mark) XR_HIP(hipEventRecord(l->fork_event, l->stream));
        XR_HIP(hipStreamWaitEvent(l->side, l->fork_event, 0));
        l->on_side = true;
        l->side_active = true;
        mode = 2;
        return;
    }
    if (t_exclusive_depth == 0) return; // a shared call on a caller's stream: several threads may be here, no common side stream
    Engine &e = engine();
    if (!at_mark) XR_HIP(hipEventRecord(e.fork_event, e.stream));
    XR_HIP(hipStreamWaitEvent(e.side, e.fork_event, 0));
    e.on_side = true;
    g_side_active = true;
    mode = 1;
}
SideScope::~SideScope() {
    if (mode == 2) {
        Lane *l = current_lane();
        l->on_side = false;
        (void)hipEventRecord(l->join_event, l->side);
    } else if (mode == 1) {
        g_engine.on_side = false;
        (void)hipEventRecord(g_engine.join_event, g_engine.side);
    }
}
static bool g_fork2_pending = false; // (exclusive path only) the second side stream holds work the next side_join must wait for
SideForkScope::SideForkScope() {
    Engine &e = engine();
    if (option(OPT_SIDE_FORK) == 0 || current_lane() || !e.on_side || t_stream_override || !e.side2) return;
    XR_HIP(hipEventRecord(e.fork2_event, e.side));
    XR_HIP(hipStreamWaitEvent(e.side2, e.fork2_event, 0));
    prev = t_stream_override;
    t_stream_override = e.side2;
    active = launching = true;
}
void SideForkScope::end_launches() {
    if (!launching) return;
    launching = false;
    t_stream_override = prev;
    (void)hipEventRecord(g_engine.join2_event, g_engine.side2);
}
SideForkScope::~SideForkScope() {
    if (!active) return;
    end_launches();
    // (the MAIN stream waits for the second stream's event itself, in side_join: made to wait for it here, the side stream put a
    // second event hop -- ~10 us -- between the second stream's last kernel and the main stream's next one)
    g_fork2_pending = true;
}
void side_join() {
    if (side_disabled()) return;
    if (Lane *l = current_lane()) {
        if (!l->side_active) return;
        XR_HIP(hipStreamWaitEvent(l->stream, l->join_event, 0));
        l->side_active = false; // everything enqueued on the lane's stream from here on runs behind the side work
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto &mine = g_lane_free[l];
        for (auto &kv : g_lane_deferred[l]) mine.emplace(kv.first, kv.second);
        g_lane_deferred[l].clear();
        return;
    }
    if (t_exclusive_depth == 0) return;
    Engine &e = engine();
    XR_HIP(hipStreamWaitEvent(e.stream, e.join_event, 0));
    if (g_fork2_pending) {
        XR_HIP(hipStreamWaitEvent(e.stream, e.join2_event, 0));
        g_fork2_pending = false;
    }
}

void prof_flush() {
    if (g_pending.empty()) return;
    (void)hipStreamSynchronize(g_engine.stream);
    (void)hipStreamSynchronize(g_engine.side);
    for (auto &l : g_lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        if (l.side) (void)hipStreamSynchronize(l.side);
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &p : g_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            auto it = g_prof.find(p.name);
            if (it == g_prof.end()) {
                g_prof_order.push_back(p.name);
                it = g_prof.emplace(p.name, ProfRec{}).first;
            }
            it->second.launches += 1;
            it->second.ms += ms;
        }
        g_event_pool.push_back(p.e0);
        g_event_pool.push_back(p.e1);
    }
    g_pending.clear();
}

} // namespace xr

using namespace xr;

extern "C" {

const char *xr_last_error(void) { return g_err; }

int xr_version(void) { return 100; }

int xr_set_stream(void *hip_stream, int external, int async_dev) {
    XR_API_BEGIN
    Engine &e = engine();
    XR_HIP(hipStreamSynchronize(e.stream)); // nothing of the engine may still be in flight on the old stream
    e.stream = external ? static_cast<hipStream_t>(hip_stream) : e.own_stream;
    e.async_dev = external && async_dev;
    XR_API_END
}

// ---- host-side helpers of the Python layer (no device involved): the copies the reference's constructors make
// (Ugrid2d.__init__: face_node_connectivity.copy(), contiguous node coordinates) through the library's host thread pool --
// a 24 MB numpy copy into a fresh array is 2-3 ms of single-threaded page faults, eight threads do it in 0.5 ms.
int xr_host_copy(void *dst, const void *src, int64_t bytes) {
    try {
        XR_REQUIRE(bytes >= 0 && ((dst && src) || bytes == 0), XR_ERR_INVALID, "xr_host_copy: bad arguments");
        char *d = static_cast<char *>(dst);
        const char *c = static_cast<const char *>(src);
        parallel_ranges((size_t)bytes, 4096, [=](size_t b, size_t e) { memcpy(d + b, c + b, e - b); });
        return XR_OK;
    } catch (const Failure &f) {
        return f.code;
    }
}

int xr_host_interleave2(const double *x, int64_t x_stride, const double *y, int64_t y_stride, int64_t n, double *out_xy) {
    try {
        XR_REQUIRE(n >= 0 && ((x && y && out_xy) || n == 0), XR_ERR_INVALID, "xr_host_interleave2: bad arguments");
        parallel_ranges((size_t)n, 512, [=](size_t b, size_t e) {
            for (size_t i = b; i < e; i++) {
                out_xy[2 * i] = x[(int64_t)i * x_stride];
                out_xy[2 * i + 1] = y[(int64_t)i * y_stride];
            }
        });
        return XR_OK;
    } catch (const Failure &f) {
        return f.code;
    }
}

int xr_set_option(const char *name, int64_t value) {
    XR_API_BEGIN
    XR_REQUIRE(name, XR_ERR_INVALID, "xr_set_option: NULL name");
    const int i = option_index(name);
    XR_REQUIRE(i >= 0, XR_ERR_INVALID, "xr_set_option: unknown option '%s'", name);
    options_init();
    g_options[i].store(value, std::memory_order_relaxed);
    XR_API_END
}

int xr_get_option(const char *name, int64_t *value) {
    XR_API_BEGIN
    XR_REQUIRE(name && value, XR_ERR_INVALID, "xr_get_option: NULL argument");
    const int i = option_index(name);
    XR_REQUIRE(i >= 0, XR_ERR_INVALID, "xr_get_option: unknown option '%s'", name);
    *value = option((Option)i);
    XR_API_END
}

int xr_set_async(int on) {
    XR_API_BEGIN
    Engine &e = engine();
    XR_HIP(hipStreamSynchronize(e.stream));
    e.main_busy = false;
    e.own_async = on != 0;
    XR_API_END
}

int xr_device_count(int *count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    if (count) *count = c;
    return XR_OK;
}

int xr_init(int device) {
    XR_API_BEGIN
    engine_init(device);
    XR_API_END
}

int xr_current_device(int *device) {
    XR_API_BEGIN
    *device = engine().device;
    XR_API_END
}

int xr_trim_pool(void) {
    XR_API_BEGIN
    pool_trim();
    XR_API_END
}

int xr_dev_alloc(int64_t bytes, void **ptr_out) {
    XR_API_BEGIN
    XR_REQUIRE(bytes >= 0 && ptr_out, XR_ERR_INVALID, "xr_dev_alloc: bad arguments");
    *ptr_out = pool_alloc((size_t)bytes);
    XR_API_END
}

int xr_dev_free(void *ptr) {
    XR_API_BEGIN
    pool_free(ptr);
    XR_API_END
}

int xr_dev_upload(void *dst_dev, const void *src_host, int64_t bytes) {
    XR_API_BEGIN
    h2d(dst_dev, src_host, (size_t)bytes);
    XR_API_END
}

int xr_dev_download(void *dst_host, const void *src_dev, int64_t bytes) {
    XR_API_BEGIN
    if (bytes > 0) {
        XR_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, engine().stream));
        XR_HIP(hipStreamSynchronize(engine().stream));
    }
    XR_API_END
}

int xr_dev_sync(void) {
    XR_API_BEGIN
    stream_sync();
    XR_API_END
}

int xr_prof_enable(int on) {
    XR_API_BEGIN
    engine();
    prof_flush();
    engine().prof = on != 0;
    XR_API_END
}

int xr_prof_reset(void) {
    XR_API_BEGIN
    prof_flush();
    g_prof.clear();
    g_prof_order.clear();
    XR_API_END
}

int xr_prof_count(int *count) {
    XR_API_BEGIN
    prof_flush();
    *count = (int)g_prof_order.size();
    XR_API_END
}

int xr_prof_get(int i, char *name, int name_cap, int64_t *launches, double *total_ms) {
    XR_API_BEGIN
    prof_flush();
    XR_REQUIRE(i >= 0 && i < (int)g_prof_order.size(), XR_ERR_INVALID, "xr_prof_get: index out of range");
    const std::string &n = g_prof_order[i];
    const ProfRec &r = g_prof[n];
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s", n.c_str());
    }
    if (launches) *launches = r.launches;
    if (total_ms) *total_ms = r.ms;
    XR_API_END
}

} // extern "C"
