// xr_internal.h -- shared internals of libxugrid_amd.so (engine context, HBM block pool,
// launch/profiling helpers, device scan).  gfx950 / wave64 only.
#pragma once
#include <memory>

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/xugrid_amd.h"

namespace xr {

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);

struct Failure {
    int code;
};

#define XR_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            xr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,      \
                          __LINE__);                                                            \
            throw xr::Failure{XR_ERR_HIP};                                                      \
        }                                                                                       \
    } while (0)

#define XR_REQUIRE(cond, code, ...)                                                             \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            xr::set_error(__VA_ARGS__);                                                         \
            throw xr::Failure{code};                                                            \
        }                                                                                       \
    } while (0)

// Every extern "C" entry point wraps its body:  XR_API_BEGIN ... XR_API_END.  XR_API_BEGIN holds the engine EXCLUSIVELY
// (everything that builds or changes objects); the apply entry points use XR_API_BEGIN_SHARED instead: any number of
// threads apply finished weights at the same time (dask's threaded scheduler calls _regrid from several threads,
// regridder.py:177-185), each on a lane of its own -- a stream, staging buffers and a stream-ordered free list -- and
// only wait for each other where they share a lane.
#define XR_API_BEGIN                                                                            \
    try {                                                                                       \
        xr::ExclusiveScope _guard;
#define XR_API_BEGIN_SHARED                                                                     \
    try {                                                                                       \
        xr::SharedScope _guard;
#define XR_API_END                                                                              \
    return XR_OK;                                                                               \
    }                                                                                           \
    catch (const xr::Failure &f) {                                                              \
        return f.code;                                                                          \
    }                                                                                           \
    catch (const std::exception &e) {                                                           \
        xr::set_error("internal error: %s", e.what());                                          \
        return XR_ERR_INVALID;                                                                  \
    }

std::recursive_mutex &engine_mutex();
std::shared_mutex &engine_rw();

// exclusive use of the engine by this thread (re-entrant: entry points may call each other)
struct ExclusiveScope {
    ExclusiveScope();
    ~ExclusiveScope();
};
// shared use: concurrent applies; takes a lane for the calling thread (see Lane below)
struct SharedScope {
    SharedScope();
    ~SharedScope();
    bool leased = false;
    bool serial = false; // no lane (caller's stream / asynchronous mode): holds the main stream's turn
};

// ---------------------------------------------------------------------------------------------
// engine context: one device + one stream per process
// ---------------------------------------------------------------------------------------------
struct Engine {
    int device = -1;
    hipStream_t stream = nullptr;     // the stream every kernel, copy and event of the engine goes to
    hipStream_t own_stream = nullptr; // created by engine_init; `stream` points elsewhere after xr_set_stream
    bool async_dev = false;           // *_dev entry points return without waiting (caller shares the stream)
    int num_cu = 256;
    void *pinned = nullptr; // small pinned staging buffer for scalar read-backs
    // "mailbox": pinned host memory that kernels write the few scalars the host is waiting for into
    // (candidate count, nnz, overflow flag).  The host then waits on an event instead of issuing a
    // device-to-host blit + stream synchronisation per scalar.
    volatile int32_t *mailbox = nullptr; // [1024]
    hipEvent_t mail_event = nullptr;
    bool prof = false;
    // side stream for work that is independent of the main chain for a while (the big target faces of xr_overlap):
    // forked / joined with events, never synchronised with the host on its own
    hipStream_t side = nullptr;
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    // a second side stream, forked FROM the side stream and joined back into it (SideForkScope): two independent tails of the
    // side chain (the light and the bitmap rows of the big faces) beside each other instead of one behind the other
    hipStream_t side2 = nullptr;
    hipEvent_t fork2_event = nullptr, join2_event = nullptr;
    hipEvent_t aux_event = nullptr; // a point INSIDE the side stream's work the main stream may wait for before the full join
    bool on_side = false; // XR_LAUNCH and the kernel timer go to the side stream while set
    int32_t mail_seq = 0; // sequence number of the last mailbox publication the host asked for (mailbox_wait_seq)
    // xr_set_async(1): the *_dev entry points, xr_mesh_invalidate and xr_overlap_apply_dev return with their kernels
    // still in flight on the engine's own stream; the caller orders itself with xr_dev_sync (or any synchronous call)
    bool own_async = false;
    bool main_busy = false; // an asynchronous call returned: the main stream may hold work the host has not waited for
};
inline hipStream_t launch_stream();
void mailbox_wait(); // everything enqueued so far has executed and its mailbox writes are visible
// The same without an event: the publishing kernel stores `seq` in mailbox[MAIL_SEQ_SLOT] (system scope) BEHIND its other
// mailbox words and the host polls that word -- no event record / barrier packet on the stream, no interrupt-driven wake-up
// (an event costs ~5 us of stream time and ~10 us of host latency per wait; XR_MAIL_POLL=0 restores it).  Bounded: after
// ~20 ms without the word the stream is synchronised and the word checked once more (a failed kernel raises the HIP error).
static constexpr int MAIL_SEQ_SLOT = 1023;
int32_t mailbox_next_seq();
void mailbox_wait_seq(int32_t seq);
// host side of the same protocol for any pinned word a kernel publishes (the mesh statistics)
bool poll_pinned_f64(const volatile double *word, double expected);
Engine &engine();      // initialises device 0 on first use
void engine_init(int device);

// ---------------------------------------------------------------------------------------------
// HBM block pool (size-bucketed cache so the timed path never calls hipMalloc/hipFree)
// ---------------------------------------------------------------------------------------------
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_trim();

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        p = static_cast<T *>(pool_alloc((count ? count : 1) * sizeof(T)));
    }
    void release() {
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
    }
    T *get() const { return p; }
    size_t bytes() const { return n * sizeof(T); }
};

// A buffer of points that is either the holder's own or SHARED with the mesh whose face centroids it holds (xr_mesh::centroids_dev:
// the reference caches `Ugrid2d.centroids` on the grid, ugrid2d.py; here the device copy is kept until the mesh is invalidated or
// destroyed -- a handle that shares it keeps it alive beyond that).
struct PointsBuf {
    DevBuf<double> own;
    std::shared_ptr<DevBuf<double>> shared;
    void alloc(size_t count) {
        shared.reset();
        own.alloc(count);
    }
    void share(std::shared_ptr<DevBuf<double>> s) {
        own.release();
        shared = std::move(s);
    }
    double *get() const { return shared ? shared->get() : own.get(); }
};

// "Zero at rest" scratch words: int32 arrays the engine keeps between calls and that are ALL ZERO whenever no call is
// using them -- so the hot path has no memset (a hipMemsetAsync is a 4-5 us kernel plus its launch gap; an odd length
// even two).  The user's own kernels restore the zeros: a histogram that is counted up and handed out down to zero
// (mesh index build), counters that the kernel publishing them clears (xr_overlap).  zero_scratch() -> the words (all
// zero in stream order), or nullptr when the cache does not apply (not the engine's own stream, too large): the caller
// then allocates and clears as before.  zero_scratch_done() says the restoring kernel has been enqueued; a call that
// fails before that leaves the slot marked dirty and the next user clears it.  Exclusive calls only.
static constexpr int ZERO_SLOTS = 2;                                     // 0: bucket histogram, 1: overlap counters
static constexpr size_t ZERO_SCRATCH_MAX_WORDS = (size_t)16 << 20;       // 64 MB; larger histograms are not kept
int32_t *zero_scratch(int slot, size_t n_words);
void zero_scratch_done(int slot);

void h2d(void *dst, const void *src, size_t bytes);       // synchronous w.r.t. the host
void d2h(void *dst, const void *src, size_t bytes);       // stream-ordered, then synchronised
// ... with work for the device to do meanwhile: `behind` is called once the copy has been enqueued and before the host waits for it
// -- what it launches runs while the value travels (a kernel that needs the device-side result but not the host's knowledge of it).
void d2h(void *dst, const void *src, size_t bytes, const std::function<void()> &behind);
void stream_sync();
// XR_HOST_STAMPS=1: wall-clock stamps at a few points of the host path, averaged per point and printed at exit (where the
// host's time goes between the mailbox of one weight build and the first kernel of the next -- profiles/r04_host_stamps.txt)
void host_stamp(int point);
void dev_call_done(); // end of a *_dev entry point: stream_sync() unless the caller shares the engine's stream
// before a handle's blocks go back to the pool: wait for the device -- unless the engine runs asynchronously on its one
// own stream (xr_set_async), where the pool's stream-ordered reuse already orders every later user behind the last one
void release_point();
// Large host <-> device copies of pageable memory through two pinned staging buffers: worker threads move the data
// between the caller's pages and the staging buffer (page faults of a fresh result array included) while the DMA
// engine moves the previous piece -- 2-3x the rate of a plain hipMemcpy on pageable memory.  Synchronous.
void h2d_big(void *dst, const void *src, size_t bytes);
// pieces of a host array through the pinned staging buffers, filled by the host thread pool while the previous piece is on
// its way (fill(pinned_dst, first_byte_of_the_device_data, n_bytes)); does NOT wait for the last DMA
void h2d_staged(void *dst, size_t bytes, const std::function<void(char *, size_t, size_t)> &fill);
void parallel_ranges(size_t n, size_t align, const std::function<void(size_t, size_t)> &fn);
void d2h_big(void *dst, const void *src, size_t bytes);
template <typename T> T read_scalar(const T *dev) {
    T v;
    d2h(&v, dev, sizeof(T));
    return v;
}
template <typename T> T read_scalar(const T *dev, const std::function<void()> &behind) {
    T v;
    d2h(&v, dev, sizeof(T), behind);
    return v;
}

// ---------------------------------------------------------------------------------------------
// launch helper with optional hipEvent timing per kernel name
// ---------------------------------------------------------------------------------------------
struct ProfScope {
    const char *name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on_side = false;
    hipStream_t lane_stream = nullptr;
    explicit ProfScope(const char *name);
    ~ProfScope();
};
void prof_flush(); // resolve pending events into the per-name table

#define XR_LAUNCH(name, kernel, grid, block, shmem, ...)                                        \
    do {                                                                                        \
        xr::ProfScope _ps(name);                                                                \
        hipLaunchKernelGGL(kernel, grid, block, shmem, xr::launch_stream(), __VA_ARGS__);       \
        XR_HIP(hipGetLastError());                                                              \
    } while (0)

// A lane: what a thread needs to use the device next to other threads.  Lane streams carry whole apply calls; a call
// ends with the lane drained, so blocks a lane frees can go back to the common pool then.
struct Lane {
    hipStream_t stream = nullptr;
    char *stage[2] = {nullptr, nullptr}; // pinned staging buffers of the big host <-> device copies
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // the lane's own side stream (SideScope inside a shared call: e.g. the long rows of an apply beside the short ones)
    hipStream_t side = nullptr;
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    bool on_side = false;     // launches go to `side`
    bool side_active = false; // forked and not yet joined: blocks freed meanwhile are parked (see pool_free)
    std::mutex busy;
};
Lane *current_lane(); // the calling thread's lane, nullptr on the engine's own (exclusive) path
bool exclusive_held(); // the calling thread is inside an entry point that holds the engine exclusively

hipStream_t stream_override(); // a stream the calling thread has redirected its launches to (StreamOverride), or nullptr
inline hipStream_t launch_stream() {
    if (hipStream_t s = stream_override()) return s;
    if (Lane *l = current_lane()) return l->on_side ? l->side : l->stream;
    return engine().on_side ? engine().side : engine().stream;
}
// RAII: launches of the calling thread go to `s` (a stream the caller owns and orders itself with events)
struct StreamOverride {
    explicit StreamOverride(hipStream_t s);
    ~StreamOverride();
    hipStream_t prev;
};

// RAII: launches inside the scope go to the side stream, which first waits for everything enqueued on the main
// stream so far; join() makes the main stream wait for the side stream's work
// Works on the engine's own (exclusive) path and inside a shared call that has a lane; elsewhere (a shared call bound to
// a caller's stream) the scope is a no-op and the work simply runs in line.
struct SideScope {
    // at_mark: the side work depends on what was enqueued up to the last side_mark() of this thread, not on what followed it
    // (a long kernel launched on the main stream first: the side work runs beside it, and the host reaches that launch
    // without the side launches in front of it)
    explicit SideScope(bool at_mark = false);
    ~SideScope();
    int mode = 0; // 0: in line, 1: engine side stream, 2: the lane's side stream
};
// RAII, inside a SideScope of the engine's own path: launches go to a SECOND side stream that waits for what the side stream
// holds so far; the destructor makes the side stream wait for them (so the one join of the SideScope covers both).  Anywhere
// else (lanes, no side stream) the scope does nothing and the work runs in line on whatever stream is current.
struct SideForkScope {
    SideForkScope();
    void end_launches(); // launches go to the side stream again (what follows there runs BESIDE the second stream's work)
    ~SideForkScope();    // the side stream waits for the second stream's work
    bool active = false, launching = false;
    hipStream_t prev = nullptr;
};
void side_join();
bool side_mark(); // record the fork point now; false: no side stream here (SideScope(true) would run in line, i.e. BEHIND what follows)

// ---------------------------------------------------------------------------------------------
// Run-time options: every switch a test or a measurement script flips.  The table (name, default, meaning) is in
// xr_engine.hip; values are read ONCE from the environment (XR_<NAME IN CAPITALS>) when the library is first used -- the one
// getenv call site of the library, no getenv on any call path or worker thread -- and changed afterwards only through
// xr_set_option.  None of them changes a result.
// ---------------------------------------------------------------------------------------------
enum Option : int {
    OPT_OVERLAP_FUSED,     // 1: dense meshes of <= 4 nodes per face take the one-round-trip pipeline; 0: the general kernel chain
    OPT_QUEUE_MARGIN,      // > 0: capacity of the big faces' pair queue (tests force the regrow path with a tiny one)
    OPT_CLIP_QUAD,         // 1: quadrilateral targets x triangle source through k_clip_quad_tri (general chain); 0: k_clip_small
    OPT_DUST,              // 1: rounding dust confirmed by the reference's pre-clip tests; 0: the round-3 behaviour
    OPT_NO_SIDE,           // 1: side-stream work in line on the main stream (every kernel alone on the device)
    OPT_SIDE_FORK,         // 1: the big faces' bitmap rows on a second side stream beside the light ones
    OPT_DEBUG,             // bits: 1 one line per weight build, 2 apply-plan statistics, 4 Voronoi sizes (stderr)
    OPT_HOST_STAMPS,       // 1: wall-clock stamps along the host path of a weight build, averages at exit
    OPT_APPLY_PLAN,        // 1: K >= 8 variables through the planned kernels; 0: the direct kernel
    OPT_APPLY_CONTRACT,    // 1: fused multiply-adds + one reciprocal per row in the planned kernel (NOT bit-identical; <= (n + 2) ulp)
    OPT_PLAN_MERGE,        // -1: per matrix from the blocks' use of the source lines; 0 / 1: force the per-block / the merged plan
    OPT_PLAN_DBG,          // measurement bits of k_apply_plan: 1 no gathers, 2 no stores, 4 plain instead of non-temporal stores
    OPT_APPLY_CHUNK_BYTES, // > 0: staging budget of the host <-> device chunks of xr_apply_csr (tests force many chunks)
    OPT_OUTER_APPLY,       // 0: auto; 1: matrix-free apply of factored raster weights; 2: through the materialised CSR
    OPT_EDGE_BIG,          // > 0: cells of an edge's box from which on it goes to the wave-per-edge kernels
    OPT_EDGE_STAGE,        // > 0: candidates a wave of 64 edges may stage (<= 1024; tests force the overflow into the wave-per-edge kernel)
    OPT_EDGE_QUEUE,        // > 0: candidate pairs the edge queue holds at first (tests force the regrow path with a tiny one)
    OPT_EDGE_SORT,         // 1: the edges are walked in the order of the grid tiles of their midpoints; 0: as they come
    OPT_MAIL_POLL,         // 1: the host polls the mailbox's sequence word, small copies without a stream synchronisation
    OPT_POINTS_DEFER,      // 1: xr_locate_flags_begin defers its kernels to the next call that has the device to spare
    OPT_INGEST_DEVICE,     // 1 ("device"): connectivity validated and narrowed on the device instead of while it is staged
    OPT_STATS_SAMPLE,      // 1: tree statistics from a sample of the faces (exact on demand); 0: always exact
    OPT_FORCE_QUERY_SORT,  // 1: Morton query order even for coherent numberings
    OPT_EARLY_APPLY,       // 1: the apply of xr_overlap_apply_dev enqueued in front of the size read-back
    OPT_STAR_FLAG,         // 1: "inside the source grid" of a barycentric construction from the faces around the located Voronoi
                           //    cell (k_star_flag), the grid walk only for the points they leave open; 0: the grid walk for all
    OPT_COUNT
};
int64_t option(Option o);

inline unsigned div_up(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// device primitives implemented in xr_scan.hip
// ---------------------------------------------------------------------------------------------
// out[0..n] = exclusive prefix sum of in[0..n-1] (out has n+1 entries; out[n] = total).
// in and out may alias when out == in is NOT required; separate buffers expected.
// Optionally the grand total (and one extra device word, e.g. an overflow counter) is also deposited in host
// memory (engine().mailbox) by the kernel that writes out[n].
void exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *host_total = nullptr,
                        const int32_t *extra_src = nullptr, int32_t *extra_host = nullptr);
void fill_i32(int32_t *p, int32_t v, int64_t n);
void fill_f64(double *p, double v, int64_t n);

} // namespace xr
