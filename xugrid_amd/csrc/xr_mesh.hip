// xr_mesh.hip -- mesh handle: upload, per-face preparation (fill/length/CCW/bbox/area/vertex
// gather), the two spatial orderings (query order, hierarchical-grid tree index), area / centroid
// kernels.
//
// Replaces, on the device, what the reference does when it constructs
// numba_celltree.CellTree2d(node_coordinates, face_node_connectivity, -1)
// (xugrid/ugrid/ugrid2d.py:908-921) and what intersect_faces does to the query mesh
// (cast, counter-clockwise normalisation, bounding boxes).  The index is NOT a port of the
// cell tree: a bounding-interval hierarchy is a pointer-chasing structure; on CDNA4 a
// counting-sorted hierarchical grid gives contiguous, coalescable record runs per query row.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include <memory>

#include <atomic>

#include "xr_objects.h"
#include "xr_agg.h"

namespace xr {

static constexpr int PREP_BLOCK = 256;

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_down(v, d, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_down(v, d, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// per-face preparation.  MC > 0: compile-time vertices per face; MC == 0: runtime m <= 32.
// connectivity.area on the caller's order (connectivity.py:372-382, 615-633): fill slots and the closing slot
// repeat node 0; everything relative to node 0.  Only `relative` overlap weights and xr_mesh_area read it, so it
// is computed on demand (mesh_area) and not by the preparation pass.
template <int MC>
__global__ void __launch_bounds__(256)
k_face_area(const double *__restrict__ node_xy, const int32_t *__restrict__ faces_raw, int64_t n_face, int m_rt,
            double *__restrict__ area) {
    constexpr int MA = MC > 0 ? MC : XR_MAX_FACE_NODES;
    const int m = MC > 0 ? MC : m_rt;
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    int face[MA];
#pragma unroll
    for (int j = 0; j < MA; j++)
        if (j < m) face[j] = faces_raw[f * m + j];
    const P2 p0 = load_p2(node_xy, face[0]);
    double det = 0.0;
#pragma unroll
    for (int i = 0; i < MA; i++) {
        if (i < m) {
            const int ia = face[i] < 0 ? face[0] : face[i];
            int ib = face[0];
            if (i + 1 < m) ib = face[i + 1] < 0 ? face[0] : face[i + 1];
            const P2 a = load_p2(node_xy, ia), b = load_p2(node_xy, ib);
            const double ax = a.x - p0.x, ay = a.y - p0.y;
            const double bx = b.x - p0.x, by = b.y - p0.y;
            det += ax * by - ay * bx;
        }
    }
    area[f] = 0.5 * fabs(det);
}

// Tail of the two statistics kernels: the block's partial goes to HBM with agent-scope (sc1) stores, a counter says how
// many blocks have done so, and the LAST block to arrive reduces all partials -- in the fixed order of k_reduce_stats, so the
// result does not depend on which block that is -- and publishes the eight numbers: device copy, pinned host copy and,
// behind them, the host's sequence word (system scope), which the host polls (mesh_read_stats).  One launch and one
// cross-queue hop less per mesh than a separate one-block reduction kernel, and no event on the stream.
// Hand-off: 8-byte agent-scope atomics on both sides (MI355X_MICROARCH.md, inter-workgroup visibility) + an agent acquire
// in the one reducing block; the counter is zero at rest (the last block clears it).
static constexpr int STATS_SHARDS = 64;
struct StatsTail {
    double *partials;     // [nb][8]
    unsigned *done;       // [16 * (1 + STATS_SHARDS)] arrival counters, one per 64-byte line, zero at rest
    double *stats;        // [8] device copy
    double *stats_host;   // [9] pinned: 8 statistics + the sequence word
    double seq;
};

__device__ __forceinline__ void stats_tail(const StatsTail &t, const double (&a)[8], double (*lds)[PREP_BLOCK / 64]) {
    __shared__ int s_last;
    const int64_t nb = gridDim.x;
    if (threadIdx.x == 0) {
        double *p = t.partials + (int64_t)blockIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) __hip_atomic_store(&p[i], a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Two-level arrival count: thousands of returning atomics on ONE address serialise (~10 ns each: 3900 blocks that
        // finish together = 40 us, measured); STATS_SHARDS counters on lines of their own take 1/STATS_SHARDS of the blocks
        // each, and only the last arrival of a shard goes on to the second-level word.
        const int shard = (int)(blockIdx.x % STATS_SHARDS);
        const unsigned in_shard = (unsigned)((nb - shard + STATS_SHARDS - 1) / STATS_SHARDS);
        unsigned *first = t.done + 16 * (1 + shard);
        bool last = false;
        if (__hip_atomic_fetch_add(first, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(first, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned shards = (unsigned)(nb < STATS_SHARDS ? nb : STATS_SHARDS);
            last = __hip_atomic_fetch_add(t.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(t.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    double a0 = INFINITY, a1 = -INFINITY, a2 = INFINITY, a3 = -INFINITY, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
    for (int64_t i = threadIdx.x; i < nb; i += PREP_BLOCK) {
        const double *p = t.partials + i * 8;
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __hip_atomic_load(&p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a0 = fmin(a0, v[0]); a1 = fmax(a1, v[1]); a2 = fmin(a2, v[2]);
        a3 = fmax(a3, v[3]); a4 += v[4]; a5 = fmax(a5, v[5]); a6 = fmax(a6, v[6]); a7 += v[7];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a0 = wave_min(a0); a1 = wave_max(a1); a2 = wave_min(a2); a3 = wave_max(a3); a4 = wave_sum(a4); a5 = wave_max(a5);
    a6 = wave_max(a6);
    a7 = wave_sum(a7);
    __syncthreads(); // (lds still holds the block's own wave partials for thread 0 above)
    if (lane == 0) {
        lds[0][wave] = a0; lds[1][wave] = a1; lds[2][wave] = a2;
        lds[3][wave] = a3; lds[4][wave] = a4; lds[5][wave] = a5; lds[6][wave] = a6; lds[7][wave] = a7;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PREP_BLOCK / 64; w++) {
            a0 = fmin(a0, lds[0][w]); a1 = fmax(a1, lds[1][w]); a2 = fmin(a2, lds[2][w]);
            a3 = fmax(a3, lds[3][w]); a4 += lds[4][w]; a5 = fmax(a5, lds[5][w]); a6 = fmax(a6, lds[6][w]); a7 += lds[7][w];
        }
        const double r[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            t.stats[k] = r[k];
            t.stats_host[k] = r[k];
        }
        __threadfence_system();
        __hip_atomic_store(&t.stats_host[8], t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// LIGHT: statistics only (a mesh that is only ever the TREE: its records are built from the raw mesh by
// k_spatial_scatter, nothing reads caller-order per-face arrays)
template <int MC, bool LIGHT>
__global__ void __launch_bounds__(PREP_BLOCK)
k_prepare_faces(const double *__restrict__ node_xy, const int32_t *__restrict__ faces_raw, int64_t n_face,
                int m_rt, double *__restrict__ fxy, uint8_t *__restrict__ len_out, double *__restrict__ bbox,
                double *__restrict__ partials, StatsTail tail) {
    constexpr int MA = MC > 0 ? MC : XR_MAX_FACE_NODES;
    const int m = MC > 0 ? MC : m_rt;
    const int64_t f = (int64_t)blockIdx.x * PREP_BLOCK + threadIdx.x;

    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, ext = 0.0, diag = 0.0, jump = 0.0;
    if (f < n_face) {
        int face[MA];
#pragma unroll
        for (int j = 0; j < MA; j++)
            if (j < m) face[j] = faces_raw[f * m + j];

        int n;
        bool flip;
        face_shape<MA>(node_xy, face, m, n, flip);
        if (!LIGHT) len_out[f] = (uint8_t)n;
        double2 *fx = reinterpret_cast<double2 *>(fxy) + f * m;
#pragma unroll
        for (int j = 0; j < MA; j++) {
            if (j < n) {
                const P2 p = load_p2(node_xy, face[j]);
                xmin = fmin(xmin, p.x);
                xmax = fmax(xmax, p.x);
                ymin = fmin(ymin, p.y);
                ymax = fmax(ymax, p.y);
                if (!LIGHT && fxy) fx[flip ? n - 1 - j : j] = make_double2(p.x, p.y);
            }
        }
        if (!LIGHT) reinterpret_cast<double4 *>(bbox)[f] = make_double4(xmin, xmax, ymin, ymax);
        ext = fmax(xmax - xmin, ymax - ymin);
        const double dx = xmax - xmin, dy = ymax - ymin;
        diag = sqrt(dx * dx + dy * dy);
    }

    // numbering coherence: distance between the bbox centres of consecutive faces (within a wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const double cx = 0.5 * (xmin + xmax), cy = 0.5 * (ymin + ymax);
        const double px = __shfl_up(cx, 1, 64), py = __shfl_up(cy, 1, 64);
        if (lane > 0 && f < n_face) jump = fmax(fabs(cx - px), fabs(cy - py));
    }
    // block partials: bounds, sum and max of the bbox extents (fixed order -> deterministic)
    __shared__ double lds[8][PREP_BLOCK / 64];
    double r0 = wave_min(xmin), r1 = wave_max(xmax), r2 = wave_min(ymin), r3 = wave_max(ymax);
    double r4 = wave_sum(ext), r5 = wave_max(ext), r6 = wave_max(diag), r7 = wave_sum(jump);
    if (lane == 0) {
        lds[0][wave] = r0; lds[1][wave] = r1; lds[2][wave] = r2;
        lds[3][wave] = r3; lds[4][wave] = r4; lds[5][wave] = r5; lds[6][wave] = r6; lds[7][wave] = r7;
    }
    __syncthreads();
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0) {
        double a0 = lds[0][0], a1 = lds[1][0], a2 = lds[2][0], a3 = lds[3][0], a4 = lds[4][0], a5 = lds[5][0], a6 = lds[6][0], a7 = lds[7][0];
        for (int w = 1; w < PREP_BLOCK / 64; w++) {
            a0 = fmin(a0, lds[0][w]); a1 = fmax(a1, lds[1][w]); a2 = fmin(a2, lds[2][w]);
            a3 = fmax(a3, lds[3][w]); a4 += lds[4][w]; a5 = fmax(a5, lds[5][w]); a6 = fmax(a6, lds[6][w]); a7 += lds[7][w];
        }
        if (tail.done == nullptr) { // (separate reduction kernel behind this one: XR_STATS_TAIL=0)
            double *p = partials + (int64_t)blockIdx.x * 8;
            p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3; p[4] = a4; p[5] = a5; p[6] = a6; p[7] = a7;
        }
        a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3; a[4] = a4; a[5] = a5; a[6] = a6; a[7] = a7;
    }
    if (tail.done != nullptr) stats_tail(tail, a, lds);
}

// Statistics of a mesh that is only ever the TREE, from a SAMPLE of its faces: the index only needs the bounds (exact:
// taken from the node array, a coalesced stream) and the mean bbox extent (to choose the cell size: every `stride`-th block
// of 256 faces is looked at).  Block b reads faces [b * stride * 256, + 256) and its slice of the nodes; partials in the
// layout of k_prepare_faces with [7] = faces sampled.  Nothing the RESULTS depend on comes from the sample: the grid is an
// accelerator (any cell size gives the same pairs), the number of levels is then derived from the domain size, and the one
// statistic with a meaning outside the index -- the largest bbox diagonal behind the default tolerance -- is recomputed over
// all faces when somebody asks for it (mesh_read_stats(need_exact)).
template <int MC>
__global__ void __launch_bounds__(PREP_BLOCK)
k_sample_stats(const double *__restrict__ node_xy, int64_t n_node, const int32_t *__restrict__ faces_raw, int64_t n_face,
               int m_rt, int stride, double *__restrict__ partials, StatsTail tail) {
    constexpr int MA = MC > 0 ? MC : XR_MAX_FACE_NODES;
    const int m = MC > 0 ? MC : m_rt;
    const int64_t f = ((int64_t)blockIdx.x * stride) * PREP_BLOCK + threadIdx.x;
    double ext = 0.0, diag = 0.0, cnt = 0.0;
    if (f < n_face) {
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        bool open = true;
#pragma unroll
        for (int j = 0; j < MA; j++) {
            if (j < m) {
                const int v = faces_raw[f * m + j];
                open = open && !(j >= 3 && v < 0);
                if (open) {
                    const P2 p = load_p2(node_xy, v);
                    xmin = fmin(xmin, p.x);
                    xmax = fmax(xmax, p.x);
                    ymin = fmin(ymin, p.y);
                    ymax = fmax(ymax, p.y);
                }
            }
        }
        const double dx = xmax - xmin, dy = ymax - ymin;
        ext = fmax(dx, dy);
        diag = sqrt(dx * dx + dy * dy);
        cnt = 1.0;
    }
    // this block's slice of the node array
    double nx0 = INFINITY, nx1 = -INFINITY, ny0 = INFINITY, ny1 = -INFINITY;
    const int64_t per = (n_node + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n_node ? i0 + per : n_node;
    const double2 *nodes = reinterpret_cast<const double2 *>(node_xy);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += PREP_BLOCK) {
        const double2 p = nodes[i];
        nx0 = fmin(nx0, p.x);
        nx1 = fmax(nx1, p.x);
        ny0 = fmin(ny0, p.y);
        ny1 = fmax(ny1, p.y);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double lds[8][PREP_BLOCK / 64];
    double r0 = wave_min(nx0), r1 = wave_max(nx1), r2 = wave_min(ny0), r3 = wave_max(ny1);
    double r4 = wave_sum(ext), r5 = wave_max(ext), r6 = wave_max(diag), r7 = wave_sum(cnt);
    if (lane == 0) {
        lds[0][wave] = r0; lds[1][wave] = r1; lds[2][wave] = r2;
        lds[3][wave] = r3; lds[4][wave] = r4; lds[5][wave] = r5; lds[6][wave] = r6; lds[7][wave] = r7;
    }
    __syncthreads();
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0) {
        double a0 = lds[0][0], a1 = lds[1][0], a2 = lds[2][0], a3 = lds[3][0], a4 = lds[4][0], a5 = lds[5][0], a6 = lds[6][0], a7 = lds[7][0];
        for (int w = 1; w < PREP_BLOCK / 64; w++) {
            a0 = fmin(a0, lds[0][w]); a1 = fmax(a1, lds[1][w]); a2 = fmin(a2, lds[2][w]);
            a3 = fmax(a3, lds[3][w]); a4 += lds[4][w]; a5 = fmax(a5, lds[5][w]); a6 = fmax(a6, lds[6][w]); a7 += lds[7][w];
        }
        if (tail.done == nullptr) { // (separate reduction kernel behind this one: XR_STATS_TAIL=0)
            double *p = partials + (int64_t)blockIdx.x * 8;
            p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3; p[4] = a4; p[5] = a5; p[6] = a6; p[7] = a7;
        }
        a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3; a[4] = a4; a[5] = a5; a[6] = a6; a[7] = a7;
    }
    if (tail.done != nullptr) stats_tail(tail, a, lds);
}

// ingest of the caller's connectivity (xr_mesh_create): fill -> -1, narrow to int32, validate
template <typename I>
__global__ void __launch_bounds__(256)
k_ingest_faces(const I *__restrict__ raw, int64_t cnt, int m, int64_t fill_value, int64_t n_node,
               int32_t *__restrict__ out, int64_t *__restrict__ err) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const int64_t v = (int64_t)raw[k];
    const int64_t f = k / m;
    const int j = (int)(k - f * m);
    if (v == fill_value || v == -1) {
        if (j < 3) atomicMin(reinterpret_cast<unsigned long long *>(err), (unsigned long long)f);
        out[k] = -1;
    } else {
        if (v < 0 || v >= n_node) atomicMin(reinterpret_cast<unsigned long long *>(err + 1), (unsigned long long)f);
        out[k] = (int32_t)v;
    }
}

// CCW-normalised connectivity (xr_mesh_faces; not on the hot path)
__global__ void __launch_bounds__(256) k_faces_ccw(const double *__restrict__ node_xy,
                                                  const int32_t *__restrict__ faces_raw, int64_t n_face, int m,
                                                  int64_t *__restrict__ out, bool caller_order) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    int face[XR_MAX_FACE_NODES];
    for (int j = 0; j < m; j++) face[j] = faces_raw[f * m + j];
    int n;
    bool flip;
    face_shape<XR_MAX_FACE_NODES>(node_xy, face, m, n, flip);
    if (caller_order) flip = false; // (fill normalised to -1, vertex order untouched)
    for (int j = 0; j < m; j++) out[f * m + j] = (flip && j < n) ? face[n - 1 - j] : face[j];
}

__global__ void __launch_bounds__(256) k_reduce_stats(const double *__restrict__ partials, int64_t nb,
                                                     double *__restrict__ stats, double *stats_host, double seq) {
    double a0 = INFINITY, a1 = -INFINITY, a2 = INFINITY, a3 = -INFINITY, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
    for (int64_t i = threadIdx.x; i < nb; i += 256) {
        const double *p = partials + i * 8;
        a0 = fmin(a0, p[0]); a1 = fmax(a1, p[1]); a2 = fmin(a2, p[2]);
        a3 = fmax(a3, p[3]); a4 += p[4]; a5 = fmax(a5, p[5]); a6 = fmax(a6, p[6]); a7 += p[7];
    }
    __shared__ double lds[8][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a0 = wave_min(a0); a1 = wave_max(a1); a2 = wave_min(a2); a3 = wave_max(a3); a4 = wave_sum(a4); a5 = wave_max(a5);
    a6 = wave_max(a6);
    a7 = wave_sum(a7);
    if (lane == 0) {
        lds[0][wave] = a0; lds[1][wave] = a1; lds[2][wave] = a2;
        lds[3][wave] = a3; lds[4][wave] = a4; lds[5][wave] = a5; lds[6][wave] = a6; lds[7][wave] = a7;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            a0 = fmin(a0, lds[0][w]); a1 = fmax(a1, lds[1][w]); a2 = fmin(a2, lds[2][w]);
            a3 = fmax(a3, lds[3][w]); a4 += lds[4][w]; a5 = fmax(a5, lds[5][w]); a6 = fmax(a6, lds[6][w]); a7 += lds[7][w];
        }
        stats[0] = a0; stats[1] = a1; stats[2] = a2; stats[3] = a3; stats[4] = a4; stats[5] = a5; stats[6] = a6;
        stats[7] = a7;
        stats_host[0] = a0; stats_host[1] = a1; stats_host[2] = a2; stats_host[3] = a3; stats_host[4] = a4;
        stats_host[5] = a5; stats_host[6] = a6; stats_host[7] = a7;
        // the host polls the sequence word (mesh_read_stats): behind the statistics, system scope
        __threadfence_system();
        __hip_atomic_store(&stats_host[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- flat vertex blocks of polygon meshes (m > DENSE_MAX_NODES): offsets = exclusive scan of the lengths, then one
// pass writes the len[r] CCW-normalised vertices of face perm[r] (perm == nullptr: r) from the raw mesh
__global__ void __launch_bounds__(256) k_widen_len(const uint8_t *__restrict__ len, int64_t n, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = len[i];
}

// The vertex blocks of a block's 256 faces are one contiguous stretch of the output: they are staged in LDS and written
// out as whole lines (a thread writing its own 16-byte vertices at a stride of ~100 bytes touched 64 lines per store
// instruction: 0.15 ms for the 3M vertices of a Voronoi tessellation).  The node ids are read where they are needed
// (no 32-entry private copy of the row); orientation as face_shape(): the first non-collinear vertex triple decides.
static constexpr int RAGGED_STAGE = 2048; // vertices one block stages (32 KiB: five blocks per CU; a tessellation's typical block holds 1536); fuller blocks write directly

__global__ void __launch_bounds__(256)
k_fill_ragged(const double *__restrict__ node_xy, const int32_t *__restrict__ faces_raw, int64_t n_face, int m,
              const int32_t *__restrict__ perm, const uint8_t *__restrict__ len, const int32_t *__restrict__ off,
              double *__restrict__ out_xy) {
    __shared__ double2 sh_xy[RAGGED_STAGE];
    const int64_t r0 = (int64_t)blockIdx.x * 256, r = r0 + threadIdx.x;
    const int64_t r1 = r0 + 256 < n_face ? r0 + 256 : n_face;
    const int base = off[r0], total = off[r1] - base;
    const bool staged = total <= RAGGED_STAGE;
    if (r < n_face) {
        const int64_t f = perm ? perm[r] : r;
        const int32_t *face = faces_raw + f * m;
        const int n = len[r];
        bool flip = false;
        for (int i = 0; i < n; i++) {
            const int ia = face[i >= 2 ? i - 2 : i + n - 2], ib = face[i >= 1 ? i - 1 : n - 1], ic = face[i];
            const P2 a = load_p2(node_xy, ia), b = load_p2(node_xy, ib), c = load_p2(node_xy, ic);
            const double ux = b.x - a.x, uy = b.y - a.y, vx = c.x - a.x, vy = c.y - a.y;
            const double prod = ux * vy - uy * vx;
            if (prod == 0) continue;
            flip = prod < 0;
            break;
        }
        double2 *dst = staged ? sh_xy + (off[r] - base) : reinterpret_cast<double2 *>(out_xy) + off[r];
        // eight corners at a time as two rounds of independent loads (ids, then coordinates) instead of a chain of dependent pairs
        for (int j0 = 0; j0 < n; j0 += 8) {
            int id[8];
            P2 p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) id[u] = face[j0 + u < n ? j0 + u : n - 1];
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = load_p2(node_xy, id[u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < n) dst[flip ? n - 1 - (j0 + u) : j0 + u] = make_double2(p[u].x, p[u].y);
        }
    }
    if (!staged) return; // (uniform)
    __syncthreads();
    double2 *out = reinterpret_cast<double2 *>(out_xy) + base;
    for (int k = threadIdx.x; k < total; k += 256) out[k] = sh_xy[k];
}

static void ragged_fill(xr_mesh *mesh, const int32_t *perm, const uint8_t *len, DevBuf<int32_t> &off, DevBuf<double> &xy) {
    const int64_t F = mesh->n_face;
    off.alloc((size_t)F + 1);
    DevBuf<int32_t> len32((size_t)std::max<int64_t>(F, 1));
    if (F > 0) XR_LAUNCH("widen_len", k_widen_len, dim3(div_up(F, 256)), dim3(256), 0, len, F, len32.get());
    exclusive_scan_i32(len32.get(), off.get(), F);
    const int64_t total = F > 0 ? read_scalar(off.get() + F) : 0;
    XR_REQUIRE(total >= 0, XR_ERR_LIMIT, "mesh has too many face vertices for int32 offsets");
    xy.alloc((size_t)std::max<int64_t>(total, 1) * 2);
    if (F > 0)
        XR_LAUNCH("fill_ragged", k_fill_ragged, dim3(div_up(F, 256)), dim3(256), 0, mesh->node_xy.get(),
                  mesh->faces_raw.get(), F, mesh->m, perm, len, off.get(), xy.get());
}

void mesh_face_coords(xr_mesh *mesh) {
    mesh_prepare(mesh, true);
    if (mesh->fxy_valid) return;
    ragged_fill(mesh, nullptr, mesh->len.get(), mesh->fxy_off, mesh->fxy); // (dense blocks are written by the prepare pass)
    mesh->fxy_valid = true;
}

static constexpr int64_t SAMPLE_MIN_FACES = 1 << 17; // smaller meshes: the full pass costs a launch either way
static constexpr int SAMPLE_STRIDE = 8;              // every 8th block of 256 faces

// pinned statistics page + (unless XR_STATS_TAIL=0) the hand-off words of the in-kernel reduction
static constexpr int64_t STATS_TAIL_MAX_BLOCKS = 1024;
static StatsTail stats_tail_for(xr_mesh *mesh, double *partials, int64_t nb) {
    if (!mesh->stats_host) {
        void *p = nullptr;
        XR_HIP(hipHostMalloc(&p, sizeof(double) * 16, hipHostMallocCoherent));
        mesh->stats_host = static_cast<double *>(p);
        memset(mesh->stats_host, 0, sizeof(double) * 16);
        XR_HIP(hipEventCreateWithFlags(&mesh->stats_event, hipEventDisableTiming | hipEventReleaseToSystem));
    }
    // The statistics leave through the kernel's last block for grids of up to STATS_TAIL_MAX_BLOCKS blocks (beyond that a separate
    // reduction kernel, as until round 3; "never" / "always" were measured behind a switch in round 4).  Every block pays ~3 us at its end for the hand-off (agent-scope stores, their
    // acknowledgement, a returning atomic): nothing for the 490 blocks of sample_stats, whose result the host is waiting
    // for (16 instead of 12 + 10 + 20 us until the tree's statistics are there), but +15 us for the 3900 blocks of
    // prepare_faces, whose statistics are only read after the index build -- those keep the one-block kernel on the side
    // stream (measured, 1M x 1M step: never 0.507, always 0.520, default see DESIGN section 5).
    constexpr int tail_mode = 2;
    mesh->stats_seq += 1.0;
    mesh->stats_polled = true; // (both forms end with the sequence word)
    // (the in-kernel form is for the engine's own main stream)
    if (tail_mode == 0 || (tail_mode == 2 && nb > STATS_TAIL_MAX_BLOCKS) || stream_override() || current_lane() || engine().on_side)
        return StatsTail{partials, nullptr, nullptr, nullptr, mesh->stats_seq};
    if (!mesh->stats_done.get()) {
        mesh->stats_done.alloc(16 * (1 + STATS_SHARDS));
        XR_HIP(hipMemsetAsync(mesh->stats_done.get(), 0, sizeof(unsigned) * 16 * (1 + STATS_SHARDS), launch_stream()));
    }
    return StatsTail{partials, mesh->stats_done.get(), mesh->stats.get(), mesh->stats_host, mesh->stats_seq};
}

void mesh_prepare(xr_mesh *mesh, bool want_fxy, bool stats_on_side, bool allow_sampled) {
    // two depths: statistics only (want_fxy = false: the mesh is used as a tree) or statistics + the caller-order
    // len / bbox / vertex blocks a query needs.  A mesh prepared light and later used as a query is prepared again;
    // so is one whose statistics come from a sample when exact ones are asked for.
    if (mesh->prepared && (mesh->has_attrs || !want_fxy) && (!mesh->stats_sampled || allow_sampled)) return;
    const int64_t F = mesh->n_face;
    const int m = mesh->m;
    const bool sampling_off = option(OPT_STATS_SAMPLE) == 0; // measurement switch
    // Polygon (ragged) meshes always keep len / bbox: the index build then reads one coalesced 32-byte box per face instead
    // of gathering the face's 6-20 vertices again in both of its passes (Voronoi tessellation of 1M triangles:
    // index_count 0.134 -> see DESIGN, index_scatter 0.067 ms).
    if (mesh->ragged()) want_fxy = true;
    if (!want_fxy && allow_sampled && !sampling_off && F >= SAMPLE_MIN_FACES && !mesh->has_attrs) {
        mesh->stats.alloc(8);
        const int64_t nb_all = (F + PREP_BLOCK - 1) / PREP_BLOCK;
        const int64_t nb = (nb_all + SAMPLE_STRIDE - 1) / SAMPLE_STRIDE;
        DevBuf<double> partials((size_t)nb * 8);
        dim3 grid((unsigned)nb), block(PREP_BLOCK);
        const StatsTail tail = stats_tail_for(mesh, partials.get(), nb);
        if (m == 3)
            XR_LAUNCH("sample_stats", k_sample_stats<3>, grid, block, 0, mesh->node_xy.get(), mesh->n_node, mesh->faces_raw.get(),
                      F, m, SAMPLE_STRIDE, partials.get(), tail);
        else if (m == 4)
            XR_LAUNCH("sample_stats", k_sample_stats<4>, grid, block, 0, mesh->node_xy.get(), mesh->n_node, mesh->faces_raw.get(),
                      F, m, SAMPLE_STRIDE, partials.get(), tail);
        else
            XR_LAUNCH("sample_stats", k_sample_stats<0>, grid, block, 0, mesh->node_xy.get(), mesh->n_node, mesh->faces_raw.get(),
                      F, m, SAMPLE_STRIDE, partials.get(), tail);
        if (!tail.done) {
            std::unique_ptr<SideScope> side;
            if (stats_on_side) side.reset(new SideScope);
            XR_LAUNCH("reduce_stats", k_reduce_stats, dim3(1), dim3(256), 0, partials.get(), nb, mesh->stats.get(),
                      mesh->stats_host, tail.seq);
            XR_HIP(hipEventRecord(mesh->stats_event, launch_stream()));
        }
        mesh->stats_event_pending = !tail.done;
        mesh->prepared = true;
        mesh->stats_valid = false;
        mesh->stats_sampled = true;
        return;
    }
    mesh->stats_sampled = false;
    const bool dense_fxy = want_fxy && !mesh->ragged(); // (ragged: flat blocks, filled on demand by mesh_face_coords)
    if (want_fxy) {
        if (dense_fxy) mesh->fxy.alloc((size_t)F * m * 2);
        mesh->len.alloc((size_t)F);
        mesh->bbox.alloc((size_t)F * 4);
    }
    mesh->fxy_valid = dense_fxy;
    mesh->has_attrs = want_fxy;
    mesh->stats.alloc(8);
    const int64_t nb = std::max<int64_t>(1, (F + PREP_BLOCK - 1) / PREP_BLOCK);
    DevBuf<double> partials((size_t)nb * 8);
    dim3 grid((unsigned)nb), block(PREP_BLOCK);
    const StatsTail tail = stats_tail_for(mesh, partials.get(), nb);
#define XR_PREP(MC)                                                                                                    \
    do {                                                                                                               \
        if (want_fxy)                                                                                                  \
            XR_LAUNCH("prepare_faces", (k_prepare_faces<MC, false>), grid, block, 0, mesh->node_xy.get(),              \
                      mesh->faces_raw.get(), F, m, dense_fxy ? mesh->fxy.get() : (double *)nullptr, mesh->len.get(),  \
                      mesh->bbox.get(), partials.get(), tail);                                                         \
        else                                                                                                           \
            XR_LAUNCH("prepare_stats", (k_prepare_faces<MC, true>), grid, block, 0, mesh->node_xy.get(),               \
                      mesh->faces_raw.get(), F, m, (double *)nullptr, (uint8_t *)nullptr, (double *)nullptr,           \
                      partials.get(), tail);                                                                           \
    } while (0)
    if (m == 3) XR_PREP(3);
    else if (m == 4) XR_PREP(4);
    else XR_PREP(0);
#undef XR_PREP
    if (!tail.done) {
        // (a one-block kernel that ends with writes to pinned host memory: ~10 us, which only the host waits for)
        std::unique_ptr<SideScope> side;
        if (stats_on_side) side.reset(new SideScope);
        XR_LAUNCH("reduce_stats", k_reduce_stats, dim3(1), dim3(256), 0, partials.get(), nb, mesh->stats.get(),
                  mesh->stats_host, tail.seq);
        XR_HIP(hipEventRecord(mesh->stats_event, launch_stream()));
    }
    mesh->stats_event_pending = !tail.done;
    mesh->prepared = true;
    mesh->stats_valid = false;
}

const double *mesh_area(xr_mesh *mesh) {
    if (mesh->area_valid) return mesh->area.get();
    const int64_t F = mesh->n_face;
    mesh->area.alloc((size_t)F);
    if (F > 0) {
        const dim3 grid(div_up(F, 256)), block(256);
        if (mesh->m == 3)
            XR_LAUNCH("face_area", k_face_area<3>, grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F, mesh->m,
                      mesh->area.get());
        else if (mesh->m == 4)
            XR_LAUNCH("face_area", k_face_area<4>, grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F, mesh->m,
                      mesh->area.get());
        else
            XR_LAUNCH("face_area", k_face_area<0>, grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F, mesh->m,
                      mesh->area.get());
    }
    mesh->area_valid = true;
    return mesh->area.get();
}

void mesh_read_stats(xr_mesh *mesh, bool need_exact) {
    if (!mesh->prepared) mesh_prepare(mesh, false);
    else if (need_exact && mesh->stats_sampled) mesh_prepare(mesh, mesh->has_attrs, false, false); // over all faces this time
    if (mesh->stats_valid) return;
    // the reducing block / kernel stores the sequence word behind the statistics: poll it (bounded), else wait for the
    // event behind the reduction kernel or drain the stream
    if (!(mesh->stats_polled && poll_pinned_f64(mesh->stats_host + 8, mesh->stats_seq))) {
        if (mesh->stats_event_pending) XR_HIP(hipEventSynchronize(mesh->stats_event));
        else XR_HIP(hipStreamSynchronize(engine().stream));
        XR_REQUIRE(!mesh->stats_polled || mesh->stats_host[8] == mesh->stats_seq, XR_ERR_HIP, "mesh statistics did not arrive");
    }
    for (int i = 0; i < 8; i++) mesh->h_stats[i] = mesh->stats_host[i];
    mesh->stats_valid = true;
}

// ---------------------------------------------------------------------------------------------
// spatial ordering = counting sort of the faces by a small integer key (the ONLY sort the engine
// needs), fused with the permutation of the per-face arrays:
//   pass 1 (k_spatial_count)   key of every face (Morton code of a coarse cell / grid cell of the
//                              tree index) + bucket histogram (global atomics)
//   scan                       bucket offsets (= cell_start for the tree index)
//   pass 2 (k_spatial_scatter) slot = offset + atomic cursor; the face's vertex block, length,
//                              bbox (f64, or the conservative f32 record bbox) are read
//                              coalesced in the caller's order and written to the slot
// Order inside a bucket follows the atomics, i.e. is unspecified -- every consumer is written
// so that final results do not depend on it.
// ---------------------------------------------------------------------------------------------

// (KeyTable / agg_insert -- one global atomic per DISTINCT key of a block -- live in xr_agg.h: the edge kernels use them too)

// (`lv` = the per-level grid sizes in LDS, [0..L) nx, [L..2L) ny, [2L..3L) base: the level differs from lane to lane, and
// indexing the kernel ARGUMENT g.nx[l] with it compiles to three dependent global loads per face)
__device__ __forceinline__ int face_cell(const GridParams &g, const int32_t *lv, double4 bb) {
    const double e = fmax(bb.y - bb.x, bb.w - bb.z);
    const int l = level_of_extent(g, e);
    const double inv_h = level_inv_h(g, l);
    const int nx = lv[l], ny = lv[MAX_LEVELS + l], base = lv[2 * MAX_LEVELS + l];
    const int cx = cell_coord(bb.x, g.x0, inv_h, nx);
    const int cy = cell_coord(bb.z, g.y0, inv_h, ny);
    return base + cy * nx + cx;
}

// (the bbox comes from the raw mesh, like in the scatter pass: node gathers hit the L2-resident node array)
template <bool INDEX, int MC>
__global__ void __launch_bounds__(256)
k_spatial_count(const double *__restrict__ node_xy, const int32_t *__restrict__ faces_raw, int64_t n, int m_rt,
                GridParams g, MortonParams mp, int32_t *__restrict__ key, int32_t *__restrict__ count,
                const double *__restrict__ bbox_opt = nullptr /* caller-order boxes of a prepared mesh (polygon meshes) */) {
    constexpr int MA = MC > 0 ? MC : XR_MAX_FACE_NODES;
    const int m = MC > 0 ? MC : m_rt;
    __shared__ int32_t sh_lv[3 * MAX_LEVELS];
    if (INDEX) {
        if (threadIdx.x < MAX_LEVELS) {
            sh_lv[threadIdx.x] = g.nx[threadIdx.x];
            sh_lv[MAX_LEVELS + threadIdx.x] = g.ny[threadIdx.x];
            sh_lv[2 * MAX_LEVELS + threadIdx.x] = g.base[threadIdx.x];
        }
        __syncthreads();
    }
    __shared__ KeyTable sh_tab;
    agg_clear(sh_tab);
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f < n) {
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        if (MC == 0 && bbox_opt) {
            const double4 b = reinterpret_cast<const double4 *>(bbox_opt)[f];
            xmin = b.x, xmax = b.y, ymin = b.z, ymax = b.w;
        } else {
            bool open = true;
#pragma unroll
            for (int j = 0; j < MA; j++) {
                if (j < m) {
                    const int v = faces_raw[f * m + j];
                    open = open && !(j >= 3 && v < 0); // polygon_length: stop at the first fill value
                    if (open) {
                        const P2 p = load_p2(node_xy, v);
                        xmin = fmin(xmin, p.x);
                        xmax = fmax(xmax, p.x);
                        ymin = fmin(ymin, p.y);
                        ymax = fmax(ymax, p.y);
                    }
                }
            }
        }
        const double4 bb = make_double4(xmin, xmax, ymin, ymax);
        const int k = INDEX ? face_cell(g, sh_lv, bb) : morton_key(mp, bb);
        key[f] = k;
        int rank;
        agg_insert(sh_tab, k, rank);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < AGG_SLOTS; s += 256)
        if (sh_tab.key[s] >= 0) atomicAdd(&count[sh_tab.key[s]], sh_tab.cnt[s]);
}

// The face's CCW-normalised vertex block and its bbox are recomputed from the raw mesh (node gathers hit the
// L2-resident node array) instead of being copied from the caller-order arrays: a mesh that is only ever the
// TREE never materialises `fxy`, and the scatter reads 17 bytes per face instead of 97.
template <bool INDEX, int MC>
__global__ void __launch_bounds__(256)
k_spatial_scatter(const int32_t *__restrict__ key, int64_t n, int m_rt, const int32_t *__restrict__ start,
                  int32_t *__restrict__ cursor, const double *__restrict__ node_xy,
                  const int32_t *__restrict__ faces_raw, int32_t *__restrict__ perm, double *__restrict__ o_fxy,
                  uint8_t *__restrict__ o_len, double *__restrict__ o_bbox, float *__restrict__ o_recbb, double x0,
                  double y0, const double *__restrict__ bbox_opt = nullptr, const uint8_t *__restrict__ len_opt = nullptr) {
    constexpr int MA = MC > 0 ? MC : XR_MAX_FACE_NODES;
    const int m = MC > 0 ? MC : m_rt;
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // Slots: `cursor` still holds the bucket histogram of the count pass and is handed out from the top down (the order inside
    // a bucket is unspecified anyway), which leaves every bucket at zero for the next build.  One returning atomic per DISTINCT
    // key of the block, as in the count pass (device-scope atomics are executed at the memory side of the fabric, ~40 G/s: one
    // per face was half of this kernel); a face's slot inside the block's stretch is its rank in the LDS table.
    __shared__ KeyTable sh_tab;
    agg_clear(sh_tab);
    const bool valid = f < n;
    const bool stored_only = MC == 0 && bbox_opt && len_opt && !o_fxy;
    // (the face's node ids go out together with the key, in front of the barriers: their gathers follow the slot computation
    // without a further round trip)
    int face[MA];
    if (valid && !stored_only) {
#pragma unroll
        for (int j = 0; j < MA; j++)
            if (j < m) face[j] = faces_raw[f * m + j];
    }
    const int k = valid ? key[f] : -1;
    __syncthreads();
    int tab_slot = 0, rank = 0;
    if (valid) tab_slot = agg_insert(sh_tab, k, rank);
    __syncthreads();
    for (int s = threadIdx.x; s < AGG_SLOTS; s += 256)
        if (sh_tab.key[s] >= 0) sh_tab.base[s] = start[sh_tab.key[s]] + atomicSub(&cursor[sh_tab.key[s]], sh_tab.cnt[s]) - sh_tab.cnt[s];
    __syncthreads();
    if (!valid) return;
    const int64_t r = (int64_t)sh_tab.base[tab_slot] + rank;
    if (stored_only) {
        // a prepared polygon mesh (its vertex blocks are written by ragged_fill): the record is the stored box and length
        const double4 b = reinterpret_cast<const double4 *>(bbox_opt)[f];
        perm[r] = (int32_t)f;
        o_len[r] = len_opt[f];
        if (INDEX)
            reinterpret_cast<float4 *>(o_recbb)[r] = make_float4(f32_below(b.x - x0), f32_above(b.y - x0), f32_below(b.z - y0), f32_above(b.w - y0));
        else
            reinterpret_cast<double4 *>(o_bbox)[r] = b;
        return;
    }
    int nl;
    bool flip;
    face_shape<MA>(node_xy, face, m, nl, flip);
    constexpr bool PREFETCH = MC > 0; // (general polygons, up to 32 nodes: gathered in the loop below, not held in registers)
    P2 pts[PREFETCH ? MA : 1];
    if (PREFETCH) {
#pragma unroll
        for (int j = 0; j < MA; j++)
            if (j < nl) pts[j] = load_p2(node_xy, face[j]);
    }
    perm[r] = (int32_t)f;
    o_len[r] = (uint8_t)nl;
    double2 *dst = reinterpret_cast<double2 *>(o_fxy) + r * m;
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
    for (int j = 0; j < MA; j++) {
        if (j < nl) {
            const P2 p = PREFETCH ? pts[PREFETCH ? j : 0] : load_p2(node_xy, face[j]);
            xmin = fmin(xmin, p.x);
            xmax = fmax(xmax, p.x);
            ymin = fmin(ymin, p.y);
            ymax = fmax(ymax, p.y);
            if (o_fxy) dst[flip ? nl - 1 - j : j] = make_double2(p.x, p.y); // (ragged meshes: ragged_fill writes them)
        }
    }
    if (INDEX) {
        reinterpret_cast<float4 *>(o_recbb)[r] =
            make_float4(f32_below(xmin - x0), f32_above(xmax - x0), f32_below(ymin - y0), f32_above(ymax - y0));
    } else {
        reinterpret_cast<double4 *>(o_bbox)[r] = make_double4(xmin, xmax, ymin, ymax);
    }
}

template <bool INDEX>
static void spatial_sort(xr_mesh *mesh, const GridParams &g, const MortonParams &mp, int64_t n_buckets,
                         int32_t *bucket_start, int32_t *perm, double *o_fxy, uint8_t *o_len, double *o_bbox,
                         float *o_recbb) {
    const int64_t F = mesh->n_face;
    // (polygon meshes are always prepared with their caller-order boxes and lengths, mesh_prepare)
    const double *boxes = mesh->has_attrs && mesh->m != 3 && mesh->m != 4 ? mesh->bbox.get() : nullptr;
    const uint8_t *lens = boxes ? mesh->len.get() : nullptr;
    // The histogram lives in the engine's zero-at-rest scratch: the count pass raises it, the scatter pass hands the slots
    // out from the top down -- every bucket is back at zero when the scatter has run, so the next build needs no memset
    // (one launch, or two when the length is odd: 9 us + gaps of a 0.56 ms step).
    DevBuf<int32_t> key((size_t)F), count_own;
    int32_t *count = zero_scratch(0, (size_t)n_buckets);
    const bool cached = count != nullptr;
    if (!cached) {
        count_own.alloc((size_t)n_buckets);
        count = count_own.get();
        XR_HIP(hipMemsetAsync(count, 0, sizeof(int32_t) * (((size_t)n_buckets + 3) & ~(size_t)3), launch_stream())); // (blocks are >= 4 KB: one fill kernel)
    }
    if (F > 0) {
        const char *name = INDEX ? "index_count" : "order_count";
        const dim3 grid(div_up(F, 256)), block(256);
        if (mesh->m == 3)
            XR_LAUNCH(name, (k_spatial_count<INDEX, 3>), grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F,
                      mesh->m, g, mp, key.get(), count);
        else if (mesh->m == 4)
            XR_LAUNCH(name, (k_spatial_count<INDEX, 4>), grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F,
                      mesh->m, g, mp, key.get(), count);
        else
            XR_LAUNCH(name, (k_spatial_count<INDEX, 0>), grid, block, 0, mesh->node_xy.get(), mesh->faces_raw.get(), F,
                      mesh->m, g, mp, key.get(), count, boxes);
    }
    exclusive_scan_i32(count, bucket_start, n_buckets);
    if (F > 0) {
        const char *name = INDEX ? "index_scatter" : "order_scatter";
        const dim3 grid(div_up(F, 256)), block(256);
        if (mesh->m == 3)
            XR_LAUNCH(name, (k_spatial_scatter<INDEX, 3>), grid, block, 0, key.get(), F, mesh->m, bucket_start, count,
                      mesh->node_xy.get(), mesh->faces_raw.get(), perm, o_fxy, o_len, o_bbox, o_recbb, g.x0, g.y0);
        else if (mesh->m == 4)
            XR_LAUNCH(name, (k_spatial_scatter<INDEX, 4>), grid, block, 0, key.get(), F, mesh->m, bucket_start, count,
                      mesh->node_xy.get(), mesh->faces_raw.get(), perm, o_fxy, o_len, o_bbox, o_recbb, g.x0, g.y0);
        else
            XR_LAUNCH(name, (k_spatial_scatter<INDEX, 0>), grid, block, 0, key.get(), F, mesh->m, bucket_start, count,
                      mesh->node_xy.get(), mesh->faces_raw.get(), perm, o_fxy, o_len, o_bbox, o_recbb, g.x0, g.y0, boxes, lens);
    }
    if (cached) zero_scratch_done(0);
}

void mesh_query_order(xr_mesh *mesh) {
    if (mesh->query_ready) return;
    mesh_read_stats(mesh, /*need_exact=*/true); // (the ordering heuristic reads sums over ALL faces: sampled statistics are redone)
    const int64_t F = mesh->n_face;
    const int m = mesh->m;
    // coherent numbering (consecutive faces are, on average, within a few face extents of each
    // other -- typical for mesh generators, not for qhull output): keep the caller's order
    const bool force_sort = option(OPT_FORCE_QUERY_SORT) != 0;
    const double mean_ext = F > 0 ? mesh->h_stats[4] / (double)F : 0.0;
    const double mean_jump = F > 0 ? mesh->h_stats[7] / ((double)F * 63.0 / 64.0) : 0.0;
    mesh->query_identity = !force_sort && (F == 0 || mean_jump <= 4.0 * mean_ext);
    if (mesh->query_identity) {
        mesh_face_coords(mesh); // the caller-order vertex blocks ARE the query-order ones (no-op if prepared with them)
        mesh->query_ready = true;
        return;
    }
    mesh->q_perm.alloc((size_t)F);
    if (!mesh->ragged()) mesh->q_fxy.alloc((size_t)F * m * 2);
    mesh->q_len.alloc((size_t)F);
    mesh->q_bbox.alloc((size_t)F * 4);
    if (F > 0) {
        const double xmin = mesh->h_stats[0], xmax = mesh->h_stats[1], ymin = mesh->h_stats[2], ymax = mesh->h_stats[3];
        double span = std::max(xmax - xmin, ymax - ymin);
        if (!(span > 0)) span = 1.0;
        double h = 2.0 * mesh->h_stats[4] / (double)F; // two mean extents per coarse cell
        if (!(h > 0)) h = span;
        int bits = 1;
        while (bits < 10 && ldexp(h, bits) < span) bits++;
        MortonParams mp{xmin, ymin, (double)(1 << bits) / (span * (1.0 + 1e-9)), 1 << bits};
        GridParams g{};
        DevBuf<int32_t> start(((size_t)1 << (2 * bits)) + 1);
        spatial_sort<false>(mesh, g, mp, (int64_t)1 << (2 * bits), start.get(), mesh->q_perm.get(),
                            mesh->ragged() ? (double *)nullptr : mesh->q_fxy.get(), mesh->q_len.get(), mesh->q_bbox.get(),
                            nullptr);
    }
    if (mesh->ragged()) ragged_fill(mesh, mesh->q_perm.get(), mesh->q_len.get(), mesh->q_off, mesh->q_fxy);
    mesh->query_ready = true;
}

void mesh_build_index(xr_mesh *mesh) {
    if (mesh->indexed) return;
    mesh_read_stats(mesh);
    const int64_t F = mesh->n_face;
    const int m = mesh->m;
    XR_REQUIRE(F < (int64_t)1 << 31, XR_ERR_LIMIT, "mesh has too many faces for int32 indices");
    const double xmin = mesh->h_stats[0], xmax = mesh->h_stats[1], ymin = mesh->h_stats[2], ymax = mesh->h_stats[3];
    // (sampled statistics: [7] = faces looked at; the largest extent is then only bounded by the domain)
    const bool sampled = mesh->stats_sampled;
    const double n_ext = sampled ? std::max(mesh->h_stats[7], 1.0) : (double)F;
    const double sum_ext = mesh->h_stats[4] * ((double)F / n_ext);
    const double max_ext = sampled ? std::max(xmax - xmin, ymax - ymin) : mesh->h_stats[5];
    GridParams g{};
    double W = F > 0 ? xmax - xmin : 1.0, H = F > 0 ? ymax - ymin : 1.0;
    if (!(W > 0)) W = 1.0;
    if (!(H > 0)) H = 1.0;
    // level-0 cell size = 1.4 mean bbox extents (measured on the 1M benchmark.  Round 1's search kernel: 0.75 -> 0.168 ms,
    // 1.0 -> 0.155, 1.25 -> 0.137, 1.5 -> 0.134, 2.0 -> 0.144; round 5's, interleaved on one box: 1.25 -> 0.0924,
    // 1.4 -> 0.0898, 1.5 -> 0.0976, 1.6 -> 0.111 -- at 1.5 the second level still holds more than 1/64 of the records, so every
    // face walks it, yet it is too empty to pay for its cell bounds; XR_H0_FACTOR is the A/B switch)
    constexpr double h0_factor = 1.4; // (measured optimum, round 5: 1.25 -> 0.0924, 1.4 -> 0.0898, 1.5 -> 0.0976 ms of search)
    double h0 = F > 0 ? h0_factor * sum_ext / (double)F : 1.0;
    if (!(h0 > 0)) h0 = std::max(W, H);
    // bound the level-0 cell count by ~4 cells per face
    const double max_cells0 = std::max(4.0 * (double)F, 1024.0);
    while ((floor(W / h0) + 1.0) * (floor(H / h0) + 1.0) > max_cells0) h0 *= 2.0;
    g.x0 = F > 0 ? xmin : 0.0;
    g.y0 = F > 0 ? ymin : 0.0;
    g.h0 = h0;
    g.inv_h0 = 1.0 / h0;
    int L = 1;
    while (L < MAX_LEVELS && !(max_ext <= 0.999 * ldexp(h0, (L - 1) * LEVEL_SHIFT))) L++;
    g.n_levels = L;
    int64_t total = 0;
    for (int l = 0; l < L; l++) {
        const double inv_h = ldexp(g.inv_h0, -l * LEVEL_SHIFT);
        const int64_t nx = (int64_t)floor(W * inv_h) + 1, ny = (int64_t)floor(H * inv_h) + 1;
        XR_REQUIRE(total + nx * ny < ((int64_t)1 << 31), XR_ERR_LIMIT, "spatial index too large");
        g.base[l] = (int)total;
        g.nx[l] = (int)nx;
        g.ny[l] = (int)ny;
        total += nx * ny;
    }
    g.n_cells = (int)total;
    mesh->grid = g;

    mesh->cell_start.alloc((size_t)total + 1);
    mesh->rec_bb.alloc((size_t)F * 4 + 4 * 8); // (+ 8 records of padding: k_search reads up to WALK_LOADS - 1 records past a run, masked)
    mesh->rec_face.alloc((size_t)F);
    if (!mesh->ragged()) mesh->rec_fxy.alloc((size_t)F * m * 2);
    mesh->rec_len.alloc((size_t)F);
    MortonParams mp{};
    spatial_sort<true>(mesh, g, mp, total, mesh->cell_start.get(), mesh->rec_face.get(),
                       mesh->ragged() ? (double *)nullptr : mesh->rec_fxy.get(), mesh->rec_len.get(), nullptr,
                       mesh->rec_bb.get());
    if (mesh->ragged()) ragged_fill(mesh, mesh->rec_face.get(), mesh->rec_len.get(), mesh->rec_off, mesh->rec_fxy);
    mesh->indexed = true;
}

// ---------------------------------------------------------------------------------------------
// centroids -- connectivity.centroids (connectivity.py:636-664) on the caller's vertex order
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_centroids(const double *__restrict__ node_xy,
                                                  const int32_t *__restrict__ faces_raw, int64_t n_face, int m,
                                                  double *__restrict__ cxy) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    const int32_t *face = faces_raw + f * m;
    if (m == 3) {
        double sx = 0.0, sy = 0.0;
        for (int i = 0; i < 3; i++) {
            const P2 p = load_p2(node_xy, face[i]);
            sx += p.x;
            sy += p.y;
        }
        cxy[2 * f] = sx / 3.0;
        cxy[2 * f + 1] = sy / 3.0;
        return;
    }
    const int f0 = face[0];
    const P2 p0 = load_p2(node_xy, f0);
    double det = 0.0, sx = 0.0, sy = 0.0;
    for (int i = 0; i < m; i++) {
        const int ia = face[i] < 0 ? f0 : face[i];
        int ib = f0;
        if (i + 1 < m) ib = face[i + 1] < 0 ? f0 : face[i + 1];
        const P2 a = load_p2(node_xy, ia), b = load_p2(node_xy, ib);
        const double ax = a.x - p0.x, ay = a.y - p0.y, bx = b.x - p0.x, by = b.y - p0.y;
        const double d = ax * by - ay * bx;
        det += d;
        sx += (ax + bx) * d;
        sy += (ay + by) * d;
    }
    const double aw = 1.0 / (3.0 * det);
    cxy[2 * f] = aw * sx + p0.x;
    cxy[2 * f + 1] = aw * sy + p0.y;
}

// device-resident variants used by other translation units (no host copy)
void mesh_centroids_dev(xr_mesh *mesh, double *cxy_dev) {
    if (mesh->n_face > 0)
        XR_LAUNCH("centroids", k_centroids, dim3(div_up(mesh->n_face, 256)), dim3(256), 0, mesh->node_xy.get(),
                  mesh->faces_raw.get(), mesh->n_face, mesh->m, cxy_dev);
}

std::shared_ptr<DevBuf<double>> mesh_centroids_shared(xr_mesh *mesh) {
    hipStream_t st = launch_stream();
    if (!mesh->centroids_dev) {
        auto c = std::make_shared<DevBuf<double>>((size_t)std::max<int64_t>(mesh->n_face, 1) * 2);
        mesh_centroids_dev(mesh, c->get());
        if (!mesh->centroids_event) XR_HIP(hipEventCreateWithFlags(&mesh->centroids_event, hipEventDisableTiming));
        XR_HIP(hipEventRecord(mesh->centroids_event, st));
        mesh->centroids_stream = st;
        mesh->centroids_dev = std::move(c);
    } else if (st != mesh->centroids_stream) {
        // (filled on another stream -- a points handle launches on the side stream --: this stream's work comes behind it)
        XR_HIP(hipStreamWaitEvent(st, mesh->centroids_event, 0));
    }
    return mesh->centroids_dev;
}

void mesh_faces_ccw_dev(xr_mesh *mesh, int64_t *faces_dev, bool caller_order) {
    if (mesh->n_face > 0)
        XR_LAUNCH("faces_ccw", k_faces_ccw, dim3(div_up(mesh->n_face, 256)), dim3(256), 0, mesh->node_xy.get(),
                  mesh->faces_raw.get(), mesh->n_face, mesh->m, faces_dev, caller_order);
}

// Ugrid2d.from_structured_bounds -> _from_intervals_helper (xugrid/ugrid/ugrid2d.py:1973-2034, :1894-1912) on the
// device: the (ny + 1) x (nx + 1) vertices of a rectilinear grid (node id = j * (nx + 1) + i, the meshgrid order) and
// one quad per cell (face id = j * nx + i), corners lower-left, lower-right, upper-right, upper-left with "left" and
// "lower" following the direction of the vertex arrays (a descending axis swaps them).
__global__ void __launch_bounds__(256)
k_rect_nodes(const double *__restrict__ xv, const double *__restrict__ yv, int64_t nx1, int64_t n_node,
             double *__restrict__ node_xy) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_node) return;
    const int64_t j = v / nx1, i = v - j * nx1;
    reinterpret_cast<double2 *>(node_xy)[v] = make_double2(xv[i], yv[j]);
}

__global__ void __launch_bounds__(256)
k_rect_faces(int64_t nx, int64_t n_face, bool flip_x, bool flip_y, int32_t *__restrict__ faces) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    const int64_t j = f / nx, i = f - j * nx;
    const int64_t left = flip_x ? i + 1 : i, right = flip_x ? i : i + 1;
    const int64_t lower = flip_y ? j + 1 : j, upper = flip_y ? j : j + 1;
    const int64_t nx1 = nx + 1;
    reinterpret_cast<int4 *>(faces)[f] = make_int4((int)(lower * nx1 + left), (int)(lower * nx1 + right),
                                                   (int)(upper * nx1 + right), (int)(upper * nx1 + left));
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_mesh_create_rectilinear(const double *x_vertices, int64_t nx, const double *y_vertices, int64_t ny,
                               xr_mesh **out) {
    XR_API_BEGIN
    XR_REQUIRE(out && x_vertices && y_vertices, XR_ERR_INVALID, "xr_mesh_create_rectilinear: NULL argument");
    XR_REQUIRE(nx >= 1 && ny >= 1, XR_ERR_INVALID, "xr_mesh_create_rectilinear: at least one cell per axis expected");
    const int64_t n_node = (nx + 1) * (ny + 1), n_face = nx * ny;
    XR_REQUIRE((nx + 1) < ((int64_t)1 << 31) / (ny + 1) && n_face * 4 < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_mesh_create_rectilinear: mesh exceeds the int32 index range");
    for (int64_t i = 0; i <= nx; i++)
        XR_REQUIRE(x_vertices[i] == x_vertices[i], XR_ERR_INVALID, "xr_mesh_create_rectilinear: NaN x vertex %lld", (long long)i);
    for (int64_t j = 0; j <= ny; j++)
        XR_REQUIRE(y_vertices[j] == y_vertices[j], XR_ERR_INVALID, "xr_mesh_create_rectilinear: NaN y vertex %lld", (long long)j);
    engine();
    xr_mesh *mesh = new xr_mesh();
    try {
        mesh->n_node = n_node;
        mesh->n_face = n_face;
        mesh->m = 4;
        mesh->node_xy.alloc((size_t)n_node * 2);
        mesh->faces_raw.alloc((size_t)n_face * 4);
        DevBuf<double> xv((size_t)nx + 1), yv((size_t)ny + 1);
        h2d(xv.get(), x_vertices, sizeof(double) * (size_t)(nx + 1));
        h2d(yv.get(), y_vertices, sizeof(double) * (size_t)(ny + 1));
        XR_LAUNCH("rect_nodes", k_rect_nodes, dim3(div_up(n_node, 256)), dim3(256), 0, xv.get(), yv.get(), nx + 1, n_node,
                  mesh->node_xy.get());
        // (ugrid2d.py:1904-1909: the tests read the flattened vertex arrays; see _from_intervals_helper in ugrid2d.py)
        XR_LAUNCH("rect_faces", k_rect_faces, dim3(div_up(n_face, 256)), dim3(256), 0, nx, n_face,
                  x_vertices[1] < x_vertices[0], y_vertices[1] < y_vertices[0], mesh->faces_raw.get());
        stream_sync();
    } catch (...) {
        delete mesh;
        throw;
    }
    *out = mesh;
    XR_API_END
}

int xr_mesh_create(const double *node_xy, int64_t n_node, const void *faces, int faces_itemsize, int64_t n_face,
                   int64_t n_max_node, int64_t fill_value, xr_mesh **out) {
    XR_API_BEGIN
    XR_REQUIRE(out != nullptr, XR_ERR_INVALID, "xr_mesh_create: out is NULL");
    XR_REQUIRE(n_node >= 0 && n_face >= 0, XR_ERR_INVALID, "xr_mesh_create: negative sizes");
    XR_REQUIRE(n_max_node >= 3 && n_max_node <= XR_MAX_FACE_NODES, XR_ERR_LIMIT,
               "xr_mesh_create: n_max_node_per_face must be in [3, %d], got %lld", XR_MAX_FACE_NODES,
               (long long)n_max_node);
    XR_REQUIRE(faces_itemsize == 4 || faces_itemsize == 8, XR_ERR_INVALID, "xr_mesh_create: faces_itemsize must be 4 or 8");
    XR_REQUIRE(n_node < ((int64_t)1 << 31) && n_face * n_max_node < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_mesh_create: mesh exceeds the int32 index range");
    XR_REQUIRE((node_xy && faces) || n_face == 0, XR_ERR_INVALID, "xr_mesh_create: NULL arrays");
    engine();
    // Ingest on the way to the device.  Both arrays go through the pinned staging buffers in pieces, filled by the host
    // thread pool while the previous piece is in flight (pageable numpy arrays handed to hipMemcpy move at a third of the
    // link rate).  The connectivity is NARROWED while it is copied: fill value -> -1, int64 -> int32 (half the bytes over
    // PCIe), validated (every face has at least three nodes, every node id is inside [0, n_node)); the first offending
    // face is reported.  Nothing waits for the last DMA: the arrays are consumed, later work is stream-ordered behind it.
    // (XR_INGEST=device: raw upload + k_ingest_faces on the device, as until round 3 -- measurement switch.)
    const size_t cnt = (size_t)n_face * (size_t)n_max_node;
    // (Meshes of less than 1 MB take the device-side ingest: two plain copies and a kernel, no pinned staging buffers --
    // those are 2 x 64 MiB of pinned host memory, allocated on first use.)
    const bool device_ingest_env = option(OPT_INGEST_DEVICE) != 0;
    const bool device_ingest =
        device_ingest_env || cnt * (size_t)faces_itemsize + sizeof(double) * 2 * (size_t)n_node < ((size_t)1 << 20);
    xr_mesh *mesh = new xr_mesh();
    try {
        mesh->n_node = n_node;
        mesh->n_face = n_face;
        mesh->m = (int)n_max_node;
        mesh->node_xy.alloc((size_t)n_node * 2);
        mesh->faces_raw.alloc(cnt);
        if (device_ingest) {
            h2d_big(mesh->node_xy.get(), node_xy, sizeof(double) * 2 * (size_t)n_node);
        } else {
            const char *xy_bytes = reinterpret_cast<const char *>(node_xy);
            h2d_staged(mesh->node_xy.get(), sizeof(double) * 2 * (size_t)n_node, [=](char *dst, size_t off, size_t n) {
                parallel_ranges(n, 64, [=](size_t b, size_t e) { memcpy(dst + b, xy_bytes + off + b, e - b); });
            });
        }
        if (cnt > 0 && !device_ingest) {
            std::atomic<int64_t> bad_short(INT64_MAX), bad_node(INT64_MAX);
            const int m = (int)n_max_node;
            auto narrow = [&](auto *raw) {
                h2d_staged(mesh->faces_raw.get(), cnt * sizeof(int32_t), [&, raw](char *dst, size_t off, size_t n) {
                    int32_t *out32 = reinterpret_cast<int32_t *>(dst);
                    const size_t k0 = off / sizeof(int32_t), nk = n / sizeof(int32_t);
                    parallel_ranges(nk, 16, [&, raw, out32, k0](size_t b, size_t e) {
                        int64_t first_short = INT64_MAX, first_node = INT64_MAX;
                        for (size_t i = b; i < e; i++) {
                            const size_t k = k0 + i;
                            const int64_t v = (int64_t)raw[k];
                            if (v == fill_value || v == -1) {
                                if ((int)(k % (size_t)m) < 3 && first_short == INT64_MAX) first_short = (int64_t)(k / (size_t)m);
                                out32[i] = -1;
                            } else {
                                if ((v < 0 || v >= n_node) && first_node == INT64_MAX) first_node = (int64_t)(k / (size_t)m);
                                out32[i] = (int32_t)v;
                            }
                        }
                        int64_t cur = bad_short.load();
                        while (first_short < cur && !bad_short.compare_exchange_weak(cur, first_short)) {}
                        cur = bad_node.load();
                        while (first_node < cur && !bad_node.compare_exchange_weak(cur, first_node)) {}
                    });
                });
            };
            if (faces_itemsize == 8) narrow(static_cast<const int64_t *>(faces));
            else narrow(static_cast<const int32_t *>(faces));
            if (bad_short.load() != INT64_MAX || bad_node.load() != INT64_MAX) stream_sync(); // (the handle is dropped below)
            XR_REQUIRE(bad_short.load() == INT64_MAX, XR_ERR_INVALID, "xr_mesh_create: face %lld has fewer than 3 nodes",
                       (long long)bad_short.load());
            XR_REQUIRE(bad_node.load() == INT64_MAX, XR_ERR_INVALID,
                       "xr_mesh_create: face %lld references a node outside [0,%lld)", (long long)bad_node.load(),
                       (long long)n_node);
        } else if (cnt > 0) {
            DevBuf<char> raw(cnt * (size_t)faces_itemsize);
            DevBuf<int64_t> err(2); // [0] first face with fewer than 3 nodes, [1] first face with a node id out of range
            const int64_t none[2] = {INT64_MAX, INT64_MAX};
            h2d(err.get(), none, sizeof(none));
            h2d_big(raw.get(), faces, cnt * (size_t)faces_itemsize);
            if (faces_itemsize == 8)
                XR_LAUNCH("ingest_faces", k_ingest_faces<int64_t>, dim3(div_up((int64_t)cnt, 256)), dim3(256), 0,
                          reinterpret_cast<const int64_t *>(raw.get()), (int64_t)cnt, (int)n_max_node, fill_value, n_node,
                          mesh->faces_raw.get(), err.get());
            else
                XR_LAUNCH("ingest_faces", k_ingest_faces<int32_t>, dim3(div_up((int64_t)cnt, 256)), dim3(256), 0,
                          reinterpret_cast<const int32_t *>(raw.get()), (int64_t)cnt, (int)n_max_node, fill_value, n_node,
                          mesh->faces_raw.get(), err.get());
            int64_t h_err[2];
            d2h(h_err, err.get(), sizeof(h_err));
            XR_REQUIRE(h_err[0] == INT64_MAX, XR_ERR_INVALID, "xr_mesh_create: face %lld has fewer than 3 nodes",
                       (long long)h_err[0]);
            XR_REQUIRE(h_err[1] == INT64_MAX, XR_ERR_INVALID,
                       "xr_mesh_create: face %lld references a node outside [0,%lld)", (long long)h_err[1],
                       (long long)n_node);
        }
    } catch (...) {
        delete mesh;
        throw;
    }
    *out = mesh;
    XR_API_END
}

int xr_mesh_create_dev(const double *node_xy_dev, int64_t n_node, const void *faces_dev, int faces_itemsize, int64_t n_face,
                       int64_t n_max_node, int64_t fill_value, xr_mesh **out) {
    XR_API_BEGIN
    XR_REQUIRE(out != nullptr, XR_ERR_INVALID, "xr_mesh_create_dev: out is NULL");
    XR_REQUIRE(n_node >= 0 && n_face >= 0, XR_ERR_INVALID, "xr_mesh_create_dev: negative sizes");
    XR_REQUIRE(n_max_node >= 3 && n_max_node <= XR_MAX_FACE_NODES, XR_ERR_LIMIT,
               "xr_mesh_create_dev: n_max_node_per_face must be in [3, %d], got %lld", XR_MAX_FACE_NODES,
               (long long)n_max_node);
    XR_REQUIRE(faces_itemsize == 4 || faces_itemsize == 8, XR_ERR_INVALID, "xr_mesh_create_dev: faces_itemsize must be 4 or 8");
    XR_REQUIRE(n_node < ((int64_t)1 << 31) && n_face * n_max_node < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_mesh_create_dev: mesh exceeds the int32 index range");
    XR_REQUIRE((node_xy_dev && faces_dev) || n_face == 0, XR_ERR_INVALID, "xr_mesh_create_dev: NULL arrays");
    engine();
    const size_t cnt = (size_t)n_face * (size_t)n_max_node;
    xr_mesh *mesh = new xr_mesh();
    try {
        mesh->n_node = n_node;
        mesh->n_face = n_face;
        mesh->m = (int)n_max_node;
        mesh->node_xy.alloc((size_t)n_node * 2);
        mesh->faces_raw.alloc(cnt);
        if (n_node > 0)
            XR_HIP(hipMemcpyAsync(mesh->node_xy.get(), node_xy_dev, sizeof(double) * 2 * (size_t)n_node, hipMemcpyDeviceToDevice,
                                  launch_stream()));
        if (cnt > 0) {
            // [0] first face with fewer than 3 nodes, [1] first face with a node id out of range
            DevBuf<int64_t> err(2);
            XR_HIP(hipMemsetAsync(err.get(), 0xff, 2 * sizeof(int64_t), launch_stream())); // (all ones: "none" for the unsigned atomicMin)
            if (faces_itemsize == 8)
                XR_LAUNCH("ingest_faces", k_ingest_faces<int64_t>, dim3(div_up((int64_t)cnt, 256)), dim3(256), 0,
                          reinterpret_cast<const int64_t *>(faces_dev), (int64_t)cnt, (int)n_max_node, fill_value, n_node,
                          mesh->faces_raw.get(), err.get());
            else
                XR_LAUNCH("ingest_faces", k_ingest_faces<int32_t>, dim3(div_up((int64_t)cnt, 256)), dim3(256), 0,
                          reinterpret_cast<const int32_t *>(faces_dev), (int64_t)cnt, (int)n_max_node, fill_value, n_node,
                          mesh->faces_raw.get(), err.get());
            uint64_t h_err[2];
            d2h(h_err, err.get(), sizeof(h_err));
            XR_REQUIRE(h_err[0] == UINT64_MAX, XR_ERR_INVALID, "xr_mesh_create_dev: face %lld has fewer than 3 nodes",
                       (long long)h_err[0]);
            XR_REQUIRE(h_err[1] == UINT64_MAX, XR_ERR_INVALID,
                       "xr_mesh_create_dev: face %lld references a node outside [0,%lld)", (long long)h_err[1],
                       (long long)n_node);
        } else {
            stream_sync();
        }
    } catch (...) {
        delete mesh;
        throw;
    }
    *out = mesh;
    XR_API_END
}

int xr_mesh_destroy(xr_mesh *mesh) {
    XR_API_BEGIN
    if (mesh) {
        flush_pending_points_of(mesh); // (an xr_points handle made from this mesh whose kernels have not been launched yet)
        release_point();
        delete mesh;
    }
    XR_API_END
}

int xr_mesh_info(const xr_mesh *mesh, int64_t *n_node, int64_t *n_face, int64_t *n_max_node) {
    XR_API_BEGIN
    XR_REQUIRE(mesh, XR_ERR_INVALID, "xr_mesh_info: NULL mesh");
    if (n_node) *n_node = mesh->n_node;
    if (n_face) *n_face = mesh->n_face;
    if (n_max_node) *n_max_node = mesh->m;
    XR_API_END
}

int xr_mesh_device_bytes(const xr_mesh *mesh, int64_t *bytes) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && bytes, XR_ERR_INVALID, "xr_mesh_device_bytes: NULL argument");
    size_t b = (mesh->centroids_dev ? mesh->centroids_dev->bytes() : 0) + mesh->node_xy.bytes() + mesh->faces_raw.bytes() + mesh->fxy.bytes() + mesh->fxy_off.bytes() +
               mesh->len.bytes() + mesh->bbox.bytes() + mesh->area.bytes() + mesh->stats.bytes() + mesh->q_perm.bytes() +
               mesh->q_fxy.bytes() + mesh->q_off.bytes() + mesh->q_len.bytes() + mesh->q_bbox.bytes() +
               mesh->cell_start.bytes() + mesh->rec_bb.bytes() + mesh->rec_face.bytes() + mesh->rec_fxy.bytes() +
               mesh->rec_off.bytes() + mesh->rec_len.bytes();
    *bytes = (int64_t)b;
    XR_API_END
}

int xr_mesh_prepare(xr_mesh *mesh) {
    XR_API_BEGIN
    XR_REQUIRE(mesh, XR_ERR_INVALID, "xr_mesh_prepare: NULL mesh");
    mesh_prepare(mesh);
    stream_sync();
    XR_API_END
}

int xr_mesh_build_index(xr_mesh *mesh) {
    XR_API_BEGIN
    XR_REQUIRE(mesh, XR_ERR_INVALID, "xr_mesh_build_index: NULL mesh");
    mesh_build_index(mesh);
    stream_sync();
    XR_API_END
}

int xr_mesh_invalidate(xr_mesh *mesh) {
    XR_API_BEGIN
    XR_REQUIRE(mesh, XR_ERR_INVALID, "xr_mesh_invalidate: NULL mesh");
    // (the blocks go back to the pool, which hands them out again in stream order on the one main stream: in asynchronous
    // mode nothing has to wait here)
    flush_pending_points_of(mesh); // (an xr_points handle made from this mesh whose kernels have not been launched yet)
    release_point();
    mesh->prepared = false;
    mesh->has_attrs = false;
    mesh->area_valid = false;
    mesh->fxy_valid = false;
    mesh->query_ready = false;
    mesh->query_identity = false;
    mesh->indexed = false;
    mesh->stats_valid = false;
    mesh->fxy.release(); mesh->len.release(); mesh->bbox.release(); mesh->area.release(); mesh->stats.release();
    mesh->centroids_dev.reset(); // (a handle that shares them keeps them)
    mesh->q_perm.release(); mesh->q_fxy.release(); mesh->q_len.release(); mesh->q_bbox.release(); mesh->q_off.release();
    mesh->fxy_off.release(); mesh->rec_off.release();
    mesh->cell_start.release(); mesh->rec_bb.release(); mesh->rec_face.release(); mesh->rec_fxy.release();
    mesh->rec_len.release();
    XR_API_END
}

int xr_mesh_area(xr_mesh *mesh, double *area_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && area_out, XR_ERR_INVALID, "xr_mesh_area: NULL argument");
    const double *area = mesh_area(mesh);
    if (mesh->n_face > 0) {
        d2h(area_out, area, sizeof(double) * (size_t)mesh->n_face);
    }
    stream_sync();
    XR_API_END
}

int xr_mesh_centroids(xr_mesh *mesh, double *centroids_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && centroids_out, XR_ERR_INVALID, "xr_mesh_centroids: NULL argument");
    const int64_t F = mesh->n_face;
    if (F > 0) {
        const auto c = mesh_centroids_shared(mesh);
        d2h(centroids_out, c->get(), sizeof(double) * 2 * (size_t)F);
        stream_sync();
    }
    XR_API_END
}

int xr_mesh_faces(xr_mesh *mesh, int64_t *faces_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && faces_out, XR_ERR_INVALID, "xr_mesh_faces: NULL argument");
    const int64_t n = mesh->n_face * mesh->m;
    if (n > 0) {
        DevBuf<int64_t> wide((size_t)n);
        XR_LAUNCH("faces_ccw", k_faces_ccw, dim3(div_up(mesh->n_face, 256)), dim3(256), 0, mesh->node_xy.get(),
                  mesh->faces_raw.get(), mesh->n_face, mesh->m, wide.get(), false);
        d2h(faces_out, wide.get(), sizeof(int64_t) * (size_t)n);
        stream_sync();
    }
    XR_API_END
}

int xr_mesh_download(xr_mesh *mesh, double *node_xy_out, int64_t *faces_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh, XR_ERR_INVALID, "xr_mesh_download: NULL handle");
    if (node_xy_out && mesh->n_node > 0) d2h(node_xy_out, mesh->node_xy.get(), sizeof(double) * 2 * (size_t)mesh->n_node);
    const int64_t n = mesh->n_face * mesh->m;
    if (faces_out && n > 0) {
        std::vector<int32_t> raw((size_t)n);
        d2h(raw.data(), mesh->faces_raw.get(), sizeof(int32_t) * (size_t)n);
        for (int64_t i = 0; i < n; i++) faces_out[i] = raw[(size_t)i];
    }
    XR_API_END
}

} // extern "C"
