// xr_edges.hip -- line segments against the faces of a mesh: the weights of NetworkGridder.
//
// Replaces, in one device pass from edge coordinates to CSR,
//   numba_celltree.CellTree2d.intersect_edges(edge_coords) -> (edge_index, face_index, intersections[n, 2, 2])
// as called from UnstructuredGrid2d.intersection_length (xugrid/regrid/unstructured.py:203-215), the
//   length = norm(diff(intersections))                                                  (:212)
// that follows it, its argsort by face (:211) and MatrixCSR.from_triplet (xugrid/regrid/gridder.py:66-73).
//
// Per edge the hierarchical grid of the mesh is walked over the edge's bounding box (long edges: one wave per
// edge, walking the cells along the segment's major axis); every candidate face is clipped with the Cyrus-Beck parametric
// line clip against its CCW-normalised (convex) polygon.  A pair is kept iff the clipped parameter interval has
// t0 < t1 -- touching a corner or an edge from outside yields no entry (tests/test_regrid/test_network_gridder.py:
// nnz == 8 for the four-edge network on the 4 x 4 grid).  The arithmetic mirrors oracle/xr_oracle.c
// (cyrus_beck_clip) operation for operation; rows come out ordered by edge id.
#include <cstring>

#include "xr_objects.h"
#include "xr_agg.h"

namespace xr {

static constexpr int EDGE_WALK_PAD = 8;    // records of padding behind rec_bb (xr_mesh.hip allocates them; WALK_PAD of the face search)
static constexpr int EDGE_BIG_CELLS = 64;  // edges whose box covers more grid cells (all levels) get a wave of their own
static constexpr int ROW_SORT_SMALL = 16;  // rows up to this length are insertion-sorted by one thread (k_edge_rows_sort)
static constexpr int ROW_SORT_LDS = 4096;  // rows up to this length are sorted in LDS by one block

// a + t (b - a), t in [0, 1], against every half-plane of the CCW polygon.  -> length of the clipped piece, or -1;
// its end points in c / d when asked for
__device__ __forceinline__ double cyrus_beck_length(const double *__restrict__ poly, int n, P2 a, P2 b,
                                                    P2 *c = nullptr, P2 *d = nullptr) {
    const double sx = b.x - a.x, sy = b.y - a.y;
    double t0 = 0.0, t1 = 1.0;
    P2 v0 = load_p2(poly, 0);
    for (int i = 0; i < n; i++) {
        const P2 v1 = load_p2(poly, (i + 1 < n) ? i + 1 : 0);
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        if (wx != 0.0 || wy != 0.0) {
            const double nx = -wy, ny = wx; // inward normal of a CCW polygon
            const double den = nx * sx + ny * sy;
            const double num = nx * (v0.x - a.x) + ny * (v0.y - a.y);
            if (den == 0.0) {
                if (num > 0.0) return -1.0; // parallel and outside
            } else {
                const double t = num / den;
                if (den > 0.0) {
                    if (t > t0) t0 = t;
                } else {
                    if (t < t1) t1 = t;
                }
            }
        }
        v0 = v1;
    }
    if (!(t0 < t1)) return -1.0;
    const double cx = a.x + t0 * sx, cy = a.y + t0 * sy;
    const double dx = a.x + t1 * sx, dy = a.y + t1 * sy;
    if (c) {
        *c = P2{cx, cy};
        *d = P2{dx, dy};
    }
    const double ex = dx - cx, ey = dy - cy;
    return sqrt(ex * ex + ey * ey);
}

// the same clip on a polygon held in registers (MP slots, n <= MP corners; static indices): every operation of
// cyrus_beck_length in the same order -- only the vertex loads have moved in front of the loop
template <int MP>
__device__ __forceinline__ double cyrus_beck_length_regs(const P2 (&v)[MP], int n, P2 a, P2 b) {
    const double sx = b.x - a.x, sy = b.y - a.y;
    double t0 = 0.0, t1 = 1.0;
    bool outside = false;
    P2 v0 = v[0];
#pragma unroll
    for (int i = 0; i < MP; i++) {
        if (i < n) {
            const P2 v1 = (i + 1 < MP && i + 1 < n) ? v[i + 1 < MP ? i + 1 : 0] : v[0];
            const double wx = v1.x - v0.x, wy = v1.y - v0.y;
            if (wx != 0.0 || wy != 0.0) {
                const double nx = -wy, ny = wx; // inward normal of a CCW polygon
                const double den = nx * sx + ny * sy;
                const double num = nx * (v0.x - a.x) + ny * (v0.y - a.y);
                if (den == 0.0) {
                    if (num > 0.0) outside = true; // parallel and outside
                } else {
                    const double t = num / den;
                    if (den > 0.0) {
                        if (t > t0) t0 = t;
                    } else {
                        if (t < t1) t1 = t;
                    }
                }
            }
            v0 = v1;
        }
    }
    if (outside || !(t0 < t1)) return -1.0;
    const double cx = a.x + t0 * sx, cy = a.y + t0 * sy;
    const double dx = a.x + t1 * sx, dy = a.y + t1 * sy;
    const double ex = dx - cx, ey = dy - cy;
    return sqrt(ex * ex + ey * ey);
}

struct EdgeBox {
    P2 a, b;
    double xmin, xmax, ymin, ymax;
    float qx0, qx1, qy0, qy1; // conservative float box relative to the grid origin
};

__device__ __forceinline__ EdgeBox load_edge(const double *__restrict__ edge_xy, int64_t e, const GridParams &g) {
    EdgeBox q;
    q.a = load_p2(edge_xy, (int)(2 * e));
    q.b = load_p2(edge_xy, (int)(2 * e + 1));
    q.xmin = fmin(q.a.x, q.b.x);
    q.xmax = fmax(q.a.x, q.b.x);
    q.ymin = fmin(q.a.y, q.b.y);
    q.ymax = fmax(q.a.y, q.b.y);
    q.qx0 = f32_below(q.xmin - g.x0);
    q.qx1 = f32_above(q.xmax - g.x0);
    q.qy0 = f32_below(q.ymin - g.y0);
    q.qy1 = f32_above(q.ymax - g.y0);
    return q;
}

// Level-0 cells of the box corners, once: the level-l cell of x is (level-0 cell) >> l exactly (cell sizes are power-of-two
// multiples and the clamped ranges nest), and a record that can overlap starts at most one cell below the box's lower corner --
// the face search's rule (xr_overlap.hip: k_search), no per-level floating-point cell arithmetic.
struct EdgeCells {
    int x0, x1, y0, y1;
};
__device__ __forceinline__ EdgeCells edge_cells0(const EdgeBox &q, const GridParams &g) {
    return EdgeCells{cell_coord(q.xmin, g.x0, g.inv_h0, g.nx[0]), cell_coord(q.xmax, g.x0, g.inv_h0, g.nx[0]),
                     cell_coord(q.ymin, g.y0, g.inv_h0, g.ny[0]), cell_coord(q.ymax, g.y0, g.inv_h0, g.ny[0])};
}
__device__ __forceinline__ void edge_level_range(const EdgeCells &c, int l, int &cx0, int &cx1, int &cy0, int &cy1) {
    const int sh = l * LEVEL_SHIFT;
    cx0 = max((c.x0 >> sh) - 1, 0);
    cx1 = c.x1 >> sh;
    cy0 = max((c.y0 >> sh) - 1, 0);
    cy1 = c.y1 >> sh;
}
// number of grid cells (all levels) under the edge's box
__device__ __forceinline__ int64_t edge_cells(const EdgeCells &c, const GridParams &g) {
    int64_t total = 0;
    for (int l = 0; l < g.n_levels; l++) {
        int cx0, cx1, cy0, cy1;
        edge_level_range(c, l, cx0, cx1, cy0, cy1);
        total += (int64_t)(cx1 - cx0 + 1) * (cy1 - cy0 + 1);
    }
    return total;
}

// the records [r0, r1) of a run of grid cells against the edge's f32 box: CAND(record) for every record that passes
template <typename Cand>
__device__ __forceinline__ void edge_cell(const EdgeBox &q, int r0, int r1, const float4 *__restrict__ rbb, Cand &&cand) {
    // four independent loads in flight per step (one at a time, a run of n records was n dependent round trips: the walk kernel
    // waited 79 % of its wave cycles); a load beyond the run reads the next run or rec_bb's padding and is masked
    constexpr int LOADS = 4;
    static_assert(LOADS - 1 <= EDGE_WALK_PAD, "rec_bb padding");
    for (int r = r0; r < r1; r += LOADS) {
        float4 bb[LOADS];
#pragma unroll
        for (int u = 0; u < LOADS; u++) bb[u] = rbb[r + u];
#pragma unroll
        for (int u = 0; u < LOADS; u++)
            if (r + u < r1 && box_gap(bb[u], q.qx0, q.qx1, q.qy0, q.qy1) <= 0.0f) cand(r + u);
    }
}

// The walk along an edge.  Per level the cells along the edge's MAJOR axis are visited; for each of them the piece
// of the segment inside the slab of that cell's records ([origin, origin + 2 h): a record starts in its cell and is
// shorter than h) gives the few cells of the minor axis that can hold a candidate -- O(length / h) cells per level
// instead of the O((length / h)^2) of the edge's box.
struct EdgeWalk {
    bool xmajor;
    double a_maj, s_maj, a_min, s_min, org_maj, org_min, lo_maj, hi_maj, lo_min, hi_min;
};

__device__ __forceinline__ EdgeWalk edge_walk_setup(const EdgeBox &q, const GridParams &g) {
    EdgeWalk w;
    const double sx = q.b.x - q.a.x, sy = q.b.y - q.a.y;
    w.xmajor = fabs(sx) >= fabs(sy);
    w.a_maj = w.xmajor ? q.a.x : q.a.y;
    w.s_maj = w.xmajor ? sx : sy;
    w.a_min = w.xmajor ? q.a.y : q.a.x;
    w.s_min = w.xmajor ? sy : sx;
    w.org_maj = w.xmajor ? g.x0 : g.y0;
    w.org_min = w.xmajor ? g.y0 : g.x0;
    w.lo_maj = w.xmajor ? q.xmin : q.ymin;
    w.hi_maj = w.xmajor ? q.xmax : q.ymax;
    w.lo_min = w.xmajor ? q.ymin : q.xmin;
    w.hi_min = w.xmajor ? q.ymax : q.xmax;
    return w;
}

// minor-axis cell range [ka, kb] of major cell cm on a level (empty: ka > kb)
__device__ __forceinline__ void edge_minor_range(const EdgeWalk &w, int cm, double h, double inv_h, int n_maj, int n_min,
                                                 int o0, int o1, int &ka, int &kb) {
    const double pad = 1e-6 * h;
    // (the first / last cell of a level also holds whatever the clamping of cell_coord put there: no bound there)
    const double slab_lo = cm == 0 ? -INFINITY : w.org_maj + cm * h - pad;
    const double slab_hi = cm == n_maj - 1 ? INFINITY : w.org_maj + (cm + 2) * h + pad;
    double t0 = 0.0, t1 = 1.0;
    ka = 1;
    kb = 0;
    if (w.s_maj == 0.0) {
        if (w.a_maj < slab_lo || w.a_maj > slab_hi) return;
    } else {
        double u = (slab_lo - w.a_maj) / w.s_maj, v = (slab_hi - w.a_maj) / w.s_maj;
        if (u > v) { const double x = u; u = v; v = x; }
        t0 = fmax(t0, u);
        t1 = fmin(t1, v);
        if (t0 > t1) return;
    }
    const double m0 = w.a_min + t0 * w.s_min, m1 = w.a_min + t1 * w.s_min;
    const double mlo = fmin(m0, m1) - pad, mhi = fmax(m0, m1) + pad;
    ka = cell_coord(mlo - 2.0 * h, w.org_min, inv_h, n_min);
    kb = cell_coord(mhi, w.org_min, inv_h, n_min);
    if (ka < o0) ka = o0;
    if (kb > o1) kb = o1;
}

// the 64 lanes of a wave share one (long) edge: lane i takes items i, i + 64, ... of every level's (major cell, minor
// residue) list.  SYNC() is called by all lanes together after every round of 64 items (the wave flushes its stage there).
template <typename Cand, typename Sync>
__device__ __forceinline__ void edge_walk_wave(const EdgeBox &q, const GridParams &g, const int32_t *__restrict__ cell_start,
                                               const float4 *__restrict__ rbb, Cand &&cand, Sync &&sync) {
    constexpr int MINOR_W = 6;
    const EdgeWalk w = edge_walk_setup(q, g);
    const int lane = threadIdx.x & 63;
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
        if (cell_start[base] == cell_start[base + nx * ny]) continue; // (a level without records; uniform)
        const int n_maj = w.xmajor ? nx : ny, n_min = w.xmajor ? ny : nx;
        const int c0 = cell_coord(w.lo_maj - h, w.org_maj, inv_h, n_maj), c1 = cell_coord(w.hi_maj, w.org_maj, inv_h, n_maj);
        const int o0 = cell_coord(w.lo_min - h, w.org_min, inv_h, n_min), o1 = cell_coord(w.hi_min, w.org_min, inv_h, n_min);
        const int64_t work = (int64_t)(c1 - c0 + 1) * MINOR_W;
        for (int64_t c_first = 0; c_first < work; c_first += 64) { // (wave-uniform)
            const int64_t c = c_first + lane;
            if (c < work) {
                const int cm = c0 + (int)(c / MINOR_W), k = (int)(c % MINOR_W);
                int ka, kb;
                edge_minor_range(w, cm, h, inv_h, n_maj, n_min, o0, o1, ka, kb);
                for (int cn = ka + k; cn <= kb; cn += MINOR_W) {
                    const int cx = w.xmajor ? cm : cn, cy = w.xmajor ? cn : cm;
                    const int r0 = cell_start[base + cy * nx + cx], r1 = cell_start[base + cy * nx + cx + 1];
                    if (r0 != r1) edge_cell(q, r0, r1, rbb, cand);
                }
            }
            sync();
        }
    }
}

// all grid cells of the edge's box, level by level (short edges: a handful of cells, no per-cell arithmetic).  The record runs
// of FOUR levels (up to four grid rows each) are fetched together before any of them is walked: three round trips for the ten
// levels of the benchmark mesh -- eight of them all but empty -- instead of one per level and row.
template <typename Cand>
__device__ __forceinline__ void edge_walk_box(const EdgeBox &q, const EdgeCells &c, const GridParams &g,
                                              const int32_t *__restrict__ cell_start, const float4 *__restrict__ rbb, Cand &&cand) {
    constexpr int LCH = 4, RCH = 4;
    for (int l0 = 0; l0 < g.n_levels; l0 += LCH) { // (uniform)
        int r0[LCH][RCH], r1[LCH][RCH];
#pragma unroll
        for (int j = 0; j < LCH; j++) {
            const int l = l0 + j < g.n_levels ? l0 + j : g.n_levels - 1;
            const int nx = g.nx[l], base = g.base[l];
            int cx0, cx1, cy0, cy1;
            edge_level_range(c, l, cx0, cx1, cy0, cy1);
#pragma unroll
            for (int k = 0; k < RCH; k++) {
                const bool has = l0 + j < g.n_levels && cy0 + k <= cy1;
                const int cy = has ? cy0 + k : cy0;
                r0[j][k] = cell_start[base + cy * nx + cx0];
                r1[j][k] = has ? cell_start[base + cy * nx + cx1 + 1] : r0[j][k];
            }
        }
#pragma unroll
        for (int j = 0; j < LCH; j++) {
#pragma unroll
            for (int k = 0; k < RCH; k++) edge_cell(q, r0[j][k], r1[j][k], rbb, cand);
        }
        // (rows beyond the four fetched ones: a tall box on a fine level)
#pragma unroll
        for (int j = 0; j < LCH; j++) {
            if (l0 + j < g.n_levels) {
                const int l = l0 + j, nx = g.nx[l], base = g.base[l];
                int cx0, cx1, cy0, cy1;
                edge_level_range(c, l, cx0, cx1, cy0, cy1);
                for (int cy = cy0 + RCH; cy <= cy1; cy++)
                    edge_cell(q, cell_start[base + cy * nx + cx0], cell_start[base + cy * nx + cx1 + 1], rbb, cand);
            }
        }
    }
}

// append `item` to a list for the lanes with `flag`: one returning atomic per wave instead of one per lane
// (every lane of the wave that is still active must call this together)
__device__ __forceinline__ void wave_append(bool flag, int32_t item, int32_t *__restrict__ list,
                                            int32_t *__restrict__ count) {
    const unsigned long long mask = __ballot(flag);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, leader);
    if (flag) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = item;
}


// ---- the pipeline: walk -> flat candidate queue -> clip -> scan -> fill -> row sort -------------------------------------
// Until round 5 the thread-per-edge pass walked the grid, parked up to 48 candidates per edge in LDS (50 KB a block: three
// blocks per CU for a kernel that is a chain of dependent loads), dealt the Cyrus-Beck clips of a wave's 64 edges out over its
// lanes, kept six hits per edge in a side buffer and sent edges with more hits through the whole walk a second time; long edges
// had a wave each that walked AND clipped.  Round 6: the walk only EMITS (edge, record) candidates into a flat queue -- a wave's
// 64 edges share one LDS stage of 1024 entries (20 KB a block), so an edge may have hundreds of candidates as long as its wave
// stays under the stage -- and one thread per CANDIDATE clips: perfectly balanced, no slots, no second walk, no hit pool.  The
// clip leaves (edge, face, length) in the queue entry itself; after the scan of the row counts the fill pass runs over the queue
// once more.  1M edges of exponentially distributed length over 1M triangles: device part 1.56 -> see DESIGN ms.
static constexpr int EDGE_STAGE = 1024;     // candidates a wave of 64 edges may stage (16 per edge on average)
static constexpr int EDGE_BIG_STAGE = 2048; // candidates one long edge stages between two flushes
static constexpr int EQ_STRIDE = 32;        // words between the eight region cursors: a 128-byte line each
// counters: [0] wave-per-edge list length, [1] rows to sort (k_edge_rows_sort), [2] bit 0: a queue region overflowed,
// [8 + x * EQ_STRIDE] cursor of queue region x
static constexpr int EQ_CURSORS = 8;
static constexpr int EQ_WORDS = EQ_CURSORS + 8 * EQ_STRIDE;

// reserve n entries of region x; -> first entry, or -1 (overflow: the flag is set, the host regrows the queue and starts over)
__device__ __forceinline__ int64_t queue_reserve(int32_t *__restrict__ counters, int x, int n, int64_t region_cap) {
    const int32_t b = atomicAdd(&counters[EQ_CURSORS + x * EQ_STRIDE], n);
    if (b < 0 || (int64_t)b + n > region_cap) {
        atomicOr(&counters[2], 1);
        return -1;
    }
    return (int64_t)x * region_cap + b;
}

// pass 0: the edges in a spatially coherent order.  A network's edges come in any order (the benchmark's are random): the 64
// edges of a wave then walk 64 different neighbourhoods of the grid, the clips of a wave gather 64 unrelated faces and the
// fill scatters into 64 unrelated rows.  Counting sort by the tile (EDGE_TILE x EDGE_TILE level-0 cells) of the edge's
// midpoint: histogram, scan, scatter of the edge ids -- the walk then takes edge perm[i].  The order inside a tile is whatever
// the atomics make it; no result depends on it (rows are ordered by edge id at the end).
static constexpr int EDGE_TILE = 4;
__device__ __forceinline__ int edge_tile(const double *__restrict__ edge_xy, int64_t e, const GridParams &g, int ntx) {
    const P2 a = load_p2(edge_xy, (int)(2 * e)), b = load_p2(edge_xy, (int)(2 * e + 1));
    const int cx = cell_coord(0.5 * (a.x + b.x), g.x0, g.inv_h0, g.nx[0]), cy = cell_coord(0.5 * (a.y + b.y), g.y0, g.inv_h0, g.ny[0]);
    return (cy / EDGE_TILE) * ntx + cx / EDGE_TILE; // (NaN coordinates: cell 0)
}
__global__ void __launch_bounds__(256)
k_edge_tile_count(const double *__restrict__ edge_xy, int64_t n_edge, int64_t e_base, GridParams g, int ntx,
                  int32_t *__restrict__ tile_of, int32_t *__restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_edge) return;
    const int t = edge_tile(edge_xy, e_base + i, g, ntx);
    // (the returning atomic gives the edge its rank inside the tile: the scatter needs no second one)
    reinterpret_cast<int2 *>(tile_of)[i] = make_int2(t, atomicAdd(hist + t, 1));
}
__global__ void __launch_bounds__(256)
k_edge_tile_scatter(const int32_t *__restrict__ tile_of, int64_t n_edge, int64_t e_base, const int32_t *__restrict__ start,
                    int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_edge) return;
    const int2 tr = reinterpret_cast<const int2 *>(tile_of)[i];
    perm[start[tr.x] + tr.y] = (int32_t)(e_base + i);
}

// pass 1, one thread per edge: walk the boxes of all levels, stage the candidates of the wave's 64 edges in LDS, then write
// the block's candidates as one stretch of the queue (ONE reservation per block).  Edges whose box spans many cells, and the
// edges of a wave whose stage ran full, are listed for the wave-per-edge kernel (their staged candidates are dropped: edge -1).
__global__ void __launch_bounds__(256)
k_edge_walk(const double *__restrict__ edge_xy, int64_t n_edge, int64_t e_base, GridParams g, const int32_t *__restrict__ cell_start,
            const float *__restrict__ rec_bb, int2 *__restrict__ queue, int64_t region_cap, int32_t *__restrict__ counters,
            int32_t *__restrict__ big_list, int big_cells, int stage_cap /* <= EDGE_STAGE (test hook: a tiny stage) */,
            const int32_t *__restrict__ perm /* optional: item i is edge perm[i] (k_edge_tile_scatter) instead of e_base + i */) {
    __shared__ int32_t sh_rec[4][EDGE_STAGE];
    __shared__ uint16_t sh_tag[4][EDGE_STAGE]; // owner lane | the candidate's number among its owner's << 6
    __shared__ int32_t sh_edge[256];
    __shared__ int32_t sh_first[256];          // first queue slot (inside the wave's stretch) of every lane's candidates
    __shared__ int32_t sh_n[4];
    __shared__ long long sh_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool live = i < n_edge;
    const int64_t e = !live ? 0 : perm ? (int64_t)perm[i] : e_base + i;
    sh_edge[tid] = (int32_t)e;
    if (lane == 0) sh_n[wv] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool big = false, over = false;
    int stored = 0;
    if (live) {
        const EdgeBox q = load_edge(edge_xy, e, g);
        const bool finite = q.xmin == q.xmin && q.ymin == q.ymin && q.xmax == q.xmax && q.ymax == q.ymax; // no NaN
        const EdgeCells cells = edge_cells0(q, g);
        big = finite && edge_cells(cells, g) > big_cells;
        if (finite && !big) {
            auto cand = [&](int r) {
                const int k = atomicAdd(&sh_n[wv], 1);
                if (k < stage_cap) {
                    sh_rec[wv][k] = r;
                    sh_tag[wv][k] = (uint16_t)(lane | (stored << 6));
                    stored++;
                } else {
                    over = true;
                }
            };
            edge_walk_box(q, cells, g, cell_start, reinterpret_cast<const float4 *>(rec_bb), cand);
        }
    }
    // (an edge that met a full stage may have candidates missing: it walks again in the wave kernel)
    const unsigned long long dropped = __ballot(over);
    wave_append(live && (big || over), (int32_t)e, big_list, counters);
    // The stage holds the wave's candidates in the order the lanes' appends happened to interleave.  They leave it GROUPED BY
    // EDGE, in walk order (lane l's candidates behind those of the lanes below it): the clip's 64 lanes then read the same two
    // end points and neighbouring records -- a few lines per load instead of 64 (the clip is bound by the address rate of
    // its gathers, not by arithmetic: 7 loads x 64 lines per round).
    {
        int incl = stored;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(incl, d, 64);
            if (lane >= d) incl += u;
        }
        sh_first[tid] = incl - stored;
    }
    __syncthreads();
    int n_w[4], total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        n_w[w] = sh_n[w] < stage_cap ? sh_n[w] : stage_cap;
        total += n_w[w];
    }
    if (tid == 0) sh_base = total > 0 ? (long long)queue_reserve(counters, blockIdx.x & 7, total, region_cap) : -1;
    __syncthreads();
    const long long base = sh_base;
    if (base < 0) return;
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) woff += w < wv ? n_w[w] : 0;
    for (int k = lane; k < n_w[wv]; k += 64) {
        const int tag = sh_tag[wv][k], owner = tag & 63;
        queue[base + woff + sh_first[wv * 64 + owner] + (tag >> 6)] =
            make_int2(((dropped >> owner) & 1ull) ? -1 : sh_edge[wv * 64 + owner], sh_rec[wv][k]);
    }
}

// pass 1b, long edges (and the edges of the waves whose stage ran full): one wave per edge walks the cells along the segment's
// major axis and stages the candidates; the stage is flushed -- one reservation -- whenever it is half full at the end of a
// round of 64 cells, and at the edge's end.  A candidate that meets a full stage in the middle of a round reserves its own entry.
__global__ void __launch_bounds__(256)
k_edge_walk_big(const double *__restrict__ edge_xy, GridParams g, const int32_t *__restrict__ cell_start,
                const float *__restrict__ rec_bb, int2 *__restrict__ queue, int64_t region_cap, int32_t *__restrict__ counters,
                const int32_t *__restrict__ big_list) {
    __shared__ int32_t sh_rec[4][EDGE_BIG_STAGE];
    __shared__ int32_t sh_n[4];
    const int nb = counters[0], wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x & 7;
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int i = blockIdx.x * 4 + wv; i < nb; i += gridDim.x * 4) { // (wave-uniform)
        const int32_t e = big_list[i];
        const EdgeBox q = load_edge(edge_xy, e, g);
        if (lane == 0) sh_n[wv] = 0;
        wave_sync();
        auto flush = [&](int at_least) {
            wave_sync();
            const int n = sh_n[wv] < EDGE_BIG_STAGE ? sh_n[wv] : EDGE_BIG_STAGE;
            if (n < at_least) return; // (uniform)
            long long base = -1;
            if (lane == 0) base = queue_reserve(counters, x, n, region_cap);
            base = __shfl(base, 0, 64);
            if (base >= 0)
                for (int k = lane; k < n; k += 64) queue[base + k] = make_int2(e, sh_rec[wv][k]);
            wave_sync();
            if (lane == 0) sh_n[wv] = 0;
            wave_sync();
        };
        auto cand = [&](int r) {
            const int k = atomicAdd(&sh_n[wv], 1);
            if (k < EDGE_BIG_STAGE) {
                sh_rec[wv][k] = r;
            } else {
                const int64_t pos = queue_reserve(counters, x, 1, region_cap);
                if (pos >= 0) queue[pos] = make_int2(e, r);
            }
        };
        edge_walk_wave(q, g, cell_start, reinterpret_cast<const float4 *>(rec_bb), cand, [&]() { flush(EDGE_BIG_STAGE / 2); });
        flush(1);
    }
}

// pass 2, one thread per candidate (persistent grid over the eight queue regions): the exact clip.  A piece of positive length
// turns the entry into (edge, face) + its length + its RANK in the face's row (the returning atomic that counts the row:
// the fill pass needs no second one); everything else becomes edge -1.
// MC = 3 / 4: dense meshes -- the corners, the end points and the length byte are fetched together, in front of the arithmetic
// (as a loop over the polygon in memory a candidate was five dependent round trips: 0.22 ms for 8.5M candidates, all latency);
// MC = 0: any mesh, from memory.
template <int MC>
__global__ void __launch_bounds__(256)
k_edge_clip(const double *__restrict__ edge_xy, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
            const int32_t *__restrict__ rec_off, int m, const int32_t *__restrict__ rec_face, int2 *__restrict__ queue,
            double *__restrict__ queue_len, int32_t *__restrict__ queue_rank, int64_t region_cap,
            const int32_t *__restrict__ counters, int32_t *__restrict__ row_count) {
    if (counters[2] & 1) return; // (a region overflowed: the host starts over with a longer queue)
    // One returning atomic per DISTINCT face of the block's 256 candidates (the edges come tile by tile, so these hit a few dozen
    // faces) instead of one per hit: device-scope atomics run at the memory side of the fabric at ~20 G/s -- 4M of them were
    // most of this kernel.  The block's rounds are software-pipelined: the entry, corners, end points and face id of round k + 1
    // are in flight while round k is clipped and counted (as a chain entry -> gathers -> face id -> atomic -> stores a round was
    // five dependent round trips with three barriers in between: 69 % of the wave cycles waited, PMC).
    __shared__ KeyTable sh_tab;
    constexpr int MV = MC > 0 ? MC : 1;
    const double2 *__restrict__ fx = reinterpret_cast<const double2 *>(rec_fxy);
    const double2 *__restrict__ ex = reinterpret_cast<const double2 *>(edge_xy);
    // the chunks of 256 entries of all eight regions as one list (uniform arithmetic on eight scalars)
    int64_t first_chunk[9];
    first_chunk[0] = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) first_chunk[x + 1] = first_chunk[x] + ((int64_t)counters[EQ_CURSORS + x * EQ_STRIDE] + 255) / 256;
    const int64_t n_chunks = first_chunk[8];
    struct Round {
        int64_t c;   // queue index of this thread's entry (-1: none)
        int2 pr;
        double2 vv[MV], pa, pb;
        int nl, face;
    };
    auto fetch = [&](int64_t chunk, Round &r) {
        r.c = -1;
        r.pr = make_int2(-1, 0);
        r.nl = 0;
        r.face = 0;
        if (chunk >= n_chunks) return;
        int x = 0;
#pragma unroll
        for (int k = 1; k < 8; k++) x += chunk >= first_chunk[k] ? 1 : 0;
        const int64_t i = (chunk - first_chunk[x]) * 256 + threadIdx.x;
        if (i >= counters[EQ_CURSORS + x * EQ_STRIDE]) return;
        r.c = (int64_t)x * region_cap + i;
        r.pr = queue[r.c];
        if (r.pr.x < 0) return;
        if constexpr (MC > 0) {
#pragma unroll
            for (int k = 0; k < MC; k++) r.vv[k] = fx[(int64_t)r.pr.y * MC + k];
            r.nl = rec_len[r.pr.y];
        }
        r.pa = ex[2 * (int64_t)r.pr.x];
        r.pb = ex[2 * (int64_t)r.pr.x + 1];
        r.face = rec_face[r.pr.y];
    };
    Round cur, nxt;
    fetch(blockIdx.x, cur);
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) { // (block-uniform)
        agg_clear(sh_tab);
        fetch(chunk + gridDim.x, nxt);
        double len = -1.0;
        if (cur.pr.x >= 0) {
            const P2 a{cur.pa.x, cur.pa.y}, b{cur.pb.x, cur.pb.y};
            if constexpr (MC > 0) {
                P2 v[MC];
#pragma unroll
                for (int k = 0; k < MC; k++) v[k] = P2{cur.vv[k].x, cur.vv[k].y};
                len = cyrus_beck_length_regs<MC>(v, cur.nl, a, b);
            } else {
                len = cyrus_beck_length(rec_fxy + 2 * face_vertex_base(rec_off, cur.pr.y, m), rec_len[cur.pr.y], a, b);
            }
        }
        const bool hit = len > 0.0; // (a degenerate piece of zero length is no intersection)
        __syncthreads();
        int slot = 0, rank = 0;
        if (hit) slot = agg_insert(sh_tab, cur.face, rank);
        __syncthreads();
        for (int s = threadIdx.x; s < AGG_SLOTS; s += 256)
            if (sh_tab.key[s] >= 0) sh_tab.base[s] = atomicAdd(row_count + sh_tab.key[s], sh_tab.cnt[s]);
        __syncthreads();
        if (hit) {
            queue_rank[cur.c] = sh_tab.base[slot] + rank;
            queue[cur.c] = make_int2(cur.pr.x, cur.face);
            queue_len[cur.c] = len;
        } else if (cur.pr.x >= 0) {
            queue[cur.c] = make_int2(-1, 0);
        }
        __syncthreads(); // (the table is cleared at the top of the next round)
        cur = nxt;
    }
}

// pass 3: the kept entries go to their rows at the rank the clip's atomic gave them (arbitrary order within a row;
// k_edge_rows_sort orders the rows by edge id)
__global__ void __launch_bounds__(256)
k_edge_fill(const int2 *__restrict__ queue, const double *__restrict__ queue_len, const int32_t *__restrict__ queue_rank,
            int64_t region_cap, const int32_t *__restrict__ counters, const int32_t *__restrict__ indptr,
            int32_t *__restrict__ indices, double *__restrict__ data,
            int64_t capacity /* entries the two arrays hold: the launch may precede the host's read of nnz */) {
    if (counters[2] & 1) return; // (a region overflowed: nothing was clipped)
    constexpr int U = 4; // entries per thread and round, their loads in flight together
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int x = 0; x < 8; x++) {
        const int64_t n = counters[EQ_CURSORS + x * EQ_STRIDE];
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
            int2 pr[U];
            int pos[U];
            double len[U];
#pragma unroll
            for (int u = 0; u < U; u++) pr[u] = i + u * stride < n ? queue[(int64_t)x * region_cap + i + u * stride] : make_int2(-1, 0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t c = (int64_t)x * region_cap + i + u * stride;
                pos[u] = pr[u].x >= 0 ? indptr[pr[u].y] + queue_rank[c] : 0;
                len[u] = pr[u].x >= 0 ? queue_len[c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (pr[u].x >= 0 && pos[u] < capacity) {
                    indices[pos[u]] = pr[u].x;
                    data[pos[u]] = len[u];
                }
            }
        }
    }
}

// rows ordered by edge id: short rows by one thread each, the others are queued.  The rows of a block's 256 faces are one
// stretch of the CSR: it is staged in LDS (coalesced), every thread insertion-sorts its row there, and the stretch goes back as
// whole lines.  (Until round 5 the insertion sort ran in global memory, every step waiting for the store before it: 0.097 ms for
// the 1M rows of the benchmark network; bitonic networks in registers -- 4 / 8 / 16 slots by row length -- took 0.178 ms: nearly
// every wave holds a row of more than eight entries and runs all three.)
static constexpr int ROWS_STAGE = 3072; // entries a block stages (36 KB); a denser block sorts in global memory
template <typename K, typename V>
__device__ __forceinline__ void row_insertion_sort(K *__restrict__ k, V *__restrict__ v, int n) {
    for (int i = 1; i < n; i++) {
        const int32_t key = k[i];
        const double val = v[i];
        int j = i - 1;
        while (j >= 0 && k[j] > key) {
            k[j + 1] = k[j];
            v[j + 1] = v[j];
            j--;
        }
        k[j + 1] = key;
        v[j + 1] = val;
    }
}

__global__ void __launch_bounds__(256)
k_edge_rows_sort(const int32_t *__restrict__ indptr, int64_t n_face, int32_t *__restrict__ indices,
                 double *__restrict__ data, int32_t *__restrict__ sort_list, int32_t *__restrict__ n_sort,
                 int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long) {
    __shared__ int32_t sh_k[ROWS_STAGE];
    __shared__ double sh_v[ROWS_STAGE];
    const int64_t f0 = (int64_t)blockIdx.x * 256, f = f0 + threadIdx.x;
    const int64_t f1 = f0 + 256 < n_face ? f0 + 256 : n_face;
    const int s0 = indptr[f0], total = indptr[f1] - s0;
    const bool staged = total <= ROWS_STAGE; // (uniform)
    if (staged) {
        for (int i = threadIdx.x; i < total; i += 256) {
            sh_k[i] = indices[s0 + i];
            sh_v[i] = data[s0 + i];
        }
        __syncthreads();
    }
    if (f < n_face) {
        const int s = indptr[f], n = indptr[f + 1] - s;
        if (n > XR_APPLY_LONG_ROW) long_rows[atomicAdd(n_long, 1)] = (int32_t)f;
        if (n > ROW_SORT_SMALL) sort_list[atomicAdd(n_sort, 1)] = (int32_t)f; // (k_edge_rows_sort_big, behind this kernel)
        else if (staged) row_insertion_sort(sh_k + (s - s0), sh_v + (s - s0), n);
        else row_insertion_sort(indices + s, data + s, n);
    }
    if (!staged) return;
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += 256) {
        indices[s0 + i] = sh_k[i];
        data[s0 + i] = sh_v[i];
    }
}

// one block per queued row.  Edge ids are distinct within a row, so the rank of an entry is the number of smaller
// ids: up to ROW_SORT_LDS entries the keys sit in LDS (bitonic network on (id, position) words); longer rows are
// ranked against global memory (quadratic, but a face crossed by > 4096 edges is not a regridding workload).
__global__ void __launch_bounds__(256)
k_edge_rows_sort_big(const int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data,
                     const int32_t *__restrict__ sort_list, const int32_t *__restrict__ n_sort,
                     int32_t *__restrict__ tmp_idx, double *__restrict__ tmp_val) {
    __shared__ unsigned long long keys[ROW_SORT_LDS];
    const int ns = *n_sort;
    for (int i = blockIdx.x; i < ns; i += gridDim.x) {
        const int f = sort_list[i];
        const int s = indptr[f], n = indptr[f + 1] - s;
        if (n <= ROW_SORT_LDS) {
            int np2 = 1;
            while (np2 < n) np2 <<= 1;
            for (int j = threadIdx.x; j < np2; j += 256)
                keys[j] = j < n ? (((unsigned long long)(uint32_t)indices[s + j] << 32) | (uint32_t)j) : ~0ull;
            for (int j = threadIdx.x; j < n; j += 256) tmp_val[s + j] = data[s + j];
            __syncthreads();
            for (int k = 2; k <= np2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < np2; t += 256) {
                        const int p = t ^ j;
                        if (p > t) {
                            const unsigned long long x = keys[t], y = keys[p];
                            const bool up = (t & k) == 0;
                            if ((x > y) == up) {
                                keys[t] = y;
                                keys[p] = x;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int j = threadIdx.x; j < n; j += 256) {
                indices[s + j] = (int32_t)(keys[j] >> 32);
                data[s + j] = tmp_val[s + (uint32_t)keys[j]];
            }
            __syncthreads();
        } else {
            for (int j = threadIdx.x; j < n; j += 256) {
                tmp_idx[s + j] = indices[s + j];
                tmp_val[s + j] = data[s + j];
            }
            __syncthreads();
            for (int j = threadIdx.x; j < n; j += 256) {
                const int key = tmp_idx[s + j];
                int rank = 0;
                for (int o = 0; o < n; o++) rank += tmp_idx[s + o] < key;
                indices[s + rank] = key;
                data[s + rank] = tmp_val[s + j];
            }
            __syncthreads();
        }
    }
}

// the end points of every (face, edge) entry of the CSR, in entry order: intersections[entry][2][2]
__global__ void __launch_bounds__(256)
k_edge_pieces(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_face,
              const double *__restrict__ fxy, const uint8_t *__restrict__ len, const int32_t *__restrict__ off, int m,
              const double *__restrict__ edge_xy, double *__restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    for (int p = indptr[f]; p < indptr[f + 1]; p++) {
        const int e = indices[p];
        P2 c{NAN, NAN}, d{NAN, NAN};
        cyrus_beck_length(fxy + 2 * face_vertex_base(off, f, m), len[f], load_p2(edge_xy, 2 * e), load_p2(edge_xy, 2 * e + 1), &c, &d);
        double *o = out + (int64_t)p * 4;
        o[0] = c.x; o[1] = c.y; o[2] = d.x; o[3] = d.y;
    }
}

// edge_xy_dev != nullptr: the end points already live in HBM (xr_edge_length_csr_dev): no upload, one walk launch
static void edge_length_csr(xr_mesh *tree, const double *edge_xy_host, int64_t n_edge, xr_csr *csr, const double *edge_xy_dev = nullptr) {
    const int64_t F = tree->n_face;
    csr->n = F;
    csr->m = n_edge;
    csr->nnz = 0;
    csr->indptr.alloc((size_t)F + 1);
    hipStream_t st = launch_stream();
    if (F == 0 || n_edge == 0) {
        XR_HIP(hipMemsetAsync(csr->indptr.get(), 0, sizeof(int32_t) * ((size_t)F + 1), st));
        csr->indices.alloc(0);
        csr->data.alloc(0);
        return;
    }
    mesh_prepare(tree, false);
    mesh_build_index(tree);
    DevBuf<double> edge_xy_own((size_t)(edge_xy_dev ? 1 : n_edge * 4));
    const double *edge_xy = edge_xy_dev ? edge_xy_dev : edge_xy_own.get();
    DevBuf<int32_t> row_count((size_t)F), big_list((size_t)n_edge), counters((size_t)EQ_WORDS);
    const size_t edge_bytes = sizeof(double) * 4 * (size_t)n_edge;
    const GridParams &g = tree->grid;
    const int grid_persistent = engine().num_cu * 8;
    const int big_cells = option(OPT_EDGE_BIG) > 0 ? (int)option(OPT_EDGE_BIG) : EDGE_BIG_CELLS;                          // test / tuning hook
    const int stage_cap = option(OPT_EDGE_STAGE) > 0 ? (int)std::min<int64_t>(option(OPT_EDGE_STAGE), EDGE_STAGE) : EDGE_STAGE; // test hook
    // candidates the queue holds: 16 per edge (the benchmark network has 10) and at least a million; a region that runs full
    // sets a flag, the clip and the fill do nothing, and everything is redone with a queue as long as the cursors say
    int64_t queue_pairs = option(OPT_EDGE_QUEUE) > 0 ? option(OPT_EDGE_QUEUE) : std::max<int64_t>(16 * n_edge, (int64_t)1 << 20);
    // (edges [e0, e0 + ne): tile sort -- histogram, scan, scatter -- then the walk in that order; option edge_sort = 0: as they come)
    const int ntx = (int)div_up(g.nx[0], EDGE_TILE), n_tiles = ntx * (int)div_up(g.ny[0], EDGE_TILE);
    const bool sorted = option(OPT_EDGE_SORT) != 0 && n_edge >= 4096;
    DevBuf<int32_t> tile_of((size_t)(sorted ? 2 * n_edge : 2)), perm((size_t)(sorted ? n_edge : 1)), tile_hist((size_t)(sorted ? n_tiles : 1)),
        tile_start((size_t)(sorted ? n_tiles + 1 : 1));
    auto walk = [&](int64_t e0, int64_t ne, int2 *queue, int64_t region_cap) {
        if (sorted) {
            XR_HIP(hipMemsetAsync(tile_hist.get(), 0, sizeof(int32_t) * (size_t)n_tiles, launch_stream()));
            XR_LAUNCH("edges_tile_count", k_edge_tile_count, dim3(div_up(ne, 256)), dim3(256), 0, edge_xy, ne, e0, g, ntx,
                      tile_of.get() + 2 * e0, tile_hist.get());
            exclusive_scan_i32(tile_hist.get(), tile_start.get(), n_tiles);
            XR_LAUNCH("edges_tile_scatter", k_edge_tile_scatter, dim3(div_up(ne, 256)), dim3(256), 0, tile_of.get() + 2 * e0, ne, e0,
                      tile_start.get(), perm.get() + e0);
        }
        XR_LAUNCH("edges_walk", k_edge_walk, dim3(div_up(ne, 256)), dim3(256), 0, edge_xy, ne, e0, g, tree->cell_start.get(),
                  tree->rec_bb.get(), queue, region_cap, counters.get(), big_list.get(), big_cells, stage_cap,
                  sorted ? perm.get() + e0 : (const int32_t *)nullptr);
    };
    bool uploaded = edge_xy_dev != nullptr;
    for (int attempt = 0;; attempt++) {
        XR_REQUIRE(attempt < 6, XR_ERR_LIMIT, "xr_edge_length_csr: the candidate queue does not fit");
        const int64_t region_cap = (div_up(queue_pairs, 8) + 63) / 64 * 64;
        XR_REQUIRE(region_cap < ((int64_t)1 << 30), XR_ERR_LIMIT, "xr_edge_length_csr: too many candidate pairs");
        DevBuf<int2> queue((size_t)(8 * region_cap));
        DevBuf<double> queue_len((size_t)(8 * region_cap));
        DevBuf<int32_t> queue_rank((size_t)(8 * region_cap));
        XR_HIP(hipMemsetAsync(row_count.get(), 0, sizeof(int32_t) * (size_t)F, st));
        XR_HIP(hipMemsetAsync(counters.get(), 0, sizeof(int32_t) * (size_t)EQ_WORDS, st));
        // The edge coordinates come from the host (32 bytes per edge: 0.86 ms of PCIe for 1M edges).  The walk is independent per
        // edge, so the upload goes in the staging pipeline's pieces and the walk of what has arrived runs on the side stream while
        // the next pieces are on their way (launches of 16 MB of coordinates: smaller ones lose to their tails what the overlap gains).
        constexpr size_t pipe_bytes = (size_t)16 << 20;
        const bool piped = !uploaded && edge_bytes >= pipe_bytes + ((size_t)4 << 20) && !current_lane() && !stream_override();
        if (piped) {
            // fill(pinned, off, n) is called for piece k BEFORE its DMA is enqueued, i.e. right after the DMA of piece k - 1 was: the
            // walk of piece k - 1 is forked behind that DMA (SideScope) and runs beside the DMA of piece k
            const char *src_bytes = reinterpret_cast<const char *>(edge_xy_host);
            size_t done = 0; // bytes whose walk has been enqueued
            auto walk_upto = [&](size_t upto, bool last = false) {
                if (upto <= done || (!last && upto - done < pipe_bytes)) return;
                SideScope side;
                walk((int64_t)(done / 32), (int64_t)((upto - done) / 32), queue.get(), region_cap);
                done = upto;
            };
            h2d_staged(edge_xy_own.get(), edge_bytes, [&](char *pinned, size_t off, size_t n) {
                walk_upto(off);
                parallel_ranges(n, 64, [=](size_t b, size_t e) { memcpy(pinned + b, src_bytes + off + b, e - b); });
            });
            walk_upto(edge_bytes, true);
            side_join();
        } else {
            if (!uploaded) h2d(edge_xy_own.get(), edge_xy_host, edge_bytes);
            walk(0, n_edge, queue.get(), region_cap);
        }
        uploaded = true;
        XR_LAUNCH("edges_walk_big", k_edge_walk_big, dim3(grid_persistent), dim3(256), 0, edge_xy, g, tree->cell_start.get(),
                  tree->rec_bb.get(), queue.get(), region_cap, counters.get(), big_list.get());
        // (a persistent grid of what is resident: 92-96 registers with the next round's operands in flight, five waves per SIMD)
#define XR_EDGE_CLIP(MC)                                                                                                           \
    XR_LAUNCH("edges_clip", k_edge_clip<MC>, dim3(engine().num_cu * (MC > 0 ? 5 : 7)), dim3(256), 0, edge_xy, tree->rec_fxy.get(), tree->rec_len.get(), \
              tree->record_off(), tree->m, tree->rec_face.get(), queue.get(), queue_len.get(), queue_rank.get(), region_cap,        \
              counters.get(), row_count.get())
        if (tree->record_off() == nullptr && tree->m == 3) XR_EDGE_CLIP(3);
        else if (tree->record_off() == nullptr && tree->m == 4) XR_EDGE_CLIP(4);
        else XR_EDGE_CLIP(0);
#undef XR_EDGE_CLIP
        exclusive_scan_i32(row_count.get(), csr->indptr.get(), F);
        // The fill needs the row pointers, not the host: it goes into arrays sized by a guess (five pieces per edge; the benchmark
        // network has four) BEFORE the host reads the cursors and nnz -- the two read-backs, two allocations and the launch no
        // longer sit between the scan and the fill with the device idle.  A matrix that does not fit is filled again.
        auto fill = [&](int64_t capacity) {
            csr->indices.alloc((size_t)capacity);
            csr->data.alloc((size_t)capacity);
            XR_LAUNCH("edges_fill", k_edge_fill, dim3(grid_persistent), dim3(256), 0, queue.get(), queue_len.get(), queue_rank.get(),
                      region_cap, counters.get(), csr->indptr.get(), csr->indices.get(), csr->data.get(), capacity);
        };
        const int64_t guess = std::min<int64_t>(8 * region_cap, 5 * n_edge + ((int64_t)1 << 16));
        // (ONE read-back on the way: nnz, with the fill enqueued behind the copy -- it runs while the value travels and the host
        // sizes the row sort.  A queue region that overflowed left the clip and the fill idle and every row empty: nnz = 0 then,
        // and only then -- or for the debug line -- the cursors are read as well.)
        const int32_t P = read_scalar(csr->indptr.get() + F, [&] { fill(guess); });
        XR_REQUIRE(P >= 0, XR_ERR_LIMIT, "nnz exceeds the int32 range");
        if (P == 0 || (option(OPT_DEBUG) & 8)) {
            int32_t c[EQ_WORDS];
            d2h(c, counters.get(), sizeof(c));
            int64_t longest = 0, pairs = 0;
            for (int x = 0; x < 8; x++) {
                const int64_t n = c[EQ_CURSORS + x * EQ_STRIDE] < 0 ? ((int64_t)1 << 31) : c[EQ_CURSORS + x * EQ_STRIDE];
                longest = std::max(longest, n);
                pairs += n;
            }
            if (c[2] & 1) { // (the cursors kept counting: they say what the regions need)
                queue_pairs = std::max(2 * queue_pairs, 8 * (longest + longest / 8 + 1024));
                continue;
            }
            if (option(OPT_DEBUG) & 8)
                fprintf(stderr, "[edges] %lld edges: %d with a wave of their own, %lld candidate pairs (queue %lld), nnz %d\n",
                        (long long)n_edge, c[0], (long long)pairs, (long long)(8 * region_cap), P);
        }
        csr->nnz = P;
        if (P > guess) fill(P);
        if (P == 0) {
            csr->indices.alloc(0);
            csr->data.alloc(0);
            return;
        }
        break;
    }
    const int32_t P = (int32_t)csr->nnz;
    DevBuf<int32_t> sort_list((size_t)(P / ROW_SORT_SMALL + 1));
    csr->long_rows.alloc((size_t)(P / XR_APPLY_LONG_ROW + 1));
    csr->n_long.alloc(1);
    XR_HIP(hipMemsetAsync(csr->n_long.get(), 0, sizeof(int32_t), st));
    XR_LAUNCH("edge_rows_sort", k_edge_rows_sort, dim3(div_up(F, 256)), dim3(256), 0, csr->indptr.get(), F,
              csr->indices.get(), csr->data.get(), sort_list.get(), counters.get() + 1, csr->long_rows.get(),
              csr->n_long.get());
    // (the queued rows' kernel reads their number on the device: it is launched without the host knowing it -- one read-back at
    // the end, of the long-row count the apply wants, instead of two in front of this launch)
    DevBuf<int32_t> tmp_idx((size_t)P);
    DevBuf<double> tmp_val((size_t)P);
    XR_LAUNCH("edge_rows_sort_big", k_edge_rows_sort_big, dim3(engine().num_cu * 4), dim3(256), 0, csr->indptr.get(),
              csr->indices.get(), csr->data.get(), sort_list.get(), counters.get() + 1, tmp_idx.get(), tmp_val.get());
    csr->has_long = read_scalar(csr->n_long.get()) > 0;
    stream_sync();
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_edge_length_csr(xr_mesh *tree, const double *edge_xy, int64_t n_edge, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && out && (edge_xy || n_edge == 0), XR_ERR_INVALID, "xr_edge_length_csr: NULL argument");
    XR_REQUIRE(n_edge >= 0 && n_edge < ((int64_t)1 << 30), XR_ERR_LIMIT, "xr_edge_length_csr: too many edges");
    xr_csr *csr = new xr_csr();
    try {
        edge_length_csr(tree, edge_xy, n_edge, csr);
        stream_sync();
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_edge_length_csr_dev(xr_mesh *tree, const double *edge_xy_dev, int64_t n_edge, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && out && (edge_xy_dev || n_edge == 0), XR_ERR_INVALID, "xr_edge_length_csr_dev: NULL argument");
    XR_REQUIRE(n_edge >= 0 && n_edge < ((int64_t)1 << 30), XR_ERR_LIMIT, "xr_edge_length_csr_dev: too many edges");
    xr_csr *csr = new xr_csr();
    try {
        edge_length_csr(tree, nullptr, n_edge, csr, edge_xy_dev);
        stream_sync();
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_edge_pieces(xr_mesh *tree, const xr_csr *csr, const double *edge_xy, int64_t n_edge, double *intersections) {
    XR_API_BEGIN
    XR_REQUIRE(tree && csr && (intersections || csr->nnz == 0), XR_ERR_INVALID, "xr_edge_pieces: NULL argument");
    XR_REQUIRE(csr->n == tree->n_face && csr->m == n_edge && (edge_xy || n_edge == 0), XR_ERR_INVALID,
               "xr_edge_pieces: the matrix does not belong to this mesh / these edges");
    if (csr->nnz > 0) {
        mesh_prepare(tree, true);
        mesh_face_coords(tree);
        DevBuf<double> xy((size_t)n_edge * 4), out((size_t)csr->nnz * 4);
        h2d(xy.get(), edge_xy, sizeof(double) * 4 * (size_t)n_edge);
        XR_LAUNCH("edge_pieces", k_edge_pieces, dim3(div_up(csr->n, 256)), dim3(256), 0, csr->indptr.get(),
                  csr->indices.get(), csr->n, tree->fxy.get(), tree->len.get(), tree->caller_off(), tree->m, xy.get(), out.get());
        d2h(intersections, out.get(), sizeof(double) * 4 * (size_t)csr->nnz);
        stream_sync();
    }
    XR_API_END
}

} // extern "C"
