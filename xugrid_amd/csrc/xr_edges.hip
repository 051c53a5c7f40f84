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

namespace xr {

static constexpr int EDGE_BIG_CELLS = 64;  // edges whose box covers more grid cells (all levels) get a wave of their own
static constexpr int EDGE_SLOTS = 6;       // hits per edge the count pass keeps for the fill pass (slot-major side buffer)
static constexpr int ROW_SORT_SMALL = 16;  // rows up to this length are insertion-sorted by one thread (in global memory: every step waits for the
                                           // store before it -- at 48 the few rows of 30-48 entries WERE the kernel, 0.23 ms; 16: 0.10)
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

// number of grid cells (all levels) under the edge's box
__device__ __forceinline__ int64_t edge_cells(const EdgeBox &q, const GridParams &g) {
    int64_t total = 0;
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int cx0 = cell_coord(q.xmin - h, g.x0, inv_h, g.nx[l]), cx1 = cell_coord(q.xmax, g.x0, inv_h, g.nx[l]);
        const int cy0 = cell_coord(q.ymin - h, g.y0, inv_h, g.ny[l]), cy1 = cell_coord(q.ymax, g.y0, inv_h, g.ny[l]);
        total += (int64_t)(cx1 - cx0 + 1) * (cy1 - cy0 + 1);
    }
    return total;
}

// the records of one grid cell against the edge's f32 box: CAND(record) for every record that passes.  The exact clip
// (edge_test) is NOT run from here by the thread-per-edge kernels: they park the candidates in LDS and clip them
// afterwards in a loop all lanes of a wave step through together (a clip is ~150 instructions with a division per face
// side; run from inside the walk, every lane that found a candidate made the whole wave execute it).
template <typename Cand>
__device__ __forceinline__ void edge_cell(const EdgeBox &q, int r0, int r1, const float4 *__restrict__ rbb,
                                          const double *__restrict__, const uint8_t *__restrict__, const int32_t *__restrict__,
                                          int, const int32_t *__restrict__, Cand &&cand) {
    for (int r = r0; r < r1; r++) {
        const float4 bb = rbb[r];
        if (box_gap(bb, q.qx0, q.qx1, q.qy0, q.qy1) <= 0.0f) cand(r);
    }
}

// exact clip of the edge against record r; HIT(face id, length) for a piece of positive length
template <typename Hit>
__device__ __forceinline__ void edge_test(const EdgeBox &q, int r, const double *__restrict__ rec_fxy,
                                          const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m,
                                          const int32_t *__restrict__ rec_face, Hit &&hit) {
    const double len = cyrus_beck_length(rec_fxy + 2 * face_vertex_base(rec_off, r, m), rec_len[r], q.a, q.b);
    if (len > 0.0) hit(rec_face[r], len); // (a degenerate piece of zero length is no intersection)
}

static constexpr int EDGE_PARK = 16; // candidates parked per edge; further ones are clipped on the spot

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

// one thread walks the whole edge (BLOCK = false), or the 64 lanes of a wave share it (BLOCK = true)
template <bool BLOCK, typename Hit>
__device__ __forceinline__ void edge_walk(const EdgeBox &q, const GridParams &g, const int32_t *__restrict__ cell_start,
                                          const float4 *__restrict__ rbb, const double *__restrict__ rec_fxy,
                                          const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m,
                                          const int32_t *__restrict__ rec_face, Hit &&hit) {
    constexpr int MINOR_W = 6;
    const EdgeWalk w = edge_walk_setup(q, g);
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
        const int n_maj = w.xmajor ? nx : ny, n_min = w.xmajor ? ny : nx;
        const int c0 = cell_coord(w.lo_maj - h, w.org_maj, inv_h, n_maj), c1 = cell_coord(w.hi_maj, w.org_maj, inv_h, n_maj);
        const int o0 = cell_coord(w.lo_min - h, w.org_min, inv_h, n_min), o1 = cell_coord(w.hi_min, w.org_min, inv_h, n_min);
        if (BLOCK) {
            const int64_t work = (int64_t)(c1 - c0 + 1) * MINOR_W;
            for (int64_t c = threadIdx.x & 63; c < work; c += 64) {
                const int cm = c0 + (int)(c / MINOR_W), k = (int)(c % MINOR_W);
                int ka, kb;
                edge_minor_range(w, cm, h, inv_h, n_maj, n_min, o0, o1, ka, kb);
                for (int cn = ka + k; cn <= kb; cn += MINOR_W) {
                    const int cx = w.xmajor ? cm : cn, cy = w.xmajor ? cn : cm;
                    const int r0 = cell_start[base + cy * nx + cx], r1 = cell_start[base + cy * nx + cx + 1];
                    if (r0 != r1) edge_cell(q, r0, r1, rbb, rec_fxy, rec_len, rec_off, m, rec_face, hit);
                }
            }
        } else {
            for (int cm = c0; cm <= c1; cm++) {
                int ka, kb;
                edge_minor_range(w, cm, h, inv_h, n_maj, n_min, o0, o1, ka, kb);
                if (ka > kb) continue;
                if (w.xmajor) { // the minor cells of one major cell are cy = ka..kb at fixed cx: separate runs
                    for (int cy = ka; cy <= kb; cy++) {
                        const int r0 = cell_start[base + cy * nx + cm], r1 = cell_start[base + cy * nx + cm + 1];
                        if (r0 != r1) edge_cell(q, r0, r1, rbb, rec_fxy, rec_len, rec_off, m, rec_face, hit);
                    }
                } else { // cells cx = ka..kb of row cm are one contiguous record run
                    edge_cell(q, cell_start[base + cm * nx + ka], cell_start[base + cm * nx + kb + 1], rbb, rec_fxy, rec_len, rec_off,
                              m, rec_face, hit);
                }
            }
        }
    }
}

// all grid cells of the edge's box, level by level (short edges: a handful of cells, no per-cell arithmetic)
template <typename Hit>
__device__ __forceinline__ void edge_walk_box(const EdgeBox &q, const GridParams &g, const int32_t *__restrict__ cell_start,
                                              const float4 *__restrict__ rbb, const double *__restrict__ rec_fxy,
                                              const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m,
                                              const int32_t *__restrict__ rec_face, Hit &&hit) {
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], base = g.base[l];
        const int cx0 = cell_coord(q.xmin - h, g.x0, inv_h, nx), cx1 = cell_coord(q.xmax, g.x0, inv_h, nx);
        const int cy0 = cell_coord(q.ymin - h, g.y0, inv_h, g.ny[l]), cy1 = cell_coord(q.ymax, g.y0, inv_h, g.ny[l]);
        for (int cy = cy0; cy <= cy1; cy++)
            edge_cell(q, cell_start[base + cy * nx + cx0], cell_start[base + cy * nx + cx1 + 1], rbb, rec_fxy, rec_len, rec_off, m,
                      rec_face, hit);
    }
}

// pass 1, one thread per edge: count the hits per face; the first EDGE_SLOTS hits of every edge are kept in a
// slot-major side buffer so that the fill pass need not walk the grid again.  Edges with more hits are queued for
// a second walk (redo_list), edges whose box spans many cells for the block-per-edge kernels (big_list).
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

template <bool MAJOR_WALK>
__global__ void __launch_bounds__(256)
k_edges_count(const double *__restrict__ edge_xy, int64_t n_edge, GridParams g, const int32_t *__restrict__ cell_start,
              const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off,
              int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
              int32_t *__restrict__ big_list, int32_t *__restrict__ n_big, int32_t *__restrict__ redo_list,
              int32_t *__restrict__ n_redo, int32_t *__restrict__ edge_hits, int32_t *__restrict__ side_face,
              double *__restrict__ side_len, int big_cells) {
    __shared__ int32_t sh_park[EDGE_PARK][256];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = e < n_edge;
    EdgeBox q{};
    bool walk = false, big = false;
    if (live) {
        q = load_edge(edge_xy, e, g);
        const bool finite = q.xmin == q.xmin && q.ymin == q.ymin && q.xmax == q.xmax && q.ymax == q.ymax; // no NaN
        big = finite && edge_cells(q, g) > big_cells;
        walk = finite && !big;
    }
    int nh = 0;
    if (walk) {
        auto hit = [&](int face, double len) {
            atomicAdd(row_count + face, 1);
            if (nh < EDGE_SLOTS) {
                side_face[(int64_t)nh * n_edge + e] = face;
                side_len[(int64_t)nh * n_edge + e] = len;
            }
            nh++;
        };
        const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
        int np = 0;
        auto park = [&](int r) {
            if (np < EDGE_PARK) sh_park[np][threadIdx.x] = r;
            else edge_test(q, r, rec_fxy, rec_len, rec_off, m, rec_face, hit);
            np++;
        };
        if (MAJOR_WALK) edge_walk<false>(q, g, cell_start, rbb, rec_fxy, rec_len, rec_off, m, rec_face, park);
        else edge_walk_box(q, g, cell_start, rbb, rec_fxy, rec_len, rec_off, m, rec_face, park);
        const int parked = np < EDGE_PARK ? np : EDGE_PARK;
        for (int k = 0; k < parked; k++) edge_test(q, sh_park[k][threadIdx.x], rec_fxy, rec_len, rec_off, m, rec_face, hit);
    }
    const bool redo = nh > EDGE_SLOTS;
    wave_append(big, (int32_t)e, big_list, n_big);
    wave_append(redo, (int32_t)e, redo_list, n_redo);
    if (live) edge_hits[e] = redo ? 0 : nh;
}

// ---- thread-per-edge passes with the exact clips dealt out over the wave ------------------------------------------
// Diagnosis (1M edges of exponentially distributed length over 1M triangles, variants of k_edges_count): the walk alone
// takes 0.35 of the count pass's 0.98 ms; the rest are the Cyrus-Beck clips (~135 vector instructions each).  An edge has
// ~10 candidate faces on average but the longest of a wave 30-40, and a loop "clip my candidates" -- worse, the clips
// made on the spot once an edge's 16 parking slots were full -- makes the whole wave step through the longest lists one
// after the other.  Here the walk only PARKS (EDGE_DEAL slots per edge); the wave then treats the parked candidates of its
// 64 edges as ONE list and lane i clips items i, i + 64, ... (owner lane by a binary search over the wave's running
// counts, its edge by a cross-lane read; a hit takes the owner's next slot by an LDS atomic).  Edges with more candidates
// than slots go to the wave-per-edge kernels (big_list), which deal an edge's candidates over the lanes by construction.
// FILL = false: count pass; FILL = true: the listed (redo) edges write their rows.  EDGE_DEAL: parking slots per edge.
template <bool FILL, int EDGE_DEAL>
__global__ void __launch_bounds__(256)
k_edges_deal(const double *__restrict__ edge_xy, int64_t n_edge, GridParams g, const int32_t *__restrict__ cell_start,
             const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
             const int32_t *__restrict__ rec_off, int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
             int32_t *__restrict__ big_list, int32_t *__restrict__ n_big, int32_t *__restrict__ redo_list,
             int32_t *__restrict__ n_redo, int32_t *__restrict__ edge_hits, int32_t *__restrict__ side_face,
             double *__restrict__ side_len, int big_cells, const int32_t *__restrict__ indptr, int32_t *__restrict__ indices,
             double *__restrict__ data, const int32_t *__restrict__ todo_list, const int32_t *__restrict__ n_todo,
             int64_t e_base = 0 /* count pass over a PIECE of the edges: item i is edge e_base + i (n_edge = the piece's length; edge_xy
                                  and every per-edge output stay indexed by the edge's own id) */,
             int64_t n_edge_all = 0 /* ... and the number of ALL edges, the stride of the slot-major side buffers (0: n_edge) */) {
    __shared__ int32_t sh_park[EDGE_DEAL][256];
    __shared__ int32_t sh_hits[256];
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const int64_t n_items = FILL ? (int64_t)*n_todo : n_edge;
    const int64_t side_stride = n_edge_all > 0 ? n_edge_all : n_edge;
    const int64_t n_rounded = (n_items + 255) / 256 * 256; // (every lane of a wave takes part in the cross-lane reads)
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n_rounded; i += (int64_t)gridDim.x * 256) {
        const bool live = i < n_items;
        const int64_t e = live ? (FILL ? (int64_t)todo_list[i] : e_base + i) : 0;
        EdgeBox q{};
        bool walk = false, big = false;
        if (live) {
            q = load_edge(edge_xy, e, g);
            const bool finite = q.xmin == q.xmin && q.ymin == q.ymin && q.xmax == q.xmax && q.ymax == q.ymax; // no NaN
            big = !FILL && finite && edge_cells(q, g) > big_cells;
            walk = finite && !big;
        }
        sh_hits[tid] = 0;
        int np = 0;
        if (walk) {
            auto park = [&](int r) {
                sh_park[np < EDGE_DEAL ? np : EDGE_DEAL - 1][tid] = r;
                np++;
            };
            edge_walk_box(q, g, cell_start, rbb, rec_fxy, rec_len, rec_off, m, rec_face, park);
        }
        // (FILL: the listed edges had at most EDGE_DEAL candidates in the count pass -- the same walk)
        if (np > EDGE_DEAL) {
            big = !FILL;
            np = 0;
        }
        int incl = np;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int total = __shfl(incl, 63, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int first = 0; first < total; first += 64) { // (wave-uniform)
            const int item = first + lane;
            const bool has = item < total;
            int lo = 0, hi = 63; // owner = first lane whose running count exceeds the item
#pragma unroll
            for (int step = 0; step < 6; step++) {
                const int mid = (lo + hi) >> 1;
                const int v = __shfl(incl, mid, 64);
                if (v > item) hi = mid;
                else lo = mid + 1;
            }
            const int ol = has ? lo : 0;
            const int slot = item - (__shfl(incl, ol, 64) - __shfl(np, ol, 64));
            const P2 a{__shfl(q.a.x, ol, 64), __shfl(q.a.y, ol, 64)}, b{__shfl(q.b.x, ol, 64), __shfl(q.b.y, ol, 64)};
            const long long e_owner = __shfl((long long)e, ol, 64);
            if (has) {
                const int rr = sh_park[slot][wbase + ol];
                const double len = cyrus_beck_length(rec_fxy + 2 * face_vertex_base(rec_off, rr, m), rec_len[rr], a, b);
                if (len > 0.0) { // (a degenerate piece of zero length is no intersection)
                    const int face = rec_face[rr];
                    const int k_row = atomicAdd(row_count + face, 1);
                    if (FILL) {
                        indices[indptr[face] + k_row] = (int32_t)e_owner;
                        data[indptr[face] + k_row] = len;
                    } else {
                        const int k_edge = atomicAdd(&sh_hits[wbase + ol], 1);
                        if (k_edge < EDGE_SLOTS) {
                            side_face[(int64_t)k_edge * side_stride + e_owner] = face;
                            side_len[(int64_t)k_edge * side_stride + e_owner] = len;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!FILL) {
            const int nh = sh_hits[tid];
            const bool redo = nh > EDGE_SLOTS;
            wave_append(big, (int32_t)e, big_list, n_big);
            wave_append(redo, (int32_t)e, redo_list, n_redo);
            if (live) edge_hits[e] = redo ? 0 : nh;
        }
    }
}

// pass 2a: the kept hits go to their rows (arbitrary order within a row)
__global__ void __launch_bounds__(256)
k_edges_replay(int64_t n_edge, const int32_t *__restrict__ edge_hits, const int32_t *__restrict__ side_face,
               const double *__restrict__ side_len, int32_t *__restrict__ row_count,
               const int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edge) return;
    const int nh = edge_hits[e];
    for (int k = 0; k < nh; k++) {
        const int face = side_face[(int64_t)k * n_edge + e];
        const int pos = indptr[face] + atomicAdd(row_count + face, 1);
        indices[pos] = (int32_t)e;
        data[pos] = side_len[(int64_t)k * n_edge + e];
    }
}

// pass 2b: the edges with more than EDGE_SLOTS hits walk the grid once more
template <bool MAJOR_WALK>
__global__ void __launch_bounds__(256)
k_edges_redo(const double *__restrict__ edge_xy, GridParams g, const int32_t *__restrict__ cell_start,
             const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off,
             int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
             const int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data,
             const int32_t *__restrict__ redo_list, const int32_t *__restrict__ n_redo) {
    __shared__ int32_t sh_park[EDGE_PARK][256];
    const int n = *n_redo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t e = redo_list[i];
        const EdgeBox q = load_edge(edge_xy, e, g);
        auto hit = [&](int face, double len) {
            const int pos = indptr[face] + atomicAdd(row_count + face, 1);
            indices[pos] = (int32_t)e;
            data[pos] = len;
        };
        const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
        int np = 0;
        auto park = [&](int r) {
            if (np < EDGE_PARK) sh_park[np][threadIdx.x] = r;
            else edge_test(q, r, rec_fxy, rec_len, rec_off, m, rec_face, hit);
            np++;
        };
        if (MAJOR_WALK) edge_walk<false>(q, g, cell_start, rbb, rec_fxy, rec_len, rec_off, m, rec_face, park);
        else edge_walk_box(q, g, cell_start, rbb, rec_fxy, rec_len, rec_off, m, rec_face, park);
        const int parked = np < EDGE_PARK ? np : EDGE_PARK;
        for (int k = 0; k < parked; k++) edge_test(q, sh_park[k][threadIdx.x], rec_fxy, rec_len, rec_off, m, rec_face, hit);
    }
}

// long edges: one wave per edge
template <bool FILL>
__global__ void __launch_bounds__(256)
k_edges_big(const double *__restrict__ edge_xy, GridParams g, const int32_t *__restrict__ cell_start,
            const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off,
            int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
            const int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data,
            const int32_t *__restrict__ big_list, const int32_t *__restrict__ n_big) {
    const int nb = *n_big;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nb; i += gridDim.x * 4) {
        const int64_t e = big_list[i];
        const EdgeBox q = load_edge(edge_xy, e, g);
        auto hit = [&](int face, double len) {
            const int k = atomicAdd(row_count + face, 1);
            if (FILL) {
                indices[indptr[face] + k] = (int32_t)e;
                data[indptr[face] + k] = len;
            }
        };
        auto cand = [&](int r) { edge_test(q, r, rec_fxy, rec_len, rec_off, m, rec_face, hit); };
        edge_walk<true>(q, g, cell_start, reinterpret_cast<const float4 *>(rec_bb), rec_fxy, rec_len, rec_off, m, rec_face, cand);
    }
}

// Count pass of the wave-per-edge kernels that KEEPS its hits: the lanes append (face, length) to a stage of their wave in
// LDS; when the edge is done the wave reserves a stretch of the hit pool with one atomic and copies the stage out.  The
// fill pass is then a replay of the pool (one thread per hit) instead of a second walk over the long edges.  Edges with
// more hits than the stage holds, or that find the pool full, are listed for the walking fill pass (k_edges_big<true>).
static constexpr int BIG_STAGE_HITS = 256;
struct EdgeHit {
    int32_t edge, face;
    double len;
};

__global__ void __launch_bounds__(256)
k_edges_big_pool(const double *__restrict__ edge_xy, GridParams g, const int32_t *__restrict__ cell_start,
                 const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
                 const int32_t *__restrict__ rec_off, int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
                 const int32_t *__restrict__ big_list, const int32_t *__restrict__ n_big, EdgeHit *__restrict__ pool,
                 int32_t *__restrict__ pool_cursor /* [0] cursor, [1] capacity - first refused base (0: none refused) */, int pool_cap,
                 int32_t *__restrict__ walk_list, int32_t *__restrict__ n_walk) {
    __shared__ int32_t sh_face[4][BIG_STAGE_HITS];
    __shared__ double sh_len[4][BIG_STAGE_HITS];
    __shared__ int32_t sh_n[4];
    const int nb = *n_big, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + wv; i < nb; i += gridDim.x * 4) { // (wave-uniform)
        const int64_t e = big_list[i];
        const EdgeBox q = load_edge(edge_xy, e, g);
        if (lane == 0) sh_n[wv] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto hit = [&](int face, double len) {
            atomicAdd(row_count + face, 1);
            const int k = atomicAdd(&sh_n[wv], 1);
            if (k < BIG_STAGE_HITS) {
                sh_face[wv][k] = face;
                sh_len[wv][k] = len;
            }
        };
        auto cand = [&](int r) { edge_test(q, r, rec_fxy, rec_len, rec_off, m, rec_face, hit); };
        edge_walk<true>(q, g, cell_start, reinterpret_cast<const float4 *>(rec_bb), rec_fxy, rec_len, rec_off, m, rec_face, cand);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = sh_n[wv];
        int base = -1;
        if (lane == 0) {
            if (n <= BIG_STAGE_HITS && n > 0) {
                const int b = atomicAdd(pool_cursor, n);
                if (b >= 0 && b <= pool_cap - n) base = b;
                else if (b >= 0 && b < pool_cap) atomicMax(pool_cursor + 1, pool_cap - b); // (the cursor only grows: every later stretch is refused too)
            }
            if (n > 0 && base < 0) walk_list[atomicAdd(n_walk, 1)] = (int32_t)e;
        }
        base = __shfl(base, 0, 64);
        if (base >= 0)
            for (int k = lane; k < n; k += 64) pool[base + k] = EdgeHit{(int32_t)e, sh_face[wv][k], sh_len[wv][k]};
    }
}

// fill pass of the pooled hits: the granted stretches are exactly [0, end), end = the base of the first refused stretch
// (the cursor only grows, so nothing behind a refused stretch is granted) or the cursor itself
__global__ void __launch_bounds__(256)
k_edges_pool_replay(const EdgeHit *__restrict__ pool, const int32_t *__restrict__ pool_cursor, int pool_cap,
                    int32_t *__restrict__ row_count, const int32_t *__restrict__ indptr, int32_t *__restrict__ indices,
                    double *__restrict__ data) {
    const int end = pool_cap - (pool_cursor[1] > 0 ? pool_cursor[1] : 0);
    const int cur = pool_cursor[0] < 0 ? 0 : pool_cursor[0]; // (a wrapped cursor: nothing was granted after the wrap)
    const int n = cur < end ? cur : end;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const EdgeHit h = pool[i];
        const int pos = indptr[h.face] + atomicAdd(row_count + h.face, 1);
        indices[pos] = h.edge;
        data[pos] = h.len;
    }
}

// rows ordered by edge id: short rows by one thread each, the others are queued
__global__ void __launch_bounds__(256)
k_edge_rows_sort(const int32_t *__restrict__ indptr, int64_t n_face, int32_t *__restrict__ indices,
                 double *__restrict__ data, int32_t *__restrict__ sort_list, int32_t *__restrict__ n_sort,
                 int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_face) return;
    const int s = indptr[f], e = indptr[f + 1];
    if (e - s > XR_APPLY_LONG_ROW) long_rows[atomicAdd(n_long, 1)] = (int32_t)f;
    if (e - s > ROW_SORT_SMALL) {
        sort_list[atomicAdd(n_sort, 1)] = (int32_t)f;
        return;
    }
    for (int i = s + 1; i < e; i++) {
        const int key = indices[i];
        const double val = data[i];
        int j = i - 1;
        while (j >= s && indices[j] > key) {
            indices[j + 1] = indices[j];
            data[j + 1] = data[j];
            j--;
        }
        indices[j + 1] = key;
        data[j + 1] = val;
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

// edge_xy_dev != nullptr: the end points already live in HBM (xr_edge_length_csr_dev): no upload, one count launch
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
    struct { const double *p; const double *get() const { return p; } } edge_xy{edge_xy_dev ? edge_xy_dev : edge_xy_own.get()};
    DevBuf<int32_t> row_count((size_t)F), big_list((size_t)n_edge), counters(8); // [0] big, [1] rows to sort, [2] redo, [4] pool cursor, [5] its refusal mark, [6] big edges that walk again
    DevBuf<int32_t> edge_hits((size_t)n_edge), side_face((size_t)n_edge * EDGE_SLOTS), redo_list((size_t)n_edge);
    DevBuf<double> side_len((size_t)n_edge * EDGE_SLOTS);
    XR_HIP(hipMemsetAsync(row_count.get(), 0, sizeof(int32_t) * (size_t)F, st));
    XR_HIP(hipMemsetAsync(counters.get(), 0, sizeof(int32_t) * 8, st));
    // The edge coordinates come from the host (32 bytes per edge: 0.86 ms of PCIe for 1M edges, a third of the whole call).  With the
    // thread-per-edge count pass -- independent per edge -- the upload goes in the staging pipeline's pieces (4 MiB = 131072 edges)
    // and the count kernel of what has arrived runs on the side stream while the next pieces are on their way: XR_EDGE_PIPE=0
    // restores the single upload + single launch (A/B switch).
    const size_t edge_bytes = sizeof(double) * 4 * (size_t)n_edge;
    constexpr bool pipe_off = false;
    const GridParams &g = tree->grid;
    const int big_grid = engine().num_cu * 8;
    const int big_cells = option(OPT_EDGE_BIG) > 0 ? (int)option(OPT_EDGE_BIG) : EDGE_BIG_CELLS; // test / tuning hook
    const bool major = option(OPT_EDGE_WALK) != 0; // test / tuning hook
    const bool deal = !major && option(OPT_EDGE_KERNEL) == 0; // (test / A/B switch)
    const int deal_slots = (int)option(OPT_EDGE_DEAL); // test / tuning hook: parking slots per edge (24 / 32 / 40 / 48)
    // count pass of the edges [e0, e0 + ne) with the thread-per-edge kernel
    auto deal_count = [&](int64_t e0, int64_t ne) {
#define XR_DEAL_COUNT(P)                                                                                                              \
    XR_LAUNCH("edges_count", (k_edges_deal<false, P>), dim3(div_up(ne, 256)), dim3(256), 0, edge_xy.get(), ne, g,                      \
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,      \
              tree->rec_face.get(), row_count.get(), big_list.get(), counters.get(), redo_list.get(), counters.get() + 2,             \
              edge_hits.get(), side_face.get(), side_len.get(), big_cells, (const int32_t *)nullptr, (int32_t *)nullptr,              \
              (double *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, e0, n_edge)
        if (deal_slots <= 24) XR_DEAL_COUNT(24);
        else if (deal_slots <= 32) XR_DEAL_COUNT(32);
        else if (deal_slots <= 40) XR_DEAL_COUNT(40);
        else XR_DEAL_COUNT(48);
#undef XR_DEAL_COUNT
    };
    constexpr size_t pipe_bytes = (size_t)16 << 20; // (launches of 16 MB: smaller ones lose to their tails what the overlap gains)
    const bool piped = !edge_xy_dev && deal && !pipe_off && edge_bytes >= pipe_bytes + ((size_t)4 << 20) && !current_lane() && !stream_override();
    if (piped) {
        // fill(pinned, off, n) is called for piece k BEFORE its DMA is enqueued, i.e. right after the DMA of piece k - 1 was: the
        // count kernel of piece k - 1 is forked behind that DMA (SideScope) and runs beside the DMA of piece k
        const char *src_bytes = reinterpret_cast<const char *>(edge_xy_host);
        size_t done = 0; // bytes whose DMA has been enqueued
        auto count_upto = [&](size_t upto, bool last = false) {
            // (launches of pipe_bytes of coordinates: a launch ends with its slowest waves and at three 50 KB blocks per CU a small
            // grid is a round and a third -- 131072 edges per launch took 142 us each against 95 us for an eighth of the single
            // launch, 262144 took 265 us: what the overlap gained the tails lost.  Half a million edges per launch by default.)
            if (upto <= done || (!last && upto - done < pipe_bytes)) return;
            SideScope side;
            deal_count((int64_t)(done / 32), (int64_t)((upto - done) / 32));
            done = upto;
        };
        h2d_staged(edge_xy_own.get(), edge_bytes, [&](char *pinned, size_t off, size_t n) {
            count_upto(off);
            parallel_ranges(n, 64, [=](size_t b, size_t e) { memcpy(pinned + b, src_bytes + off + b, e - b); });
        });
        count_upto(edge_bytes, true);
        side_join();
    } else if (!edge_xy_dev) {
        h2d(edge_xy_own.get(), edge_xy_host, edge_bytes);
    }
    if (piped) {
    } else if (deal) {
        deal_count(0, n_edge);
    }
    else if (major)
    XR_LAUNCH("edges_count", k_edges_count<true>, dim3(div_up(n_edge, 256)), dim3(256), 0, edge_xy.get(), n_edge, g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
              tree->rec_face.get(), row_count.get(), big_list.get(), counters.get(), redo_list.get(), counters.get() + 2,
              edge_hits.get(), side_face.get(), side_len.get(), big_cells);
    else
    XR_LAUNCH("edges_count", k_edges_count<false>, dim3(div_up(n_edge, 256)), dim3(256), 0, edge_xy.get(), n_edge, g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
              tree->rec_face.get(), row_count.get(), big_list.get(), counters.get(), redo_list.get(), counters.get() + 2,
              edge_hits.get(), side_face.get(), side_len.get(), big_cells);
    // the wave-per-edge count pass keeps its hits in a pool (16 B each; 4 per edge of the network, at least 1M): the fill pass
    // replays them instead of walking the long edges again (XR_EDGE_POOL=0: both passes walk, as before)
    // (XR_EDGE_POOL = n > 0: a pool of n hits -- test hook for the refusal path)
    const int pool_env = (int)option(OPT_EDGE_POOL);
    const bool pooled = pool_env != 0;
    const int pool_cap = !pooled ? 1 : pool_env > 0 ? pool_env : (int)std::min<int64_t>(std::max<int64_t>(4 * n_edge, (int64_t)1 << 20), (int64_t)1 << 28);
    DevBuf<EdgeHit> pool((size_t)pool_cap);
    DevBuf<int32_t> walk_list((size_t)(pooled ? n_edge : 1));
    if (pooled)
    XR_LAUNCH("edges_big_count", k_edges_big_pool, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
              tree->rec_face.get(), row_count.get(), big_list.get(), counters.get(), pool.get(), counters.get() + 4, pool_cap,
              walk_list.get(), counters.get() + 6);
    else
    XR_LAUNCH("edges_big_count", k_edges_big<false>, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
              tree->rec_face.get(), row_count.get(), (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr,
              big_list.get(), counters.get());
    exclusive_scan_i32(row_count.get(), csr->indptr.get(), F);
    const int32_t P = read_scalar(csr->indptr.get() + F);
    XR_REQUIRE(P >= 0, XR_ERR_LIMIT, "nnz exceeds the int32 range");
    csr->nnz = P;
    csr->indices.alloc((size_t)P);
    csr->data.alloc((size_t)P);
    if (P == 0) return;
    XR_HIP(hipMemsetAsync(row_count.get(), 0, sizeof(int32_t) * (size_t)F, st));
    XR_LAUNCH("edges_replay", k_edges_replay, dim3(div_up(n_edge, 256)), dim3(256), 0, n_edge, edge_hits.get(),
              side_face.get(), side_len.get(), row_count.get(), csr->indptr.get(), csr->indices.get(), csr->data.get());
#define XR_DEAL_FILL(P)                                                                                                               \
    XR_LAUNCH("edges_redo", (k_edges_deal<true, P>), dim3((unsigned)std::min<int64_t>(div_up(n_edge, 256), engine().num_cu * 8)),     \
              dim3(256), 0, edge_xy.get(), n_edge, g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(),                \
              tree->rec_len.get(), tree->record_off(), tree->m, tree->rec_face.get(), row_count.get(), (int32_t *)nullptr,            \
              (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr,  \
              big_cells, csr->indptr.get(), csr->indices.get(), csr->data.get(), redo_list.get(), counters.get() + 2)
    if (deal) {
        if (deal_slots <= 24) XR_DEAL_FILL(24);
        else if (deal_slots <= 32) XR_DEAL_FILL(32);
        else if (deal_slots <= 40) XR_DEAL_FILL(40);
        else XR_DEAL_FILL(48);
    }
#undef XR_DEAL_FILL
    else if (major)
    XR_LAUNCH("edges_redo", k_edges_redo<true>, dim3((unsigned)std::min<int64_t>(div_up(n_edge, 256), engine().num_cu * 8)),
              dim3(256), 0, edge_xy.get(), g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(),
              tree->rec_len.get(), tree->record_off(), tree->m, tree->rec_face.get(), row_count.get(), csr->indptr.get(), csr->indices.get(),
              csr->data.get(), redo_list.get(), counters.get() + 2);
    else
    XR_LAUNCH("edges_redo", k_edges_redo<false>, dim3((unsigned)std::min<int64_t>(div_up(n_edge, 256), engine().num_cu * 8)),
              dim3(256), 0, edge_xy.get(), g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(),
              tree->rec_len.get(), tree->record_off(), tree->m, tree->rec_face.get(), row_count.get(), csr->indptr.get(), csr->indices.get(),
              csr->data.get(), redo_list.get(), counters.get() + 2);
    if (pooled) {
        XR_LAUNCH("edges_pool_replay", k_edges_pool_replay, dim3(engine().num_cu * 8), dim3(256), 0, pool.get(), counters.get() + 4,
                  pool_cap, row_count.get(), csr->indptr.get(), csr->indices.get(), csr->data.get());
        XR_LAUNCH("edges_big_fill", k_edges_big<true>, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
                  tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
                  tree->rec_face.get(), row_count.get(), csr->indptr.get(), csr->indices.get(), csr->data.get(),
                  walk_list.get(), counters.get() + 6);
    } else
    XR_LAUNCH("edges_big_fill", k_edges_big<true>, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m,
              tree->rec_face.get(), row_count.get(), csr->indptr.get(), csr->indices.get(), csr->data.get(),
              big_list.get(), counters.get());
    DevBuf<int32_t> sort_list((size_t)(P / ROW_SORT_SMALL + 1));
    csr->long_rows.alloc((size_t)(P / XR_APPLY_LONG_ROW + 1));
    csr->n_long.alloc(1);
    XR_HIP(hipMemsetAsync(csr->n_long.get(), 0, sizeof(int32_t), st));
    XR_LAUNCH("edge_rows_sort", k_edge_rows_sort, dim3(div_up(F, 256)), dim3(256), 0, csr->indptr.get(), F,
              csr->indices.get(), csr->data.get(), sort_list.get(), counters.get() + 1, csr->long_rows.get(),
              csr->n_long.get());
    int32_t h[2];
    d2h(h, counters.get(), sizeof(h));
    csr->has_long = read_scalar(csr->n_long.get()) > 0;
    if (h[1] > 0) {
        DevBuf<int32_t> tmp_idx((size_t)P);
        DevBuf<double> tmp_val((size_t)P);
        XR_LAUNCH("edge_rows_sort_big", k_edge_rows_sort_big, dim3(std::min<int>(h[1], engine().num_cu * 4)), dim3(256),
                  0, csr->indptr.get(), csr->indices.get(), csr->data.get(), sort_list.get(), counters.get() + 1,
                  tmp_idx.get(), tmp_val.get());
        stream_sync();
    }
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
