// xr_locate.hip -- point location and generalized barycentric weights on the device.
//
// Replaces numba_celltree.CellTree2d.locate_points (called from
// xugrid/regrid/unstructured.py:139,189 and xugrid/ugrid/ugridbase.py:1323) and
// CellTree2d.compute_barycentric_weights (xugrid/ugrid/ugrid2d.py:1054-1078).
// One thread per query point walks the hierarchical grid of the mesh; the arithmetic mirrors
// oracle/xr_oracle.c (point_in_poly_or_on_edge / bary_weights) operation for operation.
// A point matched by several faces (within tolerance of a shared edge) gets the LOWEST face id.
#include <memory>

#include "xr_objects.h"
#include "xr_point_in_face.h"

// query points of a barycentric construction + "inside the source grid" flags, resident in HBM (xr_locate_flags_begin)
struct xr_points {
    int64_t n = 0;
    xr_mesh *source = nullptr;
    xr::PointsBuf pts; // (the query mesh's centroids are shared with the mesh, not copied)
    xr::DevBuf<uint8_t> inside;
    // The kernels that fill the two buffers (centroids of the query, point location in the source grid) go to the engine's
    // SIDE stream -- the high-priority stream the big faces of xr_overlap use, known to run beside the main one -- as soon as
    // the handle is made; the construction that consumes the handle joins the side stream first.  They then run beside the
    // latency-bound kernels of the Voronoi pre-step instead of in front of them.  (Versions that did not survive: a stream
    // per handle -- creating and destroying a HIP stream costs 1-3 ms; one extra plain stream -- whether it really runs
    // beside the engine's depends on how the runtime maps streams onto its few hardware queues: it did in a small script,
    // not in bench.py; launching them on the main stream right before the host computes the boundary cells -- that host
    // part turned out too short to hide anything, the gain had come from kernel-beside-kernel.)
    xr_mesh *query = nullptr;
    double tol_source = 0.0;
    bool on_side = false; // launched on the side stream: join before use
    bool pts_marked = false; // the engine's aux_event was recorded behind the kernel that fills `pts` (before locate_flag)
    // round 5: the kernels are not launched when the handle is made but when somebody flushes the pending handles -- the Voronoi
    // pre-step at the point where its host round trips begin (the device idles there for ~0.2 ms; launched at the start the
    // locate pass only shared the vector ALUs with the pre-step's kernels, and the first of those that needs LDS waited for
    // it to drain), or the construction that consumes the handle (a cached tessellation: no pre-step in between)
    bool deferred = false;
    // round 6: the flags are not computed here at all but by the construction that consumes the handle, from the faces around
    // each point's Voronoi cell (k_star_flag); the handle then only carries the points
    bool flags_pending = false;
};


namespace xr {

// crossing-number test + "strictly within tol of an edge's line, projection on the segment" (xr_point_in_face.h)
__device__ __forceinline__ bool point_in_face(const double *__restrict__ poly, int n, P2 p, double tol) {
    return point_in_face_impl([&](int i) { return load_p2(poly, i); }, n, p, tol);
}

// Level split (as in xr_overlap.hip:k_search): the records of the upper grid levels -- the hull slivers of a Delaunay
// mesh, a few hundred of a million faces on ~10 all but empty levels -- are the TAIL of the record arrays.  When that tail
// is short every block filters it once against the bounding box of its 256 points into an LDS list, and the points walk
// only the levels below: each skipped level was two dependent cell_start -> record round trips per point.
static constexpr int LOC_BIG_MAX = 1024;  // upper-level records handled as a list
static constexpr int LOC_BIG_BLOCK = 64;  // ... of which at most this many may touch one block's bounding box
// The walk only PARKS the records whose f32 box holds the point (LDS, [slot][thread]); the exact tests -- f64 box, crossing
// number with a division, on-edge test with a square root: ~300 instructions per triangle -- run afterwards in a loop all
// lanes of a wave step through together.  Run from inside the walk, every lane that found a candidate made the whole wave
// execute them: 6700 VALU instructions per wave (PMC), 9x the polygon clip.
static constexpr int LOC_CAND = 8;        // parked candidates per point; further ones are tested on the spot

struct LocateBig { // per block, in LDS
    int32_t cand[LOC_CAND][256];
    unsigned long long best[256];       // per point: (face id << 32) | record of the lowest matching face so far; ~0: none
    uint16_t owner[4][64 * LOC_CAND];   // per wave: the parked candidates of its lanes as one list, lane | slot << 8
    float4 bb[LOC_BIG_BLOCK];
    int32_t rec[LOC_BIG_BLOCK];
    float box[4][4];
    int32_t n;
};

// Block-cooperative (call with all 256 threads, `valid` = the thread has a point).  -> number of levels to walk;
// sh.n = length of the block's list of upper-level records
__device__ int locate_prepare(LocateBig &sh, const GridParams &g, const int32_t *__restrict__ cell_start,
                              const float *__restrict__ rec_bb, int64_t n_tree, P2 p, bool valid, double tol) {
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    int l_split = g.n_levels, big0 = (int)n_tree;
    for (int l = g.n_levels - 1; l >= 1; l--) { // uniform: scalar loads
        const int first = cell_start[g.base[l]];
        if ((int)n_tree - first > LOC_BIG_MAX) break;
        l_split = l;
        big0 = first;
    }
    if (threadIdx.x == 0) sh.n = 0;
    if (big0 < (int)n_tree) {
        float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
        if (valid) {
            bx0 = f32_below(p.x - tol - g.x0), bx1 = f32_above(p.x + tol - g.x0);
            by0 = f32_below(p.y - tol - g.y0), by1 = f32_above(p.y + tol - g.y0);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, d, 64));
            bx1 = fmaxf(bx1, __shfl_xor(bx1, d, 64));
            by0 = fminf(by0, __shfl_xor(by0, d, 64));
            by1 = fmaxf(by1, __shfl_xor(by1, d, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            sh.box[w][0] = bx0; sh.box[w][1] = bx1; sh.box[w][2] = by0; sh.box[w][3] = by1;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++) {
            bx0 = fminf(bx0, sh.box[w][0]); bx1 = fmaxf(bx1, sh.box[w][1]);
            by0 = fminf(by0, sh.box[w][2]); by1 = fmaxf(by1, sh.box[w][3]);
        }
        for (int r = big0 + threadIdx.x; r < (int)n_tree; r += 256) {
            const float4 b = rbb[r];
            if (box_gap(b, bx0, bx1, by0, by1) <= 0.0f) {
                const int k = atomicAdd(&sh.n, 1);
                if (k < LOC_BIG_BLOCK) {
                    sh.rec[k] = r;
                    sh.bb[k] = b;
                }
            }
        }
        __syncthreads();
        if (sh.n > LOC_BIG_BLOCK) { // too many for the list: this block walks every level
            l_split = g.n_levels;
            __syncthreads();
            if (threadIdx.x == 0) sh.n = 0;
        }
    }
    __syncthreads();
    return l_split;
}

// record r passed the f32 bbox test of point q: exact bbox, then the polygon test; a match lowers *best = (face id << 32) | record
__device__ __forceinline__ void consider_record(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
                                                const int32_t *__restrict__ rec_off, int m, const int32_t *__restrict__ rec_face, int r,
                                                P2 q, double tol, unsigned long long *best) {
    const int f = rec_face[r];
    if ((uint32_t)(*best >> 32) < (uint32_t)f) return; // a lower face already holds the point
    const double *poly = rec_fxy + 2 * face_vertex_base(rec_off, r, m);
    const int n = rec_len[r];
    // exact bbox of the face (the same min/max the prepare kernel stored)
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int j = 0; j < n; j++) {
        const P2 v = load_p2(poly, j);
        xmin = fmin(xmin, v.x);
        xmax = fmax(xmax, v.x);
        ymin = fmin(ymin, v.y);
        ymax = fmax(ymax, v.y);
    }
    if (q.x < xmin - tol || q.x > xmax + tol || q.y < ymin - tol || q.y > ymax + tol) return;
    if (point_in_face(poly, n, q, tol))
        atomicMin(best, ((unsigned long long)(uint32_t)f << 32) | (unsigned long long)(uint32_t)r);
}

// The plain walk -- every record whose f32 box holds the point is tested on the spot -- for the rare point with more
// candidates than LOC_CAND parking slots (stacked or sliver faces).  Kept apart from the fast path below, which then holds ONE copy
// of the exact test in its loop (a called function would cost scratch and 50 registers: 3 instead of 5 waves per SIMD).
__device__ __forceinline__ void locate_point_slow(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
                                               const int32_t *__restrict__ rec_off, int m, const GridParams &g,
                                               const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
                                               const int32_t *__restrict__ rec_face, P2 p, double tol, LocateBig &big, int l_split,
                                               unsigned long long *best) {
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const float qx0 = f32_below(p.x - tol - g.x0), qx1 = f32_above(p.x + tol - g.x0);
    const float qy0 = f32_below(p.y - tol - g.y0), qy1 = f32_above(p.y + tol - g.y0);
    for (int l = 0; l < l_split; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
        const int cx0 = cell_coord(p.x - tol - h, g.x0, inv_h, nx), cx1 = cell_coord(p.x + tol, g.x0, inv_h, nx);
        const int cy0 = cell_coord(p.y - tol - h, g.y0, inv_h, ny), cy1 = cell_coord(p.y + tol, g.y0, inv_h, ny);
        for (int cy = cy0; cy <= cy1; cy++) {
            const int r0 = cell_start[base + cy * nx + cx0], r1 = cell_start[base + cy * nx + cx1 + 1];
            for (int r = r0; r < r1; r++)
                if (box_gap(rbb[r], qx0, qx1, qy0, qy1) <= 0.0f) consider_record(rec_fxy, rec_len, rec_off, m, rec_face, r, p, tol, best);
        }
    }
    const int nb = big.n;
    for (int k = 0; k < nb; k++)
        if (box_gap(big.bb[k], qx0, qx1, qy0, qy1) <= 0.0f) consider_record(rec_fxy, rec_len, rec_off, m, rec_face, big.rec[k], p, tol, best);
}

// -> record index of the matching face with the LOWEST caller face id, or -1.  BLOCK-cooperative: call with all 256
// threads (`valid` = the thread has a point).
// The walk PARKS the records whose f32 box holds the point; the exact tests are then dealt out evenly over the lanes of the
// wave.  Per point the walk parks ~2 candidates (triangles) but the fullest lane of a wave has 4-5, and a loop over "my
// candidates" makes all 64 lanes step through that many exact tests -- the kernels are VALU-bound on them (PMC: 2200-3500
// vector instructions per wave, vector ALUs busy 57-63 %).  Instead every wave lists the parked candidates of its lanes
// in LDS, lane i takes items i, i + 64, ... (the point comes from its owner by a cross-lane read), and a match lowers the
// owner's (face id, record) word by an LDS atomic min.
// Cells: the level-0 cells of the point's box, once; the level-l cell is that >> l exactly and the cell of x - h_l is one
// less (as in k_search, xr_overlap.hip): no per-level floating-point cell arithmetic.
__device__ int locate_point(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m,
                            const GridParams &g, const int32_t *__restrict__ cell_start,
                            const float *__restrict__ rec_bb, const int32_t *__restrict__ rec_face, P2 p, bool valid, double tol,
                            LocateBig &big, int l_split) {
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float qx0 = f32_below(p.x - tol - g.x0), qx1 = f32_above(p.x + tol - g.x0);
    const float qy0 = f32_below(p.y - tol - g.y0), qy1 = f32_above(p.y + tol - g.y0);
    big.best[tid] = ~0ull;
    int nc = 0;
    auto park = [&](int r) {
        big.cand[nc < LOC_CAND ? nc : LOC_CAND - 1][tid] = r; // (beyond the slots: the point takes the plain walk below)
        nc++;
    };
    const int c_x0 = cell_coord(p.x - tol, g.x0, g.inv_h0, g.nx[0]), c_x1 = cell_coord(p.x + tol, g.x0, g.inv_h0, g.nx[0]);
    const int c_y0 = cell_coord(p.y - tol, g.y0, g.inv_h0, g.ny[0]), c_y1 = cell_coord(p.y + tol, g.y0, g.inv_h0, g.ny[0]);
    for (int l = 0; l < l_split && valid; l++) {
        const int nx = g.nx[l], base = g.base[l];
        const int sh = l * LEVEL_SHIFT;
        const int cx0 = max((c_x0 >> sh) - 1, 0), cx1 = c_x1 >> sh;
        const int cy0 = max((c_y0 >> sh) - 1, 0), cy1 = c_y1 >> sh;
        for (int cy = cy0; cy <= cy1; cy++) {
            const int r0 = cell_start[base + cy * nx + cx0];
            const int r1 = cell_start[base + cy * nx + cx1 + 1];
            for (int r = r0; r < r1; r += 4) {
                // four independent 16-byte loads in flight per step (indices clamped; the range guard is folded into the gap)
                const int last = r1 - 1;
                const float4 b0 = rbb[r];
                const float4 b1 = rbb[r + 1 <= last ? r + 1 : last];
                const float4 b2 = rbb[r + 2 <= last ? r + 2 : last];
                const float4 b3 = rbb[r + 3 <= last ? r + 3 : last];
                const float g0 = box_gap(b0, qx0, qx1, qy0, qy1);
                const float g1 = fmaxf(box_gap(b1, qx0, qx1, qy0, qy1), r + 1 <= last ? -INFINITY : 1.0f);
                const float g2 = fmaxf(box_gap(b2, qx0, qx1, qy0, qy1), r + 2 <= last ? -INFINITY : 1.0f);
                const float g3 = fmaxf(box_gap(b3, qx0, qx1, qy0, qy1), r + 3 <= last ? -INFINITY : 1.0f);
                if (g0 <= 0.0f) park(r);
                if (g1 <= 0.0f) park(r + 1);
                if (g2 <= 0.0f) park(r + 2);
                if (g3 <= 0.0f) park(r + 3);
            }
        }
    }
    const int nb = valid ? big.n : 0;
    for (int k = 0; k < nb; k++) { // the upper-level records that touch this block (from LDS)
        const float4 b = big.bb[k];
        if (box_gap(b, qx0, qx1, qy0, qy1) <= 0.0f) park(big.rec[k]);
    }
    // the wave's list of parked candidates
    const bool overflow = nc > LOC_CAND;
    const int parked = overflow ? 0 : nc;
    int incl = parked;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    const int total = __shfl(incl, 63, 64), excl = incl - parked;
    for (int k = 0; k < parked; k++) big.owner[wv][excl + k] = (uint16_t)(lane | (k << 8));
    __syncthreads(); // (the list, the candidates and every point's `best` word are in LDS)
    for (int first = 0; first < total; first += 64) { // (wave-uniform)
        const int item = first + lane;
        const bool has = item < total;
        const int ow = has ? big.owner[wv][item] : 0;
        const int ol = ow & 63, slot = ow >> 8;
        const P2 q{__shfl(p.x, ol, 64), __shfl(p.y, ol, 64)};
        if (has) consider_record(rec_fxy, rec_len, rec_off, m, rec_face, big.cand[slot][wv * 64 + ol], q, tol, &big.best[wv * 64 + ol]);
    }
    if (overflow) locate_point_slow(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, tol, big, l_split, &big.best[tid]);
    __syncthreads();
    const unsigned long long b = big.best[tid];
    return b == ~0ull ? -1 : (int)(uint32_t)b;
}

// (the locate kernels wait on memory ~70 % of their wave cycles: eight waves per SIMD -- 63 registers and a few bytes of scratch
// instead of 65-72 registers and seven waves -- buy 6 % on the 4M-point barycentric kernel, 0.528 -> 0.497 ms)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_locate(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m, GridParams g,
         const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
         const int32_t *__restrict__ rec_face, int64_t n_tree, const double *__restrict__ pts, int64_t n, double tol,
         int64_t *__restrict__ out) {
    __shared__ LocateBig sh_big;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const P2 p = valid ? load_p2(pts, (int)i) : P2{0.0, 0.0};
    const int l_split = locate_prepare(sh_big, g, cell_start, rec_bb, n_tree, p, valid, tol);
    const int r = locate_point(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, valid, tol, sh_big, l_split);
    if (!valid) return;
    out[i] = r >= 0 ? rec_face[r] : -1;
}

// Wachspress coordinates with the on-edge special case; triangles use plain area coordinates.
// w points at this point's row of the (n, m) weight table (global memory, zero-initialised);
// weights are aligned with the CCW-normalised vertex order of the face (as numba_celltree's).
// -> the number of weights > 0 among the n it wrote (what the masking of unstructured.py:188-191 keeps for a point
// inside the source grid whose cell has no substitute vertex)
__device__ int bary_weights(const double *__restrict__ poly, int n, P2 p, double tol, double *__restrict__ w,
                            int64_t ws = 1) {
    // (vertices roll through registers and indices wrap by a compare: `% n` with a run-time n is a ~40-instruction integer
    // division, and the loops below had five of them per vertex)
    // pass 1: on-edge detection
    {
        P2 v0 = load_p2(poly, 0);
        for (int i = 0; i < n; i++) {
            const int in = i + 1 == n ? 0 : i + 1;
            const P2 v1 = load_p2(poly, in);
            const double wx = v1.x - v0.x, wy = v1.y - v0.y;
            const double ux = p.x - v0.x, uy = p.y - v0.y;
            const double a = wx * uy - wy * ux;
            const double len2 = wx * wx + wy * wy;
            if (len2 > 0 && !edge_certainly_far(fabs(a), len2, tol)) { // (xr_point_in_face.h: the square root only where it decides)
                const double len = sqrt(len2);
                if (fabs(a) < tol * len) {
                    const double tpar = ux * wx + uy * wy;
                    if (tpar >= 0 && tpar <= len2) {
                        double tt = tpar / len2;
                        if (tt < 0) tt = 0;
                        if (tt > 1) tt = 1;
                        for (int j = 0; j < n; j++) w[(j) * ws] = j == i ? 1.0 - tt : j == in ? tt : 0.0;
                        return (1.0 - tt > 0 ? 1 : 0) + (tt > 0 ? 1 : 0);
                    }
                }
            }
            v0 = v1;
        }
    }
    auto cross_at = [&](P2 v0, P2 v1) { // cross(v_i - p, v_{i+1} - p) in the oracle's form
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        const double ux = p.x - v0.x, uy = p.y - v0.y;
        return wx * uy - wy * ux;
    };
    if (n == 3) {
        const P2 v0 = load_p2(poly, 0), v1 = load_p2(poly, 1), v2 = load_p2(poly, 2);
        const double a0 = cross_at(v0, v1), a1 = cross_at(v1, v2), a2 = cross_at(v2, v0);
        const double s = a0 + a1 + a2;
        const double w0 = a1 / s, w1 = a2 / s, w2 = a0 / s;
        w[(0) * ws] = w0;
        w[(1) * ws] = w1;
        w[(2) * ws] = w2;
        return (w0 > 0 ? 1 : 0) + (w1 > 0 ? 1 : 0) + (w2 > 0 ? 1 : 0);
    }
    double wsum = 0.0;
    P2 vp = load_p2(poly, n - 1), vi = load_p2(poly, 0);
    double a_prev = cross_at(vp, vi); // A(n - 1)
    for (int i = 0; i < n; i++) {
        const P2 vn = load_p2(poly, i + 1 == n ? 0 : i + 1);
        const double a_i = cross_at(vi, vn);
        const double cx = (vi.x - vp.x) * (vn.y - vi.y) - (vi.y - vp.y) * (vn.x - vi.x);
        const double wi = cx / (a_prev * a_i);
        w[(i) * ws] = wi;
        wsum += wi;
        vp = vi;
        vi = vn;
        a_prev = a_i;
    }
    int n_pos = 0;
    for (int i = 0; i < n; i++) {
        const double wi = w[(i) * ws] / wsum;
        w[(i) * ws] = wi;
        n_pos += wi > 0 ? 1 : 0;
    }
    return n_pos;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_barycentric(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m, GridParams g,
              const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
              const int32_t *__restrict__ rec_face, int64_t n_tree, const double *__restrict__ pts, int64_t n, double tol,
              int64_t *__restrict__ face_out, double *__restrict__ weights) {
    __shared__ LocateBig sh_big;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const P2 p = valid ? load_p2(pts, (int)i) : P2{0.0, 0.0};
    const int l_split = locate_prepare(sh_big, g, cell_start, rec_bb, n_tree, p, valid, tol);
    const int r = locate_point(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, valid, tol, sh_big, l_split);
    if (!valid) return;
    face_out[i] = r >= 0 ? rec_face[r] : -1;
    double *w = weights + i * m;
    for (int j = 0; j < m; j++) w[j] = 0.0;
    if (r >= 0) bary_weights(rec_fxy + 2 * face_vertex_base(rec_off, r, m), rec_len[r], p, tol, w);
}

static constexpr uint8_t NPOS_FIX = 255;

// ---- BarycentricInterpolator weights assembled on the device (xr_barycentric_csr) ----------------
// The (n, m) weight table of this pipeline is kept COLUMN-major (weight j of point i at [j * n + i]): the lanes of a
// wave are neighbouring points, so every access is coalesced, and only the first len(cell) slots of a point are ever
// touched (m is the largest cell of the tessellation, 15-25 corners at the hull; the typical cell has 6).

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_barycentric_cm(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m, GridParams g,
                 const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
                 const int32_t *__restrict__ rec_face, int64_t n_tree, const double *__restrict__ pts, int64_t n, double tol,
                 int64_t *__restrict__ face_out, double *__restrict__ weights,
                 const uint8_t *__restrict__ cell_flag /* cells with a substitute vertex: their weights are rewritten later */,
                 uint8_t *__restrict__ n_pos /* weights > 0 of the point; NPOS_FIX: read the weights (flagged cell) */) {
    __shared__ LocateBig sh_big;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const P2 p = valid ? load_p2(pts, (int)i) : P2{0.0, 0.0};
    const int l_split = locate_prepare(sh_big, g, cell_start, rec_bb, n_tree, p, valid, tol);
    const int r = locate_point(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, valid, tol, sh_big, l_split);
    if (!valid) return;
    const int cell = r >= 0 ? rec_face[r] : -1;
    face_out[i] = cell;
    if (r < 0) {
        n_pos[i] = 0;
        return;
    }
    double *w = weights + i;
    const int len = rec_len[r];
    // (bary_weights writes every one of the cell's len slots; nothing beyond them is ever read)
    const int c = bary_weights(rec_fxy + 2 * face_vertex_base(rec_off, r, m), len, p, tol, w, n);
    n_pos[i] = cell_flag[cell] ? NPOS_FIX : (uint8_t)c;
}

// cells of the tessellation that hold a substitute vertex (id >= threshold): only their points go through the weight
// replacement of bary_fix_count; for all others the barycentric kernel already knows how many weights are positive
__global__ void __launch_bounds__(256)
k_bary_cell_flag(const int32_t *__restrict__ faces_raw, int64_t n_cell, int m, int64_t threshold, uint8_t *__restrict__ flag) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cell) return;
    const int32_t *face = faces_raw + c * m;
    bool any = false;
    for (int j = 0; j < m && face[j] >= 0; j++) any = any || face[j] >= threshold;
    flag[c] = any;
}

// OPEN_ONLY: only the points whose flag is STAR_OPEN (left open by k_star_flag) are located; a block without one leaves at once
static constexpr uint8_t STAR_OPEN = 2;
template <bool OPEN_ONLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_locate_flag(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m, GridParams g,
              const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
              const int32_t *__restrict__ rec_face, int64_t n_tree, const double *__restrict__ pts, int64_t n, double tol,
              uint8_t *__restrict__ inside) {
    __shared__ LocateBig sh_big;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool valid = i < n;
    if (OPEN_ONLY) {
        valid = valid && inside[i] == STAR_OPEN;
        if (!__syncthreads_or(valid)) return;
    }
    const P2 p = valid ? load_p2(pts, (int)i) : P2{0.0, 0.0};
    const int l_split = locate_prepare(sh_big, g, cell_start, rec_bb, n_tree, p, valid, tol);
    const int r = locate_point(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, valid, tol, sh_big, l_split);
    if (valid) inside[i] = r >= 0;
}

// "Inside the source grid" (unstructured.py:188-190: grid.locate_points(points) == -1 with the default tolerance) WITHOUT a walk
// through the source grid's index (round 6).  The vertices of the Voronoi cell that holds the point are the centroids of the source
// faces around the cell's node (vertex id < n_identity: the face itself; behind that: vertex_face) -- the faces whose union the
// cell lies in.  The point is tested against exactly those faces with the locate kernels' own test (xr_point_in_face.h, on the
// face's counter-clockwise-normalised vertices as the tree holds them: the same directed edges, the same roundings): a hit means
// locate_points finds A face, i.e. != -1, which is all the reference asks (:189).  No hit (a point in a concave exterior cell beyond
// the hull, a degenerate star) leaves the flag OPEN and k_locate_flag<true> walks the grid for those points alone.  Per point ~6
// triangle tests instead of a grid walk + ~18 box tests + the exact tests of the parked records: 1M faces / 4M points,
// locate_flag 284 us (175 us of vector instructions, PMC) -> see DESIGN.
// The faces' vertices come from the source mesh's own face-major block (fxy: counter-clockwise-normalised, caller's face order --
// what the tree's records hold, so the test sees the same directed edges and roundings), TWO faces per round with all their loads
// in flight together: as a chain cell row -> connectivity -> nodes per face the kernel took 295 us for 4M points.
template <int MS> // nodes per face of a dense source mesh (3, 4); 0: any mesh, one face at a time
__global__ void __launch_bounds__(256)
k_star_flag(const int64_t *__restrict__ cell_of_point, const int32_t *__restrict__ cells, int mv, int64_t n_identity,
            const int64_t *__restrict__ vertex_face, int64_t n_real /* vertices below it have a source face */,
            const double *__restrict__ src_fxy, const uint8_t *__restrict__ src_len, const int32_t *__restrict__ src_off, int ms,
            const double *__restrict__ pts, int64_t n, double tol, uint8_t *__restrict__ inside) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t cell = cell_of_point[i];
    if (cell < 0) { // (no weights either way)
        inside[i] = 0;
        return;
    }
    const P2 p = load_p2(pts, (int)i);
    const int32_t *row = cells + cell * mv;
    uint8_t flag = STAR_OPEN;
    constexpr int B = 2; // faces per round (4M points, 1M-face Delaunay source: 1 face 0.234 ms, 2: 0.132, 3: 0.150, 4: 0.158, 6: 0.201 -- registers against rounds)
    bool open = true; // (the row's fill has not been met)
    for (int j0 = 0; j0 < mv && open && flag == STAR_OPEN; j0 += B) {
        int64_t f[B];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int64_t v = j0 + u < mv ? row[j0 + u] : -1;
            open = open && v >= 0;
            f[u] = (!open || v >= n_real) ? -1 : v; // (a substitute vertex has no face of its own)
        }
#pragma unroll
        for (int u = 0; u < B; u++)
            if (f[u] >= n_identity) f[u] = vertex_face[f[u]];
        if constexpr (MS > 0) {
            const double2 *fx = reinterpret_cast<const double2 *>(src_fxy);
            double2 vtx[B][MS];
            int len[B];
#pragma unroll
            for (int u = 0; u < B; u++) {
                const int64_t ff = f[u] < 0 ? 0 : f[u];
                len[u] = f[u] < 0 ? 0 : src_len[ff];
#pragma unroll
                for (int k = 0; k < MS; k++) vtx[u][k] = fx[ff * MS + k];
            }
#pragma unroll
            for (int u = 0; u < B; u++) {
                if (flag != STAR_OPEN || len[u] < 3) continue;
                const bool hit = point_in_face_regs<MS>(vtx[u], len[u], p, tol);
                if (hit) flag = 1;
            }
        } else {
            for (int u = 0; u < B; u++) {
                if (flag != STAR_OPEN || f[u] < 0) continue;
                const double *poly = src_fxy + 2 * face_vertex_base(src_off, f[u], ms);
                if (point_in_face(poly, src_len[f[u]], p, tol)) flag = 1;
            }
        }
    }
    inside[i] = flag;
}

// replace_interpolated_weights (xugrid/regrid/unstructured.py:17-57) on the point's own row, then the
// masking of unstructured.py:188-191 (points outside the source grid; weights > 0), counted per point.
// Same statement order as oracle/xr_oracle.c:xo_replace_interpolated_weights.
template <typename FI> // int64_t: a normalised table (tree order); int32_t: the mesh's own connectivity (caller's order)
__global__ void __launch_bounds__(256)
k_bary_fix_count(const int64_t *__restrict__ face_of_point, double *__restrict__ weights, int m,
                 const FI *__restrict__ faces_ccw, const double *__restrict__ vxy,
                 const int64_t *__restrict__ node_to_node_map, int64_t threshold,
                 const uint8_t *__restrict__ inside, int64_t n, int32_t *__restrict__ count,
                 const uint8_t *__restrict__ n_pos = nullptr /* optional: see k_barycentric_cm */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (n_pos && n_pos[i] != NPOS_FIX) {
        // (no substitute vertex in the point's cell: nothing to rewrite, and the count is known -- 2 bytes read per point
        // instead of its weights, vertex ids and face id: 0.157 -> see DESIGN ms for 4M points)
        count[i] = inside[i] ? n_pos[i] : 0;
        return;
    }
    const int64_t f = face_of_point[i];
    int c = 0;
    if (f >= 0 && inside[i]) { // (a point outside the source grid keeps no weight, unstructured.py:189-190)
        double *w = weights + i;
        const FI *face = faces_ccw + f * m;
        int len = 0;
        while (len < m && face[len] >= 0) len++; // (-1 fill behind the cell's corners)
        for (int j = 0; j < len; j++) {
            const int64_t pidx = face[j];
            const double wj = w[(int64_t)j * n];
            if (pidx < threshold || wj <= 0) continue;
            const int64_t index = pidx - threshold;
            const int64_t q = node_to_node_map[2 * index], r = node_to_node_map[2 * index + 1];
            const double px = vxy[2 * pidx], py = vxy[2 * pidx + 1];
            const double qx = vxy[2 * q], qy = vxy[2 * q + 1];
            const double rx = vxy[2 * r], ry = vxy[2 * r + 1];
            const double p_q = sqrt((qx - px) * (qx - px) + (qy - py) * (qy - py));
            const double p_r = sqrt((rx - px) * (rx - px) + (ry - py) * (ry - py));
            const double total = p_q + p_r;
            const double weight_q = (p_r / total) * wj;
            const double weight_r = (p_q / total) * wj;
            w[(int64_t)j * n] = 0.0;
            for (int jj = 0; jj < len; jj++) {
                if (face[jj] == q) w[(int64_t)jj * n] += weight_q;
                if (face[jj] == r) w[(int64_t)jj * n] += weight_r;
            }
        }
        for (int j = 0; j < len; j++) c += w[(int64_t)j * n] > 0;
    }
    count[i] = c;
}

// The entries of a block's 256 points are contiguous in the CSR: they are collected in LDS and written out as whole lines
// (one thread writing its own 4- and 8-byte pieces cost 1.45 GB of HBM writes for 300 MB of entries, PMC).
static constexpr int FILL_STAGE = 2048; // entries one block stages (24 KiB: six blocks per CU; the typical block holds 1600); fuller blocks write directly

template <typename FI>
__global__ void __launch_bounds__(256)
k_bary_fill(const int64_t *__restrict__ face_of_point, const double *__restrict__ weights, int m,
            const FI *__restrict__ faces_ccw, const int64_t *__restrict__ vertex_face,
            const int32_t *__restrict__ indptr, int64_t n, int32_t *__restrict__ indices,
            double *__restrict__ data, int64_t n_identity /* vertices below it ARE their face (the centroids): no table gather */,
            int64_t capacity /* entries the two arrays hold: a block whose rows end beyond it writes nothing (the host redoes the fill) */) {
    __shared__ int32_t sh_idx[FILL_STAGE];
    __shared__ double sh_val[FILL_STAGE];
    const int64_t i0 = (int64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
    const int64_t i1 = i0 + 256 < n ? i0 + 256 : n;
    if ((int64_t)indptr[i1] > capacity) return; // (uniform)
    const int base = indptr[i0], total = indptr[i1] - base;
    const bool staged = total <= FILL_STAGE;
    if (i < n) {
        int pos = indptr[i];
        if (indptr[i + 1] != pos) {
            const FI *face = faces_ccw + face_of_point[i] * m;
            const double *w = weights + i;
            auto emit = [&](int64_t vtx, double wj) {
                const int32_t col = vtx < n_identity ? (int32_t)vtx : (int32_t)vertex_face[vtx];
                if (staged) {
                    sh_idx[pos - base] = col;
                    sh_val[pos - base] = wj;
                } else {
                    indices[pos] = col;
                    data[pos] = wj;
                }
                pos++;
            };
            // the first eight slots (the typical cell has six corners) as two rounds of independent loads -- the corner ids,
            // then the weights of the corners that exist -- instead of a chain of dependent (id, weight) pairs
            constexpr int HEAD = 8;
            FI id[HEAD];
            double wv[HEAD];
#pragma unroll
            for (int j = 0; j < HEAD; j++) id[j] = j < m ? face[j] : (FI)-1;
#pragma unroll
            for (int j = 0; j < HEAD; j++) wv[j] = id[j] >= 0 ? w[(int64_t)j * n] : 0.0; // (-1 fill: nothing behind it either)
            bool open = true;
#pragma unroll
            for (int j = 0; j < HEAD; j++) {
                open = open && id[j] >= 0;
                if (open && wv[j] > 0) emit((int64_t)id[j], wv[j]);
            }
            for (int j = HEAD; open && j < m && face[j] >= 0; j++) {
                const double wj = w[(int64_t)j * n];
                if (wj > 0) emit((int64_t)face[j], wj);
            }
        }
    }
    if (!staged) return; // (uniform)
    __syncthreads();
    for (int k = threadIdx.x; k < total; k += 256) {
        indices[base + k] = sh_idx[k];
        data[base + k] = sh_val[k];
    }
}

// ---- locator weights as CSR (xr_locate_csr) -------------------------------------------------------
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_locate_col(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, const int32_t *__restrict__ rec_off, int m, GridParams g,
             const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
             const int32_t *__restrict__ rec_face, int64_t n_tree, const double *__restrict__ pts, int64_t n, double tol,
             int32_t *__restrict__ col, int32_t *__restrict__ found) {
    __shared__ LocateBig sh_big;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const P2 p = valid ? load_p2(pts, (int)i) : P2{0.0, 0.0};
    const int l_split = locate_prepare(sh_big, g, cell_start, rec_bb, n_tree, p, valid, tol);
    const int r = locate_point(rec_fxy, rec_len, rec_off, m, g, cell_start, rec_bb, rec_face, p, valid, tol, sh_big, l_split);
    if (!valid) return;
    col[i] = r >= 0 ? rec_face[r] : -1;
    found[i] = r >= 0;
}

__global__ void __launch_bounds__(256)
k_locate_fill(const int32_t *__restrict__ col, const int32_t *__restrict__ indptr, int64_t n,
              int32_t *__restrict__ indices, double *__restrict__ data) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || col[i] < 0) return;
    indices[indptr[i]] = col[i];
    data[indptr[i]] = 1.0;
}

// the nodes of a raster, row-major (y, x): point j * nx + i = (x[i], y[j])
__global__ void __launch_bounds__(256)
k_raster_points(const double *__restrict__ x, const double *__restrict__ y, int64_t nx, int64_t n,
                double *__restrict__ pts) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const int64_t j = v / nx, i = v - j * nx;
    reinterpret_cast<double2 *>(pts)[v] = make_double2(x[i], y[j]);
}

static double resolve_tolerance(xr_mesh *mesh, double tolerance) {
    if (tolerance >= 0) return tolerance;
    mesh_read_stats(mesh, /*need_exact=*/true); // (the largest bbox diagonal over ALL faces)
    return 1e-12 * mesh->h_stats[6]; // ugridbase.py:1165-1170
}

static void launch_points(xr_points *h) {
    if (h->query) h->pts.share(mesh_centroids_shared(h->query));
    if (engine().on_side && !current_lane()) {
        // the consumer needs the points long before it needs the flags: it waits for this mark first and joins the side
        // stream only in front of the kernel that reads `inside` (on a cached tessellation locate_flag then runs BESIDE the
        // barycentric kernel instead of in front of it)
        XR_HIP(hipEventRecord(engine().aux_event, engine().side));
        h->pts_marked = true;
    }
    if (h->flags_pending) return; // (the consumer fills `inside`: k_star_flag)
    xr_mesh *source = h->source;
    XR_LAUNCH("locate_flag", k_locate_flag<false>, dim3(div_up(h->n, 256)), dim3(256), 0, source->rec_fxy.get(),
              source->rec_len.get(), source->record_off(), source->m, source->grid, source->cell_start.get(), source->rec_bb.get(),
              source->rec_face.get(), source->n_face, h->pts.get(), h->n, h->tol_source, h->inside.get());
}

static std::vector<xr_points *> &pending_points() {
    static std::vector<xr_points *> list; // (exclusive entry points only: no lock)
    return list;
}
void flush_pending_points() {
    auto &list = pending_points();
    for (xr_points *h : list) {
        if (!h->deferred) continue;
        {
            SideScope side; // (forks behind everything enqueued so far)
            launch_points(h);
        }
        h->on_side = true;
        h->deferred = false;
    }
    list.clear();
}
// A mesh is about to lose its derived arrays (xr_mesh_invalidate) or to go away (xr_mesh_destroy): the deferred kernels of
// every pending handle that reads it -- the source's index, the query's centroids -- are launched NOW, while the arrays are
// still there (the pool parks blocks freed while the side stream is in flight; xr_mesh_destroy synchronises).
void flush_pending_points_of(const xr_mesh *mesh) {
    for (xr_points *h : pending_points())
        if (h->deferred && (h->source == mesh || h->query == mesh)) {
            flush_pending_points();
            return;
        }
}
static void forget_pending_points(xr_points *h) {
    auto &list = pending_points();
    for (size_t i = 0; i < list.size(); i++)
        if (list[i] == h) list[i] = list.back(), list.pop_back(), i--;
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_locate_points(xr_mesh *mesh, const double *points, int64_t n, double tolerance, int64_t *face_index_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && n >= 0 && (n == 0 || (points && face_index_out)), XR_ERR_INVALID,
               "xr_locate_points: bad arguments");
    XR_REQUIRE(n < ((int64_t)1 << 31), XR_ERR_LIMIT, "xr_locate_points: too many points");
    if (n > 0) {
        mesh_prepare(mesh, false);
        mesh_build_index(mesh);
        const double tol = resolve_tolerance(mesh, tolerance);
        DevBuf<double> pts((size_t)n * 2);
        DevBuf<int64_t> out((size_t)n);
        h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
        if (mesh->n_face > 0) {
            XR_LAUNCH("locate_points", k_locate, dim3(div_up(n, 256)), dim3(256), 0, mesh->rec_fxy.get(),
                      mesh->rec_len.get(), mesh->record_off(), mesh->m, mesh->grid, mesh->cell_start.get(), mesh->rec_bb.get(),
                      mesh->rec_face.get(), mesh->n_face, pts.get(), n, tol, out.get());
            d2h(face_index_out, out.get(), sizeof(int64_t) * (size_t)n);
            stream_sync();
        } else {
            for (int64_t i = 0; i < n; i++) face_index_out[i] = -1;
        }
    }
    XR_API_END
}

int xr_locate_raster(xr_mesh *mesh, const double *x, int64_t nx, const double *y, int64_t ny, double tolerance,
                     int64_t *face_index_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && nx >= 0 && ny >= 0 && (nx * ny == 0 || (x && y && face_index_out)), XR_ERR_INVALID,
               "xr_locate_raster: bad arguments");
    XR_REQUIRE(ny == 0 || nx < ((int64_t)1 << 31) / ny, XR_ERR_LIMIT, "xr_locate_raster: too many points");
    const int64_t n = nx * ny;
    if (n > 0) {
        if (mesh->n_face > 0) {
            mesh_prepare(mesh, false);
            mesh_build_index(mesh);
            const double tol = resolve_tolerance(mesh, tolerance);
            DevBuf<double> dx((size_t)nx), dy((size_t)ny), pts((size_t)n * 2);
            DevBuf<int64_t> out((size_t)n);
            h2d(dx.get(), x, sizeof(double) * (size_t)nx);
            h2d(dy.get(), y, sizeof(double) * (size_t)ny);
            XR_LAUNCH("raster_points", k_raster_points, dim3(div_up(n, 256)), dim3(256), 0, dx.get(), dy.get(), nx, n,
                      pts.get());
            XR_LAUNCH("locate_points", k_locate, dim3(div_up(n, 256)), dim3(256), 0, mesh->rec_fxy.get(),
                      mesh->rec_len.get(), mesh->record_off(), mesh->m, mesh->grid, mesh->cell_start.get(), mesh->rec_bb.get(),
                      mesh->rec_face.get(), mesh->n_face, pts.get(), n, tol, out.get());
            d2h(face_index_out, out.get(), sizeof(int64_t) * (size_t)n);
            stream_sync();
        } else {
            for (int64_t i = 0; i < n; i++) face_index_out[i] = -1;
        }
    }
    XR_API_END
}

int xr_barycentric(xr_mesh *mesh, const double *points, int64_t n, double tolerance, int64_t *face_index_out,
                   double *weights_out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && n >= 0 && (n == 0 || (points && face_index_out && weights_out)), XR_ERR_INVALID,
               "xr_barycentric: bad arguments");
    XR_REQUIRE(n < ((int64_t)1 << 31), XR_ERR_LIMIT, "xr_barycentric: too many points");
    if (n > 0) {
        mesh_prepare(mesh, false);
        mesh_build_index(mesh);
        const double tol = resolve_tolerance(mesh, tolerance);
        const int m = mesh->m;
        if (mesh->n_face > 0) {
            DevBuf<double> pts((size_t)n * 2), w((size_t)n * m);
            DevBuf<int64_t> out((size_t)n);
            h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
            XR_LAUNCH("barycentric", k_barycentric, dim3(div_up(n, 256)), dim3(256), 0, mesh->rec_fxy.get(),
                      mesh->rec_len.get(), mesh->record_off(), m, mesh->grid, mesh->cell_start.get(), mesh->rec_bb.get(),
                      mesh->rec_face.get(), mesh->n_face, pts.get(), n, tol, out.get(), w.get());
            d2h(face_index_out, out.get(), sizeof(int64_t) * (size_t)n);
            d2h(weights_out, w.get(), sizeof(double) * (size_t)n * m);
            stream_sync();
        } else {
            for (int64_t i = 0; i < n; i++) face_index_out[i] = -1;
            for (int64_t i = 0; i < n * m; i++) weights_out[i] = 0.0;
        }
    }
    XR_API_END
}

// vertex v of the tessellation belongs to source face v for v < n_identity (the face centroids come first) and to
// vertex_face[v - n_identity] beyond (projections: their face; substitute vertices: -1)

// the source-side part of UnstructuredGrid2d.barycentric (unstructured.py:147, 188-190) -- the query points and
// `grid.locate_points(points) == -1` -- enqueued WITHOUT a final wait: it needs nothing of the Voronoi tessellation
static void locate_flags(xr_mesh *source, xr_mesh *query, const double *points, int64_t n, PointsBuf &pts,
                         DevBuf<uint8_t> &inside, bool points_only = false) {
    mesh_prepare(source, false);
    mesh_build_index(source);
    const double tol_source = resolve_tolerance(source, -1.0); // unstructured.py:189: default tolerance
    inside.alloc((size_t)n);
    if (query) {
        pts.share(mesh_centroids_shared(query));
    } else {
        pts.alloc((size_t)n * 2);
        h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
    }
    if (points_only) return; // (the flags come from k_star_flag)
    XR_LAUNCH("locate_flag", k_locate_flag<false>, dim3(div_up(n, 256)), dim3(256), 0, source->rec_fxy.get(),
              source->rec_len.get(), source->record_off(), source->m, source->grid, source->cell_start.get(), source->rec_bb.get(),
              source->rec_face.get(), source->n_face, pts.get(), n, tol_source, inside.get());
}

static void barycentric_csr(xr_mesh *voronoi, xr_mesh *source, xr_mesh *query, const double *points, int64_t n,
                            double tolerance, int64_t n_identity, const int64_t *vertex_face,
                            const int64_t *node_to_node_map, int64_t n_extra, bool reference_order, xr_csr **out,
                            xr_points *pre = nullptr) {
    {
    XR_REQUIRE(voronoi && source && out, XR_ERR_INVALID, "xr_barycentric_csr: NULL argument");
    if (pre) {
        if (pre->deferred) flush_pending_points(); // (nobody launched the source-side kernels yet: a cached tessellation)
        XR_REQUIRE(pre->source == source && !query && !points, XR_ERR_INVALID,
                   "xr_barycentric_csr: the prepared points belong to another source grid (or points were given twice)");
        n = pre->n;
    }
    XR_REQUIRE(pre || ((query != nullptr) != (points != nullptr)), XR_ERR_INVALID,
               "xr_barycentric_csr: give either a query mesh (its face centroids are the points) or points");
    XR_REQUIRE(n_identity >= 0 && n_identity <= voronoi->n_node && n_identity <= source->n_face &&
                   (vertex_face || n_identity == voronoi->n_node),
               XR_ERR_INVALID, "xr_barycentric_csr: bad vertex_face");
    if (query) n = query->n_face;
    XR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) - 1, XR_ERR_LIMIT, "xr_barycentric_csr: too many points");
    XR_REQUIRE(n_extra >= 0 && n_extra <= voronoi->n_node && (n_extra == 0 || node_to_node_map), XR_ERR_INVALID,
               "xr_barycentric_csr: bad node_to_node_map");
    const int64_t nv = voronoi->n_node;
    for (int64_t i = 0; i < 2 * n_extra; i++)
        XR_REQUIRE(node_to_node_map[i] >= 0 && node_to_node_map[i] < nv, XR_ERR_INVALID,
                   "xr_barycentric_csr: node_to_node_map entry %lld outside [0,%lld)", (long long)i, (long long)nv);
    XR_REQUIRE(n_identity <= nv - n_extra, XR_ERR_INVALID, "xr_barycentric_csr: more substitute vertices than vertices");
    for (int64_t i = n_identity; i < nv - n_extra; i++)
        XR_REQUIRE(vertex_face[i - n_identity] >= 0 && vertex_face[i - n_identity] < source->n_face, XR_ERR_INVALID,
                   "xr_barycentric_csr: vertex_face[%lld] outside the source grid", (long long)i);
    const int m = voronoi->m;
    xr_csr *csr = new xr_csr();
    try {
        csr->n = n; csr->m = source->n_face; csr->nnz = 0;
        csr->indptr.alloc((size_t)n + 1);
        if (n == 0 || voronoi->n_face == 0 || source->n_face == 0) {
            fill_i32(csr->indptr.get(), 0, n + 1);
            csr->indices.alloc(0);
            csr->data.alloc(0);
            stream_sync();
        } else {
            mesh_prepare(voronoi, false); mesh_build_index(voronoi);
            const double tol = resolve_tolerance(voronoi, tolerance);
            PointsBuf own_pts;
            DevBuf<double> w((size_t)n * m);
            DevBuf<uint8_t> own_inside;
            // the flags from the faces around each point's cell (k_star_flag, behind the barycentric kernel) unless the handle
            // brings them (option "star_flag" = 0, or a handle made while it was)
            const bool star = pre ? pre->flags_pending : option(OPT_STAR_FLAG) != 0;
            if (!pre) locate_flags(source, query, points, n, own_pts, own_inside, star);
            bool join_later = false;
            if (pre && pre->on_side) { // (filled on the side stream)
                if (pre->pts_marked) { // the points now, the flags in front of bary_fix_count
                    XR_HIP(hipStreamWaitEvent(engine().stream, engine().aux_event, 0));
                    join_later = true;
                } else {
                    side_join();
                    pre->on_side = false;
                }
            }
            PointsBuf &pts = pre ? pre->pts : own_pts;
            DevBuf<uint8_t> &inside = pre ? pre->inside : own_inside;
            // the vertex table the weight slots are paired with: the caller's order as the reference does
            // (unstructured.py:175,193) = the mesh's own int32 connectivity, read as it is; or -- tree_order -- the tree's
            // counter-clockwise-normalised copy, materialised as a table first
            // (vertex -> face table and the interpolation map behind it in ONE buffer: their host parts are adjacent, one upload)
            DevBuf<int64_t> face((size_t)n), faces_ccw((size_t)(reference_order ? 1 : voronoi->n_face * m));
            DevBuf<int32_t> count((size_t)n);
            // (vertices below n_identity are their own face -- the centroids --: bary_fill never reads their table entries)
            // The table and the flags of the cells with a substitute vertex stay on the tessellation: a second construction
            // on it (another target, another tolerance) finds them there.
            {
                const size_t n_tail = (size_t)(nv - n_identity), n_map = (size_t)(2 * n_extra);
                std::vector<int64_t> host(n_tail + n_map);
                if (n_tail > 0) memcpy(host.data(), vertex_face, sizeof(int64_t) * n_tail);
                if (n_map > 0) memcpy(host.data() + n_tail, node_to_node_map, sizeof(int64_t) * n_map);
                const bool same = voronoi->bary_ids.get() && voronoi->bary_n_identity == n_identity &&
                                  voronoi->bary_n_extra == n_extra && voronoi->bary_ids_host == host;
                if (!same) {
                    voronoi->bary_flag_valid = false;
                    voronoi->bary_ids.alloc((size_t)(nv + 2 * n_extra + 1));
                    if (!host.empty()) h2d(voronoi->bary_ids.get() + n_identity, host.data(), sizeof(int64_t) * host.size());
                    voronoi->bary_ids_host = std::move(host);
                    voronoi->bary_n_identity = n_identity;
                    voronoi->bary_n_extra = n_extra;
                }
            }
            int64_t *const vface = voronoi->bary_ids.get(), *const n2n = voronoi->bary_ids.get() + nv;
            if (!reference_order) mesh_faces_ccw_dev(voronoi, faces_ccw.get(), false);
            DevBuf<uint8_t> n_pos((size_t)n);
            if (!voronoi->bary_flag_valid) {
                voronoi->bary_cell_flag.alloc((size_t)voronoi->n_face);
                XR_LAUNCH("bary_cell_flag", k_bary_cell_flag, dim3(div_up(voronoi->n_face, 256)), dim3(256), 0, voronoi->faces_raw.get(),
                          voronoi->n_face, m, nv - n_extra, voronoi->bary_cell_flag.get());
                voronoi->bary_flag_valid = true;
            }
            const uint8_t *const cell_flag_p = voronoi->bary_cell_flag.get();
            XR_LAUNCH("barycentric", k_barycentric_cm, dim3(div_up(n, 256)), dim3(256), 0, voronoi->rec_fxy.get(),
                      voronoi->rec_len.get(), voronoi->record_off(), m, voronoi->grid, voronoi->cell_start.get(), voronoi->rec_bb.get(),
                      voronoi->rec_face.get(), voronoi->n_face, pts.get(), n, tol, face.get(), w.get(), cell_flag_p, n_pos.get());
            if (join_later) {
                side_join();
                pre->on_side = false;
            }
            if (star) {
                const double tol_source = pre ? pre->tol_source : resolve_tolerance(source, -1.0);
                const int64_t *vface_tab = voronoi->bary_ids.get();
                mesh_face_coords(source); // (its face-major vertex block in the caller's order: built once per mesh)
                mesh_prepare(source, false); // (the index for the points the star leaves open: still there unless the mesh was
                mesh_build_index(source);    // invalidated since the handle was made)
#define XR_STAR(MSV)                                                                                                                 \
    XR_LAUNCH("star_flag", k_star_flag<MSV>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), voronoi->faces_raw.get(), m, n_identity,  \
              vface_tab, nv - n_extra, source->fxy.get(), source->len.get(), source->caller_off(), source->m, pts.get(), n,           \
              tol_source, inside.get())
                if (source->m == 3 && !source->ragged()) XR_STAR(3);
                else if (source->m == 4 && !source->ragged()) XR_STAR(4);
                else XR_STAR(0);
#undef XR_STAR
                XR_LAUNCH("locate_flag", k_locate_flag<true>, dim3(div_up(n, 256)), dim3(256), 0, source->rec_fxy.get(),
                          source->rec_len.get(), source->record_off(), source->m, source->grid, source->cell_start.get(),
                          source->rec_bb.get(), source->rec_face.get(), source->n_face, pts.get(), n, tol_source, inside.get());
            }
            if (reference_order)
                XR_LAUNCH("bary_fix_count", k_bary_fix_count<int32_t>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), w.get(), m,
                          voronoi->faces_raw.get(), voronoi->node_xy.get(), n2n, nv - n_extra, inside.get(), n, count.get(), n_pos.get());
            else
                XR_LAUNCH("bary_fix_count", k_bary_fix_count<int64_t>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), w.get(), m,
                          faces_ccw.get(), voronoi->node_xy.get(), n2n, nv - n_extra, inside.get(), n, count.get(), n_pos.get());
            exclusive_scan_i32(count.get(), csr->indptr.get(), n);
            // The fill needs the row pointers, not the host: it is launched into arrays sized by a guess -- seven entries per point
            // (a Delaunay source gives six) -- BEFORE the host reads the number of entries, so that read-back, the two allocations and
            // the launch no longer sit between the scan and the fill (55 us of an idle device per construction, timeline).  A
            // matrix that does not fit is filled again into arrays of its real size.
            auto fill = [&](int64_t capacity) {
                csr->indices.alloc((size_t)capacity);
                csr->data.alloc((size_t)capacity);
                if (reference_order)
                    XR_LAUNCH("bary_fill", k_bary_fill<int32_t>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), w.get(), m,
                              voronoi->faces_raw.get(), vface, csr->indptr.get(), n, csr->indices.get(), csr->data.get(), n_identity,
                              capacity);
                else
                    XR_LAUNCH("bary_fill", k_bary_fill<int64_t>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), w.get(), m,
                              faces_ccw.get(), vface, csr->indptr.get(), n, csr->indices.get(), csr->data.get(), n_identity, capacity);
            };
            const int64_t guess = std::min<int64_t>(n * (int64_t)m, 7 * n + ((int64_t)1 << 16));
            const int64_t nnz = read_scalar(csr->indptr.get() + n, [&] { fill(guess); }); // (the fill behind the copy: it runs while nnz travels)
            csr->nnz = nnz;
            if (nnz > guess) fill(nnz);
            stream_sync();
        }
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    }
}

int xr_barycentric_csr(xr_mesh *voronoi, xr_mesh *source, xr_mesh *query, const double *points, int64_t n,
                       double tolerance, const int64_t *vertex_face, const int64_t *node_to_node_map, int64_t n_extra,
                       xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(vertex_face, XR_ERR_INVALID, "xr_barycentric_csr: NULL argument");
    barycentric_csr(voronoi, source, query, points, n, tolerance, 0, vertex_face, node_to_node_map, n_extra, true, out);
    XR_API_END
}

int xr_barycentric_csr_tail(xr_mesh *voronoi, xr_mesh *source, xr_mesh *query, const double *points, int64_t n,
                            double tolerance, int64_t n_identity, const int64_t *vertex_face_tail,
                            const int64_t *node_to_node_map, int64_t n_extra, int tree_order, xr_csr **out) {
    XR_API_BEGIN
    barycentric_csr(voronoi, source, query, points, n, tolerance, n_identity, vertex_face_tail, node_to_node_map, n_extra,
                    tree_order == 0, out);
    XR_API_END
}

int xr_locate_flags_begin(xr_mesh *source, xr_mesh *query, const double *points, int64_t n, xr_points **out) {
    XR_API_BEGIN
    XR_REQUIRE(source && out, XR_ERR_INVALID, "xr_locate_flags_begin: NULL argument");
    XR_REQUIRE((query != nullptr) != (points != nullptr), XR_ERR_INVALID,
               "xr_locate_flags_begin: give either a query mesh (its face centroids are the points) or points");
    if (query) n = query->n_face;
    XR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) - 1, XR_ERR_LIMIT, "xr_locate_flags_begin: too many points");
    xr_points *h = new xr_points();
    try {
        h->n = n;
        h->source = source;
        if (n > 0 && source->n_face > 0) {
            mesh_prepare(source, false);
            mesh_build_index(source);
            h->tol_source = resolve_tolerance(source, -1.0); // unstructured.py:189: default tolerance
            h->inside.alloc((size_t)n);
            h->query = query;
            if (!query) { // (a query mesh: its centroids, shared with the mesh, when the kernels are launched -- launch_points)
                h->pts.alloc((size_t)n * 2);
                h2d(h->pts.get(), points, sizeof(double) * 2 * (size_t)n);
            }
            h->flags_pending = option(OPT_STAR_FLAG) != 0;
            const bool defer = option(OPT_POINTS_DEFER) != 0; // (A/B switch)
            if (defer) {
                h->deferred = true; // launched by flush_pending_points
                pending_points().push_back(h);
            } else {
                {
                    SideScope side; // (forks behind everything enqueued so far: the index of the source grid, the points)
                    launch_points(h);
                }
                h->on_side = true;
            }
        }
        else if (n > 0) { // (no source faces: every point is outside)
            h->inside.alloc((size_t)n);
            XR_HIP(hipMemsetAsync(h->inside.get(), 0, (size_t)n, launch_stream()));
            if (query) {
                h->pts.share(mesh_centroids_shared(query));
            } else {
                h->pts.alloc((size_t)n * 2);
                h2d(h->pts.get(), points, sizeof(double) * 2 * (size_t)n);
            }
        }
    } catch (...) {
        delete h;
        throw;
    }
    *out = h; // (no wait: the kernels run on the side stream beside whatever the caller does next)
    XR_API_END
}

int xr_points_destroy(xr_points *points) {
    XR_API_BEGIN
    if (points) {
        forget_pending_points(points);
        stream_sync();
        delete points;
    }
    XR_API_END
}

int xr_barycentric_csr_points(xr_mesh *voronoi, xr_mesh *source, xr_points *points, double tolerance, int64_t n_identity,
                              const int64_t *vertex_face_tail, const int64_t *node_to_node_map, int64_t n_extra,
                              int tree_order, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(points, XR_ERR_INVALID, "xr_barycentric_csr_points: NULL argument");
    barycentric_csr(voronoi, source, nullptr, nullptr, 0, tolerance, n_identity, vertex_face_tail, node_to_node_map, n_extra,
                    tree_order == 0, out, points);
    XR_API_END
}

int xr_replace_interpolated_weights(const double *vertices, int64_t n_vertex, const int64_t *faces, int64_t n_face,
                                    int64_t m, const int64_t *face_index, double *weights, int64_t n,
                                    const int64_t *node_to_node_map, int64_t n_map) {
    XR_API_BEGIN
    XR_REQUIRE(n >= 0 && m >= 0 && n_vertex >= 0 && n_face >= 0 && n_map >= 0 && n_map <= n_vertex, XR_ERR_INVALID,
               "xr_replace_interpolated_weights: bad sizes");
    XR_REQUIRE(m <= XR_MAX_FACE_NODES * 4, XR_ERR_LIMIT, "xr_replace_interpolated_weights: too many slots per face");
    if (n == 0 || m == 0 || n_map == 0) return XR_OK;
    XR_REQUIRE(vertices && faces && face_index && weights && node_to_node_map, XR_ERR_INVALID,
               "xr_replace_interpolated_weights: NULL argument");
    for (int64_t i = 0; i < n; i++)
        XR_REQUIRE(face_index[i] >= -1 && face_index[i] < n_face, XR_ERR_INVALID,
                   "xr_replace_interpolated_weights: face_index[%lld] out of range", (long long)i);
    for (int64_t i = 0; i < n_face * m; i++)
        XR_REQUIRE(faces[i] >= -1 && faces[i] < n_vertex, XR_ERR_INVALID,
                   "xr_replace_interpolated_weights: faces entry %lld out of range", (long long)i);
    for (int64_t i = 0; i < 2 * n_map; i++)
        XR_REQUIRE(node_to_node_map[i] >= 0 && node_to_node_map[i] < n_vertex, XR_ERR_INVALID,
                   "xr_replace_interpolated_weights: node_to_node_map entry %lld out of range", (long long)i);
    // the kernel walks a COLUMN-major weight table (weight j of point i at [j * n + i])
    std::vector<double> cm((size_t)n * m);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++) cm[(size_t)j * n + i] = weights[(size_t)i * m + j];
    DevBuf<double> w((size_t)n * m), vxy((size_t)(n_vertex > 0 ? n_vertex : 1) * 2);
    DevBuf<int64_t> face((size_t)n), fc((size_t)(n_face > 0 ? n_face : 1) * m), n2n((size_t)n_map * 2);
    DevBuf<uint8_t> inside((size_t)n);
    DevBuf<int32_t> count((size_t)n);
    h2d(w.get(), cm.data(), sizeof(double) * cm.size());
    h2d(vxy.get(), vertices, sizeof(double) * 2 * (size_t)n_vertex);
    h2d(face.get(), face_index, sizeof(int64_t) * (size_t)n);
    h2d(fc.get(), faces, sizeof(int64_t) * (size_t)n_face * m);
    h2d(n2n.get(), node_to_node_map, sizeof(int64_t) * 2 * (size_t)n_map);
    XR_HIP(hipMemsetAsync(inside.get(), 1, (size_t)n, launch_stream()));
    XR_LAUNCH("bary_fix_count", k_bary_fix_count<int64_t>, dim3(div_up(n, 256)), dim3(256), 0, face.get(), w.get(), (int)m,
              fc.get(), vxy.get(), n2n.get(), n_vertex - n_map, inside.get(), n, count.get());
    d2h(cm.data(), w.get(), sizeof(double) * cm.size());
    stream_sync();
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++) weights[(size_t)i * m + j] = cm[(size_t)j * n + i];
    XR_API_END
}

int xr_locate_csr(xr_mesh *tree, xr_mesh *query, const double *points, int64_t n, double tolerance, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && out, XR_ERR_INVALID, "xr_locate_csr: NULL argument");
    XR_REQUIRE((query != nullptr) != (points != nullptr) || (n == 0 && !query), XR_ERR_INVALID,
               "xr_locate_csr: give either a query mesh (its face centroids are the points) or points");
    if (query) n = query->n_face;
    XR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) - 1, XR_ERR_LIMIT, "xr_locate_csr: too many points");
    xr_csr *csr = new xr_csr();
    try {
        csr->n = n; csr->m = tree->n_face; csr->nnz = 0;
        csr->indptr.alloc((size_t)n + 1);
        if (n == 0 || tree->n_face == 0) {
            fill_i32(csr->indptr.get(), 0, n + 1);
            csr->indices.alloc(0);
            csr->data.alloc(0);
            stream_sync();
        } else {
            mesh_prepare(tree, false);
            mesh_build_index(tree);
            const double tol = resolve_tolerance(tree, tolerance);
            PointsBuf pts;
            if (query) {
                pts.share(mesh_centroids_shared(query));
            } else {
                pts.alloc((size_t)n * 2);
                h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
            }
            DevBuf<int32_t> col((size_t)n), found((size_t)n);
            XR_LAUNCH("locate_col", k_locate_col, dim3(div_up(n, 256)), dim3(256), 0, tree->rec_fxy.get(),
                      tree->rec_len.get(), tree->record_off(), tree->m, tree->grid, tree->cell_start.get(), tree->rec_bb.get(),
                      tree->rec_face.get(), tree->n_face, pts.get(), n, tol, col.get(), found.get());
            exclusive_scan_i32(found.get(), csr->indptr.get(), n);
            // (a point has at most one entry: the arrays hold n, and the fill is enqueued in front of the read-back of nnz)
            csr->indices.alloc((size_t)n);
            csr->data.alloc((size_t)n);
            csr->nnz = read_scalar(csr->indptr.get() + n, [&] {
                XR_LAUNCH("locate_fill", k_locate_fill, dim3(div_up(n, 256)), dim3(256), 0, col.get(), csr->indptr.get(), n,
                          csr->indices.get(), csr->data.get());
            });
            stream_sync();
        }
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

} // extern "C"
