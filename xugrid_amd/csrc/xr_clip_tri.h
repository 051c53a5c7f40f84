// xr_clip_tri.h -- triangle x triangle Sutherland-Hodgman clip + fan area, per lane, few instructions.
//
// Same arithmetic, in the same order, as clip_polygons / sh_polygon_area of oracle/xr_oracle.c (the restatement of
// numba_celltree's clip the parity tests compare with bit for bit): every floating-point value below is produced by
// the expression the oracle uses.  What differs is the bookkeeping around it.
//
// One clip stage (clipper edge r -> s) of the oracle walks the subject polygon vertex by vertex and, per vertex,
// evaluates an inside test AND (under divergence: for every slot in which any lane of the wave crosses the edge)
// an intersection with its division.  Here a stage is
//   1. the inside flags of all vertices (5 flops each), packed into a per-lane bit mask F,
//   2. the transition slots (f_{j-1} != f_j) by bit arithmetic on F; a convex polygon cut by a line has 0 or 2,
//   3. exactly two intersections, their end points fetched from the lane's LDS column by dynamic index,
//   4. a compaction in the oracle's emission order (for j ascending: [crossing point of edge j-1 -> j] [vertex j if
//      inside]): the positions come from a 64-entry table indexed by (n, F), built once per block in LDS; vertices
//      outside go to a trash slot, so the stores are unconditional,
//   5. the next stage starts by reading the column back with static indices.
// Lanes whose polygon is not "regular" run the oracle's loop itself for that stage (tri_stage_generic: same results
// by construction, only slower, and only for those lanes): more than two transitions, a crossing edge parallel to
// the clip line (the oracle's keep-b quirk), more vertices than the stage was unrolled for, or a REPEATED vertex
// (the oracle skips zero-length edges).  Repeated vertices are detected where they are made -- a crossing point
// equal to an end point of its edge or to the other crossing point, a degenerate input triangle, any output of the
// generic loop -- and make the lane "dirty" for its remaining stages.  Shared-vertex mesh pairs (a mesh against
// itself or its refinement) take that branch often; random mesh pairs practically never.
#pragma once

#include "xr_geom.h"

namespace xr {

static constexpr double TRI_AREA_OVERFLOW = -1.0; // sentinel: more than 6 vertices (floating-point degenerate pair)
static constexpr int TRI_MAXV = 6;                // slots 0..5 of the lane's LDS column; slot 6 = trash
static constexpr int TRI_LUT = 64;                // compaction table entries (uint2 each)

// LDS layout: every lane owns TRI_MAXV + 1 consecutive double2 slots (col[slot]); 16 lanes x 112 bytes fall on
// disjoint banks for the 16-byte accesses.  Positions in the table are BYTE offsets (slot * 16), one per byte, so an
// address is lane base + one extracted byte.
//
// Compaction table: entry (1 << n) + F, n = 3..5 vertices, F = inside mask.  .x bytes 0..3: output offset of vertex
// 0..3; .y byte 0: vertex 4, byte 1 / 2: the crossing points of the first / second transition slot.  A vertex that is
// outside (or beyond the polygon) goes to the trash slot.  Call with all threads of the block, then __syncthreads().
// MAXV = 6: triangle subjects (entries (1 << n) + F, n = 3..5); MAXV = 7: subjects of up to four vertices (n = 3..6, 128
// entries; vertex 5 in .y byte 3).  The trash slot is slot MAXV.
template <int MAXV>
__device__ __forceinline__ void poly_lut_init(uint2 *lut) {
    constexpr int N_LUT = 1 << MAXV;
    const int i = threadIdx.x;
    if (i >= N_LUT) return;
    uint32_t pos[8] = {MAXV, MAXV, MAXV, MAXV, MAXV, MAXV, MAXV, MAXV}; // vertex 0..4, crossing point 1, 2, vertex 5
    if (i >= 8) {
        const int n = 31 - __clz(i);
        const uint32_t F = (uint32_t)i - (1u << n);
        int before = 0, n_t = 0;
        for (int j = 0; j < n; j++) {
            const bool fj = (F >> j) & 1u, fp = (F >> ((j + n - 1) % n)) & 1u;
            if (fj != fp) {
                if (n_t < 2) pos[5 + n_t] = (uint32_t)before;
                n_t++;
                before++;
            }
            if (fj) pos[j < 5 ? j : 7] = (uint32_t)(before++);
        }
    }
    uint2 e;
    e.x = (pos[0] << 4) | (pos[1] << 12) | (pos[2] << 20) | (pos[3] << 28);
    e.y = (pos[4] << 4) | (pos[5] << 12) | (pos[6] << 20) | (pos[7] << 28);
    lut[i] = e;
}
__device__ __forceinline__ void tri_lut_init(uint2 *lut) { poly_lut_init<TRI_MAXV>(lut); }

__device__ __forceinline__ bool p2_eq(P2 a, P2 b) { return a.x == b.x && a.y == b.y; }
__device__ __forceinline__ bool p2_eq(P2 a, double2 b) { return a.x == b.x && a.y == b.y; }
template <int STRIDE>
__device__ __forceinline__ double2 *tri_slot(double2 *col, uint32_t byte_offset) {
    return reinterpret_cast<double2 *>(reinterpret_cast<char *>(col) + byte_offset * STRIDE);
}

// The oracle's stage loop on a register polygon (static indexing, predicated on the current length); output
// pushed into the lane's LDS column (slot TRI_MAXV = trash for clamped pushes).
template <int MAXV, int STRIDE>
__device__ __forceinline__ void poly_stage_generic(const P2 (&v)[MAXV], int &n, const P2 r, const P2 U, bool &alive,
                                                   bool &overflow, double2 *col) {
    const P2 N{-U.y, U.x};
    int n_output = 0;
    P2 a = v[0];
#pragma unroll
    for (int j = 1; j < MAXV; j++)
        if (j < n) a = v[j];
    bool a_inside = U.x * (a.y - r.y) > U.y * (a.x - r.x);
#pragma unroll
    for (int j = 0; j < MAXV; j++) {
        if (j < n) {
            const P2 b = v[j];
            const P2 V{b.x - a.x, b.y - a.y};
            const bool live = !(V.x == 0 && V.y == 0);
            bool b_inside = U.x * (b.y - r.y) > U.y * (b.x - r.x);
            const bool cross = live && (b_inside != a_inside);
            P2 pt{0.0, 0.0};
            bool have_pt = false;
            if (cross) {
                const double wx = r.x - a.x, wy = r.y - a.y;
                const double nw = N.x * wx + N.y * wy;
                const double nv = N.x * V.x + N.y * V.y;
                if (nv != 0) {
                    const double tt = nw / nv;
                    pt.x = a.x + tt * V.x;
                    pt.y = a.y + tt * V.y;
                    have_pt = true;
                }
            }
            const bool quirk = cross && !b_inside && !have_pt; // parallel edge: keep b, which then counts as inside
            if (cross && have_pt) {
                col[(n_output < MAXV ? n_output : MAXV) * STRIDE] = make_double2(pt.x, pt.y);
                n_output++;
            }
            b_inside = b_inside || quirk;
            if (live && b_inside) {
                col[(n_output < MAXV ? n_output : MAXV) * STRIDE] = make_double2(b.x, b.y);
                n_output++;
            }
            if (live) {
                a = b;
                a_inside = b_inside;
            }
        }
    }
    if (n_output > MAXV) {
        overflow = true;
        alive = false;
    } else if (n_output < 3) {
        alive = false;
    }
    n = n_output;
}

// One stage.  NIN = number of vertices the fast path is unrolled for (3, 4, 5 for the three edges of a triangle
// clipper: a regular stage adds at most one vertex).  The lane's LDS column holds the current polygon before and
// after; registers are only a per-stage copy.
template <int MAXV, int NIN, int STRIDE, bool LAST = false>
__device__ __forceinline__ void poly_stage(int &n, P2 &r, const P2 s, bool &alive, bool &dirty, bool &overflow,
                                           double2 *col, const uint2 *lut) {
    const P2 U{s.x - r.x, s.y - r.y};
    const bool work = alive && !(U.x == 0 && U.y == 0); // zero-length clipper edge: the oracle skips the stage, r stays
    P2 v[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; j++) {
        if (j < NIN) {
            const double2 q = col[j * STRIDE]; // (slots >= n: stale values, masked below)
            v[j] = P2{q.x, q.y};
        } else {
            v[j] = P2{0.0, 0.0};
        }
    }
    // ---- inside flags as a bit mask (slots >= n masked off)
    uint32_t F = 0;
#pragma unroll
    for (int j = 0; j < NIN; j++) F |= (U.x * (v[j].y - r.y) > U.y * (v[j].x - r.x)) ? (1u << j) : 0u;
    const uint32_t nmask = (1u << n) - 1u;
    F &= nmask;
    const uint32_t Tm = (F ^ ((F << 1) | (F >> ((n - 1) & 31)))) & nmask; // bit j: edge (j-1 -> j) crosses the clip line
    const int n_tr = __popc(Tm);
    bool irregular = work && (dirty || n > NIN || (n_tr != 0 && n_tr != 2));
    const bool two = work && !irregular && n_tr == 2;
    if (two) {
        // end points of the two crossing edges from the lane's LDS column
        const int j1 = __ffs(Tm) - 1, j2 = 31 - __clz(Tm);
        const int p1 = j1 == 0 ? n - 1 : j1 - 1, p2 = j2 - 1;
        const double2 a1 = col[p1 * STRIDE], b1 = col[j1 * STRIDE], a2 = col[p2 * STRIDE], b2 = col[j2 * STRIDE];
        const uint2 e = lut[F + (1u << n)];
        const P2 N{-U.y, U.x};
        const P2 V1{b1.x - a1.x, b1.y - a1.y}, V2{b2.x - a2.x, b2.y - a2.y};
        const double nw1 = N.x * (r.x - a1.x) + N.y * (r.y - a1.y), nv1 = N.x * V1.x + N.y * V1.y;
        const double nw2 = N.x * (r.x - a2.x) + N.y * (r.y - a2.y), nv2 = N.x * V2.x + N.y * V2.y;
        const double tt1 = nw1 / nv1, tt2 = nw2 / nv2;
        const P2 pt1{a1.x + tt1 * V1.x, a1.y + tt1 * V1.y}, pt2{a2.x + tt2 * V2.x, a2.y + tt2 * V2.y};
        irregular = nv1 == 0 || nv2 == 0; // parallel crossing edge: the oracle's keep-b quirk
        if (!irregular) {
            // a crossing point that coincides with a neighbour in the output is a repeated vertex for later stages
            // (x first: the y comparisons only run when some lane has an equal x)
            // (the last stage has no successor: a repeated vertex only adds an empty fan triangle to the area)
            if (!LAST && (pt1.x == a1.x || pt1.x == b1.x || pt2.x == a2.x || pt2.x == b2.x || pt1.x == pt2.x))
                dirty = p2_eq(pt1, a1) || p2_eq(pt1, b1) || p2_eq(pt2, a2) || p2_eq(pt2, b2) || p2_eq(pt1, pt2);
            // compaction in the oracle's emission order; vertices outside land in the trash slot
            *tri_slot<STRIDE>(col, e.x & 0xffu) = make_double2(v[0].x, v[0].y);
            *tri_slot<STRIDE>(col, (e.x >> 8) & 0xffu) = make_double2(v[1].x, v[1].y);
            *tri_slot<STRIDE>(col, (e.x >> 16) & 0xffu) = make_double2(v[2].x, v[2].y);
            if (NIN > 3) *tri_slot<STRIDE>(col, e.x >> 24) = make_double2(v[3].x, v[3].y);
            if (NIN > 4) *tri_slot<STRIDE>(col, e.y & 0xffu) = make_double2(v[4].x, v[4].y);
            if (NIN > 5) *tri_slot<STRIDE>(col, e.y >> 24) = make_double2(v[5].x, v[5].y);
            *tri_slot<STRIDE>(col, (e.y >> 8) & 0xffu) = make_double2(pt1.x, pt1.y);
            *tri_slot<STRIDE>(col, (e.y >> 16) & 0xffu) = make_double2(pt2.x, pt2.y);
            n = __popc(F) + 2; // (>= 3: a transition implies an inside vertex)
        }
    } else if (work && !irregular && !(F & 1u)) {
        alive = false; // no transition, first vertex outside: everything is outside
    }
    // (no transition, first vertex inside: the polygon is unchanged)
    if (irregular) {
        if (n > NIN) { // (only after an earlier generic stage: fetch the vertices the fast path does not unroll)
#pragma unroll
            for (int j = NIN; j < MAXV; j++) {
                if (j < n) {
                    const double2 q = col[j * STRIDE];
                    v[j] = P2{q.x, q.y};
                }
            }
        }
        poly_stage_generic<MAXV, STRIDE>(v, n, r, U, alive, overflow, col);
        dirty = true; // (the generic loop may emit repeated vertices)
    }
    if (work) r = s;
}

// area of (subject polygon tv, n0 <= N0 vertices: a target face) clipped by (source triangle sv, counter-clockwise), or
// TRI_AREA_OVERFLOW.  col: the lane's LDS column, col[0 .. MAXV] (MAXV + 1 double2 slots, contiguous); lut: poly_lut_init<MAXV>'s
// table.  N0 = 3, MAXV = 6: triangle x triangle; N0 = 4, MAXV = 7: the faces of a quadrilateral (raster) target, which may
// be triangles with a fill slot (n0 = 3).
template <int MAXV, int N0, int STRIDE = 1>
__device__ __forceinline__ double poly_clip_area(const P2 (&tv)[N0], int n0, const P2 (&sv)[3], double2 *col, const uint2 *lut,
                                                 bool active) {
    int n = n0;
    bool alive = active, overflow = false;
    bool dirty = false;
#pragma unroll
    for (int j = 0; j < N0; j++)
        if (j < n0) dirty = dirty || p2_eq(tv[j], tv[j + 1 < n0 ? j + 1 : 0]);
    if (N0 > 3 && n0 > 3) dirty = dirty || p2_eq(tv[0], tv[2]) || p2_eq(tv[1], tv[3 < N0 ? 3 : 0]);
    if (active) {
#pragma unroll
        for (int j = 0; j < N0; j++)
            if (j < n0) col[j * STRIDE] = make_double2(tv[j].x, tv[j].y);
    }
    P2 r = sv[2];
    poly_stage<MAXV, N0, STRIDE>(n, r, sv[0], alive, dirty, overflow, col, lut);
    poly_stage<MAXV, N0 + 1, STRIDE>(n, r, sv[1], alive, dirty, overflow, col, lut);
    poly_stage<MAXV, N0 + 2, STRIDE, true>(n, r, sv[2], alive, dirty, overflow, col, lut);
    if (overflow) return TRI_AREA_OVERFLOW;
    double area = 0.0;
    if (alive) {
        // fan area from the first clipped vertex (local origin)
        const double2 q0 = col[0], q1 = col[STRIDE];
        const P2 a0{q0.x, q0.y};
        double ux = q1.x - a0.x, uy = q1.y - a0.y;
#pragma unroll
        for (int i = 2; i < MAXV; i++) {
            if (i < n) {
                const double2 q = col[i * STRIDE];
                const double vx = a0.x - q.x, vy = a0.y - q.y;
                area += fabs(ux * vy - uy * vx);
                ux = vx;
                uy = vy;
            }
        }
        area = 0.5 * area;
    }
    return area;
}

template <int STRIDE = 1>
__device__ __forceinline__ double tri_clip_area(const P2 (&tv)[3], const P2 (&sv)[3], double2 *col, const uint2 *lut,
                                                bool active) {
    return poly_clip_area<TRI_MAXV, 3, STRIDE>(tv, 3, sv, col, lut, active);
}

static constexpr int QUAD_MAXV = 7;              // quadrilateral subject: up to 7 vertices after three clip edges
static constexpr int QUAD_LUT = 1 << QUAD_MAXV;  // compaction table entries

} // namespace xr
