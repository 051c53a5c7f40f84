// xr_point_in_face.h -- the exact point-in-face test of the locate kernels, per lane.
//
// Same booleans as point_in_poly_or_on_edge of oracle/xr_oracle.c (the restatement of numba_celltree's locate_points the
// parity tests compare with): crossing number + "strictly within tol of an edge's line, projection on the segment".
// The oracle's two expensive expressions per edge -- sqrt(len2) of the on-edge test and the division of the crossing
// test, ~30 f64 instructions each, and the locate kernels are VALU-bound (PMC: vector ALUs busy 57-63 % of the kernel) --
// are only EVALUATED where their outcome is in doubt:
//   * `twice_area < tol * sqrt(len2)` is certainly false when twice_area^2 > 1.0000001 tol^2 len2 (the two sides differ
//     by 1e-7 relative, every rounding involved is below 1e-15 relative; products too small or too large for that bound
//     fall through to the exact expression);
//   * `p.x < wx * (p.y - v0.y) / wy + v0.x` is decided from the quotient formed with the hardware reciprocal (relative
//     error far below the 1e-5 the margin allows) unless p.x is within the margin of the crossing; NaN / infinity of the
//     approximation fail both comparisons and fall through as well.
// Whatever the filters decide is what the exact expression gives, so results stay bit-identical to the oracle's; the
// exact expressions themselves are unchanged.  Compiles as plain C++ for tests/test_point_in_face_host.py (which also
// degrades the reciprocal on purpose).
#pragma once

#include "xr_geom.h"

#ifndef XR_FAST_RCP
#define XR_FAST_RCP(x) __builtin_amdgcn_rcp(x) // v_rcp_f64
#endif

namespace xr {

// the on-edge test can be skipped: twice_area < tol * sqrt(len2) is certainly false
__device__ __forceinline__ bool edge_certainly_far(double twice_area, double len2, double tol) {
    const double rhs = (1.0000001 * (tol * tol)) * len2;
    return twice_area * twice_area > rhs && (rhs >= 1e-290 || tol == 0.0);
}

// p.x < num / wy + v0x, with num = wx * (p.y - v0.y)
__device__ __forceinline__ bool left_of_crossing(double px, double num, double wy, double v0x) {
    const double qa = num * XR_FAST_RCP(wy);
    const double d = px - (qa + v0x);
    const double margin = 1e-5 * fabs(qa) + 1e-12 * (fabs(v0x) + fabs(px));
    if (d < -margin) return true;
    if (d > margin) return false;
    return px < num / wy + v0x;
}

// load(i) -> vertex i of the face (CCW)
template <typename Load> __device__ __forceinline__ bool point_in_face_impl(Load load, int n, P2 p, double tol) {
    bool c = false;
    P2 v0 = load(n - 1);
    for (int i = 0; i < n; i++) {
        const P2 v1 = load(i);
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        const double len2 = wx * wx + wy * wy;
        if (len2 > 0) {
            const double ux = p.x - v0.x, uy = p.y - v0.y;
            const double twice_area = fabs(wx * uy - wy * ux);
            if (!edge_certainly_far(twice_area, len2, tol)) {
                const double len = sqrt(len2);
                if (twice_area < tol * len) {
                    const double tpar = ux * wx + uy * wy;
                    if (tpar >= 0 && tpar <= len2) return true;
                }
            }
            if ((v0.y > p.y) != (v1.y > p.y)) {
                if (left_of_crossing(p.x, wx * (p.y - v0.y), wy, v0.x)) c = !c;
            }
        }
        v0 = v1;
    }
    return c;
}

// the same test on MS vertices held in REGISTERS (n <= MS of them valid): the loop is unrolled with constant indices, so the
// vertex array never needs dynamic addressing (which would put it into scratch memory)
template <int MS, typename V> __device__ __forceinline__ bool point_in_face_regs(const V (&vtx)[MS], int n, P2 p, double tol) {
    bool c = false, on_edge = false;
    P2 v0{vtx[0].x, vtx[0].y};
#pragma unroll
    for (int k = 1; k < MS; k++)
        if (k == n - 1) v0 = P2{vtx[k].x, vtx[k].y};
#pragma unroll
    for (int i = 0; i < MS; i++) {
        if (i < n) {
            const P2 v1{vtx[i].x, vtx[i].y};
            const double wx = v1.x - v0.x, wy = v1.y - v0.y;
            const double len2 = wx * wx + wy * wy;
            if (len2 > 0) {
                const double ux = p.x - v0.x, uy = p.y - v0.y;
                const double twice_area = fabs(wx * uy - wy * ux);
                if (!edge_certainly_far(twice_area, len2, tol)) {
                    const double len = sqrt(len2);
                    if (twice_area < tol * len) {
                        const double tpar = ux * wx + uy * wy;
                        if (tpar >= 0 && tpar <= len2) on_edge = true;
                    }
                }
                if ((v0.y > p.y) != (v1.y > p.y)) {
                    if (left_of_crossing(p.x, wx * (p.y - v0.y), wy, v0.x)) c = !c;
                }
            }
            v0 = v1;
        }
    }
    return on_edge || c; // (point_in_face_impl returns at the first edge the point lies on: the same boolean)
}

} // namespace xr
