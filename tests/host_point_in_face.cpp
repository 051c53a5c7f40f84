// CPU harness of xugrid_amd/csrc/xr_point_in_face.h (test infrastructure): the per-lane exact test as plain C++, with a
// reciprocal that is WRONG by a chosen relative error -- up to the 1e-6 the filter's margin is built for; the hardware
// instruction is an order of magnitude better.
#include <hip/hip_runtime.h>
static double g_rcp_err = 0.0;
#define XR_FAST_RCP(x) ((1.0 / (x)) * (1.0 + g_rcp_err))
#include "xr_point_in_face.h"

extern "C" void host_set_rcp_error(double e) { g_rcp_err = e; }

extern "C" void host_point_in_face_many(const double *poly, int n, const double *pts, long n_pts, double tol, unsigned char *out) {
    const xr::P2 *v = reinterpret_cast<const xr::P2 *>(poly);
    for (long i = 0; i < n_pts; i++)
        out[i] = xr::point_in_face_impl([&](int j) { return v[j]; }, n, xr::P2{pts[2 * i], pts[2 * i + 1]}, tol) ? 1 : 0;
}

// how often the exact expressions were needed (the filters must leave little for them): counts of a second pass
extern "C" void host_filter_stats(const double *poly, int n, const double *pts, long n_pts, double tol, long *stats /* [4] */) {
    const xr::P2 *v = reinterpret_cast<const xr::P2 *>(poly);
    for (long i = 0; i < n_pts; i++) {
        const xr::P2 p{pts[2 * i], pts[2 * i + 1]};
        xr::P2 v0 = v[n - 1];
        for (int j = 0; j < n; j++) {
            const xr::P2 v1 = v[j];
            const double wx = v1.x - v0.x, wy = v1.y - v0.y, len2 = wx * wx + wy * wy;
            if (len2 > 0) {
                const double ux = p.x - v0.x, uy = p.y - v0.y;
                stats[0]++;
                if (!xr::edge_certainly_far(fabs(wx * uy - wy * ux), len2, tol)) stats[1]++;
                if ((v0.y > p.y) != (v1.y > p.y)) {
                    stats[2]++;
                    const double num = wx * (p.y - v0.y), qa = num * XR_FAST_RCP(wy), d = p.x - (qa + v0.x);
                    const double margin = 1e-5 * fabs(qa) + 1e-12 * (fabs(v0.x) + fabs(p.x));
                    if (!(d < -margin) && !(d > margin)) stats[3]++;
                }
            }
            v0 = v1;
        }
    }
}
