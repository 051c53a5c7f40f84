// xr_locate.hip -- point location and generalized barycentric weights on the device.
//
// Replaces numba_celltree.CellTree2d.locate_points (called from
// xugrid/regrid/unstructured.py:139,189 and xugrid/ugrid/ugridbase.py:1323) and
// CellTree2d.compute_barycentric_weights (xugrid/ugrid/ugrid2d.py:1054-1078).
// One thread per query point walks the hierarchical grid of the mesh; the arithmetic mirrors
// oracle/xr_oracle.c (point_in_poly_or_on_edge / bary_weights) operation for operation.
// A point matched by several faces (within tolerance of a shared edge) gets the LOWEST face id.
#include "xr_objects.h"

namespace xr {

// crossing-number test + "strictly within tol of an edge's line, projection on the segment"
__device__ bool point_in_face(const double *__restrict__ poly, int n, P2 p, double tol) {
    bool c = false;
    P2 v0 = load_p2(poly, n - 1);
    for (int i = 0; i < n; i++) {
        const P2 v1 = load_p2(poly, i);
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        const double len2 = wx * wx + wy * wy;
        if (len2 > 0) {
            const double ux = p.x - v0.x, uy = p.y - v0.y;
            const double twice_area = fabs(wx * uy - wy * ux);
            const double len = sqrt(len2);
            if (twice_area < tol * len) {
                const double tpar = ux * wx + uy * wy;
                if (tpar >= 0 && tpar <= len2) return true;
            }
            if ((v0.y > p.y) != (v1.y > p.y)) {
                const double xint = wx * (p.y - v0.y) / wy + v0.x;
                if (p.x < xint) c = !c;
            }
        }
        v0 = v1;
    }
    return c;
}

// -> record index of the matching face with the LOWEST caller face id, or -1
__device__ int locate_point(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, int m,
                            const GridParams &g, const int32_t *__restrict__ cell_start,
                            const float *__restrict__ rec_bb, const int32_t *__restrict__ rec_face, P2 p, double tol) {
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const float qx0 = f32_below(p.x - tol - g.x0), qx1 = f32_above(p.x + tol - g.x0);
    const float qy0 = f32_below(p.y - tol - g.y0), qy1 = f32_above(p.y + tol - g.y0);
    int best = -1, best_rec = -1;
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
        const int cx0 = cell_coord(p.x - tol - h, g.x0, inv_h, nx), cx1 = cell_coord(p.x + tol, g.x0, inv_h, nx);
        const int cy0 = cell_coord(p.y - tol - h, g.y0, inv_h, ny), cy1 = cell_coord(p.y + tol, g.y0, inv_h, ny);
        for (int cy = cy0; cy <= cy1; cy++) {
            const int r0 = cell_start[base + cy * nx + cx0];
            const int r1 = cell_start[base + cy * nx + cx1 + 1];
            for (int r = r0; r < r1; r++) {
                const float4 b = rbb[r];
                if (!(qx0 <= b.y && b.x <= qx1 && qy0 <= b.w && b.z <= qy1)) continue;
                const int f = rec_face[r];
                if (best >= 0 && f > best) continue;
                const double *poly = rec_fxy + (int64_t)r * m * 2;
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
                if (p.x < xmin - tol || p.x > xmax + tol || p.y < ymin - tol || p.y > ymax + tol) continue;
                if (point_in_face(poly, n, p, tol)) {
                    best = f;
                    best_rec = r;
                }
            }
        }
    }
    return best_rec;
}

__global__ void __launch_bounds__(256)
k_locate(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, int m, GridParams g,
         const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
         const int32_t *__restrict__ rec_face, const double *__restrict__ pts, int64_t n, double tol,
         int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const P2 p = load_p2(pts, (int)i);
    const int r = locate_point(rec_fxy, rec_len, m, g, cell_start, rec_bb, rec_face, p, tol);
    out[i] = r >= 0 ? rec_face[r] : -1;
}

// Wachspress coordinates with the on-edge special case; triangles use plain area coordinates.
// w points at this point's row of the (n, m) weight table (global memory, zero-initialised);
// weights are aligned with the CCW-normalised vertex order of the face (as numba_celltree's).
__device__ void bary_weights(const double *__restrict__ poly, int n, P2 p, double tol, double *__restrict__ w) {
    // pass 1: on-edge detection
    for (int i = 0; i < n; i++) {
        const P2 v0 = load_p2(poly, i), v1 = load_p2(poly, (i + 1) % n);
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        const double ux = p.x - v0.x, uy = p.y - v0.y;
        const double a = wx * uy - wy * ux;
        const double len2 = wx * wx + wy * wy;
        if (len2 > 0) {
            const double len = sqrt(len2);
            if (fabs(a) < tol * len) {
                const double tpar = ux * wx + uy * wy;
                if (tpar >= 0 && tpar <= len2) {
                    double tt = tpar / len2;
                    if (tt < 0) tt = 0;
                    if (tt > 1) tt = 1;
                    w[i] = 1.0 - tt;
                    w[(i + 1) % n] = tt;
                    return;
                }
            }
        }
    }
    auto A = [&](int i) { // cross(v_i - p, v_{i+1} - p) in the oracle's form
        const P2 v0 = load_p2(poly, i), v1 = load_p2(poly, (i + 1) % n);
        const double wx = v1.x - v0.x, wy = v1.y - v0.y;
        const double ux = p.x - v0.x, uy = p.y - v0.y;
        return wx * uy - wy * ux;
    };
    if (n == 3) {
        const double a0 = A(0), a1 = A(1), a2 = A(2);
        const double s = a0 + a1 + a2;
        w[0] = a1 / s;
        w[1] = a2 / s;
        w[2] = a0 / s;
        return;
    }
    double wsum = 0.0;
    for (int i = 0; i < n; i++) {
        const int ip = (i + n - 1) % n, in = (i + 1) % n;
        const P2 vp = load_p2(poly, ip), vi = load_p2(poly, i), vn = load_p2(poly, in);
        const double cx = (vi.x - vp.x) * (vn.y - vi.y) - (vi.y - vp.y) * (vn.x - vi.x);
        const double wi = cx / (A(ip) * A(i));
        w[i] = wi;
        wsum += wi;
    }
    for (int i = 0; i < n; i++) w[i] = w[i] / wsum;
}

__global__ void __launch_bounds__(256)
k_barycentric(const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, int m, GridParams g,
              const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
              const int32_t *__restrict__ rec_face, const double *__restrict__ pts, int64_t n, double tol,
              int64_t *__restrict__ face_out, double *__restrict__ weights) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const P2 p = load_p2(pts, (int)i);
    const int r = locate_point(rec_fxy, rec_len, m, g, cell_start, rec_bb, rec_face, p, tol);
    face_out[i] = r >= 0 ? rec_face[r] : -1;
    double *w = weights + i * m;
    for (int j = 0; j < m; j++) w[j] = 0.0;
    if (r >= 0) bary_weights(rec_fxy + (int64_t)r * m * 2, rec_len[r], p, tol, w);
}

static double resolve_tolerance(xr_mesh *mesh, double tolerance) {
    if (tolerance >= 0) return tolerance;
    mesh_read_stats(mesh);
    return 1e-12 * mesh->h_stats[6]; // ugridbase.py:1165-1170
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
        mesh_prepare(mesh);
        mesh_build_index(mesh);
        const double tol = resolve_tolerance(mesh, tolerance);
        DevBuf<double> pts((size_t)n * 2);
        DevBuf<int64_t> out((size_t)n);
        h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
        if (mesh->n_face > 0) {
            XR_LAUNCH("locate_points", k_locate, dim3(div_up(n, 256)), dim3(256), 0, mesh->rec_fxy.get(),
                      mesh->rec_len.get(), mesh->m, mesh->grid, mesh->cell_start.get(), mesh->rec_bb.get(),
                      mesh->rec_face.get(), pts.get(), n, tol, out.get());
            XR_HIP(hipMemcpyAsync(face_index_out, out.get(), sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost,
                                  engine().stream));
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
        mesh_prepare(mesh);
        mesh_build_index(mesh);
        const double tol = resolve_tolerance(mesh, tolerance);
        const int m = mesh->m;
        if (mesh->n_face > 0) {
            DevBuf<double> pts((size_t)n * 2), w((size_t)n * m);
            DevBuf<int64_t> out((size_t)n);
            h2d(pts.get(), points, sizeof(double) * 2 * (size_t)n);
            XR_LAUNCH("barycentric", k_barycentric, dim3(div_up(n, 256)), dim3(256), 0, mesh->rec_fxy.get(),
                      mesh->rec_len.get(), m, mesh->grid, mesh->cell_start.get(), mesh->rec_bb.get(),
                      mesh->rec_face.get(), pts.get(), n, tol, out.get(), w.get());
            XR_HIP(hipMemcpyAsync(face_index_out, out.get(), sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost,
                                  engine().stream));
            XR_HIP(hipMemcpyAsync(weights_out, w.get(), sizeof(double) * (size_t)n * m, hipMemcpyDeviceToHost,
                                  engine().stream));
            stream_sync();
        } else {
            for (int64_t i = 0; i < n; i++) face_index_out[i] = -1;
            for (int64_t i = 0; i < n * m; i++) weights_out[i] = 0.0;
        }
    }
    XR_API_END
}

} // extern "C"
