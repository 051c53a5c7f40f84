// xr_voronoi.hip -- the O(n) part of the centroidal Voronoi pre-step of BarycentricInterpolator on the
// device.
//
// Replaces, for the call made at xugrid/regrid/unstructured.py:151-165
// (voronoi_topology(..., add_exterior=True, add_vertices=True, skip_concave=True)):
//   * Ugrid2d.node_face_connectivity (ugrid/ugrid2d.py:700-713 -> connectivity.invert_dense_to_sparse,
//     ugrid/connectivity.py:247-259)                      -> node -> faces CSR by counting sort
//   * Ugrid2d.edge_node_connectivity / edge_face_connectivity (ugrid2d.py:497-509, :661-677 ->
//     connectivity.edge_connectivity, connectivity.py:419-457) -- of which the Voronoi step only uses
//     the EXTERIOR edges (voronoi.py:77-97)               -> exterior half-edges found through the CSR
//   * the interior cells: centroids of the faces around every node that touches no exterior edge,
//     ordered counter-clockwise about the node (voronoi.py:355-372: lexsort((arctan2, node)))
//   * the assembly of the dense, -1 padded cell table and of the vertex array into a device-resident mesh
// The cells of nodes ON the boundary (projections on exterior edges, substitute vertices, convexity
// choice: voronoi.py:59-327) are O(boundary) and stay host numpy (xugrid_amd/voronoi.py); they are handed
// back in as a small table.
#include <algorithm>
#include <vector>

#include <cstring>

#include "xr_objects.h"

struct xr_voronoi {
    xr_mesh *mesh = nullptr; // borrowed: the caller keeps the source mesh alive
    int64_t n_node = 0, n_face = 0, nnz = 0;
    xr::DevBuf<int32_t> indptr;    // [n_node+1]
    xr::DevBuf<int32_t> faces_asc; // [nnz] faces around each node, ascending (scipy CSR order)
    xr::DevBuf<int32_t> faces_ccw; // [nnz] interior nodes: counter-clockwise about the node
    xr::DevBuf<uint8_t> interior;  // [n_node] 1 = has faces and touches no exterior edge
    xr::DevBuf<int32_t> cell_rank; // [n_node+1] exclusive scan of `interior`
    xr::DevBuf<double> centroids;  // [n_face*2]
    int64_t n_interior = 0;
    int max_degree = 0, min_degree = 0; // over interior nodes
    std::vector<int64_t> edge_lo, edge_hi, edge_face; // exterior edges, lexicographic (lo, hi)
    // the rows of the boundary nodes (what the O(boundary) host part reads), gathered on first request
    bool boundary_ready = false;
    std::vector<int64_t> b_nodes, b_ptr, b_faces; // ascending node ids, CSR offsets, faces ascending per node
    std::vector<double> b_face_xy, b_edge_face_xy; // centroid of every listed face / of every exterior edge's face
};

namespace xr {

__global__ void __launch_bounds__(256)
k_vor_count(const int32_t *__restrict__ faces, int64_t total, int32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int v = faces[i];
    if (v >= 0) atomicAdd(&count[v], 1);
}

__global__ void __launch_bounds__(256)
k_vor_scatter(const int32_t *__restrict__ faces, int64_t total, int m, const int32_t *__restrict__ indptr,
              int32_t *__restrict__ cursor, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int v = faces[i];
    if (v >= 0) out[indptr[v] + atomicAdd(&cursor[v], 1)] = (int32_t)(i / m);
}

// ascending face ids per node (the scatter order is arbitrary); rows are short
__global__ void __launch_bounds__(256)
k_vor_sort_rows(const int32_t *__restrict__ indptr, int64_t n_node, int32_t *__restrict__ rows) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_node) return;
    const int s = indptr[v], e = indptr[v + 1];
    for (int i = s + 1; i < e; i++) {
        const int key = rows[i];
        int j = i - 1;
        while (j >= s && rows[j] > key) {
            rows[j + 1] = rows[j];
            j--;
        }
        rows[j + 1] = key;
    }
}

__device__ __forceinline__ int face_len(const int32_t *__restrict__ face, int m) {
    int n = m;
    for (int i = m - 1; i >= 3; i--)
        if (face[i] < 0) n = i;
    return n;
}

// One thread per half-edge (face f, slot j): a -> b.  The edge is exterior iff no OTHER face around node a
// has a and b as ring neighbours (either direction).  Exterior edges are appended (unordered).
__global__ void __launch_bounds__(256)
k_vor_exterior(const int32_t *__restrict__ faces, int64_t n_face, int m, const int32_t *__restrict__ indptr,
               const int32_t *__restrict__ rows, uint8_t *__restrict__ on_boundary, int32_t *__restrict__ n_edges,
               int32_t *__restrict__ e_lo, int32_t *__restrict__ e_hi, int32_t *__restrict__ e_face) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_face * m) return;
    const int64_t f = i / m;
    const int j = (int)(i - f * m);
    const int32_t *face = faces + f * m;
    const int n = face_len(face, m);
    if (j >= n) return;
    const int a = face[j], b = face[(j + 1) % n];
    if (a == b) return;
    bool shared = false;
    for (int r = indptr[a]; r < indptr[a + 1] && !shared; r++) {
        const int g = rows[r];
        if (g == f) continue;
        const int32_t *other = faces + (int64_t)g * m;
        const int k = face_len(other, m);
        for (int t = 0; t < k; t++) {
            if (other[t] != a) continue;
            if (other[(t + 1) % k] == b || other[(t + k - 1) % k] == b) shared = true;
        }
    }
    if (shared) return;
    on_boundary[a] = 1;
    on_boundary[b] = 1;
    const int at = atomicAdd(n_edges, 1);
    e_lo[at] = a < b ? a : b;
    e_hi[at] = a < b ? b : a;
    e_face[at] = (int32_t)f;
}

// angle class of a direction in arctan2's order (-pi, pi]: 0: dy < 0, 1: dy == 0 & dx >= 0, 2: dy > 0,
// 3: dy == 0 & dx < 0
__device__ __forceinline__ int angle_class(double dx, double dy) {
    if (dy < 0) return 0;
    if (dy > 0) return 2;
    return dx < 0 ? 3 : 1;
}

// true iff direction a comes strictly before direction b in arctan2 order
__device__ __forceinline__ bool angle_less(double ax, double ay, double bx, double by) {
    const int ca = angle_class(ax, ay), cb = angle_class(bx, by);
    if (ca != cb) return ca < cb;
    if (ca == 1 || ca == 3) return false; // same ray
    return ax * by - ay * bx > 0;
}

// Interior nodes: order the surrounding face centroids counter-clockwise about the node (stable on ties:
// ascending face id, as lexsort).  Also flags the node and records the degree range.
__global__ void __launch_bounds__(256)
k_vor_interior(const double *__restrict__ node_xy, const double *__restrict__ cxy,
               const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows_asc,
               const uint8_t *__restrict__ on_boundary, int64_t n_node, int32_t *__restrict__ rows_ccw,
               uint8_t *__restrict__ interior, int32_t *__restrict__ flag32, int32_t *__restrict__ deg_minmax) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_node) return;
    const int s = indptr[v], e = indptr[v + 1];
    const bool ok = e > s && !on_boundary[v];
    interior[v] = ok;
    flag32[v] = ok;
    if (!ok) {
        for (int i = s; i < e; i++) rows_ccw[i] = rows_asc[i];
        return;
    }
    atomicMin(&deg_minmax[0], e - s);
    atomicMax(&deg_minmax[1], e - s);
    const P2 p = load_p2(node_xy, (int)v);
    for (int i = s; i < e; i++) { // insertion sort in global memory (rows are short, L2 resident)
        const int key = rows_asc[i];
        const P2 c = load_p2(cxy, key);
        const double kx = c.x - p.x, ky = c.y - p.y;
        int j = i - 1;
        while (j >= s) {
            const P2 d = load_p2(cxy, rows_ccw[j]);
            if (!angle_less(kx, ky, d.x - p.x, d.y - p.y)) break;
            rows_ccw[j + 1] = rows_ccw[j];
            j--;
        }
        rows_ccw[j + 1] = key;
    }
}

__global__ void __launch_bounds__(256)
k_vor_cells(const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows_ccw,
            const uint8_t *__restrict__ interior, const int32_t *__restrict__ cell_rank, int64_t n_node, int m,
            int32_t *__restrict__ cells) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_node || !interior[v]) return;
    int32_t *row = cells + (int64_t)cell_rank[v] * m;
    const int s = indptr[v], n = indptr[v + 1] - s;
    for (int j = 0; j < m; j++) row[j] = j < n ? rows_ccw[s + j] : -1;
}

__global__ void __launch_bounds__(256)
k_vor_boundary_cells(const int32_t *__restrict__ table, int64_t n_rows, int mb, int m, int32_t *__restrict__ cells) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * m) return;
    const int64_t r = i / m;
    const int j = (int)(i - r * m);
    cells[i] = j < mb ? table[r * mb + j] : -1;
}

__global__ void __launch_bounds__(256) k_vor_widen(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// rows of selected nodes: degree, then faces + their centroids
__global__ void k_vor_row_degree(const int32_t *__restrict__ indptr, const int64_t *__restrict__ nodes, int64_t n,
                                 int64_t *__restrict__ deg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) deg[i] = indptr[nodes[i] + 1] - indptr[nodes[i]];
}

__global__ void k_vor_row_gather(const int32_t *__restrict__ indptr, const int32_t *__restrict__ faces_asc,
                                 const double *__restrict__ centroids, const int64_t *__restrict__ nodes,
                                 const int64_t *__restrict__ ptr, int64_t n, int64_t *__restrict__ faces,
                                 double *__restrict__ xy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = indptr[nodes[i]], e = indptr[nodes[i] + 1];
    for (int r = s; r < e; r++) {
        const int f = faces_asc[r];
        const int64_t o = ptr[i] + (r - s);
        faces[o] = f;
        xy[2 * o] = centroids[2 * f];
        xy[2 * o + 1] = centroids[2 * f + 1];
    }
}

__global__ void k_vor_face_xy(const double *__restrict__ centroids, const int64_t *__restrict__ faces, int64_t n,
                              double *__restrict__ xy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    xy[2 * i] = centroids[2 * faces[i]];
    xy[2 * i + 1] = centroids[2 * faces[i] + 1];
}

static void voronoi_boundary(xr_voronoi *v) {
    if (v->boundary_ready) return;
    std::vector<int64_t> nodes(v->edge_lo);
    nodes.insert(nodes.end(), v->edge_hi.begin(), v->edge_hi.end());
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    const int64_t nb = (int64_t)nodes.size(), ne = (int64_t)v->edge_face.size();
    v->b_nodes = nodes;
    v->b_ptr.assign((size_t)nb + 1, 0);
    v->b_faces.clear();
    v->b_face_xy.clear();
    v->b_edge_face_xy.assign((size_t)ne * 2, 0.0);
    if (nb > 0) {
        DevBuf<int64_t> d_nodes((size_t)nb), d_deg((size_t)nb), d_ptr((size_t)nb + 1);
        h2d(d_nodes.get(), nodes.data(), sizeof(int64_t) * (size_t)nb);
        XR_LAUNCH("vor_row_degree", k_vor_row_degree, dim3(div_up(nb, 256)), dim3(256), 0, v->indptr.get(), d_nodes.get(), nb,
                  d_deg.get());
        std::vector<int64_t> deg((size_t)nb);
        d2h(deg.data(), d_deg.get(), sizeof(int64_t) * (size_t)nb);
        for (int64_t i = 0; i < nb; i++) v->b_ptr[(size_t)i + 1] = v->b_ptr[(size_t)i] + deg[(size_t)i];
        const int64_t total = v->b_ptr[(size_t)nb];
        v->b_faces.resize((size_t)total);
        v->b_face_xy.resize((size_t)total * 2);
        if (total > 0) {
            DevBuf<int64_t> d_faces((size_t)total);
            DevBuf<double> d_xy((size_t)total * 2);
            h2d(d_ptr.get(), v->b_ptr.data(), sizeof(int64_t) * (size_t)(nb + 1));
            XR_LAUNCH("vor_row_gather", k_vor_row_gather, dim3(div_up(nb, 256)), dim3(256), 0, v->indptr.get(),
                      v->faces_asc.get(), v->centroids.get(), d_nodes.get(), d_ptr.get(), nb, d_faces.get(), d_xy.get());
            d2h(v->b_faces.data(), d_faces.get(), sizeof(int64_t) * (size_t)total);
            d2h(v->b_face_xy.data(), d_xy.get(), sizeof(double) * 2 * (size_t)total);
        }
    }
    if (ne > 0) {
        DevBuf<int64_t> d_ef((size_t)ne);
        DevBuf<double> d_xy((size_t)ne * 2);
        h2d(d_ef.get(), v->edge_face.data(), sizeof(int64_t) * (size_t)ne);
        XR_LAUNCH("vor_face_xy", k_vor_face_xy, dim3(div_up(ne, 256)), dim3(256), 0, v->centroids.get(), d_ef.get(), ne,
                  d_xy.get());
        d2h(v->b_edge_face_xy.data(), d_xy.get(), sizeof(double) * 2 * (size_t)ne);
    }
    v->boundary_ready = true;
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_voronoi_create(xr_mesh *mesh, xr_voronoi **out) {
    XR_API_BEGIN
    XR_REQUIRE(mesh && out, XR_ERR_INVALID, "xr_voronoi_create: NULL argument");
    xr_voronoi *v = new xr_voronoi();
    try {
        const int64_t N = mesh->n_node, F = mesh->n_face;
        const int m = mesh->m;
        const int64_t total = F * m;
        v->mesh = mesh; v->n_node = N; v->n_face = F;
        DevBuf<int32_t> count((size_t)N + 1), cursor((size_t)N + 1);
        v->indptr.alloc((size_t)N + 1);
        fill_i32(count.get(), 0, N + 1);
        fill_i32(cursor.get(), 0, N + 1);
        if (total > 0)
            XR_LAUNCH("vor_count", k_vor_count, dim3(div_up(total, 256)), dim3(256), 0, mesh->faces_raw.get(), total, count.get());
        exclusive_scan_i32(count.get(), v->indptr.get(), N);
        v->nnz = read_scalar(v->indptr.get() + N);
        v->faces_asc.alloc((size_t)v->nnz);
        v->faces_ccw.alloc((size_t)v->nnz);
        v->interior.alloc((size_t)N);
        v->cell_rank.alloc((size_t)N + 1);
        v->centroids.alloc((size_t)F * 2);
        DevBuf<uint8_t> on_boundary((size_t)N);
        DevBuf<int32_t> flag32((size_t)N), counters(4), e_lo((size_t)total), e_hi((size_t)total), e_face((size_t)total);
        XR_HIP(hipMemsetAsync(on_boundary.get(), 0, (size_t)(N > 0 ? N : 1), launch_stream()));
        const int32_t init[4] = {0, INT32_MAX, 0, 0}; // n_edges, min degree, max degree
        h2d(counters.get(), init, sizeof(init));
        if (total > 0) {
            XR_LAUNCH("vor_scatter", k_vor_scatter, dim3(div_up(total, 256)), dim3(256), 0, mesh->faces_raw.get(), total, m,
                      v->indptr.get(), cursor.get(), v->faces_asc.get());
            XR_LAUNCH("vor_sort_rows", k_vor_sort_rows, dim3(div_up(N, 256)), dim3(256), 0, v->indptr.get(), N,
                      v->faces_asc.get());
            XR_LAUNCH("vor_exterior", k_vor_exterior, dim3(div_up(total, 256)), dim3(256), 0, mesh->faces_raw.get(), F, m,
                      v->indptr.get(), v->faces_asc.get(), on_boundary.get(), counters.get(), e_lo.get(), e_hi.get(),
                      e_face.get());
        }
        mesh_centroids_dev(mesh, v->centroids.get());
        if (N > 0)
            XR_LAUNCH("vor_interior", k_vor_interior, dim3(div_up(N, 256)), dim3(256), 0, mesh->node_xy.get(),
                      v->centroids.get(), v->indptr.get(), v->faces_asc.get(), on_boundary.get(), N, v->faces_ccw.get(),
                      v->interior.get(), flag32.get(), counters.get() + 1);
        exclusive_scan_i32(flag32.get(), v->cell_rank.get(), N);
        int32_t h[4];
        d2h(h, counters.get(), sizeof(h));
        v->n_interior = read_scalar(v->cell_rank.get() + N);
        v->min_degree = v->n_interior > 0 ? h[1] : 0;
        v->max_degree = h[2];
        const int64_t ne = h[0];
        std::vector<int32_t> lo((size_t)ne), hi((size_t)ne), fc((size_t)ne);
        if (ne > 0) {
            d2h(lo.data(), e_lo.get(), sizeof(int32_t) * (size_t)ne);
            d2h(hi.data(), e_hi.get(), sizeof(int32_t) * (size_t)ne);
            d2h(fc.data(), e_face.get(), sizeof(int32_t) * (size_t)ne);
        }
        std::vector<int64_t> order((size_t)ne);
        for (int64_t i = 0; i < ne; i++) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (lo[(size_t)a] != lo[(size_t)b]) return lo[(size_t)a] < lo[(size_t)b];
            if (hi[(size_t)a] != hi[(size_t)b]) return hi[(size_t)a] < hi[(size_t)b];
            return fc[(size_t)a] < fc[(size_t)b];
        });
        v->edge_lo.resize((size_t)ne); v->edge_hi.resize((size_t)ne); v->edge_face.resize((size_t)ne);
        for (int64_t i = 0; i < ne; i++) {
            v->edge_lo[(size_t)i] = lo[(size_t)order[(size_t)i]];
            v->edge_hi[(size_t)i] = hi[(size_t)order[(size_t)i]];
            v->edge_face[(size_t)i] = fc[(size_t)order[(size_t)i]];
        }
    } catch (...) {
        delete v;
        throw;
    }
    *out = v;
    XR_API_END
}

int xr_voronoi_info(const xr_voronoi *v, int64_t *n_node, int64_t *nnz, int64_t *n_exterior_edge, int64_t *n_interior_cell,
                    int64_t *max_interior_degree) {
    XR_API_BEGIN
    XR_REQUIRE(v, XR_ERR_INVALID, "xr_voronoi_info: NULL handle");
    if (n_node) *n_node = v->n_node;
    if (nnz) *nnz = v->nnz;
    if (n_exterior_edge) *n_exterior_edge = (int64_t)v->edge_lo.size();
    if (n_interior_cell) *n_interior_cell = v->n_interior;
    if (max_interior_degree) *max_interior_degree = v->max_degree;
    XR_API_END
}

int xr_voronoi_download(const xr_voronoi *v, int64_t *indptr, int64_t *indices, int64_t *edge_nodes, int64_t *edge_face,
                        double *centroids) {
    XR_API_BEGIN
    XR_REQUIRE(v && indptr && (v->nnz == 0 || indices), XR_ERR_INVALID, "xr_voronoi_download: NULL argument");
    {
        DevBuf<int64_t> wide((size_t)std::max<int64_t>(v->n_node + 1, v->nnz));
        XR_LAUNCH("vor_widen", k_vor_widen, dim3(div_up(v->n_node + 1, 256)), dim3(256), 0, v->indptr.get(), v->n_node + 1,
                  wide.get());
        d2h(indptr, wide.get(), sizeof(int64_t) * (size_t)(v->n_node + 1));
        if (v->nnz > 0) {
            XR_LAUNCH("vor_widen", k_vor_widen, dim3(div_up(v->nnz, 256)), dim3(256), 0, v->faces_asc.get(), v->nnz,
                      wide.get());
            d2h(indices, wide.get(), sizeof(int64_t) * (size_t)v->nnz);
        }
    }
    const size_t ne = v->edge_lo.size();
    XR_REQUIRE(ne == 0 || (edge_nodes && edge_face), XR_ERR_INVALID, "xr_voronoi_download: NULL edge arrays");
    for (size_t i = 0; i < ne; i++) {
        edge_nodes[2 * i] = v->edge_lo[i];
        edge_nodes[2 * i + 1] = v->edge_hi[i];
        edge_face[i] = v->edge_face[i];
    }
    if (centroids && v->n_face > 0) d2h(centroids, v->centroids.get(), sizeof(double) * 2 * (size_t)v->n_face);
    XR_API_END
}

int xr_voronoi_boundary_info(xr_voronoi *v, int64_t *n_boundary_node, int64_t *n_boundary_entry) {
    XR_API_BEGIN
    XR_REQUIRE(v && n_boundary_node && n_boundary_entry, XR_ERR_INVALID, "xr_voronoi_boundary_info: NULL argument");
    voronoi_boundary(v);
    *n_boundary_node = (int64_t)v->b_nodes.size();
    *n_boundary_entry = (int64_t)v->b_faces.size();
    XR_API_END
}

int xr_voronoi_boundary(xr_voronoi *v, int64_t *nodes, int64_t *row_ptr, int64_t *faces, double *face_xy,
                        int64_t *edge_nodes, int64_t *edge_face, double *edge_face_xy) {
    XR_API_BEGIN
    XR_REQUIRE(v && row_ptr, XR_ERR_INVALID, "xr_voronoi_boundary: NULL argument");
    voronoi_boundary(v);
    const size_t nb = v->b_nodes.size(), nf = v->b_faces.size(), ne = v->edge_face.size();
    XR_REQUIRE((nb == 0 || nodes) && (nf == 0 || (faces && face_xy)) && (ne == 0 || (edge_nodes && edge_face && edge_face_xy)),
               XR_ERR_INVALID, "xr_voronoi_boundary: NULL output array");
    if (nb) memcpy(nodes, v->b_nodes.data(), sizeof(int64_t) * nb);
    memcpy(row_ptr, v->b_ptr.data(), sizeof(int64_t) * (nb + 1));
    if (nf) {
        memcpy(faces, v->b_faces.data(), sizeof(int64_t) * nf);
        memcpy(face_xy, v->b_face_xy.data(), sizeof(double) * 2 * nf);
    }
    for (size_t i = 0; i < ne; i++) {
        edge_nodes[2 * i] = v->edge_lo[i];
        edge_nodes[2 * i + 1] = v->edge_hi[i];
        edge_face[i] = v->edge_face[i];
    }
    if (ne) memcpy(edge_face_xy, v->b_edge_face_xy.data(), sizeof(double) * 2 * ne);
    XR_API_END
}

int xr_voronoi_mesh(const xr_voronoi *v, const double *extra_xy, int64_t n_extra_vertex, const int64_t *boundary_cells,
                    int64_t n_boundary_cell, int64_t n_max_boundary, xr_mesh **out) {
    XR_API_BEGIN
    XR_REQUIRE(v && out, XR_ERR_INVALID, "xr_voronoi_mesh: NULL argument");
    XR_REQUIRE(n_extra_vertex >= 0 && n_boundary_cell >= 0 && n_max_boundary >= 0, XR_ERR_INVALID,
               "xr_voronoi_mesh: negative sizes");
    XR_REQUIRE((n_extra_vertex == 0 || extra_xy) && (n_boundary_cell == 0 || boundary_cells), XR_ERR_INVALID,
               "xr_voronoi_mesh: NULL arrays");
    XR_REQUIRE(v->n_interior == 0 || v->min_degree >= 3, XR_ERR_INVALID,
               "xr_voronoi_mesh: an interior node is surrounded by only %d faces (non-manifold mesh)", v->min_degree);
    const int64_t n_vertex = v->n_face + n_extra_vertex;
    const int64_t n_cell = v->n_interior + n_boundary_cell;
    const int64_t m64 = std::max<int64_t>(std::max<int64_t>(v->max_degree, n_boundary_cell > 0 ? n_max_boundary : 0), 3);
    XR_REQUIRE(m64 <= XR_MAX_FACE_NODES, XR_ERR_LIMIT, "xr_voronoi_mesh: a Voronoi cell has %lld corners (limit %d)",
               (long long)m64, XR_MAX_FACE_NODES);
    XR_REQUIRE(n_vertex < ((int64_t)1 << 31) && n_cell * m64 < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_voronoi_mesh: mesh exceeds the int32 index range");
    const int m = (int)m64;
    const int mb = (int)n_max_boundary;
    std::vector<int32_t> table((size_t)(n_boundary_cell * mb) + 1);
    for (int64_t r = 0; r < n_boundary_cell; r++) {
        for (int j = 0; j < mb; j++) {
            const int64_t c = boundary_cells[r * mb + j];
            XR_REQUIRE(c >= -1 && c < n_vertex, XR_ERR_INVALID, "xr_voronoi_mesh: boundary cell %lld references vertex %lld",
                       (long long)r, (long long)c);
            XR_REQUIRE(c >= 0 || j >= 3, XR_ERR_INVALID, "xr_voronoi_mesh: boundary cell %lld has fewer than 3 corners",
                       (long long)r);
            table[(size_t)(r * mb + j)] = (int32_t)c;
        }
    }
    xr_mesh *mesh = new xr_mesh();
    try {
        mesh->n_node = n_vertex;
        mesh->n_face = n_cell;
        mesh->m = m;
        mesh->node_xy.alloc((size_t)n_vertex * 2);
        mesh->faces_raw.alloc((size_t)(n_cell * m));
        if (v->n_face > 0)
            XR_HIP(hipMemcpyAsync(mesh->node_xy.get(), v->centroids.get(), sizeof(double) * 2 * (size_t)v->n_face,
                                  hipMemcpyDeviceToDevice, launch_stream()));
        if (n_extra_vertex > 0)
            h2d(mesh->node_xy.get() + 2 * v->n_face, extra_xy, sizeof(double) * 2 * (size_t)n_extra_vertex);
        if (v->n_node > 0 && v->n_interior > 0)
            XR_LAUNCH("vor_cells", k_vor_cells, dim3(div_up(v->n_node, 256)), dim3(256), 0, v->indptr.get(),
                      v->faces_ccw.get(), v->interior.get(), v->cell_rank.get(), v->n_node, m, mesh->faces_raw.get());
        if (n_boundary_cell > 0) {
            DevBuf<int32_t> dtable((size_t)(n_boundary_cell * mb));
            h2d(dtable.get(), table.data(), sizeof(int32_t) * (size_t)(n_boundary_cell * mb));
            XR_LAUNCH("vor_boundary_cells", k_vor_boundary_cells, dim3(div_up(n_boundary_cell * m, 256)), dim3(256), 0,
                      dtable.get(), n_boundary_cell, mb, m, mesh->faces_raw.get() + v->n_interior * m);
            stream_sync();
        }
        stream_sync();
    } catch (...) {
        delete mesh;
        throw;
    }
    *out = mesh;
    XR_API_END
}

int xr_voronoi_destroy(xr_voronoi *v) {
    XR_API_BEGIN
    if (v) {
        stream_sync();
        delete v;
    }
    XR_API_END
}

} // extern "C"
