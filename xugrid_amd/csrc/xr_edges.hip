// xr_edges.hip -- line segments against the faces of a mesh: the weights of NetworkGridder.
//
// Replaces, in one device pass from edge coordinates to CSR,
//   numba_celltree.CellTree2d.intersect_edges(edge_coords) -> (edge_index, face_index, intersections[n, 2, 2])
// as called from UnstructuredGrid2d.intersection_length (xugrid/regrid/unstructured.py:203-215), the
//   length = norm(diff(intersections))                                                  (:212)
// that follows it, its argsort by face (:211) and MatrixCSR.from_triplet (xugrid/regrid/gridder.py:66-73).
//
// Per edge the hierarchical grid of the mesh is walked over the edge's bounding box (long edges: only the cells
// their segment can reach, one block per edge); every candidate face is clipped with the Cyrus-Beck parametric
// line clip against its CCW-normalised (convex) polygon.  A pair is kept iff the clipped parameter interval has
// t0 < t1 -- touching a corner or an edge from outside yields no entry (tests/test_regrid/test_network_gridder.py:
// nnz == 8 for the four-edge network on the 4 x 4 grid).  The arithmetic mirrors oracle/xr_oracle.c
// (cyrus_beck_clip) operation for operation; rows come out ordered by edge id.
#include "xr_objects.h"

namespace xr {

static constexpr int EDGE_BIG_CELLS = 192; // edges whose box covers more grid cells go to the block-per-edge kernel
static constexpr int ROW_SORT_SMALL = 48;  // rows up to this length are insertion-sorted by one thread
static constexpr int ROW_SORT_LDS = 4096;  // rows up to this length are sorted in LDS by one block

// a + t (b - a), t in [0, 1], against every half-plane of the CCW polygon.  -> length of the clipped piece, or -1
__device__ __forceinline__ double cyrus_beck_length(const double *__restrict__ poly, int n, P2 a, P2 b) {
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
    const double ex = dx - cx, ey = dy - cy;
    return sqrt(ex * ex + ey * ey);
}

// does the segment reach the closed box?  (slab test; conservative: used only to skip grid cells)
__device__ __forceinline__ bool segment_reaches_box(P2 a, P2 b, double x0, double x1, double y0, double y1) {
    double t0 = 0.0, t1 = 1.0;
    const double sx = b.x - a.x, sy = b.y - a.y;
    if (sx == 0.0) {
        if (a.x < x0 || a.x > x1) return false;
    } else {
        double u = (x0 - a.x) / sx, v = (x1 - a.x) / sx;
        if (u > v) { const double w = u; u = v; v = w; }
        t0 = fmax(t0, u);
        t1 = fmin(t1, v);
    }
    if (sy == 0.0) {
        if (a.y < y0 || a.y > y1) return false;
    } else {
        double u = (y0 - a.y) / sy, v = (y1 - a.y) / sy;
        if (u > v) { const double w = u; u = v; v = w; }
        t0 = fmax(t0, u);
        t1 = fmin(t1, v);
    }
    return t0 <= t1 + 1e-9; // (slack: a cell is only ever skipped when clearly unreachable)
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

// number of grid cells (all levels) the edge's box has to look at
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

// the records of one grid cell against the edge; HIT(face id, length) for every kept pair
template <typename Hit>
__device__ __forceinline__ void edge_cell(const EdgeBox &q, int r0, int r1, const float4 *__restrict__ rbb,
                                          const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
                                          int m, const int32_t *__restrict__ rec_face, Hit &&hit) {
    for (int r = r0; r < r1; r++) {
        const float4 bb = rbb[r];
        if (!(q.qx0 <= bb.y && bb.x <= q.qx1 && q.qy0 <= bb.w && bb.z <= q.qy1)) continue;
        const double len = cyrus_beck_length(rec_fxy + (int64_t)r * m * 2, rec_len[r], q.a, q.b);
        if (len > 0.0) hit(rec_face[r], len); // (a degenerate piece of zero length is no intersection)
    }
}

// FILL = false: count the hits per face;  FILL = true: place (edge, length) into the rows (arbitrary order)
template <bool FILL>
__global__ void __launch_bounds__(256)
k_edges(const double *__restrict__ edge_xy, int64_t n_edge, GridParams g, const int32_t *__restrict__ cell_start,
        const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len, int m,
        const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count, const int32_t *__restrict__ indptr,
        int32_t *__restrict__ indices, double *__restrict__ data, int32_t *__restrict__ big_list,
        int32_t *__restrict__ n_big) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edge) return;
    const EdgeBox q = load_edge(edge_xy, e, g);
    if (!(q.xmin == q.xmin && q.ymin == q.ymin && q.xmax == q.xmax && q.ymax == q.ymax)) return; // NaN coordinates
    if (edge_cells(q, g) > EDGE_BIG_CELLS) {
        if (!FILL) big_list[atomicAdd(n_big, 1)] = (int32_t)e;
        return;
    }
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    auto hit = [&](int face, double len) {
        const int k = atomicAdd(row_count + face, 1);
        if (FILL) {
            indices[indptr[face] + k] = (int32_t)e;
            data[indptr[face] + k] = len;
        }
    };
    for (int l = 0; l < g.n_levels; l++) {
        const double h = level_h(g, l), inv_h = level_inv_h(g, l);
        const int nx = g.nx[l], base = g.base[l];
        const int cx0 = cell_coord(q.xmin - h, g.x0, inv_h, nx), cx1 = cell_coord(q.xmax, g.x0, inv_h, nx);
        const int cy0 = cell_coord(q.ymin - h, g.y0, inv_h, g.ny[l]), cy1 = cell_coord(q.ymax, g.y0, inv_h, g.ny[l]);
        for (int cy = cy0; cy <= cy1; cy++)
            edge_cell(q, cell_start[base + cy * nx + cx0], cell_start[base + cy * nx + cx1 + 1], rbb, rec_fxy, rec_len, m,
                      rec_face, hit);
    }
}

// long edges: one block per edge, the threads stride over the cells of its box and skip those the segment cannot
// reach (a record lies within [cell origin, cell origin + 2 h) in both directions)
template <bool FILL>
__global__ void __launch_bounds__(256)
k_edges_big(const double *__restrict__ edge_xy, GridParams g, const int32_t *__restrict__ cell_start,
            const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const uint8_t *__restrict__ rec_len,
            int m, const int32_t *__restrict__ rec_face, int32_t *__restrict__ row_count,
            const int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data,
            const int32_t *__restrict__ big_list, const int32_t *__restrict__ n_big) {
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const int nb = *n_big;
    for (int i = blockIdx.x; i < nb; i += gridDim.x) {
        const int64_t e = big_list[i];
        const EdgeBox q = load_edge(edge_xy, e, g);
        auto hit = [&](int face, double len) {
            const int k = atomicAdd(row_count + face, 1);
            if (FILL) {
                indices[indptr[face] + k] = (int32_t)e;
                data[indptr[face] + k] = len;
            }
        };
        for (int l = 0; l < g.n_levels; l++) {
            const double h = level_h(g, l), inv_h = level_inv_h(g, l);
            const int nx = g.nx[l], base = g.base[l];
            const int cx0 = cell_coord(q.xmin - h, g.x0, inv_h, nx), cx1 = cell_coord(q.xmax, g.x0, inv_h, nx);
            const int cy0 = cell_coord(q.ymin - h, g.y0, inv_h, g.ny[l]), cy1 = cell_coord(q.ymax, g.y0, inv_h, g.ny[l]);
            const int w = cx1 - cx0 + 1;
            const int64_t cells = (int64_t)w * (cy1 - cy0 + 1);
            for (int64_t c = threadIdx.x; c < cells; c += 256) {
                const int cy = cy0 + (int)(c / w), cx = cx0 + (int)(c % w);
                const int r0 = cell_start[base + cy * nx + cx], r1 = cell_start[base + cy * nx + cx + 1];
                if (r0 == r1) continue;
                const double bx = g.x0 + cx * h, by = g.y0 + cy * h;
                // (the first / last cell of a level also holds the records clamped into it)
                const bool edge_cell_of_grid = cx == 0 || cy == 0 || cx == nx - 1 || cy == g.ny[l] - 1;
                if (!edge_cell_of_grid && !segment_reaches_box(q.a, q.b, bx, bx + 2.0 * h, by, by + 2.0 * h)) continue;
                edge_cell(q, r0, r1, rbb, rec_fxy, rec_len, m, rec_face, hit);
            }
        }
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

static void edge_length_csr(xr_mesh *tree, const double *edge_xy_host, int64_t n_edge, xr_csr *csr) {
    const int64_t F = tree->n_face;
    csr->n = F;
    csr->m = n_edge;
    csr->nnz = 0;
    csr->indptr.alloc((size_t)F + 1);
    hipStream_t st = engine().stream;
    if (F == 0 || n_edge == 0) {
        XR_HIP(hipMemsetAsync(csr->indptr.get(), 0, sizeof(int32_t) * ((size_t)F + 1), st));
        csr->indices.alloc(0);
        csr->data.alloc(0);
        return;
    }
    mesh_prepare(tree, false);
    mesh_build_index(tree);
    DevBuf<double> edge_xy((size_t)n_edge * 4);
    h2d(edge_xy.get(), edge_xy_host, sizeof(double) * 4 * (size_t)n_edge);
    DevBuf<int32_t> row_count((size_t)F), big_list((size_t)n_edge), counters(4);
    XR_HIP(hipMemsetAsync(row_count.get(), 0, sizeof(int32_t) * (size_t)F, st));
    XR_HIP(hipMemsetAsync(counters.get(), 0, sizeof(int32_t) * 4, st));
    const GridParams &g = tree->grid;
    const int big_grid = engine().num_cu * 4;
    XR_LAUNCH("edges_count", k_edges<false>, dim3(div_up(n_edge, 256)), dim3(256), 0, edge_xy.get(), n_edge, g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->m,
              tree->rec_face.get(), row_count.get(), (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr,
              big_list.get(), counters.get());
    XR_LAUNCH("edges_big_count", k_edges_big<false>, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->m,
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
    XR_LAUNCH("edges_fill", k_edges<true>, dim3(div_up(n_edge, 256)), dim3(256), 0, edge_xy.get(), n_edge, g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->m,
              tree->rec_face.get(), row_count.get(), csr->indptr.get(), csr->indices.get(), csr->data.get(),
              big_list.get(), counters.get());
    XR_LAUNCH("edges_big_fill", k_edges_big<true>, dim3(big_grid), dim3(256), 0, edge_xy.get(), g,
              tree->cell_start.get(), tree->rec_bb.get(), tree->rec_fxy.get(), tree->rec_len.get(), tree->m,
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

} // extern "C"
