/*
 * xr_oracle.h -- CPU ORACLE for the xugrid regridding hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and there only as the checker / the
 * timed CPU baseline.  The product path (xugrid_amd/) never links, imports or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - apply kernel, reducers, quickselect, CSR assembly, area, centroids,
 *     replace_interpolated_weights: restated line-by-line from files that ARE in
 *     /root/reference and PINNED against golden vectors generated from those files
 *     (tests/golden/gen_goldens.py -> tests/golden/ npz files).
 *   - polygon clipping, bbox search, point location, barycentric weights, segment-in-face
 *     clipping (intersect_edges): the arithmetic
 *     lives in the third-party package numba_celltree 0.4.2 (pixi.lock:298), which is
 *     neither in /root/reference nor installed.  Restated from its published algorithm
 *     (Sutherland-Hodgman, bounding-box cell tree after Garth & Joy 2010, Wachspress
 *     coordinates, Cyrus-Beck line clipping).  PARITY UNPINNED for general polygon pairs; pinned only through the
 *     reference's own invariants/known answers (rectilinear overlap == overlap_1d goldens,
 *     self-overlap identity, barycentric known answers of tests/test_ugrid2d.py:751-791, the
 *     NetworkGridder known answers of tests/test_regrid/test_network_gridder.py)
 *     and an exact rational-arithmetic clip (oracle/exact_clip.py).
 *
 * All index arrays are int64 (np.intp in the reference, xugrid/constants.py:10),
 * all floats are float64 (constants.py:9), fill value is -1 (constants.py:28).
 */
#ifndef XR_ORACLE_H
#define XR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- method ids shared by oracle and tests (names: xugrid/regrid/reduce.py:254-272) ---- */
enum {
    XO_MEAN = 0, XO_HARMONIC_MEAN = 1, XO_GEOMETRIC_MEAN = 2, XO_SUM = 3, XO_MINIMUM = 4,
    XO_MAXIMUM = 5, XO_MODE = 6, XO_PERCENTILE = 7, XO_FIRST_ORDER_CONSERVATIVE = 8,
    XO_MAX_OVERLAP = 9
};

/* One reducer call: (values, weights, workspace) -> float; reduce.py:16-238. */
double xo_reduce(int method, double p, const double *values, const double *weights,
                 double *workspace, int64_t n);

/* make_regrid(func)._regrid, regridder.py:41-67.  source is (K, S) row-major,
 * out is (K, T).  parallel_rows = 0 -> threads over K only (as numba prange, :50);
 * 1 -> additionally over target rows ("favourable" CPU variant). */
int xo_regrid_csr(int method, double p, const double *source, int64_t K, int64_t S,
                  const double *data, const int64_t *indices, const int64_t *indptr,
                  int64_t T, double *out, int parallel_rows);

/* CentroidLocatorRegridder._regrid, regridder.py:400-409 (COO scatter). */
int xo_regrid_coo(const double *source, int64_t K, int64_t S, const int64_t *row,
                  const int64_t *col, int64_t nnz, int64_t T, double *out);

/* MatrixCOO.to_csr indptr, core/sparse.py:61-78. */
int xo_to_csr_indptr(const int64_t *row, int64_t nnz, int64_t n, int64_t *indptr);

/* connectivity.area (connectivity.py:615-633) and centroids (:636-664). */
int xo_area(const double *node_xy, const int64_t *faces, int64_t n_face, int64_t m, double *area);
int xo_centroids(const double *node_xy, const int64_t *faces, int64_t n_face, int64_t m, double *cxy);

/* replace_interpolated_weights, regrid/unstructured.py:17-57 (in place on weights). */
int xo_replace_interpolated_weights(const double *vertices, const int64_t *faces, int64_t m,
                                    const int64_t *face_index, double *weights, int64_t n,
                                    const int64_t *node_to_node_map, int64_t node_index_threshold);

/* ---- numba_celltree 0.4.2 restatement (external; parity unpinned, see header) ---- */
typedef struct xo_tree xo_tree;

/* CellTree2d(vertices, faces, fill_value): copies + CCW-normalises faces, builds bboxes and
 * the bounding-box cell tree (n_buckets=4, cells_per_leaf=2).  ugrid2d.py:915-921. */
xo_tree *xo_tree_create(const double *node_xy, int64_t n_node, const int64_t *faces,
                        int64_t n_face, int64_t m, int64_t fill_value);
void xo_tree_destroy(xo_tree *t);
int64_t xo_tree_n_face(const xo_tree *t);
/* copy of the CCW-normalised connectivity (n_face x m) */
int xo_tree_faces(const xo_tree *t, int64_t *faces_out);

/* CellTree2d.intersect_edges(edge_coords (n_edge, 2, 2)) -> (edge_index, face_index, intersections (n, 2, 2)),
 * unstructured.py:203-215: Cyrus-Beck clip of every edge against the convex faces its box reaches; pairs with a
 * positive-length piece only, ordered by (edge, face).  Two-phase like intersect_faces. */
int xo_intersect_edges_count(xo_tree *t, const double *edge_xy, int64_t n_edge, int64_t *n_found);
int xo_intersect_edges_fill(xo_tree *t, int64_t *edge_idx, int64_t *face_idx, double *intersections);

/* CellTree2d.intersect_faces(vertices, faces, fill_value), unstructured.py:124-132.
 * Two-phase: _count runs the whole search+clip and stores the result in the tree,
 * _fill copies it out.  Output is ordered by (query face, tree face).
 * use_sat: 1 = separating-axis pre-filter before clipping (as the reference), 0 = clip all
 * bbox candidates (the final `area > 0` filter makes both identical; tested).
 * n_candidates (optional) receives the number of bbox-overlap pairs. */
int xo_intersect_faces_count(xo_tree *t, const double *q_xy, int64_t q_n_node,
                             const int64_t *q_faces, int64_t q_n_face, int64_t q_m,
                             int64_t q_fill, int use_sat, int64_t *nnz, int64_t *n_candidates);
int xo_intersect_faces_fill(xo_tree *t, int64_t *query_idx, int64_t *tree_idx, double *area);

/* Brute-force O(T*S) variant of the same (no tree), for tiny cases: validates the tree. */
int xo_intersect_faces_bruteforce(xo_tree *t, const double *q_xy, int64_t q_n_node,
                                  const int64_t *q_faces, int64_t q_n_face, int64_t q_m,
                                  int64_t q_fill, int64_t cap, int64_t *query_idx,
                                  int64_t *tree_idx, double *area, int64_t *nnz);

/* area of clip(subject, clipper) for two explicit convex CCW polygons (n x 2 doubles). */
double xo_clip_area(const double *subject, int64_t ns, const double *clipper, int64_t nc);

/* CellTree2d.locate_points(points, tolerance); tolerance < 0 -> default
 * (1e-12 * max bbox diagonal, ugridbase.py:1165-1170).  -1 where not found.
 * Ties (point within tolerance of an edge shared by two faces): LOWEST face index. */
int xo_locate_points(const xo_tree *t, const double *pts, int64_t n, double tolerance,
                     int64_t *face_index);

/* CellTree2d.compute_barycentric_weights(points, tolerance), ugrid2d.py:1078.
 * weights is (n, m) row-major, zero where outside. */
int xo_barycentric(const xo_tree *t, const double *pts, int64_t n, double tolerance,
                   int64_t *face_index, double *weights);

double xo_default_tolerance(const xo_tree *t);

int xo_num_threads(void);
void xo_set_num_threads(int n); /* OpenMP team size of the following calls (bench.py: 1-thread baseline) */

#ifdef __cplusplus
}
#endif
#endif
