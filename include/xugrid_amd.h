/*
 * xugrid_amd.h -- C ABI of the MI355X-native regridding engine (libxugrid_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of Deltares/xugrid 0.15.3: regridding weight
 * construction + sparse weight x data apply.  xugrid has no FFI of its own;
 * the path sits behind three Python seams, and every entry point below names the seam it
 * replaces (paths relative to the reference tree):
 *
 *   seam 1  the tree object returned by Ugrid2d.celltree        xugrid/ugrid/ugrid2d.py:908-921
 *           (external numba_celltree.CellTree2d) and its methods
 *             .intersect_faces(vertices, faces, fill_value)     xugrid/regrid/unstructured.py:124-132
 *             .locate_points(points, tolerance)                 xugrid/regrid/unstructured.py:139,189
 *             .compute_barycentric_weights(points, tolerance)   xugrid/ugrid/ugrid2d.py:1078
 *   seam 2  the apply callable  self._regrid(source(K,S), A, size)  xugrid/regrid/regridder.py:41-67,
 *           looked up at :124-141, invoked at :187; COO variant :400-409
 *   seam 3  MatrixCSR.from_triplet(target, source, w, n, m)     xugrid/regrid/regridder.py:433-435,
 *           xugrid/core/sparse.py:61-78,119-127
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns an int
 *     status: 0 = ok, < 0 = error (message via xr_last_error(), thread-local).  No exception or
 *     abort crosses the boundary.
 *   - Host arrays are caller-owned numpy-style buffers: float64 coordinates/weights, int64
 *     indices (np.intp, xugrid/constants.py:10), fill value -1 after ingestion.
 *     On the device the engine keeps int32 connectivity/indices and float64 geometry.
 *   - Functions with the suffix _dev take DEVICE pointers (HBM addresses on the current device,
 *     e.g. torch tensor .data_ptr()) and do no host transfer; they run on the engine stream and
 *     synchronise it before returning unless stated otherwise.
 *   - Handles (xr_mesh, xr_csr) own HBM; they are not tied to a host thread.  Re-entrancy (dask calls
 *     the apply seam from several threads, regridder.py:177-185): the apply entry points
 *     xr_apply_csr[_dev], which only READ a weight matrix, run concurrently, each calling thread on a
 *     HIP stream of its own; every other entry point (building or mutating a handle, and the
 *     partial-state entries of the multi-GPU split, one thread per process by design) is exclusive:
 *     callable from any thread, serialised inside the library.
 *   - There is NO CPU fallback: without a HIP device every compute entry point fails with
 *     XR_ERR_NO_DEVICE.
 */
#ifndef XUGRID_AMD_H
#define XUGRID_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XR_OK 0
#define XR_ERR_INVALID (-1)   /* bad argument (maps to ValueError/TypeError on the Python side) */
#define XR_ERR_NO_DEVICE (-2) /* no HIP device / HIP runtime error at init */
#define XR_ERR_HIP (-3)       /* HIP runtime error during a call */
#define XR_ERR_LIMIT (-4)     /* size limit exceeded (int32 index range, max vertices per face) */

/* Reducer ids.  Names and semantics: xugrid/regrid/reduce.py:16-272. */
#define XR_MEAN 0
#define XR_HARMONIC_MEAN 1
#define XR_GEOMETRIC_MEAN 2
#define XR_SUM 3
#define XR_MINIMUM 4
#define XR_MAXIMUM 5
#define XR_MODE 6
#define XR_PERCENTILE 7 /* + percentile argument; median = 50 */
#define XR_FIRST_ORDER_CONSERVATIVE 8 /* = conductance */
#define XR_MAX_OVERLAP 9
/* Not a reduce.py reducer: the value of the row's LAST entry, NaN included, weights ignored -- the COO scatter
 * of CentroidLocatorRegridder._regrid (xugrid/regrid/regridder.py:400-409: out[k, row] = source[k, col], later
 * entries overwrite earlier ones) expressed on CSR rows, so that locator weights stay in HBM too. */
#define XR_SELECT 10

/* Source data dtypes accepted by the apply seam (output is always float64, regridder.py:44). */
#define XR_F64 0
#define XR_F32 1

#define XR_MAX_FACE_NODES 32 /* numba_celltree MAX_N_VERTEX */

typedef struct xr_mesh xr_mesh; /* device-resident Ugrid2d face topology + spatial index */
typedef struct xr_csr xr_csr;   /* device-resident MatrixCSR weights */
typedef struct xr_outer xr_outer; /* separable (rectilinear) weights in factored form */

/* ---- engine ---------------------------------------------------------------------------- */
const char *xr_last_error(void);
/* Number of HIP devices (0 if none / no runtime). */
int xr_device_count(int *count);
/* Bind the engine (this process) to one device; creates the engine stream.  One process per
 * GPU is the multi-GPU model (torch.distributed ranks call this with LOCAL_RANK).  Calling any
 * compute entry point before xr_init implies xr_init(0). */
int xr_init(int device);
int xr_current_device(int *device);
/* Release cached HBM held by the engine's block pool. */
int xr_trim_pool(void);
int xr_version(void);
/* Share a HIP stream with the caller (e.g. PyTorch's current stream, `torch.cuda.current_stream().cuda_stream`;
 * the null stream is a valid handle): with external != 0 every kernel, copy and event of the engine goes to
 * `hip_stream`, so device work of caller and engine is ordered without host synchronisation, and with
 * async_dev != 0 the *_dev entry points return as soon as their kernels are enqueued.  external == 0 restores the
 * engine's own stream.  The multi-GPU layer uses this between its kernels and the RCCL collectives. */
int xr_set_stream(void *hip_stream, int external, int async_dev);
/* Asynchronous mode on the engine's OWN stream (on != 0): the *_dev entry points, xr_overlap_apply_dev and
 * xr_mesh_invalidate return with their kernels still in flight; results are complete after xr_dev_sync() or any call that
 * returns data to the host.  All work then stays on the one engine stream (no per-thread lanes): calls from several host
 * threads are safe but take turns -- each is enqueued whole before the next starts -- instead of running on streams of their
 * own (the same holds after xr_set_stream).  For pipelines that keep their data on the device and issue many calls back to
 * back (a time loop; bench.py's step). */
int xr_set_async(int on);
/* Run-time options of the library -- the switches tests and measurement scripts flip (no reference interface: xugrid has no
 * such knobs).  None changes a result, except "apply_contract" (documented there).  Names and meanings: DESIGN.md section 8 /
 * xugrid_amd/csrc/xr_internal.h (enum Option); the value an option starts with is read ONCE from the environment variable
 * XR_<NAME IN CAPITALS> when the library is first used.  Unknown names are an error (XR_ERR_INVALID).  The calls are
 * thread-safe; an option changed while another thread is inside a call takes effect for that call or the next. */
int xr_set_option(const char *name, int64_t value);
int xr_get_option(const char *name, int64_t *value);
/* Host-only helpers of the Python layer (no device, usable without one): the copies xugrid's constructors make
 * (Ugrid2d.__init__, xugrid/ugrid/ugrid2d.py:86-96: contiguous node_x / node_y, face_node_connectivity.copy()) done by the
 * library's host thread pool.  xr_host_interleave2: out_xy[i] = (x[i * x_stride], y[i * y_stride]), strides in elements. */
int xr_host_copy(void *dst, const void *src, int64_t bytes);
int xr_host_interleave2(const double *x, int64_t x_stride, const double *y, int64_t y_stride, int64_t n, double *out_xy);

/* ---- seam 1: mesh handle = CellTree2d(vertices, faces, fill_value) --------------------- */
/* ugrid2d.py:915-921.  node_xy: float64[n_node,2] (node_coordinates, ugridbase.py:576-579);
 * faces: int32 or int64 [n_face, n_max_node] dense face_node_connectivity with `fill_value`
 * in unused slots (faces_itemsize = 4 or 8).  Uploads the raw arrays only; the derived state
 * (CCW-normalised connectivity, per-face length/bbox/area, spatial index) is built on the
 * device lazily by the first call that needs it, or explicitly by xr_mesh_prepare /
 * xr_mesh_build_index. */
int xr_mesh_create(const double *node_xy, int64_t n_node, const void *faces, int faces_itemsize,
                   int64_t n_face, int64_t n_max_node, int64_t fill_value, xr_mesh **out);
/* The same from arrays that are ALREADY in HBM of the engine's device (a multi-GPU rank that cut its shard out of the
 * replicated mesh on the device, xugrid_amd/distributed.py; any pipeline that produced the mesh there).  node_xy_dev:
 * float64[n_node,2]; faces_dev: int32 or int64 [n_face, n_max_node].  Both are COPIED (the handle owns its arrays; the
 * copy of the connectivity is the same fill -> -1 / narrowing / validation pass xr_mesh_create runs).  The caller's arrays
 * must be complete in the engine's stream order (xr_set_stream: the caller's own stream) or on the host's view
 * (after a device synchronisation). */
int xr_mesh_create_dev(const double *node_xy_dev, int64_t n_node, const void *faces_dev, int faces_itemsize,
                       int64_t n_face, int64_t n_max_node, int64_t fill_value, xr_mesh **out);
/* The quad mesh of a rectilinear grid, generated on the device: Ugrid2d.from_structured_bounds ->
 * _from_intervals_helper (xugrid/ugrid/ugrid2d.py:1973-2034, :1894-1912), which is how a raster enters the polygon
 * path when the other grid is unstructured (StructuredGrid2d.convert_to, regrid/structured.py:489-501).
 * x_vertices float64[nx + 1], y_vertices float64[ny + 1]: the cell edges in the raster's own order (ascending or
 * descending).  Nodes in meshgrid order (node id = j * (nx + 1) + i), face id = j * nx + i, corners
 * counter-clockwise for ascending axes.  Only the two 1-D arrays cross PCIe. */
int xr_mesh_create_rectilinear(const double *x_vertices, int64_t nx, const double *y_vertices, int64_t ny,
                               xr_mesh **out);
int xr_mesh_destroy(xr_mesh *mesh);
int xr_mesh_info(const xr_mesh *mesh, int64_t *n_node, int64_t *n_face, int64_t *n_max_node);
/* HBM currently held by the handle (raw arrays, prepared arrays, query order, tree index).  Vertex blocks of meshes with
 * more than 4 nodes per face are stored flat with offsets (sum of the real polygon lengths, not n_face * n_max_node). */
int xr_mesh_device_bytes(const xr_mesh *mesh, int64_t *bytes);
/* Per-face preparation on the device: fill->-1, polygon length, CCW normalisation, bbox, area. */
int xr_mesh_prepare(xr_mesh *mesh);
/* Spatial index over the faces of this mesh (the "tree" side): hierarchical uniform grid. */
int xr_mesh_build_index(xr_mesh *mesh);
/* Drop all derived state (prepare + index + the face centroids a locate / barycentric call left on the mesh -- the
 * reference caches Ugrid2d.centroids on the grid the same way); used by benchmarks to time the whole path. */
int xr_mesh_invalidate(xr_mesh *mesh);
/* Face areas, connectivity.area (xugrid/ugrid/connectivity.py:615-633) -> float64[n_face]. */
int xr_mesh_area(xr_mesh *mesh, double *area_out);
/* Face centroids, connectivity.centroids (connectivity.py:636-664) -> float64[n_face,2]. */
int xr_mesh_centroids(xr_mesh *mesh, double *centroids_out);
/* CCW-normalised connectivity as held on the device -> int64[n_face, n_max_node]. */
int xr_mesh_faces(xr_mesh *mesh, int64_t *faces_out);
/* The arrays the handle was created from: node_xy float64[n_node, 2] and the connectivity int64[n_face, m] in
 * the caller's vertex order, fill -1 (either pointer may be NULL).  Mainly for meshes that were built on the
 * device (xr_voronoi_mesh). */
int xr_mesh_download(xr_mesh *mesh, double *node_xy_out, int64_t *faces_out);

/* CellTree2d.intersect_faces (unstructured.py:124-132) + relative normalisation (:133-134)
 * + MatrixCSR.from_triplet (regridder.py:433-435): all (query face, tree face) pairs with
 * intersection area > 0, as a CSR matrix with one row per QUERY (= regridding target) face,
 * columns = TREE (= source) faces sorted ascending within a row, data = overlap area, or
 * area / tree-face area if relative != 0.  The result stays in HBM.
 * As in numba_celltree a pair only counts if the two faces' exact bounding boxes overlap strictly and the separating-axis
 * test does not separate them: faces that merely touch (a mesh against itself, meshes sharing nodes) give no entry even
 * where the clip's floating-point result is a sliver of rounding dust. */
int xr_overlap(xr_mesh *tree, xr_mesh *query, int relative, xr_csr **out);
/* Statistics of the last xr_overlap on this tree: bbox candidate pairs tested by the clipper. */
int xr_overlap_stats(const xr_mesh *tree, int64_t *n_candidates);
/* Weights AND their first use in one call: OverlapRegridder(source, target, method).regrid(data) -- constructor
 * (regridder.py:505-512 -> _compute_weights :428-436) followed by regrid (:212-262 -> _regrid :41-67).  Same results as
 * xr_overlap + xr_apply_csr_dev; the matrix is returned for further applies.  For one variable and a streaming reducer the
 * apply is enqueued before the host has read the sizes of the matrix back, so the build's one host round trip hides behind
 * the apply kernel instead of separating the two.  source_dev (K, S) / out_dev (K, T) are device pointers. */
int xr_overlap_apply_dev(xr_mesh *tree, xr_mesh *query, int relative, int method, double percentile,
                         const void *source_dev, int source_dtype, int64_t K, double *out_dev, xr_csr **out);

/* CellTree2d.locate_points(points, tolerance) (unstructured.py:139,189; ugridbase.py:1323).
 * points float64[n,2] -> face index int64[n], -1 = not found.  tolerance < 0 selects the
 * default 1e-12 x max bbox diagonal (ugridbase.py:1165-1170).  A point within tolerance of an
 * edge shared by several faces is assigned to the LOWEST face index. */
int xr_locate_points(xr_mesh *mesh, const double *points, int64_t n, double tolerance,
                     int64_t *face_index_out);
/* Ugrid2d.rasterize_like (xugrid/ugrid/ugrid2d.py:1080-1100): locate_points on the nodes of a raster given by its
 * two 1-D coordinate arrays; the ny * nx points are generated on the device (point j * nx + i = (x[i], y[j])),
 * face_index_out int64[ny * nx]. */
int xr_locate_raster(xr_mesh *mesh, const double *x, int64_t nx, const double *y, int64_t ny,
                     double tolerance, int64_t *face_index_out);
/* CellTree2d.compute_barycentric_weights(points, tolerance) (ugrid2d.py:1054-1078).
 * -> face index int64[n] and float64[n, n_max_node] generalized barycentric weights
 * (all zero, face -1 when outside). */
int xr_barycentric(xr_mesh *mesh, const double *points, int64_t n, double tolerance,
                   int64_t *face_index_out, double *weights_out);

/* replace_interpolated_weights(vertices, faces, face_index, weights, node_to_node_map, node_index_threshold)
 * (xugrid/regrid/unstructured.py:17-57) as a stand-alone call: the device kernel xr_barycentric_csr runs inside its
 * pipeline, on caller-owned arrays.  vertices float64[n_vertex, 2]; faces int64[n_face, m] (-1 fill behind the
 * corners); face_index int64[n] (-1 = point in no cell: row untouched); weights float64[n, m] row-major, updated
 * IN PLACE as the reference does; node_to_node_map int64[n_map, 2]; node_index_threshold = n_vertex - n_map. */
int xr_replace_interpolated_weights(const double *vertices, int64_t n_vertex, const int64_t *faces, int64_t n_face,
                                    int64_t m, const int64_t *face_index, double *weights, int64_t n,
                                    const int64_t *node_to_node_map, int64_t n_map);

/* UnstructuredGrid2d.locate_centroids (xugrid/regrid/unstructured.py:137-144) + MatrixCOO.from_triplet
 * (regridder.py:386-398) without leaving the device: one row per query point holding (face containing it, 1.0),
 * no entry when the point is in no face.  Query points: `points` float64[n, 2] (query == NULL) or the face
 * centroids of the mesh `query` (points == NULL).  Apply with XR_SELECT. */
int xr_locate_csr(xr_mesh *tree, xr_mesh *query, const double *points, int64_t n, double tolerance,
                  xr_csr **out);

/* The whole of UnstructuredGrid2d.barycentric after the Voronoi pre-step (xugrid/regrid/unstructured.py:166-201),
 * assembled on the device: compute_barycentric_weights of the query points in the centroidal Voronoi mesh
 * `voronoi` (ugrid2d.py:1054-1078), replace_interpolated_weights (unstructured.py:17-57) for the synthetic exterior
 * vertices (the last n_extra Voronoi vertices; node_to_node_map int64[n_extra, 2]), zero rows for points outside
 * `source` (locate_points == -1 with the default tolerance, :188-190), weights > 0 only (:191-198), Voronoi
 * vertices mapped to source faces through vertex_face int64[n_vertex] (node_to_face_index).  The query points are
 * either `points` float64[n, 2] (query == NULL) or the face centroids of the mesh `query` (points == NULL; computed
 * on the device, nothing crosses PCIe).  Result: MatrixCSR.from_triplet(target, source, weights) (regridder.py:634-636)
 * with n rows (points) and source->n_face columns, entries of a row in Voronoi-vertex slot order. */
int xr_barycentric_csr(xr_mesh *voronoi, xr_mesh *source, xr_mesh *query, const double *points,
                       int64_t n, double tolerance, const int64_t *vertex_face,
                       const int64_t *node_to_node_map, int64_t n_extra, xr_csr **out);
/* the same with vertex_face given only for the vertices >= n_identity (vertex v < n_identity, a face centroid,
 * belongs to source face v): spares the host an O(n) array and its upload.
 * The weight of slot j of a cell is paired with vertex j of the cell in the CALLER's vertex order, exactly as the
 * reference does (unstructured.py:175,193) -- also in xr_barycentric_csr above.  tree_order = 1 (opt-in, NOT the
 * reference's result): paired with vertex j in the TREE's own counter-clockwise-normalised order, the order the
 * weights were computed in.  The two differ only for cells the tree stores reversed (clockwise cells, or concave
 * exterior cells that start at a reflex corner). */
int xr_barycentric_csr_tail(xr_mesh *voronoi, xr_mesh *source, xr_mesh *query, const double *points, int64_t n,
                            double tolerance, int64_t n_identity, const int64_t *vertex_face_tail,
                            const int64_t *node_to_node_map, int64_t n_extra, int tree_order, xr_csr **out);

/* The source-side half of UnstructuredGrid2d.barycentric -- the query points (points, or the face centroids of `query`,
 * unstructured.py:147) and `grid.locate_points(points) == -1` with the default tolerance (:188-190) -- as a device-resident
 * handle whose kernels run on the engine's side stream: they need nothing of the Voronoi tessellation, so they run beside
 * the (latency-bound) kernels of the Voronoi pre-step instead of behind them (1M faces / 4M points: construction 3.3 -> 2.96
 * ms).  Since round 5 they are DEFERRED: enqueued not by this call but by whichever call needs them or has the device idle
 * next -- xr_voronoi_create / xr_voronoi_mesh_auto where their host round trips begin, xr_barycentric_csr_points at the
 * latest.  LIFETIME: the handle keeps raw pointers to `source` and `query`; both meshes must stay alive until
 * xr_barycentric_csr_points has consumed the handle or xr_points_destroy has released it.  xr_mesh_invalidate and
 * xr_mesh_destroy of either mesh launch the deferred kernels first (they read the arrays as they are at that moment), so a
 * mesh that is invalidated or destroyed in between is safe, a handle used AFTER its source was destroyed is not.
 * xr_barycentric_csr_points is xr_barycentric_csr_tail on such a handle; it joins the side stream before it reads the flags. */
typedef struct xr_points xr_points;
int xr_locate_flags_begin(xr_mesh *source, xr_mesh *query, const double *points, int64_t n, xr_points **out);
int xr_points_destroy(xr_points *points);
int xr_barycentric_csr_points(xr_mesh *voronoi, xr_mesh *source, xr_points *points, double tolerance, int64_t n_identity,
                              const int64_t *vertex_face_tail, const int64_t *node_to_node_map, int64_t n_extra,
                              int tree_order, xr_csr **out);

/* ---- Voronoi pre-step of BarycentricInterpolator (xugrid/ugrid/voronoi.py:330-458 as called from
 * xugrid/regrid/unstructured.py:151-165: add_exterior, add_vertices, skip_concave) ---------------------
 * xr_voronoi_create does the O(n) part on the device: the node -> face inversion
 * (ugrid2d.py:700-713, connectivity.py:247-259), the exterior edges (the only part of edge_node_connectivity /
 * edge_face_connectivity the Voronoi step reads, voronoi.py:77-97; ugrid2d.py:497-509, :661-677) and the cells
 * of all nodes that touch no exterior edge: the centroids of the surrounding faces ordered counter-clockwise
 * about the node (voronoi.py:355-372).  The cells of the boundary nodes (projections on the exterior edges, one extra
 * corner per boundary node, the convexity choice: voronoi.py:59-327) are O(boundary) work on a few KB the device gathers:
 * done by the library itself in native host code (xr_voronoi_mesh_auto; a dozen dependent sorts / scans over a few thousand
 * items take tens of microseconds there, as device kernels each would be a launch and a round trip) or by the caller
 * (xr_voronoi_boundary -> xr_voronoi_mesh).  The Voronoi tessellation is assembled as a device-resident mesh:
 *   vertices = [face centroids ; extra_xy]      cells = [interior nodes ascending ; boundary_cells]. */
typedef struct xr_voronoi xr_voronoi;
int xr_voronoi_create(xr_mesh *mesh, xr_voronoi **out); /* `mesh` must outlive the handle */
int xr_voronoi_info(const xr_voronoi *v, int64_t *n_node, int64_t *nnz, int64_t *n_exterior_edge,
                    int64_t *n_interior_cell, int64_t *max_interior_degree);
/* node_face_connectivity as CSR (indptr int64[n_node+1], indices int64[nnz], faces ascending per node), the
 * exterior edges in lexicographic order (edge_nodes int64[n_edge, 2] = (lower, higher node id), edge_face
 * int64[n_edge]) and, optionally, the face centroids float64[n_face, 2]. */
int xr_voronoi_download(const xr_voronoi *v, int64_t *indptr, int64_t *indices, int64_t *edge_nodes,
                        int64_t *edge_face, double *centroids);
/* The same information restricted to what the O(boundary) host part reads -- nothing of size O(n) crosses PCIe:
 * the boundary nodes (end points of exterior edges, ascending), their rows of node_face_connectivity as a small
 * CSR (row_ptr int64[n_boundary_node + 1], faces int64[n_boundary_entry]) with the centroid of every listed face
 * (face_xy float64[n_boundary_entry, 2]), the exterior edges as above and the centroid of each edge's face. */
int xr_voronoi_boundary_info(xr_voronoi *v, int64_t *n_boundary_node, int64_t *n_boundary_entry);
int xr_voronoi_boundary(xr_voronoi *v, int64_t *nodes, int64_t *row_ptr, int64_t *faces, double *face_xy,
                        int64_t *edge_nodes, int64_t *edge_face, double *edge_face_xy);
/* extra_xy float64[n_extra_vertex, 2]: projections and substitute vertices (ids n_face ...);
 * boundary_cells int64[n_boundary_cell, n_max_boundary], -1 padded, counter-clockwise. */
int xr_voronoi_mesh(const xr_voronoi *v, const double *extra_xy, int64_t n_extra_vertex,
                    const int64_t *boundary_cells, int64_t n_boundary_cell, int64_t n_max_boundary,
                    xr_mesh **out);
/* The whole pre-step in one call: boundary cells by the library, assembly on the device.  n_tail = vertices added behind
 * the n_face centroids, n_map = rows of the interpolation map (voronoi.py:interpolation_map); xr_voronoi_tail copies
 * tail_face_index int64[n_tail] (source face of every added vertex, -1 for the per-node extra corners) and
 * interpolation_map int64[n_map, 2] (global vertex ids).  xr_voronoi_boundary_cells[_info]: the boundary cells themselves
 * (extra_xy float64[n_extra_vertex, 2], cells int64[n_cell, n_max], -1 padded) as xr_voronoi_mesh expects them. */
int xr_voronoi_mesh_auto(xr_voronoi *v, xr_mesh **out, int64_t *n_tail, int64_t *n_map);
int xr_voronoi_tail(xr_voronoi *v, int64_t *tail_face_index, int64_t *interpolation_map);
int xr_voronoi_boundary_cells_info(xr_voronoi *v, int64_t *n_extra_vertex, int64_t *n_cell, int64_t *n_max);
int xr_voronoi_boundary_cells(xr_voronoi *v, double *extra_xy, int64_t *cells);
int xr_voronoi_destroy(xr_voronoi *v);

/* ---- seam 3: MatrixCSR handle ----------------------------------------------------------- */
int xr_csr_info(const xr_csr *csr, int64_t *n, int64_t *m, int64_t *nnz);
/* Copy out as the reference's MatrixCSR fields (core/sparse.py:81-137): data float64[nnz],
 * indices int64[nnz], indptr int64[n+1]. */
int xr_csr_download(const xr_csr *csr, double *data, int64_t *indices, int64_t *indptr);
/* Build a device CSR from host MatrixCSR fields (from_weights / from_dataset path,
 * regridder.py:299-312,334-348). */
int xr_csr_upload(const double *data, const int64_t *indices, const int64_t *indptr, int64_t n,
                  int64_t m, int64_t nnz, xr_csr **out);
/* MatrixCSR.from_triplet(row, col, data, n, m): rows must be non-decreasing
 * (core/sparse.py:65); indptr = [0, cumsum(bincount(row, minlength=n))] computed on device. */
int xr_csr_from_triplet(const int64_t *row, const int64_t *col, const double *data, int64_t nnz,
                        int64_t n, int64_t m, xr_csr **out);
/* Separable (rectilinear) weights: StructuredGrid2d.broadcast_sorted (xugrid/regrid/structured.py:503-531 =
 * utils.broadcast, regrid/utils.py:17-36, followed by an argsort on the target index) for the per-axis
 * triplets of StructuredGrid1d.overlap / linear_weights (structured.py:335-356, :379-403).  Each axis is a
 * small sparse matrix in CSR form: indptr int64[n_target+1], source index int64[nnz] (ascending within a
 * row) and weight float64[nnz].  The result is the CSR of their outer product: row jt * n_target_x + it holds
 * the entries (source_y * n_source_x + source_x, weight_y * weight_x), ordered by column id, assembled on the
 * device without a sort. */
int xr_csr_from_outer(const int64_t *indptr_y, const int64_t *source_y, const double *weight_y,
                      int64_t n_target_y, int64_t n_source_y, const int64_t *indptr_x,
                      const int64_t *source_x, const double *weight_x, int64_t n_target_x,
                      int64_t n_source_x, xr_csr **out);
/* The same separable weights kept in FACTORED form (the two per-axis sparse matrices, a few hundred KB for
 * 4000 x 4000 grids) instead of their P_y * P_x product: xr_apply_outer reduces each target cell by walking its
 * c_y x c_x entries in the order of the product's CSR row (= the order of the reference's loop,
 * xugrid/regrid/structured.py:503-601 -> regrid/reduce.py), forming every weight w_y * w_x on the fly, so rows of
 * ANY length are reduced sequentially and bit-identically to the reference, and the apply reads the source data
 * once instead of the 12 bytes per entry of a stored matrix.  mode / percentiles (which need a row's values side by
 * side) and xr_outer_csr materialise the product once and keep it inside the handle. */
int xr_outer_create(const int64_t *indptr_y, const int64_t *source_y, const double *weight_y,
                    int64_t n_target_y, int64_t n_source_y, const int64_t *indptr_x,
                    const int64_t *source_x, const double *weight_x, int64_t n_target_x,
                    int64_t n_source_x, xr_outer **out);
int xr_outer_info(const xr_outer *outer, int64_t *n, int64_t *m, int64_t *nnz);
/* borrowed handle of the materialised product (owned by `outer`; do not destroy) */
int xr_outer_csr(xr_outer *outer, const xr_csr **out);
int xr_outer_destroy(xr_outer *outer);
/* NetworkGridder weights (xugrid/regrid/gridder.py:66-73 through UnstructuredGrid2d.intersection_length,
 * xugrid/regrid/unstructured.py:203-215): replaces numba_celltree's
 *     celltree.intersect_edges(edge_coords) -> (edge_index, face_index, intersections[n, 2, 2]),
 * the length = norm(diff(intersections)) that follows, the argsort by face and MatrixCSR.from_triplet.
 * edge_xy: (n_edge, 2, 2) float64 end-point coordinates (Ugrid1d.edge_node_coordinates).  Result: rows = faces of
 * `tree`, columns = edge ids (ascending within a row), data = length of the piece of the edge inside the face
 * (Cyrus-Beck clip against the convex, CCW-normalised face); only pieces of positive length are entries, so an
 * edge that merely touches a face corner or runs outside along its boundary adds nothing.  Applied with
 * xr_apply_csr like any other weights (source variables live on the EDGES).  `relative` lengths are not offered:
 * the reference never requests them (gridder.py:49) and its formula indexes the edge lengths by face id (:213-214). */
int xr_edge_length_csr(xr_mesh *tree, const double *edge_xy, int64_t n_edge, xr_csr **out);
/* ... with the end points already in HBM (edge_xy_dev: float64 [n_edge][2][2], a device pointer): no upload -- the device part
 * of the call above on its own (a third of which is the PCIe transfer of 32 bytes per edge). */
int xr_edge_length_csr_dev(xr_mesh *tree, const double *edge_xy_dev, int64_t n_edge, xr_csr **out);
/* The third return value of intersect_edges for the entries of such a matrix: intersections float64[nnz, 2, 2]
 * (begin and end point of every piece, along the direction of its edge), in the entry order of the CSR. */
int xr_edge_pieces(xr_mesh *tree, const xr_csr *csr, const double *edge_xy, int64_t n_edge,
                   double *intersections);
/* Optional locality hint for matrices that were uploaded (xr_csr_upload / xr_csr_from_triplet, i.e. the
 * from_weights path): one small integer per row such that rows with equal keys are spatial neighbours (e.g. the
 * Morton code of a coarse cell holding the target face's centroid).  With many source variables (K >= 8) the apply
 * regroups the STORED rows by key once -- results and xr_csr_download are unaffected.  xr_overlap attaches such
 * keys itself (runs of 16 consecutive target ids share a key, so that a variable's outputs leave in 128-byte pieces);
 * keys given for such a matrix are per CALLER row and replace that grouping -- worth it with xr_csr_output_stored_order,
 * where the output no longer cares about the caller's numbering and every row can go to its own tile. */
int xr_csr_set_row_keys(xr_csr *csr, const int64_t *keys, int64_t key_range);
/* The same for the COLUMNS (source cells): one small integer per column such that columns with equal keys are spatial
 * neighbours.  The columns are renumbered once by key (stable), so that the source values a block of target rows
 * gathers are neighbours in memory whatever the caller's cell numbering (a qhull-numbered mesh: half the fetched bytes
 * were unused).  Entry order inside the rows is untouched -- reducers add in the same order, results are bit-identical
 * -- and xr_csr_download returns the caller's column ids.  The source block handed to the apply entry points is either
 * in the caller's cell order (default: one gather pass per call puts it in the stored order) or, after
 * xr_csr_expect_permuted(csr, 1), already in the stored order: source_permuted[k][j] = source[k][order[j]] with
 * order = xr_csr_col_order (int64[m]) -- for data that stays on the device across many applies, or producers that can
 * write in that order. */
int xr_csr_set_col_keys(xr_csr *csr, const int64_t *keys, int64_t key_range);
int xr_csr_col_order(const xr_csr *csr, int64_t *order_out);
int xr_csr_expect_permuted(xr_csr *csr, int permuted);
/* The mirror image for the OUTPUT: after xr_csr_output_stored_order(csr, 1) the applies write stored row r to out[k, r]
 * (fully coalesced whatever the caller's numbering) instead of out[k, row_order[r]]; xr_csr_row_order returns the permutation
 * (stored row r = caller's row order[r]; K_hint is ignored and kept for binary compatibility).  The regrouping of the rows
 * into 2-D tiles -- otherwise deferred to the first apply with 8 or more variables -- is part of the stored order: both calls
 * settle it at once, and from xr_csr_output_stored_order(csr, 1) on the order is FROZEN: xr_csr_set_row_keys refuses
 * (XR_ERR_INVALID) until the stored-order output is switched off again, so no apply can ever write rows in a permutation
 * the caller has not read.  With columns AND rows in the engine's order a pipeline that keeps its blocks on the device pays
 * the caller's numbering once, not per apply. */
int xr_csr_output_stored_order(xr_csr *csr, int stored);
int xr_csr_row_order(const xr_csr *csr, int64_t K_hint, int64_t *order_out);
int xr_csr_destroy(xr_csr *csr);

/* ---- seam 2: apply ---------------------------------------------------------------------- */
/* make_regrid(func)._regrid(source, A, size), regridder.py:41-67.
 * source: (K, S) row-major, S = csr.m, dtype XR_F64 / XR_F32; out: float64 (K, T), T = csr.n,
 * NaN for empty rows.  `percentile` is used by XR_PERCENTILE only (0..100, reduce.py:241-243). */
int xr_apply_csr(const xr_csr *csr, int method, double percentile, const void *source,
                 int source_dtype, int64_t K, double *out);
int xr_apply_csr_dev(const xr_csr *csr, int method, double percentile, const void *source_dev,
                     int source_dtype, int64_t K, double *out_dev);
/* the same two calls on factored separable weights (xr_outer_create) */
int xr_apply_outer(xr_outer *outer, int method, double percentile, const void *source,
                   int source_dtype, int64_t K, double *out);
int xr_apply_outer_dev(xr_outer *outer, int method, double percentile, const void *source_dev,
                       int source_dtype, int64_t K, double *out_dev);
/* CentroidLocatorRegridder._regrid, regridder.py:400-409: out[k,row[i]] = source[k,col[i]],
 * NaN elsewhere.  rows must be unique (they are target indices of located centroids). */
int xr_apply_coo(const int64_t *row, const int64_t *col, int64_t nnz, int64_t T,
                 const void *source, int source_dtype, int64_t K, int64_t S, double *out);

/* ---- multi-GPU set-up: which source faces are mine, which targets can they reach ------------------
 * The reference has no multi-device path (its only parallel loop is the dask map of regridder.py:167-185); SURVEY 8(e) shards
 * the SOURCE faces over the ranks and replicates the target.  xr_shard_plan_dev evaluates the partition rule on the rank's
 * device from the replicated raw meshes (float64 [N, 2] coordinates, int64 [F, m] connectivity with -1 fill, device pointers):
 *   mode 0 "hash"      owner = face id mod world (the north star's wording)
 *   mode 1 "morton"    cells of a 1024 x 1024 raster over the source centroids, ordered along the Morton curve, cut into
 *                      `world` stretches of equal face COUNT (owner per cell: floor(faces in front of the cell * world / S))
 *   mode 2 "balanced"  the same cut by estimated WORK, round(4096 (1 + 4 n_tgt / n_src)) per source face from a coarse raster
 *                      both meshes are counted into
 * and keeps the target faces whose box touches the 128 x 128 occupancy raster of the rank's own source faces.  Integer
 * arithmetic decides every owner, so all ranks agree without communication.  Outputs: ascending global ids (device arrays of
 * capacity n_src_face / n_tgt_face), their counts, optionally the owner of every source face (int32 [n_src_face]). */
int xr_shard_plan_dev(const double *src_xy_dev, const int64_t *src_faces_dev, int64_t n_src_face, int src_m,
                      const double *tgt_xy_dev, const int64_t *tgt_faces_dev, int64_t n_tgt_face, int tgt_m, int world, int rank,
                      int mode, int64_t *local_faces_dev, int64_t *n_local_faces, int64_t *local_targets_dev,
                      int64_t *n_local_targets, int32_t *owner_dev);

/* Multi-GPU (source faces sharded over ranks).  For EVERY reducer that decomposes over source shards
 * (xugrid/regrid/reduce.py:16-123, 206-222) each rank reduces the entries of its own columns to a few partial STATE components per (target, variable),
 * the ranks combine the components element-wise with ONE collective (sum, or max for minimum / maximum), the owner of a
 * target finalises.  Components (v = value, w = weight; "valid" = v not NaN):
 *   XR_MEAN, XR_FIRST_ORDER_CONSERVATIVE   [sum w v, sum w]                     over valid          -> c0 / c1 resp. c0
 *   XR_SUM                                 [sum v,   sum w]                     over valid          -> c0
 *   XR_HARMONIC_MEAN                       [sum w,   sum w / v]                 valid, v != 0, w > 0 -> c0 / c1
 *   XR_GEOMETRIC_MEAN                      [sum w (ALL entries), sum w ln v, sum w (v > 0, w > 0), #(v < 0)]
 *                                                                                                     -> exp(c1 / c2)
 *   XR_MINIMUM / XR_MAXIMUM                [max(-v) resp. max v, max w]         over valid (combine with MAX)
 * with NaN where the reference returns NaN (zero weight sum, a negative value under the geometric mean ...).
 * xr_partial_components(method) -> number of components (0: the reducer needs whole rows: mode, percentiles,
 * max_overlap), xr_partial_combine_is_max(method) -> 1 if the collective is MAX.
 * Layouts: planes float64 [C, K, T] (dense exchange) or rows float64 [T, C * K] (sparse exchange, component-major
 * inside a row); t runs over the CALLER's row order of the matrix. */
int xr_partial_components(int method);
int xr_partial_combine_is_max(int method);
int xr_apply_partial_dev(const xr_csr *csr, int method, const void *source_dev, int source_dtype, int64_t K,
                         double *out_dev, int rows_layout);
/* Weight build + partial state of ONE call: what a rank of the sharded regridder does per rebuild (xugrid/regrid/regridder.py:386-398
 * + the per-target reduction of :41-67 restricted to the rank's columns).  As xr_overlap_apply_dev: for K = 1 the partial-state
 * kernel is enqueued before the host has read the matrix' sizes back, so the one host round trip of the build hides behind it. */
int xr_overlap_partial_dev(xr_mesh *tree, xr_mesh *query, int relative, int method, const void *source_dev, int source_dtype,
                           int64_t K, double *out_dev, int rows_layout, xr_csr **out);
/* identity element of the combine step in the same layouts (buffers of targets a rank has no weight for) */
int xr_partial_fill_identity_dev(int method, double *planes_dev, int64_t K, int64_t T);
/* out_dev float64 [K, T] from combined planes [C, K, T] */
int xr_finalize_partial_dev(int method, const double *planes_dev, int64_t K, int64_t T, double *out_dev);
/* owner side of the sparse exchange in one launch: rows_dev float64 [R, C * K] received partial rows; target t of this
 * rank's slice combines rows order_dev[indptr_dev[t] .. indptr_dev[t+1]) in that (sender) order and is finalised:
 * out_dev float64 [K, n_targets]. */
int xr_reduce_partial_rows_dev(int method, const double *rows_dev, const int64_t *indptr_dev, const int64_t *order_dev,
                               int64_t n_targets, int64_t K, double *out_dev);

/* ---- raw HBM helpers for hosts that do not bring their own allocator -------------------- */
int xr_dev_alloc(int64_t bytes, void **ptr_out);
int xr_dev_free(void *ptr);
int xr_dev_upload(void *dst_dev, const void *src_host, int64_t bytes);
int xr_dev_download(void *dst_host, const void *src_dev, int64_t bytes);
int xr_dev_sync(void);

/* ---- in-library kernel timing (HIP events on the engine stream) -------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on the engine stream; durations
 * are accumulated per kernel name. */
int xr_prof_enable(int on);
int xr_prof_reset(void);
/* Number of distinct kernels recorded since the last reset. */
int xr_prof_count(int *count);
/* i-th record: name (copied, NUL-terminated, at most name_cap bytes), launches, total ms. */
int xr_prof_get(int i, char *name, int name_cap, int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* XUGRID_AMD_H */
