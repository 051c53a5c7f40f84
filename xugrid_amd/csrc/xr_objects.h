// xr_objects.h -- the two device-resident handle types behind the C ABI.
#pragma once
#include <memory>
#include <vector>

#include "xr_geom.h"
#include "xr_internal.h"

// Device-resident face topology + derived per-face data, in three stages (layout: xr_geom.h):
//   raw      what the caller uploaded (node_xy, faces_raw)
//   prepared per-face arrays in the CALLER's face order (fxy, len, bbox, area, stats)
//   query    the same arrays permuted into a spatially coherent order (Morton order of coarse
//            cells) -- used when the mesh is the QUERY (regridding target) side
//   index    the same arrays permuted into grid-cell order + cell_start -- used when the mesh is
//            the TREE (regridding source) side
struct xr_mesh {
    int64_t n_node = 0, n_face = 0;
    int m = 0; // n_max_node_per_face

    xr::DevBuf<double> node_xy;    // [n_node*2]
    xr::DevBuf<int32_t> faces_raw; // [n_face*m] caller's vertex order, fill -> -1

    // ---- prepared (caller's face order)
    bool prepared = false;  // statistics known
    bool has_attrs = false; // ... and len / bbox / fxy filled (prepared as a query)
    bool area_valid = false;
    xr::DevBuf<double> fxy;   // [n_face*m*2] CCW-normalised vertex coordinates per face (only if fxy_valid:
                              // needed when the mesh is a query kept in the caller's numbering)
    bool fxy_valid = false;
    xr::DevBuf<int32_t> fxy_off; // [n_face+1] only for ragged() meshes: fxy is flat, face f starts at vertex fxy_off[f]
    xr::DevBuf<uint8_t> len;  // [n_face]
    xr::DevBuf<double> bbox;  // [n_face*4] xmin,xmax,ymin,ymax
    xr::DevBuf<double> area;  // [n_face]   connectivity.area on the caller's vertex order (mesh_area, on demand)
    hipEvent_t centroids_event = nullptr;   // recorded behind the kernel that filled centroids_dev, on centroids_stream: a user on
    hipStream_t centroids_stream = nullptr; // another stream (the handles' kernels run on the side stream) waits for it
    std::shared_ptr<xr::DevBuf<double>> centroids_dev; // [n_face*2] connectivity.centroids, on demand (mesh_centroids_shared); kept like
                                                       // the reference's cached Ugrid2d.centroids until the mesh is invalidated
    xr::DevBuf<double> stats; // [8] xmin,xmax,ymin,ymax,sum_extent,max_extent,max_diagonal,sum_jump (device)
    bool stats_valid = false;
    bool stats_sampled = false; // h_stats[4..6] come from a sample of the faces ([7] = faces sampled): enough to size a grid, NOT for the default tolerance
    double h_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the reduction kernel also deposits the statistics in pinned host memory; the host waits on an event
    // recorded right behind it, so kernels queued later (the other mesh's prepare) keep the GPU busy meanwhile
    double *stats_host = nullptr; // [9] pinned: the statistics + the sequence word the publishing block stores last
    hipEvent_t stats_event = nullptr;
    xr::DevBuf<unsigned> stats_done; // [1] blocks that have delivered their partial (zero at rest; stats_tail, xr_mesh.hip)
    double stats_seq = 0.0;          // sequence number of the last statistics launch (what stats_host[8] must show)
    bool stats_event_pending = false; // a reduction kernel + event were enqueued for the last launch (fallback of the poll)
    bool stats_polled = false;       // the statistics of the last launch arrive through the sequence word (no event recorded)
    xr_mesh() = default;
    xr_mesh(const xr_mesh &) = delete;
    xr_mesh &operator=(const xr_mesh &) = delete;
    ~xr_mesh() {
        if (stats_host) (void)hipHostFree(stats_host);
        if (centroids_event) (void)hipEventDestroy(centroids_event);
        if (stats_event) (void)hipEventDestroy(stats_event);
    }

    // ---- query order.  If the caller's face numbering is already spatially coherent (mean distance
    // between consecutive faces <= a few face extents) the caller's order IS the query order:
    // no permutation, no copies (query_identity); the accessors below hide the difference.
    bool query_ready = false;
    bool query_identity = false;
    xr::DevBuf<int32_t> q_perm; // [n_face] position -> caller's face id
    xr::DevBuf<double> q_fxy;   // [n_face*m*2], ragged(): [sum len * 2]
    xr::DevBuf<int32_t> q_off;  // [n_face+1] ragged() only
    xr::DevBuf<uint8_t> q_len;  // [n_face]
    xr::DevBuf<double> q_bbox;  // [n_face*4]

    // ---- tree index (records in grid-cell order)
    bool indexed = false;
    xr::GridParams grid{};
    xr::DevBuf<int32_t> cell_start; // [n_cells+1]
    xr::DevBuf<float> rec_bb;       // [n_face*4] conservative f32 bbox relative to the grid origin
    xr::DevBuf<int32_t> rec_face;   // [n_face]   record -> caller's face id
    xr::DevBuf<double> rec_fxy;     // [n_face*m*2], ragged(): [sum len * 2]
    xr::DevBuf<int32_t> rec_off;    // [n_face+1] ragged() only
    xr::DevBuf<uint8_t> rec_len;    // [n_face]

    int64_t last_candidates = 0;

    // ---- kept by the barycentric construction when this mesh is a Voronoi tessellation (xr_locate.hip:barycentric_csr): the
    // vertex -> face table with the interpolation map behind it, and the flags of the cells that hold a substitute vertex --
    // a second interpolator on a cached tessellation uploads and recomputes nothing (its first kernel then runs BESIDE the
    // source-side locate pass instead of behind a copy)
    std::vector<int64_t> bary_ids_host;  // what was uploaded (tail of the vertex table, then the map)
    xr::DevBuf<int64_t> bary_ids;        // [n_node + 2 n_extra + 1]
    int64_t bary_n_identity = -1, bary_n_extra = -1;
    xr::DevBuf<uint8_t> bary_cell_flag;  // [n_face], valid for bary_n_extra
    bool bary_flag_valid = false;

    // vertex blocks flat + offsets instead of dense [n_face][m] (xr_geom.h)
    bool ragged() const { return m > xr::DENSE_MAX_NODES; }
    const int32_t *caller_off() const { return ragged() ? fxy_off.get() : nullptr; }
    const int32_t *record_off() const { return ragged() ? rec_off.get() : nullptr; }
    const int32_t *qo_off() const { return !ragged() ? nullptr : query_identity ? fxy_off.get() : q_off.get(); }
    const double *qo_fxy() const { return query_identity ? fxy.get() : q_fxy.get(); }
    const uint8_t *qo_len() const { return query_identity ? len.get() : q_len.get(); }
    const double *qo_bbox() const { return query_identity ? bbox.get() : q_bbox.get(); }
    const int32_t *qo_perm() const { return query_identity ? nullptr : q_perm.get(); } // nullptr = identity
};

// Rows with more entries than this are not walked by one thread but reduced cooperatively in the apply kernels:
// by one wave each up to XR_APPLY_WAVE_ROW entries, by a whole block beyond.  (Cooperative = fixed butterfly
// order: deterministic, ~1e-15 from the sequential order of the reference loop.)
static constexpr int XR_APPLY_LONG_ROW = 32;
static constexpr int XR_APPLY_WAVE_ROW = 2048;

// Device-resident MatrixCSR (xugrid/core/sparse.py:81-137), int32 structure + float64 data.
struct xr_csr {
    int64_t n = 0, m = 0, nnz = 0;
    xr::DevBuf<int32_t> indptr;    // [n+1]
    xr::DevBuf<int32_t> indices;   // [nnz]
    xr::DevBuf<double> data;       // [nnz]
    // Rows may be STORED in a spatially coherent order instead of the caller's: stored row r is the
    // caller's row row_order[r] (weights built by xr_overlap: the query mesh's query order).  The
    // apply kernels scatter their outputs through it; xr_csr_download un-permutes.
    xr::DevBuf<int32_t> row_order; // [n]
    bool has_row_order = false;
    // stored rows with more than XR_APPLY_LONG_ROW entries (reduced by one wave or block each in the apply)
    xr::DevBuf<int32_t> long_rows; // [<= n]
    xr::DevBuf<int32_t> n_long;    // [1] device-side count ([2] for matrices of the triangle pipeline: [1] = apply gate)
    bool apply_gated = false;      // the K = 1 apply being enqueued must check n_long[1] (matrix not yet confirmed by the host)
    int64_t max_row_len = -1;      // entries of the longest row if the builder reported it (-1: unknown)
    bool has_long = false;
    // Coarse Morton key per STORED row (tile of ~12 target extents).  Set by xr_overlap when the rows are kept
    // in the caller's order, or by xr_csr_set_row_keys; consumed once by the many-variable apply, which regroups
    // the stored rows into compact 2-D tiles (xr_apply.hip: ensure_tiled) and then drops the keys.
    xr::DevBuf<int32_t> tile_key; // [n]
    bool has_tile_key = false;
    int64_t tile_key_range = 0;   // keys are in [0, tile_key_range)
    // Columns renumbered by a spatial key (xr_csr_set_col_keys): stored column j is the caller's column col_of[j].
    // The apply gathers the caller's source block into the stored order first unless the caller says it already is.
    xr::DevBuf<int32_t> col_of; // [m]
    bool has_col_perm = false;
    bool source_permuted = false;
    bool output_stored = false; // applies write row r of the STORED order to out[k, r] (xr_csr_output_stored_order)
    // "apply plan" for many source variables (built lazily, xr_apply.hip): per block of 256 stored
    // rows the sorted list of DISTINCT column ids and, per entry, its 16-bit position in that list
    bool plan_ready = false;
    bool plan_merged = false;           // the lists are per GROUP of row blocks (k_plan_build_group), not per block
    xr::DevBuf<int32_t> plan_unplanned; // blocks the plan could not take (handled by the direct kernel)
    int plan_n_unplanned = 0;
    int plan_lmax = 0;               // entries of the largest planned block, rounded up to 256 (LDS stage of the apply)
    xr::DevBuf<int32_t> plan_ucol;   // [n_blocks * PLAN_UMAX]
    xr::DevBuf<int32_t> plan_nuniq;  // [n_blocks], -1 = block not planned (falls back to direct gathers)
    xr::DevBuf<uint16_t> plan_loc;   // [nnz]
};

// Separable (rectilinear) weights kept as the two per-axis sparse matrices (xugrid/regrid/structured.py:503-601): the
// P = P_y * P_x entries of their outer product are never stored unless somebody asks for the CSR.
struct xr_outer {
    int64_t nty = 0, nsy = 0, ntx = 0, nsx = 0, Py = 0, Px = 0;
    xr::DevBuf<int32_t> ipy, ipx; // [nt + 1]
    xr::DevBuf<int32_t> sy, sx;   // [P] source index per entry (ascending within a row)
    xr::DevBuf<double> wy, wx;    // [P]
    xr::DevBuf<int32_t> tx;       // [Px] x-target owning each x-entry (CSR assembly)
    int64_t max_cy = 0, max_cx = 0;
    xr_csr *csr = nullptr;        // materialised on demand (mode / percentiles, downloads)
    ~xr_outer() { delete csr; }
};

namespace xr {
void mesh_prepare(xr_mesh *mesh, bool want_fxy = true, bool stats_on_side = false, bool allow_sampled = false); // stats_on_side: the reduction of the
    // statistics (only the HOST reads them) leaves the main stream: kernels queued behind the prepare pass do not wait for it
const double *mesh_area(xr_mesh *mesh); // connectivity.area in the caller's face order (computed on first use)
void mesh_face_coords(xr_mesh *mesh); // make sure the caller-order vertex blocks exist
void mesh_query_order(xr_mesh *mesh);
void mesh_build_index(xr_mesh *mesh);
void flush_pending_points(); // launch the deferred source-side kernels of pending xr_points handles (xr_locate.hip)
void flush_pending_points_of(const xr_mesh *mesh); // ... if one of them reads `mesh`, before its arrays are released
void mesh_read_stats(xr_mesh *mesh, bool need_exact = false); // need_exact: statistics over ALL faces (sampled ones are redone)
// the apply of a finished (or, K = 1, of a just-enqueued) matrix on the calling thread's launch stream (xr_apply.hip)
void csr_apply_dev(const xr_csr *csr, int method, double percentile, const void *src_dev, int dtype, int64_t K, double *out_dev);
void csr_partial_dev(const xr_csr *csr, int method, const void *src_dev, int dtype, int64_t K, double *out_dev, int rows_layout);
void mesh_centroids_dev(xr_mesh *mesh, double *cxy_dev);    // connectivity.centroids into device memory [n_face*2]
std::shared_ptr<xr::DevBuf<double>> mesh_centroids_shared(xr_mesh *mesh); // ... computed once per mesh, shared with the callers
void mesh_faces_ccw_dev(xr_mesh *mesh, int64_t *faces_dev, bool caller_order = false); // CCW-normalised (or the caller's) connectivity [n_face*m]
} // namespace xr
