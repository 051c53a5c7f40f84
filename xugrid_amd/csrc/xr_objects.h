// xr_objects.h -- the two device-resident handle types behind the C ABI.
#pragma once

#include "xr_geom.h"
#include "xr_internal.h"

// Device-resident face topology + derived per-face data + spatial index.  Layout: xr_geom.h.
struct xr_mesh {
    int64_t n_node = 0, n_face = 0;
    int m = 0; // n_max_node_per_face

    xr::DevBuf<double> node_xy;    // [n_node*2]
    xr::DevBuf<int32_t> faces_raw; // [n_face*m] caller's vertex order, fill -> -1

    // derived by xr_mesh_prepare
    bool prepared = false;
    xr::DevBuf<int32_t> faces; // [n_face*m] CCW
    xr::DevBuf<uint8_t> len;   // [n_face]
    xr::DevBuf<double> bbox;   // [n_face*4] xmin,xmax,ymin,ymax
    xr::DevBuf<double> area;   // [n_face]
    xr::DevBuf<double> stats;  // [7] xmin,xmax,ymin,ymax,sum_extent,max_extent,max_diagonal (device)
    bool stats_valid = false;
    double h_stats[7] = {0, 0, 0, 0, 0, 0, 0};

    // derived by xr_mesh_build_index
    bool indexed = false;
    xr::GridParams grid{};
    xr::DevBuf<int32_t> cell_start; // [n_cells+1]
    xr::DevBuf<float> rec_bb;       // [n_face*4]
    xr::DevBuf<int32_t> rec_face;   // [n_face]

    int64_t last_candidates = 0;
};

// Device-resident MatrixCSR (xugrid/core/sparse.py:81-137), int32 structure + float64 data.
struct xr_csr {
    int64_t n = 0, m = 0, nnz = 0;
    xr::DevBuf<int32_t> indptr;  // [n+1]
    xr::DevBuf<int32_t> indices; // [nnz]
    xr::DevBuf<double> data;     // [nnz]
    int32_t max_row = -1;        // longest row (np.diff(indptr).max(), regridder.py:48); -1 = unknown
};

namespace xr {
void mesh_prepare(xr_mesh *mesh);
void mesh_build_index(xr_mesh *mesh);
void mesh_read_stats(xr_mesh *mesh);
} // namespace xr
