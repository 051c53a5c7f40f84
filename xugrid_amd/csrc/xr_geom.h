// xr_geom.h -- device-side geometry types shared by the mesh / overlap / locate kernels.
//
// HBM layout of a mesh (struct xr_mesh), all face-major so that no hot kernel chases
// connectivity -> node indirections:
//   fxy   f64 [n_face][M][2]  CCW-normalised vertex coordinates of every face (M = n_max_node;
//                             48 B per triangle); slots >= len are never read.  Meshes with M > DENSE_MAX_NODES
//                             (polygon meshes, e.g. the M = 22 Voronoi mesh of the barycentric path with ~6 real
//                             vertices per cell) keep the blocks FLAT with offsets instead: f64 [sum len][2] +
//                             off i32 [n_face + 1] (fxy_off / q_off / rec_off); face_vertex_base() hides the difference
//   len   u8  [n_face]        number of valid vertices (polygon_length)
//   bbox  f64 [n_face][4]     xmin, xmax, ymin, ymax
//   area  f64 [n_face]        connectivity.area on the caller's vertex order
// The raw upload (node_xy f64[N][2], faces_raw i32[F][M], the CSR-style flat connectivity with the
// implicit offset f*M) is only read once, by the prepare kernel.
// Two spatially sorted copies exist (counting sort, xr_mesh.hip):
//   query order  q_perm/q_fxy/q_len/q_bbox: faces grouped by the Morton code of a coarse cell, so a
//                256-face block is a compact 2-D patch (search/clip/row kernels walk this order)
//   tree index   hierarchical uniform grid, one insertion per face: level l has square cells of
//                size h0 * 2^l; a face lives on the lowest level whose cell size exceeds its bbox
//                extent, in the cell holding its bbox lower-left corner
//     cell_start i32 [n_cell+1]  CSR offsets cell -> record run, all levels concatenated
//     rec_bb     f32 [n_face][4] conservative (outward-rounded) bbox relative to the grid origin
//     rec_face   i32 [n_face]    caller's face id of each record
//     rec_fxy / rec_len          the face arrays in record order (clip reads them contiguously)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace xr {

struct P2 {
    double x, y;
};

// vertex blocks of meshes with at most this many nodes per face are dense [n_face][M]; wider meshes: flat + offsets
static constexpr int DENSE_MAX_NODES = 4;

static constexpr int MAX_LEVELS = 24;
// cell size ratio between consecutive grid levels = 2^LEVEL_SHIFT
static constexpr int LEVEL_SHIFT = 1; // (4x steps halve the level count but were 3 % slower in the search)

struct GridParams {
    double x0, y0;    // grid origin (domain lower-left)
    double h0;        // level-0 cell size
    double inv_h0;    // 1 / h0
    int n_levels;
    int n_cells;      // total over all levels
    int base[MAX_LEVELS];
    int nx[MAX_LEVELS];
    int ny[MAX_LEVELS];
};

// level-l cell size and its inverse (exact power-of-two scalings of h0 / inv_h0)
__host__ __device__ inline double level_h(const GridParams &g, int l) { return ldexp(g.h0, l * LEVEL_SHIFT); }
__host__ __device__ inline double level_inv_h(const GridParams &g, int l) { return ldexp(g.inv_h0, -l * LEVEL_SHIFT); }

// monotone non-decreasing in x: (x - x0) * inv_h, floor, clamp
__host__ __device__ inline int cell_coord(double x, double origin, double inv_h, int n) {
    double c = floor((x - origin) * inv_h);
    if (!(c > 0.0)) return 0; // also catches NaN
    if (c >= (double)(n - 1)) return n - 1;
    return (int)c;
}

// lowest level whose cell size strictly dominates the extent (0.1 % slack covers every rounding)
__host__ __device__ inline int level_of_extent(const GridParams &g, double e) {
    int l = 0;
    while (l < g.n_levels - 1 && !(e <= 0.999 * level_h(g, l))) l++;
    return l;
}

#ifdef __HIPCC__
// index (in vertices) of the first vertex of face / record r: flat layout with offsets, or dense [.][m]
__device__ __forceinline__ int64_t face_vertex_base(const int32_t *__restrict__ off, int64_t r, int m) {
    return off ? (int64_t)off[r] : r * m;
}

__device__ __forceinline__ P2 load_p2(const double *__restrict__ xy, int i) {
    const double2 v = reinterpret_cast<const double2 *>(xy)[i];
    return P2{v.x, v.y};
}

// f32 box test as ONE float.  b = (xmin, xmax, ymin, ymax) of a record, q = the query box: the value is negative iff all
// four STRICT inequalities qx0 < b.xmax, b.xmin < qx1, qy0 < b.ymax, b.ymin < qy1 hold, and <= 0 iff the non-strict ones do
// (a < b <=> a - b < 0 exactly in IEEE arithmetic; NaN fails both).  Seven VALU instructions and no branch: the obvious
// `a && b && c && d` compiles to a compare / select / shift / bit-op chain of twice that, and `in_range && hit` to a branch
// per record with the record's load and its wait INSIDE the branch (the loads are then no longer in flight together).
__device__ __forceinline__ float box_gap(float4 b, float qx0, float qx1, float qy0, float qy1) {
    return fmaxf(fmaxf(qx0 - b.y, b.x - qx1), fmaxf(qy0 - b.w, b.z - qy1));
}

// ---------------------------------------------------------------------------------------------
// polygon_length (a minimal polygon is a triangle; stop at the first fill value) and
// counter_clockwise (the first non-collinear vertex triple decides) of one face
template <int MA>
__device__ __forceinline__ void face_shape(const double *__restrict__ node_xy, const int (&face)[MA], int m, int &n,
                                           bool &flip) {
    n = m;
#pragma unroll
    for (int i = MA - 1; i >= 3; i--)
        if (i < m && face[i] < 0) n = i;
    flip = false;
    for (int i = 0; i < n; i++) {
        const int ia = face[(i + n - 2) % n], ib = face[(i + n - 1) % n], ic = face[i];
        const P2 a = load_p2(node_xy, ia), b = load_p2(node_xy, ib), c = load_p2(node_xy, ic);
        const double ux = b.x - a.x, uy = b.y - a.y, vx = c.x - a.x, vy = c.y - a.y;
        const double prod = ux * vy - uy * vx;
        if (prod == 0) continue;
        flip = prod < 0;
        break;
    }
}

// conservative float bounds: strictly below / above the double value
__device__ __forceinline__ float f32_below(double v) {
    return nextafterf(__double2float_rd(v), -INFINITY);
}
__device__ __forceinline__ float f32_above(double v) {
    return nextafterf(__double2float_ru(v), INFINITY);
}
#endif

// Morton (Z-order) key of the coarse cell that holds a bbox centre: spatial orderings of faces / matrix rows
__device__ __forceinline__ uint32_t spread_bits16(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

struct MortonParams {
    double x0, y0, inv_h;
    int n_side;
    int n_run = 1;
};

__device__ __forceinline__ int morton_key(const MortonParams &mp, double4 bb) {
    const int cx = cell_coord(0.5 * (bb.x + bb.y), mp.x0, mp.inv_h, mp.n_side);
    const int cy = cell_coord(0.5 * (bb.z + bb.w), mp.y0, mp.inv_h, mp.n_side);
    return (int)(spread_bits16((uint32_t)cx) | (spread_bits16((uint32_t)cy) << 1));
}


} // namespace xr
