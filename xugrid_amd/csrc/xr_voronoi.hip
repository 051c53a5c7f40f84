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
#include <chrono>
#include <cmath>
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
    std::vector<double> b_node_xy;                 // coordinates of the boundary nodes
    // the cells of the boundary nodes (voronoi_boundary_cells: native O(boundary) host part)
    bool cells_ready = false;
    std::vector<double> c_extra_xy;   // vertices added behind the n_face centroids: kept projections, then one per boundary node
    std::vector<int64_t> c_cells;     // [c_n_cell][c_m] global vertex ids, -1 padded, counter-clockwise
    int64_t c_n_cell = 0, c_m = 0;
    std::vector<int64_t> c_tail;      // source face of every added vertex (-1: the per-node extras)
    std::vector<int64_t> c_interp;    // [n_extra][2] the two projection vertex ids an extra corner sits between
};

namespace xr {

// Node -> face inversion = a counting sort of the (node, face) slots by node.  Device-scope atomics are executed at the memory
// side of the fabric on this part (every one of them leaves the XCD's L2: PMC TCC_EA0_ATOMIC = TCC_ATOMIC), ~40 G/s in total:
// one atomic per SLOT made both passes atomic-bound (3M slots of a 1M-triangle mesh: 75 + 82 us).  A block therefore counts
// its 1024 slots per DISTINCT node in an LDS table first (faces arrive spatially coherent: a node's ~6 faces mostly sit in
// the same block) and issues one global atomic per distinct node -- the count pass a plain add, the scatter pass one
// returning add that reserves the block's stretch of the node's row; a slot's place in it is its rank in the table.
static constexpr int VOR_SLOTS = 1024, VOR_TABLE = 2048; // slots per block (4 per thread); open-addressing table, load <= 1/2
struct VorTable {
    int32_t key[VOR_TABLE];
    int32_t cnt[VOR_TABLE];
    int32_t base[VOR_TABLE];
};
__device__ __forceinline__ void vor_table_clear(VorTable &t) {
    for (int s = threadIdx.x; s < VOR_TABLE; s += 256) {
        t.key[s] = -1;
        t.cnt[s] = 0;
    }
}
// -> slot of node v in the table; rank = position of this (node, face) slot among the block's slots of the same node
__device__ __forceinline__ int vor_table_insert(VorTable &t, int v, int &rank) {
    int s = (int)(((unsigned)v * 2654435761u) >> 21) & (VOR_TABLE - 1);
    while (true) {
        const int prev = atomicCAS(&t.key[s], -1, v);
        if (prev == -1 || prev == v) break;
        s = (s + 1) & (VOR_TABLE - 1);
    }
    rank = atomicAdd(&t.cnt[s], 1);
    return s;
}

__global__ void __launch_bounds__(256)
k_vor_count(const int32_t *__restrict__ faces, int64_t total, int32_t *__restrict__ count) {
    __shared__ VorTable sh;
    vor_table_clear(sh);
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * VOR_SLOTS + threadIdx.x;
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = i0 + u * 256 < total ? faces[i0 + u * 256] : -1;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int rank;
        if (v[u] >= 0) vor_table_insert(sh, v[u], rank);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < VOR_TABLE; s += 256)
        if (sh.key[s] >= 0) atomicAdd(&count[sh.key[s]], sh.cnt[s]);
}

__global__ void __launch_bounds__(256)
k_vor_scatter(const int32_t *__restrict__ faces, int64_t total, int m, const int32_t *__restrict__ indptr,
              int32_t *__restrict__ cursor, int32_t *__restrict__ out) {
    __shared__ VorTable sh;
    vor_table_clear(sh);
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * VOR_SLOTS + threadIdx.x;
    int v[4], slot[4], rank[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = i0 + u * 256 < total ? faces[i0 + u * 256] : -1;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        slot[u] = 0, rank[u] = 0;
        if (v[u] >= 0) slot[u] = vor_table_insert(sh, v[u], rank[u]);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < VOR_TABLE; s += 256) {
        const int key = sh.key[s];
        if (key >= 0) sh.base[s] = indptr[key] + atomicAdd(&cursor[key], sh.cnt[s]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (v[u] >= 0) out[sh.base[slot[u]] + rank[u]] = (int32_t)((i0 + u * 256) / m);
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
               int32_t *__restrict__ e_all /* (lo, hi, face) per exterior edge: one download */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_face * m) return;
    const int64_t f = i / m;
    const int j = (int)(i - f * m);
    const int32_t *face = faces + f * m;
    const int n = face_len(face, m);
    if (j >= n) return;
    const int a = face[j], b = face[j + 1 == n ? 0 : j + 1]; // (wrap by compare: `% n` with a run-time n is an integer division)
    if (a == b) return;
    bool shared = false;
    if (m == 3) {
        // triangles: any two nodes of a face are ring neighbours, so the edge is shared iff a face other than f is in the rows
        // of BOTH nodes -- a merge of two short ascending rows (k_vor_sort_rows ran before), contiguous reads, instead of a
        // gather of every neighbouring face's node list
        int ia = indptr[a], ib = indptr[b];
        const int ea = indptr[a + 1], eb = indptr[b + 1];
        constexpr int ROW_REG = 8; // (a Delaunay node has ~6 faces)
        if (ea - ia <= ROW_REG && eb - ib <= ROW_REG) {
            // both rows in registers with ONE round of independent loads, then the intersection by all pairs: the merge below is a
            // chain of ~10 dependent pairs of loads per half-edge (the kernel waited on memory for 57 % of its wave cycles)
            int ra[ROW_REG], rb[ROW_REG];
#pragma unroll
            for (int u = 0; u < ROW_REG; u++) ra[u] = ia + u < ea ? rows[ia + u] : -1;
#pragma unroll
            for (int u = 0; u < ROW_REG; u++) rb[u] = ib + u < eb ? rows[ib + u] : -2;
#pragma unroll
            for (int u = 0; u < ROW_REG; u++) {
                bool in_b = false;
#pragma unroll
                for (int w = 0; w < ROW_REG; w++) in_b = in_b || ra[u] == rb[w];
                shared = shared || (in_b && ra[u] != (int)f);
            }
        } else
        while (ia < ea && ib < eb) {
            const int ga = rows[ia], gb = rows[ib];
            if (ga == gb) {
                if (ga != f) {
                    shared = true;
                    break;
                }
                ia++;
                ib++;
            } else if (ga < gb) {
                ia++;
            } else {
                ib++;
            }
        }
    } else
    for (int r = indptr[a]; r < indptr[a + 1] && !shared; r++) {
        const int g = rows[r];
        if (g == f) continue;
        const int32_t *other = faces + (int64_t)g * m;
        const int k = face_len(other, m);
        for (int t = 0; t < k; t++) {
            if (other[t] != a) continue;
            if (other[t + 1 == k ? 0 : t + 1] == b || other[t == 0 ? k - 1 : t - 1] == b) shared = true;
        }
    }
    if (shared) return;
    on_boundary[a] = 1;
    on_boundary[b] = 1;
    const int at = atomicAdd(n_edges, 1);
    e_all[3 * (int64_t)at] = a < b ? a : b;
    e_all[3 * (int64_t)at + 1] = a < b ? b : a;
    e_all[3 * (int64_t)at + 2] = (int32_t)f;
}

__global__ void k_vor_init_counters(int32_t *__restrict__ c) {
    if (threadIdx.x < 8) c[threadIdx.x] = threadIdx.x == 1 ? INT32_MAX : 0;
}
__global__ void k_vor_totals(const int32_t *__restrict__ n_interior, const int32_t *__restrict__ nnz, int32_t *__restrict__ c) {
    if (threadIdx.x == 0) {
        c[3] = *n_interior;
        c[4] = *nnz;
    }
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
// The row -- face ids and centroid offsets -- is gathered ONCE into the thread's LDS column (independent loads, four in
// flight) and insertion-sorted there: the same insertions and comparisons, in the same order, as the sort in global memory
// this replaces, whose every step was a load that depended on the store before it and re-gathered a centroid through L2
// (0.35 ms for 0.5M nodes; PMC: 164 MB written for 12 MB of rows).  Rows longer than VOR_CAP keep that path.
static constexpr int VOR_BLOCK = 128, VOR_CAP = 16;

__global__ void __launch_bounds__(VOR_BLOCK)
k_vor_interior(const double *__restrict__ node_xy, const double *__restrict__ cxy,
               const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows_asc,
               const uint8_t *__restrict__ on_boundary, int64_t n_node, int32_t *__restrict__ rows_ccw,
               uint8_t *__restrict__ interior, int32_t *__restrict__ flag32, int32_t *__restrict__ deg_minmax) {
    __shared__ double2 sh_d[VOR_CAP][VOR_BLOCK];
    __shared__ int32_t sh_k[VOR_CAP][VOR_BLOCK];
    const int t = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * VOR_BLOCK + t;
    const bool in_range = v < n_node;
    const int s = in_range ? indptr[v] : 0, e = in_range ? indptr[v + 1] : 0;
    const bool ok = in_range && e > s && !on_boundary[v];
    // degree range: reduced over the wave, and an atomic only when it improves what the word already holds -- one
    // atomicMin + one atomicMax per NODE on the same two words cost 0.13 ms of the kernel's 0.23 (A/B without them: 0.097)
    {
        int dmin = ok ? e - s : INT32_MAX, dmax = ok ? e - s : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            dmin = min(dmin, __shfl_xor(dmin, d, 64));
            dmax = max(dmax, __shfl_xor(dmax, d, 64));
        }
        if ((t & 63) == 0) {
            if (dmin < __hip_atomic_load(&deg_minmax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&deg_minmax[0], dmin);
            if (dmax > __hip_atomic_load(&deg_minmax[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&deg_minmax[1], dmax);
        }
    }
    if (!in_range) return;
    interior[v] = ok;
    flag32[v] = ok;
    if (!ok) {
        for (int i = s; i < e; i++) rows_ccw[i] = rows_asc[i];
        return;
    }
    const P2 p = load_p2(node_xy, (int)v);
    const int deg = e - s;
    if (deg <= VOR_CAP) {
        for (int i0 = 0; i0 < deg; i0 += 8) { // (eight at a time: the typical row of six in ONE pair of dependent round trips)
            int key[8];
            P2 c[8];
#pragma unroll
            for (int u = 0; u < 8; u++) key[u] = rows_asc[s + (i0 + u < deg ? i0 + u : deg - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] = load_p2(cxy, key[u]);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (i0 + u < deg) {
                    sh_k[i0 + u][t] = key[u];
                    sh_d[i0 + u][t] = make_double2(c[u].x - p.x, c[u].y - p.y);
                }
            }
        }
        for (int i = 1; i < deg; i++) { // insertion sort of the thread's own column (no other thread touches it)
            const int key = sh_k[i][t];
            const double2 k = sh_d[i][t];
            int j = i - 1;
            while (j >= 0) {
                const double2 d = sh_d[j][t];
                if (!angle_less(k.x, k.y, d.x, d.y)) break;
                sh_d[j + 1][t] = d;
                sh_k[j + 1][t] = sh_k[j][t];
                j--;
            }
            sh_d[j + 1][t] = k;
            sh_k[j + 1][t] = key;
        }
        for (int i = 0; i < deg; i++) rows_ccw[s + i] = sh_k[i][t];
        return;
    }
    for (int i = s; i < e; i++) { // long rows: insertion sort in global memory
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

// one thread per SLOT of the cell table (the lanes of a wave write consecutive words; a thread per node wrote its own
// 4 m bytes -- 64 partial lines per store instruction: 0.135 ms for a 44 MB table)
__global__ void __launch_bounds__(256)
k_vor_cells(const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows_ccw,
            const uint8_t *__restrict__ interior, const int32_t *__restrict__ cell_rank, int64_t n_node, int m,
            int32_t *__restrict__ cells) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_node * m) return;
    const int64_t v = i / m;
    const int j = (int)(i - v * m);
    if (!interior[v]) return;
    const int s = indptr[v], n = indptr[v + 1] - s;
    cells[(int64_t)cell_rank[v] * m + j] = j < n ? rows_ccw[s + j] : -1;
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

// everything the host needs about the boundary except the rows themselves, in ONE launch: per boundary node its degree and
// coordinates, per exterior edge the centroid of its face.  `in` = [nodes (nb) | edge faces (ne)]; `out` (8-byte words) =
// [degree (nb, int64) | node xy (2 nb) | edge-face xy (2 ne)]
__global__ void __launch_bounds__(256)
k_vor_boundary_info(const int32_t *__restrict__ indptr, const double *__restrict__ node_xy, const double *__restrict__ centroids,
                    const int64_t *__restrict__ in, int64_t nb, int64_t ne, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double *out_d = reinterpret_cast<double *>(out);
    if (i < nb) {
        const int64_t v = in[i];
        out[i] = indptr[v + 1] - indptr[v];
        out_d[nb + 2 * i] = node_xy[2 * v];
        out_d[nb + 2 * i + 1] = node_xy[2 * v + 1];
    }
    if (i < ne) {
        const int64_t f = in[nb + i];
        out_d[3 * nb + 2 * i] = centroids[2 * f];
        out_d[3 * nb + 2 * i + 1] = centroids[2 * f + 1];
    }
}

// exclusive scan of the degrees by one block (a few thousand boundary nodes): ptr[0..n].  256 threads, a wave per SIMD: a block
// of 1024 threads needs four free wave slots on every SIMD of ONE CU, and beside a kernel that fills the device (the locate pass of
// a barycentric construction runs on the side stream during this phase) it waited 200 us for them.
__global__ void __launch_bounds__(256) k_vor_scan_deg(const int64_t *__restrict__ deg, int64_t n, int64_t *__restrict__ ptr) {
    constexpr int NT = 256;
    __shared__ long long sh[NT];
    const int t = threadIdx.x;
    const int64_t chunk = (n + NT - 1) / NT, i0 = (int64_t)t * chunk < n ? (int64_t)t * chunk : n, i1 = i0 + chunk < n ? i0 + chunk : n;
    long long sum = 0;
    for (int64_t i = i0; i < i1; i++) sum += deg[i];
    sh[t] = sum;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {
        const long long add = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    long long run = sh[t] - sum; // exclusive prefix of this thread's chunk
    for (int64_t i = i0; i < i1; i++) {
        ptr[i] = run;
        run += deg[i];
    }
    if (t == NT - 1) ptr[n] = sh[NT - 1];
}

// rows of the boundary nodes: `out` (8-byte words) = [faces (total, int64) | their centroids (2 total)]
__global__ void k_vor_row_gather(const int32_t *__restrict__ indptr, const int32_t *__restrict__ faces_asc,
                                 const double *__restrict__ centroids, const int64_t *__restrict__ nodes,
                                 const int64_t *__restrict__ ptr, int64_t n, int64_t total, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double *xy = reinterpret_cast<double *>(out) + total;
    const int s = indptr[nodes[i]], e = indptr[nodes[i] + 1];
    for (int r = s; r < e; r++) {
        const int f = faces_asc[r];
        const int64_t o = ptr[i] + (r - s);
        out[o] = f;
        xy[2 * o] = centroids[2 * f];
        xy[2 * o + 1] = centroids[2 * f + 1];
    }
}

// Three host <-> device round trips (one upload, two downloads of packed buffers) instead of the nine single-array copies
// this used to make: each is a stream synchronisation, ~20 us with the device idle in between.
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
    v->b_node_xy.assign((size_t)nb * 2, 0.0);
    if (nb + ne > 0) {
        std::vector<int64_t> in(nodes);
        in.insert(in.end(), v->edge_face.begin(), v->edge_face.end());
        DevBuf<int64_t> d_in((size_t)(nb + ne)), d_info((size_t)(3 * nb + 2 * ne)), d_ptr((size_t)nb + 1);
        h2d(d_in.get(), in.data(), sizeof(int64_t) * (size_t)(nb + ne));
        XR_LAUNCH("vor_boundary_info", k_vor_boundary_info, dim3(div_up(std::max(nb, ne), 256)), dim3(256), 0, v->indptr.get(),
                  v->mesh->node_xy.get(), v->centroids.get(), d_in.get(), nb, ne, d_info.get());
        if (nb > 0) XR_LAUNCH("vor_scan_deg", k_vor_scan_deg, dim3(1), dim3(256), 0, d_info.get(), nb, d_ptr.get());
        std::vector<int64_t> info((size_t)(3 * nb + 2 * ne));
        d2h(info.data(), d_info.get(), sizeof(int64_t) * info.size());
        for (int64_t i = 0; i < nb; i++) v->b_ptr[(size_t)i + 1] = v->b_ptr[(size_t)i] + info[(size_t)i];
        if (nb > 0) memcpy(v->b_node_xy.data(), info.data() + nb, sizeof(double) * 2 * (size_t)nb);
        if (ne > 0) memcpy(v->b_edge_face_xy.data(), info.data() + 3 * nb, sizeof(double) * 2 * (size_t)ne);
        const int64_t total = v->b_ptr[(size_t)nb];
        v->b_faces.resize((size_t)total);
        v->b_face_xy.resize((size_t)total * 2);
        if (total > 0) {
            DevBuf<int64_t> d_rows((size_t)(3 * total));
            XR_LAUNCH("vor_row_gather", k_vor_row_gather, dim3(div_up(nb, 256)), dim3(256), 0, v->indptr.get(),
                      v->faces_asc.get(), v->centroids.get(), d_in.get(), d_ptr.get(), nb, total, d_rows.get());
            std::vector<int64_t> rows((size_t)(3 * total));
            d2h(rows.data(), d_rows.get(), sizeof(int64_t) * rows.size());
            memcpy(v->b_faces.data(), rows.data(), sizeof(int64_t) * (size_t)total);
            memcpy(v->b_face_xy.data(), rows.data() + total, sizeof(double) * 2 * (size_t)total);
        }
    }
    v->boundary_ready = true;
}


// ---------------------------------------------------------------------------------------------------------------------
// The cells of the boundary nodes (voronoi.py:59-327 for add_exterior = add_vertices = skip_concave = True): native host
// code on the few KB voronoi_boundary() gathered -- a LOCAL problem (boundary nodes 0..nb-1, the faces around them
// 0..nl-1, both ascending like their global ids, so every grouping and stable sort sees the order it would see globally).
// O(boundary) items in a dozen dependent steps (sorts, a scan, a per-cell convexity choice): tens of microseconds here; as
// device kernels each step would be a launch plus a round trip.  The arithmetic follows xugrid_amd/voronoi.py
// (_boundary_records) operation for operation, including numpy's pairwise summation in the polygon areas: on a straight
// boundary the two candidate areas of the convexity choice differ by rounding only.
// ---------------------------------------------------------------------------------------------------------------------
static double np_pairwise_sum(const double *a, int64_t n) { // numpy's DOUBLE_pairwise_sum for n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

static void voronoi_boundary_cells(xr_voronoi *v) {
    if (v->cells_ready) return;
    voronoi_boundary(v);
    // (the boundary rows are on the host now; whatever device work is waiting for a good moment -- the source-side half of a
    // barycentric construction -- goes out here, and runs while the host builds the boundary cells below)
    flush_pending_points();
    const auto tk0 = std::chrono::steady_clock::now();
    const int64_t nb = (int64_t)v->b_nodes.size(), ne = (int64_t)v->edge_face.size(), n_face = v->n_face;
    v->c_extra_xy.clear(); v->c_cells.clear(); v->c_tail.clear(); v->c_interp.clear();
    v->c_n_cell = 0; v->c_m = 0;
    if (ne == 0) { // closed surface: nothing to add
        v->cells_ready = true;
        return;
    }
    XR_REQUIRE(n_face + 3 * ne < ((int64_t)1 << 31), XR_ERR_LIMIT, "voronoi: vertex ids exceed the int32 range");
    // Vertex ids are GLOBAL from the start (face centroid f -> f, kept projection r -> n_face + r, extra corner k ->
    // n_face + n_proj + k): the local renumbering of the numpy restatement only exists to keep its arrays small; every
    // record carries its corner's coordinates, so no vertex table is needed either.  Cell keys are local node ranks
    // (ascending like the global node ids).
    struct Rec {
        int64_t key, id;
        double x, y, angle;
    };
    std::vector<Rec> rec;
    rec.reserve(v->b_faces.size() + 3 * (size_t)ne);
    for (int64_t i = 0; i < nb; i++) // corners that are face centroids: nodes shared by several faces ...
        if (v->b_ptr[(size_t)i + 1] - v->b_ptr[(size_t)i] > 1)
            for (int64_t r = v->b_ptr[(size_t)i]; r < v->b_ptr[(size_t)i + 1]; r++)
                rec.push_back({i, v->b_faces[(size_t)r], v->b_face_xy[2 * (size_t)r], v->b_face_xy[2 * (size_t)r + 1], 0.0});
    for (int64_t i = 0; i < nb; i++) // ... then corner nodes owned by exactly one face
        if (v->b_ptr[(size_t)i + 1] - v->b_ptr[(size_t)i] == 1) {
            const size_t r = (size_t)v->b_ptr[(size_t)i];
            rec.push_back({i, v->b_faces[r], v->b_face_xy[2 * r], v->b_face_xy[2 * r + 1], 0.0});
        }
    // projections of the adjacent face centroid on every exterior edge
    auto local_node = [&](int64_t g) { return (int64_t)(std::lower_bound(v->b_nodes.begin(), v->b_nodes.end(), g) - v->b_nodes.begin()); };
    const double *nxy = v->b_node_xy.data();
    const double merge_tol = 1.0e-8 * 1.0e-8;
    std::vector<int64_t> e_n0((size_t)ne), e_n1((size_t)ne), kept_rank((size_t)ne, -1);
    std::vector<double> proj_all((size_t)ne * 2);
    int64_t n_proj = 0;
    for (int64_t e = 0; e < ne; e++) {
        e_n0[(size_t)e] = local_node(v->edge_lo[(size_t)e]);
        e_n1[(size_t)e] = local_node(v->edge_hi[(size_t)e]);
        const double ax = nxy[2 * e_n0[(size_t)e]], ay = nxy[2 * e_n0[(size_t)e] + 1];
        const double bx = nxy[2 * e_n1[(size_t)e]], by = nxy[2 * e_n1[(size_t)e] + 1];
        const double cx = v->b_edge_face_xy[2 * (size_t)e], cy = v->b_edge_face_xy[2 * (size_t)e + 1];
        const double vx = bx - ax, vy = by - ay, ux = cx - ax, uy = cy - ay;
        const double sc = (ux * vx + uy * vy) / (vx * vx + vy * vy);
        const double px = ax + sc * vx, py = ay + sc * vy;
        proj_all[2 * (size_t)e] = px;
        proj_all[2 * (size_t)e + 1] = py;
        const double dx = px - cx, dy = py - cy;
        if (std::sqrt(dx * dx + dy * dy) > merge_tol) kept_rank[(size_t)e] = n_proj++;
    }
    const int64_t first_new = n_face + n_proj; // id of the first extra corner
    for (int64_t e = 0; e < ne; e++)
        if (kept_rank[(size_t)e] >= 0) { // both end nodes use the projection
            const int64_t id = n_face + kept_rank[(size_t)e];
            rec.push_back({e_n0[(size_t)e], id, proj_all[2 * (size_t)e], proj_all[2 * (size_t)e + 1], 0.0});
            rec.push_back({e_n1[(size_t)e], id, proj_all[2 * (size_t)e], proj_all[2 * (size_t)e + 1], 0.0});
            v->c_tail.push_back(v->edge_face[(size_t)e]);
        }
    // one extra corner per boundary node, between the node's two projections: the (edge, end) records sorted by node id
    // (stable), paired off two by two; the interpolation map refers to the UNFILTERED projection numbering exactly as the
    // reference does
    std::vector<int64_t> by_node((size_t)(2 * ne));
    for (int64_t j = 0; j < 2 * ne; j++) by_node[(size_t)j] = j;
    auto flat_node = [&](int64_t j) { return (j & 1) ? e_n1[(size_t)(j >> 1)] : e_n0[(size_t)(j >> 1)]; };
    std::stable_sort(by_node.begin(), by_node.end(), [&](int64_t a, int64_t b) { return flat_node(a) < flat_node(b); });
    const int64_t n_extra = ne; // (2 ne records, two per extra corner)
    std::vector<double> extra_xy((size_t)n_extra * 2), true_corner((size_t)n_extra * 2);
    v->c_interp.resize((size_t)n_extra * 2);
    for (int64_t k = 0; k < n_extra; k++) {
        const int64_t p0 = by_node[(size_t)(2 * k)] >> 1, p1 = by_node[(size_t)(2 * k + 1)] >> 1;
        extra_xy[2 * (size_t)k] = 0.5 * (proj_all[2 * (size_t)p0] + proj_all[2 * (size_t)p1]);
        extra_xy[2 * (size_t)k + 1] = 0.5 * (proj_all[2 * (size_t)p0 + 1] + proj_all[2 * (size_t)p1 + 1]);
        const int64_t node = flat_node(by_node[(size_t)(2 * k)]);
        rec.push_back({node, first_new + k, extra_xy[2 * (size_t)k], extra_xy[2 * (size_t)k + 1], 0.0});
        v->c_interp[2 * (size_t)k] = p0 + n_face;
        v->c_interp[2 * (size_t)k + 1] = p1 + n_face;
        true_corner[2 * (size_t)k] = nxy[2 * node];
        true_corner[2 * (size_t)k + 1] = nxy[2 * node + 1];
    }
    v->c_tail.insert(v->c_tail.end(), (size_t)n_extra, (int64_t)-1);
    const auto tk1 = std::chrono::steady_clock::now();
    // ---- counter-clockwise order about the mean of each cell's corners (sums in record order, as np.bincount)
    const int64_t n_rec = (int64_t)rec.size();
    std::vector<double> sx((size_t)nb, 0.0), sy((size_t)nb, 0.0), cnt((size_t)nb, 0.0);
    for (const Rec &r : rec) {
        sx[(size_t)r.key] += r.x;
        sy[(size_t)r.key] += r.y;
        cnt[(size_t)r.key] += 1.0;
    }
    for (Rec &r : rec) {
        const double px = sx[(size_t)r.key] / cnt[(size_t)r.key], py = sy[(size_t)r.key] / cnt[(size_t)r.key];
        r.angle = std::atan2(r.y - py, r.x - px);
    }
    std::stable_sort(rec.begin(), rec.end(), [](const Rec &a, const Rec &b) {
        if (a.key != b.key) return a.key < b.key;
        return a.angle < b.angle;
    });
    const auto tk2 = std::chrono::steady_clock::now();
    // ---- dense table: one row per distinct key (ascending), -1 padded
    std::vector<int64_t> row_start;
    for (int64_t r = 0; r < n_rec; r++)
        if (r == 0 || rec[(size_t)r].key != rec[(size_t)r - 1].key) row_start.push_back(r);
    const int64_t n_cell = (int64_t)row_start.size();
    row_start.push_back(n_rec);
    int64_t m = 0;
    for (int64_t c = 0; c < n_cell; c++) m = std::max(m, row_start[(size_t)c + 1] - row_start[(size_t)c]);
    XR_REQUIRE(m <= 128, XR_ERR_LIMIT, "voronoi: a boundary cell has %lld corners", (long long)m);
    std::vector<int64_t> cells((size_t)(n_cell * m), -1);
    // ---- keep the true boundary node where it does not make the cell concave: the cell's area with the true node against
    // its area with the midpoint substitute (closed polygon: fill slots and the closing slot repeat corner 0)
    std::vector<double> term((size_t)m);
    for (int64_t c = 0; c < n_cell; c++) {
        const Rec *row = rec.data() + row_start[(size_t)c];
        const int64_t len = row_start[(size_t)c + 1] - row_start[(size_t)c];
        for (int64_t j = 0; j < len; j++) cells[(size_t)(c * m + j)] = row[j].id;
        auto vertex = [&](int64_t j, bool use_true, double &x, double &y) { // corner j of the closed polygon
            const Rec &r = row[j < len ? j : 0];
            if (use_true && r.id >= first_new) {
                x = true_corner[2 * (size_t)(r.id - first_new)];
                y = true_corner[2 * (size_t)(r.id - first_new) + 1];
            } else {
                x = r.x;
                y = r.y;
            }
        };
        double area[2];
        for (int t = 0; t < 2; t++) {
            double x0, y0;
            vertex(0, t == 1, x0, y0);
            for (int64_t i = 0; i < m; i++) { // closed[i], closed[i + 1] with closed[m] = corner 0
                double xa, ya, xb, yb;
                vertex(i, t == 1, xa, ya);
                vertex(i + 1 < m ? i + 1 : len, t == 1, xb, yb);
                const double a0 = xa - x0, a1 = ya - y0, b0 = xb - x0, b1 = yb - y0;
                term[(size_t)i] = a0 * b1 - a1 * b0;
            }
            area[t] = 0.5 * std::fabs(np_pairwise_sum(term.data(), m));
        }
        if (area[1] >= area[0])
            for (int64_t j = 0; j < len; j++)
                if (row[j].id >= first_new) {
                    const int64_t k = row[j].id - first_new;
                    extra_xy[2 * (size_t)k] = true_corner[2 * (size_t)k];
                    extra_xy[2 * (size_t)k + 1] = true_corner[2 * (size_t)k + 1];
                }
    }
    const auto tk3 = std::chrono::steady_clock::now();
    if (option(OPT_DEBUG) & 4)
        fprintf(stderr, "[voronoi] records %lld: set-up %.3f, sort %.3f, cells + convexity %.3f ms\n", (long long)n_rec,
                std::chrono::duration<double, std::milli>(tk1 - tk0).count(), std::chrono::duration<double, std::milli>(tk2 - tk1).count(),
                std::chrono::duration<double, std::milli>(tk3 - tk2).count());
    v->c_cells = std::move(cells);
    v->c_n_cell = n_cell;
    v->c_m = m;
    v->c_extra_xy.resize((size_t)(n_proj + n_extra) * 2);
    for (int64_t e = 0; e < ne; e++)
        if (kept_rank[(size_t)e] >= 0) {
            v->c_extra_xy[2 * (size_t)kept_rank[(size_t)e]] = proj_all[2 * (size_t)e];
            v->c_extra_xy[2 * (size_t)kept_rank[(size_t)e] + 1] = proj_all[2 * (size_t)e + 1];
        }
    std::copy(extra_xy.begin(), extra_xy.end(), v->c_extra_xy.begin() + 2 * n_proj);
    v->cells_ready = true;
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
        DevBuf<int32_t> count_cursor(2 * ((size_t)N + 1)); // (histogram and scatter cursors: one buffer, one fill)
        int32_t *const count = count_cursor.get(), *const cursor = count_cursor.get() + N + 1;
        v->indptr.alloc((size_t)N + 1);
        fill_i32(count_cursor.get(), 0, 2 * (N + 1));
        if (total > 0)
            XR_LAUNCH("vor_count", k_vor_count, dim3(div_up(total, VOR_SLOTS)), dim3(256), 0, mesh->faces_raw.get(), total, count);
        exclusive_scan_i32(count, v->indptr.get(), N);
        // (sized by the slot count, an upper bound of the entries: their number is read at the end with the other counters --
        // the whole construction has TWO host round trips, counters and exterior edges, instead of seven)
        v->faces_asc.alloc((size_t)std::max<int64_t>(total, 1));
        v->faces_ccw.alloc((size_t)std::max<int64_t>(total, 1));
        v->interior.alloc((size_t)N);
        v->cell_rank.alloc((size_t)N + 1);
        v->centroids.alloc((size_t)F * 2);
        DevBuf<uint8_t> on_boundary((size_t)N);
        DevBuf<int32_t> flag32((size_t)N), counters(8), e_all((size_t)std::max<int64_t>(3 * total, 1));
        XR_HIP(hipMemsetAsync(on_boundary.get(), 0, (size_t)(N > 0 ? N : 1), launch_stream()));
        // counters: [0] exterior edges, [1] min / [2] max interior degree, [3] interior nodes, [4] entries of the node -> face rows
        XR_LAUNCH("vor_init", k_vor_init_counters, dim3(1), dim3(64), 0, counters.get());
        if (total > 0) {
            XR_LAUNCH("vor_scatter", k_vor_scatter, dim3(div_up(total, VOR_SLOTS)), dim3(256), 0, mesh->faces_raw.get(), total, m,
                      v->indptr.get(), cursor, v->faces_asc.get());
            XR_LAUNCH("vor_sort_rows", k_vor_sort_rows, dim3(div_up(N, 256)), dim3(256), 0, v->indptr.get(), N,
                      v->faces_asc.get());
            XR_LAUNCH("vor_exterior", k_vor_exterior, dim3(div_up(total, 256)), dim3(256), 0, mesh->faces_raw.get(), F, m,
                      v->indptr.get(), v->faces_asc.get(), on_boundary.get(), counters.get(), e_all.get());
        }
        mesh_centroids_dev(mesh, v->centroids.get());
        if (N > 0)
            XR_LAUNCH("vor_interior", k_vor_interior, dim3(div_up(N, VOR_BLOCK)), dim3(VOR_BLOCK), 0, mesh->node_xy.get(),
                      v->centroids.get(), v->indptr.get(), v->faces_asc.get(), on_boundary.get(), N, v->faces_ccw.get(),
                      v->interior.get(), flag32.get(), counters.get() + 1);
        exclusive_scan_i32(flag32.get(), v->cell_rank.get(), N);
        XR_LAUNCH("vor_totals", k_vor_totals, dim3(1), dim3(64), 0, v->cell_rank.get() + N, v->indptr.get() + N, counters.get());
        // (every kernel of the O(n) part is enqueued; from here on the host reads counters and lists back -- the device idles for
        // most of the next 0.2-0.3 ms.  A pending source-side locate pass of a barycentric construction goes out here.)
        flush_pending_points();
        int32_t h[8];
        d2h(h, counters.get(), sizeof(h));
        v->n_interior = h[3];
        v->nnz = h[4];
        v->min_degree = v->n_interior > 0 ? h[1] : 0;
        v->max_degree = h[2];
        const int64_t ne = h[0];
        std::vector<int32_t> lo((size_t)ne), hi((size_t)ne), fc((size_t)ne);
        if (ne > 0) {
            std::vector<int32_t> all((size_t)(3 * ne));
            d2h(all.data(), e_all.get(), sizeof(int32_t) * all.size());
            for (int64_t i = 0; i < ne; i++) {
                lo[(size_t)i] = all[3 * (size_t)i];
                hi[(size_t)i] = all[3 * (size_t)i + 1];
                fc[(size_t)i] = all[3 * (size_t)i + 2];
            }
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
            XR_LAUNCH("vor_cells", k_vor_cells, dim3(div_up(v->n_node * m, 256)), dim3(256), 0, v->indptr.get(),
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

/* The whole pre-step without a host language in between: the cells of the boundary nodes are computed by the library
 * (voronoi_boundary_cells) and the tessellation is assembled on the device.  n_tail = vertices added behind the face
 * centroids, n_map = rows of the interpolation map; fetch them with xr_voronoi_tail. */
int xr_voronoi_mesh_auto(xr_voronoi *v, xr_mesh **out, int64_t *n_tail, int64_t *n_map) {
    XR_API_BEGIN
    XR_REQUIRE(v && out && n_tail && n_map, XR_ERR_INVALID, "xr_voronoi_mesh_auto: NULL argument");
    const bool dbg = (option(OPT_DEBUG) & 4) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    voronoi_boundary(v);
    const auto t1 = std::chrono::steady_clock::now();
    voronoi_boundary_cells(v);
    const auto t2 = std::chrono::steady_clock::now();
    if (dbg)
        fprintf(stderr, "[voronoi] gather %.3f ms, boundary cells %.3f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    *n_tail = (int64_t)v->c_tail.size();
    *n_map = (int64_t)v->c_interp.size() / 2;
    const int rc = xr_voronoi_mesh(v, v->c_extra_xy.data(), (int64_t)v->c_extra_xy.size() / 2, v->c_cells.data(), v->c_n_cell,
                                   v->c_m, out);
    if (rc != XR_OK) return rc;
    XR_API_END
}

int xr_voronoi_tail(xr_voronoi *v, int64_t *tail_face_index, int64_t *interpolation_map) {
    XR_API_BEGIN
    XR_REQUIRE(v, XR_ERR_INVALID, "xr_voronoi_tail: NULL argument");
    voronoi_boundary_cells(v);
    XR_REQUIRE((v->c_tail.empty() || tail_face_index) && (v->c_interp.empty() || interpolation_map), XR_ERR_INVALID,
               "xr_voronoi_tail: NULL output array");
    if (!v->c_tail.empty()) memcpy(tail_face_index, v->c_tail.data(), sizeof(int64_t) * v->c_tail.size());
    if (!v->c_interp.empty()) memcpy(interpolation_map, v->c_interp.data(), sizeof(int64_t) * v->c_interp.size());
    XR_API_END
}

/* the boundary cells themselves (tests, hosts that assemble on their own): sizes, then the arrays */
int xr_voronoi_boundary_cells_info(xr_voronoi *v, int64_t *n_extra_vertex, int64_t *n_cell, int64_t *n_max) {
    XR_API_BEGIN
    XR_REQUIRE(v && n_extra_vertex && n_cell && n_max, XR_ERR_INVALID, "xr_voronoi_boundary_cells_info: NULL argument");
    voronoi_boundary_cells(v);
    *n_extra_vertex = (int64_t)v->c_extra_xy.size() / 2;
    *n_cell = v->c_n_cell;
    *n_max = v->c_m;
    XR_API_END
}

int xr_voronoi_boundary_cells(xr_voronoi *v, double *extra_xy, int64_t *cells) {
    XR_API_BEGIN
    XR_REQUIRE(v, XR_ERR_INVALID, "xr_voronoi_boundary_cells: NULL argument");
    voronoi_boundary_cells(v);
    XR_REQUIRE((v->c_extra_xy.empty() || extra_xy) && (v->c_cells.empty() || cells), XR_ERR_INVALID,
               "xr_voronoi_boundary_cells: NULL output array");
    if (!v->c_extra_xy.empty()) memcpy(extra_xy, v->c_extra_xy.data(), sizeof(double) * v->c_extra_xy.size());
    if (!v->c_cells.empty()) memcpy(cells, v->c_cells.data(), sizeof(int64_t) * v->c_cells.size());
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
