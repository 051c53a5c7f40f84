// xr_shard.hip -- the set-up of a rank of the source-sharded regridder on the device ("source faces partitioned over
// the GPUs, target replicated"; the reference's only multi-worker analogue is the dask loop of xugrid/regrid/regridder.py:167-185):
// which source faces are mine, which target faces can receive weight from them.  Until round 4 this was a page of torch tensor
// operations (argsort over all centroids, index_put rasters, nonzero: ~6 ms per rebuild of a 1M + 1M pair, ten times the step it
// prepares); here it is a dozen O(S + T) kernels, integer arithmetic wherever ranks have to agree, ONE host round trip (the two
// list lengths).
//
// The rule (every rank evaluates it on its own device and must get the same owner for every face, so everything that decides an
// owner is exact):
//   * centroid of a face = sum of its valid nodes in connectivity order / their number
//   * "hash": owner = face id mod world
//   * "morton" / "balanced": the source centroids' bounding square is cut into 1024 x 1024 cells, cells are ordered along the
//     Morton curve, every face adds its WORK to its cell -- 1 ("morton": equal counts) or the fixed-point value
//     round(4096 (1 + 4 n_tgt / n_src)) of the coarse work raster both meshes were counted into ("balanced": a rank's cost is
//     ~1 per source face + ~4 per target face it meets) -- and owner(cell) = floor(work in front of the cell * world / total work).
//     All faces of a cell share the owner: the cut is exact to one cell of ~1-8 faces, no sort of the faces is needed, and
//     64-bit integer sums do not depend on the order of the atomics.
//   * a target face is kept iff its box touches an occupied cell of the 128 x 128 occupancy raster of MY source faces' boxes
//     (difference array + two prefix sums + integral image, all int32; conservative at cell granularity, not spoiled by a few
//     long hull slivers the way one box per shard would be).
#include "xr_internal.h"

namespace xr {

static constexpr int SHARD_BITS = 10;                 // Morton cells per axis: 2^10
static constexpr int SHARD_CELLS = 1 << (2 * SHARD_BITS);
static constexpr int OCC_GRID = 128;                  // occupancy raster of the near-shard filter
static constexpr int OCC_STRIDE = OCC_GRID + 1;

// doubles as order-preserving unsigned keys: atomicMin / atomicMax on them are exact and order-independent
__device__ __forceinline__ unsigned long long dkey(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
struct Bounds { // [0..3] min x, min y, max x, max y as keys
    unsigned long long k[4];
};
// centroid + box of face f of a dense (F, m) int64 connectivity with -1 fill
// -> number of valid nodes (a face without any: centroid (0, 0), an empty box -- it takes no part in the bounds)
__device__ __forceinline__ int face_geometry(const double *__restrict__ xy, const int64_t *__restrict__ faces, int64_t f, int m,
                                             double &cx, double &cy, double &x0, double &y0, double &x1, double &y1) {
    double sx = 0.0, sy = 0.0;
    int n = 0;
    x0 = y0 = INFINITY;
    x1 = y1 = -INFINITY;
    for (int j = 0; j < m; j++) {
        const int64_t v = faces[f * m + j];
        if (v < 0) continue;
        const double x = xy[2 * v], y = xy[2 * v + 1];
        sx += x;
        sy += y;
        n++;
        x0 = fmin(x0, x); x1 = fmax(x1, x);
        y0 = fmin(y0, y); y1 = fmax(y1, y);
    }
    cx = sx / (double)(n > 0 ? n : 1);
    cy = sy / (double)(n > 0 ? n : 1);
    return n;
}

// Bounds of a set of boxes, two stages: every block reduces its boxes (waves by shuffles, the block through LDS) and writes ONE
// partial (min x, min y, max x, max y) to its slot; a one-block kernel folds the partials.  (One set of same-address atomics
// per wave -- 375 k of them for a 1M + 1M pair -- cost 3 ms: the L2 serves a contended word at a few hundred million a second.)
__device__ __forceinline__ void bounds_block_partial(double4 *__restrict__ partial, bool valid, double x0, double y0, double x1, double y1) {
    __shared__ double4 sh[4];
    double a = valid ? x0 : INFINITY, c = valid ? y0 : INFINITY, d = valid ? x1 : -INFINITY, e = valid ? y1 : -INFINITY;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        a = fmin(a, __shfl_xor(a, s, 64));
        c = fmin(c, __shfl_xor(c, s, 64));
        d = fmax(d, __shfl_xor(d, s, 64));
        e = fmax(e, __shfl_xor(e, s, 64));
    }
    __syncthreads(); // (the LDS slots may still be read from an earlier call)
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = make_double4(a, c, d, e);
    __syncthreads();
    if (threadIdx.x == 0) {
        double4 r = sh[0];
        for (int w = 1; w < 4; w++) {
            r.x = fmin(r.x, sh[w].x); r.y = fmin(r.y, sh[w].y);
            r.z = fmax(r.z, sh[w].z); r.w = fmax(r.w, sh[w].w);
        }
        partial[blockIdx.x] = r;
    }
}
// partials [n_sets][n_blocks] -> Bounds[n_sets] (one block of 256 threads per set)
__global__ void __launch_bounds__(256) k_shard_fold_bounds(const double4 *__restrict__ partial, int64_t n_blocks, Bounds *__restrict__ out) {
    __shared__ double4 sh[4];
    const double4 *p = partial + (int64_t)blockIdx.x * n_blocks;
    double a = INFINITY, c = INFINITY, d = -INFINITY, e = -INFINITY;
    for (int64_t i = threadIdx.x; i < n_blocks; i += 256) {
        const double4 v = p[i];
        a = fmin(a, v.x); c = fmin(c, v.y); d = fmax(d, v.z); e = fmax(e, v.w);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        a = fmin(a, __shfl_xor(a, s, 64));
        c = fmin(c, __shfl_xor(c, s, 64));
        d = fmax(d, __shfl_xor(d, s, 64));
        e = fmax(e, __shfl_xor(e, s, 64));
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = make_double4(a, c, d, e);
    __syncthreads();
    if (threadIdx.x == 0) {
        double4 r = sh[0];
        for (int w = 1; w < 4; w++) {
            r.x = fmin(r.x, sh[w].x); r.y = fmin(r.y, sh[w].y);
            r.z = fmax(r.z, sh[w].z); r.w = fmax(r.w, sh[w].w);
        }
        out[blockIdx.x].k[0] = dkey(r.x); out[blockIdx.x].k[1] = dkey(r.y);
        out[blockIdx.x].k[2] = dkey(r.z); out[blockIdx.x].k[3] = dkey(r.w);
    }
}

// pass 1: centroids of both meshes (stored) and their bounds
__global__ void __launch_bounds__(256)
k_shard_centroids(const double *__restrict__ sxy, const int64_t *__restrict__ sf, int64_t S, int ms, const double *__restrict__ txy,
                  const int64_t *__restrict__ tf, int64_t T, int mt, double2 *__restrict__ scen, double2 *__restrict__ tcen,
                  double4 *__restrict__ partial /* [2][gridDim.x]: source centroids, target centroids */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool is_src = i < S, valid = i < S + T;
    double cx = 0, cy = 0, x0, y0, x1, y1;
    int n_nodes = 0;
    if (is_src) n_nodes = face_geometry(sxy, sf, i, ms, cx, cy, x0, y0, x1, y1);
    else if (valid) n_nodes = face_geometry(txy, tf, i - S, mt, cx, cy, x0, y0, x1, y1);
    if (is_src) scen[i] = make_double2(cx, cy);
    else if (valid) tcen[i - S] = make_double2(cx, cy);
    // (a block holds faces of one mesh except the one straddling S: the two reductions are masked per mesh; a face without a
    // valid node has no centroid -- its (0, 0) must not stretch the bounds the Morton cells are cut from)
    bounds_block_partial(partial, is_src && n_nodes > 0, cx, cy, cx, cy);
    bounds_block_partial(partial + gridDim.x, valid && !is_src && n_nodes > 0, cx, cy, cx, cy);
}

__device__ __forceinline__ int raster_cell(double v, double lo, double f, int n) {
    const double c = floor((v - lo) * f);
    if (!(c > 0.0)) return 0;
    return c >= (double)(n - 1) ? n - 1 : (int)c;
}

// pass 2: the coarse work raster -- source and target faces counted per cell
__global__ void __launch_bounds__(256)
k_shard_count(const double2 *__restrict__ scen, int64_t S, const double2 *__restrict__ tcen, int64_t T, const Bounds *__restrict__ b,
              int n_grid, int32_t *__restrict__ n_src, int32_t *__restrict__ n_tgt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S + T) return;
    const double lox = fmin(dkey_inv(b[0].k[0]), dkey_inv(b[1].k[0])), loy = fmin(dkey_inv(b[0].k[1]), dkey_inv(b[1].k[1]));
    const double hix = fmax(dkey_inv(b[0].k[2]), dkey_inv(b[1].k[2])), hiy = fmax(dkey_inv(b[0].k[3]), dkey_inv(b[1].k[3]));
    const double fx = (double)n_grid / fmax(hix - lox, 1e-300), fy = (double)n_grid / fmax(hiy - loy, 1e-300);
    const double2 c = i < S ? scen[i] : tcen[i - S];
    const int cell = raster_cell(c.y, loy, fy, n_grid) * n_grid + raster_cell(c.x, lox, fx, n_grid);
    atomicAdd(i < S ? &n_src[cell] : &n_tgt[cell], 1);
}

__device__ __forceinline__ uint32_t shard_spread(uint32_t v) { // 10 bits -> every second bit
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// pass 3: Morton cell of every source face and its work, summed per cell (64-bit integer atomics: exact in any order)
__global__ void __launch_bounds__(256)
k_shard_work(const double2 *__restrict__ scen, int64_t S, const Bounds *__restrict__ b, int n_grid, const int32_t *__restrict__ n_src,
             const int32_t *__restrict__ n_tgt, bool balanced, int32_t *__restrict__ mcell, unsigned long long *__restrict__ cell_work) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const double2 c = scen[i];
    const double slox = dkey_inv(b[0].k[0]), sloy = dkey_inv(b[0].k[1]), shix = dkey_inv(b[0].k[2]), shiy = dkey_inv(b[0].k[3]);
    const int side = 1 << SHARD_BITS;
    const int qx = raster_cell(c.x, slox, (double)side / fmax(shix - slox, 1e-300), side);
    const int qy = raster_cell(c.y, sloy, (double)side / fmax(shiy - sloy, 1e-300), side);
    const int mc = (int)(shard_spread((uint32_t)qx) | (shard_spread((uint32_t)qy) << 1));
    mcell[i] = mc;
    unsigned long long w = 1;
    if (balanced) {
        const double lox = fmin(slox, dkey_inv(b[1].k[0])), loy = fmin(sloy, dkey_inv(b[1].k[1]));
        const double hix = fmax(shix, dkey_inv(b[1].k[2])), hiy = fmax(shiy, dkey_inv(b[1].k[3]));
        const double fx = (double)n_grid / fmax(hix - lox, 1e-300), fy = (double)n_grid / fmax(hiy - loy, 1e-300);
        const int cell = raster_cell(c.y, loy, fy, n_grid) * n_grid + raster_cell(c.x, lox, fx, n_grid);
        const int ns = n_src[cell] > 0 ? n_src[cell] : 1;
        const double cost = (1.0 + 4.0 * (double)n_tgt[cell] / (double)ns) * 4096.0;
        // (clamped to 2^20 per face -- 63 target faces per source face of the cell --: with S < 2^31 faces the total stays below
        // 2^51 and `before * world` in k_shard_owner below 2^63 for world <= 4096, which xr_shard_plan_dev requires)
        const double r = fmin(rint(cost), 1048576.0);
        w = r >= 1.0 ? (unsigned long long)r : 1ull;
    }
    atomicAdd(&cell_work[mc], w);
}

// exclusive scan of the 2^20 cell sums (uint64): three small launches
static constexpr int S64_TILE = 1024;
__global__ void __launch_bounds__(256) k_scan64_reduce(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long sh[256];
    const int64_t base = (int64_t)blockIdx.x * S64_TILE;
    unsigned long long s = 0;
    for (int j = 0; j < 4; j++) s += in[base + threadIdx.x * 4 + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(1024) k_scan64_partials(unsigned long long *__restrict__ sums, int n, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long sh[1024];
    const unsigned long long v = (int)threadIdx.x < n ? sums[threadIdx.x] : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // (Hillis-Steele: 1024 partial sums)
        const unsigned long long add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0ull;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if ((int)threadIdx.x < n) sums[threadIdx.x] = sh[threadIdx.x] - v;
    if (threadIdx.x == 1023) *total = sh[1023];
}
__global__ void __launch_bounds__(256) k_scan64_apply(unsigned long long *__restrict__ data, const unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long sh[256];
    const int64_t base = (int64_t)blockIdx.x * S64_TILE + threadIdx.x * 4;
    unsigned long long v[4], s = 0;
    for (int j = 0; j < 4; j++) {
        v[j] = data[base + j];
        s += v[j];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned long long add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0ull;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = sums[blockIdx.x] + sh[threadIdx.x] - s;
    for (int j = 0; j < 4; j++) {
        data[base + j] = run;
        run += v[j];
    }
}

// pass 4: the owner of every source face -> flag of MY faces (+ the owners themselves on request)
__global__ void __launch_bounds__(256)
k_shard_owner(int64_t S, const int32_t *__restrict__ mcell, const unsigned long long *__restrict__ before,
              const unsigned long long *__restrict__ total, int world, int rank, bool hash, int32_t *__restrict__ flag,
              int32_t *__restrict__ owner_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    int o;
    if (hash) {
        o = (int)(i % world);
    } else {
        const unsigned long long t = *total > 0 ? *total : 1ull;
        const unsigned long long q = before[mcell[i]] * (unsigned long long)world / t; // (work < 2^51 by the clamp above, world <= 2^12: no overflow)
        o = q < (unsigned long long)world ? (int)q : world - 1;
    }
    flag[i] = o == rank ? 1 : 0;
    if (owner_out) owner_out[i] = o;
}

__global__ void __launch_bounds__(256)
k_shard_compact(const int32_t *__restrict__ flag, const int32_t *__restrict__ pos, int64_t n, int64_t *__restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) ids[pos[i]] = i;
}

// pass 5: bounds of the boxes of MY source faces and of all target faces
__global__ void __launch_bounds__(256)
k_shard_box_bounds(const double *__restrict__ sxy, const int64_t *__restrict__ sf, int64_t S, int ms, const int32_t *__restrict__ mine,
                   const double *__restrict__ txy, const int64_t *__restrict__ tf, int64_t T, int mt, double4 *__restrict__ partial) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool valid = false;
    double cx, cy, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (i < S) {
        if (mine[i]) {
            face_geometry(sxy, sf, i, ms, cx, cy, x0, y0, x1, y1);
            valid = x0 <= x1;
        }
    } else if (i < S + T) {
        face_geometry(txy, tf, i - S, mt, cx, cy, x0, y0, x1, y1);
        valid = x0 <= x1;
    }
    bounds_block_partial(partial, valid, x0, y0, x1, y1);
}

// pass 6: difference array of my source faces' boxes on the occupancy raster
__global__ void __launch_bounds__(256)
k_shard_occupy(const double *__restrict__ sxy, const int64_t *__restrict__ sf, int64_t S, int ms, const int32_t *__restrict__ mine,
               const Bounds *__restrict__ b, int32_t *__restrict__ diff /* [OCC_STRIDE^2], zeroed */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S || !mine[i]) return;
    double cx, cy, x0, y0, x1, y1;
    face_geometry(sxy, sf, i, ms, cx, cy, x0, y0, x1, y1);
    if (!(x0 <= x1)) return;
    const double lox = dkey_inv(b[2].k[0]), loy = dkey_inv(b[2].k[1]);
    const double fx = (double)OCC_GRID / fmax(dkey_inv(b[2].k[2]) - lox, 1e-300), fy = (double)OCC_GRID / fmax(dkey_inv(b[2].k[3]) - loy, 1e-300);
    const int cx0 = raster_cell(x0, lox, fx, OCC_GRID), cx1 = raster_cell(x1, lox, fx, OCC_GRID);
    const int cy0 = raster_cell(y0, loy, fy, OCC_GRID), cy1 = raster_cell(y1, loy, fy, OCC_GRID);
    atomicAdd(&diff[cy0 * OCC_STRIDE + cx0], 1);
    atomicAdd(&diff[cy0 * OCC_STRIDE + cx1 + 1], -1);
    atomicAdd(&diff[(cy1 + 1) * OCC_STRIDE + cx0], -1);
    atomicAdd(&diff[(cy1 + 1) * OCC_STRIDE + cx1 + 1], 1);
}

// difference array -> occupied cells -> integral image (one block of 16 waves; a wave takes a line -- a row, then a column -- and
// scans it 64 elements at a time by shuffles: four passes of 129 lines over the 129 x 129 words)
__device__ __forceinline__ void scan_line(int32_t *line, int stride, int n, int lane) {
    int carry = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        int v = c < n ? line[c * stride] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(v, d, 64);
            if (lane >= d) v += t;
        }
        v += carry;
        if (c < n) line[c * stride] = v;
        carry = __shfl(v, 63, 64);
    }
}
__global__ void __launch_bounds__(1024) k_shard_integral(int32_t *__restrict__ diff, int32_t *__restrict__ integral) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < OCC_STRIDE; r += 16) scan_line(diff + r * OCC_STRIDE, 1, OCC_STRIDE, lane); // along x
    __syncthreads();
    for (int c = wave; c < OCC_STRIDE; c += 16) scan_line(diff + c, OCC_STRIDE, OCC_STRIDE, lane);     // along y: coverage counts
    __syncthreads();
    // integral[r + 1][c + 1] = occupied(r, c); row 0 and column 0 stay zero; then the two prefix sums
    for (int i = threadIdx.x; i < OCC_STRIDE * OCC_STRIDE; i += 1024) {
        const int r = i / OCC_STRIDE, c = i - r * OCC_STRIDE;
        integral[i] = (r > 0 && c > 0 && diff[(r - 1) * OCC_STRIDE + (c - 1)] > 0) ? 1 : 0;
    }
    __syncthreads();
    for (int r = wave; r < OCC_STRIDE; r += 16) scan_line(integral + r * OCC_STRIDE, 1, OCC_STRIDE, lane);
    __syncthreads();
    for (int c = wave; c < OCC_STRIDE; c += 16) scan_line(integral + c, OCC_STRIDE, OCC_STRIDE, lane);
}

// pass 7: the target faces whose box touches an occupied cell
__global__ void __launch_bounds__(256)
k_shard_targets(const double *__restrict__ txy, const int64_t *__restrict__ tf, int64_t T, int mt, const Bounds *__restrict__ b,
                const int32_t *__restrict__ integral, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    double cx, cy, x0, y0, x1, y1;
    face_geometry(txy, tf, i, mt, cx, cy, x0, y0, x1, y1);
    int hit = 0;
    if (x0 <= x1) {
        const double lox = dkey_inv(b[2].k[0]), loy = dkey_inv(b[2].k[1]);
        const double fx = (double)OCC_GRID / fmax(dkey_inv(b[2].k[2]) - lox, 1e-300), fy = (double)OCC_GRID / fmax(dkey_inv(b[2].k[3]) - loy, 1e-300);
        const int qx0 = raster_cell(x0, lox, fx, OCC_GRID), qx1 = raster_cell(x1, lox, fx, OCC_GRID) + 1;
        const int qy0 = raster_cell(y0, loy, fy, OCC_GRID), qy1 = raster_cell(y1, loy, fy, OCC_GRID) + 1;
        hit = integral[qy1 * OCC_STRIDE + qx1] - integral[qy0 * OCC_STRIDE + qx1] - integral[qy1 * OCC_STRIDE + qx0] +
                  integral[qy0 * OCC_STRIDE + qx0] > 0;
    }
    flag[i] = hit;
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_shard_plan_dev(const double *src_xy_dev, const int64_t *src_faces_dev, int64_t n_src_face, int src_m,
                      const double *tgt_xy_dev, const int64_t *tgt_faces_dev, int64_t n_tgt_face, int tgt_m, int world, int rank,
                      int mode, int64_t *local_faces_dev, int64_t *n_local_faces, int64_t *local_targets_dev,
                      int64_t *n_local_targets, int32_t *owner_dev) {
    XR_API_BEGIN
    const int64_t S = n_src_face, T = n_tgt_face;
    XR_REQUIRE(S >= 0 && T >= 0 && src_m >= 1 && tgt_m >= 1, XR_ERR_INVALID, "xr_shard_plan_dev: bad sizes");
    XR_REQUIRE(world >= 1 && world <= 4096 && rank >= 0 && rank < world, XR_ERR_INVALID, "xr_shard_plan_dev: rank %d of %d (at most 4096 ranks)", rank, world);
    XR_REQUIRE(mode >= 0 && mode <= 2, XR_ERR_INVALID, "xr_shard_plan_dev: mode %d (0 hash, 1 morton, 2 balanced)", mode);
    XR_REQUIRE(n_local_faces && n_local_targets, XR_ERR_INVALID, "xr_shard_plan_dev: NULL argument");
    XR_REQUIRE((S == 0 || (src_xy_dev && src_faces_dev && local_faces_dev)) && (T == 0 || (tgt_xy_dev && tgt_faces_dev && local_targets_dev)),
               XR_ERR_INVALID, "xr_shard_plan_dev: NULL array");
    XR_REQUIRE(S + T < ((int64_t)1 << 31), XR_ERR_LIMIT, "xr_shard_plan_dev: more than 2^31 faces");
    *n_local_faces = *n_local_targets = 0;
    if (S == 0) {
        dev_call_done();
        return XR_OK;
    }
    hipStream_t st = launch_stream();
    const int n_grid = (int)std::min(256.0, std::max(4.0, std::sqrt((double)S / 16.0)));
    DevBuf<double2> scen((size_t)S), tcen((size_t)std::max<int64_t>(T, 1));
    DevBuf<Bounds> bounds(3);
    DevBuf<int32_t> n_src((size_t)n_grid * n_grid), n_tgt((size_t)n_grid * n_grid), mcell((size_t)S), sflag((size_t)S + 1),
        spos((size_t)S + 1), tflag((size_t)T + 1), tpos((size_t)T + 1), diff((size_t)OCC_STRIDE * OCC_STRIDE),
        integral((size_t)OCC_STRIDE * OCC_STRIDE);
    DevBuf<unsigned long long> cell_work((size_t)SHARD_CELLS), part((size_t)SHARD_CELLS / S64_TILE + 1);
    const int64_t nb_all = div_up(S + T, 256);
    DevBuf<double4> partial((size_t)(2 * nb_all));
    XR_LAUNCH("shard_centroids", k_shard_centroids, dim3(nb_all), dim3(256), 0, src_xy_dev, src_faces_dev, S, src_m,
              tgt_xy_dev, tgt_faces_dev, T, tgt_m, scen.get(), tcen.get(), partial.get());
    XR_LAUNCH("shard_bounds", k_shard_fold_bounds, dim3(2), dim3(256), 0, partial.get(), nb_all, bounds.get());
    const bool hash = mode == 0, balanced = mode == 2;
    if (!hash) {
        if (balanced) {
            fill_i32(n_src.get(), 0, (int64_t)n_grid * n_grid);
            fill_i32(n_tgt.get(), 0, (int64_t)n_grid * n_grid);
            XR_LAUNCH("shard_count", k_shard_count, dim3(div_up(S + T, 256)), dim3(256), 0, scen.get(), S, tcen.get(), T, bounds.get(),
                      n_grid, n_src.get(), n_tgt.get());
        }
        XR_HIP(hipMemsetAsync(cell_work.get(), 0, sizeof(unsigned long long) * SHARD_CELLS, st));
        XR_LAUNCH("shard_work", k_shard_work, dim3(div_up(S, 256)), dim3(256), 0, scen.get(), S, bounds.get(), n_grid, n_src.get(),
                  n_tgt.get(), balanced, mcell.get(), cell_work.get());
        constexpr int NB = SHARD_CELLS / S64_TILE; // 1024
        XR_LAUNCH("shard_scan", k_scan64_reduce, dim3(NB), dim3(256), 0, cell_work.get(), part.get());
        XR_LAUNCH("shard_scan", k_scan64_partials, dim3(1), dim3(1024), 0, part.get(), NB, part.get() + NB);
        XR_LAUNCH("shard_scan", k_scan64_apply, dim3(NB), dim3(256), 0, cell_work.get(), part.get());
    }
    XR_LAUNCH("shard_owner", k_shard_owner, dim3(div_up(S, 256)), dim3(256), 0, S, mcell.get(), cell_work.get(),
              part.get() + SHARD_CELLS / S64_TILE, world, rank, hash, sflag.get(), owner_dev);
    exclusive_scan_i32(sflag.get(), spos.get(), S);
    XR_LAUNCH("shard_compact", k_shard_compact, dim3(div_up(S, 256)), dim3(256), 0, sflag.get(), spos.get(), S, local_faces_dev);
    if (T > 0) {
        XR_LAUNCH("shard_box_bounds", k_shard_box_bounds, dim3(nb_all), dim3(256), 0, src_xy_dev, src_faces_dev, S, src_m,
                  sflag.get(), tgt_xy_dev, tgt_faces_dev, T, tgt_m, partial.get());
        XR_LAUNCH("shard_bounds", k_shard_fold_bounds, dim3(1), dim3(256), 0, partial.get(), nb_all, bounds.get() + 2);
        fill_i32(diff.get(), 0, (int64_t)OCC_STRIDE * OCC_STRIDE);
        XR_LAUNCH("shard_occupy", k_shard_occupy, dim3(div_up(S, 256)), dim3(256), 0, src_xy_dev, src_faces_dev, S, src_m, sflag.get(),
                  bounds.get(), diff.get());
        XR_LAUNCH("shard_integral", k_shard_integral, dim3(1), dim3(1024), 0, diff.get(), integral.get());
        XR_LAUNCH("shard_targets", k_shard_targets, dim3(div_up(T, 256)), dim3(256), 0, tgt_xy_dev, tgt_faces_dev, T, tgt_m, bounds.get(),
                  integral.get(), tflag.get());
        exclusive_scan_i32(tflag.get(), tpos.get(), T);
        XR_LAUNCH("shard_compact", k_shard_compact, dim3(div_up(T, 256)), dim3(256), 0, tflag.get(), tpos.get(), T, local_targets_dev);
    }
    int32_t n_s = 0, n_t = 0;
    d2h(&n_s, spos.get() + S, sizeof(int32_t));
    if (T > 0) d2h(&n_t, tpos.get() + T, sizeof(int32_t));
    *n_local_faces = n_s;
    *n_local_targets = n_t;
    dev_call_done();
    XR_API_END
}

} // extern "C"
