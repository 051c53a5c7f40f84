// xr_scan.hip -- device exclusive prefix sum (int32) and fills.  wave64, 256-thread blocks.
// Three-phase reduce / scan-of-partials / scan-with-offset; every element is read twice and
// written once (HBM-bound, 12 B per element).
#include "xr_internal.h"

namespace xr {

static constexpr int SCAN_BLOCK = 256;
static constexpr int SCAN_ITEMS = 8; // per thread
static constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total via *total
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *lds /* >= 4 ints */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; w++) {
        int s = lds[w];
        if (w < wave) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

// a thread's SCAN_ITEMS consecutive elements as two 16-byte accesses (the tile base and every thread's offset are multiples of
// eight elements, the arrays come from the pool: 32-byte aligned); element by element only in the one tile that holds the end
__device__ __forceinline__ void load_items(const int32_t *__restrict__ in, int64_t base, int64_t n, int (&v)[SCAN_ITEMS]) {
    static_assert(SCAN_ITEMS == 8, "two int4 per thread");
    if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base), b = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + i < n ? in[base + i] : 0;
    }
}
__device__ __forceinline__ void store_scanned(int32_t *__restrict__ out, int64_t base, int64_t n, const int (&v)[SCAN_ITEMS], int ex) {
    int o[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        o[i] = ex;
        ex += v[i];
    }
    if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        *reinterpret_cast<int4 *>(out + base) = make_int4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<int4 *>(out + base + 4) = make_int4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (base + i < n) out[base + i] = o[i];
    }
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_reduce(const int32_t *__restrict__ in,
                                                            int32_t *__restrict__ block_sums, int64_t n) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t j = base + (int64_t)i * SCAN_BLOCK + threadIdx.x;
        if (j < n) s += in[j];
    }
    int tot;
    (void)block_excl_scan(s, &tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block scans up to any number of partials sequentially by tiles
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_partials(int32_t *__restrict__ sums, int64_t nb,
                                                              int32_t *__restrict__ total_out, int32_t *host_total,
                                                              const int32_t *extra_src, int32_t *extra_host) {
    __shared__ int lds[4];
    int carry = 0;
    for (int64_t base = 0; base < nb; base += SCAN_BLOCK) {
        int64_t j = base + threadIdx.x;
        int v = j < nb ? sums[j] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot, lds);
        if (j < nb) sums[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
    if (threadIdx.x == 0 && host_total) *host_total = carry;
    if (threadIdx.x == 0 && extra_host) *extra_host = *extra_src;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const int32_t *__restrict__ in,
                                                           const int32_t *__restrict__ block_off,
                                                           int32_t *__restrict__ out, int64_t n) {
    __shared__ int lds[4];
    // thread owns SCAN_ITEMS consecutive elements so the scan order is the array order
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    load_items(in, base, n, v);
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    int tot;
    const int ex = block_excl_scan(s, &tot, lds) + block_off[blockIdx.x];
    store_scanned(out, base, n, v, ex);
}

__global__ void k_fill_i32(int32_t *p, int32_t v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_fill_f64(double *p, double v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// two-kernel variant for up to SCAN_FUSED_TILES tiles: every block of the apply pass first scans the
// (few) tile sums itself in LDS instead of waiting for a separate single-block kernel
static constexpr int SCAN_FUSED_TILES = 8192;

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply_fused(const int32_t *__restrict__ in,
                                                                 const int32_t *__restrict__ tile_sums, int64_t nb,
                                                                 int32_t *__restrict__ out, int64_t n, int32_t *host_total,
                                                                 const int32_t *extra_src, int32_t *extra_host) {
    __shared__ int lds[4];
    __shared__ int sh_base;
    // offset of this tile = sum of the tile sums before it (strided partial sums + block reduce)
    int part = 0;
    for (int64_t i = threadIdx.x; i < blockIdx.x; i += SCAN_BLOCK) part += tile_sums[i];
    int tot;
    (void)block_excl_scan(part, &tot, lds);
    if (threadIdx.x == 0) sh_base = tot;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    load_items(in, base, n, v);
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    const int ex = block_excl_scan(s, &tot, lds) + sh_base;
    store_scanned(out, base, n, v, ex);
    if (blockIdx.x == nb - 1 && threadIdx.x == 0) {
        out[n] = sh_base + tot; // grand total
        if (host_total) *host_total = sh_base + tot;
        if (extra_host) *extra_host = *extra_src;
    }
}

void exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *host_total, const int32_t *extra_src,
                        int32_t *extra_host) {
    if (n <= 0) {
        fill_i32(out, 0, 1);
        if (host_total) {
            stream_sync();
            *host_total = 0;
            if (extra_host) d2h(extra_host, extra_src, sizeof(int32_t));
        }
        return;
    }
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    DevBuf<int32_t> sums((size_t)nb);
    XR_LAUNCH("scan_reduce", k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, in, sums.get(), n);
    if (nb <= SCAN_FUSED_TILES) {
        XR_LAUNCH("scan_apply", k_scan_apply_fused, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, in, sums.get(), nb, out, n,
                  host_total, extra_src, extra_host);
        return;
    }
    XR_LAUNCH("scan_partials", k_scan_partials, dim3(1), dim3(SCAN_BLOCK), 0, sums.get(), nb, out + n, host_total, extra_src,
              extra_host);
    XR_LAUNCH("scan_apply", k_scan_apply, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, in, sums.get(), out, n);
}

void fill_i32(int32_t *p, int32_t v, int64_t n) {
    if (n <= 0) return;
    XR_LAUNCH("fill_i32", k_fill_i32, dim3(div_up(n, 256)), dim3(256), 0, p, v, n);
}
void fill_f64(double *p, double v, int64_t n) {
    if (n <= 0) return;
    XR_LAUNCH("fill_f64", k_fill_f64, dim3(div_up(n, 256)), dim3(256), 0, p, v, n);
}

} // namespace xr
