// xr_apply.hip -- the sparse weight x data apply and the MatrixCSR handle.
//
// Replaces make_regrid(func)._regrid (xugrid/regrid/regridder.py:41-67) with one kernel
// specialisation per reducer of xugrid/regrid/reduce.py, CentroidLocatorRegridder._regrid
// (regridder.py:400-409) and MatrixCSR.from_triplet / MatrixCOO.to_csr
// (xugrid/core/sparse.py:61-78,119-127).
//
// Reducer semantics are those of reduce.py line by line; entries of a
// row are consumed in CSR order by one thread per (row, k-tile), so that results are
// bit-identical to the reference loop on the same CSR (except exp/log in geometric_mean).
//
// HBM layout: source (K, S) row-major, out (K, T) row-major (regridder.py:163, :44);
// CSR = indptr i32[T+1], indices i32[nnz], data f64[nnz].
#include <cstdlib>
#include <cstring>
#include <vector>

#include <memory>

#include "xr_objects.h"

namespace xr {

static constexpr int AP_BLOCK = 256;
static constexpr int KT = 16; // source variables reduced per pass over the CSR (K >= 2)

template <typename SRC> __device__ __forceinline__ double ld_src(const SRC *p, int64_t i) { return (double)p[i]; }

// ---------------------------------------------------------------------------------------------
// streaming reducers: constant-size state per (row, k), one pass over the row
// ---------------------------------------------------------------------------------------------
// State = up to three doubles (a, b, c) so that partial states can be shuffled between lanes and
// merged (long rows are reduced by a whole block, see k_apply_long).
template <int METHOD> struct Red;

template <> struct Red<XR_MEAN> { // reduce.py:16-27   a = sum w v, b = sum w
    double a = 0.0, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool ok = v == v; // select instead of branch: lanes stay converged
        const double na = a + w * v, nb = b + w;
        a = ok ? na : a;
        b = ok ? nb : b;
    }
    __device__ void merge(const Red &o) { a += o.a; b += o.b; }
    __device__ double fin() const { return b == 0 ? NAN : a / b; }
};
template <> struct Red<XR_HARMONIC_MEAN> { // reduce.py:30-42   a = sum w / v, b = sum w
    double a = 0.0, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        if (v != v || v == 0) return;
        if (w > 0) {
            b += w;
            a += w / v;
        }
    }
    __device__ void merge(const Red &o) { a += o.a; b += o.b; }
    __device__ double fin() const { return (a == 0 || b == 0) ? NAN : b / a; }
};
template <> struct Red<XR_GEOMETRIC_MEAN> { // reduce.py:45-72; normsum = sum of ALL row weights
    double a = 0.0, b = 0.0, c = 0.0;        // a = sum wn ln v, b = sum wn, c = 1 if a negative value was seen
    __device__ void add(double v, double w, double normsum) {
        if (c != 0.0) return;
        const double wn = w / normsum;
        if (v > 0 && wn > 0) {
            a += wn * log(fabs(v));
            b += wn;
        } else if (v < 0) {
            c = 1.0;
        }
    }
    __device__ void merge(const Red &o) { a += o.a; b += o.b; if (o.c != 0.0) c = 1.0; }
    __device__ double fin() const { return (c != 0.0 || b == 0) ? NAN : exp((1.0 / b) * a); }
};
template <> struct Red<XR_SUM> { // reduce.py:75-87   a = sum v, b = sum w
    double a = 0.0, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool ok = v == v;
        const double na = a + v, nb = b + w;
        a = ok ? na : a;
        b = ok ? nb : b;
    }
    __device__ void merge(const Red &o) { a += o.a; b += o.b; }
    __device__ double fin() const { return b == 0 ? NAN : a; }
};
template <> struct Red<XR_MINIMUM> { // reduce.py:90-106   a = min v, b = max w
    double a = INFINITY, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool ok = v == v;
        a = (ok && v < a) ? v : a;
        b = (ok && w > b) ? w : b;
    }
    __device__ void merge(const Red &o) { if (o.a < a) a = o.a; if (o.b > b) b = o.b; }
    __device__ double fin() const { return b == 0.0 ? NAN : a; }
};
template <> struct Red<XR_MAXIMUM> { // reduce.py:109-123   a = max v, b = max w
    double a = -INFINITY, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool ok = v == v;
        a = (ok && v > a) ? v : a;
        b = (ok && w > b) ? w : b;
    }
    __device__ void merge(const Red &o) { if (o.a > a) a = o.a; if (o.b > b) b = o.b; }
    __device__ double fin() const { return b == 0.0 ? NAN : a; }
};
template <> struct Red<XR_FIRST_ORDER_CONSERVATIVE> { // reduce.py:206-222   a = sum v w, b = sum w
    double a = 0.0, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool ok = v == v;
        const double na = a + v * w, nb = b + w;
        a = ok ? na : a;
        b = ok ? nb : b;
    }
    __device__ void merge(const Red &o) { a += o.a; b += o.b; }
    __device__ double fin() const { return b == 0 ? NAN : a; }
};
template <> struct Red<XR_MAX_OVERLAP> { // reduce.py:225-238   a = value of the largest weight b
    double a = -INFINITY, b = 0.0, c = 0.0;
    __device__ void add(double v, double w, double) {
        const bool take = (v == v) && ((w > b) || (w == b && v > a));
        b = take ? w : b;
        a = take ? v : a;
    }
    __device__ void merge(const Red &o) {
        if ((o.b > b) || (o.b == b && o.a > a)) {
            b = o.b;
            a = o.a;
        }
    }
    __device__ double fin() const { return b == 0.0 ? NAN : a; }
};

template <> struct Red<XR_SELECT> { // regridder.py:400-409 on CSR rows: the last entry wins, NaN included
    double a = NAN, b = 0.0, c = 0.0;
    __device__ void add(double v, double, double) {
        a = v;
        b = 1.0;
    }
    __device__ void merge(const Red &o) {
        if (o.b != 0.0) {
            a = o.a;
            b = 1.0;
        }
    }
    __device__ double fin() const { return a; }
};

// Rows longer than APPLY_LONG entries are not reduced by one thread: a coarse target cell over a fine source
// has tens to thousands of entries, and a thread walking such a row alone reads its CSR entries uncoalesced and
// waits one memory latency per entry.  They are reduced by one WAVE each (lanes stride the row: coalesced
// entries, neighbouring columns gathered together) or, beyond APPLY_WAVE entries, by a whole block: strided
// partial states merged in a fixed butterfly order -> deterministic, but the summation order differs from
// the reference's sequential loop (agreement ~1e-15 relative instead of bit-exact).
static constexpr int APPLY_LONG = XR_APPLY_LONG_ROW;
static constexpr int APPLY_WAVE = XR_APPLY_WAVE_ROW;

template <int METHOD> __device__ __forceinline__ void wave_merge(Red<METHOD> &r) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        Red<METHOD> o;
        o.a = __shfl_xor(r.a, d, 64);
        o.b = __shfl_xor(r.b, d, 64);
        o.c = __shfl_xor(r.c, d, 64);
        if ((lane & d) == 0) { // lane-independent order: every lane ends with the same value
            r.merge(o);
        } else {
            o.merge(r);
            r = o;
        }
    }
}

template <int METHOD> __device__ __forceinline__ void block_merge(Red<METHOD> &r, double (*lds)[3]) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        Red<METHOD> o;
        o.a = __shfl_xor(r.a, d, 64);
        o.b = __shfl_xor(r.b, d, 64);
        o.c = __shfl_xor(r.c, d, 64);
        // merge in a lane-independent order so that every lane holds the same value
        if ((threadIdx.x & d) == 0) {
            r.merge(o);
        } else {
            o.merge(r);
            r = o;
        }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        lds[wave][0] = r.a;
        lds[wave][1] = r.b;
        lds[wave][2] = r.c;
    }
    __syncthreads();
    Red<METHOD> t;
    t.a = lds[0][0]; t.b = lds[0][1]; t.c = lds[0][2];
#pragma unroll
    for (int w = 1; w < AP_BLOCK / 64; w++) {
        Red<METHOD> o;
        o.a = lds[w][0]; o.b = lds[w][1]; o.c = lds[w][2];
        t.merge(o);
    }
    r = t;
}

// one listed row reduced by the whole block (all threads call; rows beyond APPLY_WAVE entries)
template <int METHOD, typename SRC>
__device__ __forceinline__ void apply_row_block(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                const double *__restrict__ data, const int32_t *__restrict__ row_order,
                                                int t, int64_t T, int64_t S, const SRC *__restrict__ source, int64_t K,
                                                double *__restrict__ out, double (*lds)[3]) {
    const int s = indptr[t], e = indptr[t + 1];
    const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
    double normsum = 0.0;
    if (METHOD == XR_GEOMETRIC_MEAN) {
        Red<XR_SUM> ws; // b accumulates the weights
        for (int j = s + threadIdx.x; j < e; j += AP_BLOCK) ws.b += data[j];
        block_merge<XR_SUM>(ws, lds);
        normsum = ws.b;
    }
    for (int64_t k = blockIdx.y; k < K; k += gridDim.y) {
        const SRC *src = source + k * S;
        Red<METHOD> r;
        int j = s + threadIdx.x;
        // four entries in flight per thread; the additions keep their sequential order
        for (; j + 3 * AP_BLOCK < e; j += 4 * AP_BLOCK) {
            const int c0 = indices[j], c1 = indices[j + AP_BLOCK], c2 = indices[j + 2 * AP_BLOCK],
                      c3 = indices[j + 3 * AP_BLOCK];
            const double w0 = data[j], w1 = data[j + AP_BLOCK], w2 = data[j + 2 * AP_BLOCK],
                         w3 = data[j + 3 * AP_BLOCK];
            const double v0 = ld_src(src, c0), v1 = ld_src(src, c1), v2 = ld_src(src, c2), v3 = ld_src(src, c3);
            r.add(v0, w0, normsum);
            r.add(v1, w1, normsum);
            r.add(v2, w2, normsum);
            r.add(v3, w3, normsum);
        }
        for (; j < e; j += AP_BLOCK) r.add(ld_src(src, indices[j]), data[j], normsum);
        block_merge<METHOD>(r, lds);
        if (threadIdx.x == 0) {
            double v = r.fin();
            if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) v = NAN;
            out[k * T + t_out] = v;
        }
    }
}

template <int METHOD, typename SRC>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_long(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
             const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
             const int32_t *__restrict__ n_long, int64_t T, int64_t S, const SRC *__restrict__ source, int64_t K,
             double *__restrict__ out) {
    __shared__ double lds[AP_BLOCK / 64][3];
    const int nl = *n_long;
    for (int li = blockIdx.x; li < nl; li += gridDim.x)
        apply_row_block<METHOD, SRC>(indptr, indices, data, row_order, long_rows[li], T, S, source, K, out, lds);
}

// Listed rows with APPLY_LONG < entries <= APPLY_WAVE, KTILE source variables per pass, reduced by a GROUP of
// G lanes each: G = 16 (four rows per wave) up to APPLY_GROUP16 entries, G = 64 (one row per wave) beyond.
// The lanes of a group stride the row (entry j goes to lane j % G), so the CSR entries are read coalesced and
// -- rows are sorted by column -- neighbouring lanes gather neighbouring source values; the group's partial
// states are merged by a butterfly of log2(G) shuffles.  A lane's first four entries stay in registers across
// the variable tiles.
static constexpr int APPLY_GROUP16 = 512;

template <int METHOD, int G> __device__ __forceinline__ void group_merge(Red<METHOD> &r) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) {
        Red<METHOD> o;
        o.a = __shfl_xor(r.a, d, 64);
        o.b = __shfl_xor(r.b, d, 64);
        o.c = __shfl_xor(r.c, d, 64);
        if ((lane & d) == 0) { // lane-independent order: every lane of the group ends with the same value
            r.merge(o);
        } else {
            o.merge(r);
            r = o;
        }
    }
}

template <int METHOD, typename SRC, int KTILE, int G>
__device__ __forceinline__ void apply_row_group(const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                int s, int e, int64_t t_out, int64_t T, int64_t S,
                                                const SRC *__restrict__ source, int64_t K, double *__restrict__ out) {
    const int gl = threadIdx.x & (G - 1); // lane within the group
    int col4[4];
    double w4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = s + gl + G * u;
        col4[u] = j < e ? indices[j] : -1;
        w4[u] = j < e ? data[j] : 0.0;
    }
    double normsum = 0.0;
    if (METHOD == XR_GEOMETRIC_MEAN) {
        Red<XR_SUM> ws; // b accumulates the weights
        for (int j = s + gl; j < e; j += G) ws.b += data[j];
        group_merge<XR_SUM, G>(ws);
        normsum = ws.b;
    }
    for (int64_t k0 = (int64_t)blockIdx.y * KTILE; k0 < K; k0 += (int64_t)gridDim.y * KTILE) {
        const int kn = (int)((K - k0) < KTILE ? (K - k0) : KTILE);
        const SRC *src = source + k0 * S;
        Red<METHOD> r[KTILE];
        double v[4][KTILE];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++)
                v[u][kk] = (col4[u] >= 0 && kk < kn) ? ld_src(src, (int64_t)kk * S + col4[u]) : 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (col4[u] >= 0) {
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++)
                    if (kk < kn) r[kk].add(v[u][kk], w4[u], normsum);
            }
        }
        for (int j = s + gl + 4 * G; j < e; j += G) {
            const int64_t col = indices[j];
            const double w = data[j];
            double vv[KTILE];
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++) vv[kk] = kk < kn ? ld_src(src, (int64_t)kk * S + col) : 0.0;
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++)
                if (kk < kn) r[kk].add(vv[kk], w, normsum);
        }
#pragma unroll
        for (int kk = 0; kk < KTILE; kk++) group_merge<METHOD, G>(r[kk]);
        if (gl == 0) {
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++) {
                if (kk < kn) {
                    double res = r[kk].fin();
                    if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) res = NAN;
                    out[(k0 + kk) * T + t_out] = res;
                }
            }
        }
    }
}

// the listed rows of APPLY_LONG + 1 ... APPLY_WAVE entries, dealt to `n_waves` waves (this one is `wave`); rows beyond
// APPLY_WAVE entries are queued in huge_rows ([0] = count) when given, else left to the caller
template <int METHOD, typename SRC, int KTILE>
__device__ __forceinline__ void apply_wave_rows(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                const double *__restrict__ data, const int32_t *__restrict__ row_order,
                                                const int32_t *__restrict__ long_rows, int nl, int wave, int n_waves,
                                                int64_t T, int64_t S, const SRC *__restrict__ source, int64_t K,
                                                double *__restrict__ out, int32_t *__restrict__ huge_rows) {
    const int lane = threadIdx.x & 63;
    // pass 1: four listed rows per wave, 16 lanes each
    for (int li0 = wave * 4; li0 < nl; li0 += n_waves * 4) {
        const int li = li0 + (lane >> 4);
        int s = 0, e = 0;
        int64_t t_out = 0;
        if (li < nl) {
            const int t = long_rows[li];
            s = indptr[t];
            e = indptr[t + 1];
            t_out = row_order ? (int64_t)row_order[t] : t;
            if (e - s > APPLY_GROUP16) e = s; // other passes
        }
        // (the shuffles inside are wave-wide: every lane takes part, idle groups carry empty rows)
        if (__any(e > s)) {
            if (e > s) apply_row_group<METHOD, SRC, KTILE, 16>(indices, data, s, e, t_out, T, S, source, K, out);
        }
    }
    // pass 2: one listed row per wave; rows beyond APPLY_WAVE entries are queued for the block kernel
    for (int li = wave; li < nl; li += n_waves) {
        const int t = long_rows[li];
        const int s = indptr[t], e = indptr[t + 1];
        if (e - s <= APPLY_GROUP16) continue;
        if (e - s > APPLY_WAVE) { // [0] = count
            if (huge_rows && lane == 0 && blockIdx.y == 0) huge_rows[1 + atomicAdd(huge_rows, 1)] = t;
            continue;
        }
        const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
        apply_row_group<METHOD, SRC, KTILE, 64>(indices, data, s, e, t_out, T, S, source, K, out);
    }
}

template <int METHOD, typename SRC, int KTILE>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_wave(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
             const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
             const int32_t *__restrict__ n_long, int64_t T, int64_t S, const SRC *__restrict__ source, int64_t K,
             double *__restrict__ out, int32_t *__restrict__ huge_rows) {
    const int wave = (blockIdx.x * AP_BLOCK + threadIdx.x) >> 6, n_waves = gridDim.x * (AP_BLOCK / 64);
    apply_wave_rows<METHOD, SRC, KTILE>(indptr, indices, data, row_order, long_rows, *n_long, wave, n_waves, T, S, source, K,
                                        out, huge_rows);
}

// K = 1, every row in ONE launch.  Blocks [0, n_long_blocks) take the listed long rows (lane groups / waves, then the rows
// beyond APPLY_WAVE entries block by block) -- they are dispatched first and run beside the short rows, no second and third
// launch behind the kernel, no fork / join.  The other blocks take AP_BLOCK stored rows each, every WAVE on its own: a wave
// stages the CSR segment of its 64 consecutive rows -- contiguous -- in a private LDS window in chunks of W1_CAP entries
// (column and weight loaded coalesced, the source value gathered by the same lane and parked next to the weight), then every
// lane reduces ITS row from the window, sequentially in CSR order (bit-identical to the reference loop,
// regridder.py:52-62).  No block barrier anywhere and 6 KB of LDS per wave: 24 waves per CU keep three dependent
// round trips (row pointers -> entries -> source values) in flight, where the block-wide version of rounds 1-2 (32 KB
// and four barriers per block) held 20 (MI355X, 1M x 1M benchmark: 42.6 -> see DESIGN section 5).
static constexpr int W1_CAP = 384; // entries per wave window: 64 rows x 4 entries (the mean of a triangle pair) + slack for the tail
                                   // (measured on the 1M x 1M matrix: 256 / 320 / 384 / 448 entries -> kernel 31 / 30 / 27 / 27 us)
template <int METHOD, typename SRC>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_rows1(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
              const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
              const int32_t *__restrict__ n_long, int n_long_blocks, bool any_huge, int64_t T, int64_t S,
              const SRC *__restrict__ source, double *__restrict__ out, const int32_t *__restrict__ gate) {
    __shared__ double2 sh_win[AP_BLOCK / 64][W1_CAP]; // (.x = weight, .y = source value); 24 KB: 6 blocks per CU
    if (gate && *gate == 0) return; // (enqueued behind a weight build whose attempt failed: the host redoes both)
    double(*sh_merge)[3] = reinterpret_cast<double(*)[3]>(&sh_win[0][0]); // (long-row blocks stage nothing)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    if ((int)blockIdx.x < n_long_blocks) {
        const int nl = *n_long;
        apply_wave_rows<METHOD, SRC, 1>(indptr, indices, data, row_order, long_rows, nl, (int)blockIdx.x * (AP_BLOCK / 64) + wib,
                                        n_long_blocks * (AP_BLOCK / 64), T, S, source, 1, out, nullptr);
        if (any_huge) {
            // rows beyond the wave kernel's reach, a block each.  The block's share of the list is looked at by all its
            // threads AT ONCE (entry li = block + n_long_blocks * thread: two dependent loads per thread, in flight
            // together) and the few hits are parked in LDS -- walking the share entry by entry was a chain of ~9 x 2
            // dependent round trips that made these blocks the last to finish (the kernel's 28 us on the benchmark).
            int *sh_list = reinterpret_cast<int *>(&sh_win[1][0]), *sh_cnt = reinterpret_cast<int *>(&sh_win[2][0]);
            for (int base = 0; base < nl; base += n_long_blocks * AP_BLOCK) {
                if (threadIdx.x == 0) *sh_cnt = 0;
                __syncthreads();
                const int li = base + (int)blockIdx.x + n_long_blocks * (int)threadIdx.x;
                if (li < nl) {
                    const int t = long_rows[li];
                    if (indptr[t + 1] - indptr[t] > APPLY_WAVE) sh_list[atomicAdd(sh_cnt, 1)] = t; // (at most one per thread)
                }
                __syncthreads();
                const int n_hit = *sh_cnt;
                for (int i = 0; i < n_hit; i++) // (uniform: every thread of the block works on the row)
                    apply_row_block<METHOD, SRC>(indptr, indices, data, row_order, sh_list[i], T, S, source, 1, out, sh_merge);
                __syncthreads();
            }
        }
        return;
    }
    // (last row blocks first: weights built by xr_overlap keep the rows of the big target faces at the end)
    const int64_t row0 = ((int64_t)(gridDim.x - 1 - blockIdx.x) * (AP_BLOCK / 64) + wib) * 64;
    if (row0 >= T) return;
    const int64_t t = row0 + lane;
    int s = 0, e = 0;
    if (t < T) {
        s = indptr[t];
        e = indptr[t + 1];
    }
    const bool skip = n_long_blocks > 0 && (e - s > APPLY_LONG); // reduced by the long-row blocks
    const int seg0 = __shfl(s, 0, 64);
    int seg1 = e; // (lanes beyond T hold 0: the maximum is the end of the last valid row)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) seg1 = max(seg1, __shfl_xor(seg1, d, 64));
    if (skip) e = s;
    const bool jumpy = __any(skip);
    double2 *win = sh_win[wib];
    // first entry any lane still needs at or behind c0 (a window must not stream the entries of skipped long rows)
    auto next_chunk = [&](int c0) -> int {
        if (!jumpy) return c0;
        int mine = (e > s && e > c0) ? (s > c0 ? s : c0) : INT_MAX;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine = min(mine, __shfl_xor(mine, d, 64));
        return mine;
    };
    double normsum = 0.0;
    if (METHOD == XR_GEOMETRIC_MEAN) {
        for (int c0 = seg0; c0 < seg1; c0 += W1_CAP) {
            c0 = next_chunk(c0);
            if (c0 >= seg1) break;
            for (int j = c0 + lane; j < c0 + W1_CAP && j < seg1; j += 64) win[j - c0].x = data[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int a = s > c0 ? s : c0, b = e < c0 + W1_CAP ? e : c0 + W1_CAP;
            for (int j = a; j < b; j++) normsum += win[j - c0].x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    Red<METHOD> red;
    for (int c0 = seg0; c0 < seg1; c0 += W1_CAP) {
        c0 = next_chunk(c0);
        if (c0 >= seg1) break;
        {
            constexpr int PER = W1_CAP / 64;
            int col[PER];
            double w[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int j = c0 + u * 64 + lane;
                col[u] = j < seg1 ? indices[j] : -1;
                w[u] = j < seg1 ? data[j] : 0.0;
            }
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = col[u] >= 0 ? ld_src(source, (int64_t)col[u]) : 0.0;
#pragma unroll
            for (int u = 0; u < PER; u++)
                if (col[u] >= 0) win[u * 64 + lane] = make_double2(w[u], v[u]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int a = s > c0 ? s : c0, b = e < c0 + W1_CAP ? e : c0 + W1_CAP;
        for (int j = a; j < b; j++) {
            const double2 wv = win[j - c0];
            red.add(wv.y, wv.x, normsum);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the window is overwritten by the next chunk)
        __builtin_amdgcn_wave_barrier();
    }
    if (t < T && !skip) {
        const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
        double r = NAN; // regridder.py:44,62: rows without entries stay NaN
        if (e > s) {
            r = red.fin();
            if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) r = NAN;
        }
        out[t_out] = r;
    }
}

// Many source variables (K >= 2): one thread per row keeps KTILE reducer states in registers and
// walks its row once per k-tile; no LDS, so occupancy is limited by registers only and every
// thread has KTILE independent gathers in flight per entry.  Rows are visited in stored (spatial)
// order, so the gathers of neighbouring threads hit the same cache lines.
template <int METHOD, typename SRC, int KTILE>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_direct(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
               const double *__restrict__ data, const int32_t *__restrict__ row_order, bool skip_long, int64_t T,
               int64_t S, const SRC *__restrict__ source, int64_t K, double *__restrict__ out,
               const int32_t *__restrict__ block_list) {
    // block_list != nullptr: only these 256-row blocks (the ones the apply plan could not take)
    const int64_t t = (int64_t)(block_list ? block_list[blockIdx.x] : blockIdx.x) * AP_BLOCK + threadIdx.x;
    if (t >= T) return;
    const int64_t k0 = (int64_t)blockIdx.y * KTILE;
    const int kn = (int)((K - k0) < KTILE ? (K - k0) : KTILE);
    const int s = indptr[t], e = indptr[t + 1];
    if (skip_long && e - s > APPLY_LONG) return; // reduced by k_apply_long
    const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
    const SRC *src = source + k0 * S;
    double normsum = 0.0;
    if (METHOD == XR_GEOMETRIC_MEAN)
        for (int j = s; j < e; j++) normsum += data[j];
    Red<METHOD> red[KTILE];
    int j = s;
    if (KTILE <= 4) {
        // four entries' gathers in flight before their (sequential, CSR-order) additions: a thread that walks a
        // long row alone is otherwise one HBM latency per entry
        for (; j + 3 < e; j += 4) {
            int64_t col[4];
            double w[4], v[4][KTILE];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                col[u] = indices[j + u];
                w[u] = data[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++) v[u][kk] = kk < kn ? ld_src(src, (int64_t)kk * S + col[u]) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++)
                    if (kk < kn) red[kk].add(v[u][kk], w[u], normsum);
        }
    }
    for (; j < e; j++) {
        const int64_t col = indices[j];
        const double w = data[j];
        double v[KTILE];
#pragma unroll
        for (int kk = 0; kk < KTILE; kk++) v[kk] = kk < kn ? ld_src(src, (int64_t)kk * S + col) : 0.0;
#pragma unroll
        for (int kk = 0; kk < KTILE; kk++)
            if (kk < kn) red[kk].add(v[kk], w, normsum);
    }
#pragma unroll
    for (int kk = 0; kk < KTILE; kk++) {
        if (kk < kn) {
            double r = NAN;
            if (e > s) {
                r = red[kk].fin();
                if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) r = NAN;
            }
            out[(k0 + kk) * T + t_out] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// apply plan: blocked CSR with per-block distinct-column lists.
// A block of 256 spatially adjacent rows references ~5x fewer DISTINCT source faces than it has
// entries.  With many source variables the gathers dominate, so each distinct source value is
// loaded ONCE per (block, variable) -- lanes run over the ascending column list, neighbours share
// cache lines -- into LDS, and the rows are reduced from LDS through 16-bit local indices.
// ---------------------------------------------------------------------------------------------
static constexpr int PLAN_LMAX = 4096; // entries of a block the builder can sort in LDS
static constexpr int PLAN_UMAX = 512;  // distinct columns per block kept in the plan
static constexpr int PLAN_KT = 8;      // source variables per pipeline stage (LDS: PLAN_KT * PLAN_UMAX doubles)

__global__ void __launch_bounds__(AP_BLOCK)
k_plan_build(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t T,
             int32_t *__restrict__ ucol, int32_t *__restrict__ nuniq, uint16_t *__restrict__ loc,
             int32_t *__restrict__ max_entries, int32_t *__restrict__ unplanned, int32_t *__restrict__ n_unplanned,
             int entry_cap) {
    __shared__ int32_t keys[PLAN_LMAX];
    __shared__ int32_t uniq[PLAN_UMAX];
    __shared__ int32_t sh_wave[4];
    __shared__ int32_t sh_lines;
    const int64_t row0 = (int64_t)blockIdx.x * AP_BLOCK;
    const int64_t row_end = row0 + AP_BLOCK < T ? row0 + AP_BLOCK : T;
    const int seg0 = indptr[row0], seg1 = indptr[row_end];
    const int n = seg1 - seg0;
    if (n > entry_cap) {
        if (threadIdx.x == 0) {
            nuniq[blockIdx.x] = -1;
            unplanned[atomicAdd(n_unplanned, 1)] = (int32_t)blockIdx.x;
        }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += AP_BLOCK) keys[i] = i < n ? indices[seg0 + i] : 0x7fffffff;
    __syncthreads();
    // bitonic sort, ascending
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += AP_BLOCK) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // unique: each thread owns a contiguous slice of the sorted keys
    const int per = (n + AP_BLOCK - 1) / AP_BLOCK;
    const int a0 = threadIdx.x * per, a1 = a0 + per < n ? a0 + per : n;
    int cnt = 0;
    for (int i = a0; i < a1; i++) cnt += (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) sh_wave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) woff += sh_wave[w];
        total += sh_wave[w];
    }
    if (total > PLAN_UMAX) {
        if (threadIdx.x == 0) {
            nuniq[blockIdx.x] = -1;
            unplanned[atomicAdd(n_unplanned, 1)] = (int32_t)blockIdx.x;
        }
        return;
    }
    int pos = woff + incl - cnt;
    for (int i = a0; i < a1; i++) {
        if (i == 0 || keys[i] != keys[i - 1]) uniq[pos++] = keys[i];
    }
    __syncthreads();
    for (int u = threadIdx.x; u < total; u += AP_BLOCK) ucol[(int64_t)blockIdx.x * PLAN_UMAX + u] = uniq[u];
    // how well the block uses the source lines it touches (16 values of 8 bytes per 128-byte line): the sums over all blocks
    // decide whether a merged plan pays (ensure_plan)
    if (threadIdx.x == 0) sh_lines = 0;
    __syncthreads();
    {
        int mine = 0;
        for (int u = threadIdx.x; u < total; u += AP_BLOCK) mine += (u == 0 || (uniq[u] >> 4) != (uniq[u - 1] >> 4)) ? 1 : 0;
        if (mine) atomicAdd(&sh_lines, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(max_entries + 2, total);
        atomicAdd(max_entries + 3, sh_lines);
    }
    if (threadIdx.x == 0) atomicMax(max_entries, n); // the apply kernel sizes its LDS stage for the largest planned block
    if (threadIdx.x == 0) nuniq[blockIdx.x] = total;
    // local index of every entry: binary search in the distinct list
    for (int i = threadIdx.x; i < n; i += AP_BLOCK) {
        const int c = indices[seg0 + i];
        int lo = 0, hi = total - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        loc[seg0 + i] = (uint16_t)lo;
    }
}

// MERGED plan (round 5): ONE distinct-column list per GROUP of PLAN_GROUP neighbouring row blocks.  On a qhull-numbered mesh a
// block of 256 rows uses ~4 of the 16 values of every source line it touches, a group of 512 rows 4.5, of 1024 rows 5 (66 / 52
// lines per 256 rows instead of 85: profiles/r05_experiments/analysis_column_locality.*): the group's workgroup gathers every
// line once for all its row blocks.  (Blocks in lockstep with a list EACH -- XR_PLAN_SUBS -- do not get there: the lines in
// flight, 4 x 85 x 8 variables x 128 bytes, are ten times the L1, the siblings' requests miss again.)  Groups of two and of four
// measured the same on the benchmark matrix (K = 256: 1.64 -> 1.53 / 1.56 ms: what the larger group saves in line fills its
// 16-wave barriers cost); two blocks need 58 KB of LDS (two workgroups per CU) and cost a lattice-numbered matrix 2 %, four 7 %.
static constexpr int PLAN_GROUP = 2;
static constexpr int PLAN_GUMAX = 1024; // distinct columns per group kept in the plan
__global__ void __launch_bounds__(AP_BLOCK * PLAN_GROUP)
k_plan_build_group(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t T,
                   int32_t *__restrict__ ucol, int32_t *__restrict__ nuniq, uint16_t *__restrict__ loc,
                   int32_t *__restrict__ max_entries, int32_t *__restrict__ unplanned, int32_t *__restrict__ n_unplanned,
                   int entry_cap) {
    constexpr int NT = AP_BLOCK * PLAN_GROUP, KMAX = PLAN_LMAX * 2; // (8192 keys: the group's blocks of up to 2048 entries each)
    __shared__ int32_t keys[KMAX];
    __shared__ int32_t uniq[PLAN_GUMAX];
    __shared__ int32_t sh_wave[NT / 64];
    __shared__ int32_t sh_bad;
    const int64_t n_blocks = (T + AP_BLOCK - 1) / AP_BLOCK;
    const int64_t b0 = (int64_t)blockIdx.x * PLAN_GROUP;
    const int64_t row0 = b0 * AP_BLOCK;
    const int64_t row_end = row0 + NT < T ? row0 + NT : T;
    const int seg0 = indptr[row0], seg1 = indptr[row_end];
    const int n = seg1 - seg0;
    if (threadIdx.x == 0) sh_bad = 0;
    __syncthreads();
    // every block of the group has to fit the apply's per-block stage
    if (threadIdx.x < PLAN_GROUP) {
        const int64_t r0 = row0 + (int64_t)threadIdx.x * AP_BLOCK;
        if (r0 < T) {
            const int64_t r1 = r0 + AP_BLOCK < T ? r0 + AP_BLOCK : T;
            const int nb_e = indptr[r1] - indptr[r0];
            if (nb_e > entry_cap) sh_bad = 1;
        }
    }
    __syncthreads();
    auto give_up = [&]() {
        if (threadIdx.x == 0) {
            nuniq[blockIdx.x] = -1;
            for (int g = 0; g < PLAN_GROUP; g++)
                if (b0 + g < n_blocks) unplanned[atomicAdd(n_unplanned, 1)] = (int32_t)(b0 + g);
        }
    };
    if (sh_bad || n > KMAX) {
        give_up();
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += NT) keys[i] = i < n ? indices[seg0 + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) { // bitonic sort, ascending
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    const int per = (n + NT - 1) / NT;
    const int a0 = threadIdx.x * per < n ? threadIdx.x * per : n, a1 = a0 + per < n ? a0 + per : n;
    int cnt = 0;
    for (int i = a0; i < a1; i++) cnt += (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    __syncthreads();
    if (lane == 63) sh_wave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < NT / 64; w++) {
        if (w < wave) woff += sh_wave[w];
        total += sh_wave[w];
    }
    if (total > PLAN_GUMAX) {
        give_up();
        return;
    }
    int pos = woff + incl - cnt;
    for (int i = a0; i < a1; i++)
        if (i == 0 || keys[i] != keys[i - 1]) uniq[pos++] = keys[i];
    __syncthreads();
    for (int u = threadIdx.x; u < total; u += NT) ucol[(int64_t)blockIdx.x * PLAN_GUMAX + u] = uniq[u];
    if (threadIdx.x < PLAN_GROUP) { // the apply kernel sizes its per-block stage for the largest planned block
        const int64_t r0 = row0 + (int64_t)threadIdx.x * AP_BLOCK;
        if (r0 < T) {
            const int64_t r1 = r0 + AP_BLOCK < T ? r0 + AP_BLOCK : T;
            atomicMax(max_entries, indptr[r1] - indptr[r0]);
        }
    }
    if (threadIdx.x == 0) nuniq[blockIdx.x] = total;
    for (int i = threadIdx.x; i < n; i += NT) { // local index of every entry: binary search in the group's list
        const int c = indices[seg0 + i];
        int lo = 0, hi = total - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        loc[seg0 + i] = (uint16_t)lo;
    }
}

// One block = 256 stored rows, ALL variables: the block's CSR entries (weight + 16-bit local column)
// are staged in LDS once, then the block walks the variables in tiles of KTILE.  Software pipeline
// (global -> registers -> LDS): the distinct source values of tile i+1 are requested into registers
// before tile i is reduced, and written to LDS after it -- HBM latency overlaps the reduction, so one
// resident block per CU is enough to keep its share of the memory system busy.
// SUBS > 1 (round 5): one workgroup = SUBS neighbouring row blocks (SUBS x 256 threads, every sub-block with its own plan and its
// own LDS region) that walk the variable tiles in LOCKSTEP -- the group's barriers are the workgroup's.  On a mesh whose
// numbering is coherent but not compact (qhull) a row block uses ~4 of the 16 source values of a 128-byte line and its
// neighbours use the rest: as separate workgroups they drift apart and every one of them fetches the line on its own -- the L1
// of a CU holds 256 lines, the L2 of an XCD four microseconds of traffic --, in lockstep on ONE CU the requests for a line meet
// in that CU's L1.
// FAST (round 6; opt-in for `mean` / `first_order_conservative`: option "apply_contract"): on NaN-free tiles the
// products are CONTRACTED into the running sums (one fused multiply-add per entry and variable instead of a multiplication and
// an addition) and the mean is the sum times ONE reciprocal of the row's weight sum (computed once per row, outside the loop
// over the variable tiles) instead of an IEEE division per variable -- the reduction was the kernel's floor (0.71-0.81 ms of
// vector work per K = 256 apply against 0.52 ms of memory time).  The results then differ from the reference's sequential
// loop (regridder.py:41-67) by the roundings of at most n products (n = entries of the row, 4 on average) and one
// multiplication: <= (n + 2) ulp of sum |w v| / sum w -- measured at full size 8e-16 of the field's range, but 1e-9 RELATIVE where a
// mean of mixed-sign data cancels to ~1e-7 of that range.  What it buys (profiles/r06_apply_fast_ab.txt): K = 256 on the
// lattice-numbered pair 1.01 -> 0.92 ms (50 -> 56 % of HBM), on the qhull-numbered benchmark matrix 1.57 -> 1.53 ms (that one is
// bound by its line fills, not by the reduction).  The default stays the reference's operation order, bit for bit.
// The staged tile of distinct source values in LDS is VARIABLE-minor (round 6): the KTILE values of a column in KTILE / 2 consecutive
// 16-byte slots, read with ds_read_b128 (which reaches its rate from one wave per SIMD; ds_read_b64 needs ~4 and the kernel has 3) --
// the slot of a pair XOR-swizzled with the column so that, for a fixed pair, the columns spread over all 16 slot positions of a
// 256-byte LDS row.  Until round 5 it was variable-MAJOR (vals[kk][u]: a ds_read_b64 at a random column per variable).  Same values,
// same arithmetic; measured with gathers and stores off (option plan_dbg = 3, the reduction's floor): merged plan 0.785 -> 0.685 ms,
// per-block plan 0.588 -> 0.502 ms per K = 256 apply; whole kernel - 1.5 ... 2 % (profiles/r06_experiments/apply_lds_layout_ab.txt).
template <int KTILE> __device__ __forceinline__ int plan_slot(int u, int k2) {
    static_assert(KTILE == 8 || KTILE == 4, "tiles of 4 or 8 variables");
    return KTILE == 8 ? u * 4 + (k2 ^ ((u >> 2) & 3)) : u * 2 + (k2 ^ ((u >> 3) & 1));
}
template <int METHOD, typename SRC, int KTILE, int SUBS, bool MERGE = false, bool FAST = false>
__global__ void __launch_bounds__(AP_BLOCK * SUBS)
k_apply_plan(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
             const int32_t *__restrict__ ucol, const int32_t *__restrict__ nuniq, const uint16_t *__restrict__ loc,
             const int32_t *__restrict__ row_order, bool skip_long, int64_t T, int64_t S,
             const SRC *__restrict__ source, int64_t K, double *__restrict__ out, int lmax, int super_blocks, int item_tiles, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int sub = SUBS > 1 ? (int)threadIdx.x / AP_BLOCK : 0, tid = SUBS > 1 ? (int)threadIdx.x % AP_BLOCK : (int)threadIdx.x;
    // MERGE: ONE value table for the workgroup (the group's distinct columns), the entries of every row block behind it
    static_assert(!MERGE || SUBS == PLAN_GROUP, "a merged plan is built for groups of PLAN_GROUP row blocks");
    constexpr int UMAX = MERGE ? PLAN_GUMAX : PLAN_UMAX;                  // columns of the value table
    constexpr int GATHERERS = MERGE ? AP_BLOCK * SUBS : AP_BLOCK;         // threads that share a list
    const size_t stage_bytes = ((sizeof(double) * (size_t)lmax + sizeof(uint16_t) * (size_t)lmax) + 15) / 16 * 16;
    char *smem = MERGE ? smem_all
                       : smem_all + (size_t)sub * (((sizeof(double) * (KTILE * PLAN_UMAX + lmax) + sizeof(uint16_t) * lmax) + 15) / 16 * 16);
    double *vals = reinterpret_cast<double *>(smem);                      // [UMAX][KTILE / 2] 16-byte slots, swizzled (plan_slot)
    double *sh_w = MERGE ? reinterpret_cast<double *>(smem_all + sizeof(double) * KTILE * UMAX + (size_t)sub * stage_bytes)
                         : vals + KTILE * PLAN_UMAX;                      // [lmax] = entries of the largest planned block
    uint16_t *sh_loc = reinterpret_cast<uint16_t *>(sh_w + lmax);         // [lmax]
    constexpr int UPT = UMAX / GATHERERS;                                 // distinct columns per thread
    const int gtid = MERGE ? (int)threadIdx.x : tid;                      // index among the threads that share the list
    // XCD-aware block order: hardware block b runs on XCD b % 8; give every XCD a CONTIGUOUS range of
    // row blocks (= one spatial region), so that lines shared by neighbouring blocks stay in one L2
    const int64_t n_blocks = (T + AP_BLOCK - 1) / AP_BLOCK;           // row blocks
    const int64_t n_groups = (n_blocks + SUBS - 1) / SUBS;            // workgroups' worth of them
    const int64_t per_xcd = (n_groups + 7) / 8;
    int64_t lg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    int64_t k_begin = 0, k_end = K;
    if (super_blocks > 0) {
        // L2-blocked order (round 5).  A block of 256 rows uses ~4 of the 16 source values of every 128-byte line it touches
        // (a qhull-numbered mesh: spatial neighbours are not id neighbours); the rest of the line belongs to NEIGHBOURING row
        // blocks, and a block that walks all K variables on its own drifts away from them -- the line comes from HBM once per
        // block (PMC, K = 128: 24 M fabric reads against 7 M for a lattice-numbered pair).  Here a work item is one row block x
        // `item_tiles` variable tiles, and an XCD's items are ordered (super tile of `super_blocks` neighbouring row blocks) ->
        // (variable item) -> (row block): the blocks resident on an XCD at any time share one super tile and one variable
        // item, whose source lines (~1.4 MB for 64 blocks x 8 variables) stay in that XCD's 4 MB L2.
        const int64_t n_items = (K + (int64_t)KTILE * item_tiles - 1) / ((int64_t)KTILE * item_tiles);
        const int64_t i = blockIdx.x >> 3;
        const int64_t per_super = (int64_t)super_blocks * n_items;
        const int64_t sup = i / per_super, rem = i - sup * per_super;
        const int64_t item = rem / super_blocks, b = rem - item * super_blocks;
        const int64_t local = sup * super_blocks + b;
        if (local >= per_xcd) return;
        lg = (int64_t)(blockIdx.x & 7) * per_xcd + local;
        k_begin = item * KTILE * item_tiles;
        k_end = k_begin + (int64_t)KTILE * item_tiles < K ? k_begin + (int64_t)KTILE * item_tiles : K;
    }
    if (lg >= n_groups) return; // (uniform over the workgroup)
    const int64_t lb = lg * SUBS + sub;
    // (a sub-block without work -- past the last row block, or unplanned: too many entries / distinct columns, k_apply_direct takes
    // those over its block list -- stays for the workgroup's barriers; SUBS == 1: it leaves)
    const int64_t row0 = (lb < n_blocks ? lb : 0) * AP_BLOCK;
    const int nu = MERGE ? nuniq[lg] : (lb < n_blocks ? nuniq[lb] : -1);
    if ((SUBS == 1 || MERGE) && nu < 0) return; // (uniform over the workgroup)
    const bool active = nu >= 0 && lb < n_blocks;
    const int64_t t = active ? row0 + tid : T;
    const int64_t row_end = row0 + AP_BLOCK < T ? row0 + AP_BLOCK : T;
    int s = 0, e = 0;
    if (t < T) {
        s = indptr[t];
        e = indptr[t + 1];
    }
    const bool is_long = skip_long && (e - s > APPLY_LONG); // reduced by k_apply_long
    const int64_t t_out = (t < T && row_order) ? (int64_t)row_order[t] : t;
    // stage the block's entries once
    const int seg0 = indptr[row0], seg1 = active ? indptr[row_end] : seg0;
    for (int j = seg0 + tid; j < seg1; j += AP_BLOCK) {
        sh_w[j - seg0] = data[j];
        sh_loc[j - seg0] = loc[j];
    }
    // this thread's distinct columns
    int64_t mycol[UPT];
#pragma unroll
    for (int q = 0; q < UPT; q++) {
        const int u = q * GATHERERS + gtid;
        mycol[q] = (u < nu && !(dbg & 1)) ? (int64_t)ucol[(MERGE ? lg : lb) * UMAX + u] : -1; // (dbg: measurement switches, XR_PLAN_DBG)
    }
    __syncthreads();
    double normsum = 0.0;
    if (METHOD == XR_GEOMETRIC_MEAN && !is_long)
        for (int j = s; j < e; j++) normsum += sh_w[j - seg0];
    double wsum_row = 0.0; // sum of the row's weights in entry order (= Red::b when no value is NaN)
    if (!is_long)
        for (int j = s; j < e; j++) wsum_row += sh_w[j - seg0];
    const double inv_wsum = FAST ? 1.0 / wsum_row : 0.0; // (FAST: one division per row for all K variables)
    // prologue: request tile 0
    double stage[UPT][KTILE];
    {
        const int kn = (int)(k_end - k_begin < KTILE ? k_end - k_begin : KTILE);
        const SRC *src0 = source + k_begin * S;
#pragma unroll
        for (int q = 0; q < UPT; q++)
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++)
                stage[q][kk] = (mycol[q] >= 0 && kk < kn) ? ld_src(src0, (int64_t)kk * S + mycol[q]) : 0.0;
    }
    for (int64_t k0 = k_begin; k0 < k_end; k0 += KTILE) {
        const int kn = (int)((k_end - k0) < KTILE ? (k_end - k0) : KTILE);
        __syncthreads(); // everybody finished reading the previous tile
        int my_nan = 0;
#pragma unroll
        for (int q = 0; q < UPT; q++) {
            const int u = q * GATHERERS + gtid;
            if (u < nu) {
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++) my_nan |= (stage[q][kk] != stage[q][kk]) ? 1 : 0;
#pragma unroll
                for (int k2 = 0; k2 < KTILE / 2; k2++)
                    reinterpret_cast<double2 *>(vals)[plan_slot<KTILE>(u, k2)] = make_double2(stage[q][2 * k2], stage[q][2 * k2 + 1]);
            }
        }
        // block-uniform: does this tile hold any NaN?  (NaN-free tiles take the short path below)
        const bool tile_has_nan = __syncthreads_or(my_nan) != 0;
        // request the next tile while this one is reduced
        const int64_t k1 = k0 + KTILE;
        if (k1 < k_end) {
            const int kn1 = (int)((k_end - k1) < KTILE ? (k_end - k1) : KTILE);
            const SRC *src1 = source + k1 * S;
#pragma unroll
            for (int q = 0; q < UPT; q++)
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++)
                    stage[q][kk] = (mycol[q] >= 0 && kk < kn1) ? ld_src(src1, (int64_t)kk * S + mycol[q]) : 0.0;
        }
        constexpr bool LINEAR = METHOD == XR_MEAN || METHOD == XR_SUM || METHOD == XR_FIRST_ORDER_CONSERVATIVE;
        if (LINEAR && !tile_has_nan) {
            // no NaN anywhere in the tile: the weight sum of a row is the same for every variable
            // (computed once, wsum_row) and the value sums need no per-entry test.  Same additions
            // in the same order as the general path -> bit-identical results.
            if (t < T && !is_long) {
                double acc[KTILE];
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++) acc[kk] = 0.0;
                for (int j = s - seg0; j < e - seg0; j++) {
                    const int l = sh_loc[j];
                    const double w = sh_w[j];
                    double v[KTILE];
#pragma unroll
                    for (int k2 = 0; k2 < KTILE / 2; k2++) {
                        const double2 t2 = reinterpret_cast<const double2 *>(vals)[plan_slot<KTILE>(l, k2)];
                        v[2 * k2] = t2.x;
                        v[2 * k2 + 1] = t2.y;
                    }
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++) {
                        if (METHOD == XR_SUM) acc[kk] += v[kk];
                        else if (FAST) acc[kk] = fma(w, v[kk], acc[kk]);
                        else if (METHOD == XR_MEAN) acc[kk] += w * v[kk];
                        else acc[kk] += v[kk] * w;
                    }
                }
                if (FAST && METHOD == XR_MEAN) {
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++) acc[kk] *= inv_wsum;
                }
                const bool defined = e > s && wsum_row != 0;
                double *o = out + k0 * T + t_out;
                if (dbg & 2) { // (measurement: no stores -- the compiler must not know)
                    if (acc[0] != 12345.678) continue;
                }
                if (kn == KTILE && !(dbg & 4)) {
                    // whole tile: no per-variable branch around the stores.  Non-temporal: the result is written once and not
                    // read here -- it should not push the source lines that neighbouring row blocks still need out of the L2
                    // (1M x 1M, K = 256: 1.52 -> 1.44 ms qhull-numbered, 1.04 -> 0.99 ms lattice-numbered; XR_PLAN_DBG=4: plain stores)
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++)
                        __builtin_nontemporal_store(defined ? (METHOD == XR_MEAN && !FAST ? acc[kk] / wsum_row : acc[kk]) : NAN, &o[kk * T]);
                } else if (kn == KTILE) {
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++) o[kk * T] = defined ? (METHOD == XR_MEAN && !FAST ? acc[kk] / wsum_row : acc[kk]) : NAN;
                } else {
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++)
                        if (kk < kn) o[kk * T] = defined ? (METHOD == XR_MEAN && !FAST ? acc[kk] / wsum_row : acc[kk]) : NAN;
                }
            }
        } else if (t < T && !is_long) {
            Red<METHOD> red[KTILE];
            for (int j = s - seg0; j < e - seg0; j++) {
                const int l = sh_loc[j];
                const double w = sh_w[j];
                double v[KTILE];
#pragma unroll
                for (int k2 = 0; k2 < KTILE / 2; k2++) {
                    const double2 t2 = reinterpret_cast<const double2 *>(vals)[plan_slot<KTILE>(l, k2)];
                    v[2 * k2] = t2.x;
                    v[2 * k2 + 1] = t2.y;
                }
#pragma unroll
                for (int kk = 0; kk < KTILE; kk++)
                    if (kk < kn) red[kk].add(v[kk], w, normsum);
            }
#pragma unroll
            for (int kk = 0; kk < KTILE; kk++) {
                if (kk < kn) {
                    double r = NAN;
                    if (e > s) {
                        r = red[kk].fin();
                        if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) r = NAN;
                    }
                    out[(k0 + kk) * T + t_out] = r;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row tiling for many source variables.  The stored row order is free (row_order maps stored -> caller
// rows), and with K >= 8 variables the apply is bound by HBM traffic: a block of 256 stored rows that is a
// long strip of targets (consecutive ids of a lattice-numbered mesh) touches a few source values in each
// of very many cache lines, a compact 2-D tile touches long runs (measured on the 1M -> 1M benchmark,
// K = 256: 1.42 -> 1.06 ms).  Rows are regrouped once, by the coarse Morton key xr_overlap attached
// (stable: ascending stored row inside a tile), before the plan is built.
// ---------------------------------------------------------------------------------------------
__global__ void k_bincount(const int32_t *__restrict__ row, int64_t nnz, int32_t *__restrict__ count);

__global__ void k_tile_scatter(const int32_t *__restrict__ key, int64_t n, const int32_t *__restrict__ start,
                               int32_t *__restrict__ cursor, int32_t *__restrict__ members) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) members[start[key[r]] + atomicAdd(&cursor[key[r]], 1)] = (int32_t)r;
}
// position i of the unordered member list -> its rank inside the tile by ascending row id
__global__ void k_tile_rank(const int32_t *__restrict__ key, const int32_t *__restrict__ start,
                            const int32_t *__restrict__ members, int64_t n, int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = members[i];
    const int s = start[key[r]], e = start[key[r] + 1];
    int rank = 0;
    for (int j = s; j < e; j++) rank += members[j] < r;
    perm[s + rank] = r;
}
__global__ void k_tile_len(const int32_t *__restrict__ indptr, const int32_t *__restrict__ perm, int64_t n,
                           int32_t *__restrict__ len) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) len[r] = indptr[perm[r] + 1] - indptr[perm[r]];
}
// one wave per new stored row
__global__ void k_tile_rows(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                            const double *__restrict__ data, const int32_t *__restrict__ row_order,
                            const int32_t *__restrict__ perm, int64_t n, const int32_t *__restrict__ t_indptr,
                            int32_t *__restrict__ t_indices, double *__restrict__ t_data,
                            int32_t *__restrict__ t_row_order, int32_t *__restrict__ long_rows,
                            int32_t *__restrict__ n_long) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    const int old = perm[r];
    const int s = indptr[old], e = indptr[old + 1], d = t_indptr[r];
    for (int j = s + lane; j < e; j += 64) {
        t_indices[d + (j - s)] = indices[j];
        t_data[d + (j - s)] = data[j];
    }
    if (lane == 0) {
        t_row_order[r] = row_order ? row_order[old] : old;
        if (e - s > APPLY_LONG) long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
    }
}

static void ensure_tiled(const xr_csr *ccsr) {
    xr_csr *csr = const_cast<xr_csr *>(ccsr); // like the plan: a cache-friendly re-layout of the same matrix
    if (!csr->has_tile_key) return;
    // (a re-layout of the matrix: only under the exclusive scope -- the apply entry points call prepare_for_apply first,
    // so that the concurrent, shared part of an apply finds this done)
    XR_REQUIRE(exclusive_held(), XR_ERR_INVALID, "internal: row tiling requested outside the exclusive scope");
    csr->has_tile_key = false;
    const int64_t n = csr->n, R = csr->tile_key_range;
    if (n == 0 || R <= 1) {
        csr->tile_key.release();
        return;
    }
    hipStream_t st = launch_stream();
    DevBuf<int32_t> hist((size_t)R + 1), start((size_t)R + 1), members((size_t)n), perm((size_t)n), len((size_t)n);
    fill_i32(hist.get(), 0, R + 1);
    XR_LAUNCH("bincount", k_bincount, dim3(div_up(n, 256)), dim3(256), 0, csr->tile_key.get(), n, hist.get());
    exclusive_scan_i32(hist.get(), start.get(), R);
    fill_i32(hist.get(), 0, R + 1);
    XR_LAUNCH("tile_scatter", k_tile_scatter, dim3(div_up(n, 256)), dim3(256), 0, csr->tile_key.get(), n, start.get(),
              hist.get(), members.get());
    XR_LAUNCH("tile_rank", k_tile_rank, dim3(div_up(n, 256)), dim3(256), 0, csr->tile_key.get(), start.get(),
              members.get(), n, perm.get());
    XR_LAUNCH("tile_len", k_tile_len, dim3(div_up(n, 256)), dim3(256), 0, csr->indptr.get(), perm.get(), n, len.get());
    DevBuf<int32_t> t_indptr((size_t)n + 1), t_indices((size_t)csr->nnz), t_row_order((size_t)n),
        t_long((size_t)(csr->nnz / APPLY_LONG + 1)), t_nlong(1);
    DevBuf<double> t_data((size_t)csr->nnz);
    exclusive_scan_i32(len.get(), t_indptr.get(), n);
    XR_HIP(hipMemsetAsync(t_nlong.get(), 0, sizeof(int32_t), st));
    XR_LAUNCH("tile_rows", k_tile_rows, dim3(div_up(n * 64, 256)), dim3(256), 0, csr->indptr.get(), csr->indices.get(),
              csr->data.get(), csr->has_row_order ? csr->row_order.get() : (const int32_t *)nullptr, perm.get(), n,
              t_indptr.get(), t_indices.get(), t_data.get(), t_row_order.get(), t_long.get(), t_nlong.get());
    const int32_t nl = read_scalar(t_nlong.get());
    csr->indptr = std::move(t_indptr);
    csr->indices = std::move(t_indices);
    csr->data = std::move(t_data);
    csr->row_order = std::move(t_row_order);
    csr->has_row_order = true;
    csr->long_rows = std::move(t_long);
    csr->n_long = std::move(t_nlong);
    csr->has_long = nl > 0;
    csr->plan_ready = false;
    csr->tile_key.release();
}

// option "plan_merge": 1 / 0 force a merged / per-block plan; -1 (default): merged when the blocks use less than PLAN_MERGE_UTIL of the 16
// values of the source lines they touch (measured by the per-block builder: ~4 on a qhull-numbered mesh, ~12 on a
// lattice-numbered one -- there the lockstep of a group costs 2 % and saves nothing)
static constexpr double PLAN_MERGE_UTIL = 6.0;
static int plan_merge_mode() { // (read when a matrix' plan is built, once per matrix: a test can switch between matrices)
    const int64_t e = option(OPT_PLAN_MERGE);
    return e < 0 ? -1 : (e == 1 ? 1 : 0);
}

static void ensure_plan(const xr_csr *ccsr) {
    xr_csr *csr = const_cast<xr_csr *>(ccsr); // the plan is a cache attached to the weights
    if (csr->plan_ready) return;
    XR_REQUIRE(exclusive_held(), XR_ERR_INVALID, "internal: apply plan requested outside the exclusive scope");
    const int64_t nb = (csr->n + AP_BLOCK - 1) / AP_BLOCK;
    bool merged = plan_merge_mode() == 1;
    int64_t n_lists = nb;
    csr->plan_loc.alloc((size_t)csr->nnz);
    csr->plan_lmax = 256;
    csr->plan_n_unplanned = 0;
    // The apply sizes its LDS stage for the LARGEST planned block, and that decides how many blocks a CU holds: one block
    // of 4096 entries among thousands of 1000 (a Delaunay hull) costs every block a third of the occupancy (K = 256 on
    // the benchmark matrix: 1.76 -> 1.2 ms).  Blocks beyond 2048 entries (twice the typical triangle-mesh block) go to
    // the direct kernel instead.
    constexpr int plan_cap = 2048;
    if (nb > 0) {
        DevBuf<int32_t> max_entries(4); // [0] largest planned block, [1] number of unplanned blocks, [2] distinct columns, [3] distinct lines (sums over the planned blocks)
        csr->plan_unplanned.alloc((size_t)nb);
        auto build = [&](bool group) {
            n_lists = group ? (nb + PLAN_GROUP - 1) / PLAN_GROUP : nb;
            csr->plan_ucol.alloc((size_t)n_lists * (group ? PLAN_GUMAX : PLAN_UMAX));
            csr->plan_nuniq.alloc((size_t)n_lists);
            fill_i32(max_entries.get(), 0, 4);
            if (group)
                XR_LAUNCH("plan_build", k_plan_build_group, dim3((unsigned)n_lists), dim3(AP_BLOCK * PLAN_GROUP), 0, csr->indptr.get(),
                          csr->indices.get(), csr->n, csr->plan_ucol.get(), csr->plan_nuniq.get(), csr->plan_loc.get(),
                          max_entries.get(), csr->plan_unplanned.get(), max_entries.get() + 1, plan_cap);
            else
                XR_LAUNCH("plan_build", k_plan_build, dim3((unsigned)nb), dim3(AP_BLOCK), 0, csr->indptr.get(),
                          csr->indices.get(), csr->n, csr->plan_ucol.get(), csr->plan_nuniq.get(), csr->plan_loc.get(),
                          max_entries.get(), csr->plan_unplanned.get(), max_entries.get() + 1, plan_cap);
        };
        build(merged);
        int32_t h[4];
        d2h(h, max_entries.get(), sizeof(h));
        if (!merged && plan_merge_mode() < 0 && h[3] > 0 && nb >= 4 * PLAN_GROUP && (double)h[2] / (double)h[3] < PLAN_MERGE_UTIL) {
            // (poor use of the source lines: the group plan gathers a line once for the group's row blocks)
            merged = true;
            build(true);
            d2h(h, max_entries.get(), sizeof(h));
        }
        const int m = h[0];
        csr->plan_n_unplanned = h[1];
        csr->plan_lmax = std::min(PLAN_LMAX, std::max(256, (m + 255) / 256 * 256));
        if (option(OPT_DEBUG) & 2) {
            std::vector<int32_t> nu((size_t)n_lists);
            d2h(nu.data(), csr->plan_nuniq.get(), sizeof(int32_t) * (size_t)n_lists);
            double sum = 0; int cnt = 0, mx = 0;
            for (auto v : nu) if (v >= 0) { sum += v; cnt++; mx = std::max(mx, (int)v); }
            fprintf(stderr, "[plan] %s blocks %lld planned %d unplanned %d lmax %d avg distinct columns %.1f max %d nnz/block %.1f\n", merged ? "merged (a list per group of row blocks)" : "per block", (long long)nb, cnt,
                    csr->plan_n_unplanned, csr->plan_lmax, cnt ? sum / cnt : 0.0, mx, (double)csr->nnz / nb);
        }
    }
    csr->plan_merged = merged;
    csr->plan_ready = true;
}

// ---------------------------------------------------------------------------------------------
// workspace reducers: mode and percentile.  One thread per (row, k); per-row scratch lives in a
// global workspace laid out like the CSR data (ws[k_local][nnz]).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool nan_le(double a, double b) { // nanpercentile.py:19-27
    if (a != a) return false;
    if (b != b) return true;
    return a < b;
}
__device__ __forceinline__ void dswap(double *A, int i, int j) {
    const double t = A[i];
    A[i] = A[j];
    A[j] = t;
}
__device__ int q_partition(double *A, int low, int high) { // nanpercentile.py:30-63
    const int mid = (low + high) >> 1;
    if (nan_le(A[mid], A[low])) dswap(A, low, mid);
    if (nan_le(A[high], A[mid])) dswap(A, high, mid);
    if (nan_le(A[mid], A[low])) dswap(A, low, mid);
    const double pivot = A[mid];
    dswap(A, high, mid);
    int i = low, j = high - 1;
    for (;;) {
        while (i < high && nan_le(A[i], pivot)) i++;
        while (j >= low && nan_le(pivot, A[j])) j--;
        if (i >= j) break;
        dswap(A, i, j);
        i++;
        j--;
    }
    dswap(A, i, high);
    return i;
}
__device__ double q_select(double *A, int k, int low, int high) { // nanpercentile.py:66-77
    int i = q_partition(A, low, high);
    while (i != k) {
        if (i < k) {
            low = i + 1;
            i = q_partition(A, low, high);
        } else {
            high = i - 1;
            i = q_partition(A, low, high);
        }
    }
    return A[k];
}
__device__ void q_select_two(double *A, int k, int low, int high, double &lo, double &hi) { // :80-102
    for (;;) {
        const int i = q_partition(A, low, high);
        if (i < k) {
            low = i + 1;
        } else if (i > k + 1) {
            high = i - 1;
        } else if (i == k) {
            q_select(A, k + 1, i + 1, high);
            break;
        } else {
            q_select(A, k, low, i - 1);
            break;
        }
    }
    lo = A[k];
    hi = A[k + 1];
}

template <int METHOD, typename SRC>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_workspace(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                  const double *__restrict__ data, const int32_t *__restrict__ row_order, int64_t T, int64_t S,
                  int64_t nnz,
                  const SRC *__restrict__ source, int64_t k_base, double p, double *__restrict__ ws,
                  double *__restrict__ out, bool skip_sorted) {
    const int64_t t = (int64_t)blockIdx.x * AP_BLOCK + threadIdx.x;
    if (t >= T) return;
    const int64_t k = k_base + blockIdx.y;
    const int s = indptr[t], e = indptr[t + 1], n = e - s;
    if (skip_sorted && n > APPLY_LONG && n <= APPLY_WAVE) return; // k_apply_sorted
    double res = NAN;
    if (n > 0) {
        const SRC *src = source + k * S;
        double *w_row = ws + (int64_t)blockIdx.y * nnz + s;
        if (METHOD == XR_MODE) { // reduce.py:126-158
            for (int i = 0; i < n; i++) w_row[i] = data[s + i];
            int w_sum = 0;
            double w_max = 0.0;
            for (int i = 0; i < n; i++) {
                const double v = ld_src(src, indices[s + i]);
                if (v != v) continue;
                const double w = data[s + i];
                if (w > w_max) w_max = w;
                w_sum += 1;
                for (int j = 0; j < i; j++) {
                    if (ld_src(src, indices[s + j]) == v) {
                        w_row[j] += w;
                        break;
                    }
                }
            }
            if (!(w_sum == 0 || w_max == 0.0)) {
                w_max = 0;
                double mode_value = ld_src(src, indices[s]);
                for (int i = 0; i < n; i++) {
                    const double v = ld_src(src, indices[s + i]);
                    if (v == v) {
                        const double wa = w_row[i];
                        if ((wa > w_max) || (wa == w_max && v > mode_value)) {
                            w_max = wa;
                            mode_value = v;
                        }
                    }
                }
                res = mode_value;
            }
        } else { // percentile, reduce.py:161-203
            double w_max = 0.0;
            for (int i = 0; i < n; i++) {
                const double w = data[s + i];
                if (w > w_max) w_max = w;
            }
            if (w_max != 0.0) {
                if (p == 0 || p == 100) {
                    double v_ext = p == 0 ? INFINITY : -INFINITY, wm = 0.0;
                    for (int i = 0; i < n; i++) {
                        const double v = ld_src(src, indices[s + i]);
                        if (v != v) continue;
                        if (p == 0 ? (v < v_ext) : (v > v_ext)) v_ext = v;
                        const double w = data[s + i];
                        if (w > wm) wm = w;
                    }
                    res = wm == 0.0 ? NAN : v_ext;
                } else {
                    int m = 0;
                    for (int i = 0; i < n; i++) {
                        const double v = ld_src(src, indices[s + i]);
                        if (v == v) w_row[m++] = v;
                    }
                    if (m == 1) {
                        res = w_row[0];
                    } else if (m > 1) {
                        const double rank = 1 + (double)(m - 1) * p / 100.0;
                        const double f = floor(rank);
                        const double frac = rank - f;
                        double lower, upper;
                        q_select_two(w_row, (int)(f - 1), 0, m - 1, lower, upper);
                        res = lower * (1 - frac) + upper * frac;
                    }
                }
            }
        }
    }
    out[k * T + (row_order ? (int64_t)row_order[t] : t)] = res;
}

// mode / percentile of rows with APPLY_LONG < entries <= APPLY_WAVE: one WAVE per (row, variable).  The row's
// values are sorted in LDS (bitonic network on (value, entry index) pairs -- a total order, i.e. a stable sort)
// instead of the reference's per-row O(n^2) linear search (mode) / serial quickselect (percentile):
//   percentile  order statistics do not depend on how they are found: sorted[k], sorted[k + 1], same
//               interpolation as reduce.py:196-203 -> identical values
//   mode        equal values become contiguous in entry order, so the weight of a distinct value is summed
//               left to right exactly as the reference accumulates it onto the first occurrence
//               (reduce.py:134-142); the winner is the largest (total weight, value) pair (:150-157)
static constexpr int SORT_MAX = XR_APPLY_WAVE_ROW; // 2048

template <int METHOD, typename SRC>
__global__ void __launch_bounds__(64)
k_apply_sorted(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
               const int32_t *__restrict__ row_order, const int32_t *__restrict__ long_rows,
               const int32_t *__restrict__ n_long, int64_t T, int64_t S, const SRC *__restrict__ source, int64_t K,
               double p, double *__restrict__ out) {
    __shared__ double sv[SORT_MAX];   // value (NaN and padding as +inf: sorted to the end)
    __shared__ uint16_t si[SORT_MAX]; // entry index within the row; bit 15 = not a valid value (NaN / padding), so that
                                      // among equal sort keys (+inf) the valid entries come first
    __shared__ double sw[METHOD == XR_MODE ? SORT_MAX : 1]; // weights in entry order (mode)
    const int lane = threadIdx.x;
    const int nl = *n_long;
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {
        const int t = long_rows[li];
        const int s = indptr[t], n = indptr[t + 1] - s;
        if (n > SORT_MAX) continue; // thread-per-row kernel
        const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
        int np2 = 64;
        while (np2 < n) np2 <<= 1;
        for (int64_t k = blockIdx.y; k < K; k += gridDim.y) {
            const SRC *src = source + k * S;
            __syncthreads();
            // load: values, validity count, largest weight of a valid (mode) / any (percentile) entry
            int n_valid = 0;
            double w_max = 0.0, v_min = INFINITY, v_max = -INFINITY;
            for (int i = lane; i < np2; i += 64) {
                double v = INFINITY;
                if (i < n) {
                    const double x = ld_src(src, indices[s + i]);
                    const double w = data[s + i];
                    if (METHOD == XR_MODE) sw[i] = w;
                    const bool ok = x == x;
                    if (ok) {
                        v = x;
                        n_valid++;
                        v_min = fmin(v_min, x);
                        v_max = fmax(v_max, x);
                    }
                    if (METHOD != XR_MODE || ok) w_max = fmax(w_max, w);
                }
                sv[i] = v;
                si[i] = (uint16_t)(v == INFINITY && !(i < n && ld_src(src, indices[s + i]) == INFINITY) ? (i | 0x8000) : i);
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                n_valid += __shfl_xor(n_valid, d, 64);
                w_max = fmax(w_max, __shfl_xor(w_max, d, 64));
                v_min = fmin(v_min, __shfl_xor(v_min, d, 64));
                v_max = fmax(v_max, __shfl_xor(v_max, d, 64));
            }
            double res = NAN;
            if (METHOD == XR_PERCENTILE && (p == 0 || p == 100)) {
                // reduce.py:172-176: _minimum / _maximum (NaN skipped; NaN if every weight of a valid entry is 0)
                double wm = 0.0;
                for (int i = lane; i < n; i += 64)
                    if (!(si[i] & 0x8000)) wm = fmax(wm, data[s + i]);
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) wm = fmax(wm, __shfl_xor(wm, d, 64));
                if (w_max != 0.0 && n_valid > 0 && wm != 0.0) res = p == 0 ? v_min : v_max;
                if (lane == 0) out[k * T + t_out] = res;
                continue;
            }
            const bool trivial = w_max == 0.0 || n_valid == 0;
            if (!trivial && !(METHOD == XR_PERCENTILE && n_valid == 1)) {
                // bitonic sort of (value, index) ascending
                for (int size = 2; size <= np2; size <<= 1) {
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        __syncthreads();
                        for (int q = lane; q < np2 / 2; q += 64) {
                            const int lo = 2 * q - (q & (stride - 1));
                            const int hi = lo + stride;
                            const bool up = (lo & size) == 0;
                            const double a = sv[lo], b = sv[hi];
                            const uint16_t ia = si[lo], ib = si[hi];
                            const bool a_gt_b = (a > b) || (a == b && ia > ib);
                            if (a_gt_b == up) {
                                sv[lo] = b; sv[hi] = a;
                                si[lo] = ib; si[hi] = ia;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (METHOD == XR_PERCENTILE) {
                if (!trivial) {
                    if (n_valid == 1) {
                        res = v_min;
                    } else {
                        // +inf VALUES sort among the padding; their count does not matter: ranks only reach n_valid - 1
                        const double rank = 1 + (double)(n_valid - 1) * p / 100.0;
                        const double f = floor(rank);
                        const double frac = rank - f;
                        const int kk = (int)(f - 1);
                        const double lower = sv[kk], upper = sv[kk + 1];
                        res = lower * (1 - frac) + upper * frac;
                    }
                }
            } else if (!trivial) {
                // group starts sum their group left to right; then the largest (total, value) pair wins
                double best_w = -1.0, best_v = -INFINITY;
                for (int i = lane; i < n_valid; i += 64) {
                    const double v = sv[i];
                    if (i == 0 || sv[i - 1] != v) {
                        double tot = sw[si[i]];
                        for (int j = i + 1; j < n_valid && sv[j] == v; j++) tot += sw[si[j]];
                        if (tot > best_w || (tot == best_w && v > best_v)) {
                            best_w = tot;
                            best_v = v;
                        }
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const double ow = __shfl_xor(best_w, d, 64), ov = __shfl_xor(best_v, d, 64);
                    if (ow > best_w || (ow == best_w && ov > best_v)) {
                        best_w = ow;
                        best_v = ov;
                    }
                }
                res = best_v;
            }
            if (lane == 0) out[k * T + t_out] = res;
        }
    }
}

// ---- partial states of every shard-decomposable reducer (multi-GPU split over the source faces) ------------------
// (reduce.py:16-123, 206-222; component table in include/xugrid_amd.h)
__host__ __device__ inline int partial_components(int method) {
    switch (method) {
    case XR_MEAN: case XR_FIRST_ORDER_CONSERVATIVE: case XR_SUM: case XR_HARMONIC_MEAN: case XR_MINIMUM: case XR_MAXIMUM: return 2;
    case XR_GEOMETRIC_MEAN: return 4;
    default: return 0;
    }
}
__host__ __device__ inline bool partial_is_max(int method) { return method == XR_MINIMUM || method == XR_MAXIMUM; }

struct PartialState {
    double c[4];
};
__device__ __forceinline__ PartialState partial_identity(int method) {
    PartialState st{{0.0, 0.0, 0.0, 0.0}};
    if (partial_is_max(method)) st.c[0] = -INFINITY;
    return st;
}
__device__ __forceinline__ void partial_add(int method, PartialState &st, double v, double w) {
    switch (method) {
    case XR_MEAN: case XR_FIRST_ORDER_CONSERVATIVE:
        if (v == v) { st.c[0] += w * v; st.c[1] += w; }
        break;
    case XR_SUM:
        if (v == v) { st.c[0] += v; st.c[1] += w; }
        break;
    case XR_HARMONIC_MEAN:
        if (v == v && v != 0 && w > 0) { st.c[0] += w; st.c[1] += w / v; }
        break;
    case XR_GEOMETRIC_MEAN:
        st.c[0] += w;
        if (v > 0 && w > 0) { st.c[1] += w * log(fabs(v)); st.c[2] += w; }
        else if (v < 0) st.c[3] += 1.0;
        break;
    case XR_MINIMUM:
        if (v == v) { st.c[0] = fmax(st.c[0], -v); st.c[1] = fmax(st.c[1], w); }
        break;
    case XR_MAXIMUM:
        if (v == v) { st.c[0] = fmax(st.c[0], v); st.c[1] = fmax(st.c[1], w); }
        break;
    }
}
__device__ __forceinline__ void partial_combine(int method, PartialState &a, const double *b, int64_t stride, int C) {
    // (static component indices: a loop over the run-time C puts the state into scratch memory)
    if (partial_is_max(method)) {
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (c < C) a.c[c] = fmax(a.c[c], b[c * stride]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (c < C) a.c[c] += b[c * stride];
    }
}
__device__ __forceinline__ double partial_finalize(int method, const PartialState &st) {
    switch (method) {
    case XR_MEAN: return st.c[1] == 0 ? NAN : st.c[0] / st.c[1];
    case XR_FIRST_ORDER_CONSERVATIVE: case XR_SUM: return st.c[1] == 0 ? NAN : st.c[0];
    case XR_HARMONIC_MEAN: return (st.c[1] == 0 || st.c[0] == 0) ? NAN : st.c[0] / st.c[1];
    case XR_GEOMETRIC_MEAN:
        // (the reference normalises the weights by their sum first: exp(sum (w / W) ln v / sum (w / W)) -- the same
        // number up to rounding)
        if (st.c[0] == 0 || st.c[3] > 0 || st.c[2] == 0) return NAN;
        return exp((1.0 / (st.c[2] / st.c[0])) * (st.c[1] / st.c[0]));
    case XR_MINIMUM: return st.c[1] == 0 ? NAN : -st.c[0];
    case XR_MAXIMUM: return st.c[1] == 0 ? NAN : st.c[0];
    }
    return NAN;
}

// one thread per (stored row, variable)
template <typename SRC>
__global__ void __launch_bounds__(256)
k_apply_partial(int method, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                const double *__restrict__ data, const int32_t *__restrict__ row_order, int64_t T, int64_t S,
                const SRC *__restrict__ source, int64_t K, double *__restrict__ out, bool rows_layout, bool skip_long) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k = blockIdx.y;
    if (t >= T) return;
    if (skip_long && indptr[t + 1] - indptr[t] > APPLY_LONG) return; // reduced by one wave each (k_apply_partial_long)
    PartialState st = partial_identity(method);
    const SRC *src = source + k * S;
    for (int j = indptr[t]; j < indptr[t + 1]; j++) partial_add(method, st, ld_src(src, indices[j]), data[j]);
    const int C = partial_components(method);
    const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
    for (int c = 0; c < C; c++) {
        if (rows_layout) out[t_out * (C * K) + c * K + k] = st.c[c];
        else out[((int64_t)c * K + k) * T + t_out] = st.c[c];
    }
}

// K = 1, one specialisation per decomposable reducer (round 5): the partial state of 64 stored rows per WAVE through the wave's
// private LDS window, exactly as k_apply_rows1 reduces them -- the CSR segment of the 64 rows is contiguous: column and weight
// loaded coalesced, the source value gathered by the same lane and parked next to the weight, then every lane walks ITS row in
// CSR order (the same additions in the same order as k_apply_partial's thread-per-row loop, so the states are bit-identical).
// With the reducer a run-time switch this form was slower than the plain kernel (73 against 61 us, round 4: four state
// components and a switch per entry in the walk); as a template the walk of `mean` is two selects per entry.
// Rows beyond APPLY_LONG entries are left to k_apply_partial_long.
template <int METHOD, typename SRC>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_partial_w1(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ data,
                   const int32_t *__restrict__ row_order, int64_t T, int64_t S, const SRC *__restrict__ source,
                   double *__restrict__ out, bool rows_layout, bool skip_long, const int32_t *__restrict__ gate,
                   const int32_t *__restrict__ long_rows, const int32_t *__restrict__ n_long, int n_long_blocks) {
    __shared__ double2 sh_win[AP_BLOCK / 64][W1_CAP]; // (.x = weight, .y = source value)
    if (gate && *gate == 0) return; // (enqueued behind a weight build whose attempt failed: the host redoes both)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    constexpr int C = METHOD == XR_GEOMETRIC_MEAN ? 4 : 2;
    if ((int)blockIdx.x < n_long_blocks) {
        // the listed long rows (hull slivers), one WAVE each, in the blocks dispatched first: beside the short rows instead of
        // in a launch of their own behind them (as k_apply_rows1; the same strided walk and butterfly as k_apply_partial_long)
        const int nl = *n_long;
        for (int64_t li = (int64_t)blockIdx.x * (AP_BLOCK / 64) + wib; li < nl; li += (int64_t)n_long_blocks * (AP_BLOCK / 64)) {
            const int t = long_rows[li];
            PartialState st = partial_identity(METHOD);
            for (int j = indptr[t] + lane; j < indptr[t + 1]; j += 64) partial_add(METHOD, st, ld_src(source, indices[j]), data[j]);
#pragma unroll
            for (int c = 0; c < C; c++) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const double o = __shfl_xor(st.c[c], d, 64);
                    st.c[c] = partial_is_max(METHOD) ? fmax(st.c[c], o) : st.c[c] + o;
                }
            }
            if (lane == 0) {
                const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if (rows_layout) out[t_out * C + c] = st.c[c];
                    else out[(int64_t)c * T + t_out] = st.c[c];
                }
            }
        }
        return;
    }
    const int64_t row0 = ((int64_t)(gridDim.x - 1 - blockIdx.x) * (AP_BLOCK / 64) + wib) * 64;
    if (row0 >= T) return;
    const int64_t t = row0 + lane;
    int s = 0, e = 0;
    if (t < T) {
        s = indptr[t];
        e = indptr[t + 1];
    }
    const bool skip = skip_long && (e - s > APPLY_LONG);
    const int seg0 = __shfl(s, 0, 64);
    int seg1 = e;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) seg1 = max(seg1, __shfl_xor(seg1, d, 64));
    if (skip) e = s;
    const bool jumpy = __any(skip);
    double2 *win = sh_win[wib];
    auto next_chunk = [&](int c0) -> int { // first entry any lane still needs at or behind c0 (skipped long rows are not streamed)
        if (!jumpy) return c0;
        int mine = (e > s && e > c0) ? (s > c0 ? s : c0) : INT_MAX;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine = min(mine, __shfl_xor(mine, d, 64));
        return mine;
    };
    PartialState st = partial_identity(METHOD);
    for (int c0 = seg0; c0 < seg1; c0 += W1_CAP) {
        c0 = next_chunk(c0);
        if (c0 >= seg1) break;
        {
            constexpr int PER = W1_CAP / 64;
            int col[PER];
            double w[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int j = c0 + u * 64 + lane;
                col[u] = j < seg1 ? indices[j] : -1;
                w[u] = j < seg1 ? data[j] : 0.0;
            }
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) v[u] = col[u] >= 0 ? ld_src(source, (int64_t)col[u]) : 0.0;
#pragma unroll
            for (int u = 0; u < PER; u++)
                if (col[u] >= 0) win[u * 64 + lane] = make_double2(w[u], v[u]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int a = s > c0 ? s : c0, b = e < c0 + W1_CAP ? e : c0 + W1_CAP;
        for (int j = a; j < b; j++) {
            const double2 wv = win[j - c0];
            partial_add(METHOD, st, wv.y, wv.x); // (METHOD is a constant: the switch folds to the reducer's own lines)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the window is overwritten by the next chunk)
        __builtin_amdgcn_wave_barrier();
    }
    if (t < T && !skip) {
        const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
#pragma unroll
        for (int c = 0; c < C; c++) {
            if (rows_layout) out[t_out * C + c] = st.c[c];
            else out[(int64_t)c * T + t_out] = st.c[c];
        }
    }
}

// the listed long rows (hull slivers: thousands of entries): one wave per (row, variable), lanes stride the row,
// butterfly combine of the states (fixed order)
// KT variables per thread (K >= KT): the row's entries are read once per tile instead of once per variable, the KT
// gathers of an entry are independent loads in flight, and in the rows layout a thread writes KT consecutive doubles per
// component (whole 64-byte lines) where the one-variable kernel scatters single doubles C * K apart.  Same additions in
// the same order per variable.  (One rank, 1M x 1M matrix, 32 variables per call: see DESIGN section 6.)
template <typename SRC, int KT>
__global__ void __launch_bounds__(256)
k_apply_partial_kt(int method, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                   const double *__restrict__ data, const int32_t *__restrict__ row_order, int64_t T, int64_t S,
                   const SRC *__restrict__ source, int64_t K, double *__restrict__ out, bool rows_layout, bool skip_long) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.y * KT;
    if (t >= T) return;
    const int s = indptr[t], e = indptr[t + 1];
    if (skip_long && e - s > APPLY_LONG) return; // reduced by one wave each (k_apply_partial_long)
    const int kn = (int)((K - k0) < KT ? (K - k0) : KT);
    PartialState st[KT];
#pragma unroll
    for (int kk = 0; kk < KT; kk++) st[kk] = partial_identity(method);
    const SRC *src = source + k0 * S;
    for (int j = s; j < e; j++) {
        const int64_t col = indices[j];
        const double w = data[j];
        double v[KT];
#pragma unroll
        for (int kk = 0; kk < KT; kk++) v[kk] = kk < kn ? ld_src(src, (int64_t)kk * S + col) : 0.0;
#pragma unroll
        for (int kk = 0; kk < KT; kk++)
            if (kk < kn) partial_add(method, st[kk], v[kk], w);
    }
    const int C = partial_components(method);
    const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
#pragma unroll
    for (int c = 0; c < 4; c++) { // (static indices into the states: see partial_combine)
        if (c >= C) break;
#pragma unroll
        for (int kk = 0; kk < KT; kk++) {
            if (kk < kn) {
                if (rows_layout) out[t_out * (C * K) + c * K + k0 + kk] = st[kk].c[c];
                else out[((int64_t)c * K + k0 + kk) * T + t_out] = st[kk].c[c];
            }
        }
    }
}
// The same for the ROWS layout (T, C * K) of the sparse exchange.  There a row's states lie C * K doubles apart from the
// next row's: a thread that stores its own doubles makes every store instruction of the wave touch 64 different lines, 8
// bytes each (measured: 2.2 ms for 32 variables x 1M rows, one rank -- 7 x what the bytes cost).  The block therefore
// parks its 128 rows x C x KT states in LDS and writes them out with 8 lanes per (row, component): whole 64-byte lines.
template <typename SRC, int KT>
__global__ void __launch_bounds__(128)
k_apply_partial_rows(int method, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                     const double *__restrict__ data, const int32_t *__restrict__ row_order, int64_t T, int64_t S,
                     const SRC *__restrict__ source, int64_t K, double *__restrict__ out, bool skip_long) {
    static_assert(KT == 8, "one 64-byte line per (row, component)");
    extern __shared__ double sh_dyn[]; // [128][C * KT + 1] states (sized for the reducer's C: 17 KB for two components)
    __shared__ int32_t sh_tout[128];
    const int64_t t = (int64_t)blockIdx.x * 128 + threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.y * KT;
    const int kn = (int)((K - k0) < KT ? (K - k0) : KT);
    const int C = partial_components(method);
    const int ld = C * KT + 1;
    int s = 0, e = 0;
    bool mine = false;
    if (t < T) {
        s = indptr[t];
        e = indptr[t + 1];
        mine = !(skip_long && e - s > APPLY_LONG); // (long rows: one wave each, k_apply_partial_long)
    }
    PartialState st[KT];
#pragma unroll
    for (int kk = 0; kk < KT; kk++) st[kk] = partial_identity(method);
    const SRC *src = source + k0 * S;
    if (mine) {
        for (int j = s; j < e; j++) {
            const int64_t col = indices[j];
            const double w = data[j];
            double v[KT];
#pragma unroll
            for (int kk = 0; kk < KT; kk++) v[kk] = kk < kn ? ld_src(src, (int64_t)kk * S + col) : 0.0;
#pragma unroll
            for (int kk = 0; kk < KT; kk++)
                if (kk < kn) partial_add(method, st[kk], v[kk], w);
        }
    }
    sh_tout[threadIdx.x] = mine ? (int32_t)(row_order ? row_order[t] : t) : -1;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (c >= C) break;
#pragma unroll
        for (int kk = 0; kk < KT; kk++) sh_dyn[threadIdx.x * ld + c * KT + kk] = st[kk].c[c];
    }
    __syncthreads();
    // 8 lanes per (row, component) line
    const int n_lines = 128 * C;
    for (int line = threadIdx.x >> 3; line < n_lines; line += 16) {
        const int r = line / C, c = line - r * C, kk = threadIdx.x & 7;
        const int t_out = sh_tout[r];
        if (t_out >= 0 && kk < kn) out[(int64_t)t_out * (C * K) + (int64_t)c * K + k0 + kk] = sh_dyn[r * ld + c * KT + kk];
    }
}

// Combination + finalisation of the received rows (sparse exchange), coalesced both ways: a block takes 64 owned targets
// x a chunk of 32 variables; a thread reads the states of ITS (target, variable) with the variable running fastest across
// the lanes -- 256-byte runs of the (R, C * K) rows -- and the finalised values cross LDS so that the (K, n_targets)
// output is written with the target running fastest.  (One thread per output element read one double per 64-byte line
// and line 8 times over: 1.4 ms for 32 variables x 1M targets.)
__global__ void __launch_bounds__(256)
k_reduce_partial_rows_t(int method, const double *__restrict__ rows, const int64_t *__restrict__ indptr,
                        const int64_t *__restrict__ order, int64_t n_targets, int64_t K, double *__restrict__ out) {
    constexpr int TT = 64, KC = 32;
    __shared__ double sh_val[KC][TT + 1];
    const int C = partial_components(method);
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    for (int64_t kc0 = (int64_t)blockIdx.y * KC; kc0 < K; kc0 += (int64_t)gridDim.y * KC) {
        const int kk = threadIdx.x & (KC - 1);
        const int64_t k = kc0 + kk;
        for (int tl = threadIdx.x / KC; tl < TT; tl += 256 / KC) {
            const int64_t t = t0 + tl;
            double v = NAN;
            if (t < n_targets && k < K) {
                PartialState st = partial_identity(method);
                for (int64_t j = indptr[t]; j < indptr[t + 1]; j++) partial_combine(method, st, rows + order[j] * (C * K) + k, K, C);
                v = partial_finalize(method, st);
            }
            sh_val[kk][tl] = v;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < KC * TT; q += 256) {
            const int kq = q / TT, tl = q - kq * TT;
            if (t0 + tl < n_targets && kc0 + kq < K) out[(kc0 + kq) * n_targets + t0 + tl] = sh_val[kq][tl];
        }
        __syncthreads();
    }
}

template <typename SRC>
__global__ void __launch_bounds__(256)
k_apply_partial_long(int method, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                     const double *__restrict__ data, const int32_t *__restrict__ row_order,
                     const int32_t *__restrict__ long_rows, const int32_t *__restrict__ n_long, int64_t T, int64_t S,
                     const SRC *__restrict__ source, int64_t K, double *__restrict__ out, bool rows_layout,
                     const int32_t *__restrict__ gate = nullptr) {
    if (gate && *gate == 0) return; // (see k_apply_partial_w1)
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (int64_t)gridDim.x * 4;
    const int64_t k = blockIdx.y;
    const int nl = *n_long;
    const int C = partial_components(method);
    const SRC *src = source + k * S;
    for (int64_t li = wave; li < nl; li += n_waves) {
        const int t = long_rows[li];
        PartialState st = partial_identity(method);
        for (int j = indptr[t] + lane; j < indptr[t + 1]; j += 64) partial_add(method, st, ld_src(src, indices[j]), data[j]);
        for (int c = 0; c < C; c++) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double o = __shfl_xor(st.c[c], d, 64);
                st.c[c] = partial_is_max(method) ? fmax(st.c[c], o) : st.c[c] + o;
            }
        }
        if (lane == 0) {
            const int64_t t_out = row_order ? (int64_t)row_order[t] : t;
            for (int c = 0; c < C; c++) {
                if (rows_layout) out[t_out * (C * K) + c * K + k] = st.c[c];
                else out[((int64_t)c * K + k) * T + t_out] = st.c[c];
            }
        }
    }
}

__global__ void k_partial_identity(int method, double *__restrict__ planes, int64_t per_component) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int C = partial_components(method);
    if (i >= per_component * C) return;
    const PartialState st = partial_identity(method);
    planes[i] = st.c[i / per_component];
}

__global__ void k_finalize_partial(int method, const double *__restrict__ planes, int64_t n /* K * T */,
                                   double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    PartialState st;
    const int C = partial_components(method);
    for (int c = 0; c < C; c++) st.c[c] = planes[(int64_t)c * n + i];
    out[i] = partial_finalize(method, st);
}

__global__ void k_reduce_partial_rows(int method, const double *__restrict__ rows, const int64_t *__restrict__ indptr,
                                      const int64_t *__restrict__ order, int64_t n_targets, int64_t K,
                                      double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_targets * K) return;
    const int64_t k = i / n_targets, t = i - k * n_targets;
    const int C = partial_components(method);
    PartialState st = partial_identity(method);
    for (int64_t j = indptr[t]; j < indptr[t + 1]; j++) partial_combine(method, st, rows + order[j] * (C * K) + k, K, C);
    out[i] = partial_finalize(method, st);
}

// One variable: a thread per owned target, one specialisation per reducer (the components are compile-time registers, a row of
// the (R, C) state table is one or two 16-byte loads) -- the run-time form above indexes the state by a loop over C and took
// 40 us for 1M targets of one row each, 2.5 x what its 40 MB need.
template <int METHOD>
__global__ void __launch_bounds__(256)
k_reduce_partial_rows_k1(const double *__restrict__ rows, const int64_t *__restrict__ indptr, const int64_t *__restrict__ order,
                         int64_t n_targets, double *__restrict__ out) {
    constexpr int C = METHOD == XR_GEOMETRIC_MEAN ? 4 : 2;
    constexpr bool IS_MAX = METHOD == XR_MINIMUM || METHOD == XR_MAXIMUM;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_targets) return;
    const int64_t j0 = indptr[t], j1 = indptr[t + 1];
    PartialState st = partial_identity(METHOD);
    for (int64_t j = j0; j < j1; j++) {
        const double2 *row = reinterpret_cast<const double2 *>(rows + order[j] * C);
        const double2 a = row[0];
        if (IS_MAX) {
            st.c[0] = fmax(st.c[0], a.x);
            st.c[1] = fmax(st.c[1], a.y);
        } else {
            st.c[0] += a.x;
            st.c[1] += a.y;
        }
        if (C == 4) {
            const double2 b = row[1];
            st.c[2] += b.x;
            st.c[3] += b.y;
        }
    }
    out[t] = partial_finalize(METHOD, st);
}

template <typename SRC>
__global__ void k_apply_coo(const int32_t *__restrict__ row, const int32_t *__restrict__ col, int64_t nnz, int64_t T,
                            int64_t S, const SRC *__restrict__ source, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const int64_t k = blockIdx.y;
    out[k * T + row[i]] = ld_src(source + k * S, col[i]);
}

static inline const int32_t *row_order_of(const xr_csr *csr) {
    return (csr->has_row_order && !csr->output_stored) ? csr->row_order.get() : nullptr;
}

template <int METHOD, typename SRC>
static void launch_stream(const xr_csr *csr, const SRC *src, int64_t K, double *out) {
    // The long rows (hull slivers: a few hundred latency-bound rows).  With many variables they run on the side stream BESIDE
    // the short ones -- disjoint output rows (0.25 ms of the 1.9 ms at K = 256).  For one variable the fork / join events
    // cost more than the 30 us they could hide (0.069 -> 0.075 ms): there they follow apply_stream in line, which also
    // zeroes their queue counter (no memset launch), and the block-per-row kernel is skipped when the builder reported
    // that no row exceeds the wave kernel's reach.
    DevBuf<int32_t> huge;
    if (csr->has_long) huge.alloc((size_t)(csr->nnz / APPLY_WAVE + 2));
    auto launch_long_rows = [&](bool zeroed) {
        if (!zeroed) XR_HIP(hipMemsetAsync(huge.get(), 0, sizeof(int32_t), launch_stream()));
        if (K == 1) {
            // rows of APPLY_LONG + 1 ... APPLY_WAVE entries: lane groups / waves; a single variable
            dim3 wgrid((unsigned)engine().num_cu * 16, 1);
            XR_LAUNCH("apply_wave", (k_apply_wave<METHOD, SRC, 1>), wgrid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(),
                      csr->n, csr->m, src, K, out, huge.get());
        } else {
            // ... 8 variables per pass
            constexpr int WT = 8;
            const unsigned wy = (unsigned)std::min<int64_t>(div_up(K, WT), 8);
            dim3 wgrid((unsigned)engine().num_cu * 16 / wy, wy);
            XR_LAUNCH("apply_wave", (k_apply_wave<METHOD, SRC, WT>), wgrid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(),
                      csr->n, csr->m, src, K, out, huge.get());
        }
        if (csr->max_row_len >= 0 && csr->max_row_len <= APPLY_WAVE) return; // no row for the block kernel
        // rows beyond APPLY_WAVE entries, as queued by the wave kernel: one block each
        const unsigned gy = (unsigned)(K < 8 ? K : 8);
        dim3 grid((unsigned)engine().num_cu * (8 / gy), gy); // 8 blocks per CU; blocks past the count exit at once
        XR_LAUNCH("apply_long", (k_apply_long<METHOD, SRC>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                  csr->indices.get(), csr->data.get(), row_order_of(csr), huge.get() + 1, huge.get(),
                  csr->n, csr->m, src, K, out);
    };
    const bool long_on_side = csr->has_long && K >= PLAN_KT;
    const bool use_plan = K > 1 && K >= PLAN_KT && option(OPT_APPLY_PLAN) != 0; // (measurement / test switch)
    if (use_plan) { // (built once per matrix, on the main stream, in front of the fork)
        ensure_tiled(csr);
        ensure_plan(csr);
    }
    // the few blocks the plan could not take (hull slivers: too many entries or distinct columns): direct gathers, parallel
    // over the variable tiles as well so that no thread walks a long row K / 8 times (tiles of 4 variables: these blocks hold
    // rows of up to 256 entries that one thread walks alone).  Beside the planned blocks when the side stream is in use
    // anyway (disjoint output rows; in line they were 0.07 ms behind the 1.5 ms of a K = 256 apply).
    auto launch_unplanned = [&]() {
        if (csr->plan_n_unplanned <= 0) return;
        dim3 grid((unsigned)csr->plan_n_unplanned, div_up(K, 4));
        XR_LAUNCH("apply_direct", (k_apply_direct<METHOD, SRC, 4>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                  csr->indices.get(), csr->data.get(), row_order_of(csr), csr->has_long, csr->n, csr->m, src, K, out,
                  csr->plan_unplanned.get());
    };
    // with a plan the side work is launched BEHIND the plan kernel but depends only on what precedes it (side_mark): the host
    // reaches the long kernel's launch first, the three short side launches follow while it runs
    const bool side_late = long_on_side && use_plan && side_mark();
    if (long_on_side && !side_late) {
        SideScope side;
        launch_long_rows(false);
        if (use_plan) launch_unplanned();
    }
    if (K == 1) {
        // one launch: long-row blocks in front, then one wave per 64 stored rows
        const int n_long_blocks = csr->has_long ? engine().num_cu / 2 : 0;
        const bool any_huge = csr->has_long && !(csr->max_row_len >= 0 && csr->max_row_len <= APPLY_WAVE);
        dim3 grid((unsigned)(div_up(csr->n, AP_BLOCK) + n_long_blocks), 1);
        XR_LAUNCH("apply_rows1", (k_apply_rows1<METHOD, SRC>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(), csr->indices.get(),
                  csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(), n_long_blocks, any_huge,
                  csr->n, csr->m, src, out, csr->apply_gated ? csr->n_long.get() + 1 : (const int32_t *)nullptr);
    } else {
        if (use_plan) {
            // many variables: rows regrouped into 2-D tiles, then a blocked CSR with per-block distinct-column
            // lists (both built once per matrix, above)
            // LDS: one tile of distinct source values + the block's entries (weight + 16-bit local column); sized
            // for the largest planned block, so typical matrices run three blocks per CU instead of two
            // (tuning hooks: row blocks per workgroup, variables per tile, the L2-blocked order of k_apply_plan: super tiles of
            // `super_blocks` workgroups x items of `item_tiles` variable tiles)
            const bool merged = csr->plan_merged;
            const int plan_subs = merged ? PLAN_GROUP : 1;
            // contracted products + one reciprocal per row (k_apply_plan, FAST): opt-in
            constexpr bool FAST_OK = METHOD == XR_MEAN || METHOD == XR_FIRST_ORDER_CONSERVATIVE;
            const bool fast = FAST_OK && option(OPT_APPLY_CONTRACT) != 0;
            // default: items of 64 variables, super tile = the XCD's whole range of row blocks -- every XCD sweeps its row blocks
            // once per 64 variables (the source planes in flight: 0.5 GB instead of all K of them; what the host-side split
            // into groups of 128 variables did until round 4, without its extra launches, forks and joins)
            const int plan_dbg = (int)option(OPT_PLAN_DBG); // measurement: 1 no gathers, 2 no stores, 4 plain instead of non-temporal stores
            // LDS per row block: a tile of distinct source values + the block's entries (weight + 16-bit local column), sized for
            // the largest planned block of the matrix.  4 row blocks per workgroup: tiles of 4 variables (4 x 34 KB)
            const int plan_kt = plan_subs > 1 ? 4 : PLAN_KT;
            auto sub_bytes = [&](int kt, int lmax) {
                return ((sizeof(double) * ((size_t)kt * PLAN_UMAX + lmax) + sizeof(uint16_t) * lmax) + 15) / 16 * 16;
            };
            const size_t stage_bytes = ((sizeof(double) * (size_t)csr->plan_lmax + sizeof(uint16_t) * (size_t)csr->plan_lmax) + 15) / 16 * 16;
            const size_t shmem = merged ? sizeof(double) * (size_t)plan_kt * PLAN_GUMAX + stage_bytes * PLAN_GROUP
                                        : sub_bytes(plan_kt, csr->plan_lmax) * plan_subs;
            XR_REQUIRE(shmem <= (size_t)160 * 1024, XR_ERR_LIMIT, "internal: apply plan needs %zu bytes of LDS", shmem);
            // (dynamic LDS beyond 64 KB has to be allowed per kernel; applies run concurrently under the shared scope, so the
            // high-water mark per instantiation sits behind a mutex)
            auto allow_lds = [&](const void *kernel, size_t &granted) {
                static std::mutex attr_mutex;
                std::lock_guard<std::mutex> lock(attr_mutex);
                if (shmem <= granted) return;
                XR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                granted = shmem;
            };
            static size_t granted1 = 0, granted_m = 0, granted1f = 0, granted_mf = 0;
            if (merged && fast) allow_lds(reinterpret_cast<const void *>(&k_apply_plan<METHOD, SRC, 4, PLAN_GROUP, true, FAST_OK>), granted_mf);
            else if (merged) allow_lds(reinterpret_cast<const void *>(&k_apply_plan<METHOD, SRC, 4, PLAN_GROUP, true>), granted_m);
            else if (fast) allow_lds(reinterpret_cast<const void *>(&k_apply_plan<METHOD, SRC, PLAN_KT, 1, false, FAST_OK>), granted1f);
            else allow_lds(reinterpret_cast<const void *>(&k_apply_plan<METHOD, SRC, PLAN_KT, 1>), granted1);
            const int64_t n_groups = div_up(div_up(csr->n, AP_BLOCK), plan_subs);
            const int item_tiles = 64 / plan_kt;
            const int super_blocks = K > (int64_t)plan_kt * item_tiles ? (int)std::min<int64_t>((n_groups + 7) / 8, 1 << 30) : 0;
            int64_t plan_grid = (n_groups + 7) / 8 * 8;
            if (super_blocks > 0) {
                const int64_t per_xcd = (n_groups + 7) / 8;
                const int64_t n_items = div_up(K, (int64_t)plan_kt * item_tiles);
                plan_grid = 8 * div_up(per_xcd, super_blocks) * super_blocks * n_items;
            }
#define XR_PLAN_LAUNCH(KT, SUBS, MERGED, FASTF)                                                                                      \
    XR_LAUNCH("apply_plan", (k_apply_plan<METHOD, SRC, KT, SUBS, MERGED, FASTF>), dim3((unsigned)plan_grid), dim3(AP_BLOCK * SUBS),   \
              shmem, csr->indptr.get(), csr->indices.get(), csr->data.get(), csr->plan_ucol.get(), csr->plan_nuniq.get(),            \
              csr->plan_loc.get(), row_order_of(csr), csr->has_long, csr->n, csr->m, src, K, out, csr->plan_lmax, super_blocks,      \
              item_tiles, plan_dbg)
            if (merged && fast) XR_PLAN_LAUNCH(4, PLAN_GROUP, true, FAST_OK);
            else if (merged) XR_PLAN_LAUNCH(4, PLAN_GROUP, true, false);
            else if (fast) XR_PLAN_LAUNCH(PLAN_KT, 1, false, FAST_OK);
            else XR_PLAN_LAUNCH(PLAN_KT, 1, false, false);
#undef XR_PLAN_LAUNCH
            if (side_late) {
                SideScope side(true);
                launch_long_rows(false);
                launch_unplanned();
            } else if (!long_on_side) {
                launch_unplanned();
            }
        } else {
            // a few variables: register-resident k-tiles, direct gathers, no LDS
            dim3 grid(div_up(csr->n, AP_BLOCK), div_up(K, KT));
            XR_LAUNCH("apply_direct", (k_apply_direct<METHOD, SRC, KT>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->has_long, csr->n, csr->m, src, K,
                      out, (const int32_t *)nullptr);
        }
    }
    if (csr->has_long && K > 1 && !long_on_side) launch_long_rows(false); // (a few variables: in line, behind the short rows)
    if (long_on_side) side_join();
}

template <int METHOD, typename SRC>
static void launch_workspace(const xr_csr *csr, const SRC *src, int64_t K, double p, double *out) {
    // scratch budget: at most ~512 MB of per-(k,row) workspace at a time
    int64_t kchunk = csr->nnz > 0 ? ((int64_t)64 << 20) / csr->nnz : K;
    if (kchunk < 1) kchunk = 1;
    if (kchunk > K) kchunk = K;
    if (kchunk > 65535) kchunk = 65535;
    DevBuf<double> ws((size_t)kchunk * (size_t)(csr->nnz > 0 ? csr->nnz : 1));
    for (int64_t k0 = 0; k0 < K; k0 += kchunk) {
        const int64_t kc = (K - k0) < kchunk ? (K - k0) : kchunk;
        dim3 grid(div_up(csr->n, AP_BLOCK), (unsigned)kc);
        XR_LAUNCH("apply_workspace", (k_apply_workspace<METHOD, SRC>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),
                  csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m, csr->nnz, src, k0, p,
                  ws.get(), out, csr->has_long);
    }
    if (csr->has_long) {
        // rows of APPLY_LONG + 1 ... APPLY_WAVE entries: sorted in LDS by one wave per (row, variable)
        const unsigned gy = (unsigned)std::min<int64_t>(K, 16);
        dim3 grid((unsigned)engine().num_cu * 32 / gy, gy);
        XR_LAUNCH("apply_sorted", (k_apply_sorted<METHOD, SRC>), grid, dim3(64), 0, csr->indptr.get(), csr->indices.get(),
                  csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(), csr->n, csr->m, src, K, p,
                  out);
    }
}

template <typename SRC>
static void apply_dispatch(const xr_csr *csr, int method, double p, const SRC *src, int64_t K, double *out) {
    if (csr->n == 0 || K == 0) return;
    // (Until round 4 many variables were walked in host-side groups of 128; since round 5 the many-variable kernel orders its
    // own work by groups of 64 variables -- k_apply_plan, L2-blocked order.)
    switch (method) {
    case XR_MEAN: launch_stream<XR_MEAN, SRC>(csr, src, K, out); break;
    case XR_HARMONIC_MEAN: launch_stream<XR_HARMONIC_MEAN, SRC>(csr, src, K, out); break;
    case XR_GEOMETRIC_MEAN: launch_stream<XR_GEOMETRIC_MEAN, SRC>(csr, src, K, out); break;
    case XR_SUM: launch_stream<XR_SUM, SRC>(csr, src, K, out); break;
    case XR_MINIMUM: launch_stream<XR_MINIMUM, SRC>(csr, src, K, out); break;
    case XR_MAXIMUM: launch_stream<XR_MAXIMUM, SRC>(csr, src, K, out); break;
    case XR_FIRST_ORDER_CONSERVATIVE: launch_stream<XR_FIRST_ORDER_CONSERVATIVE, SRC>(csr, src, K, out); break;
    case XR_MAX_OVERLAP: launch_stream<XR_MAX_OVERLAP, SRC>(csr, src, K, out); break;
    case XR_SELECT: launch_stream<XR_SELECT, SRC>(csr, src, K, out); break;
    case XR_MODE: launch_workspace<XR_MODE, SRC>(csr, src, K, p, out); break;
    case XR_PERCENTILE:
        XR_REQUIRE(p >= 0.0 && p <= 100.0, XR_ERR_INVALID,
                   "percentile must be in the range [0, 100], received: %g", p);
        launch_workspace<XR_PERCENTILE, SRC>(csr, src, K, p, out);
        break;
    default: XR_REQUIRE(false, XR_ERR_INVALID, "unknown reducer id %d", method);
    }
}

// the caller's source block into the stored column order: dst[k][j] = src[k][col_of[j]] (every source line is read once;
// the columns of a line are spatial neighbours, so they are consumed within a short stretch of j: L2 hits)
template <typename SRC>
__global__ void __launch_bounds__(256)
k_permute_source(const SRC *__restrict__ src, const int32_t *__restrict__ col_of, int64_t S, SRC *__restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k = blockIdx.y;
    if (j < S) dst[k * S + j] = src[k * S + col_of[j]];
}

// -> the source block in the STORED column order (the caller's block itself unless the columns were renumbered)
static const void *stored_source(const xr_csr *csr, const void *src, int dtype, int64_t K, DevBuf<char> &permuted) {
    if (!(csr->has_col_perm && !csr->source_permuted && K > 0 && csr->m > 0)) return src;
    const size_t esz = dtype == XR_F64 ? 8 : 4;
    permuted.alloc((size_t)K * (size_t)csr->m * esz);
    for (int64_t k0 = 0; k0 < K; k0 += 65535) {
        const int64_t kc = std::min<int64_t>(K - k0, 65535);
        dim3 grid(div_up(csr->m, 256), (unsigned)kc);
        if (dtype == XR_F64)
            XR_LAUNCH("permute_source", k_permute_source<double>, grid, dim3(256), 0,
                      static_cast<const double *>(src) + k0 * csr->m, csr->col_of.get(), csr->m,
                      reinterpret_cast<double *>(permuted.get()) + k0 * csr->m);
        else
            XR_LAUNCH("permute_source", k_permute_source<float>, grid, dim3(256), 0,
                      static_cast<const float *>(src) + k0 * csr->m, csr->col_of.get(), csr->m,
                      reinterpret_cast<float *>(permuted.get()) + k0 * csr->m);
    }
    return permuted.get();
}

static void apply_dev(const xr_csr *csr, int method, double p, const void *src, int dtype, int64_t K, double *out) {
    XR_REQUIRE(dtype == XR_F64 || dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d", dtype);
    DevBuf<char> permuted; // (goes back to the pool at return; the pool hands blocks out in stream order)
    src = stored_source(csr, src, dtype, K, permuted);
    if (dtype == XR_F64) apply_dispatch<double>(csr, method, p, static_cast<const double *>(src), K, out);
    else apply_dispatch<float>(csr, method, p, static_cast<const float *>(src), K, out);
}

void csr_apply_dev(const xr_csr *csr, int method, double percentile, const void *src_dev, int dtype, int64_t K, double *out_dev) {
    apply_dev(csr, method, percentile, src_dev, dtype, K, out_dev);
}

__global__ void k_narrow_i64(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}
__global__ void k_widen_i32(const int32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void k_bincount(const int32_t *__restrict__ row, int64_t nnz, int32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nnz) atomicAdd(&count[row[i]], 1);
}

__global__ void k_gather_i32(const int32_t *__restrict__ table, const int32_t *__restrict__ idx, int64_t n,
                             int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = table[idx[i]];
}
__global__ void k_scatter_iota_i32(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inverse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inverse[perm[i]] = (int32_t)i;
}
__global__ void k_relabel_i32(const int32_t *__restrict__ table, int32_t *__restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = table[idx[i]];
}

// stored row r -> caller row row_order[r]
__global__ void k_unpermute_len(const int32_t *__restrict__ indptr, const int32_t *__restrict__ row_order, int64_t n,
                                int32_t *__restrict__ len) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) len[row_order[r]] = indptr[r + 1] - indptr[r];
}
// one wave per stored row
__global__ void k_unpermute_rows(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                 const double *__restrict__ data, const int32_t *__restrict__ row_order, int64_t n,
                                 const int32_t *__restrict__ c_indptr, int32_t *__restrict__ c_indices,
                                 double *__restrict__ c_data) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    const int s = indptr[r], e = indptr[r + 1], d = c_indptr[row_order[r]];
    for (int j = s + lane; j < e; j += 64) {
        c_indices[d + (j - s)] = indices[j];
        c_data[d + (j - s)] = data[j];
    }
}

static void set_long_rows(xr_csr *csr, const std::vector<int32_t> &longs) {
    csr->has_long = !longs.empty();
    if (!csr->has_long) return;
    const int32_t count = (int32_t)longs.size();
    csr->long_rows.alloc(longs.size());
    csr->n_long.alloc(1);
    h2d(csr->long_rows.get(), longs.data(), sizeof(int32_t) * longs.size());
    h2d(csr->n_long.get(), &count, sizeof(int32_t));
}

// ---- matrix-free apply of separable weights: one thread per target cell walks the c_y x c_x entries of its row
// (y-major, x-minor = the order of the materialised CSR row and of the reference's loop) and forms each weight
// w_y * w_x on the fly.  Neighbouring threads are neighbouring x-targets, whose x-lists are adjacent source columns:
// the gathers are coalesced whatever the coarsening ratio, so even rows of thousands of entries are reduced
// sequentially -- bit-identical to the reference order -- and the P entries of the product are never stored or read.
template <int METHOD, typename SRC, int KTILE, int CX>
__global__ void __launch_bounds__(AP_BLOCK)
k_apply_outer(const int32_t *__restrict__ ipy, const int32_t *__restrict__ sy, const double *__restrict__ wy,
              const int32_t *__restrict__ ipx, const int32_t *__restrict__ sx, const double *__restrict__ wx, int64_t nty,
              int64_t ntx, int64_t nsx, int64_t S, int rows_per_block, const SRC *__restrict__ source, int64_t K,
              double *__restrict__ out) {
    // blockIdx.x: 256 adjacent x-targets; blockIdx.y: a run of target rows (everything about a y-list is wave-uniform,
    // i.e. scalar loads); blockIdx.z: tile of KTILE source variables.  CX > 0: every x-list has at most CX entries
    // and lives in registers for all rows of the run; the CX x KTILE gathers of one source row are then independent
    // loads issued back to back (lanes past the end of their list re-read their last column and discard it).
    const int64_t it = (int64_t)blockIdx.x * AP_BLOCK + threadIdx.x;
    if (it >= ntx) return;
    const int64_t T = nty * ntx;
    const int x0 = ipx[it], x1 = ipx[it + 1];
    const int nx = x1 - x0;
    constexpr int CXR = CX > 0 ? CX : 1;
    int sxr[CXR];
    double wxr[CXR];
    if (CX > 0) {
#pragma unroll
        for (int c = 0; c < CXR; c++) {
            const int b = c < nx ? x0 + c : (nx > 0 ? x1 - 1 : 0);
            sxr[c] = nx > 0 ? sx[b] : 0;
            wxr[c] = nx > 0 ? wx[b] : 0.0;
        }
    }
    const int64_t k0 = (int64_t)blockIdx.z * KTILE;
    const int kn = (int)((K - k0) < KTILE ? (K - k0) : KTILE);
    const SRC *src = source + k0 * S;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t j1 = (j0 + rows_per_block) < nty ? (j0 + rows_per_block) : nty;
    for (int64_t jt = j0; jt < j1; jt++) {
        const int y0 = ipy[jt], y1 = ipy[jt + 1];
        double normsum = 0.0;
        if (METHOD == XR_GEOMETRIC_MEAN)
            for (int a = y0; a < y1; a++)
                for (int b = x0; b < x1; b++) normsum += wy[a] * wx[b];
        Red<METHOD> red[KTILE];
        for (int a = y0; a < y1; a++) {
            const int64_t rowbase = (int64_t)sy[a] * nsx;
            const double wa = wy[a];
            if (CX > 0) {
                double v[CXR][KTILE];
#pragma unroll
                for (int c = 0; c < CXR; c++)
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++)
                        v[c][kk] = ld_src(src, (int64_t)(kk < kn ? kk : 0) * S + rowbase + sxr[c]);
#pragma unroll
                for (int c = 0; c < CXR; c++) {
                    if (c < nx) {
                        const double w = wa * wxr[c];
#pragma unroll
                        for (int kk = 0; kk < KTILE; kk++) red[kk].add(v[c][kk], w, normsum);
                    }
                }
            } else {
                for (int b = x0; b < x1; b++) {
                    const int64_t col = rowbase + sx[b];
                    const double w = wa * wx[b];
#pragma unroll
                    for (int kk = 0; kk < KTILE; kk++)
                        if (kk < kn) red[kk].add(ld_src(src, (int64_t)kk * S + col), w, normsum);
                }
            }
        }
        const bool empty = y1 == y0 || nx == 0;
        const int64_t t = jt * ntx + it;
#pragma unroll
        for (int kk = 0; kk < KTILE; kk++) {
            if (kk < kn) {
                double r = NAN;
                if (!empty) {
                    r = red[kk].fin();
                    if (METHOD == XR_GEOMETRIC_MEAN && normsum == 0) r = NAN;
                }
                out[(k0 + kk) * T + t] = r;
            }
        }
    }
}

template <int METHOD, typename SRC, int KT, int CX>
static void launch_outer_kt(const xr_outer *o, const SRC *src, int64_t K, double *out) {
    const int rpb = (int)std::max<int64_t>(4, div_up(o->nty, 65535));
    const dim3 grid((unsigned)div_up(o->ntx, AP_BLOCK), (unsigned)div_up(o->nty, rpb), (unsigned)div_up(K, KT));
    XR_LAUNCH("apply_outer", (k_apply_outer<METHOD, SRC, KT, CX>), grid, dim3(AP_BLOCK), 0, o->ipy.get(), o->sy.get(),
              o->wy.get(), o->ipx.get(), o->sx.get(), o->wx.get(), o->nty, o->ntx, o->nsx, o->nsy * o->nsx, rpb, src, K,
              out);
}

template <int METHOD, typename SRC, int KT>
static void launch_outer_cx(const xr_outer *o, const SRC *src, int64_t K, double *out) {
    if (o->max_cx <= 2) launch_outer_kt<METHOD, SRC, KT, 2>(o, src, K, out);
    else if (o->max_cx <= 4) launch_outer_kt<METHOD, SRC, KT, 4>(o, src, K, out);
    else launch_outer_kt<METHOD, SRC, KT, 0>(o, src, K, out);
}

template <int METHOD, typename SRC>
static void launch_outer(const xr_outer *o, const SRC *src, int64_t K, double *out) {
    if (o->nty * o->ntx == 0 || K == 0) return;
    if (o->Py == 0 || o->Px == 0) { // no entries at all (e.g. an empty source grid): every row is empty
        fill_f64(out, NAN, K * o->nty * o->ntx);
        return;
    }
    XR_REQUIRE(div_up(K, 4) <= 65535, XR_ERR_LIMIT, "apply: too many source variables in one call (%lld)", (long long)K);
    if (K == 1) launch_outer_cx<METHOD, SRC, 1>(o, src, K, out);
    else if (K == 2) launch_outer_cx<METHOD, SRC, 2>(o, src, K, out);
    else launch_outer_cx<METHOD, SRC, 4>(o, src, K, out);
}

// ---- separable (rectilinear) weights: CSR of the outer product of two per-axis sparse matrices.
// Row (jt, it) of the product holds cy(jt) * cx(it) entries, y-major / x-minor, i.e. ascending in the
// column id sy * n_source_x + sx when both axis rows are ascending.  The entries of one jt form a
// contiguous slab of cy * Px entries (Px = nnz of the x axis) that starts at indptr_y[jt] * Px, and
// entry e of the slab belongs to the x-target that owns x-entry floor(e / cy): offsets have a closed
// form, so there is no sort and no scan (the reference argsorts the broadcast triplets,
// xugrid/regrid/structured.py:527).
__global__ __launch_bounds__(256) void
k_outer_indptr(const int32_t *__restrict__ ipy, const int32_t *__restrict__ ipx, int64_t nty, int64_t ntx, int64_t Px,
               int64_t nnz, int32_t *__restrict__ indptr, int32_t *__restrict__ long_rows,
               int32_t *__restrict__ n_long) {
    const int64_t T = nty * ntx;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) indptr[T] = (int32_t)nnz;
    if (t >= T) return;
    const int64_t jt = t / ntx, it = t - jt * ntx;
    const int64_t cy = ipy[jt + 1] - ipy[jt], cx = ipx[it + 1] - ipx[it];
    indptr[t] = (int32_t)((int64_t)ipy[jt] * Px + cy * ipx[it]);
    if (cy * cx > XR_APPLY_LONG_ROW) long_rows[atomicAdd(n_long, 1)] = (int32_t)t;
}

__global__ __launch_bounds__(256) void
k_outer_fill(const int32_t *__restrict__ ipy, const int32_t *__restrict__ sy, const double *__restrict__ wy,
             const int32_t *__restrict__ ipx, const int32_t *__restrict__ sx, const double *__restrict__ wx,
             const int32_t *__restrict__ tx_of_entry, int64_t nty, int64_t nsx, int64_t Px,
             int32_t *__restrict__ indices, double *__restrict__ data) {
    for (int64_t jt = blockIdx.y; jt < nty; jt += gridDim.y) {
        const int y0 = ipy[jt], cy = ipy[jt + 1] - y0;
        if (cy == 0) continue;
        const int64_t slab = (int64_t)y0 * Px, n = (int64_t)cy * Px;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
            const int it = tx_of_entry[e / cy];
            const int x0 = ipx[it], cx = ipx[it + 1] - x0;
            const int r = (int)(e - (int64_t)cy * x0);
            const int a = r / cx, b = r - a * cx;
            indices[slab + e] = (int32_t)((int64_t)sy[y0 + a] * nsx + sx[x0 + b]);
            data[slab + e] = wy[y0 + a] * wx[x0 + b];
        }
    }
}

static void upload_narrow(const int64_t *host, int64_t n, int32_t *dev) {
    if (n <= 0) return;
    DevBuf<int64_t> wide((size_t)n);
    h2d(wide.get(), host, sizeof(int64_t) * (size_t)n);
    XR_LAUNCH("narrow_i64", k_narrow_i64, dim3(div_up(n, 256)), dim3(256), 0, wide.get(), dev, n);
}

static void download_widen(const int32_t *dev, int64_t n, int64_t *host) {
    if (n <= 0) return;
    DevBuf<int64_t> wide((size_t)n);
    XR_LAUNCH("widen_i32", k_widen_i32, dim3(div_up(n, 256)), dim3(256), 0, dev, wide.get(), n);
    d2h(host, wide.get(), sizeof(int64_t) * (size_t)n);
}

} // namespace xr

using namespace xr;

static void check_axis(const char *name, const int64_t *indptr, const int64_t *source, const double *weight,
                       int64_t n_target, int64_t n_source) {
    XR_REQUIRE(indptr, XR_ERR_INVALID, "xr_csr_from_outer: NULL indptr (%s axis)", name);
    XR_REQUIRE(n_target >= 0 && n_source >= 0, XR_ERR_INVALID, "xr_csr_from_outer: negative size (%s axis)", name);
    XR_REQUIRE(indptr[0] == 0, XR_ERR_INVALID, "xr_csr_from_outer: indptr[0] != 0 (%s axis)", name);
    for (int64_t i = 0; i < n_target; i++)
        XR_REQUIRE(indptr[i + 1] >= indptr[i], XR_ERR_INVALID, "xr_csr_from_outer: indptr decreases at %lld (%s axis)",
                   (long long)i, name);
    const int64_t nnz = indptr[n_target];
    XR_REQUIRE(nnz == 0 || (source && weight), XR_ERR_INVALID, "xr_csr_from_outer: NULL entries (%s axis)", name);
    for (int64_t i = 0; i < nnz; i++)
        XR_REQUIRE(source[i] >= 0 && source[i] < n_source, XR_ERR_INVALID,
                   "xr_csr_from_outer: source index %lld outside [0,%lld) (%s axis)", (long long)source[i],
                   (long long)n_source, name);
}

static xr_outer *outer_create(const int64_t *indptr_y, const int64_t *source_y, const double *weight_y, int64_t n_target_y,
                              int64_t n_source_y, const int64_t *indptr_x, const int64_t *source_x, const double *weight_x,
                              int64_t n_target_x, int64_t n_source_x) {
    check_axis("y", indptr_y, source_y, weight_y, n_target_y, n_source_y);
    check_axis("x", indptr_x, source_x, weight_x, n_target_x, n_source_x);
    const int64_t Py = indptr_y[n_target_y], Px = indptr_x[n_target_x];
    const int64_t lim = ((int64_t)1 << 31) - 1;
    XR_REQUIRE(n_target_y == 0 || n_target_x < lim / n_target_y, XR_ERR_LIMIT,
               "separable weights: target grid exceeds the int32 index range");
    XR_REQUIRE(n_source_y == 0 || n_source_x <= lim / n_source_y, XR_ERR_LIMIT,
               "separable weights: source grid exceeds the int32 index range");
    xr_outer *o = new xr_outer();
    try {
        o->nty = n_target_y; o->nsy = n_source_y; o->ntx = n_target_x; o->nsx = n_source_x; o->Py = Py; o->Px = Px;
        o->ipy.alloc((size_t)n_target_y + 1); o->ipx.alloc((size_t)n_target_x + 1);
        o->sy.alloc((size_t)Py); o->sx.alloc((size_t)Px); o->tx.alloc((size_t)Px);
        o->wy.alloc((size_t)Py); o->wx.alloc((size_t)Px);
        upload_narrow(indptr_y, n_target_y + 1, o->ipy.get());
        upload_narrow(indptr_x, n_target_x + 1, o->ipx.get());
        upload_narrow(source_y, Py, o->sy.get());
        upload_narrow(source_x, Px, o->sx.get());
        h2d(o->wy.get(), weight_y, sizeof(double) * (size_t)Py);
        h2d(o->wx.get(), weight_x, sizeof(double) * (size_t)Px);
        std::vector<int32_t> owner((size_t)Px);
        for (int64_t it = 0; it < n_target_x; it++)
            for (int64_t q = indptr_x[it]; q < indptr_x[it + 1]; q++) owner[(size_t)q] = (int32_t)it;
        h2d(o->tx.get(), owner.data(), sizeof(int32_t) * (size_t)Px);
        for (int64_t j = 0; j < n_target_y; j++) o->max_cy = std::max(o->max_cy, indptr_y[j + 1] - indptr_y[j]);
        for (int64_t i = 0; i < n_target_x; i++) o->max_cx = std::max(o->max_cx, indptr_x[i + 1] - indptr_x[i]);
        stream_sync();
    } catch (...) {
        delete o;
        throw;
    }
    return o;
}

// the CSR of the outer product, assembled on the device (closed-form offsets: no sort, no scan)
static xr_csr *outer_materialise(const xr_outer *o) {
    const int64_t lim = ((int64_t)1 << 31) - 1;
    XR_REQUIRE(o->Py == 0 || o->Px < lim / o->Py, XR_ERR_LIMIT, "separable weights: %lld x %lld entries exceed the int32 range",
               (long long)o->Py, (long long)o->Px);
    const int64_t n = o->nty * o->ntx, m = o->nsy * o->nsx, nnz = o->Py * o->Px;
    xr_csr *csr = new xr_csr();
    try {
        csr->n = n; csr->m = m; csr->nnz = nnz;
        csr->indptr.alloc((size_t)n + 1);
        csr->indices.alloc((size_t)nnz);
        csr->data.alloc((size_t)nnz);
        csr->long_rows.alloc((size_t)(nnz / XR_APPLY_LONG_ROW + 1));
        csr->n_long.alloc(1);
        XR_HIP(hipMemsetAsync(csr->n_long.get(), 0, sizeof(int32_t), launch_stream()));
        XR_LAUNCH("outer_indptr", k_outer_indptr, dim3(div_up(n + 1, 256)), dim3(256), 0, o->ipy.get(), o->ipx.get(), o->nty,
                  o->ntx, o->Px, nnz, csr->indptr.get(), csr->long_rows.get(), csr->n_long.get());
        if (nnz > 0) {
            const int64_t gx = std::min<int64_t>(div_up(o->max_cy * o->Px, 256), 1 << 16);
            const int64_t gy = std::min<int64_t>(o->nty, 32768);
            XR_LAUNCH("outer_fill", k_outer_fill, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, o->ipy.get(), o->sy.get(),
                      o->wy.get(), o->ipx.get(), o->sx.get(), o->wx.get(), o->tx.get(), o->nty, o->nsx, o->Px,
                      csr->indices.get(), csr->data.get());
        }
        const int32_t nl = read_scalar(csr->n_long.get());
        csr->has_long = nl > 0;
    } catch (...) {
        delete csr;
        throw;
    }
    return csr;
}

// x-lists of at most 4 entries (similar resolutions, refinement, any coarsening along y only): matrix-free.  Longer
// x-lists make neighbouring lanes gather columns that lie a whole list apart; there the stored product with its
// cooperative long-row kernels is the faster engine (and the only one for mode / percentiles, which need a row's
// values side by side).  Products too large to store (>= 2^31 entries) stay matrix-free.
static bool outer_matrix_free(const xr_outer *o, int method) {
    const int64_t force = option(OPT_OUTER_APPLY); // test hook: 1 "free" | 2 "csr"
    const bool reducible = method != XR_MODE && method != XR_PERCENTILE;
    const bool fits = o->Py == 0 || o->Px < (((int64_t)1 << 31) - 1) / o->Py;
    if (!reducible) return false;
    if (!fits) return true;
    if (force == 1) return true;
    if (force == 2) return false;
    return o->max_cx <= 4;
}

template <typename SRC>
static void apply_outer_dispatch(xr_outer *o, int method, double p, const SRC *src, int64_t K, double *out) {
    if (!outer_matrix_free(o, method)) {
        if (!o->csr) o->csr = outer_materialise(o);
        apply_dispatch<SRC>(o->csr, method, p, src, K, out);
        return;
    }
    switch (method) {
    case XR_MEAN: launch_outer<XR_MEAN, SRC>(o, src, K, out); break;
    case XR_HARMONIC_MEAN: launch_outer<XR_HARMONIC_MEAN, SRC>(o, src, K, out); break;
    case XR_GEOMETRIC_MEAN: launch_outer<XR_GEOMETRIC_MEAN, SRC>(o, src, K, out); break;
    case XR_SUM: launch_outer<XR_SUM, SRC>(o, src, K, out); break;
    case XR_MINIMUM: launch_outer<XR_MINIMUM, SRC>(o, src, K, out); break;
    case XR_MAXIMUM: launch_outer<XR_MAXIMUM, SRC>(o, src, K, out); break;
    case XR_FIRST_ORDER_CONSERVATIVE: launch_outer<XR_FIRST_ORDER_CONSERVATIVE, SRC>(o, src, K, out); break;
    case XR_MAX_OVERLAP: launch_outer<XR_MAX_OVERLAP, SRC>(o, src, K, out); break;
    case XR_SELECT: launch_outer<XR_SELECT, SRC>(o, src, K, out); break;
    default: XR_REQUIRE(false, XR_ERR_INVALID, "unknown reducer id %d", method);
    }
}

static void apply_outer_dev(xr_outer *o, int method, double p, const void *src, int dtype, int64_t K, double *out) {
    XR_REQUIRE(dtype == XR_F64 || dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d", dtype);
    XR_REQUIRE(method >= 0 && method <= XR_SELECT, XR_ERR_INVALID, "unknown reducer id %d", method);
    if (dtype == XR_F64) apply_outer_dispatch<double>(o, method, p, static_cast<const double *>(src), K, out);
    else apply_outer_dispatch<float>(o, method, p, static_cast<const float *>(src), K, out);
}


extern "C" {

int xr_csr_info(const xr_csr *csr, int64_t *n, int64_t *m, int64_t *nnz) {
    XR_API_BEGIN
    XR_REQUIRE(csr, XR_ERR_INVALID, "xr_csr_info: NULL csr");
    if (n) *n = csr->n;
    if (m) *m = csr->m;
    if (nnz) *nnz = csr->nnz;
    XR_API_END
}

int xr_csr_download(const xr_csr *csr, double *data, int64_t *indices, int64_t *indptr) {
    XR_API_BEGIN
    XR_REQUIRE(csr, XR_ERR_INVALID, "xr_csr_download: NULL csr");
    const int32_t *d_indptr = csr->indptr.get(), *d_indices = csr->indices.get();
    const double *d_data = csr->data.get();
    DevBuf<int32_t> c_indptr, c_indices;
    DevBuf<double> c_data;
    if (csr->has_row_order && csr->n > 0) {
        // rows are stored in query order: rebuild the caller's row order on the device
        DevBuf<int32_t> len((size_t)csr->n);
        c_indptr.alloc((size_t)csr->n + 1);
        c_indices.alloc((size_t)csr->nnz);
        c_data.alloc((size_t)csr->nnz);
        XR_LAUNCH("unpermute_len", k_unpermute_len, dim3(div_up(csr->n, 256)), dim3(256), 0, csr->indptr.get(),
                  csr->row_order.get(), csr->n, len.get());
        exclusive_scan_i32(len.get(), c_indptr.get(), csr->n);
        XR_LAUNCH("unpermute_rows", k_unpermute_rows, dim3(div_up(csr->n * 64, 256)), dim3(256), 0, csr->indptr.get(),
                  csr->indices.get(), csr->data.get(), csr->row_order.get(), csr->n, c_indptr.get(), c_indices.get(),
                  c_data.get());
        d_indptr = c_indptr.get();
        d_indices = c_indices.get();
        d_data = c_data.get();
    }
    if (data && csr->nnz > 0) {
        d2h(data, d_data, sizeof(double) * (size_t)csr->nnz);
    }
    if (indices) {
        DevBuf<int32_t> caller_ids;
        if (csr->has_col_perm && csr->nnz > 0) { // stored column ids -> the caller's
            caller_ids.alloc((size_t)csr->nnz);
            XR_LAUNCH("col_ids", k_gather_i32, dim3(div_up(csr->nnz, 256)), dim3(256), 0, csr->col_of.get(), d_indices,
                      csr->nnz, caller_ids.get());
            d_indices = caller_ids.get();
        }
        download_widen(d_indices, csr->nnz, indices);
    }
    if (indptr) download_widen(d_indptr, csr->n + 1, indptr);
    stream_sync();
    XR_API_END
}

int xr_csr_upload(const double *data, const int64_t *indices, const int64_t *indptr, int64_t n, int64_t m,
                  int64_t nnz, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(out && indptr && (nnz == 0 || (data && indices)), XR_ERR_INVALID, "xr_csr_upload: NULL argument");
    XR_REQUIRE(n >= 0 && m >= 0 && nnz >= 0, XR_ERR_INVALID, "xr_csr_upload: negative sizes");
    XR_REQUIRE(n < ((int64_t)1 << 31) - 1 && m < ((int64_t)1 << 31) && nnz < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_csr_upload: matrix exceeds the int32 index range");
    XR_REQUIRE(indptr[0] == 0 && indptr[n] == nnz, XR_ERR_INVALID, "xr_csr_upload: indptr does not span [0, nnz]");
    for (int64_t i = 0; i < n; i++)
        XR_REQUIRE(indptr[i + 1] >= indptr[i], XR_ERR_INVALID,
                   "xr_csr_upload: indptr is not non-decreasing at row %lld", (long long)i);
    for (int64_t i = 0; i < nnz; i++)
        XR_REQUIRE(indices[i] >= 0 && indices[i] < m, XR_ERR_INVALID,
                   "xr_csr_upload: column index %lld outside [0,%lld)", (long long)indices[i], (long long)m);
    xr_csr *csr = new xr_csr();
    try {
        csr->n = n; csr->m = m; csr->nnz = nnz;
        csr->indptr.alloc((size_t)n + 1);
        csr->indices.alloc((size_t)nnz);
        csr->data.alloc((size_t)nnz);
        upload_narrow(indptr, n + 1, csr->indptr.get());
        upload_narrow(indices, nnz, csr->indices.get());
        h2d(csr->data.get(), data, sizeof(double) * (size_t)nnz);
        std::vector<int32_t> longs;
        for (int64_t i = 0; i < n; i++)
            if (indptr[i + 1] - indptr[i] > APPLY_LONG) longs.push_back((int32_t)i);
        set_long_rows(csr, longs);
        stream_sync();
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_csr_from_triplet(const int64_t *row, const int64_t *col, const double *data, int64_t nnz, int64_t n,
                        int64_t m, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(out && (nnz == 0 || (row && col && data)), XR_ERR_INVALID, "xr_csr_from_triplet: NULL argument");
    XR_REQUIRE(n >= 0 && m >= 0 && nnz >= 0, XR_ERR_INVALID, "xr_csr_from_triplet: negative sizes");
    XR_REQUIRE(n < ((int64_t)1 << 31) - 1 && m < ((int64_t)1 << 31) && nnz < ((int64_t)1 << 31), XR_ERR_LIMIT,
               "xr_csr_from_triplet: matrix exceeds the int32 index range");
    for (int64_t i = 0; i < nnz; i++) {
        XR_REQUIRE(row[i] >= 0 && row[i] < n && col[i] >= 0 && col[i] < m, XR_ERR_INVALID,
                   "xr_csr_from_triplet: entry %lld outside the %lld x %lld matrix", (long long)i, (long long)n,
                   (long long)m);
        XR_REQUIRE(i == 0 || row[i] >= row[i - 1], XR_ERR_INVALID,
                   "xr_csr_from_triplet: rows must be sorted (core/sparse.py:65)");
    }
    xr_csr *csr = new xr_csr();
    try {
        csr->n = n; csr->m = m; csr->nnz = nnz;
        csr->indptr.alloc((size_t)n + 1);
        csr->indices.alloc((size_t)nnz);
        csr->data.alloc((size_t)nnz);
        DevBuf<int32_t> row32((size_t)nnz), count((size_t)(n > 0 ? n : 1));
        upload_narrow(row, nnz, row32.get());
        upload_narrow(col, nnz, csr->indices.get());
        h2d(csr->data.get(), data, sizeof(double) * (size_t)nnz);
        XR_HIP(hipMemsetAsync(count.get(), 0, sizeof(int32_t) * (size_t)(n > 0 ? n : 1), launch_stream()));
        if (nnz > 0) XR_LAUNCH("bincount", k_bincount, dim3(div_up(nnz, 256)), dim3(256), 0, row32.get(), nnz, count.get());
        exclusive_scan_i32(count.get(), csr->indptr.get(), n);
        std::vector<int32_t> longs;
        for (int64_t i = 0, j = 0; i < nnz; i = j) { // rows are sorted: run lengths
            while (j < nnz && row[j] == row[i]) j++;
            if (j - i > APPLY_LONG) longs.push_back((int32_t)row[i]);
        }
        set_long_rows(csr, longs);
        stream_sync();
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_csr_from_outer(const int64_t *indptr_y, const int64_t *source_y, const double *weight_y, int64_t n_target_y,
                      int64_t n_source_y, const int64_t *indptr_x, const int64_t *source_x, const double *weight_x,
                      int64_t n_target_x, int64_t n_source_x, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(out, XR_ERR_INVALID, "xr_csr_from_outer: NULL argument");
    xr_outer *o = outer_create(indptr_y, source_y, weight_y, n_target_y, n_source_y, indptr_x, source_x, weight_x,
                               n_target_x, n_source_x);
    xr_csr *csr = nullptr;
    try {
        csr = outer_materialise(o);
    } catch (...) {
        delete o;
        throw;
    }
    delete o;
    *out = csr;
    XR_API_END
}

int xr_outer_create(const int64_t *indptr_y, const int64_t *source_y, const double *weight_y, int64_t n_target_y,
                    int64_t n_source_y, const int64_t *indptr_x, const int64_t *source_x, const double *weight_x,
                    int64_t n_target_x, int64_t n_source_x, xr_outer **out) {
    XR_API_BEGIN
    XR_REQUIRE(out, XR_ERR_INVALID, "xr_outer_create: NULL argument");
    *out = outer_create(indptr_y, source_y, weight_y, n_target_y, n_source_y, indptr_x, source_x, weight_x, n_target_x,
                        n_source_x);
    XR_API_END
}

int xr_outer_info(const xr_outer *o, int64_t *n, int64_t *m, int64_t *nnz) {
    XR_API_BEGIN
    XR_REQUIRE(o, XR_ERR_INVALID, "xr_outer_info: NULL handle");
    if (n) *n = o->nty * o->ntx;
    if (m) *m = o->nsy * o->nsx;
    if (nnz) *nnz = o->Py * o->Px;
    XR_API_END
}

int xr_outer_csr(xr_outer *o, const xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(o && out, XR_ERR_INVALID, "xr_outer_csr: NULL argument");
    if (!o->csr) o->csr = outer_materialise(o);
    *out = o->csr;
    XR_API_END
}

int xr_outer_destroy(xr_outer *o) {
    XR_API_BEGIN
    if (o) {
        stream_sync();
        delete o;
    }
    XR_API_END
}

int xr_apply_outer_dev(xr_outer *o, int method, double percentile, const void *source_dev, int source_dtype, int64_t K,
                       double *out_dev) {
    XR_API_BEGIN
    XR_REQUIRE(o && (source_dev || K == 0) && (out_dev || K == 0), XR_ERR_INVALID, "xr_apply_outer_dev: NULL argument");
    XR_REQUIRE(K >= 0, XR_ERR_INVALID, "xr_apply_outer_dev: negative K");
    apply_outer_dev(o, method, percentile, source_dev, source_dtype, K, out_dev);
    dev_call_done();
    XR_API_END
}

int xr_apply_outer(xr_outer *o, int method, double percentile, const void *source, int source_dtype, int64_t K, double *out) {
    XR_API_BEGIN
    XR_REQUIRE(o && (source || K == 0) && (out || K == 0), XR_ERR_INVALID, "xr_apply_outer: NULL argument");
    XR_REQUIRE(K >= 0, XR_ERR_INVALID, "xr_apply_outer: negative K");
    XR_REQUIRE(source_dtype == XR_F64 || source_dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d",
               source_dtype);
    const size_t esz = source_dtype == XR_F64 ? 8 : 4;
    const int64_t n = o->nty * o->ntx, m = o->nsy * o->nsx;
    const size_t per_k = (size_t)m * esz + (size_t)n * sizeof(double);
    const int64_t chunk_opt = option(OPT_APPLY_CHUNK_BYTES); // test hook
    const size_t budget = chunk_opt > 0 ? (size_t)chunk_opt : ((size_t)4 << 30);
    int64_t kchunk = per_k > 0 ? (int64_t)(budget / per_k) : K;
    if (kchunk < 1) kchunk = 1;
    if (kchunk > K) kchunk = K;
    DevBuf<char> src((size_t)kchunk * (size_t)m * esz);
    DevBuf<double> dst((size_t)kchunk * (size_t)n);
    for (int64_t k0 = 0; k0 < K; k0 += kchunk) {
        const int64_t kc = (K - k0) < kchunk ? (K - k0) : kchunk;
        h2d(src.get(), static_cast<const char *>(source) + (size_t)k0 * (size_t)m * esz, (size_t)kc * (size_t)m * esz);
        apply_outer_dev(o, method, percentile, src.get(), source_dtype, kc, dst.get());
        d2h(out + (size_t)k0 * (size_t)n, dst.get(), (size_t)kc * (size_t)n * sizeof(double));
        stream_sync();
    }
    XR_API_END
}

int xr_csr_set_row_keys(xr_csr *csr, const int64_t *keys, int64_t key_range) {
    XR_API_BEGIN
    XR_REQUIRE(csr && (keys || csr->n == 0), XR_ERR_INVALID, "xr_csr_set_row_keys: NULL argument");
    XR_REQUIRE(key_range >= 1 && key_range <= ((int64_t)1 << 24), XR_ERR_INVALID,
               "xr_csr_set_row_keys: key_range must be in [1, 2^24]");
    // With the output in stored order the stored order is part of what the caller has been told (xr_csr_row_order): new
    // keys would regroup the rows under every result that is read through the old permutation.
    XR_REQUIRE(!csr->output_stored, XR_ERR_INVALID,
               "xr_csr_set_row_keys: the matrix delivers its output in stored order -- its row order is frozen; call "
               "xr_csr_output_stored_order(csr, 0) first, set the keys, switch it on again and re-read xr_csr_row_order");
    for (int64_t i = 0; i < csr->n; i++)
        XR_REQUIRE(keys[i] >= 0 && keys[i] < key_range, XR_ERR_INVALID, "xr_csr_set_row_keys: key %lld outside [0,%lld)",
                   (long long)keys[i], (long long)key_range);
    if (csr->n > 0) {
        csr->tile_key.alloc((size_t)csr->n);
        if (csr->has_row_order) {
            // rows already stored in another order (built by xr_overlap, or tiled before): the keys are per CALLER row, the
            // regrouping works on stored rows -- key of stored row r = keys[row_order[r]]; the permutations compose
            DevBuf<int32_t> by_caller((size_t)csr->n);
            upload_narrow(keys, csr->n, by_caller.get());
            XR_LAUNCH("gather_keys", k_gather_i32, dim3(div_up(csr->n, 256)), dim3(256), 0, by_caller.get(),
                      csr->row_order.get(), csr->n, csr->tile_key.get());
        } else {
            upload_narrow(keys, csr->n, csr->tile_key.get());
        }
        stream_sync();
        csr->plan_ready = false;
        csr->tile_key_range = key_range;
        csr->has_tile_key = key_range > 1;
    }
    XR_API_END
}

int xr_csr_set_col_keys(xr_csr *csr, const int64_t *keys, int64_t key_range) {
    XR_API_BEGIN
    XR_REQUIRE(csr && (keys || csr->m == 0), XR_ERR_INVALID, "xr_csr_set_col_keys: NULL argument");
    XR_REQUIRE(key_range >= 1 && key_range <= ((int64_t)1 << 24), XR_ERR_INVALID,
               "xr_csr_set_col_keys: key_range must be in [1, 2^24]");
    XR_REQUIRE(!csr->has_col_perm, XR_ERR_INVALID, "xr_csr_set_col_keys: the columns of this matrix are renumbered already");
    for (int64_t i = 0; i < csr->m; i++)
        XR_REQUIRE(keys[i] >= 0 && keys[i] < key_range, XR_ERR_INVALID, "xr_csr_set_col_keys: key %lld outside [0,%lld)",
                   (long long)keys[i], (long long)key_range);
    const int64_t m = csr->m;
    if (m > 0 && key_range > 1) {
        // stable counting sort of the columns by key (the tiling kernels of the rows): col_of[new] = caller's column
        DevBuf<int32_t> key((size_t)m), hist((size_t)key_range + 1), start((size_t)key_range + 1), members((size_t)m),
            rank((size_t)m);
        upload_narrow(keys, m, key.get());
        fill_i32(hist.get(), 0, key_range + 1);
        XR_LAUNCH("bincount", k_bincount, dim3(div_up(m, 256)), dim3(256), 0, key.get(), m, hist.get());
        exclusive_scan_i32(hist.get(), start.get(), key_range);
        fill_i32(hist.get(), 0, key_range + 1);
        XR_LAUNCH("tile_scatter", k_tile_scatter, dim3(div_up(m, 256)), dim3(256), 0, key.get(), m, start.get(), hist.get(),
                  members.get());
        csr->col_of.alloc((size_t)m);
        XR_LAUNCH("tile_rank", k_tile_rank, dim3(div_up(m, 256)), dim3(256), 0, key.get(), start.get(), members.get(), m,
                  csr->col_of.get());
        XR_LAUNCH("col_inverse", k_scatter_iota_i32, dim3(div_up(m, 256)), dim3(256), 0, csr->col_of.get(), m, rank.get());
        if (csr->nnz > 0)
            XR_LAUNCH("col_relabel", k_relabel_i32, dim3(div_up(csr->nnz, 256)), dim3(256), 0, rank.get(),
                      csr->indices.get(), csr->nnz);
        stream_sync();
        csr->has_col_perm = true;
        csr->plan_ready = false;
    }
    XR_API_END
}

int xr_csr_col_order(const xr_csr *csr, int64_t *order_out) {
    XR_API_BEGIN
    XR_REQUIRE(csr && (order_out || csr->m == 0), XR_ERR_INVALID, "xr_csr_col_order: NULL argument");
    if (csr->has_col_perm) {
        download_widen(csr->col_of.get(), csr->m, order_out);
        stream_sync();
    } else {
        for (int64_t i = 0; i < csr->m; i++) order_out[i] = i;
    }
    XR_API_END
}

int xr_csr_expect_permuted(xr_csr *csr, int permuted) {
    XR_API_BEGIN
    XR_REQUIRE(csr, XR_ERR_INVALID, "xr_csr_expect_permuted: NULL argument");
    csr->source_permuted = permuted != 0;
    XR_API_END
}

static int prepare_for_apply(const xr_csr *csr, int64_t K);

// The row tiling of the many-variable apply is PART of the stored order.  It is normally deferred to the first apply with
// K >= 8; whoever is told the stored order (xr_csr_row_order) or asks for results in it (xr_csr_output_stored_order) gets
// it settled right away, whatever K the coming applies have -- otherwise a later apply would regroup the rows under a
// permutation the caller has already read, and every result after that would be silently mis-ordered.
static void settle_row_layout(const xr_csr *csr) {
    if (csr && csr->n > 0 && csr->has_tile_key) {
        ensure_tiled(csr);
        stream_sync();
    }
}

int xr_csr_output_stored_order(xr_csr *csr, int stored) {
    XR_API_BEGIN
    XR_REQUIRE(csr, XR_ERR_INVALID, "xr_csr_output_stored_order: NULL argument");
    if (stored) settle_row_layout(csr); // from here on the order is frozen (xr_csr_set_row_keys refuses)
    csr->output_stored = stored != 0;
    XR_API_END
}

int xr_csr_row_order(const xr_csr *csr, int64_t K_hint, int64_t *order_out) {
    XR_API_BEGIN
    (void)K_hint; // (kept in the signature; the layout is settled whatever K is)
    XR_REQUIRE(csr && (order_out || csr->n == 0), XR_ERR_INVALID, "xr_csr_row_order: NULL argument");
    settle_row_layout(csr);
    if (csr->has_row_order) {
        download_widen(csr->row_order.get(), csr->n, order_out);
        stream_sync();
    } else {
        for (int64_t i = 0; i < csr->n; i++) order_out[i] = i;
    }
    XR_API_END
}

int xr_csr_destroy(xr_csr *csr) {
    XR_API_BEGIN
    if (csr) {
        release_point();
        delete csr;
    }
    XR_API_END
}

// Everything an apply may build lazily inside the matrix (row tiling, apply plan) is built here, under the EXCLUSIVE
// scope; the apply proper then only reads the matrix and runs under the shared scope next to other threads' applies.
static int prepare_for_apply(const xr_csr *csr, int64_t K) {
    XR_API_BEGIN
    if (csr && csr->n > 0 && K >= PLAN_KT && option(OPT_APPLY_PLAN) != 0 && (csr->has_tile_key || !csr->plan_ready)) {
        ensure_tiled(csr);
        ensure_plan(csr);
        stream_sync();
    }
    XR_API_END
}

int xr_apply_csr_dev(const xr_csr *csr, int method, double percentile, const void *source_dev, int source_dtype,
                     int64_t K, double *out_dev) {
    {
        const int rc = prepare_for_apply(csr, K);
        if (rc != XR_OK) return rc;
    }
    XR_API_BEGIN_SHARED
    XR_REQUIRE(csr && (source_dev || csr->m == 0 || K == 0) && (out_dev || csr->n == 0 || K == 0), XR_ERR_INVALID,
               "xr_apply_csr_dev: NULL argument");
    XR_REQUIRE(K >= 0, XR_ERR_INVALID, "xr_apply_csr_dev: negative K");
    apply_dev(csr, method, percentile, source_dev, source_dtype, K, out_dev);
    dev_call_done();
    XR_API_END
}

int xr_apply_csr(const xr_csr *csr, int method, double percentile, const void *source, int source_dtype, int64_t K,
                 double *out) {
    {
        const int rc = prepare_for_apply(csr, K);
        if (rc != XR_OK) return rc;
    }
    XR_API_BEGIN_SHARED
    XR_REQUIRE(csr && (source || csr->m == 0 || K == 0) && (out || csr->n == 0 || K == 0), XR_ERR_INVALID,
               "xr_apply_csr: NULL argument");
    XR_REQUIRE(K >= 0, XR_ERR_INVALID, "xr_apply_csr: negative K");
    XR_REQUIRE(source_dtype == XR_F64 || source_dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d",
               source_dtype);
    const size_t esz = source_dtype == XR_F64 ? 8 : 4;
    // The stacked variables go through the device in chunks of at most ~4 GiB of staging (source + result), so any
    // K works whatever the size of HBM; every variable is independent, results do not depend on the chunking.
    const size_t per_k = (size_t)csr->m * esz + (size_t)csr->n * sizeof(double);
    const int64_t chunk_opt = option(OPT_APPLY_CHUNK_BYTES); // test hook
    const size_t budget = chunk_opt > 0 ? (size_t)chunk_opt : ((size_t)4 << 30);
    int64_t kchunk = per_k > 0 ? (int64_t)(budget / per_k) : K;
    if (kchunk < 1) kchunk = 1;
    if (kchunk > K) kchunk = K;
    DevBuf<char> src((size_t)kchunk * (size_t)csr->m * esz);
    DevBuf<double> dst((size_t)kchunk * (size_t)csr->n);
    for (int64_t k0 = 0; k0 < K; k0 += kchunk) {
        const int64_t kc = (K - k0) < kchunk ? (K - k0) : kchunk;
        const size_t n_src = (size_t)kc * (size_t)csr->m, n_out = (size_t)kc * (size_t)csr->n;
        h2d_big(src.get(), static_cast<const char *>(source) + (size_t)k0 * (size_t)csr->m * esz, n_src * esz);
        apply_dev(csr, method, percentile, src.get(), source_dtype, kc, dst.get());
        if (n_out > 0) d2h_big(out + (size_t)k0 * (size_t)csr->n, dst.get(), n_out * sizeof(double));
        stream_sync();
    }
    XR_API_END
}

int xr_apply_coo(const int64_t *row, const int64_t *col, int64_t nnz, int64_t T, const void *source, int source_dtype,
                 int64_t K, int64_t S, double *out) {
    XR_API_BEGIN
    XR_REQUIRE(nnz >= 0 && T >= 0 && K >= 0 && S >= 0, XR_ERR_INVALID, "xr_apply_coo: negative sizes");
    XR_REQUIRE(source_dtype == XR_F64 || source_dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d",
               source_dtype);
    XR_REQUIRE(T < ((int64_t)1 << 31) && S < ((int64_t)1 << 31) && K < 65536, XR_ERR_LIMIT, "xr_apply_coo: too large");
    for (int64_t i = 0; i < nnz; i++)
        XR_REQUIRE(row[i] >= 0 && row[i] < T && col[i] >= 0 && col[i] < S, XR_ERR_INVALID,
                   "xr_apply_coo: entry %lld out of range", (long long)i);
    const size_t esz = source_dtype == XR_F64 ? 8 : 4;
    const size_t n_src = (size_t)K * (size_t)S, n_out = (size_t)K * (size_t)T;
    DevBuf<char> src(n_src * esz);
    DevBuf<double> dst(n_out);
    DevBuf<int32_t> r32((size_t)nnz), c32((size_t)nnz);
    h2d(src.get(), source, n_src * esz);
    upload_narrow(row, nnz, r32.get());
    upload_narrow(col, nnz, c32.get());
    fill_f64(dst.get(), NAN, (int64_t)n_out);
    if (nnz > 0 && K > 0) {
        dim3 grid(div_up(nnz, 256), (unsigned)K);
        if (source_dtype == XR_F64)
            XR_LAUNCH("apply_coo", k_apply_coo<double>, grid, dim3(256), 0, r32.get(), c32.get(), nnz, T, S,
                      reinterpret_cast<const double *>(src.get()), dst.get());
        else
            XR_LAUNCH("apply_coo", k_apply_coo<float>, grid, dim3(256), 0, r32.get(), c32.get(), nnz, T, S,
                      reinterpret_cast<const float *>(src.get()), dst.get());
    }
    if (n_out > 0) {
        XR_HIP(hipMemcpyAsync(out, dst.get(), n_out * sizeof(double), hipMemcpyDeviceToHost, launch_stream()));
    }
    stream_sync();
    XR_API_END
}

int xr_partial_components(int method) { return partial_components(method); }
int xr_partial_combine_is_max(int method) { return partial_is_max(method) ? 1 : 0; }

} // extern "C"

// the partial state of every stored row (xr_apply_partial_dev; also enqueued right behind a weight build by
// xr_overlap_partial_dev, then gated like the early apply: csr->apply_gated)
void xr::csr_partial_dev(const xr_csr *csr, int method, const void *source_dev, int source_dtype, int64_t K, double *out_dev,
                         int rows_layout) {
    XR_REQUIRE(partial_components(method) > 0, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
    XR_REQUIRE(source_dtype == XR_F64 || source_dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d", source_dtype);
    XR_REQUIRE(K >= 0 && K < 65536, XR_ERR_LIMIT, "xr_apply_partial_dev: K out of range (tile the variables)");
    const int32_t *gate = csr->apply_gated ? csr->n_long.get() + 1 : (const int32_t *)nullptr;
    XR_REQUIRE(!gate || K == 1, XR_ERR_INVALID, "internal: a gated partial apply is one variable");
    if (csr->n > 0 && K > 0) {
        XR_REQUIRE(source_dev || csr->m == 0, XR_ERR_INVALID, "xr_apply_partial_dev: NULL source");
        DevBuf<char> permuted;
        source_dev = stored_source(csr, source_dev, source_dtype, K, permuted);
        constexpr int PKT = 8;
        bool long_done = false; // (the long rows went with the short ones)
        constexpr bool one_var = false; // (the one-variable-per-thread kernels serve K < 8 only)
        if (K >= PKT && !one_var && rows_layout != 0) {
            dim3 grid(div_up(csr->n, 128), (unsigned)div_up(K, PKT));
            const size_t rows_shmem = sizeof(double) * 128 * (size_t)(partial_components(method) * PKT + 1);
            if (source_dtype == XR_F64)
                XR_LAUNCH("apply_partial", (k_apply_partial_rows<double, PKT>), grid, dim3(128), rows_shmem, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                          static_cast<const double *>(source_dev), K, out_dev, csr->has_long);
            else
                XR_LAUNCH("apply_partial", (k_apply_partial_rows<float, PKT>), grid, dim3(128), rows_shmem, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                          static_cast<const float *>(source_dev), K, out_dev, csr->has_long);
        } else if (K >= PKT && !one_var) {
            dim3 grid(div_up(csr->n, 256), (unsigned)div_up(K, PKT));
            if (source_dtype == XR_F64)
                XR_LAUNCH("apply_partial", (k_apply_partial_kt<double, PKT>), grid, dim3(256), 0, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                          static_cast<const double *>(source_dev), K, out_dev, rows_layout != 0, csr->has_long);
            else
                XR_LAUNCH("apply_partial", (k_apply_partial_kt<float, PKT>), grid, dim3(256), 0, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                          static_cast<const float *>(source_dev), K, out_dev, rows_layout != 0, csr->has_long);
        } else if (K == 1 && !one_var) {
            // one variable: the wave-window kernel, one specialisation per reducer; the long rows in its first blocks
            const int n_long_blocks = csr->has_long ? engine().num_cu / 2 : 0;
            dim3 grid((unsigned)(div_up(csr->n, AP_BLOCK) + n_long_blocks));
            long_done = true;
#define XR_PARTIAL_W1(M)                                                                                                            \
    case M:                                                                                                                         \
        if (source_dtype == XR_F64)                                                                                                 \
            XR_LAUNCH("apply_partial", (k_apply_partial_w1<M, double>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),                 \
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,                                       \
                      static_cast<const double *>(source_dev), out_dev, rows_layout != 0, csr->has_long, gate,                      \
                      csr->long_rows.get(), csr->n_long.get(), n_long_blocks);                                                      \
        else                                                                                                                        \
            XR_LAUNCH("apply_partial", (k_apply_partial_w1<M, float>), grid, dim3(AP_BLOCK), 0, csr->indptr.get(),                  \
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,                                       \
                      static_cast<const float *>(source_dev), out_dev, rows_layout != 0, csr->has_long, gate,                       \
                      csr->long_rows.get(), csr->n_long.get(), n_long_blocks);                                                      \
        break;
            switch (method) {
                XR_PARTIAL_W1(XR_MEAN)
                XR_PARTIAL_W1(XR_FIRST_ORDER_CONSERVATIVE)
                XR_PARTIAL_W1(XR_SUM)
                XR_PARTIAL_W1(XR_HARMONIC_MEAN)
                XR_PARTIAL_W1(XR_GEOMETRIC_MEAN)
                XR_PARTIAL_W1(XR_MINIMUM)
                XR_PARTIAL_W1(XR_MAXIMUM)
            default: XR_REQUIRE(false, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
            }
#undef XR_PARTIAL_W1
        } else {
        dim3 grid(div_up(csr->n, 256), (unsigned)K);
        if (source_dtype == XR_F64)
            XR_LAUNCH("apply_partial", k_apply_partial<double>, grid, dim3(256), 0, method, csr->indptr.get(),
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                      static_cast<const double *>(source_dev), K, out_dev, rows_layout != 0, csr->has_long);
        else
            XR_LAUNCH("apply_partial", k_apply_partial<float>, grid, dim3(256), 0, method, csr->indptr.get(),
                      csr->indices.get(), csr->data.get(), row_order_of(csr), csr->n, csr->m,
                      static_cast<const float *>(source_dev), K, out_dev, rows_layout != 0, csr->has_long);
        }
        if (csr->has_long && !long_done) {
            dim3 lgrid(64, (unsigned)K);
            if (source_dtype == XR_F64)
                XR_LAUNCH("apply_partial_long", k_apply_partial_long<double>, lgrid, dim3(256), 0, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(),
                          csr->n, csr->m, static_cast<const double *>(source_dev), K, out_dev, rows_layout != 0, gate);
            else
                XR_LAUNCH("apply_partial_long", k_apply_partial_long<float>, lgrid, dim3(256), 0, method, csr->indptr.get(),
                          csr->indices.get(), csr->data.get(), row_order_of(csr), csr->long_rows.get(), csr->n_long.get(),
                          csr->n, csr->m, static_cast<const float *>(source_dev), K, out_dev, rows_layout != 0, gate);
        }
    }
}

extern "C" {

int xr_apply_partial_dev(const xr_csr *csr, int method, const void *source_dev, int source_dtype, int64_t K,
                         double *out_dev, int rows_layout) {
    XR_API_BEGIN
    XR_REQUIRE(csr && out_dev, XR_ERR_INVALID, "xr_apply_partial_dev: NULL argument");
    csr_partial_dev(csr, method, source_dev, source_dtype, K, out_dev, rows_layout);
    dev_call_done();
    XR_API_END
}

int xr_partial_fill_identity_dev(int method, double *planes_dev, int64_t K, int64_t T) {
    XR_API_BEGIN
    XR_REQUIRE(partial_components(method) > 0, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
    XR_REQUIRE(K >= 0 && T >= 0, XR_ERR_INVALID, "xr_partial_fill_identity_dev: negative size");
    if (K * T > 0) {
        XR_REQUIRE(planes_dev, XR_ERR_INVALID, "xr_partial_fill_identity_dev: NULL argument");
        XR_LAUNCH("partial_identity", k_partial_identity, dim3(div_up(K * T * partial_components(method), 256)), dim3(256), 0,
                  method, planes_dev, K * T);
    }
    dev_call_done();
    XR_API_END
}

int xr_finalize_partial_dev(int method, const double *planes_dev, int64_t K, int64_t T, double *out_dev) {
    XR_API_BEGIN
    XR_REQUIRE(partial_components(method) > 0, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
    XR_REQUIRE(K >= 0 && T >= 0, XR_ERR_INVALID, "xr_finalize_partial_dev: negative size");
    if (K * T > 0) {
        XR_REQUIRE(planes_dev && out_dev, XR_ERR_INVALID, "xr_finalize_partial_dev: NULL argument");
        XR_LAUNCH("finalize_partial", k_finalize_partial, dim3(div_up(K * T, 256)), dim3(256), 0, method, planes_dev, K * T,
                  out_dev);
    }
    dev_call_done();
    XR_API_END
}

int xr_reduce_partial_rows_dev(int method, const double *rows_dev, const int64_t *indptr_dev, const int64_t *order_dev,
                               int64_t n_targets, int64_t K, double *out_dev) {
    XR_API_BEGIN
    XR_REQUIRE(partial_components(method) > 0, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
    XR_REQUIRE(n_targets >= 0 && K >= 0, XR_ERR_INVALID, "xr_reduce_partial_rows_dev: negative size");
    if (n_targets * K > 0) {
        XR_REQUIRE(indptr_dev && out_dev, XR_ERR_INVALID, "xr_reduce_partial_rows_dev: NULL argument");
        constexpr bool plain = false;
        if (K == 1 && !plain && (reinterpret_cast<uintptr_t>(rows_dev) & 15) == 0) {
#define XR_REDUCE_K1(M)                                                                                                             \
    case M:                                                                                                                         \
        XR_LAUNCH("reduce_partial_rows", k_reduce_partial_rows_k1<M>, dim3(div_up(n_targets, 256)), dim3(256), 0, rows_dev,         \
                  indptr_dev, order_dev, n_targets, out_dev);                                                                       \
        break;
            switch (method) {
                XR_REDUCE_K1(XR_MEAN)
                XR_REDUCE_K1(XR_FIRST_ORDER_CONSERVATIVE)
                XR_REDUCE_K1(XR_SUM)
                XR_REDUCE_K1(XR_HARMONIC_MEAN)
                XR_REDUCE_K1(XR_GEOMETRIC_MEAN)
                XR_REDUCE_K1(XR_MINIMUM)
                XR_REDUCE_K1(XR_MAXIMUM)
            default: XR_REQUIRE(false, XR_ERR_INVALID, "reducer %d does not decompose over source shards", method);
            }
#undef XR_REDUCE_K1
        } else if (K >= 8 && !plain)
            XR_LAUNCH("reduce_partial_rows", k_reduce_partial_rows_t, dim3(div_up(n_targets, 64), (unsigned)std::min<int64_t>(div_up(K, 32), 64)),
                      dim3(256), 0, method, rows_dev, indptr_dev, order_dev, n_targets, K, out_dev);
        else
            XR_LAUNCH("reduce_partial_rows", k_reduce_partial_rows, dim3(div_up(n_targets * K, 256)), dim3(256), 0, method,
                      rows_dev, indptr_dev, order_dev, n_targets, K, out_dev);
    }
    dev_call_done();
    XR_API_END
}

} // extern "C"
