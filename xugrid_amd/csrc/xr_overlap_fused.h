// xr_overlap_fused.h -- triangle x triangle pairs without the host in the loop (included by xr_overlap.hip; reference
// seam: CellTree2d.intersect_faces + weights /= area + MatrixCSR.from_triplet, xugrid/regrid/unstructured.py:109-135,
// xugrid/regrid/regridder.py:433-435).
//
// The general kernel chain search -> [host: C] -> clip (+ row counts by global atomics) -> scan (2 launches) -> [host: nnz]
// -> row_fill needs the host twice.  For triangle meshes:
//   k_search / k_search_big   as before, big faces into their own queue
//   k_clip_tri_queue          persistent clip over the regular pair queue; its length is read from device memory, so
//                             the launch does not wait for the host; chunks software-pipelined
//   k_assemble_scan           the search leaves the number of regular faces per block of 256 target faces, the clip the
//                             number of surviving pairs per block (one atomic per wave): ONE small block scans the few
//                             thousand (rows, entries) pairs into every block's first stored row and CSR base
//   k_assemble                the block that owns 256 consecutive target faces counts its survivors per row in LDS, ranks
//                             every survivor among its row and writes the final CSR at its base.  No row counters in
//                             HBM.  (xr_csr::row_order maps stored rows to the caller's faces.)  Up to 8192 blocks every
//                             block sums the counts of the blocks in front of it itself instead of k_assemble_scan
//   big faces (hull slivers, > SLOTS hits; 0.1 % of the faces, but their block-per-face kernels are chains of dependent
//   phases: 15 % of the step when run in line) on a SIDE STREAM, forked behind k_search: k_search_big -> k_clip_tri_queue on
//   their own pair queue -> k_row_fill_long (rows in face order as ranked by k_search_big; scans their lengths itself) into a small
//   CSR of their own; joined behind k_assemble, k_place_big copies those rows BEHIND the regular ones.
//   [host: sizes + error bits: ONE round trip for the whole weight matrix]
// Measured and discarded on the way (MI355X, benchmark pair): a fully fused search + clip + assembly block (pairs never
// leave LDS): correct, but the grid walk is a chain of dependent loads that needs ~7 waves per SIMD, the LDS staging
// allows 2 -- 0.68 ms against 0.31 ms for separate kernels; clip + assembly in one kernel: the clip is FP64-issue bound
// and needs 5 waves per SIMD, the staging leaves 3 -- 0.33 ms against 0.25 ms (and with the look-back inside, 1 ... 16
// clip rounds per block make every generation of resident blocks last as long as its slowest member); rows of the
// assembly placed by ONE returning 64-bit atomic per block instead of the look-back: 4000 atomics on one address cost
// 30 us (and a "last block done" counter another 27 us).
#pragma once

namespace xr {

static constexpr int FB = 256;            // faces (= threads) per block: the block granularity of k_search's queue stretches
static constexpr int FWAVES = FB / 64;

struct FusedCounters {                    // device words, zeroed before the launch
    int32_t n_apply_long;                 // rows with more than XR_APPLY_LONG_ROW entries
    int32_t error;                        // bit 0: clip buffer overflow, bit 2: CSR capacity
    int32_t p_regular;                    // entries of all regular rows (k_assemble)
    int32_t rows_regular;                 // regular rows
    int32_t p_big;                        // entries of the big faces' rows (k_big_order)
    int32_t pad0;                         // (unused)
    int32_t max_row;                      // entries of the longest row (atomicMax)
    int32_t pad2;
};
static_assert(offsetof(FusedCounters, error) == 4 && offsetof(FusedCounters, p_regular) == 8 && offsetof(FusedCounters, rows_regular) == 12 &&
                  offsetof(FusedCounters, p_big) == 16 && offsetof(FusedCounters, max_row) == 24,
              "k_publish_all addresses the fields by word");

// block-wide exclusive scan of one int per thread (FB threads); returns the exclusive value, total in *total
__device__ __forceinline__ int block_excl_scan(int v, int *sh_wave /*[FWAVES]*/, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    __syncthreads(); // (sh_wave may still be read from an earlier scan)
    if (lane == 63) sh_wave[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FWAVES; w++) {
        if (w < wave) woff += sh_wave[w];
        tot += sh_wave[w];
    }
    *total = tot;
    return woff + incl - v;
}

// Persistent triangle x triangle clip over the regular pair queue: the number of pairs is read from device memory
// (the search's queue cursor), so the launch needs no host round trip; every block walks chunks of 256 pairs.  Chunks are dealt so that every XCD works on one contiguous eighth of the queue (the order
// k_search wrote it in: the vertex blocks are in that XCD's L2).
// COUNT: which survivor counts the kernel keeps (two instantiations: both paths in one kernel cost registers -> scratch)
//   0 none, 1 per block of FB target faces (regular queue: blk_surv), 2 per target face (big faces' queue: nnz_row)
//   SOA: the LDS columns slot-major (slot s of lane l at s * BLOCK + l: every 16-byte access of a wave -- static or
//   dynamic slot -- falls on the banks of its lane alone) instead of lane-major (7 consecutive slots per lane: the dynamic-index
//   reads of a stage conflict, 18 % of the LDS cycles, round-3 PMC)
// KIND 1 (round 5): dense meshes of up to 4 nodes per face on either side (quadrilaterals, mixed triangle / quadrilateral
// meshes with fill values -- the usual D-Flow FM mesh, a raster paired with a triangle mesh): the register / LDS clip of
// k_clip_small<8> (the oracle's arithmetic in the oracle's order) inside the same persistent loop, so that these pairs take
// the one-round-trip pipeline too.  q_len / s_len / q_m / s_m are only read then.
// (waves per SIMD: the persistent grid is sized for five resident blocks per CU -- four for KIND 1 -- and a single register beyond
// 96 would leave only four: the block that does not fit runs BEHIND the others, +40 % on the kernel -- measured when one crept in)
template <int BLOCK, int COUNT, bool SOA = false, int KIND = 0>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(KIND == 0 ? 5 : 4)))
k_clip_tri_queue(const double *__restrict__ q_fxy, const double *__restrict__ rec_fxy,
                 const int32_t *__restrict__ rec_face, const int32_t *cand_tgt,
                 const int32_t *cand_src, const int32_t *__restrict__ n_cand_dev, int64_t capacity,
                 double *__restrict__ cand_area, int32_t *__restrict__ cand_sid, int32_t *__restrict__ error_bits,
                 int32_t *__restrict__ nnz_row /* optional: survivors per target face, counted by atomics */,
                 const int32_t *__restrict__ skip_if /* optional: nothing is done when this device word is > 0 */,
                 int32_t *__restrict__ blk_surv = nullptr /* optional: survivors per BLOCK of 256 target faces (k_assemble_scan) */,
                 double dust = 0.0 /* areas up to this are confirmed by the reference's pre-clip tests (xr_overlap.hip: confirm_dust) */,
                 const uint8_t *__restrict__ q_len = nullptr, const uint8_t *__restrict__ s_len = nullptr, int q_m = 3, int s_m = 3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CS = SOA ? BLOCK : 1;
    constexpr int SMALL_MAXV = 8;
    double2 *col = reinterpret_cast<double2 *>(smem) + (KIND == 1 || SOA ? threadIdx.x : threadIdx.x * (TRI_MAXV + 1)); // the lane's slots
    __shared__ uint2 sh_lut[KIND == 1 ? 1 : TRI_LUT];
    if (skip_if) __builtin_amdgcn_s_setprio(3); // (the big faces' queue, on the side stream: issue priority over the main clip)
    if (skip_if && *skip_if > 0) return; // (big faces that did not fit their queue: the host redoes everything)
    if (KIND == 0) {
        tri_lut_init(sh_lut);
        __syncthreads();
    }
    // capacity > 0: ONE dense queue of *n_cand_dev pairs, its chunks dealt so that every XCD works on a contiguous eighth;
    // capacity < 0: EIGHT dense regions of -capacity pairs each (region x filled by the search blocks of XCD x, n_cand_dev[x] pairs):
    // slot j is chunk j / 8 of region j mod 8 -- the same dealing, the regions' own lengths
    const bool regions = capacity < 0;
    const int64_t region_cap = regions ? -capacity : 0;
    int64_t n_cand = 0, per_xcd;
    auto region_len = [&](int x) -> int64_t { // (uniform: a scalar load; a region that overflowed is redone by the host)
        const int64_t n = n_cand_dev[x * QCUR_STRIDE];
        return n < region_cap ? n : region_cap;
    };
    if (regions) {
        int64_t longest = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) longest = region_len(x) > longest ? region_len(x) : longest;
        per_xcd = (longest + BLOCK - 1) / BLOCK;
    } else {
        n_cand = *n_cand_dev < capacity ? *n_cand_dev : capacity; // (a queue that overflowed is redone by the host)
        per_xcd = ((n_cand + BLOCK - 1) / BLOCK + 7) >> 3;
    }
    const int64_t n_slots = per_xcd * 8; // chunk slots incl. the empty ones of the shorter eighths
    // j-th slot -> index of its first pair; *end = one behind the last pair the slot's stretch may hold
    auto slot_first = [&](int64_t j, int64_t *end) -> int64_t {
        if (regions) {
            const int x = (int)(j & 7);
            *end = (int64_t)x * region_cap + region_len(x);
            return (int64_t)x * region_cap + (j >> 3) * BLOCK;
        }
        *end = n_cand;
        return ((j & 7) * per_xcd + (j >> 3)) * BLOCK;
    };
    const double2 *tfx = reinterpret_cast<const double2 *>(q_fxy);
    const double2 *sfx = reinterpret_cast<const double2 *>(rec_fxy);
    const int tid = threadIdx.x;
    // the pair indices of the NEXT chunk are fetched while a chunk is clipped (its vertex gathers then start at once;
    // prefetching the vertex blocks as well costs 36 more registers = a wave per SIMD, and the clip is issue bound)
    int n_tq = 0, n_s = 0;
    auto load_idx = [&](int64_t slot, int &tq, int &s) {
        tq = 0;
        s = 0;
        if (slot < n_slots) {
            int64_t end;
            const int64_t c = slot_first(slot, &end) + tid;
            if (c < end) {
                tq = cand_tgt[c];
                s = cand_src[c];
            }
        }
    };
    const int64_t stride = gridDim.x;
    int64_t slot = blockIdx.x;
    load_idx(slot, n_tq, n_s);
    bool overflow = false;
    for (; slot < n_slots; slot += stride) {
        int64_t c_end;
        const int64_t c = slot_first(slot, &c_end) + tid;
        const bool active = c < c_end;
        int sid = 0;
        const int cur_tq = n_tq;
        double area = 0.0;
        if (KIND == 1) {
            const int cur_s = n_s;
            int nt = 0, ns = 0;
            if (active) {
                sid = rec_face[cur_s];
                nt = q_len[cur_tq];
                ns = s_len[cur_s];
            }
            load_idx(slot + stride, n_tq, n_s);
            if (active) {
                const double2 *tf = tfx + (int64_t)cur_tq * q_m;
                const double2 *sf = sfx + (int64_t)cur_s * s_m;
                area = clip_small_pair<SMALL_MAXV, BLOCK, false>(tf, nt, reinterpret_cast<const double *>(sf), ns, col);
                if (area > 0 && area <= dust && !pair_passes_box_and_sat(tf, nt, sf, ns)) area = 0.0;
            }
        } else {
            P2 tv[3] = {{0, 0}, {0, 0}, {0, 0}}, sv[3] = {{0, 0}, {0, 0}, {0, 0}};
            if (active) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const double2 a = tfx[(int64_t)n_tq * 3 + j], b2 = sfx[(int64_t)n_s * 3 + j];
                    tv[j] = P2{a.x, a.y};
                    sv[j] = P2{b2.x, b2.y};
                }
                sid = rec_face[n_s];
            }
            load_idx(slot + stride, n_tq, n_s);
            // (rounding dust of a pair the reference's pre-clip tests reject: the few lanes concerned fetch their vertices again)
            area = tri_clip_area<CS>(tv, sv, col, sh_lut, active);
            const bool suspicious = active && area > 0 && area <= dust;
            if (__any(suspicious)) {
                if (suspicious && !pair_passes_box_and_sat(tfx + (int64_t)cur_tq * 3, 3, sfx + (int64_t)cand_src[c] * 3, 3)) area = 0.0;
            }
        }
        const int tq_now = active ? cur_tq : -1;
        if (active) {
            overflow = overflow || area == TRI_AREA_OVERFLOW;
            cand_area[c] = area;
            cand_sid[c] = area > 0 ? sid : 0x7fffffff;
        }
        if (COUNT == 1) {
            // survivors per block of FB target faces.  The pairs of a block are one stretch of the queue (1700 on average):
            // most waves see a single block -- one atomic by lane 0; the others add per run of equal block ids
            const int lane = tid & 63;
            const int blk_now = active ? (cur_tq >> 8) : -1;
            const unsigned long long surv = __ballot(active && area > 0);
            const int blk_first = __builtin_amdgcn_readfirstlane(blk_now), blk_last = __shfl(blk_now, 63, 64);
            if (blk_first == blk_last) { // (uniform)
                if (lane == 0 && blk_first >= 0 && surv) atomicAdd(&blk_surv[blk_first], __popcll(surv));
            } else {
                const int blk_prev = __shfl_up(blk_now, 1, 64);
                const bool head = lane == 0 || blk_prev != blk_now;
                const unsigned long long heads = __ballot(head);
                if (head && blk_now >= 0) {
                    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
                    const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
                    unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
                    run &= ~((1ull << lane) - 1);
                    const int n = __popcll(surv & run);
                    if (n > 0) atomicAdd(&blk_surv[blk_now], n);
                }
            }
        }
        if (COUNT == 2) {
            // pairs of one target face are contiguous: the head lane of each run of equal faces adds the run's survivors
            const int lane = tid & 63;
            const int t_prev = __shfl_up(tq_now, 1, 64);
            const bool head = lane == 0 || t_prev != tq_now;
            const unsigned long long heads = __ballot(head);
            const unsigned long long surv = __ballot(active && area > 0);
            if (head && tq_now >= 0) {
                const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
                const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
                unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
                run &= ~((1ull << lane) - 1);
                const int n = __popcll(surv & run);
                if (n > 0) atomicAdd(&nnz_row[tq_now], n);
            }
        }
    }
    if (overflow) atomicOr(error_bits, 1);
}

// CSR assembly for the regular faces: the block that owns 256 consecutive target faces stages the source ids of its
// (k_assemble_scan: the blocks' first stored row and CSR base from counts the search and the clip left per block -- one
// block scans the few thousand (rows, entries) pairs in the order the assembly assigns stored rows, i.e. hardware block
// order; replaces the look-back chain inside k_assemble when those counts exist)
__global__ void __launch_bounds__(1024)
k_assemble_scan(const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ blk_surv, int64_t n_blocks, int n_chain,
                bool remap, int32_t *__restrict__ base_rows, int32_t *__restrict__ base_nnz, FusedCounters *counters,
                int32_t *__restrict__ indptr) {
    __shared__ long long sh_w[16];
    __shared__ long long sh_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sh_carry = 0;
    __syncthreads();
    const int64_t per_xcd = (n_blocks + 7) >> 3;
    for (int b0 = 0; b0 < n_chain; b0 += 1024) {
        const int b = b0 + tid;
        long long v = 0;
        if (b < n_chain) {
            const int64_t lb = remap ? (int64_t)(b & 7) * per_xcd + (b >> 3) : b;
            if (lb < n_blocks) v = ((long long)blk_rows[lb] << 31) | (long long)blk_surv[lb];
        }
        long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sh_w[wave] = incl;
        __syncthreads();
        long long woff = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) woff += sh_w[w];
            tot += sh_w[w];
        }
        const long long excl = sh_carry + woff + incl - v;
        if (b < n_chain) {
            base_nnz[b] = (int32_t)(excl & 0x7fffffffll);
            base_rows[b] = (int32_t)(excl >> 31);
        }
        __syncthreads();
        if (tid == 0) sh_carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        const long long entries = sh_carry & 0x7fffffffll, rows = sh_carry >> 31;
        counters->p_regular = (int32_t)entries; // entries of all regular rows
        counters->rows_regular = (int32_t)rows;
        indptr[rows] = (int32_t)entries;        // (= indptr[T] when there are no big faces)
    }
}

// stretch of the pair queue in LDS ("dead" for area <= 0), counts the survivors per face (LDS atomics), reserves its
// rows and entries with ONE atomic, ranks every survivor among its row and writes it to its final CSR position.
__global__ void __launch_bounds__(FB, 2)
k_assemble(const double *__restrict__ q_bbox, const int32_t *__restrict__ q_perm, int64_t n_query,
           const int32_t *__restrict__ cand_tgt, const int32_t *__restrict__ cand_off,
           const int32_t *__restrict__ cand_count, const int2 *__restrict__ block_seg,
           const uint8_t *__restrict__ is_big, const double *__restrict__ cand_area,
           const int32_t *__restrict__ cand_sid, const double *__restrict__ src_area, bool relative, MortonParams tile,
           int32_t *__restrict__ tile_key, FusedCounters *__restrict__ counters,
           int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data, int32_t *__restrict__ row_order,
           int32_t *__restrict__ apply_long_rows, int64_t csr_capacity, bool remap,
           const int32_t *__restrict__ base_rows = nullptr, const int32_t *__restrict__ base_nnz = nullptr,
           const int32_t *__restrict__ blk_rows = nullptr, const int32_t *__restrict__ blk_surv = nullptr) {
    __shared__ int32_t sh_stage[SLOTS * FB];
    __shared__ int32_t sh_nnz[FB];     // survivors of the face (LDS atomics)
    __shared__ uint16_t sh_lo[FB];     // offset of the face's pairs inside the stretch
    __shared__ uint16_t sh_rowoff[FB]; // CSR offset of the face's row inside the block
    __shared__ uint8_t sh_cnt[FB];     // pairs of the face (0 for big faces)
    __shared__ int32_t sh_wave[FWAVES];
    __shared__ long long sh_base;
    const int tid = threadIdx.x;
    const int64_t n_blocks = (n_query + FB - 1) / FB;
    const int64_t lb = xcd_block(n_blocks, remap); // (blocks beyond n_blocks carry no faces but stay in the chain)
    const bool valid_block = lb < n_blocks;
    sh_nnz[tid] = 0;
    const int64_t t0 = lb * FB;
    const int64_t t = t0 + tid;
    const bool in_range = valid_block && t < n_query;
    const int2 seg = valid_block ? block_seg[lb] : make_int2(0, 0);
    const int total = seg.y;
    bool regular = false;
    int my_cnt = 0;
    if (in_range) {
        regular = !is_big[t];
        my_cnt = regular ? cand_count[t] : 0;
        sh_lo[tid] = (uint16_t)(regular ? cand_off[t] - seg.x : 0);
    } else {
        sh_lo[tid] = 0;
    }
    sh_cnt[tid] = (uint8_t)my_cnt;
    __syncthreads();
    // (the kernel is latency bound -- 84 % of its wave cycles wait, PMC -- so both entry loops keep the loads of four
    // steps in flight instead of one)
    for (int i0 = tid; i0 < total; i0 += 4 * FB) {
        int s[4], row[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * FB;
            const int64_t c = (int64_t)seg.x + (i < total ? i : total - 1);
            s[u] = cand_sid[c];
            row[u] = cand_tgt[c] - (int)t0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * FB;
            if (i < total) {
                sh_stage[i] = s[u];
                if (s[u] != 0x7fffffff) atomicAdd(&sh_nnz[row[u]], 1);
            }
        }
    }
    __syncthreads();
    const int my_nnz = regular ? sh_nnz[tid] : 0;
    int block_nnz = 0, block_rows = 0;
    const int rowoff = block_excl_scan(my_nnz, sh_wave, &block_nnz);
    const int rowidx = block_excl_scan(regular ? 1 : 0, sh_wave, &block_rows);
    sh_rowoff[tid] = (uint16_t)rowoff;
    const int b = (int)blockIdx.x, n_chain = (int)gridDim.x;
    if (blk_surv) {
        // The search left the regular faces, the clip the surviving pairs of every block of FB target faces: this block
        // sums the (rows, entries) of the blocks in front of it itself -- in the order the assembly assigns stored rows,
        // i.e. hardware block order.  A few thousand pairs of words that all blocks read from L2 (b / 256 loads per thread,
        // issued together, while the block's own entry loads are in flight): no scan kernel, no launch in between.
        const int64_t per_xcd = (n_blocks + 7) >> 3;
        long long acc = 0;
        for (int hb = tid; hb < b; hb += FB) {
            const int64_t plb = remap ? (int64_t)(hb & 7) * per_xcd + (hb >> 3) : hb;
            if (plb < n_blocks) acc += ((long long)blk_rows[plb] << 31) | (long long)blk_surv[plb];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        __shared__ long long sh_part[FWAVES];
        if ((tid & 63) == 0) sh_part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            long long tot = 0;
#pragma unroll
            for (int w = 0; w < FWAVES; w++) tot += sh_part[w];
            sh_base = tot;
        }
    } else { // scanned beforehand (k_assemble_scan)
        if (tid == 0) sh_base = ((long long)base_rows[b] << 31) | (long long)base_nnz[b];
    }
    __syncthreads();
    const long long base = sh_base & 0x7fffffffll, row_base = sh_base >> 31;
    if (!base_nnz && b == n_chain - 1 && tid == 0) { // (own prefix: the last block of the chain knows the totals)
        const long long entries = base + block_nnz, rows = row_base + block_rows;
        counters->p_regular = (int32_t)entries; // entries of all regular rows
        counters->rows_regular = (int32_t)rows;
        indptr[rows] = (int32_t)entries;        // (= indptr[T] when there are no big faces)
    }
    if (regular) {
        const long long r = row_base + rowidx; // stored row of the face
        indptr[r] = (int32_t)(base + rowoff);
        row_order[r] = q_perm ? q_perm[t] : (int32_t)t;
        if (tile_key) {
            const int64_t mid = (t & ~(int64_t)(tile.n_run - 1)) + tile.n_run / 2;
            tile_key[r] = morton_key(tile, reinterpret_cast<const double4 *>(q_bbox)[mid < n_query ? mid : n_query - 1]);
        }
        if (my_nnz > XR_APPLY_LONG_ROW) {
            apply_long_rows[atomicAdd(&counters->n_apply_long, 1)] = (int32_t)r;
            atomicMax(&counters->max_row, my_nnz); // (long rows only: the apply asks whether any row exceeds its wave kernel)
        }
    }
    bool overflow_cap = false;
    for (int i0 = tid; i0 < total; i0 += 4 * FB) {
        double area[4];
        int row[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * FB;
            const int64_t c = (int64_t)seg.x + (i < total ? i : total - 1);
            area[u] = cand_area[c];
            row[u] = cand_tgt[c] - (int)t0; // (read again -- an L2 hit -- rather than kept in LDS)
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * FB;
            if (i >= total) continue;
            const int s = sh_stage[i];
            if (s == 0x7fffffff) continue;
            const int a0 = sh_lo[row[u]], a1 = a0 + sh_cnt[row[u]];
            int rank = 0;
            for (int j = a0; j < a1; j++) rank += sh_stage[j] < s ? 1 : 0;
            const long long pos = base + sh_rowoff[row[u]] + rank;
            if (pos < csr_capacity) {
                indices[pos] = s;
                data[pos] = relative ? area[u] / src_area[s] : area[u];
            } else {
                overflow_cap = true;
            }
        }
    }
    if (overflow_cap) atomicOr(&counters->error, 4);
}

// ---- the big faces' rows (side stream) -------------------------------------------------------------------------------
// Their order (by face id) is established inside k_search_big (slot_face), their offsets inside k_row_fill_long (scan_nnz
// mode): the chain search_big -> clip -> row_fill_long is the critical path of the weight build, every launch less counts.

// The finished rows of the big faces (ranked by k_row_fill_long into big_indices / big_data on the side stream) go
// BEHIND the regular rows: stored row T - n_big + slot.  Grid-stride over rows and entries.
__global__ void __launch_bounds__(256)
k_place_big(const int32_t *__restrict__ n_big_dev, const int32_t *__restrict__ slot_face,
            const int32_t *__restrict__ big_indptr, const int32_t *__restrict__ big_indices,
            const double *__restrict__ big_data, int64_t n_query, const int32_t *__restrict__ q_perm,
            const double *__restrict__ q_bbox, MortonParams tile, int32_t *__restrict__ tile_key,
            FusedCounters *counters, int32_t *__restrict__ indptr, int32_t *__restrict__ indices,
            double *__restrict__ data, int32_t *__restrict__ row_order, int32_t *__restrict__ apply_long_rows,
            int64_t csr_capacity, const int32_t *__restrict__ skip_if) {
    const int n_big = *n_big_dev;
    if (n_big == 0 || *skip_if > 0) return;
    const int64_t p_regular = counters->p_regular, t_reg = n_query - n_big;
    const int64_t p_big = big_indptr[n_big];
    if (p_regular + p_big > csr_capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&counters->error, 4);
        return;
    }
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    for (int64_t k = i0; k < n_big; k += stride) {
        const int f = slot_face[k];
        const int64_t r = t_reg + k;
        const int n = big_indptr[k + 1] - big_indptr[k];
        indptr[r] = (int32_t)(p_regular + big_indptr[k]);
        row_order[r] = q_perm ? q_perm[f] : f;
        if (tile_key) {
            const int64_t mid = ((int64_t)f & ~(int64_t)(tile.n_run - 1)) + tile.n_run / 2;
            tile_key[r] = morton_key(tile, reinterpret_cast<const double4 *>(q_bbox)[mid < n_query ? mid : n_query - 1]);
        }
        if (n > XR_APPLY_LONG_ROW) {
            apply_long_rows[atomicAdd(&counters->n_apply_long, 1)] = (int32_t)r;
            atomicMax(&counters->max_row, n);
        }
    }
    for (int64_t e = i0; e < p_big; e += stride) {
        indices[p_regular + e] = big_indices[e];
        data[p_regular + e] = big_data[e];
    }
    if (i0 == 0) indptr[n_query] = (int32_t)(p_regular + p_big);
}

// sizes, error bits and the long-row count -> host mailbox / device word of the matrix (after everything else)
// ... and clears all of them behind itself: the counter words are the engine's zero-at-rest scratch (the next xr_overlap
// starts without a memset).  One wave.
__global__ void k_publish_all(int32_t *c /* search counters, FusedCounters right behind them (c + 8) */, FusedCounters *fc,
                              int32_t *__restrict__ n_apply_long_out, int32_t *mail, int32_t seq, int64_t cap,
                              int64_t big_capacity) {
    // one load per lane (the 16 words are one line), so the host's wait is one round trip long
    const int t = threadIdx.x;
    int32_t w = 0;
    if (t < 16) w = c[t];
    // (the cursors of the eight regions of the regular pair queue, a line each: their sum is the number of regular pairs, which
    // takes the place of the single cursor of word 0)
    const int32_t cur = (t >= 16 && t < 24) ? c[QCUR_BASE + (t - 16) * QCUR_STRIDE] : 0;
    int32_t c_reg_sum = cur;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c_reg_sum += __shfl_xor(c_reg_sum, d, 64);
    if (t == 0) w = c_reg_sum;
    // mailbox slot of word t: [0..4] regular pairs, big pairs, big faces, not fitting, (unused); then the FusedCounters
    // fields in the order the host reads them: error, rows_regular, n_apply_long, p_regular, p_big, max_row
    int slot = -1;
    if (t < 5) slot = t;
    else if (t == 8 + 1) slot = 5;
    else if (t == 8 + 3) slot = 6;
    else if (t == 8 + 0) slot = 7;
    else if (t == 8 + 2) slot = 8;
    else if (t == 8 + 4) slot = 9;
    else if (t == 8 + 6) slot = 10;
    if (slot >= 0) mail[slot] = w;
    if (t >= 16 && t < 24) mail[11 + (t - 16)] = cur; // (the regions' own lengths: XR_DEBUG_FUSED)
    if (t == 8) *n_apply_long_out = w;
    if (t == 0) {
        // gate of an apply enqueued right behind this kernel (xr_overlap_apply_dev): open only if THIS attempt produced the
        // final matrix -- the very conditions the host checks after its read-back (overlap_tri); a failed attempt leaves
        // row pointers no kernel may follow
        const int32_t c_reg = c_reg_sum, c_big = c[1], n_pending = c[3], err = c[8 + 1], p_reg = c[8 + 2], p_big = c[8 + 4];
        const bool ok = c_reg >= 0 && c_big >= 0 && !(err & (1 | 4 | 8)) && n_pending == 0 && (int64_t)c_big <= big_capacity &&
                        (int64_t)p_reg + p_big <= cap;
        n_apply_long_out[1] = ok ? 1 : 0;
    }
    __builtin_amdgcn_s_waitcnt(0); // (every load above has returned before the words are cleared)
    if (t < 16) c[t] = 0;
    if (t >= 16 && t < 24) c[QCUR_BASE + (t - 16) * QCUR_STRIDE] = 0;
    (void)fc;
    // the host polls the sequence word (mailbox_wait_seq): it goes out BEHIND the words above (one wave: a system-scope
    // release covers the stores of all its lanes)
    __threadfence_system();
    if (t == 0) __hip_atomic_store(&mail[MAIL_SEQ_SLOT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace xr
