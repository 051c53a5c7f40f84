// xr_overlap.hip -- OverlapRegridder weight construction on the device.
//
// Replaces numba_celltree.CellTree2d.intersect_faces as called from
// UnstructuredGrid2d.overlap (xugrid/regrid/unstructured.py:109-135), the relative
// normalisation (:133-134) and MatrixCSR.from_triplet (xugrid/regrid/regridder.py:433-435,
// xugrid/core/sparse.py:61-78).
//
// Two pipelines.  Triangle x triangle pairs (the benchmark) take overlap_tri() below: search -> persistent bit-mask clip
// -> per-block scan -> assembly, big faces on a side stream, ONE host round trip (kernels in xr_overlap_fused.h,
// xr_clip_tri.h).  Everything else takes the general chain (all on the engine stream):
//   search         one thread per query (target) face walks the tree mesh's hierarchical grid and
//                  parks the bbox-overlapping tree records in 16 LDS slots; the block reserves its
//                  stretch of the candidate-pair queue with one atomic and writes it compacted
//                                               -> cand_off/cand_count[T], cand_tgt/src[C]
//   search_big     faces with too many rows / records / hits: block per face, count -> reserve -> fill
//   clip           one thread per candidate pair: Sutherland-Hodgman clip of the query polygon
//                  by the tree polygon, polygon buffers staged in LDS ([vertex][thread] layout,
//                  conflict-free 16-byte accesses), fan area                 -> cand_area[C]
//                  per-row counts of pairs with area > 0 by wave-aggregated atomics -> nnz_row[T]
//   scan           -> indptr[T+1]
//   row_fill       per query face: rank the surviving pairs by tree face id and write the CSR
//                  row (indices, data); data /= tree face area if relative.  Rows with many
//                  candidates go to one block each (all-pairs or LDS bitmap rank, k_row_fill_long).
// Query faces whose bbox spans many grid rows/records are searched by one block each
// (k_search_big) instead of one thread.
//
// The clip arithmetic mirrors oracle/xr_oracle.c (clip_polygons / sh_polygon_area) operation
// for operation; both are built with -ffp-contract=off, so areas agree bit for bit.
#include "xr_objects.h"
#include "xr_clip_tri.h"

namespace xr {

// ---------------------------------------------------------------------------------------------
// candidate search
// ---------------------------------------------------------------------------------------------
// A query face is "big" when its bbox spans many grid rows or many records (hull slivers of a
// Delaunay mesh, coarse target cells over a fine source): one thread would serialise thousands
// of tests, so those faces are queued and handled by one block each (k_search_big).
static constexpr int BIG_VISITS = 160; // records visited by one thread before it gives up

__device__ __forceinline__ bool rec_hit(float4 b, float qx0, float qx1, float qy0, float qy1) {
    return box_gap(b, qx0, qx1, qy0, qy1) < 0.0f;
}

// One traversal per query face: hits are parked in SLOTS LDS slots per face and counted; the block
// then compacts them into its stretch of the candidate-pair queue.  Faces with more than SLOTS
// hits, too many grid rows or too many visited records are "big" and go to the block-per-face
// kernel instead.
static constexpr int SLOTS = 16;
static constexpr int QCUR_STRIDE = 32; // words between the cursors of the regular queue's eight regions: a 128-byte line each
static constexpr int QCUR_BASE = 32;   // first of them in the control words of overlap_tri
static constexpr int TILE_RUN = 16; // rows per run in the tiling hint (128-byte output stores per variable).  Measured, K = 256 on
                                    // the benchmark matrix: runs of 64 rows / tiles of 24 extents 2.10 ms, 16 / 12: 1.72 ms (a qhull-numbered
                                    // target: long runs of consecutive ids are not compact); a lattice-numbered pair 0.99 ms either way

// XCD-aware block order (launch the grid rounded up to a multiple of 8): hardware block b runs on XCD b % 8; every
// XCD gets a CONTIGUOUS range of logical blocks (= one spatial region of the Morton-ordered work list), so that the
// lines shared by neighbouring blocks stay in one L2.  -> logical block (>= n_blocks: nothing to do)
__device__ __forceinline__ int64_t xcd_block(int64_t n_blocks, bool remap) {
    if (!remap) return blockIdx.x;
    const int64_t per_xcd = (n_blocks + 7) >> 3;
    return (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
}

// Level split: the walk only visits the grid levels that hold the bulk of the records.  The records of the upper
// levels (a Delaunay hull's slivers: a few hundred of a million) are the TAIL of the record arrays (levels are
// concatenated); when that tail is short (<= BIGREC_MAX) every block filters it against its own bounding box once
// (coalesced) and its faces test the few survivors from LDS, instead of every face walking 5-6 all but empty levels:
// the dependent cell_start -> record round trips of those levels were most of the search time.
static constexpr int WALK_PAD = 8;       // records of padding behind rec_bb (xr_mesh.hip allocates them): the walk's unclamped loads
static constexpr int BIGREC_MAX = 1024;  // records of the upper grid levels handled as a list
static constexpr int BIGREC_BLOCK = 64;  // ... of which at most this many may touch one block's bounding box

// PACK: the owning thread of a parked candidate rides in the top 8 bits of the record id (trees of at most 2^24 faces) instead
// of a byte array of its own: 19.5 instead of 23.5 KB of LDS per block = 8 instead of 6 resident blocks per CU for a
// kernel that is a chain of dependent loads.
// (The f32 box test of a step through subtract / max, 62 vector instructions per step of four records.  Two cheaper forms --
// compares combined on the scalar unit: 37 + 28 scalar; packed adds: 47 + 10 -- were measured in round 5 and changed nothing:
// the kernel waits on memory.  They are gone.)
template <bool PACK>
__global__ void __launch_bounds__(256)
k_search(const double *__restrict__ q_bbox, int64_t n_query, GridParams g, int64_t n_tree,
         const int32_t *__restrict__ cell_start,
         const float *__restrict__ rec_bb, int32_t *__restrict__ cand_count, int32_t *__restrict__ cand_off,
         int32_t *__restrict__ cand_tgt, int32_t *__restrict__ cand_src, int32_t *__restrict__ queue_cursor,
         int2 *__restrict__ block_seg, uint8_t *__restrict__ is_big, int32_t *__restrict__ big_list,
         int32_t *__restrict__ n_big, MortonParams tile, int32_t *__restrict__ tile_key, int32_t *__restrict__ nnz_row,
         bool remap, int32_t *__restrict__ blk_rows = nullptr /* optional: regular (non-big) faces per block, written */,
         int32_t *__restrict__ blk_surv = nullptr /* optional: the clip's survivor count of the block, cleared here */,
         int region_cap = 0 /* > 0: EIGHT queue regions of this many pairs with a cursor each (queue_cursor[0..7]), one per XCD */) {
    __shared__ __attribute__((aligned(16))) int32_t sh_slots[SLOTS + 1][256]; // [slot][thread]: conflict-free; + trash row
    __shared__ uint8_t sh_owner[PACK ? 1 : SLOTS * 256];
    __shared__ __attribute__((aligned(16))) float4 sh_bigbb[BIGREC_BLOCK];
    __shared__ int32_t sh_bigrec[BIGREC_BLOCK];
    __shared__ float sh_box[4][4];
    __shared__ int32_t sh_nbig;
    __shared__ int32_t sh_wave[4];
    __shared__ int32_t sh_base;
    const int64_t n_blocks = (n_query + 255) / 256;
    const int64_t lb = xcd_block(n_blocks, remap);
    if (threadIdx.x == 0 && blk_surv) blk_surv[lb] = 0; // (every hardware block owns one word, also the idle ones of the rounded grid)
    if (lb >= n_blocks) {
        if (threadIdx.x == 0 && blk_rows) blk_rows[lb] = 0;
        return;
    }
    const int64_t t = lb * 256 + threadIdx.x;
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    // ---- level split (uniform: scalar loads)
    int l_split = g.n_levels, big0 = (int)n_tree;
    for (int l = g.n_levels - 1; l >= 1; l--) {
        const int first = cell_start[g.base[l]];
        if ((int)n_tree - first > BIGREC_MAX) break;
        l_split = l;
        big0 = first;
    }
    double4 bb = make_double4(0, 0, 0, 0);
    float qx0 = INFINITY, qx1 = -INFINITY, qy0 = INFINITY, qy1 = -INFINITY;
    if (t < n_query) {
        bb = reinterpret_cast<const double4 *>(q_bbox)[t];
        qx0 = f32_below(bb.x - g.x0), qx1 = f32_above(bb.y - g.x0);
        qy0 = f32_below(bb.z - g.y0), qy1 = f32_above(bb.w - g.y0);
    }
    // Sparse levels in between (round 4): from the lowest level on whose tail holds at most 1/64 of the records (the 1M benchmark:
    // levels 2-4, 0.6 % of the records, ~0.05 per face -- but every face paid their cell bounds: three of its ~15 dependent
    // round trips and 11 % of its instructions), the levels [l_coop, l_split) are walked ONCE PER BLOCK: thread i takes grid
    // row i of the block's box on one of those levels, and the records that touch the box join the LDS list below.
    int l_coop = l_split;
    for (int l = l_split - 1; l >= 1; l--) {
        const int first = cell_start[g.base[l]];
        if (((int64_t)big0 - first) * 64 > n_tree) break;
        l_coop = l;
    }
    if (threadIdx.x == 0) sh_nbig = 0;
    if (big0 < (int)n_tree || l_coop < l_split) {
        // the block's bounding box, then one coalesced pass over the big records
        float bx0 = qx0, bx1 = qx1, by0 = qy0, by1 = qy1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, d, 64));
            bx1 = fmaxf(bx1, __shfl_xor(bx1, d, 64));
            by0 = fminf(by0, __shfl_xor(by0, d, 64));
            by1 = fmaxf(by1, __shfl_xor(by1, d, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            sh_box[w][0] = bx0; sh_box[w][1] = bx1; sh_box[w][2] = by0; sh_box[w][3] = by1;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++) {
            bx0 = fminf(bx0, sh_box[w][0]); bx1 = fmaxf(bx1, sh_box[w][1]);
            by0 = fminf(by0, sh_box[w][2]); by1 = fmaxf(by1, sh_box[w][3]);
        }
        for (int r = big0 + threadIdx.x; r < (int)n_tree; r += 256) {
            const float4 rb = rbb[r];
            if (rec_hit(rb, bx0, bx1, by0, by1)) {
                const int k = atomicAdd(&sh_nbig, 1);
                if (k < BIGREC_BLOCK) {
                    sh_bigrec[k] = r;
                    sh_bigbb[k] = rb;
                }
            }
        }
        if (l_coop < l_split) {
            // level-0 cells of the box's corners (the float box is a superset of every face's box; same monotone cell function
            // and the same shifts as the per-thread walk below)
            const int bcx0 = cell_coord((double)bx0 + g.x0, g.x0, g.inv_h0, g.nx[0]), bcx1 = cell_coord((double)bx1 + g.x0, g.x0, g.inv_h0, g.nx[0]);
            const int bcy0 = cell_coord((double)by0 + g.y0, g.y0, g.inv_h0, g.ny[0]), bcy1 = cell_coord((double)by1 + g.y0, g.y0, g.inv_h0, g.ny[0]);
            int n_items = 0;
            for (int l = l_coop; l < l_split; l++) n_items += (bcy1 >> (l * LEVEL_SHIFT)) - max((bcy0 >> (l * LEVEL_SHIFT)) - 1, 0) + 1;
            if (n_items > 256) {
                l_coop = l_split; // (a block spanning too many grid rows: its faces walk these levels themselves)
            } else if ((int)threadIdx.x < n_items) {
                int it = threadIdx.x, l = l_coop;
                for (;; l++) {
                    const int rows = (bcy1 >> (l * LEVEL_SHIFT)) - max((bcy0 >> (l * LEVEL_SHIFT)) - 1, 0) + 1;
                    if (it < rows) break;
                    it -= rows;
                }
                const int sh = l * LEVEL_SHIFT, nx = g.nx[l], base = g.base[l];
                const int cy = max((bcy0 >> sh) - 1, 0) + it;
                const int cx0 = max((bcx0 >> sh) - 1, 0), cx1 = bcx1 >> sh;
                const int r0 = cell_start[base + cy * nx + cx0], r1 = cell_start[base + cy * nx + cx1 + 1];
                for (int r = r0; r < r1; r++) {
                    const float4 rb = rbb[r];
                    if (rec_hit(rb, bx0, bx1, by0, by1)) {
                        const int k = atomicAdd(&sh_nbig, 1);
                        if (k < BIGREC_BLOCK) {
                            sh_bigrec[k] = r;
                            sh_bigbb[k] = rb;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (sh_nbig > BIGREC_BLOCK) { // too many for the list: this block walks every level
            l_split = g.n_levels;
            l_coop = l_split;
            __syncthreads();
            if (threadIdx.x == 0) sh_nbig = 0;
        }
    }
    __syncthreads();
    const int n_bigblk = sh_nbig;
    int count = 0;
    bool big = false;
    if (t < n_query) {
        nnz_row[t] = 0; // the clip kernel counts the surviving pairs of the row into it
        if (tile_key) {
            // row tiling hint for the many-variable apply: all rows of a run of TILE_RUN consecutive ids share the
            // key of the run's middle row, so runs stay contiguous (long output stores) and neighbouring runs meet
            const int64_t mid = (t & ~(int64_t)(tile.n_run - 1)) + tile.n_run / 2;
            tile_key[t] = morton_key(tile, reinterpret_cast<const double4 *>(q_bbox)[mid < n_query ? mid : n_query - 1]);
        }
        // Level-0 cells of the bbox corners, once; the level-l cell of x is (level-0 cell) >> l exactly (cell
        // sizes are power-of-two multiples and the clamped ranges nest), and the cell of x - h_l is that minus one:
        // no per-level floating-point cell arithmetic.  (A record that can overlap has xmin > q.xmin - 0.999 h_l,
        // so its cell is >= cell(q.xmin) - 1 for the same monotone cell function the index was built with.)
        const int c_x0 = cell_coord(bb.x, g.x0, g.inv_h0, g.nx[0]), c_x1 = cell_coord(bb.y, g.x0, g.inv_h0, g.nx[0]);
        const int c_y0 = cell_coord(bb.z, g.y0, g.inv_h0, g.ny[0]), c_y1 = cell_coord(bb.w, g.y0, g.inv_h0, g.ny[0]);
        constexpr int WALK_LOADS = 4;
        int visited = 0, n_rows = 0;
        for (int l = 0; l < l_coop; l++)
            n_rows += (c_y1 >> (l * LEVEL_SHIFT)) - max((c_y0 >> (l * LEVEL_SHIFT)) - 1, 0) + 1;
        big = n_rows > 8 * l_coop + 8;
        for (int l = 0; l < l_coop && !big; l++) {
            const int nx = g.nx[l], base = g.base[l];
            const int sh = l * LEVEL_SHIFT;
            const int cx0 = max((c_x0 >> sh) - 1, 0), cx1 = c_x1 >> sh;
            const int cy0 = max((c_y0 >> sh) - 1, 0), cy1 = c_y1 >> sh;
            for (int cyb = cy0; cyb <= cy1 && !big; cyb += 4) {
                // fetch the record runs of up to four grid rows before walking them
                int r0[4], r1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int cy = cyb + k <= cy1 ? cyb + k : cy1;
                    r0[k] = cell_start[base + cy * nx + cx0];
                    r1[k] = cyb + k <= cy1 ? cell_start[base + cy * nx + cx1 + 1] : r0[k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    visited += r1[k] - r0[k];
                    if (visited > BIG_VISITS) big = true;
                    if (!big) {
                        for (int r = r0[k]; r < r1[k]; r += WALK_LOADS) {
                            // WALK_LOADS independent 16-byte loads in flight per step (6 and 8 measured in round 4: no faster)
                            const int last = r1[k] - 1;
                            float4 bx[WALK_LOADS];
                            static_assert(WALK_LOADS - 1 <= WALK_PAD, "rec_bb padding");
                            // (no clamp of the index: rec_bb is padded by WALK_PAD records, a load beyond the run reads the next
                            // run or the padding and is masked below; PACK: ONE 32-bit byte offset from the uniform base per step --
                            // at most 2^24 records of 16 bytes -- and the records behind it through the instruction's immediate offset)
                            const char *step_base = reinterpret_cast<const char *>(rbb) + ((uint32_t)r << 4);
#pragma unroll
                            for (int u = 0; u < WALK_LOADS; u++)
                                bx[u] = PACK ? *reinterpret_cast<const float4 *>(step_base + 16 * u) : rbb[r + u];
                            // branch-free parking: the slot is written unconditionally and only kept (count advances) on a hit;
                            // beyond SLOTS everything lands in a trash row
#pragma unroll
                            for (int u = 0; u < WALK_LOADS; u++) {
                                sh_slots[count < SLOTS ? count : SLOTS][threadIdx.x] = r + u;
                                count += fmaxf(box_gap(bx[u], qx0, qx1, qy0, qy1), r + u <= last ? -INFINITY : 1.0f) < 0.0f ? 1 : 0;
                            }
                        }
                    }
                }
            }
        }
        // the big records that touch the block (from LDS)
        for (int k = 0; k < n_bigblk && !big; k++) {
            const bool h = rec_hit(sh_bigbb[k], qx0, qx1, qy0, qy1);
            sh_slots[count < SLOTS ? count : SLOTS][threadIdx.x] = sh_bigrec[k];
            count += h ? 1 : 0;
        }
        if (count > SLOTS) big = true;
        is_big[t] = big ? 1 : 0;
        if (big) big_list[atomicAdd(n_big, 1)] = (int32_t)t;
    }
    // The block's candidates go straight into the pair queue: block-level exclusive scan of the counts, ONE
    // atomic reservation per block, then the entries -- compacted in LDS -- are written as whole lines.  No
    // slot array, no global scan, no compaction pass; blocks land in the queue in arbitrary order (nothing
    // downstream depends on it: rows are re-ranked by tree face id), the rows of one block stay contiguous.
    const int mine = (t < n_query && !big) ? count : 0;
    if (blk_rows) {
        const int n_regular = __syncthreads_count(t < n_query && !big);
        if (threadIdx.x == 0) blk_rows[lb] = n_regular;
    }
    int own[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; j++) own[j] = j < mine ? sh_slots[j][threadIdx.x] : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) sh_wave[wave] = incl;
    __syncthreads(); // (also: every thread has read its slots)
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) woff += sh_wave[w];
        total += sh_wave[w];
    }
    const int lo = woff + incl - mine;
    // One returning atomic per block on ONE word is served at the memory side at ~10 M a second when thousands of blocks queue up
    // for it (10 us of this kernel, measured by giving every block a fixed stretch instead).  With region_cap > 0 the queue is
    // eight dense regions, one per XCD (hardware block b runs on XCD b mod 8, and with the XCD-aware block order that XCD owns a
    // contiguous eighth of the target faces): eight words share the traffic, and the persistent clip deals its chunks per
    // region anyway.
    if (threadIdx.x == 0) {
        const int x = region_cap > 0 ? (int)(blockIdx.x & 7) : 0;
        sh_base = (total > 0 ? atomicAdd(&queue_cursor[x * QCUR_STRIDE], total) : 0) + x * region_cap;
    }
    int32_t *flat = &sh_slots[0][0];
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        if (j < mine) {
            flat[lo + j] = PACK ? (own[j] | (int32_t)(threadIdx.x << 24)) : own[j];
            if (!PACK) sh_owner[lo + j] = (uint8_t)threadIdx.x;
        }
    }
    __syncthreads();
    const int base = sh_base;
    if (t < n_query && !big) { // big faces get their offset and count from k_search_big<false>
        cand_off[t] = base + lo;
        cand_count[t] = mine;
    }
    if (threadIdx.x == 0) block_seg[lb] = make_int2(base, total);
    const int32_t t0 = (int32_t)(lb * 256);
    for (int i = threadIdx.x; i < total; i += 256) {
        const int32_t v = flat[i];
        cand_tgt[base + i] = t0 + (PACK ? (int32_t)((uint32_t)v >> 24) : (int32_t)sh_owner[i]);
        cand_src[base + i] = PACK ? (v & 0xffffff) : v;
    }
}

// deposit one device word in the host mailbox (pinned memory)
__global__ void k_publish(const int32_t *__restrict__ src, int32_t *dst, int32_t *src2, int32_t *dst2) {
    if (threadIdx.x == 0) {
        *dst = *src;
        *dst2 = *src2;
        *src2 = 0; // the counter is reused afterwards
    }
}

// One block per big query face.  The face is scan-converted against the grid: for grid row cy of
// level l only the cells under the polygon's x-extent inside the y-slab of that row are visited
// (a thin hull sliver touches O(length) cells instead of the O(length^2) cells of its bbox).
// Lanes take one grid row each (64 rows per batch, batches dealt round-robin to the block's four
// waves); runs longer than LONG_RUN records are processed by the whole wave in 64-record chunks.
//
// Superset argument: a tree face s on level l with positive-area intersection has a point p in
// both polygons; its record sits in the cell of (xmin_s, ymin_s) with p.x - h < xmin_s <= p.x and
// p.y - h < ymin_s <= p.y (extent <= 0.999 h).  So for row cy the relevant points have
// y in [Y(cy), Y(cy) + 2h) and the record's cell column lies in [cell(xlo - h), cell(xhi)], with
// [xlo, xhi] the polygon's x-extent inside that slab (inflated by 1e-6 h against rounding).
static constexpr int LONG_RUN = 128;

__device__ __forceinline__ int wave_excl_scan_i32(int v, int lane) {
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    return incl - v;
}

// FUSED: ONE walk parks the face's candidates in LDS (up to BIG_STAGE of them); the block then reserves the face's
// stretch of the pair queue with one atomic and copies them out.  Faces with more candidates than the stage holds
// walk a second time and write straight to the queue (the first walk has counted them); faces whose stretch does
// not fit the queue as currently allocated are listed and filled by a second launch (FUSED = false) after the host
// regrew it.  (The two-walk version -- count, reserve, fill -- took twice as long per face, and a big face is a
// chain of dependent phases: the kernel's duration is the slowest face's.)
static constexpr int BIG_STAGE = 5120; // size of the stage (dynamic LDS; 3072 / 2048 / 1024 / 512 measured in round 5: no gain)
static int big_stage_entries() { return BIG_STAGE; }
static constexpr int BIG_RANK_MAX = 1 << 16; // big faces ranked by id (all-pairs, inside k_search_big); longer lists keep their order

template <bool FUSED>
__global__ void __launch_bounds__(256)
k_search_big(const double *__restrict__ q_bbox, const double *__restrict__ q_fxy,
             const uint8_t *__restrict__ q_len, const int32_t *__restrict__ q_off, int q_m, GridParams g,
             const int32_t *__restrict__ cell_start, const float *__restrict__ rec_bb,
             const int32_t *__restrict__ rec_face, const int32_t *__restrict__ big_list,
             const int32_t *__restrict__ n_big, int32_t *__restrict__ cand_off, int32_t *__restrict__ cand_count,
             int32_t *__restrict__ cand_tgt, int32_t *__restrict__ cand_src, int32_t *__restrict__ queue_cursor,
             int64_t capacity, int32_t *__restrict__ pending_list, int32_t *__restrict__ n_pending,
             int big_stage /* entries of the LDS stage (FUSED) */,
             int32_t *__restrict__ slot_face = nullptr /* optional: the listed faces in ascending id order, written */) {
    __builtin_amdgcn_s_setprio(3); // side-stream kernel: its waves go first in the SIMDs' issue arbitration (see overlap_tri)
    // one BLOCK per big face: its four waves take the 64-row batches round-robin; candidates are
    // appended through a per-face cursor in LDS (their order inside the row is irrelevant: rows are
    // ranked by tree face id afterwards)
    __shared__ double2 sh_poly[XR_MAX_FACE_NODES];
    extern __shared__ int32_t sh_stage[]; // [big_stage] (FUSED) / [1]
    __shared__ int sh_cursor;
    __shared__ int sh_out0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nb = *n_big;
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    __shared__ int32_t sh_rankw[4];
    for (int bi = blockIdx.x; bi < nb; bi += gridDim.x) {
        const int t = big_list[bi];
        const int np = q_len[t];
        if (slot_face) {
            // the face's rank among the listed ones by face id (k_search lists them in finishing order; ranking makes the
            // stored matrix the same from run to run).  Lists beyond BIG_RANK_MAX faces keep the order they were found in.
            int c = 0;
            if (nb <= BIG_RANK_MAX)
                for (int j = threadIdx.x; j < nb; j += 256) c += big_list[j] < t ? 1 : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            if (lane == 0) sh_rankw[wv] = c;
        }
        __syncthreads();
        if (slot_face && threadIdx.x == 0)
            slot_face[nb <= BIG_RANK_MAX ? sh_rankw[0] + sh_rankw[1] + sh_rankw[2] + sh_rankw[3] : bi] = t;
        if (threadIdx.x < np) sh_poly[threadIdx.x] = reinterpret_cast<const double2 *>(q_fxy)[face_vertex_base(q_off, t, q_m) + threadIdx.x];
        if (threadIdx.x == 0) sh_cursor = 0;
        __syncthreads();
        const double2 *poly = sh_poly;
        const double4 bb = reinterpret_cast<const double4 *>(q_bbox)[t];
        const float qx0 = f32_below(bb.x - g.x0), qx1 = f32_above(bb.y - g.x0);
        const float qy0 = f32_below(bb.z - g.y0), qy1 = f32_above(bb.w - g.y0);
        for (int pass = FUSED ? 0 : 1; pass < 2; pass++) {
        const bool STAGE = pass == 0; // first walk: park in LDS (and count); second walk: write to the queue
        if (!STAGE) {
            if (FUSED) {
                // (the first walk left the face's total in sh_cursor)
                __syncthreads();
                const int n = sh_cursor;
                __syncthreads();
                if (threadIdx.x == 0) {
                    const int base = atomicAdd(queue_cursor, n);
                    cand_count[t] = n;
                    cand_off[t] = base;
                    const bool fits = (int64_t)base + n <= capacity && base >= 0;
                    if (!fits) pending_list[atomicAdd(n_pending, 1)] = t;
                    sh_out0 = fits ? base : -1;
                    sh_cursor = 0;
                }
                __syncthreads();
                const int out = sh_out0;
                if (out < 0) break;
                if (n <= big_stage) { // everything is parked: copy it out, no second walk
                    for (int i = threadIdx.x; i < n; i += 256) {
                        cand_tgt[out + i] = t;
                        cand_src[out + i] = sh_stage[i];
                    }
                    break;
                }
            } else {
                if (threadIdx.x == 0) sh_out0 = cand_off[t];
                __syncthreads();
            }
            if (sh_out0 < 0) break;
        }
        const int out0 = STAGE ? 0 : sh_out0;
        for (int l = 0; l < g.n_levels; l++) {
            const double h = level_h(g, l), inv_h = level_inv_h(g, l);
            const double eps = 1e-6 * h;
            const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
            const int cy0 = cell_coord(bb.z - h, g.y0, inv_h, ny), cy1 = cell_coord(bb.w, g.y0, inv_h, ny);
            int batch_id = 0;
            for (int cyb = cy0; cyb <= cy1; cyb += 64, batch_id++) {
                if (((batch_id + l) & 3) != wv) continue;
                const int cy = cyb + lane;
                int r0 = 0, r1 = 0;
                // (the record test uses the polygon's x-extent INSIDE the row's slab, not the face's whole box: a record of this
                // row lies inside the slab in y, so a common point with the polygon has its x inside that extent -- the same
                // argument that lets the walk skip the cells beside it.  For a diagonal sliver this is the difference between
                // "every record of every visited cell" and the records along the sliver: the longest face of the 1M benchmark
                // went from ~4000 candidates, 763 of them real, to a third.)
                float rx0 = qx0, rx1 = qx1;
                if (cy <= cy1) {
                    // polygon x-extent inside the slab [ya, yb] of this grid row
                    const double ya = g.y0 + (double)cy * h - eps, yb = g.y0 + (double)(cy + 2) * h + eps;
                    double xlo = INFINITY, xhi = -INFINITY;
                    double2 p = poly[np - 1];
                    for (int k = 0; k < np; k++) {
                        const double2 q = poly[k];
                        const double ylo = fmin(p.y, q.y), yhi = fmax(p.y, q.y);
                        if (yhi >= ya && ylo <= yb) {
                            double xa = p.x, xb = q.x;
                            if (yhi > ylo) {
                                // clip the edge to the slab (parameter along p -> q)
                                const double inv = 1.0 / (q.y - p.y);
                                double t0 = (ya - p.y) * inv, t1 = (yb - p.y) * inv;
                                if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
                                t0 = fmax(t0, 0.0);
                                t1 = fmin(t1, 1.0);
                                xa = p.x + t0 * (q.x - p.x);
                                xb = p.x + t1 * (q.x - p.x);
                            }
                            xlo = fmin(xlo, fmin(xa, xb));
                            xhi = fmax(xhi, fmax(xa, xb));
                        }
                        p = q;
                    }
                    if (xhi >= xlo) {
                        xlo = fmax(xlo - eps, bb.x);
                        xhi = fmin(xhi + eps, bb.y);
                        const int cx0 = cell_coord(xlo - h, g.x0, inv_h, nx), cx1 = cell_coord(xhi, g.x0, inv_h, nx);
                        r0 = cell_start[base + cy * nx + cx0];
                        r1 = cell_start[base + cy * nx + cx1 + 1];
                        rx0 = fmaxf(qx0, f32_below(xlo - g.x0));
                        rx1 = fminf(qx1, f32_above(xhi - g.x0));
                    }
                }
                const int len = r1 - r0;
                // (1) long runs: the whole wave, 64 records at a time
                unsigned long long long_mask = __ballot(len > LONG_RUN);
                while (long_mask) {
                    const int src_lane = __ffsll((long long)long_mask) - 1;
                    long_mask &= long_mask - 1;
                    const int R0 = __shfl(r0, src_lane, 64), R1 = __shfl(r1, src_lane, 64);
                    const float X0 = __shfl(rx0, src_lane, 64), X1 = __shfl(rx1, src_lane, 64);
                    for (int rb = R0; rb < R1; rb += 64) {
                        const int r = rb + lane;
                        const bool hit = r < R1 && rec_hit(rbb[r], X0, X1, qy0, qy1);
                        const unsigned long long mask = __ballot(hit);
                        const int n = __popcll(mask);
                        if (n > 0) {
                            int slot0 = 0;
                            if (lane == 0) slot0 = atomicAdd(&sh_cursor, n);
                            slot0 = __shfl(slot0, 0, 64);
                            if (hit) {
                                const int slot = slot0 + __popcll(mask & lt_mask);
                                if (STAGE) {
                                    if (slot < big_stage) sh_stage[FUSED ? slot : 0] = r;
                                } else {
                                    cand_tgt[out0 + slot] = t;
                                    cand_src[out0 + slot] = r;
                                }
                            }
                        }
                    }
                }
                // (2) short runs: one lane per grid row
                const int my_r1 = len > LONG_RUN ? r0 : r1;
                int cnt = 0;
                for (int r = r0; r < my_r1; r++) cnt += rec_hit(rbb[r], rx0, rx1, qy0, qy1) ? 1 : 0;
                const int excl = wave_excl_scan_i32(cnt, lane);
                const int batch = __shfl(excl + cnt, 63, 64);
                if (batch > 0) {
                    int slot0 = 0;
                    if (lane == 0) slot0 = atomicAdd(&sh_cursor, batch);
                    slot0 = __shfl(slot0, 0, 64);
                    int pos = slot0 + excl;
                    for (int r = r0; r < my_r1; r++) {
                        if (rec_hit(rbb[r], rx0, rx1, qy0, qy1)) {
                            if (STAGE) {
                                if (pos < big_stage) sh_stage[FUSED ? pos : 0] = r;
                            } else {
                                cand_tgt[out0 + pos] = t;
                                cand_src[out0 + pos] = r;
                            }
                            pos++;
                        }
                    }
                }
            }
        }
        } // pass
    }
}

// ---------------------------------------------------------------------------------------------
// What the reference does with a candidate pair BEFORE it clips (numba_celltree, restated in oracle/xr_oracle.c): the two
// faces' exact boxes must overlap STRICTLY (boxes_intersect, :530) and the separating-axis test must not separate them
// (sat_intersect, :757; touching counts as intersecting).  For faces that overlap properly both tests pass and the engine
// skips them.  They matter where faces merely touch or are degenerate: there the clip returns slivers of 1e-20 ... 1e-36
// that the reference never computes --
//   * a mesh against itself (or a mesh sharing its nodes): neighbours across a corner have boxes that touch, not overlap
//     (23 pairs of 60 394 on a 60k-face Delaunay mesh; the engine's float boxes are supersets and let them through),
//   * zero-area and needle source faces: the clip of a target by a segment leaves rounding dust the SAT rejects.
// A pair that fails either test has faces whose interiors are disjoint up to a penetration of a few ulp(coordinate), so the
// clip's area for it is at most ~1e-15 * |coordinate| * perimeter.  overlap_dust_threshold() bounds that for EVERY pair of
// the two meshes (largest coordinate x the smaller mesh's largest face extent, with a factor of four to spare); the clip kernels run the two
// tests -- on the float64 vertices, fetched again -- only for the lanes whose area is positive and below it: a handful
// per million pairs on unrelated meshes, the touching neighbours on related ones.  The tests' arithmetic is the oracle's
// (order and orientation of the vertices do not enter: min / max of the same products).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_passes_box_and_sat(const double2 *__restrict__ a, int na, const double2 *__restrict__ b, int nb) {
    double ax0 = INFINITY, ax1 = -INFINITY, ay0 = INFINITY, ay1 = -INFINITY, bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
    for (int j = 0; j < na; j++) {
        const double2 q = a[j];
        ax0 = fmin(ax0, q.x), ax1 = fmax(ax1, q.x), ay0 = fmin(ay0, q.y), ay1 = fmax(ay1, q.y);
    }
    for (int j = 0; j < nb; j++) {
        const double2 q = b[j];
        bx0 = fmin(bx0, q.x), bx1 = fmax(bx1, q.x), by0 = fmin(by0, q.y), by1 = fmax(by1, q.y);
    }
    if (!(ax0 < bx1 && bx0 < ax1 && ay0 < by1 && by0 < ay1)) return false;
    for (int pass = 0; pass < 2; pass++) {
        const double2 *p = pass ? b : a;
        const int np = pass ? nb : na;
        for (int i = 0; i < np; i++) {
            const double2 v0 = p[i], v1 = p[i + 1 < np ? i + 1 : 0];
            const double nx = -(v1.y - v0.y), ny = v1.x - v0.x; // edge normal
            if (nx == 0 && ny == 0) continue;
            double amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
            for (int j = 0; j < na; j++) {
                const double2 q = a[j];
                const double d = nx * q.x + ny * q.y;
                amin = d < amin ? d : amin;
                amax = d > amax ? d : amax;
            }
            for (int j = 0; j < nb; j++) {
                const double2 q = b[j];
                const double d = nx * q.x + ny * q.y;
                bmin = d < bmin ? d : bmin;
                bmax = d > bmax ? d : bmax;
            }
            if (amax < bmin || bmax < amin) return false;
        }
    }
    return true;
}
// -> the pair's area after the reference's pre-clip tests: unchanged, or 0 for rounding dust of a pair they reject
__device__ __forceinline__ double confirm_dust(double area, double dust, bool active, const double2 *__restrict__ a, int na,
                                               const double2 *__restrict__ b, int nb) {
    const bool suspicious = active && area > 0 && area <= dust;
    if (__any(suspicious)) {
        if (suspicious && !pair_passes_box_and_sat(a, na, b, nb)) area = 0.0;
    }
    return area;
}

// ---------------------------------------------------------------------------------------------
// Sutherland-Hodgman clip + fan area.  LDS: two polygon buffers per thread, [vertex][thread].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sh_inside(P2 p, P2 r, P2 U) { return U.x * (p.y - r.y) > U.y * (p.x - r.x); }

__device__ __forceinline__ bool sh_intersection(P2 a, P2 V, P2 r, P2 N, P2 &out) {
    const double wx = r.x - a.x, wy = r.y - a.y;
    const double nw = N.x * wx + N.y * wy;
    const double nv = N.x * V.x + N.y * V.y;
    if (nv != 0) {
        const double tt = nw / nv;
        out.x = a.x + tt * V.x;
        out.y = a.y + tt * V.y;
        return true;
    }
    return false;
}

static constexpr double AREA_OVERFLOW = -1.0; // sentinel: polygon buffer too small, redo with MAXV=64

template <int MAXV, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_clip(const double *__restrict__ q_fxy, const uint8_t *__restrict__ q_len, const int32_t *__restrict__ q_off, int q_m,
       const int32_t *__restrict__ q_perm, const double *__restrict__ s_fxy, const uint8_t *__restrict__ s_len, const int32_t *__restrict__ s_off,
       int s_m, const int32_t *__restrict__ cand_tgt, const int32_t *__restrict__ cand_src, int64_t n_cand,
       double *__restrict__ cand_area, bool redo_only, const int32_t *__restrict__ rec_face,
       int32_t *__restrict__ cand_sid, int32_t *__restrict__ overflow_count, int32_t *__restrict__ nnz_row, double dust) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *sh = reinterpret_cast<double2 *>(smem);
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = c < n_cand && !(redo_only && cand_area[c] != AREA_OVERFLOW);
    int t = -1;
    double area = 0.0;
    if (active) {
    t = cand_tgt[c];
    const int s = cand_src[c];
    cand_sid[c] = rec_face[s];
    const int nt = q_len[t], ns = s_len[s];
    double2 *out = sh + threadIdx.x;                // out[j * BLOCK]
    double2 *in = sh + MAXV * BLOCK + threadIdx.x;  // in[j * BLOCK]
    const double2 *tf = reinterpret_cast<const double2 *>(q_fxy) + face_vertex_base(q_off, t, q_m);
    const double *sf = s_fxy + 2 * face_vertex_base(s_off, s, s_m);
    for (int j = 0; j < nt; j++) out[j * BLOCK] = tf[j];
    int n_output = nt;
    bool overflow = false;
    P2 r = load_p2(sf, ns - 1);
    bool empty = false;
    for (int i = 0; i < ns; i++) {
        const P2 sv = load_p2(sf, i);
        const P2 U{sv.x - r.x, sv.y - r.y};
        if (U.x == 0 && U.y == 0) continue;
        const P2 N{-U.y, U.x};
        const int length = n_output;
        {
            double2 *tmp = in;
            in = out;
            out = tmp;
        }
        n_output = 0;
        const double2 a2 = in[(length - 1) * BLOCK];
        P2 a{a2.x, a2.y};
        bool a_inside = sh_inside(a, r, U);
        for (int j = 0; j < length; j++) {
            const double2 b2 = in[j * BLOCK];
            const P2 b{b2.x, b2.y};
            const P2 V{b.x - a.x, b.y - a.y};
            if (V.x == 0 && V.y == 0) continue;
            bool b_inside = sh_inside(b, r, U);
            if (b_inside) {
                if (!a_inside) {
                    P2 pt;
                    if (sh_intersection(a, V, r, N, pt)) {
                        if (n_output < MAXV) out[n_output * BLOCK] = make_double2(pt.x, pt.y);
                        else overflow = true;
                        n_output++;
                    }
                }
                if (n_output < MAXV) out[n_output * BLOCK] = b2;
                else overflow = true;
                n_output++;
            } else if (a_inside) {
                P2 pt;
                if (sh_intersection(a, V, r, N, pt)) {
                    if (n_output < MAXV) out[n_output * BLOCK] = make_double2(pt.x, pt.y);
                    else overflow = true;
                    n_output++;
                } else {
                    b_inside = true;
                    if (n_output < MAXV) out[n_output * BLOCK] = b2;
                    else overflow = true;
                    n_output++;
                }
            }
            a = b;
            a_inside = b_inside;
        }
        if (overflow) break;
        if (n_output < 3) {
            empty = true;
            break;
        }
        r = sv;
    }
    if (overflow) {
        area = AREA_OVERFLOW;
        atomicAdd(overflow_count, 1);
    } else if (!empty) {
        // fan area from the first clipped vertex (local origin)
        const double2 a2 = out[0];
        const double2 b2 = out[BLOCK];
        double ux = b2.x - a2.x, uy = b2.y - a2.y;
        for (int i = 2; i < n_output; i++) {
            const double2 c2 = out[i * BLOCK];
            const double vx = a2.x - c2.x, vy = a2.y - c2.y;
            area += fabs(ux * vy - uy * vx);
            ux = vx;
            uy = vy;
        }
        area = 0.5 * area;
    }
    if (area > 0 && area <= dust && !pair_passes_box_and_sat(tf, nt, reinterpret_cast<const double2 *>(sf), ns)) area = 0.0;
    cand_area[c] = area;
    } // active
    // per-row survivor counts: candidates of one query face are contiguous, so a wave holds a
    // few runs of equal t; the head lane of each run adds the run's survivor count.
    {
        const int lane = threadIdx.x & 63;
        const int t_prev = __shfl_up(t, 1, 64);
        const bool head = lane == 0 || t_prev != t;
        const unsigned long long heads = __ballot(head);
        const unsigned long long surv = __ballot(active && area > 0);
        if (head && t >= 0) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
            run &= ~((1ull << lane) - 1);
            const int n = __popcll(surv & run);
            if (n > 0) atomicAdd(&nnz_row[t], n);
        }
    }
}

// Small-polygon variant (query + tree vertices <= MAXV, i.e. triangles / quads): the subject
// polygon lives in REGISTERS (statically indexed, loops fully unrolled and predicated on the
// current length); only the output polygon, whose write index is data dependent, is staged in
// LDS ([vertex][thread]).  Half the LDS of the generic kernel -> twice the resident waves.
// The arithmetic and its order are those of k_clip / the oracle.
// -> the pair's area (AREA_OVERFLOW: more than MAXV vertices; before the confirmation of rounding dust); `out` = the
// thread's LDS column, out[j * BLOCK]
template <int MAXV, int BLOCK, bool TRI>
__device__ __forceinline__ double clip_small_pair(const double2 *__restrict__ tf, int nt, const double *__restrict__ sf, int ns,
                                                  double2 *out) {
    double area = 0.0;
    P2 in[MAXV];
    P2 last{0.0, 0.0};
#pragma unroll
    for (int j = 0; j < MAXV; j++) {
        if (j < nt) {
            const double2 v = tf[j];
            in[j] = P2{v.x, v.y};
            last = in[j];
        }
    }
    int length = nt;
    bool overflow = false, empty = false;
    // all clipper vertices up front (three independent 16-byte loads for triangles)
    P2 sv_all[TRI ? 3 : 1];
    if (TRI) {
#pragma unroll
        for (int i = 0; i < 3; i++) sv_all[i] = load_p2(sf, i);
    }
    P2 r = TRI ? sv_all[2] : load_p2(sf, ns - 1);
#pragma unroll
    for (int i = 0; i < (TRI ? 3 : XR_MAX_FACE_NODES); i++) {
        if (!TRI && i >= ns) break;
        const P2 sv = TRI ? sv_all[i] : load_p2(sf, i);
        const P2 U{sv.x - r.x, sv.y - r.y};
        if (U.x == 0 && U.y == 0) continue;
        const P2 N{-U.y, U.x};
        int n_output = 0;
        P2 a = last;
        bool a_inside = sh_inside(a, r, U);
#pragma unroll
        for (int j = 0; j < MAXV; j++) {
            if (j < length) {
                // flat body: the only nested divergent region is the division of a crossing
                const P2 b = in[j];
                const P2 V{b.x - a.x, b.y - a.y};
                const bool live = !(V.x == 0 && V.y == 0);
                bool b_inside = sh_inside(b, r, U);
                const bool cross = live && (b_inside != a_inside);
                P2 pt{0.0, 0.0};
                bool have_pt = false;
                if (cross) have_pt = sh_intersection(a, V, r, N, pt);
                // S-H emission: entering -> [pt] b ; inside -> b ; leaving -> pt, or (parallel-edge
                // quirk) b, which then counts as inside
                const bool quirk = cross && !b_inside && !have_pt;
                if (cross && have_pt) {
                    out[(n_output < MAXV ? n_output : MAXV) * BLOCK] = make_double2(pt.x, pt.y);
                    n_output++;
                    last = pt;
                }
                b_inside = b_inside || quirk;
                if (live && b_inside) {
                    out[(n_output < MAXV ? n_output : MAXV) * BLOCK] = make_double2(b.x, b.y);
                    n_output++;
                    last = b;
                }
                if (live) {
                    a = b;
                    a_inside = b_inside;
                }
            }
        }
        overflow = n_output > MAXV;
        if (overflow) break;
        if (n_output < 3) {
            empty = true;
            break;
        }
        length = n_output;
#pragma unroll
        for (int j = 0; j < MAXV; j++) {
            if (j < length) {
                const double2 v = out[j * BLOCK];
                in[j] = P2{v.x, v.y};
            }
        }
        r = sv;
    }
    if (overflow) return AREA_OVERFLOW;
    if (!empty) {
        const P2 a0 = in[0];
        double ux = in[1].x - a0.x, uy = in[1].y - a0.y;
#pragma unroll
        for (int i = 2; i < MAXV; i++) {
            if (i < length) {
                const double vx = a0.x - in[i].x, vy = a0.y - in[i].y;
                area += fabs(ux * vy - uy * vx);
                ux = vx;
                uy = vy;
            }
        }
        area = 0.5 * area;
    }
    return area;
}

template <int MAXV, int BLOCK, bool TRI>
__global__ void __launch_bounds__(BLOCK)
k_clip_small(const double *__restrict__ q_fxy, const uint8_t *__restrict__ q_len, const int32_t *__restrict__ q_off, int q_m,
             const int32_t *__restrict__ q_perm, const double *__restrict__ s_fxy,
             const uint8_t *__restrict__ s_len, const int32_t *__restrict__ s_off, int s_m, const int32_t *__restrict__ cand_tgt,
             const int32_t *__restrict__ cand_src, int64_t n_cand, double *__restrict__ cand_area,
             const int32_t *__restrict__ rec_face, int32_t *__restrict__ cand_sid,
             int32_t *__restrict__ overflow_count, int32_t *__restrict__ nnz_row, bool remap, double dust) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *out = reinterpret_cast<double2 *>(smem) + threadIdx.x; // out[j * BLOCK]
    const int64_t n_blocks = (n_cand + BLOCK - 1) / BLOCK;
    const int64_t lb = xcd_block(n_blocks, remap);
    if (lb >= n_blocks) return;
    const int64_t c = lb * BLOCK + threadIdx.x;
    const bool active = c < n_cand;
    int t = -1;
    double area = 0.0;
    if (active) {
        t = cand_tgt[c];
        const int s = cand_src[c];
        cand_sid[c] = rec_face[s];
        // TRI: both meshes are pure triangle meshes -> vertex counts are compile-time constants
        const int nt = TRI ? 3 : q_len[t], ns = TRI ? 3 : s_len[s];
        const double2 *tf = reinterpret_cast<const double2 *>(q_fxy) + face_vertex_base(q_off, t, q_m);
        const double *sf = s_fxy + 2 * face_vertex_base(s_off, s, s_m);
        area = clip_small_pair<MAXV, BLOCK, TRI>(tf, nt, sf, ns, out);
        if (area == AREA_OVERFLOW) atomicAdd(overflow_count, 1);
        if (area > 0 && area <= dust && !pair_passes_box_and_sat(tf, nt, reinterpret_cast<const double2 *>(sf), ns)) area = 0.0;
        cand_area[c] = area;
    }
    {
        const int lane = threadIdx.x & 63;
        const int t_prev = __shfl_up(t, 1, 64);
        const bool head = lane == 0 || t_prev != t;
        const unsigned long long heads = __ballot(head);
        const unsigned long long surv = __ballot(active && area > 0);
        if (head && t >= 0) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
            run &= ~((1ull << lane) - 1);
            const int n = __popcll(surv & run);
            if (n > 0) atomicAdd(&nnz_row[t], n);
        }
    }
}

// Triangle x triangle pairs (both meshes pure triangle meshes): the flag / compaction formulation of
// xr_clip_tri.h -- about half the instructions of k_clip_small<6, 256, true>, same areas bit for bit.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_clip_tri(const double *__restrict__ q_fxy, int q_m, const double *__restrict__ s_fxy, int s_m,
           const int32_t *__restrict__ cand_tgt, const int32_t *__restrict__ cand_src, int64_t n_cand,
           double *__restrict__ cand_area, const int32_t *__restrict__ rec_face, int32_t *__restrict__ cand_sid,
           int32_t *__restrict__ overflow_count, int32_t *__restrict__ nnz_row, bool remap, double dust) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *col = reinterpret_cast<double2 *>(smem) + threadIdx.x * (TRI_MAXV + 1); // the lane's TRI_MAXV + 1 slots
    __shared__ uint2 sh_lut[TRI_LUT];
    const int64_t n_blocks = (n_cand + BLOCK - 1) / BLOCK;
    const int64_t lb = xcd_block(n_blocks, remap);
    if (lb >= n_blocks) return;
    tri_lut_init(sh_lut);
    __syncthreads();
    const int64_t c = lb * BLOCK + threadIdx.x;
    const bool active = c < n_cand;
    int t = -1, s = 0;
    P2 tv[3] = {{0, 0}, {0, 0}, {0, 0}}, sv[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (active) {
        t = cand_tgt[c];
        s = cand_src[c];
        cand_sid[c] = rec_face[s];
        const double2 *tf = reinterpret_cast<const double2 *>(q_fxy) + (int64_t)t * q_m;
        const double2 *sf = reinterpret_cast<const double2 *>(s_fxy) + (int64_t)s * s_m;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double2 a = tf[j], b = sf[j];
            tv[j] = P2{a.x, a.y};
            sv[j] = P2{b.x, b.y};
        }
    }
    double area = tri_clip_area(tv, sv, col, sh_lut, active);
    area = confirm_dust(area, dust, active, reinterpret_cast<const double2 *>(q_fxy) + (int64_t)t * q_m, 3,
                        reinterpret_cast<const double2 *>(s_fxy) + (int64_t)s * s_m, 3);
    if (active) {
        if (area == TRI_AREA_OVERFLOW) {
            area = AREA_OVERFLOW;
            atomicAdd(overflow_count, 1);
        }
        cand_area[c] = area;
    }
    {
        const int lane = threadIdx.x & 63;
        const int t_prev = __shfl_up(t, 1, 64);
        const bool head = lane == 0 || t_prev != t;
        const unsigned long long heads = __ballot(head);
        const unsigned long long surv = __ballot(active && area > 0);
        if (head && t >= 0) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
            run &= ~((1ull << lane) - 1);
            const int n = __popcll(surv & run);
            if (n > 0) atomicAdd(&nnz_row[t], n);
        }
    }
}

// Quadrilateral (raster / quad mesh) targets against a triangle source -- the shape of the reference's unstructured ->
// raster regridding: the same flag / compaction clip with a subject of up to four vertices (xr_clip_tri.h, MAXV = 7);
// target faces may be triangles with a fill slot (q_len).  Replaces k_clip_small<8> for this pair of shapes.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_clip_quad_tri(const double *__restrict__ q_fxy, const uint8_t *__restrict__ q_len, const double *__restrict__ s_fxy,
                const int32_t *__restrict__ cand_tgt, const int32_t *__restrict__ cand_src, int64_t n_cand,
                double *__restrict__ cand_area, const int32_t *__restrict__ rec_face, int32_t *__restrict__ cand_sid,
                int32_t *__restrict__ overflow_count, int32_t *__restrict__ nnz_row, bool remap, double dust) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (QUAD_MAXV + 2 slots per lane: 144 bytes -- 128 would put the 16-byte accesses of all lanes on the same banks)
    double2 *col = reinterpret_cast<double2 *>(smem) + threadIdx.x * (QUAD_MAXV + 2);
    __shared__ uint2 sh_lut[QUAD_LUT];
    const int64_t n_blocks = (n_cand + BLOCK - 1) / BLOCK;
    const int64_t lb = xcd_block(n_blocks, remap);
    if (lb >= n_blocks) return;
    poly_lut_init<QUAD_MAXV>(sh_lut);
    __syncthreads();
    const int64_t c = lb * BLOCK + threadIdx.x;
    const bool active = c < n_cand;
    int t = -1, n0 = 3, s = 0;
    P2 tv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, sv[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (active) {
        t = cand_tgt[c];
        s = cand_src[c];
        cand_sid[c] = rec_face[s];
        n0 = q_len[t];
        const double2 *tf = reinterpret_cast<const double2 *>(q_fxy) + (int64_t)t * 4;
        const double2 *sf = reinterpret_cast<const double2 *>(s_fxy) + (int64_t)s * 3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < n0) {
                const double2 a = tf[j];
                tv[j] = P2{a.x, a.y};
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double2 b = sf[j];
            sv[j] = P2{b.x, b.y};
        }
    }
    double area = poly_clip_area<QUAD_MAXV, 4>(tv, n0, sv, col, sh_lut, active);
    area = confirm_dust(area, dust, active, reinterpret_cast<const double2 *>(q_fxy) + (int64_t)t * 4, n0,
                        reinterpret_cast<const double2 *>(s_fxy) + (int64_t)s * 3, 3);
    if (active) {
        if (area == TRI_AREA_OVERFLOW) {
            area = AREA_OVERFLOW;
            atomicAdd(overflow_count, 1);
        }
        cand_area[c] = area;
    }
    {
        const int lane = threadIdx.x & 63;
        const int t_prev = __shfl_up(t, 1, 64);
        const bool head = lane == 0 || t_prev != t;
        const unsigned long long heads = __ballot(head);
        const unsigned long long surv = __ballot(active && area > 0);
        if (head && t >= 0) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            unsigned long long run = next == 64 ? ~0ull : ((1ull << next) - 1);
            run &= ~((1ull << lane) - 1);
            const int n = __popcll(surv & run);
            if (n > 0) atomicAdd(&nnz_row[t], n);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CSR assembly
// ---------------------------------------------------------------------------------------------
// recount of the survivors per row; only used after the rare clip-buffer overflow redo
__global__ void __launch_bounds__(256) k_row_count(const int32_t *__restrict__ cand_off,
                                                  const int32_t *__restrict__ cand_count,
                                                  const double *__restrict__ cand_area, int64_t n_query,
                                                  const int32_t *__restrict__ q_perm,
                                                  int32_t *__restrict__ nnz_row) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_query) return;
    int n = 0;
    for (int c = cand_off[t]; c < cand_off[t] + cand_count[t]; c++) n += cand_area[c] > 0 ? 1 : 0;
    nnz_row[t] = n;
}

static constexpr int ROW_SHORT = 24; // rows with more candidates go to the block-per-row kernel

static constexpr int ROW_LDS = 3072; // short-row candidate entries one block can stage in LDS

// One block = 256 consecutive query faces (query order).  The candidate entries of the block's
// SHORT rows (<= ROW_SHORT candidates; long rows go to k_row_fill_long) are packed into LDS -- the
// block's candidate segment is contiguous, so the loads are coalesced -- then the block works
// ENTRY-parallel: each thread takes entries, ranks a survivor among the survivors of its row
// (adjacent in LDS) and writes it to its final CSR position.  Rows are stored in QUERY order
// (row r belongs to the caller's face q_perm[r], xr_csr::row_order).
__global__ void __launch_bounds__(256)
k_row_fill(const int32_t *__restrict__ cand_off, const int32_t *__restrict__ cand_count,
           const int2 *__restrict__ block_seg, const uint8_t *__restrict__ is_big,
           const int32_t *__restrict__ cand_tgt, const int32_t *__restrict__ cand_sid,
           const double *__restrict__ cand_area, int64_t n_query,
           const int32_t *__restrict__ indptr, const double *__restrict__ src_area, bool relative,
           int32_t *__restrict__ indices, double *__restrict__ data, int32_t *__restrict__ long_rows,
           int32_t *__restrict__ n_long, int32_t *__restrict__ apply_long_rows,
           int32_t *__restrict__ n_apply_long, bool remap) {
    __shared__ int32_t sh_src[ROW_LDS];  // tree face id of a survivor, INT_MAX for area <= 0
    __shared__ uint16_t sh_row[ROW_LDS];
    __shared__ int32_t sh_c0[256];   // first candidate of the row (global index)
    __shared__ int32_t sh_lds[257];  // LDS offset of the row (short rows only), [256] = total
    __shared__ int32_t sh_ptr[256];  // CSR offset of the row
    __shared__ int32_t sh_wave[4];
    const int64_t n_blocks = (n_query + 255) / 256;
    const int64_t lb = xcd_block(n_blocks, remap);
    if (lb >= n_blocks) return;
    const int64_t t0 = lb * 256;
    const int64_t t = t0 + threadIdx.x;
    // the candidates of the block's rows that k_search wrote itself (all but its "big" faces) are one stretch
    const int2 seg = block_seg[lb];
    const int seg0 = seg.x, seg1 = seg.x + seg.y;
    int c0 = 0, len = 0;
    bool is_short = true; // packed here; everything else (long rows, and the "big" faces of the search, whose
                          // candidates live elsewhere in the queue) goes to the block-per-row kernel
    if (t < n_query) {
        c0 = cand_off[t];
        len = cand_count[t];
        is_short = len <= ROW_SHORT && !is_big[t];
        sh_ptr[threadIdx.x] = indptr[t];
        if (!is_short) long_rows[atomicAdd(n_long, 1)] = (int32_t)t;
        if (indptr[t + 1] - indptr[t] > XR_APPLY_LONG_ROW) apply_long_rows[atomicAdd(n_apply_long, 1)] = (int32_t)t;
    }
    const int slen = is_short ? len : 0;
    sh_c0[threadIdx.x] = is_short ? c0 : -1;
    // block exclusive scan of the short-row lengths
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = slen;
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
    sh_lds[threadIdx.x] = woff + incl - slen;
    if (threadIdx.x == 0) sh_lds[256] = total;
    __syncthreads();
    if (total <= ROW_LDS) {
        for (int j = seg0 + threadIdx.x; j < seg1; j += 256) {
            const int row = cand_tgt[j] - (int)t0;
            const int rc0 = sh_c0[row];
            if (rc0 < 0) continue; // entry of a long row
            const int k = sh_lds[row] + (j - rc0);
            sh_src[k] = cand_area[j] > 0 ? cand_sid[j] : 0x7fffffff;
            sh_row[k] = (uint16_t)row;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < total; k += 256) {
            const int s = sh_src[k];
            if (s == 0x7fffffff) continue;
            const int row = sh_row[k];
            const int a0 = sh_lds[row], a1 = row == 255 ? total : sh_lds[row + 1];
            int rank = 0;
            for (int i = a0; i < a1; i++) rank += sh_src[i] < s ? 1 : 0;
            const double a = cand_area[sh_c0[row] + (k - a0)];
            const int pos = sh_ptr[row] + rank;
            indices[pos] = s;
            data[pos] = relative ? a / src_area[s] : a;
        }
    } else if (t < n_query && is_short) {
        // more short-row candidates than the LDS stage holds: rank straight from memory
        const int base = indptr[t];
        for (int i = c0; i < c0 + len; i++) {
            const double a = cand_area[i];
            if (!(a > 0)) continue;
            const int s = cand_sid[i];
            int rank = 0;
            for (int j = c0; j < c0 + len; j++) rank += (cand_area[j] > 0 && cand_sid[j] < s) ? 1 : 0;
            indices[base + rank] = s;
            data[base + rank] = relative ? a / src_area[s] : a;
        }
    }
}

// Rows that do not fit the packed short-row path (candidates > ROW_SHORT) are listed by k_row_fill and
// finished by ONE kernel, one block per row: up to ROW_BLOCK candidates by an all-pairs rank (ids staged in
// LDS, 16-byte broadcast reads: a few microseconds), longer rows by the bitmap rank below.  (Separate
// wave-per-row / block-per-row / bitmap kernels were each pure latency on a few hundred rows: 14 + 21 + 26 us
// back to back.)
static constexpr int ROW_BLOCK = 512;

// Long rows: one block per row ranks the survivors by tree face id with an LDS bitmap.
// The id space [0, S) is processed in chunks of BM_BITS ids (one chunk covers a million tree faces): set one
// bit per survivor, count the bits of each thread's segment of BM_SEG words (rotated start -> bank-conflict
// free), scan the 256 segment totals, rank = (survivors in earlier chunks) + segment base + popcounts of
// the segment's words below + popc(bits below).  O(n + S/32) per row instead of O(n^2); any row length,
// any number of tree faces.  The row is read from HBM once, four candidates in flight per thread, and
// parked in LDS (survivor id or -1) for the later passes: the kernel is a chain of dependent phases on a
// handful of rows, i.e. pure latency.
static constexpr int BM_WORDS = 32768;          // 128 KiB of bitmap
static constexpr int BM_BITS = BM_WORDS * 32;   // ids per chunk
static constexpr int BM_SEG = BM_WORDS / 256;   // words per thread segment
static constexpr int BM_STAGE = 4096;           // candidates parked in LDS (16 KiB); longer rows re-read HBM
static constexpr size_t ROW_FILL_LIGHT_LDS = sizeof(int32_t) * (8 + BM_STAGE); // LDS of a k_row_fill_long launch with row_class = 1

__global__ void __launch_bounds__(256)
k_row_fill_long(const int32_t *__restrict__ cand_off, const int32_t *__restrict__ cand_count,
                const int32_t *__restrict__ cand_sid,
                const double *__restrict__ cand_area, const int32_t *__restrict__ indptr,
                const double *__restrict__ src_area, bool relative, int64_t n_tree, int32_t *__restrict__ indices,
                double *__restrict__ data, const int32_t *__restrict__ long_rows,
                const int32_t *__restrict__ n_long, int64_t row_base /* >= 0: list entry li is stored row row_base + li */,
                const int32_t *__restrict__ skip_if /* optional: nothing is done when this device word is > 0 */,
                const int32_t *__restrict__ scan_nnz = nullptr /* optional: row lengths per FACE; the offsets of the listed rows
                                                                  are then computed here (exclusive scan in list order) */,
                int32_t *__restrict__ scan_indptr = nullptr /* [n_long + 1], written */,
                int32_t *__restrict__ scan_total = nullptr /* total entries, written */,
                int row_class = 0 /* 0: every listed row; 1: only rows of at most ROW_BLOCK candidates -- the launch then needs
                                     ROW_FILL_LIGHT_LDS bytes of LDS instead of 150 KB, so its blocks find room beside other
                                     kernels and every row gets a block of its own; 2: only the rows beyond ROW_BLOCK */) {
    __builtin_amdgcn_s_setprio(3); // side-stream kernel: its waves go first in the SIMDs' issue arbitration (see overlap_tri)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (scratch and stage first: a row_class = 1 launch allocates nothing behind them)
    int32_t *red = reinterpret_cast<int32_t *>(smem);           // [8] scratch
    int32_t *stage = red + 8;                                   // [BM_STAGE]
    uint32_t *tbase = reinterpret_cast<uint32_t *>(stage + BM_STAGE); // [256] exclusive over thread segments
    uint32_t *bm = tbase + 256;                                 // [BM_WORDS]
    uint16_t *gcnt = reinterpret_cast<uint16_t *>(bm + BM_WORDS); // [BM_WORDS / 8] bits per 8 words -> prefix in segment
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (skip_if && *skip_if > 0) return;
    const int nl = *n_long;
    // block-wide sum of the lengths of the listed rows [k0, k1) (scan_nnz mode)
    auto length_sum = [&](int k0, int k1) -> long long {
        long long v = 0;
        for (int k = k0 + tid; k < k1; k += 256) v += scan_nnz[long_rows[k]];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        __syncthreads();
        if (lane == 0) red[wave] = (int32_t)(v > 0x7fffffffll ? 0x7fffffff : v);
        __syncthreads();
        return (long long)red[0] + red[1] + red[2] + red[3];
    };
    long long scan_base = 0; // offset of row blockIdx.x, advanced by gridDim.x rows per trip
    if (scan_nnz) {
        scan_base = length_sum(0, blockIdx.x < nl ? (int)blockIdx.x : nl);
        if (blockIdx.x == 0 && nl == 0 && tid == 0) {
            scan_indptr[0] = 0;
            *scan_total = 0;
        }
    }
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {
        const int t = long_rows[li];
        const int c0 = cand_off[t], n = cand_count[t];
        int base;
        if (scan_nnz) {
            const long long capped = scan_base > 0x7fffffffll ? 0x7fffffffll : scan_base;
            base = (int)capped;
            if (tid == 0) {
                scan_indptr[li] = base;
                if (li == nl - 1) { // the last listed row closes the offsets
                    const long long all = capped + scan_nnz[t];
                    scan_indptr[nl] = (int32_t)(all > 0x7fffffffll ? 0x7fffffff : all);
                    *scan_total = scan_indptr[nl];
                }
            }
            const int next = li + (int)gridDim.x;
            scan_base += length_sum(li, next < nl ? next : nl);
        } else {
            base = row_base >= 0 ? indptr[row_base + li] : indptr[t];
        }
        __syncthreads();
        if (row_class != 0 && (row_class == 1) != (n <= ROW_BLOCK)) continue; // (uniform) the other launch's row
        // park the row: survivor id or -1 (four independent pairs of loads per thread and trip)
        for (int i0 = tid; i0 < n && i0 < BM_STAGE; i0 += 4 * 256) {
            double a[4];
            int sd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 256;
                const bool in = i < n && i < BM_STAGE;
                a[u] = in ? cand_area[c0 + i] : 0.0;
                sd[u] = in ? cand_sid[c0 + i] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 256;
                if (i < n && i < BM_STAGE) stage[i] = a[u] > 0 ? sd[u] : -1;
            }
        }
        if (n <= ROW_BLOCK) { // all-pairs rank; dead candidates compare as +inf
            __syncthreads();
            const int n4 = (n + 3) & ~3;
            if (tid < n4 - n) stage[n + tid] = -1;
            __syncthreads();
            const int4 *quad = reinterpret_cast<const int4 *>(stage);
            for (int i = tid; i < n; i += 256) {
                const int sd = stage[i];
                if (sd < 0) continue;
                int rank = 0;
                for (int j = 0; j < n4 / 4; j++) {
                    const int4 q = quad[j];
                    rank += ((unsigned)q.x < (unsigned)sd) + ((unsigned)q.y < (unsigned)sd) + ((unsigned)q.z < (unsigned)sd) +
                            ((unsigned)q.w < (unsigned)sd);
                }
                const double a = cand_area[c0 + i];
                indices[base + rank] = sd;
                data[base + rank] = relative ? a / src_area[sd] : a;
            }
            continue;
        }
        auto survivor = [&](int i) -> int { // tree face id of candidate i of the row, -1 if its area is not positive
            if (i < BM_STAGE) return stage[i];
            return cand_area[c0 + i] > 0 ? cand_sid[c0 + i] : -1;
        };
        int running = 0;
        for (int64_t cb64 = 0; cb64 < n_tree; cb64 += BM_BITS) {
            const int cb = (int)cb64;
            __syncthreads();
            {
                uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
                for (int w = tid; w < BM_WORDS / 4; w += 256) bm4[w] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                const int s = survivor(i) - cb;
                if (s >= 0 && s < BM_BITS) atomicOr(&bm[s >> 5], 1u << (s & 31));
            }
            __syncthreads();
            // bits per group of 8 words (coalesced 32-byte reads), then an exclusive prefix over the 16 groups
            // of each thread's segment
            for (int g = tid; g < BM_WORDS / 8; g += 256) {
                const uint4 x = reinterpret_cast<const uint4 *>(bm)[2 * g], y = reinterpret_cast<const uint4 *>(bm)[2 * g + 1];
                gcnt[g] = (uint16_t)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) +
                                     __popc(y.z) + __popc(y.w));
            }
            __syncthreads();
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < BM_SEG / 8; k++) {
                const uint32_t c = gcnt[tid * (BM_SEG / 8) + k];
                gcnt[tid * (BM_SEG / 8) + k] = (uint16_t)acc;
                acc += c;
            }
            // block exclusive scan of the per-thread totals
            uint32_t incl = acc;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            if (lane == 63) red[wave] = (int32_t)incl;
            __syncthreads();
            uint32_t woff = 0, chunk_total = 0;
            for (int w = 0; w < 4; w++) {
                if (w < wave) woff += (uint32_t)red[w];
                chunk_total += (uint32_t)red[w];
            }
            tbase[tid] = woff + incl - acc;
            __syncthreads();
            if (chunk_total != 0) {
                for (int i = tid; i < n; i += 256) {
                    const int sid = survivor(i);
                    const int s = sid - cb;
                    if (sid >= 0 && s >= 0 && s < BM_BITS) {
                        const int w = s >> 5;
                        int rank = running + (int)tbase[w / BM_SEG] + (int)gcnt[w >> 3] +
                                   __popc(bm[w] & ((1u << (s & 31)) - 1u));
                        for (int k = w & ~7; k < w; k++) rank += __popc(bm[k]);
                        const double a = cand_area[c0 + i];
                        indices[base + rank] = sid;
                        data[base + rank] = relative ? a / src_area[sid] : a;
                    }
                }
            }
            running += (int)chunk_total;
        }
        __syncthreads();
    }
}

} // namespace xr
#include "xr_overlap_fused.h"
namespace xr {

static bool debug_fused() { return (option(OPT_DEBUG) & 1) != 0; }
static int xcd_remap_mask() {
    // bit 0 clip, bit 1 search, bit 2 row_fill: the XCD-aware block order for clip + search -- 15 % fewer HBM bytes fetched by
    // both (PMC: clip 161 -> 130 MB, search 56 -> 48 MB per launch) at an unchanged clip time and a 5 % shorter search;
    // row_fill loses more on the then interleaved queue than it gains.
    return 3;
}
static unsigned xcd_grid(int64_t n_blocks, bool remap) { return (unsigned)(remap ? (n_blocks + 7) / 8 * 8 : n_blocks); }

// Upper bound, for every pair of faces of the two meshes, of the area the clip can return for a pair the reference's pre-clip
// tests reject (pair_passes_box_and_sat).  Such a pair's true intersection is at most ~3 ulp(coordinate) deep (the SAT's
// projections) and the clip's crossing points are off by a few ulp(coordinate) more: a sliver of width <= ~8 eps |coordinate|
// along at most half the perimeter of the smaller face.  With perimeter <= 4 extents: 16 eps M ext; the threshold is four
// times that (64 eps = 1.4e-14), M the largest coordinate magnitude, ext the smaller of the two meshes' largest face extents
// (both statistics are on the host when the clip is launched; a sampled tree statistic is bounded by the domain).  The
// smaller the threshold the fewer lanes fetch their vertices again: on the 1M benchmark 1e-12 M ext cost the clip 7 us.
static double overlap_dust_threshold(const xr_mesh *tree, const xr_mesh *query) {
    double mag = 0.0, ext = INFINITY;
    for (const xr_mesh *m : {tree, query}) {
        const double *h = m->h_stats;
        mag = std::max(mag, std::max(std::max(fabs(h[0]), fabs(h[1])), std::max(fabs(h[2]), fabs(h[3]))));
        ext = std::min(ext, m->stats_sampled ? std::max(h[1] - h[0], h[3] - h[2]) : h[5]);
    }
    return option(OPT_DUST) == 0 ? 0.0 : 5.7e-14 * mag * ext; // (option "dust" = 0: measurement switch, no confirmation)
}

template <int MAXV, int BLOCK>
static void launch_clip(const xr_mesh *tree, const xr_mesh *query, const int32_t *cand_tgt, const int32_t *cand_src,
                        int64_t C, double *cand_area, bool redo_only, int32_t *cand_sid, int32_t *overflow_count,
                        int32_t *nnz_row) {
    const size_t shmem = (size_t)2 * MAXV * BLOCK * sizeof(double2);
    static bool attr_set = false;
    if (!attr_set) {
        XR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_clip<MAXV, BLOCK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    XR_LAUNCH(MAXV == 8 ? "clip_v8" : (MAXV == 16 ? "clip_v16" : "clip_v64"), (k_clip<MAXV, BLOCK>),
              dim3(div_up(C, BLOCK)), dim3(BLOCK), shmem, query->qo_fxy(), query->qo_len(), query->qo_off(), query->m,
              query->qo_perm(), tree->rec_fxy.get(), tree->rec_len.get(), tree->record_off(), tree->m, cand_tgt, cand_src, C, cand_area,
              redo_only, tree->rec_face.get(), cand_sid, overflow_count, nnz_row, overlap_dust_threshold(tree, query));
}

static void launch_clip_for(const xr_mesh *tree, const xr_mesh *query, const int32_t *cand_tgt, const int32_t *cand_src,
                            int64_t C, double *cand_area, int32_t *cand_sid, int32_t *overflow_count,
                            int32_t *nnz_row) {
    const int vmax = query->m + tree->m;
    const bool remap = xcd_remap_mask() & 1;
    const double dust = overlap_dust_threshold(tree, query);
    if (vmax <= 6) {
        // triangle x triangle: the clipped polygon never has more than 6 vertices
        constexpr int MAXV = 6, BLOCK = 256;
        const size_t shmem = (size_t)(MAXV + 1) * BLOCK * sizeof(double2); // + one trash row for clamped pushes
        XR_LAUNCH("clip_tri", (k_clip_tri<BLOCK>), dim3(xcd_grid(div_up(C, BLOCK), remap)), dim3(BLOCK), shmem,
                      query->qo_fxy(), query->m, tree->rec_fxy.get(), tree->m, cand_tgt, cand_src, C, cand_area,
                      tree->rec_face.get(), cand_sid, overflow_count, nnz_row, remap, dust);
    } else if (query->m == 4 && tree->m == 3 && option(OPT_CLIP_QUAD) != 0) {
        // quadrilateral targets (a raster) x triangle source (option "clip_quad" = 0: the slot-loop kernel, test switch)
        constexpr int BLOCK = 256;
        const size_t shmem = (size_t)(QUAD_MAXV + 2) * BLOCK * sizeof(double2);
        XR_LAUNCH("clip_quad_tri", (k_clip_quad_tri<BLOCK>), dim3(xcd_grid(div_up(C, BLOCK), remap)), dim3(BLOCK), shmem,
                  query->qo_fxy(), query->qo_len(), tree->rec_fxy.get(), cand_tgt, cand_src, C, cand_area, tree->rec_face.get(),
                  cand_sid, overflow_count, nnz_row, remap, dust);
    } else if (vmax <= 8) {
        constexpr int MAXV = 8, BLOCK = 256;
        const size_t shmem = (size_t)(MAXV + 1) * BLOCK * sizeof(double2); // + one trash row for clamped pushes
        XR_LAUNCH("clip_small", (k_clip_small<MAXV, BLOCK, false>), dim3(xcd_grid(div_up(C, BLOCK), remap)), dim3(BLOCK),
                  shmem, query->qo_fxy(), query->qo_len(), query->qo_off(), query->m, query->qo_perm(), tree->rec_fxy.get(),
                  tree->rec_len.get(), tree->record_off(), tree->m, cand_tgt, cand_src, C, cand_area, tree->rec_face.get(), cand_sid,
                  overflow_count, nnz_row, remap, dust);
    }
    else if (vmax <= 16) launch_clip<16, 128>(tree, query, cand_tgt, cand_src, C, cand_area, false, cand_sid, overflow_count, nnz_row);
    else launch_clip<64, 64>(tree, query, cand_tgt, cand_src, C, cand_area, false, cand_sid, overflow_count, nnz_row);
}

// Triangle x triangle pairs (xr_overlap_fused.h): search -> persistent clip -> assembly with a look-back scan, the big
// faces of the search (hull slivers ...) on a side stream, their rows stored behind the regular ones; ONE host round
// trip at the very end (sizes + error bits).  -> false if the pair has to go through the general pipeline after all
// (a clip that needs more than 6 vertices: floating-point degenerate input; or the look-back chain gave up).
// An apply the caller wants right behind the weight build (xr_overlap_apply_dev).  The triangle pipeline enqueues it BEFORE
// the host has read the sizes back -- the K = 1 apply kernel takes everything it needs from device memory -- so the one host
// round trip of the build is hidden behind the apply instead of standing between the two.
struct EarlyApply {
    std::function<void(const xr_csr *)> fn;
    bool done = false;
};

static bool overlap_tri(xr_mesh *tree, xr_mesh *query, const double *tree_area, bool relative, xr_csr *csr,
                        const MortonParams &tile, EarlyApply *early) {
    const int64_t T = query->n_face;
    const GridParams &g = tree->grid;
    hipStream_t st = launch_stream();
    const int64_t n_blocks = div_up(T, FB);
    const bool remap = xcd_remap_mask() & 2;
    const unsigned grid = xcd_grid(n_blocks, remap);
    const int64_t margin_opt = option(OPT_QUEUE_MARGIN); // test hook: a tiny margin forces the regrow path
    int64_t big_capacity = margin_opt > 0 ? margin_opt : ((int64_t)4 << 20);
    // the regular pair queue: eight regions (one per XCD) of region_cap pairs each -- every block of 256 faces parks at most
    // 256 x SLOTS pairs and an XCD takes ceil(n_blocks / 8) blocks
    const int64_t region_cap = div_up(div_up(T, FB), 8) * FB * SLOTS;
    const int64_t reg_capacity = 8 * region_cap;
    int64_t per_face = 8; // CSR entries reserved per target face (+ the big queue); regrown if the matrix is denser
    int32_t *mail = const_cast<int32_t *>(engine().mailbox);
    static_assert(sizeof(FusedCounters) == 32, "FusedCounters layout");
    // ctl: [0] regular queue cursor, [1] big queue cursor, [2] big faces, [3] big faces that did not fit, [4] clip
    // overflows among the big pairs | FusedCounters
    // | per block of 256 target faces: survivors (clip), regular faces (search)
    // The 16 counter words come from the engine's zero-at-rest scratch (k_publish_all clears them again after copying them
    // to the mailbox) and k_search clears its block's survivor count: no memset in front of the search.
    constexpr size_t CTL_HEAD = QCUR_BASE + 8 * QCUR_STRIDE; // (the eight region cursors of the regular queue behind the 16 counters: a line each)
    DevBuf<int32_t> ctl_tail(2 * (size_t)grid), ctl_own;
    int32_t *blk_surv = ctl_tail.get(), *blk_rows = blk_surv + grid;
    DevBuf<int32_t> blk_base(2 * (size_t)grid); // first stored row / CSR base of every hardware block (k_assemble_scan)
    // Row bases of the assembly: every assembly block sums the counts of the blocks in front of it itself (scan_mode 2) -- it reads
    // b words in block b, O(blocks^2) L2 reads in total: fine for a few thousand blocks (1M faces: 62 MB), 6 GB for the 39k blocks
    // of a 10M-face target (measured: 10M -> 10M 4.39 -> 4.89 ms) -- so from 8192 blocks on a one-block scan kernel between clip and
    // assembly provides them (scan_mode 1, as until round 3).  (Until round 5 a decoupled look-back inside k_assemble was a third
    // form, behind a switch; it lost in round 3 and is gone.)
    const int scan_mode = grid <= 8192 ? 2 : 1;
    DevBuf<int32_t> cand_count((size_t)T), cand_off((size_t)T + 1), big_list((size_t)T), pending((size_t)T), nnz_row((size_t)T),
        slot_face((size_t)T), big_indptr((size_t)T + 1);
    DevBuf<uint8_t> is_big((size_t)T);
    DevBuf<int2> block_seg((size_t)n_blocks);
    // (measured and removed: the clip writing only its survivors, compacted in place over the queue -- the HBM bytes drop, but the
    // assembly becomes three passes of dependent loads, 0.066 -> 0.087 ms, DESIGN_HISTORY.md round 3)
    DevBuf<int32_t> cand_tgt((size_t)reg_capacity), cand_src((size_t)reg_capacity), cand_sid((size_t)reg_capacity);
    DevBuf<double> cand_area((size_t)reg_capacity);
    csr->n_long.alloc(2); // [0] rows of more than XR_APPLY_LONG_ROW entries, [1] gate of an apply enqueued behind the build
    csr->row_order.alloc((size_t)T);
    csr->has_row_order = true;
    const int big_grid = engine().num_cu * 8;
    constexpr int CLIP_BLOCK = 256;
    // triangle x triangle: the flag / compaction clip of xr_clip_tri.h; any other pair of dense meshes (<= 4 nodes per face:
    // quadrilaterals, mixed meshes with fill values, a raster against triangles): the register / LDS clip of k_clip_small<8>
    const bool tri_pair = tree->m == 3 && query->m == 3;
    const size_t clip_shmem = tri_pair ? (size_t)(TRI_MAXV + 1) * CLIP_BLOCK * sizeof(double2) : (size_t)(8 + 1) * CLIP_BLOCK * sizeof(double2);
    const double dust = overlap_dust_threshold(tree, query);
    const size_t fill_shmem = sizeof(uint32_t) * (BM_WORDS + 256) + sizeof(int32_t) * (8 + BM_STAGE) + sizeof(uint16_t) * (BM_WORDS / 8);
    static bool attr_set = false;
    if (!attr_set) {
        XR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_fill_long),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fill_shmem));
        attr_set = true;
    }
    for (int attempt = 0;; attempt++) {
        XR_REQUIRE(attempt < 8, XR_ERR_LIMIT, "xr_overlap: the weight matrix does not fit the device buffers");
        const int64_t cap = per_face * T + big_capacity;
        XR_REQUIRE(cap < ((int64_t)1 << 31), XR_ERR_LIMIT, "nnz exceeds the int32 range");
        csr->indices.alloc((size_t)cap);
        csr->data.alloc((size_t)cap);
        csr->long_rows.alloc((size_t)(cap / XR_APPLY_LONG_ROW + 1));
        DevBuf<int32_t> big_tgt((size_t)big_capacity), big_src((size_t)big_capacity), big_sid((size_t)big_capacity),
            big_indices((size_t)big_capacity);
        DevBuf<double> big_area((size_t)big_capacity), big_data((size_t)big_capacity);
        int32_t *ctl_head = zero_scratch(1, CTL_HEAD);
        const bool ctl_cached = ctl_head != nullptr;
        if (!ctl_cached) {
            ctl_own.alloc(CTL_HEAD);
            ctl_head = ctl_own.get();
            XR_HIP(hipMemsetAsync(ctl_head, 0, CTL_HEAD * sizeof(int32_t), st));
        }
        FusedCounters *fc = reinterpret_cast<FusedCounters *>(ctl_head + 8);
#define XR_SEARCH_LAUNCH(PACKED)                                                                                                    \
    XR_LAUNCH("search", (k_search<PACKED>), dim3(grid), dim3(256), 0, query->qo_bbox(), T, g, tree->n_face,                            \
              tree->cell_start.get(), tree->rec_bb.get(), cand_count.get(), cand_off.get(), cand_tgt.get(), cand_src.get(),          \
              ctl_head + QCUR_BASE, block_seg.get(), is_big.get(), big_list.get(), ctl_head + 2, tile, (int32_t *)nullptr, nnz_row.get(),   \
              remap, blk_rows, blk_surv, (int)region_cap)
        // (the owner byte rides in the record id's top byte where the tree has at most 2^24 faces)
        if (tree->n_face <= ((int64_t)1 << 24)) XR_SEARCH_LAUNCH(true);
        else XR_SEARCH_LAUNCH(false);
#undef XR_SEARCH_LAUNCH
        {
            // ---- side stream: everything about the big faces except their final placement
            SideScope side;
            XR_LAUNCH("search_big", k_search_big<true>, dim3(big_grid), dim3(256), sizeof(int32_t) * (size_t)big_stage_entries(),
                      query->qo_bbox(), query->qo_fxy(),
                      query->qo_len(), query->qo_off(), query->m, g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_face.get(),
                      big_list.get(), ctl_head + 2, cand_off.get(), cand_count.get(), big_tgt.get(), big_src.get(),
                      ctl_head + 1, big_capacity, pending.get(), ctl_head + 3, big_stage_entries(), slot_face.get());
            // (pairs of a face that did not fit are missing: the error is seen at the end and everything is redone)
            if (tri_pair)
                XR_LAUNCH("clip_big", (k_clip_tri_queue<CLIP_BLOCK, 2>), dim3(engine().num_cu), dim3(CLIP_BLOCK), clip_shmem,
                          query->qo_fxy(), tree->rec_fxy.get(), tree->rec_face.get(), big_tgt.get(), big_src.get(), ctl_head + 1,
                          big_capacity, big_area.get(), big_sid.get(), &fc->error, nnz_row.get(), ctl_head + 3, (int32_t *)nullptr,
                          dust);
            else
                XR_LAUNCH("clip_big", (k_clip_tri_queue<CLIP_BLOCK, 2, false, 1>), dim3(engine().num_cu), dim3(CLIP_BLOCK), clip_shmem,
                          query->qo_fxy(), tree->rec_fxy.get(), tree->rec_face.get(), big_tgt.get(), big_src.get(), ctl_head + 1,
                          big_capacity, big_area.get(), big_sid.get(), &fc->error, nnz_row.get(), ctl_head + 3, (int32_t *)nullptr,
                          dust, query->qo_len(), tree->rec_len.get(), query->m, tree->m);
            // (rows in face order: ranked inside search_big; their offsets: scanned inside row_fill_long -- two launches less
            // in what is the critical path of the whole weight build)
            // Two launches: the rows of at most ROW_BLOCK candidates (all but a handful) with 16 KB of LDS per block -- a block
            // per row, resident beside the clip and the assembly -- and the few longer ones with the bitmap (150 KB, a CU per
            // block).  As ONE launch every block needed a whole CU: it could not start before the clip's blocks had left and
            // then walked ~5 rows in turn -- 54-64 us, ending after the assembly (round-4 timeline).  XR_ROWFILL_SPLIT=0: one launch.
            // (round 6: the two launches are independent -- each scans the row lengths itself and fills only its own class of
            // rows -- so the bitmap rows go to a SECOND side stream and run beside the light ones: the big faces' chain
            // behind the clip is one launch shorter where it is the step's critical path, 0.506 -> 0.500 ms in an A/B)
            SideForkScope fork;
            XR_LAUNCH("row_fill_huge", k_row_fill_long, dim3(engine().num_cu), dim3(256), fill_shmem, cand_off.get(),
                      cand_count.get(), big_sid.get(), big_area.get(), big_indptr.get(), tree_area, relative,
                      tree->n_face, big_indices.get(), big_data.get(), slot_face.get(), ctl_head + 2, (int64_t)0, ctl_head + 3,
                      nnz_row.get(), big_indptr.get(), &fc->p_big, 2);
            fork.end_launches();
            XR_LAUNCH("row_fill_long", k_row_fill_long, dim3(engine().num_cu * 6), dim3(256), ROW_FILL_LIGHT_LDS, cand_off.get(),
                      cand_count.get(), big_sid.get(), big_area.get(), big_indptr.get(), tree_area, relative,
                      tree->n_face, big_indices.get(), big_data.get(), slot_face.get(), ctl_head + 2, (int64_t)0, ctl_head + 3,
                      nnz_row.get(), big_indptr.get(), &fc->p_big, 1);
        }
        // Persistent blocks per CU: five fill the LDS and the register files (best for the clip alone).  The big faces' chain
        // on the side stream then gets few wave slots while the clip runs; with its kernels at raised wave priority and only
        // three launches long (search_big -> clip -> row_fill_long) it still ends about when k_assemble does: 3, 4 and 5
        // blocks per CU all give the same step (0.635 ms) -- measured after the chain lost two launches; before, 3 was
        // 4 % faster because the chain was the critical path.
        // (slot-major LDS columns -- k_clip_tri_queue's SOA form -- were measured in round 3 and lost; the template parameter is all
        // that is left of them)
        if (!tri_pair)
            // (k_clip_small's LDS columns: 36 KB per block, four persistent blocks per CU)
            XR_LAUNCH("clip_small", (k_clip_tri_queue<CLIP_BLOCK, 1, false, 1>), dim3(engine().num_cu * 4), dim3(CLIP_BLOCK), clip_shmem,
                      query->qo_fxy(), tree->rec_fxy.get(), tree->rec_face.get(), cand_tgt.get(), cand_src.get(),
                      ctl_head + QCUR_BASE, -region_cap, cand_area.get(), cand_sid.get(), &fc->error, (int32_t *)nullptr,
                      (const int32_t *)nullptr, blk_surv, dust, query->qo_len(), tree->rec_len.get(), query->m, tree->m);
        else
            XR_LAUNCH("clip_tri", (k_clip_tri_queue<CLIP_BLOCK, 1, false>), dim3(engine().num_cu * 5), dim3(CLIP_BLOCK), clip_shmem,
                      query->qo_fxy(), tree->rec_fxy.get(), tree->rec_face.get(), cand_tgt.get(), cand_src.get(),
                      ctl_head + QCUR_BASE, -region_cap, cand_area.get(), cand_sid.get(), &fc->error, (int32_t *)nullptr,
                      (const int32_t *)nullptr, blk_surv, dust);
        if (scan_mode == 1)
            XR_LAUNCH("assemble_scan", k_assemble_scan, dim3(1), dim3(1024), 0, blk_rows, blk_surv, (int64_t)n_blocks, (int)grid,
                      remap, blk_base.get(), blk_base.get() + grid, fc, csr->indptr.get());
        XR_LAUNCH("assemble", k_assemble, dim3(grid), dim3(FB), 0, query->qo_bbox(), query->qo_perm(), T, cand_tgt.get(),
                  cand_off.get(), cand_count.get(), block_seg.get(), is_big.get(), cand_area.get(), cand_sid.get(),
                  tree_area, relative, tile, csr->has_tile_key ? csr->tile_key.get() : (int32_t *)nullptr, fc,
                  csr->indptr.get(), csr->indices.get(), csr->data.get(), csr->row_order.get(),
                  csr->long_rows.get(), cap, remap, scan_mode == 1 ? blk_base.get() : (const int32_t *)nullptr,
                  scan_mode == 1 ? blk_base.get() + grid : (const int32_t *)nullptr,
                  scan_mode == 2 ? blk_rows : (const int32_t *)nullptr, scan_mode == 2 ? blk_surv : (const int32_t *)nullptr);
        side_join();
        XR_LAUNCH("place_big", k_place_big, dim3(64), dim3(256), 0, ctl_head + 2, slot_face.get(), big_indptr.get(),
                  big_indices.get(), big_data.get(), T, query->qo_perm(), query->qo_bbox(), tile,
                  csr->has_tile_key ? csr->tile_key.get() : (int32_t *)nullptr, fc, csr->indptr.get(), csr->indices.get(),
                  csr->data.get(), csr->row_order.get(), csr->long_rows.get(), cap, ctl_head + 3);
        host_stamp(5);
        const int32_t seq = mailbox_next_seq();
        XR_LAUNCH("publish", k_publish_all, dim3(1), dim3(64), 0, ctl_head, fc, csr->n_long.get(), mail, seq, cap, big_capacity);
        if (ctl_cached) zero_scratch_done(1); // (the counters are zero again behind k_publish_all)
        if (early) {
            // sizes unknown on the host yet: pessimistic flags (long rows possible, of any length) -- they only add blocks
            // that look at the device-side list of long rows and find it short or empty; results do not depend on them
            csr->nnz = 0;
            csr->has_long = true;
            csr->max_row_len = -1;
            csr->apply_gated = true; // (the kernel looks at the gate k_publish_all has just set: a failed attempt is skipped)
            early->fn(csr);
            csr->apply_gated = false;
        }
        host_stamp(6);
        mailbox_wait_seq(seq);
        host_stamp(7);
        const int32_t C_reg = mail[0], C_big = mail[1], n_big = mail[2], n_pending = mail[3], big_overflow = mail[4];
        const int32_t err = mail[5], rows_regular = mail[6], p_regular = mail[8], p_big = mail[9];
        XR_REQUIRE(C_reg >= 0 && C_big >= 0, XR_ERR_LIMIT, "candidate pair count exceeds the int32 range");
        if (debug_fused())
            fprintf(stderr, "[tri] T=%lld C=%d big: %d faces %d pairs (%d pending) p_regular=%d p_big=%d err=%d rows=%d long=%d regions %d %d %d %d %d %d %d %d of %lld\n",
                    (long long)T, C_reg, n_big, C_big, n_pending, p_regular, p_big, err, rows_regular, mail[7], mail[11], mail[12], mail[13],
                    mail[14], mail[15], mail[16], mail[17], mail[18], (long long)region_cap);
        (void)big_overflow;
        if (err & 1) return false;
        if (n_pending > 0 || C_big > big_capacity) { // some big faces needed more room than the margin
            big_capacity = (int64_t)C_big + 1024;
            continue;
        }
        if ((err & 4) || (int64_t)p_regular + p_big > cap) {
            per_face *= 2;
            continue;
        }
        XR_REQUIRE(rows_regular == T - n_big, XR_ERR_INVALID, "xr_overlap: internal row count mismatch");
        tree->last_candidates = (int64_t)C_reg + C_big;
        csr->nnz = (int64_t)p_regular + p_big;
        csr->has_long = mail[7] > 0;   // rows of more than XR_APPLY_LONG_ROW entries (none: the apply skips their kernels)
        csr->max_row_len = mail[7] > 0 ? mail[10] : XR_APPLY_LONG_ROW;
        if (early) early->done = true; // (the apply enqueued in THIS attempt saw the final matrix)
        return true;
    }
}

static void overlap(xr_mesh *tree, xr_mesh *query, bool relative, xr_csr *csr, EarlyApply *early = nullptr) {
    // (the query side on the side stream next to the tree side was measured: the prepare kernels are bandwidth bound and
    // just slow each other down, 0.684 -> 0.706 ms per step)
    // the tree side only needs its records (built from the raw mesh).  Its statistics go to the host through a one-block
    // kernel that ends with writes to pinned memory (~20 us): on the side stream as well, beside the preparation of the
    // query mesh, which is what the host waits behind anyway
    mesh_prepare(tree, false, /*stats_on_side=*/true, /*allow_sampled=*/true); // (bounds exact, mean extent sampled)
    host_stamp(1);
    mesh_prepare(query, true, /*stats_on_side=*/true); // (its statistics are first read by mesh_query_order below)
    host_stamp(2);
    mesh_build_index(tree);
    host_stamp(3);
    mesh_query_order(query);
    host_stamp(4);
    const double *tree_area = relative ? mesh_area(tree) : nullptr;
    const int64_t T = query->n_face, S = tree->n_face;
    XR_REQUIRE(T < ((int64_t)1 << 31) - 1, XR_ERR_LIMIT, "too many query faces for int32 row offsets");
    csr->n = T;
    csr->m = S;
    csr->nnz = 0;
    csr->indptr.alloc((size_t)T + 1);
    tree->last_candidates = 0;
    if (T == 0 || S == 0) {
        XR_HIP(hipMemsetAsync(csr->indptr.get(), 0, sizeof(int32_t) * ((size_t)T + 1), launch_stream()));
        csr->indices.alloc(0);
        csr->data.alloc(0);
        return;
    }
    const GridParams &g = tree->grid;
    hipStream_t st = launch_stream();
    // Rows kept in the caller's (coherent, but typically strip-like) numbering get a coarse Morton key each:
    // tiles of 12-24 mean target extents, runs of TILE_RUN consecutive rows kept together.  A Morton-sorted query order is tiled already.
    MortonParams tile{};
    csr->has_tile_key = false;
    if (query->query_identity) {
        const double *hs = query->h_stats;
        double span = std::max(hs[1] - hs[0], hs[3] - hs[2]);
        if (!(span > 0)) span = 1.0;
        // tile edge <= 12 mean extents (the side count is a power of two): about one block of 256 triangles per tile
        double h = 12.0 * hs[4] / (double)T;
        if (!(h > 0)) h = span;
        int bits = 0;
        while (bits < 12 && ldexp(h, bits) < span) bits++;
        tile = MortonParams{hs[0], hs[2], (double)(1 << bits) / (span * (1.0 + 1e-9)), 1 << bits};
        tile.n_run = TILE_RUN;
        csr->tile_key.alloc((size_t)T);
        csr->tile_key_range = (int64_t)1 << (2 * bits);
        csr->has_tile_key = bits > 0;
    }
    {
        // triangle x triangle: the single-round-trip pipeline of xr_overlap_fused.h (option "overlap_fused" = 0: test /
        // measurement switch back to the general kernel chain)
        const bool fused_on = option(OPT_OVERLAP_FUSED) != 0;
        if (fused_on && tree->m <= DENSE_MAX_NODES && query->m <= DENSE_MAX_NODES && (T + 8 * FB) * SLOTS < ((int64_t)1 << 31)) {
            if (overlap_tri(tree, query, tree_area, relative, csr, tile, early)) return;
            csr->has_row_order = false; // (the general pipeline below stores the rows in query order)
        }
    }
    // counters: [0] clip overflow, [1] number of long rows, [2] number of big query faces, [3] pair-queue cursor
    DevBuf<int32_t> counters(4);
    XR_HIP(hipMemsetAsync(counters.get(), 0, sizeof(int32_t) * 4, st));
    // --- candidate search.  k_search appends the candidates of its 256 faces to the pair queue itself (one
    // atomic reservation per block); the few "big" faces are counted, reserve their stretch, and are filled by
    // the block-per-face kernels.  The queue is sized for the regular faces (at most SLOTS candidates each) plus
    // a margin for the big ones; if the big faces need more, it is regrown before they are filled (rare).
    DevBuf<int32_t> cand_count((size_t)T), cand_off((size_t)T + 1), big_list((size_t)T);
    DevBuf<uint8_t> is_big((size_t)T);
    const int big_grid = engine().num_cu * 8;
    const int64_t n_blocks = div_up(T, 256);
    DevBuf<int2> block_seg((size_t)n_blocks);
    const int64_t margin_opt = option(OPT_QUEUE_MARGIN); // test hook: a tiny margin forces the regrow path
    int64_t capacity = T * SLOTS + (margin_opt > 0 ? margin_opt : ((int64_t)4 << 20));
    XR_REQUIRE(capacity < ((int64_t)1 << 31), XR_ERR_LIMIT, "candidate pair queue exceeds the int32 range");
    DevBuf<int32_t> cand_tgt((size_t)capacity), cand_src((size_t)capacity), nnz_row((size_t)T);
    const bool remap_search = xcd_remap_mask() & 2, remap_rows = xcd_remap_mask() & 4;
    if (tree->n_face <= ((int64_t)1 << 24))
        XR_LAUNCH("search", (k_search<true>), dim3(xcd_grid(div_up(T, 256), remap_search)), dim3(256), 0, query->qo_bbox(), T, g,
                  tree->n_face, tree->cell_start.get(), tree->rec_bb.get(), cand_count.get(), cand_off.get(), cand_tgt.get(),
                  cand_src.get(), counters.get() + 3, block_seg.get(), is_big.get(), big_list.get(), counters.get() + 2, tile,
                  csr->has_tile_key ? csr->tile_key.get() : (int32_t *)nullptr, nnz_row.get(), remap_search);
    else
        XR_LAUNCH("search", (k_search<false>), dim3(xcd_grid(div_up(T, 256), remap_search)), dim3(256), 0, query->qo_bbox(), T, g,
                  tree->n_face, tree->cell_start.get(), tree->rec_bb.get(), cand_count.get(), cand_off.get(), cand_tgt.get(),
                  cand_src.get(), counters.get() + 3, block_seg.get(), is_big.get(), big_list.get(), counters.get() + 2, tile,
                  csr->has_tile_key ? csr->tile_key.get() : (int32_t *)nullptr, nnz_row.get(), remap_search);
    DevBuf<int32_t> pending((size_t)T);
    XR_LAUNCH("search_big", k_search_big<true>, dim3(big_grid), dim3(256), sizeof(int32_t) * (size_t)big_stage_entries(),
              query->qo_bbox(), query->qo_fxy(),
              query->qo_len(), query->qo_off(), query->m, g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_face.get(),
              big_list.get(), counters.get() + 2, cand_off.get(), cand_count.get(), cand_tgt.get(), cand_src.get(),
              counters.get() + 3, capacity, pending.get(), counters.get() + 1, big_stage_entries());
    // queue length and number of big faces still to be filled -> host
    int32_t *mail = const_cast<int32_t *>(engine().mailbox);
    XR_LAUNCH("publish", k_publish, dim3(1), dim3(64), 0, counters.get() + 3, mail + 0, counters.get() + 1, mail + 3);
    mailbox_wait();
    const int32_t C32 = mail[0], n_pending = mail[3];
    XR_REQUIRE(C32 >= 0, XR_ERR_LIMIT, "candidate pair count exceeds the int32 range");
    const int64_t C = C32;
    tree->last_candidates = C;
    if (n_pending > 0) {
        // some big faces need more room than the margin: move what is there to a larger queue, then fill them
        DevBuf<int32_t> bigger_tgt((size_t)C), bigger_src((size_t)C);
        XR_HIP(hipMemcpyAsync(bigger_tgt.get(), cand_tgt.get(), sizeof(int32_t) * (size_t)capacity, hipMemcpyDeviceToDevice, st));
        XR_HIP(hipMemcpyAsync(bigger_src.get(), cand_src.get(), sizeof(int32_t) * (size_t)capacity, hipMemcpyDeviceToDevice, st));
        stream_sync();
        cand_tgt = std::move(bigger_tgt);
        cand_src = std::move(bigger_src);
        capacity = C;
        h2d(counters.get() + 1, &n_pending, sizeof(int32_t)); // (k_publish zeroed the device copy)
        XR_LAUNCH("search_big_fill", k_search_big<false>, dim3(big_grid), dim3(256), sizeof(int32_t), query->qo_bbox(), query->qo_fxy(),
                  query->qo_len(), query->qo_off(), query->m, g, tree->cell_start.get(), tree->rec_bb.get(), tree->rec_face.get(),
                  pending.get(), counters.get() + 1, cand_off.get(), cand_count.get(), cand_tgt.get(), cand_src.get(),
                  (int32_t *)nullptr, capacity, (int32_t *)nullptr, (int32_t *)nullptr, 0);
        XR_HIP(hipMemsetAsync(counters.get() + 1, 0, sizeof(int32_t), st));
    }
    // (counters[1], zeroed again by k_publish, is reused below as the number of long rows)
    DevBuf<int32_t> cand_sid((size_t)C);
    DevBuf<double> cand_area((size_t)C);
    if (C > 0) {
        // --- clip (+ per-row survivor counts)
        launch_clip_for(tree, query, cand_tgt.get(), cand_src.get(), C, cand_area.get(), cand_sid.get(), counters.get(),
                        nnz_row.get());
    }
    // --- rows
    // P is needed on the host anyway; the overflow counter travels in the same mailbox round trip
    exclusive_scan_i32(nnz_row.get(), csr->indptr.get(), T, mail + 1, counters.get(), mail + 2);
    mailbox_wait();
    int32_t tail[4] = {mail[2], 0, 0, mail[1]};
    if (tail[0] > 0) {
        // polygon buffer overflow in the small-MAXV kernel (floating-point degenerate pairs):
        // redo those pairs with the oracle's buffer size, then recount every row.
        XR_HIP(hipMemsetAsync(counters.get(), 0, sizeof(int32_t), st));
        XR_HIP(hipMemsetAsync(nnz_row.get(), 0, sizeof(int32_t) * (size_t)T, st));
        launch_clip<64, 64>(tree, query, cand_tgt.get(), cand_src.get(), C, cand_area.get(), true, cand_sid.get(),
                            counters.get(), nnz_row.get());
        XR_LAUNCH("row_recount", k_row_count, dim3(div_up(T, 256)), dim3(256), 0, cand_off.get(), cand_count.get(),
                  cand_area.get(), T,
                  query->qo_perm(), nnz_row.get());
        exclusive_scan_i32(nnz_row.get(), csr->indptr.get(), T);
        tail[3] = read_scalar(csr->indptr.get() + T);
    }
    XR_REQUIRE(tail[3] >= 0, XR_ERR_LIMIT, "nnz exceeds the int32 range");
    const int64_t P = tail[3];
    csr->nnz = P;
    csr->indices.alloc((size_t)P);
    csr->data.alloc((size_t)P);
    // rows are best processed in the query mesh's spatial order (apply kernels)
    if (query->qo_perm()) {
        csr->row_order.alloc((size_t)T);
        XR_HIP(hipMemcpyAsync(csr->row_order.get(), query->qo_perm(), sizeof(int32_t) * (size_t)T,
                              hipMemcpyDeviceToDevice, st));
        csr->has_row_order = true;
    }
    if (P > 0) {
        DevBuf<int32_t> long_rows((size_t)T);
        csr->long_rows.alloc((size_t)(P / XR_APPLY_LONG_ROW + 1));
        csr->n_long.alloc(1);
        csr->has_long = true;
        XR_HIP(hipMemsetAsync(csr->n_long.get(), 0, sizeof(int32_t), st));
        XR_LAUNCH("row_fill", k_row_fill, dim3(xcd_grid(div_up(T, 256), remap_rows)), dim3(256), 0, cand_off.get(), cand_count.get(),
                  block_seg.get(), is_big.get(), cand_tgt.get(), cand_sid.get(), cand_area.get(), T, csr->indptr.get(), tree_area, relative,
                  csr->indices.get(), csr->data.get(), long_rows.get(), counters.get() + 1, csr->long_rows.get(),
                  csr->n_long.get(), remap_rows);
        const size_t shmem = sizeof(uint32_t) * (BM_WORDS + 256) + sizeof(int32_t) * (8 + BM_STAGE) + sizeof(uint16_t) * (BM_WORDS / 8);
        static bool attr_set = false;
        if (!attr_set) {
            XR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_fill_long),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr_set = true;
        }
        XR_LAUNCH("row_fill_long", k_row_fill_long, dim3(engine().num_cu), dim3(256), shmem, cand_off.get(),
                  cand_count.get(), cand_sid.get(), cand_area.get(), csr->indptr.get(), tree_area, relative, tree->n_face,
                  csr->indices.get(), csr->data.get(), long_rows.get(), counters.get() + 1, (int64_t)-1, (const int32_t *)nullptr);
    }
}

} // namespace xr

using namespace xr;

extern "C" {

int xr_overlap(xr_mesh *tree, xr_mesh *query, int relative, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && query && out, XR_ERR_INVALID, "xr_overlap: NULL argument");
    xr_csr *csr = new xr_csr();
    try {
        overlap(tree, query, relative != 0, csr);
        stream_sync();
    } catch (...) {
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_overlap_apply_dev(xr_mesh *tree, xr_mesh *query, int relative, int method, double percentile, const void *source_dev,
                         int source_dtype, int64_t K, double *out_dev, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && query && out, XR_ERR_INVALID, "xr_overlap_apply_dev: NULL argument");
    XR_REQUIRE(K >= 0 && (source_dev || tree->n_face == 0 || K == 0) && (out_dev || query->n_face == 0 || K == 0), XR_ERR_INVALID,
               "xr_overlap_apply_dev: NULL data argument");
    XR_REQUIRE(source_dtype == XR_F64 || source_dtype == XR_F32, XR_ERR_INVALID, "unsupported source dtype id %d", source_dtype);
    host_stamp(0);
    xr_csr *csr = new xr_csr();
    try {
        EarlyApply early;
        early.fn = [&](const xr_csr *c) { csr_apply_dev(c, method, percentile, source_dev, source_dtype, K, out_dev); };
        // one variable and a streaming reducer: the apply is one launch that needs no size on the host
        const bool can_early = option(OPT_EARLY_APPLY) != 0 && K == 1 && method != XR_MODE && method != XR_PERCENTILE;
        overlap(tree, query, relative != 0, csr, can_early ? &early : nullptr);
        host_stamp(8);
        if (!early.done && K > 0) csr_apply_dev(csr, method, percentile, source_dev, source_dtype, K, out_dev);
        dev_call_done(); // (xr_set_async(1): returns with the apply in flight)
        host_stamp(9);
    } catch (...) {
        stream_sync();
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_overlap_partial_dev(xr_mesh *tree, xr_mesh *query, int relative, int method, const void *source_dev, int source_dtype,
                           int64_t K, double *out_dev, int rows_layout, xr_csr **out) {
    XR_API_BEGIN
    XR_REQUIRE(tree && query && out, XR_ERR_INVALID, "xr_overlap_partial_dev: NULL argument");
    XR_REQUIRE(K >= 0 && (source_dev || tree->n_face == 0 || K == 0) && (out_dev || query->n_face == 0 || K == 0), XR_ERR_INVALID,
               "xr_overlap_partial_dev: NULL data argument");
    // (checked BEFORE the weight build: inside it they would fire in the early apply's callback, half-way through the build)
    XR_REQUIRE(xr_partial_components(method) > 0, XR_ERR_INVALID, "xr_overlap_partial_dev: method %d has no partial state", method);
    XR_REQUIRE(source_dtype == XR_F32 || source_dtype == XR_F64, XR_ERR_INVALID, "xr_overlap_partial_dev: source_dtype must be XR_F32 or XR_F64");
    XR_REQUIRE(K < 65536, XR_ERR_LIMIT, "xr_overlap_partial_dev: at most 65535 variables per call");
    xr_csr *csr = new xr_csr();
    try {
        EarlyApply early;
        early.fn = [&](const xr_csr *c) { csr_partial_dev(c, method, source_dev, source_dtype, K, out_dev, rows_layout); };
        const bool can_early = option(OPT_EARLY_APPLY) != 0 && K == 1; // (one variable: the wave-window kernel needs no size on the host)
        overlap(tree, query, relative != 0, csr, can_early ? &early : nullptr);
        if (!early.done && K > 0) csr_partial_dev(csr, method, source_dev, source_dtype, K, out_dev, rows_layout);
        dev_call_done();
    } catch (...) {
        stream_sync();
        delete csr;
        throw;
    }
    *out = csr;
    XR_API_END
}

int xr_overlap_stats(const xr_mesh *tree, int64_t *n_candidates) {
    XR_API_BEGIN
    XR_REQUIRE(tree && n_candidates, XR_ERR_INVALID, "xr_overlap_stats: NULL argument");
    *n_candidates = tree->last_candidates;
    XR_API_END
}

} // extern "C"
