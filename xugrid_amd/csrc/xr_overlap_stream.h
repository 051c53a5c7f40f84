// xr_overlap_stream.h -- triangle x triangle weight construction as ONE streamed kernel (included by xr_overlap.hip behind
// xr_overlap_fused.h; reference seam: CellTree2d.intersect_faces + weights /= area + MatrixCSR.from_triplet,
// xugrid/regrid/unstructured.py:109-135, xugrid/regrid/regridder.py:433-435).
//
// Until round 5 the pairs of the benchmark went through three device-filling kernels one after the other -- k_search (bound by
// the latency of its dependent loads, vector ALUs 11 % active), k_clip_tri_queue (bound by the vector ALUs), k_assemble (bound
// by latency again) -- with a 66 MB pair queue and 85 MB of per-candidate output crossing HBM in between, and the big faces
// (hull slivers) on a side stream that ended AFTER the main chain.  Here a workgroup owns 256 consecutive target faces from
// the grid walk to their finished CSR rows:
//   phase S  the grid walk of k_search (one thread per face, hits parked in 16 LDS slots), compacted in LDS into the block's
//            own pair list (packed word: owner thread << 24 | record) -- written to the block's FIXED stretch of a scratch
//            array (no queue cursor, no atomic), from where the clip reads it back (its own CU's L2; keeping the list in LDS
//            beside the clip's columns would cost two of the five resident blocks per CU)
//   phase C  the bit-mask / LUT clip of xr_clip_tri.h over the block's pairs, 256 at a time; survivors are counted per row in
//            LDS, (source id | dead) and the area go to the block's scratch stretch
//   phase A  rows and entries of the block are reserved with ONE returning 64-bit atomic (rows << 32 | entries: stored rows are
//            handed out in order of ARRIVAL -- xr_csr::row_order maps them to the caller's faces, nothing depends on the
//            order of the blocks), every survivor is ranked among its row's candidates in LDS and written to its final place
//   big      faces with more than 16 hits (or too many grid rows / records) are LISTED during phase S; every block, when its
//            own rows are done, claims listed faces one at a time (CAS on a claim counter that never passes the list's
//            length) and handles each with all 256 threads: scan-converted walk into LDS (k_search_big's), clip from the
//            block's scratch stretch, rank (all-pairs up to 512 candidates, bitonic sort in LDS beyond), one ticket, one row.
//            A block that lists a face looks at the list after that, so every listed face is claimed by somebody; nobody
//            waits for anything but a list entry whose writer is already running.
// Blocks in different phases share a CU, so the latency-bound walk and assembly of some hide under the ALU-bound clip of
// others -- the overlap two kernels on two streams would have to be scheduled into.  No side stream, no fork / join, no
// k_place_big, no pair queue, no per-candidate arrays that another kernel reads.  Arithmetic untouched: same clip, same
// dust confirmation, same ranking -- the CSR a caller downloads is bit-identical to the one of the kernel chain.
// Pairs the kernel does not take (a big face with more than SB_STRETCH candidates, a clip that needs more than 6 vertices)
// raise an error bit and the host redoes the matrix with the kernel chain of xr_overlap_fused.h.
#pragma once

namespace xr {

static constexpr int SB_STRETCH = FB * SLOTS; // pairs of scratch per block = the most its 256 faces can park
static constexpr int SB_ALLPAIRS = 512;       // big faces of at most this many candidates are ranked all-pairs

// control words (int32 units; zero at rest, k_publish_stream clears them).  Words that blocks hammer with returning atomics sit on
// 128-byte lines of their own: the memory side serialises the atomics of a line.
static constexpr int SC_TICKET = 0;    // line 0, 64-bit: rows << 32 | entries handed out so far
static constexpr int SC_BIG = 32;      // line 1, 64-bit: big faces listed << 32 | big faces claimed
static constexpr int SC_CAND = 64;     // lines 2..9: candidates of the regular faces, one counter per XCD (statistics)
static constexpr int SC_ERROR = 320;   // line 10: bit 0: clip needs > 6 vertices, bit 1: big face beyond the stage, bit 2: CSR capacity
static constexpr int SC_NLONG = 321;   // rows of more than XR_APPLY_LONG_ROW entries
static constexpr int SC_MAXROW = 322;  // entries of the longest of those
static constexpr int SC_BIGPAIRS = 323; // candidates of the big faces
static constexpr size_t SC_HEAD = 352; // the list of big faces (face id + 1, zero at rest) starts here

// one stored row: offset, caller's face, tiling hint, long-row list
__device__ __forceinline__ void stream_row_header(long long r, long long first, int n_entries, int64_t t, int64_t n_query,
                                                  const int32_t *__restrict__ q_perm, const double *__restrict__ q_bbox,
                                                  const MortonParams &tile, int32_t *__restrict__ tile_key,
                                                  int32_t *__restrict__ indptr, int32_t *__restrict__ row_order,
                                                  int32_t *__restrict__ apply_long_rows, int32_t *ctl, int64_t csr_capacity) {
    indptr[r] = (int32_t)first;
    row_order[r] = q_perm ? q_perm[t] : (int32_t)t;
    if (tile_key) {
        const int64_t mid = (t & ~(int64_t)(tile.n_run - 1)) + tile.n_run / 2;
        tile_key[r] = morton_key(tile, reinterpret_cast<const double4 *>(q_bbox)[mid < n_query ? mid : n_query - 1]);
    }
    // (a row beyond the capacity belongs to a matrix the host is about to redo: the list is sized for capacity / 32 rows)
    if (n_entries > XR_APPLY_LONG_ROW && first + n_entries <= csr_capacity) {
        apply_long_rows[atomicAdd(&ctl[SC_NLONG], 1)] = (int32_t)r;
        atomicMax(&ctl[SC_MAXROW], n_entries);
    }
}

__global__ void __launch_bounds__(FB) __attribute__((amdgpu_waves_per_eu(5)))
k_overlap_block(const double *__restrict__ q_bbox, const double *__restrict__ q_fxy, const int32_t *__restrict__ q_perm,
                int64_t n_query, GridParams g, int64_t n_tree, const int32_t *__restrict__ cell_start,
                const float *__restrict__ rec_bb, const double *__restrict__ rec_fxy, const int32_t *__restrict__ rec_face,
                int32_t *sc_pair, int32_t *sc_sid, double *sc_area, int32_t *ctl, int32_t *big_list,
                MortonParams tile, int32_t *__restrict__ tile_key, const double *__restrict__ src_area, bool relative, double dust,
                int32_t *__restrict__ indptr, int32_t *__restrict__ indices, double *__restrict__ data,
                int32_t *__restrict__ row_order, int32_t *__restrict__ apply_long_rows, int64_t csr_capacity, bool remap,
                unsigned long long *dbg = nullptr /* XR_STREAM_DEBUG: phase clocks (100 MHz) summed over the blocks */) {
    // one region, three lives: the walk's slots (18.3 KB) / the clip's columns (28 KB) / the rank stage (16 KB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint2 sh_lut[TRI_LUT];
    __shared__ int32_t sh_nnz[FB];      // survivors of the face (LDS atomics)
    __shared__ uint16_t sh_lo[FB + 2];  // offset of the face's pairs inside the block's stretch; [FB] = all pairs
    __shared__ uint16_t sh_rowoff[FB];  // CSR offset of the face's row inside the block
    __shared__ int32_t sh_wave[FWAVES];
    __shared__ long long sh_base;
    __shared__ float sh_box[4][4];
    __shared__ int32_t sh_nbig, sh_cursor, sh_claim, sh_alive, sh_nself;
    __shared__ uint8_t sh_self[FB]; // own big faces whose list slot had been given away
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_blocks = (n_query + FB - 1) / FB;
    const int64_t lb = xcd_block(n_blocks, remap);
    const float4 *__restrict__ rbb = reinterpret_cast<const float4 *>(rec_bb);
    const double2 *tfx = reinterpret_cast<const double2 *>(q_fxy);
    const double2 *sfx = reinterpret_cast<const double2 *>(rec_fxy);
    const int64_t sbase = (int64_t)blockIdx.x * SB_STRETCH; // the block's stretch of the scratch arrays
    double2 *col = reinterpret_cast<double2 *>(smem) + tid * (TRI_MAXV + 1);
    tri_lut_init(sh_lut);
    sh_nnz[tid] = 0;
    if (tid == 0) sh_nself = 0;
    __syncthreads();
    bool overflow = false, overflow_cap = false;
    unsigned long long clk0 = 0, clk1 = 0, clk2 = 0, clk3 = 0;
    if (dbg) clk0 = wall_clock64();
    if (lb < n_blocks) {
        // ================================================================ phase S: the grid walk (k_search, PACK form)
        int32_t(*sh_slots)[FB] = reinterpret_cast<int32_t(*)[FB]>(smem);                      // [SLOTS + 1][FB]
        float4 *sh_bigbb = reinterpret_cast<float4 *>(smem + sizeof(int32_t) * (SLOTS + 1) * FB); // [BIGREC_BLOCK]
        int32_t *sh_bigrec = reinterpret_cast<int32_t *>(sh_bigbb + BIGREC_BLOCK);            // [BIGREC_BLOCK]
        const int64_t t = lb * FB + tid;
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
        int l_coop = l_split;
        for (int l = l_split - 1; l >= 1; l--) {
            const int first = cell_start[g.base[l]];
            if (((int64_t)big0 - first) * 64 > n_tree) break;
            l_coop = l;
        }
        if (tid == 0) sh_nbig = 0;
        if (big0 < (int)n_tree || l_coop < l_split) {
            float bx0 = qx0, bx1 = qx1, by0 = qy0, by1 = qy1;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                bx0 = fminf(bx0, __shfl_xor(bx0, d, 64));
                bx1 = fmaxf(bx1, __shfl_xor(bx1, d, 64));
                by0 = fminf(by0, __shfl_xor(by0, d, 64));
                by1 = fmaxf(by1, __shfl_xor(by1, d, 64));
            }
            if (lane == 0) {
                sh_box[wave][0] = bx0; sh_box[wave][1] = bx1; sh_box[wave][2] = by0; sh_box[wave][3] = by1;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; w++) {
                bx0 = fminf(bx0, sh_box[w][0]); bx1 = fmaxf(bx1, sh_box[w][1]);
                by0 = fminf(by0, sh_box[w][2]); by1 = fmaxf(by1, sh_box[w][3]);
            }
            for (int r = big0 + tid; r < (int)n_tree; r += FB) {
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
                const int bcx0 = cell_coord((double)bx0 + g.x0, g.x0, g.inv_h0, g.nx[0]), bcx1 = cell_coord((double)bx1 + g.x0, g.x0, g.inv_h0, g.nx[0]);
                const int bcy0 = cell_coord((double)by0 + g.y0, g.y0, g.inv_h0, g.ny[0]), bcy1 = cell_coord((double)by1 + g.y0, g.y0, g.inv_h0, g.ny[0]);
                int n_items = 0;
                for (int l = l_coop; l < l_split; l++) n_items += (bcy1 >> (l * LEVEL_SHIFT)) - max((bcy0 >> (l * LEVEL_SHIFT)) - 1, 0) + 1;
                if (n_items > FB) {
                    l_coop = l_split; // (a block spanning too many grid rows: its faces walk these levels themselves)
                } else if (tid < n_items) {
                    int it = tid, l = l_coop;
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
                if (tid == 0) sh_nbig = 0;
            }
        }
        __syncthreads();
        const int n_bigblk = sh_nbig;
        int count = 0;
        bool big = false;
        if (t < n_query) {
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
                                const int last = r1[k] - 1;
                                float4 bx[WALK_LOADS];
                                static_assert(WALK_LOADS - 1 <= WALK_PAD, "rec_bb padding");
                                const char *step_base = reinterpret_cast<const char *>(rbb) + ((uint32_t)r << 4);
#pragma unroll
                                for (int u = 0; u < WALK_LOADS; u++) bx[u] = *reinterpret_cast<const float4 *>(step_base + 16 * u);
#pragma unroll
                                for (int u = 0; u < WALK_LOADS; u++) {
                                    sh_slots[count < SLOTS ? count : SLOTS][tid] = r + u;
                                    count += fmaxf(box_gap(bx[u], qx0, qx1, qy0, qy1), r + u <= last ? -INFINITY : 1.0f) < 0.0f ? 1 : 0;
                                }
                            }
                        }
                    }
                }
            }
            for (int k = 0; k < n_bigblk && !big; k++) {
                const bool h = rec_hit(sh_bigbb[k], qx0, qx1, qy0, qy1);
                sh_slots[count < SLOTS ? count : SLOTS][tid] = sh_bigrec[k];
                count += h ? 1 : 0;
            }
            if (count > SLOTS) big = true;
            // a big face joins the list: the slot from one returning atomic, the entry (face + 1; the list is zero at rest) as
            // an atomic store -- whoever claims the slot polls the word
            if (big) {
                // One word holds (faces listed << 32 | tickets taken).  The add returns both halves at that instant: slot =
                // listed; if a ticket for that slot is out already (a block that looked when the list had just run dry took it
                // and left), nobody will come for the entry -- the face stays with this block (sh_self).  Else it is published:
                // face + 1 as an atomic store (the list is zero at rest), the ticket's holder polls the word.
                const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(ctl + SC_BIG), 1ull << 32);
                const uint32_t slot = (uint32_t)(old >> 32);
                if ((uint32_t)old > slot) sh_self[atomicAdd(&sh_nself, 1)] = (uint8_t)tid;
                else __hip_atomic_store(&big_list[slot], (int32_t)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const bool regular = t < n_query && !big;
        const int mine = regular ? count : 0;
        int own[SLOTS];
#pragma unroll
        for (int j = 0; j < SLOTS; j++) own[j] = j < mine ? sh_slots[j][tid] : 0;
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
        for (int w = 0; w < FWAVES; w++) {
            if (w < wave) woff += sh_wave[w];
            total += sh_wave[w];
        }
        const int lo = woff + incl - mine;
        sh_lo[tid] = (uint16_t)lo;
        if (tid == 0) {
            sh_lo[FB] = (uint16_t)total; // (<= 4096)
            if (total > 0) atomicAdd(&ctl[SC_CAND + (blockIdx.x & 7) * 32], total); // (statistics; nobody waits for it)
        }
        int32_t *flat = &sh_slots[0][0];
#pragma unroll
        for (int j = 0; j < SLOTS; j++)
            if (j < mine) flat[lo + j] = own[j] | (int32_t)(tid << 24);
        __syncthreads();
        for (int i = tid; i < total; i += FB) sc_pair[sbase + i] = flat[i];
        __syncthreads(); // (the stretch is read back by other waves; the slots become the clip's columns)
        if (dbg) clk1 = wall_clock64();
        // ================================================================ phase C: clip the block's pairs
        const int32_t t0 = (int32_t)(lb * FB);
        int n_v = tid < total ? sc_pair[sbase + tid] : 0;
        for (int i0 = 0; i0 < total; i0 += FB) {
            const int i = i0 + tid;
            const bool active = i < total;
            const int v = n_v;
            const int owner = (int)((uint32_t)v >> 24), s = v & 0xffffff;
            P2 tv[3] = {{0, 0}, {0, 0}, {0, 0}}, sv[3] = {{0, 0}, {0, 0}, {0, 0}};
            int sid = 0;
            if (active) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const double2 a = tfx[(int64_t)(t0 + owner) * 3 + j], b2 = sfx[(int64_t)s * 3 + j];
                    tv[j] = P2{a.x, a.y};
                    sv[j] = P2{b2.x, b2.y};
                }
                sid = rec_face[s];
            }
            n_v = i + FB < total ? sc_pair[sbase + i + FB] : 0; // (the next round's pair, in flight during the clip)
            double area = tri_clip_area<1>(tv, sv, col, sh_lut, active);
            const bool suspicious = active && area > 0 && area <= dust;
            if (__any(suspicious)) {
                if (suspicious && !pair_passes_box_and_sat(tfx + (int64_t)(t0 + owner) * 3, 3, sfx + (int64_t)s * 3, 3)) area = 0.0;
            }
            overflow = overflow || (active && area == TRI_AREA_OVERFLOW);
            const bool keep = active && area > 0;
            if (active) sc_sid[sbase + i] = keep ? sid : 0x7fffffff;
            if (keep) {
                sc_area[sbase + i] = area;
                atomicAdd(&sh_nnz[owner], 1);
            }
        }
        __syncthreads(); // (columns dead, counts complete, scratch written)
        if (dbg) clk2 = wall_clock64();
        // ================================================================ phase A: the block's rows
        int32_t *sh_stage = reinterpret_cast<int32_t *>(smem); // [SB_STRETCH]
        for (int i0 = tid; i0 < total; i0 += 4 * FB) {
            int sd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * FB;
                sd[u] = sc_sid[sbase + (i < total ? i : total - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * FB;
                if (i < total) sh_stage[i] = sd[u];
            }
        }
        const int my_nnz = regular ? sh_nnz[tid] : 0;
        int block_nnz = 0, block_rows = 0;
        const int rowoff = block_excl_scan(my_nnz, sh_wave, &block_nnz);
        const int rowidx = block_excl_scan(regular ? 1 : 0, sh_wave, &block_rows);
        sh_rowoff[tid] = (uint16_t)rowoff;
        if (tid == 0)
            sh_base = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(ctl + SC_TICKET),
                                           ((unsigned long long)block_rows << 32) | (unsigned long long)block_nnz);
        __syncthreads();
        const long long base = sh_base & 0xffffffffll, row_base = sh_base >> 32;
        if (regular) stream_row_header(row_base + rowidx, base + rowoff, my_nnz, t, n_query, q_perm, q_bbox, tile, tile_key, indptr,
                                       row_order, apply_long_rows, ctl, csr_capacity);
        for (int i0 = tid; i0 < total; i0 += 4 * FB) {
            double area[4];
            int row[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * FB;
                const int64_t c = sbase + (i < total ? i : total - 1);
                row[u] = (int)((uint32_t)sc_pair[c] >> 24);
                area[u] = (i < total && sh_stage[i] != 0x7fffffff) ? sc_area[c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * FB;
                if (i >= total) continue;
                const int s = sh_stage[i];
                if (s == 0x7fffffff) continue;
                const int a0 = sh_lo[row[u]], a1 = sh_lo[row[u] + 1];
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
    }
    if (dbg) clk3 = wall_clock64();
    int dbg_faces = 0;
    int self_i = 0; // (thread 0's)
    // ==================================================================== the listed big faces, one at a time
    int32_t *sh_stage = reinterpret_cast<int32_t *>(smem); // [SB_STRETCH]: the walk's hits, later the rank stage
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
        __syncthreads(); // (everything of the previous item is done with LDS and the scratch stretch)
        if (tid == 0) {
            // Own faces whose slot had been given away first.  Then tickets: look (a load), and if faces are listed beyond the
            // tickets taken, take one -- a single add, never taken back (a compare-and-swap costs a round trip to the memory side
            // per SUCCESSFUL claim, all contenders' expected values going stale at once: 1189 claims by 1280 blocks took 6 ms).
            // The add returns (listed, tickets) at that instant: ticket < listed -- the entry is published or about to be, poll
            // it; else the ticket ran ahead of the list and is void -- the block that lists that slot later sees it (above).
            // A block that lists a face comes here afterwards and takes tickets until none is left below the listed count, so
            // every published entry finds a holder.
            unsigned long long *big_word = reinterpret_cast<unsigned long long *>(ctl + SC_BIG);
            int face = -1;
            if (self_i < sh_nself) {
                face = (int)(lb * FB) + sh_self[self_i++];
            } else {
                for (;;) {
                    const unsigned long long w = __hip_atomic_load(big_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)w >= (uint32_t)(w >> 32)) break;
                    const unsigned long long old = atomicAdd(big_word, 1ull);
                    const uint32_t ticket = (uint32_t)old;
                    if (ticket >= (uint32_t)(old >> 32)) continue; // (void ticket; look again)
                    int v, polls = 0;
                    while ((v = __hip_atomic_load(&big_list[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++polls < (1 << 22))
                        __builtin_amdgcn_s_sleep(2);
                    if (v == 0) { // (cannot happen: the writer of the entry is a running wave; bounded all the same)
                        atomicOr(&ctl[SC_ERROR], 8);
                        continue;
                    }
                    face = v - 1;
                    __hip_atomic_store(&big_list[ticket], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (zero at rest again)
                    break;
                }
            }
            sh_claim = face;
            sh_cursor = 0;
            sh_alive = 0;
        }
        __syncthreads();
        const int t = sh_claim;
        if (t < 0) break;
        unsigned long long clk_f = 0;
        if (dbg) clk_f = wall_clock64();
        dbg_faces++;
        P2 tv[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double2 a = tfx[(int64_t)t * 3 + j];
            tv[j] = P2{a.x, a.y};
        }
        const double4 bb = reinterpret_cast<const double4 *>(q_bbox)[t];
        const float qx0 = f32_below(bb.x - g.x0), qx1 = f32_above(bb.y - g.x0);
        const float qy0 = f32_below(bb.z - g.y0), qy1 = f32_above(bb.w - g.y0);
        // ---- the walk of k_search_big: per grid row only the cells under the polygon's x-extent inside the row's slab; the
        // block's four waves take 64-row batches in turn, hits are appended to the LDS stage through sh_cursor
        for (int l = 0; l < g.n_levels; l++) {
            const double h = level_h(g, l), inv_h = level_inv_h(g, l);
            const double eps = 1e-6 * h;
            const int nx = g.nx[l], ny = g.ny[l], base = g.base[l];
            const int cy0 = cell_coord(bb.z - h, g.y0, inv_h, ny), cy1 = cell_coord(bb.w, g.y0, inv_h, ny);
            int batch_id = 0;
            for (int cyb = cy0; cyb <= cy1; cyb += 64, batch_id++) {
                if (((batch_id + l) & 3) != wave) continue;
                const int cy = cyb + lane;
                int r0 = 0, r1 = 0;
                float rx0 = qx0, rx1 = qx1;
                if (cy <= cy1) {
                    const double ya = g.y0 + (double)cy * h - eps, yb = g.y0 + (double)(cy + 2) * h + eps;
                    double xlo = INFINITY, xhi = -INFINITY;
                    P2 p = tv[2];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const P2 q = tv[k];
                        const double ylo = fmin(p.y, q.y), yhi = fmax(p.y, q.y);
                        if (yhi >= ya && ylo <= yb) {
                            double xa = p.x, xb = q.x;
                            if (yhi > ylo) {
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
                unsigned long long long_mask = __ballot(len > LONG_RUN); // long runs: the whole wave, 64 records at a time
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
                                if (slot < SB_STRETCH) sh_stage[slot] = r;
                            }
                        }
                    }
                }
                const int my_r1 = len > LONG_RUN ? r0 : r1; // short runs: one lane per grid row
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
                            if (pos < SB_STRETCH) sh_stage[pos] = r;
                            pos++;
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int n = sh_cursor;
        if (n > SB_STRETCH) { // (uniform) more candidates than the stage holds: the host redoes the matrix with the kernel chain
            if (tid == 0) atomicOr(&ctl[SC_ERROR], 2);
            continue;
        }
        if (tid == 0) atomicAdd(&ctl[SC_BIGPAIRS], n);
        for (int i = tid; i < n; i += FB) sc_pair[sbase + i] = sh_stage[i];
        __syncthreads(); // (the stage becomes the clip's columns)
        // ---- clip
        int n_s = tid < n ? sc_pair[sbase + tid] : 0;
        for (int i0 = 0; i0 < n; i0 += FB) {
            const int i = i0 + tid;
            const bool active = i < n;
            const int s = n_s;
            P2 sv[3] = {{0, 0}, {0, 0}, {0, 0}};
            int sid = 0;
            if (active) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const double2 b2 = sfx[(int64_t)s * 3 + j];
                    sv[j] = P2{b2.x, b2.y};
                }
                sid = rec_face[s];
            }
            n_s = i + FB < n ? sc_pair[sbase + i + FB] : 0;
            double area = tri_clip_area<1>(tv, sv, col, sh_lut, active);
            const bool suspicious = active && area > 0 && area <= dust;
            if (__any(suspicious)) {
                if (suspicious && !pair_passes_box_and_sat(tfx + (int64_t)t * 3, 3, sfx + (int64_t)s * 3, 3)) area = 0.0;
            }
            overflow = overflow || (active && area == TRI_AREA_OVERFLOW);
            const bool keep = active && area > 0;
            if (active) sc_sid[sbase + i] = keep ? sid : 0x7fffffff;
            if (keep) sc_area[sbase + i] = area;
            const unsigned long long surv = __ballot(keep);
            if (lane == 0 && surv) atomicAdd(&sh_alive, __popcll(surv));
        }
        __syncthreads();
        // ---- rank + row
        const int n_alive = sh_alive;
        if (tid == 0)
            sh_base = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(ctl + SC_TICKET), (1ull << 32) | (unsigned long long)n_alive);
        if (n <= SB_ALLPAIRS) {
            // all-pairs rank among the row's candidates (dead ones compare as +inf)
            const int n4 = (n + 3) & ~3;
            for (int i = tid; i < n4; i += FB) sh_stage[i] = i < n ? sc_sid[sbase + i] : 0x7fffffff;
            __syncthreads();
            const long long base = sh_base & 0xffffffffll;
            const int4 *quad = reinterpret_cast<const int4 *>(sh_stage);
            for (int i = tid; i < n; i += FB) {
                const int sd = sh_stage[i];
                if (sd == 0x7fffffff) continue;
                int rank = 0;
                for (int j = 0; j < n4 / 4; j++) {
                    const int4 q = quad[j];
                    rank += (q.x < sd) + (q.y < sd) + (q.z < sd) + (q.w < sd);
                }
                const long long pos = base + rank;
                if (pos < csr_capacity) {
                    const double a = sc_area[sbase + i];
                    indices[pos] = sd;
                    data[pos] = relative ? a / src_area[sd] : a;
                } else {
                    overflow_cap = true;
                }
            }
        } else {
            // bitonic sort of the source ids in LDS (dead ones last); the sorted list IS the row's indices, an entry finds its
            // place by binary search (a source face occurs once per row)
            int np2 = 1024;
            while (np2 < n) np2 <<= 1;
            for (int i = tid; i < np2; i += FB) sh_stage[i] = i < n ? sc_sid[sbase + i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= np2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int p = tid; p < np2 / 2; p += FB) {
                        const int a = ((p & ~(j - 1)) << 1) | (p & (j - 1)), b = a | j;
                        const bool asc = (a & k) == 0;
                        const int va = sh_stage[a], vb = sh_stage[b];
                        if ((va > vb) == asc) {
                            sh_stage[a] = vb;
                            sh_stage[b] = va;
                        }
                    }
                    __syncthreads();
                }
            }
            const long long base = sh_base & 0xffffffffll;
            for (int i = tid; i < n_alive; i += FB)
                if (base + i < csr_capacity) indices[base + i] = sh_stage[i];
                else overflow_cap = true;
            for (int i = tid; i < n; i += FB) {
                const int sd = sc_sid[sbase + i];
                if (sd == 0x7fffffff) continue;
                int lo = 0, hi = n_alive; // first position whose id is >= sd
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sh_stage[mid] < sd) lo = mid + 1;
                    else hi = mid;
                }
                if (base + lo < csr_capacity) {
                    const double a = sc_area[sbase + i];
                    data[base + lo] = relative ? a / src_area[sd] : a;
                }
            }
        }
        if (dbg && tid == 0) {
            const unsigned long long d = wall_clock64() - clk_f;
            atomicAdd(&dbg[3], d);
            atomicAdd(&dbg[4], 1ull);
            atomicMax(&dbg[5], d);
        }
        if (tid == 0)
            stream_row_header(sh_base >> 32, sh_base & 0xffffffffll, n_alive, t, n_query, q_perm, q_bbox, tile, tile_key, indptr, row_order,
                              apply_long_rows, ctl, csr_capacity);
    }
    if (dbg && tid == 0) {
        const unsigned long long end = wall_clock64();
        if (lb < n_blocks) {
            atomicAdd(&dbg[0], clk1 - clk0);
            atomicAdd(&dbg[1], clk2 - clk1);
            atomicAdd(&dbg[2], clk3 - clk2);
        }
        atomicMax(&dbg[6], (unsigned long long)dbg_faces);
        atomicMin(&dbg[7], clk0);
        atomicMax(&dbg[8], end);
        atomicMax(&dbg[9], end - clk0);
        atomicMax(&dbg[10], clk0);
    }
    if (overflow) atomicOr(&ctl[SC_ERROR], 1);
    if (overflow_cap) atomicOr(&ctl[SC_ERROR], 4);
}

// sizes, error bits and the long-row count -> host mailbox / device words of the matrix; closes the row offsets; clears the
// control words behind itself (zero at rest).  One wave.  Mailbox: [0] candidates of the regular faces, [1] of the big ones,
// [2] big faces, [3] error bits, [4] rows, [5] entries, [6] long rows, [7] longest row.
__global__ void k_publish_stream(int32_t *ctl, int32_t *__restrict__ indptr, int64_t n_query, int32_t *__restrict__ n_apply_long_out,
                                 int32_t *mail, int32_t seq, int64_t cap) {
    const int t = threadIdx.x;
    // one load per lane: [0..1] ticket, [2..3] big word, [4..7] error / long rows / longest row / big pairs, [16..23] candidate counters
    int32_t w = 0;
    if (t < 2) w = ctl[SC_TICKET + t];
    else if (t < 4) w = ctl[SC_BIG + t - 2];
    else if (t < 8) w = ctl[SC_ERROR + t - 4];
    const int32_t cur = (t >= 16 && t < 24) ? ctl[SC_CAND + (t - 16) * 32] : 0;
    int32_t c_reg = cur;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c_reg += __shfl_xor(c_reg, d, 64);
    const int32_t entries = __shfl(w, 0, 64), rows = __shfl(w, 1, 64), n_big = __shfl(w, 3, 64),
                  err = __shfl(w, 4, 64), n_long = __shfl(w, 5, 64), max_row = __shfl(w, 6, 64), big_pairs = __shfl(w, 7, 64);
    if (t == 0) {
        mail[0] = c_reg;
        mail[1] = big_pairs;
        mail[2] = n_big;
        mail[3] = err;
        mail[4] = rows;
        mail[5] = entries;
        mail[6] = n_long;
        mail[7] = max_row;
        const bool ok = !(err & 15) && rows == (int32_t)n_query && entries >= 0 && (int64_t)entries <= cap;
        if (ok) indptr[n_query] = entries;
        n_apply_long_out[0] = n_long;
        n_apply_long_out[1] = ok ? 1 : 0; // gate of an apply enqueued right behind this kernel
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (t < 2) ctl[SC_TICKET + t] = 0;
    else if (t < 4) ctl[SC_BIG + t - 2] = 0;
    else if (t < 8) ctl[SC_ERROR + t - 4] = 0;
    if (t >= 16 && t < 24) ctl[SC_CAND + (t - 16) * 32] = 0;
    __threadfence_system();
    if (t == 0) __hip_atomic_store(&mail[MAIL_SEQ_SLOT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace xr
