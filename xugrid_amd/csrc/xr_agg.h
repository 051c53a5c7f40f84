// xr_agg.h -- block-level aggregation of global atomics (device code only; included by the kernels that use it: xr_mesh.hip's
// two sorting passes, xr_edges.hip's clip).
#pragma once

#include <cstdint>

namespace xr {

// ---- one global atomic per DISTINCT key of a block instead of one per face.  Faces arrive spatially coherent: the 256 faces
// of a block fall into ~80 cells; device-scope atomics on scattered addresses are what bounds both sorting passes
// (1 M of them: ~30 us).  Open-addressing table in LDS; key -1 = no face.
static constexpr int AGG_SLOTS = 512;
struct KeyTable {
    int32_t key[AGG_SLOTS];
    int32_t cnt[AGG_SLOTS];
    int32_t base[AGG_SLOTS];
};
__device__ __forceinline__ void agg_clear(KeyTable &t) {
    for (int s = threadIdx.x; s < AGG_SLOTS; s += 256) {
        t.key[s] = -1;
        t.cnt[s] = 0;
    }
}
// -> slot of the key in the table; `rank` = position of this face among the block's faces with the same key
__device__ __forceinline__ int agg_insert(KeyTable &t, int k, int &rank) {
    int s = (int)(((unsigned)k * 2654435761u) >> 23) & (AGG_SLOTS - 1);
    while (true) {
        const int prev = atomicCAS(&t.key[s], -1, k);
        if (prev == -1 || prev == k) break;
        s = (s + 1) & (AGG_SLOTS - 1);
    }
    rank = atomicAdd(&t.cnt[s], 1);
    return s;
}

} // namespace xr
