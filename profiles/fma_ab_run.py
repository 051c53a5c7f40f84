"""One variant of the floating-point contraction A/B (profiles/fma_ab.sh): build the benchmark pair's weights with the library named
by XUGRID_AMD_LIB (default: the product's), compare pair set and areas with the CPU oracle at full size, time search and clip.
Prints one JSON line.  `python profiles/fma_ab_run.py [steps] [--no-oracle]`"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root
import numpy as np

import xugrid_amd as xa
from xugrid_amd import engine as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
with_oracle = "--no-oracle" not in sys.argv
sxy, sf = xa.meshgen.triangle_mesh(500_000, 0)
txy, tf = xa.meshgen.triangle_mesh(500_000, 1, 30.0, 0.7)
ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
csr = ms.overlap(mt)
out = {"lib": os.environ.get("XUGRID_AMD_LIB", "default (-ffp-contract=off)"), "nnz": int(csr.nnz), "candidates": int(ms.last_candidates())}
if with_oracle:
    from oracle import oracle as O

    data, indices, indptr = csr.download()
    rows = np.repeat(np.arange(csr.n, dtype=np.int64), np.diff(indptr))
    oq, os_, oa = O.CellTree2d(sxy, sf).intersect_faces(txy, tf)
    key_dev = rows * csr.m + indices
    key_ora = oq.astype(np.int64) * csr.m + os_
    # both are sorted by (row, column)
    only_dev = np.setdiff1d(key_dev, key_ora, assume_unique=True)
    only_ora = np.setdiff1d(key_ora, key_dev, assume_unique=True)
    common_d = np.isin(key_dev, key_ora, assume_unique=True)
    common_o = np.isin(key_ora, key_dev, assume_unique=True)
    a_dev, a_ora = data[common_d], oa[common_o]
    rel = np.abs(a_dev - a_ora) / a_ora
    scale = float(np.median(oa))
    out.update({
        "oracle_pairs": int(key_ora.size),
        "pairs_only_on_device": int(only_dev.size), "pairs_only_in_oracle": int(only_ora.size),
        "largest_area_of_a_pair_only_on_device_rel_to_median": float(data[~common_d].max() / scale) if only_dev.size else 0.0,
        "largest_area_of_a_pair_only_in_oracle_rel_to_median": float(oa[~common_o].max() / scale) if only_ora.size else 0.0,
        "areas_bit_identical": int((a_dev == a_ora).sum()), "areas_differing": int((a_dev != a_ora).sum()),
        "max_rel_area_diff": float(rel.max()), "p999_rel_area_diff": float(np.quantile(rel, 0.999)),
        "max_rel_area_diff_among_areas_above_1e-6_median": float(rel[a_ora > 1e-6 * scale].max()),
        "max_abs_row_sum_diff_rel": float(np.abs(np.bincount(rows, data, csr.n) - np.bincount(oq, oa, csr.n)).max() / scale),
    })
E.set_async(True)
for _ in range(10):
    ms.invalidate(); mt.invalidate(); ms.overlap(mt)
E.dev_sync()
E.prof_enable(True)
E.prof_reset()
t0 = time.perf_counter()
for _ in range(steps):
    ms.invalidate(); mt.invalidate(); ms.overlap(mt)
E.dev_sync()
dt = (time.perf_counter() - t0) / steps
kt = E.kernel_times()
E.prof_enable(False)
out["weights_ms_profiled"] = 1e3 * dt
out["kernel_ms_per_step"] = {k: round(v[1] / steps, 5) for k, v in kt.items() if k in ("search", "clip_tri", "assemble", "search_big", "clip_big")}
E.prof_reset()
t0 = time.perf_counter()
for _ in range(steps):
    ms.invalidate(); mt.invalidate(); ms.overlap(mt)
E.dev_sync()
out["weights_ms"] = 1e3 * (time.perf_counter() - t0) / steps
print(json.dumps(out))
