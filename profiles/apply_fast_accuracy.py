"""Accuracy of the contracted many-variable apply (k_apply_plan FAST: fused multiply-adds + one reciprocal per row) at full size:
`python profiles/apply_fast_accuracy.py [delaunay|lattice] [K]` applies K variables of the benchmark's 1M x 1M matrix three ways --
exact (the default: the reference's operation order, regridder.py:41-67), contracted (XR_APPLY_CONTRACT=1) and the CPU oracle on the
downloaded weights -- and prints one JSON line with the largest relative differences.  (The oracle is the checker here, nothing
else; the product never calls it.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root
import numpy as np

import xugrid_amd as xa
from oracle import oracle as O
from xugrid_amd import engine as E

kind = sys.argv[1] if len(sys.argv) > 1 else "delaunay"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
delaunay = kind == "delaunay"
sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=delaunay)
txy, tf = xa.meshgen.triangle_mesh(500_000, 1, 30.0, 0.7, delaunay=delaunay)
csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
S, T = sf.shape[0], tf.shape[0]
cen = O.centroids(sxy, sf)
rng = np.random.default_rng(5)
data = np.stack([np.sin(6 * np.pi * cen[:, 0] + 0.37 * k) * np.cos(4 * np.pi * cen[:, 1]) + 0.1 * rng.standard_normal(S) for k in range(K)])
data_nan = data.copy()
data_nan[:, rng.random(S) < 0.01] = np.nan  # (tiles with NaNs take the exact path in both modes)
w_data, w_idx, w_ptr = csr.download()
res = {"matrix": kind, "K": K, "S": S, "T": T, "nnz": int(csr.nnz)}
for name, d in (("clean", data), ("nan_1pct", data_nan)):
    E.set_option("apply_contract", 1)
    fast = csr.apply(d, 0)
    E.set_option("apply_contract", 0)
    exact = csr.apply(d, 0)
    ref = O.regrid_csr("mean", d, w_data, w_idx, w_ptr, T)
    short = np.diff(w_ptr) <= 32
    ok = ~np.isnan(ref)
    assert np.array_equal(np.isnan(fast), np.isnan(ref)) and np.array_equal(np.isnan(exact), np.isnan(ref))
    rel = lambda a: float(np.max(np.abs(a[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-300)))
    scale = float(np.max(np.abs(ref[ok])))
    res[name] = {
        "exact_vs_oracle_bit_identical_short_rows": bool(np.array_equal(exact[:, short], ref[:, short], equal_nan=True)),
        "exact_vs_oracle_max_rel": rel(exact),
        "fast_vs_oracle_max_rel": rel(fast),
        "fast_vs_oracle_max_abs_over_scale": float(np.max(np.abs(fast[ok] - ref[ok])) / scale),
        "fast_values_differing": int((fast[ok] != ref[ok]).sum()), "values": int(ok.sum()),
    }
print(json.dumps(res))
