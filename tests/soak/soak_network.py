"""Randomised soak of the NetworkGridder weights and of the factored raster apply: xr_edge_length_csr == oracle
intersect_edges (pairs, lengths, row order) on random meshes with random, node-to-node and axis-aligned edges;
matrix-free xr_apply_outer == apply through the materialised product."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from xugrid_amd import engine as E, meshgen
from xugrid_amd.regrid.structured import Raster, StructuredGrid2d
from network_cases import csr_from_pairs, random_network
from structured_cases import random_raster


def run(seed0, n_iter):
    from oracle import oracle as O

    O.build()
    E.init(0)
    bad = 0
    t_start = time.time()
    for it in range(n_iter):
        rng = np.random.default_rng(seed0 * 104729 + it)
        try:
            msg = ""
            n = int(10 ** rng.uniform(1.0, 4.0))
            kind = int(rng.integers(3))
            if kind == 0:
                xy, f = meshgen.triangle_mesh(n, int(rng.integers(1 << 30)), delaunay=bool(rng.integers(2)))
            elif kind == 1:
                m = int(np.sqrt(n)) + 2
                xy, f = meshgen.quad_mesh(np.cumsum(rng.uniform(0.2, 1.0, m)), np.cumsum(rng.uniform(0.2, 1.0, m + 3)))
            else:
                m = int(np.sqrt(n)) + 2
                xy, q = meshgen.quad_mesh(np.linspace(0, 1, m), np.linspace(0, 1, m))
                k = q.shape[0] // 3
                f = np.vstack([np.column_stack([q[:k, 0], q[:k, 1], q[:k, 2], np.full(k, -1)]),
                               np.column_stack([q[:k, 0], q[:k, 2], q[:k, 3], np.full(k, -1)]), q[k:]])
            lo, hi = xy.min(), xy.max()
            span = hi - lo
            ne = int(10 ** rng.uniform(0.5, 4.3))
            edges = random_network(rng, ne, lo - 0.1 * span, hi + 0.1 * span, 10 ** rng.uniform(-2.5, -0.2) * span)
            nn = min(ne, 200)
            if nn:  # edges between mesh nodes (through corners, along sides) and axis-aligned edges
                a, b = rng.integers(0, xy.shape[0], nn), rng.integers(0, xy.shape[0], nn)
                edges[:nn] = np.stack([xy[a], xy[b]], axis=1)
                h = nn // 2
                edges[:h, 1, int(rng.integers(2))] = edges[:h, 0, int(rng.integers(2))]
            tree = O.CellTree2d(xy, f)
            e, fc, pts = tree.intersect_edges(edges)
            w, cols, indptr = csr_from_pairs(e, fc, pts, f.shape[0])
            # (the network kernels' tuning hooks at random: a small stage sends whole waves of edges to the wave-per-edge kernel, a
            # short queue is regrown, the edges in tile order or as they come -- the CSR must not notice; the sort needs >= 4096 edges)
            opts = {"edge_stage": int(rng.choice([0, 0, 1, 40, 300])), "edge_big": int(rng.choice([0, 0, 8, 100000])),
                    "edge_queue": int(rng.choice([0, 0, 500, 20000])), "edge_sort": int(rng.integers(2))}
            for k, v in opts.items():
                E.set_option(k, v)
            csr = E.edge_length_csr(E.DeviceMesh(xy, f), edges)
            for k, v in (("edge_stage", 0), ("edge_big", 0), ("edge_queue", 0), ("edge_sort", 1)):
                E.set_option(k, v)
            dd, di, dp = csr.download()
            if not (np.array_equal(dp, indptr) and np.array_equal(di, cols) and np.array_equal(dd, w)):
                msg = f"network (faces={f.shape[0]} edges={ne} nnz={e.size} vs {csr.nnz}; {opts})"
            if not msg:
                ks, kt = random_raster(rng, 150), random_raster(rng, 150)
                s, t = StructuredGrid2d(Raster(**ks)), StructuredGrid2d(Raster(**kt))
                dev = [s.overlap_device(t, bool(rng.integers(2))), s.linear_weights_device(t), s.locate_centroids_device(t)][int(rng.integers(3))]
                K = int(rng.integers(1, 7))
                field = rng.normal(size=(K, s.size)) + 1.5
                field[rng.random(field.shape) < 0.05] = np.nan
                mid = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 9]))
                E.set_option("outer_apply", 1)  # matrix-free
                free = dev.apply(field, mid)
                E.set_option("outer_apply", 0)
                ref = dev.csr().apply(field, mid)
                data, indices, dip = dev.download()
                short = np.diff(dip) <= 32
                same = (free == ref) | (np.isnan(free) & np.isnan(ref))
                if mid != 2 and not same[:, short].all():
                    msg = f"factored apply method {mid} (short rows differ)"
                elif not np.allclose(free, ref, rtol=1e-9 if mid == 1 else 1e-12, atol=1e-13, equal_nan=True):
                    msg = f"factored apply method {mid}"
            if msg:
                bad += 1
                print(f"MISMATCH seed={seed0} it={it}: {msg}", flush=True)
        except Exception as ex:
            import traceback
            bad += 1
            print(f"ERROR seed={seed0} it={it}: {type(ex).__name__}: {ex}", flush=True)
            traceback.print_exc()
    print(f"soak_network done: {n_iter} iterations, {bad} failures, {time.time() - t_start:.0f}s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 50) else 0)
