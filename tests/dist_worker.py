"""Worker of the world_size-2 gloo tests: ShardedOverlapRegridder with an ORACLE-backed compute
backend (tests only) -- exercises partitioning, padding, the reduce-scatter layout and the
finalise step on CPU.  Launched by test_distributed_cpu.py via torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from xugrid_amd import meshgen  # noqa: E402
from xugrid_amd.distributed import (  # noqa: E402
    ShardedOverlapRegridder,
    TargetPartitionedRegridder,
    init_process_group_from_env,
)


class OracleWeights:
    def __init__(self, data, indices, indptr, n, m=None):
        self.data, self.indices, self.indptr, self.n, self.m = data, indices, indptr, n, m

    @property
    def nnz(self):
        return int(np.asarray(self.data).size)


COMPONENTS = {0: 2, 8: 2, 3: 2, 1: 2, 2: 4, 4: 2, 5: 2}  # method id -> partial-state components (include/xugrid_amd.h)
NAMES = {0: "mean", 1: "harmonic_mean", 2: "geometric_mean", 3: "sum", 4: "minimum", 5: "maximum", 6: "mode",
         8: "first_order_conservative", 9: "max_overlap"}


class OracleBackend:
    """Same methods as xugrid_amd.distributed.HipBackend, computed on the CPU: weights and whole-row reducers by the
    oracle, the partial states of the shard-decomposable reducers by a numpy restatement of the component table of
    include/xugrid_amd.h (reduce.py:16-123, 206-222)."""

    device = None

    def build_weights(self, src_xy, src_faces, tgt_xy, tgt_faces, relative=False):
        self._last = (src_xy, src_faces, tgt_xy, tgt_faces, relative)
        q, s, a = O.CellTree2d(src_xy, src_faces).intersect_faces(tgt_xy, tgt_faces)
        if relative:
            a = a / O.area(src_xy, src_faces)[s]
        T = np.asarray(tgt_faces).shape[0]
        return OracleWeights(a, s, O.to_csr_indptr(q, T), T, np.asarray(src_faces).shape[0])

    def rebuild_weights(self):
        return self.build_weights(*self._last)

    def download_weights(self, w):
        return w.data, w.indices, w.indptr, w.n, w.m

    def upload_weights(self, data, indices, indptr, n, m):
        return OracleWeights(np.asarray(data), np.asarray(indices), np.asarray(indptr), n, m)

    def to_device(self, array):
        return torch.as_tensor(np.ascontiguousarray(array))

    def apply(self, w, source, method_id, percentile=0.0):
        method = ("percentile", percentile) if method_id == 7 else NAMES[method_id]
        return torch.as_tensor(O.regrid_csr(method, source.numpy(), w.data, w.indices, w.indptr, w.n))

    def n_components(self, method_id):
        return COMPONENTS[method_id]

    def combine_is_max(self, method_id):
        return method_id in (4, 5)

    def identity(self, method_id, K, n):
        out = np.zeros((COMPONENTS[method_id], K, n))
        if self.combine_is_max(method_id):
            out[0] = -np.inf
        return torch.as_tensor(out)

    def partial(self, w, source, method_id, rows_layout):
        src = source.numpy().astype(np.float64)
        K, C = src.shape[0], COMPONENTS[method_id]
        out = self.identity(method_id, K, w.n).numpy()
        rows = np.repeat(np.arange(w.n), np.diff(w.indptr))
        for k in range(K):
            v, wt = src[k, w.indices], w.data
            ok = ~np.isnan(v)

            def acc(mask, values):
                return np.bincount(rows[mask], weights=values[mask], minlength=w.n)

            with np.errstate(all="ignore"):
                if method_id in (0, 8):
                    out[0, k], out[1, k] = acc(ok, wt * v), acc(ok, wt)
                elif method_id == 3:
                    out[0, k], out[1, k] = acc(ok, v), acc(ok, wt)
                elif method_id == 1:
                    m = ok & (v != 0) & (wt > 0)
                    out[0, k], out[1, k] = acc(m, wt), acc(m, wt / v)
                elif method_id == 2:
                    m = (v > 0) & (wt > 0)
                    out[0, k] = acc(np.ones_like(ok), wt)
                    out[1, k], out[2, k] = acc(m, wt * np.log(np.abs(v))), acc(m, wt)
                    out[3, k] = acc(~m & (v < 0), np.ones_like(wt))
                else:
                    sv = -v if method_id == 4 else v
                    np.maximum.at(out[0, k], rows[ok], sv[ok])
                    np.maximum.at(out[1, k], rows[ok], wt[ok])
        if rows_layout:
            return torch.as_tensor(np.ascontiguousarray(out.transpose(2, 0, 1).reshape(w.n, C * K)))
        return torch.as_tensor(out)

    def finalize(self, method_id, planes):
        c = planes.numpy()
        with np.errstate(all="ignore"):
            if method_id == 0:
                out = np.where(c[1] == 0, np.nan, c[0] / c[1])
            elif method_id in (8, 3):
                out = np.where(c[1] == 0, np.nan, c[0])
            elif method_id == 1:
                out = np.where((c[1] == 0) | (c[0] == 0), np.nan, c[0] / c[1])
            elif method_id == 2:
                out = np.where((c[0] == 0) | (c[3] > 0) | (c[2] == 0), np.nan, np.exp(c[1] / c[2]))
            elif method_id == 4:
                out = np.where(c[1] == 0, np.nan, -c[0])
            else:
                out = np.where(c[1] == 0, np.nan, c[0])
        return torch.as_tensor(out)

    def reduce_rows(self, method_id, rows, indptr, order, n_targets, K):
        rows, indptr, order = rows.numpy(), indptr.numpy(), order.numpy()
        C = COMPONENTS[method_id]
        acc = self.identity(method_id, K, n_targets).numpy()
        for t in range(n_targets):
            for j in order[indptr[t]:indptr[t + 1]]:  # sender order
                r = rows[j].reshape(C, K)
                acc[:, :, t] = np.maximum(acc[:, :, t], r) if self.combine_is_max(method_id) else acc[:, :, t] + r
        return self.finalize(method_id, torch.as_tensor(acc))


def main():
    out_dir = sys.argv[1]
    dist_ = init_process_group_from_env("gloo")
    rank, world = dist_.get_rank(), dist_.get_world_size()
    sxy, sf = meshgen.triangle_mesh(1500, 0)
    txy, tf = meshgen.triangle_mesh(1203, 1, 30.0, 0.7)  # T not divisible by the world size
    data = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(3)])
    results = {}
    for mode, exchange in (("morton", "sparse"), ("hash", "sparse"), ("morton_dense", "dense"), ("balanced", "sparse")):
        rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition=mode.split("_")[0], exchange=exchange)
        owned = torch.zeros(sf.shape[0], dtype=torch.int64)
        owned[torch.as_tensor(rg.local_faces)] = 1
        dist.all_reduce(owned)
        assert bool((owned == 1).all()), "every source face must be owned by exactly one rank"
        results[mode] = rg.regrid(data)
        results[mode + "_1d"] = rg.regrid(data[0])
        results[mode + "_n_local"] = rg.local_faces.size
        results[mode + "_n_local_targets"] = rg.local_targets.size
    # sharded weights written rank by rank and read back without the meshes
    rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition="morton")
    rg.to_file(os.path.join(out_dir, "sharded"))
    dist.barrier()
    rg2 = ShardedOverlapRegridder.from_file(os.path.join(out_dir, "sharded"), OracleBackend())
    results["morton_from_file"] = rg2.regrid(data)
    results["morton_from_file_dense"] = ShardedOverlapRegridder.from_file(
        os.path.join(out_dir, "sharded"), OracleBackend(), exchange="dense").regrid(data)
    # every shard-decomposable reducer, both exchanges; the variables in tiles of 2 (pipelined collectives)
    data7 = np.stack([meshgen.smooth_field(sxy[sf].mean(axis=1), k, 0.05) for k in range(7)])
    data7[1] = np.abs(data7[1]) + 0.1      # (a variable without negatives: geometric mean defined)
    data7[2, ::3] = 0.0                    # zeros (harmonic / geometric means skip them)
    data7[5] = np.nan                      # an all-NaN variable
    for method in ("sum", "first_order_conservative", "harmonic_mean", "geometric_mean", "minimum", "maximum", "mean"):
        for exchange in ("sparse", "dense"):
            rg = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition="morton", exchange=exchange,
                                         method=method, k_tile=2)
            results[f"m_{method}_{exchange}"] = rg.regrid(data7)
    results["int_source"] = ShardedOverlapRegridder(sxy, sf, txy, tf, OracleBackend(), partition="morton").regrid(
        np.nan_to_num(10 * data).astype(np.int32))
    for method in ("mode", "median", "max_overlap", "minimum", "sum", "mean"):
        tp = TargetPartitionedRegridder(sxy, sf, txy, tf, OracleBackend(), method=method)
        results["tp_" + method] = tp.regrid(data)
        results["tp_n_local_sources"] = tp.local_faces.size
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist_out.npz"), world=world, **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
