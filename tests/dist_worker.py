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


class OracleBackend:
    """Same three methods as xugrid_amd.distributed.HipBackend, computed by the CPU oracle."""

    def build_weights(self, src_xy, src_faces, tgt_xy, tgt_faces):
        q, s, a = O.CellTree2d(src_xy, src_faces).intersect_faces(tgt_xy, tgt_faces)
        T = np.asarray(tgt_faces).shape[0]
        return OracleWeights(a, s, O.to_csr_indptr(q, T), T, np.asarray(src_faces).shape[0])

    def download_weights(self, w):
        return w.data, w.indices, w.indptr, w.n, w.m

    def upload_weights(self, data, indices, indptr, n, m):
        return OracleWeights(np.asarray(data), np.asarray(indices), np.asarray(indptr), n, m)

    def to_device(self, array):
        return torch.as_tensor(np.ascontiguousarray(array))

    def apply(self, w, source, method_id, percentile=0.0):
        names = {0: "mean", 1: "harmonic_mean", 2: "geometric_mean", 3: "sum", 4: "minimum", 5: "maximum", 6: "mode",
                 8: "first_order_conservative", 9: "max_overlap"}
        method = ("percentile", percentile) if method_id == 7 else names[method_id]
        return torch.as_tensor(O.regrid_csr(method, source.numpy(), w.data, w.indices, w.indptr, w.n))

    def partial_mean(self, w, source):
        src = source.numpy().astype(np.float64)
        K = src.shape[0]
        out = np.zeros((2, K, w.n))
        rows = np.repeat(np.arange(w.n), np.diff(w.indptr))
        for k in range(K):
            v = src[k, w.indices]
            ok = ~np.isnan(v)
            out[0, k] = np.bincount(rows[ok], weights=(w.data * v)[ok], minlength=w.n)
            out[1, k] = np.bincount(rows[ok], weights=w.data[ok], minlength=w.n)
        return torch.as_tensor(out)

    def partial_mean_rows(self, w, source):
        nd = self.partial_mean(w, source)  # (2, K, T)
        return nd.permute(2, 0, 1).reshape(nd.shape[2], -1).contiguous()

    def reduce_mean_rows(self, rows, indptr, order, n_targets, K):
        rows, indptr, order = rows.numpy(), indptr.numpy(), order.numpy()
        acc = np.zeros((n_targets, 2 * K))
        for t in range(n_targets):
            for j in order[indptr[t]:indptr[t + 1]]:  # sender order
                acc[t] += rows[j]
        return self.finalize_mean(torch.as_tensor(acc[:, :K].T.copy()), torch.as_tensor(acc[:, K:].T.copy()))

    def accumulate_rows(self, acc, ids, rows):
        acc[ids] += rows

    def finalize_mean_rows(self, acc, K):
        return self.finalize_mean(acc[:, :K].t().contiguous(), acc[:, K:].t().contiguous())

    def finalize_mean(self, num, den):
        n, d = num.numpy(), den.numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            return torch.as_tensor(np.where(d == 0, np.nan, n / d))


class _NoFused:
    """OracleBackend without reduce_mean_rows (attribute lookup fails -> the regridder accumulates sender by sender)."""

    def __init__(self):
        self._b = OracleBackend()

    def __getattr__(self, name):
        if name == "reduce_mean_rows":
            raise AttributeError(name)
        return getattr(self._b, name)


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
    results["morton_legacy"] = ShardedOverlapRegridder(sxy, sf, txy, tf, _NoFused(), partition="morton").regrid(data)
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
