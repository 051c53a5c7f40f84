"""Worker of tests/test_gpu_sharded_loopback.py::test_shard_plan_on_the_device.  For every mode and W in {1, 3, 8} the owners
computed for rank r -- by W separate calls, as W ranks would make them -- form ONE partition (every source face exactly one owner,
identical owner arrays from every call), the id lists are ascending and consistent with it, Morton shards are compact (few
targets kept per shard), the balanced cut evens out the estimated cost where equal counts do not, and every (target, source)
pair with positive overlap has its target in the owner's list."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import xugrid_amd as xa  # noqa: E402
from xugrid_amd import meshgen  # noqa: E402
from xugrid_amd.distributed import HipBackend, _t  # noqa: E402


def main():
    be = HipBackend(0)
    dev = be.device
    sxy, sf = meshgen.triangle_mesh(40000, 0)
    txy, tf = meshgen.triangle_mesh(30000, 1, 30.0, 0.6)  # (the target covers a part of the source: counts and work differ)
    mxy, mf = meshgen.mixed_mesh(20000, 2)                  # (-1 fill in the connectivity)
    morton_spread = None
    for (axy, af, bxy, bf) in ((sxy, sf, txy, tf), (mxy, mf, txy, tf)):
        full = (_t(axy, dev), _t(af.astype(np.int64), dev), _t(bxy, dev), _t(bf.astype(np.int64), dev))
        S, T = af.shape[0], bf.shape[0]
        pairs_t, pairs_s, _ = xa.CellTree2d(axy, af, -1).intersect_faces(bxy, bf, -1)
        for mode in ("hash", "morton", "balanced"):
            for W in (1, 3, 8):
                owners, counts, costs = [], np.zeros(W, dtype=np.int64), []
                for r in range(W):
                    lf, lt, owner = be.shard_plan(full, W, r, mode, want_owner=True)
                    torch.cuda.synchronize()
                    lf, lt, owner = lf.cpu().numpy(), lt.cpu().numpy(), owner.cpu().numpy()
                    owners.append(owner)
                    assert (np.diff(lf) > 0).all() and (np.diff(lt) > 0).all(), (mode, W, r)
                    assert np.array_equal(lf, np.nonzero(owner == r)[0]), (mode, W, r)
                    counts[r] = lf.size
                    costs.append(lf.size + 4 * lt.size)
                    mine = owner[pairs_s] == r  # no overlapping pair is lost: its target is in my list
                    assert np.isin(pairs_t[mine], lt).all(), (mode, W, r)
                    if mode != "hash" and W == 8:
                        assert lt.size < 0.45 * T, (mode, lt.size, T)  # (a compact shard meets a fraction of the targets)
                for o in owners[1:]:
                    assert np.array_equal(o, owners[0]), (mode, W)  # every rank computes the same owners
                assert counts.sum() == S and owners[0].min() >= 0 and owners[0].max() < W
                if mode == "hash":
                    assert np.array_equal(owners[0], np.arange(S) % W)
                if mode == "morton":
                    assert counts.max() - counts.min() <= 64, counts  # (equal counts up to a cell of faces)
                if W == 8 and af is sf:
                    c = np.asarray(costs, dtype=np.float64)
                    if mode == "morton":
                        morton_spread = c.max() / c.mean()
                    if mode == "balanced":
                        assert c.max() / c.mean() < min(1.25, morton_spread), (c, morton_spread)
    print("shard plan ok")


if __name__ == "__main__":
    main()
