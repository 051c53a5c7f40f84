"""
Multi-GPU regridding: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI),
SOURCE faces partitioned over the ranks, target mesh replicated (BASELINE north_star).

    rank r:  faces_r = {s : part(s) == r}                            # Morton blocks of equal work (or s mod N)
             targets_r = {t : bbox(t) overlaps bbox(faces_r)}         # only these can get weight from r
             W_r     = overlap(source[faces_r], target[targets_r])    # HIP, no communication
             num_r, den_r = sum_j w v, sum_j w  (v not NaN)          # HIP, per (k, target in targets_r)
    all:     reduce_scatter(sum) of [num ; den] over the target axis  # the ONE exchange step
    rank r:  out[k, t] = num / den (NaN where den == 0)  for its slice of targets

With spatially compact shards the per-rank work is ~(S + T) / N plus a boundary layer.  The
exchange step comes in two forms:
  "dense"   reduce_scatter_tensor over the whole target axis, as worded in the north star: every
            rank contributes a [2, K, T] buffer (zeros where it has no weight) -- O(T) bytes per rank;
  "sparse"  (default) the same reduction restricted to the entries that exist: each rank sends, to
            the owner of every target it touched, that target's (num, den) -- an
            all_to_all_single of ~T/N rows per rank; the owner adds the contributions sender by
            sender (deterministic order).  Index lists are exchanged once at set-up.

Only sum-decomposable reducers shard over sources; ``mean`` is implemented that way (it is the reducer
of OverlapRegridder's default and of BarycentricInterpolator).  Every other reducer -- ``mode``, percentiles,
``max_overlap``, ``minimum`` ... -- needs the whole row of a target: ``TargetPartitionedRegridder`` gives each
rank a contiguous slice of the TARGETS plus the sources near them (the same occupancy-raster filter, the other
way round), so a rank holds complete rows, applies any reducer locally, bit-identically to one GPU, and the only
communication is the optional gather of the output slices (SURVEY.md 8e "not shardable over sources").

The compute backend is a parameter: the product uses ``HipBackend`` (C ABI, device pointers of
torch tensors); the world_size-2 gloo tests on CPU inject an oracle-backed backend with the same
three methods.  There is no CPU fallback in the product path.
"""
import os

import numpy as np


def partition_faces(centroids, world_size, mode="morton", weights=None):
    """
    Owner rank of every source face.

    "morton": faces are ordered along a Z-order curve of their centroids and cut into
    ``world_size`` contiguous blocks of equal size -- each rank's shard is spatially compact, so
    the per-rank search touches ~T/world targets instead of all of them.  With ``weights`` (one
    non-negative number per face) the blocks have equal total WEIGHT instead of equal face counts.
    "hash":   ``face id mod world_size`` (BASELINE north_star wording); every rank sees every
    target with ~1/world of its pairs.
    (``ShardedOverlapRegridder(partition="balanced")`` = "morton" with ``work_weights``.)
    """
    n = centroids.shape[0]
    if mode == "hash":
        return (np.arange(n) % world_size).astype(np.int32)
    if mode != "morton":
        raise ValueError(f"unknown partition mode {mode!r}")
    lo = centroids.min(axis=0)
    span = np.maximum(centroids.max(axis=0) - lo, 1e-300)
    q = np.minimum(((centroids - lo) / span * 65536.0).astype(np.uint64), 65535)

    def spread(v):
        v = (v | (v << 8)) & np.uint64(0x00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x33333333)
        v = (v | (v << 1)) & np.uint64(0x55555555)
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    order = np.argsort(code, kind="stable")
    owner = np.empty(n, dtype=np.int32)
    if weights is None:
        owner[order] = (np.arange(n) * world_size // max(n, 1)).astype(np.int32)
    else:
        w = np.asarray(weights, dtype=np.float64)[order]
        before = np.cumsum(w) - w  # weight in front of each face along the curve
        total = float(w.sum())
        cut = before * (world_size / total) if total > 0 else np.arange(n) * (world_size / max(n, 1))
        owner[order] = np.minimum(cut.astype(np.int64), world_size - 1).astype(np.int32)
    return owner


def work_weights(source_centroids, target_centroids, target_cost=4.0, n_grid=None):
    """
    Work estimate per source face for the "balanced" partition: 1 for the face itself (prepare + index) plus
    ``target_cost`` for every target face that falls to it (search, clip, row assembly and apply are per target /
    per candidate pair; 4 is their measured share relative to the per-source kernels on the 1M x 1M benchmark).
    Targets are attributed through a coarse raster (~16 sources per cell, at most 256 x 256 cells): the targets of
    a raster cell are shared by its sources; targets in cells without sources cost nothing (they overlap nothing).
    """
    if n_grid is None:
        n_grid = int(min(256, max(4, np.sqrt(source_centroids.shape[0] / 16.0))))
    lo = np.minimum(source_centroids.min(axis=0), target_centroids.min(axis=0))
    hi = np.maximum(source_centroids.max(axis=0), target_centroids.max(axis=0))
    f = n_grid / np.maximum(hi - lo, 1e-300)

    def cell(c):
        ij = np.clip(np.floor((c - lo) * f).astype(np.int64), 0, n_grid - 1)
        return ij[:, 1] * n_grid + ij[:, 0]

    cs = cell(source_centroids)
    n_src = np.bincount(cs, minlength=n_grid * n_grid)
    n_tgt = np.bincount(cell(target_centroids), minlength=n_grid * n_grid)
    return 1.0 + target_cost * n_tgt[cs] / np.maximum(n_src[cs], 1)


class HipBackend:
    """Device compute through the C ABI; tensors are torch CUDA(HIP) tensors of this rank's GPU."""

    def __init__(self, local_rank):
        import torch

        from . import engine

        self.torch = torch
        self.engine = engine
        torch.cuda.set_device(local_rank)
        engine.init(local_rank)
        self.device = torch.device("cuda", local_rank)
        # The engine runs on torch's current stream: its kernels, torch's own ops and the RCCL collectives (which
        # torch orders against the current stream with events) are then ordered on the device, and no host
        # synchronisation is needed between them.  XR_DIST_OWN_STREAM=1 keeps the engine on its own stream with
        # host synchronisation at every hand-over (measurement hook).
        self.shared_stream = os.environ.get("XR_DIST_OWN_STREAM", "") == ""
        if self.shared_stream:
            engine.set_stream(torch.cuda.current_stream().cuda_stream, async_dev=True)

    def _typed(self, source):
        """-> (tensor, XR dtype id).  float32 / float64 go to the kernels as they are; integer, bool and half inputs
        are cast to float64 first (engine._source_2d does the same for host arrays); anything else is an error --
        never hand the kernels an element size they do not read."""
        torch = self.torch
        if source.dtype == torch.float64:
            return source.contiguous(), self.engine.XR_F64
        if source.dtype == torch.float32:
            return source.contiguous(), self.engine.XR_F32
        if source.dtype in (torch.float16, torch.bfloat16, torch.bool, torch.uint8, torch.int8, torch.int16,
                            torch.int32, torch.int64):
            return source.to(torch.float64).contiguous(), self.engine.XR_F64
        raise TypeError(f"unsupported source dtype {source.dtype}")

    def _handover(self):
        """Work enqueued by torch must be visible to the engine's stream."""
        if not self.shared_stream:
            self.torch.cuda.current_stream().synchronize()

    def build_weights(self, src_xy, src_faces, tgt_xy, tgt_faces):
        E = self.engine
        self._src_mesh = E.DeviceMesh(src_xy, src_faces)
        self._tgt_mesh = E.DeviceMesh(tgt_xy, tgt_faces)
        return self._src_mesh.overlap(self._tgt_mesh, relative=False)

    def rebuild_weights(self):
        """Benchmark hook: redo prepare + index + overlap from the HBM-resident raw meshes."""
        self._src_mesh.invalidate()
        self._tgt_mesh.invalidate()
        return self._src_mesh.overlap(self._tgt_mesh, relative=False)

    def download_weights(self, weights):
        """-> (data, indices, indptr, n, m) host arrays of this rank's shard of the weights."""
        data, indices, indptr = weights.download()
        return data, indices, indptr, weights.n, weights.m

    def upload_weights(self, data, indices, indptr, n, m):
        return self.engine.DeviceCSR.from_arrays(data, indices, indptr, n, m)

    def to_device(self, array):
        return self.torch.as_tensor(np.ascontiguousarray(array), device=self.device)

    def apply(self, weights, source, method_id, percentile=0.0):
        """Any reducer on complete rows: (K, S_local) device tensor -> (K, T_local) float64 device tensor."""
        torch = self.torch
        K = source.shape[0]
        out = torch.empty((K, weights.n), dtype=torch.float64, device=self.device)
        source, dtype = self._typed(source)
        self._handover()
        weights.apply_dev(source.data_ptr(), dtype, K, out.data_ptr(), method_id, percentile)
        self.engine.dev_sync()
        return out

    def partial_mean(self, weights, source):
        """source: (K, S_local) float64/float32 device tensor -> (2, K, T) float64 device tensor."""
        torch = self.torch
        K = source.shape[0]
        out = torch.empty((2, K, weights.n), dtype=torch.float64, device=self.device)
        source, dtype = self._typed(source)
        self._handover()  # inputs produced on torch's stream are ready
        weights.partial_mean_dev(source.data_ptr(), dtype, K, out.data_ptr())
        return out

    def finalize_mean(self, num, den):
        out = self.torch.empty_like(num)
        self._handover()
        self.engine.finalize_mean_dev(num.data_ptr(), den.data_ptr(), num.numel(), out.data_ptr())
        return out

    # ---- row layout used by the sparse exchange
    def partial_mean_rows(self, weights, source):
        """-> (T_local, 2K) float64: [num_0..num_K-1, den_0..den_K-1] per target."""
        torch = self.torch
        K = source.shape[0]
        rows = torch.empty((weights.n, 2 * K), dtype=torch.float64, device=self.device)
        source, dtype = self._typed(source)
        self._handover()
        weights.partial_mean_rows_dev(source.data_ptr(), dtype, K, rows.data_ptr())
        return rows

    def accumulate_rows(self, acc, ids, rows):
        """acc[ids] += rows (ids distinct)."""
        self._handover()
        self.engine.accumulate_rows_dev(acc.data_ptr(), ids.data_ptr(), rows.data_ptr(), rows.shape[0], rows.shape[1])

    def reduce_mean_rows(self, rows, indptr, order, n_targets, K):
        """received rows (R, 2K) + per-target lists (indptr, order) -> finalised (K, n_targets) in one launch."""
        out = self.torch.empty((K, n_targets), dtype=self.torch.float64, device=self.device)
        self._handover()
        self.engine.reduce_mean_rows_dev(rows.data_ptr(), indptr.data_ptr(), order.data_ptr(), n_targets, K, out.data_ptr())
        return out

    def finalize_mean_rows(self, acc, K):
        """(chunk, 2K) -> (K, chunk)."""
        out = self.torch.empty((K, acc.shape[0]), dtype=self.torch.float64, device=self.device)
        self._handover()
        self.engine.finalize_mean_rows_dev(acc.data_ptr(), acc.shape[0], K, out.data_ptr())
        return out


def _face_boxes(xy, faces):
    valid = faces >= 0
    safe = np.where(valid, faces, 0)
    px = np.where(valid, xy[safe, 0], np.nan)
    py = np.where(valid, xy[safe, 1], np.nan)
    return np.nanmin(px, axis=1), np.nanmax(px, axis=1), np.nanmin(py, axis=1), np.nanmax(py, axis=1)


def _targets_near_shard(src_xy, src_faces, tgt_xy, tgt_faces, n_grid=128):
    """
    ids of the target faces that can overlap the given source shard: a coarse occupancy raster of
    the shard's face bboxes (2-D difference array + cumulative sums), queried with the target face
    bboxes through an integral image.  Conservative (cell granularity), and -- unlike one bbox per
    shard -- not spoiled by a few long hull slivers.
    """
    n_t = tgt_faces.shape[0]
    if src_faces.shape[0] == 0 or n_t == 0:
        return np.zeros(0, dtype=np.int64)
    sx0, sx1, sy0, sy1 = _face_boxes(src_xy, src_faces)
    tx0, tx1, ty0, ty1 = _face_boxes(tgt_xy, tgt_faces)
    x_lo, y_lo = min(sx0.min(), tx0.min()), min(sy0.min(), ty0.min())
    x_hi, y_hi = max(sx1.max(), tx1.max()), max(sy1.max(), ty1.max())
    fx = n_grid / max(x_hi - x_lo, 1e-300)
    fy = n_grid / max(y_hi - y_lo, 1e-300)

    def cells(v, lo, f):
        return np.clip(np.floor((v - lo) * f).astype(np.int64), 0, n_grid - 1)

    cx0, cx1 = cells(sx0, x_lo, fx), cells(sx1, x_lo, fx)
    cy0, cy1 = cells(sy0, y_lo, fy), cells(sy1, y_lo, fy)
    diff = np.zeros((n_grid + 1, n_grid + 1), dtype=np.int64)
    np.add.at(diff, (cy0, cx0), 1)
    np.add.at(diff, (cy0, cx1 + 1), -1)
    np.add.at(diff, (cy1 + 1, cx0), -1)
    np.add.at(diff, (cy1 + 1, cx1 + 1), 1)
    occupied = (diff.cumsum(axis=0).cumsum(axis=1)[:n_grid, :n_grid] > 0).astype(np.int64)
    integral = np.zeros((n_grid + 1, n_grid + 1), dtype=np.int64)
    integral[1:, 1:] = occupied.cumsum(axis=0).cumsum(axis=1)
    qx0, qx1 = cells(tx0, x_lo, fx), cells(tx1, x_lo, fx) + 1
    qy0, qy1 = cells(ty0, y_lo, fy), cells(ty1, y_lo, fy) + 1
    hits = integral[qy1, qx1] - integral[qy0, qx1] - integral[qy1, qx0] + integral[qy0, qx0]
    return np.nonzero(hits > 0)[0]


def _reduce_scatter_sum(dist, tensor, world_size, group=None):
    """tensor: (world, ...) contiguous -> this rank's (...) slice of the element-wise sum."""
    import torch

    out = torch.empty_like(tensor[0])
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, tensor, op=dist.ReduceOp.SUM, group=group)
    else:  # gloo (CPU tests) has no reduce_scatter: all_reduce, then keep the own slice
        full = tensor.clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        out.copy_(full[dist.get_rank(group)])
    return out


class ShardedOverlapRegridder:
    """
    ``OverlapRegridder(source, target, method="mean")`` with the source faces sharded over the
    ranks of ``torch.distributed``.

    source_xy/source_faces, target_xy/target_faces: the FULL meshes (every rank passes the same
    arrays; each keeps only its shard of the source faces).
    partition: "balanced" (default; Morton blocks of equal estimated work, ``work_weights``), "morton" (Morton
    blocks of equal source-face counts) or "hash" (face id mod world size).
    """

    def __init__(self, source_xy, source_faces, target_xy, target_faces, backend, partition="balanced", group=None,
                 exchange="sparse"):
        import torch.distributed as dist

        if exchange not in ("sparse", "dense"):
            raise ValueError(f"unknown exchange mode {exchange!r}")
        self.exchange = exchange
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend
        source_faces = np.asarray(source_faces)
        self.n_source = source_faces.shape[0]
        self.n_target = np.asarray(target_faces).shape[0]
        valid = source_faces >= 0
        cnt = valid.sum(axis=1)
        xy = np.asarray(source_xy, dtype=np.float64)
        safe = np.where(valid, source_faces, 0)
        cen = (xy[safe] * valid[..., None]).sum(axis=1) / cnt[:, None]
        if partition == "balanced":
            # Morton blocks of equal estimated WORK: where the target mesh is denser than the source mesh (or covers
            # only a part of it) equal source counts would leave some ranks with several times the targets of others
            tfa = np.asarray(target_faces)
            tv = tfa >= 0
            tcen = (np.asarray(target_xy, dtype=np.float64)[np.where(tv, tfa, 0)] * tv[..., None]).sum(axis=1)
            tcen /= tv.sum(axis=1)[:, None]
            owner = partition_faces(cen, self.world, "morton", weights=work_weights(cen, tcen))
        else:
            owner = partition_faces(cen, self.world, partition)
        self.local_faces = np.nonzero(owner == self.rank)[0]  # global ids of this rank's sources
        # only targets whose bbox overlaps the bbox of this rank's source shard can receive weight
        target_faces = np.asarray(target_faces)
        txy = np.asarray(target_xy, dtype=np.float64)
        self.local_targets = _targets_near_shard(xy, source_faces[self.local_faces], txy, target_faces)
        # target rows are cut into `world` equal slices (padded): rank r finalises slice r
        self.t_chunk = -(-self.n_target // self.world)
        self.weights = backend.build_weights(
            xy, source_faces[self.local_faces], txy, target_faces[self.local_targets]
        )
        self._local_targets_dev = backend.to_device(self.local_targets.astype(np.int64))
        self._setup_sparse_exchange()

    def _setup_sparse_exchange(self):
        """Who gets which of my partial rows: exchanged once (the weights are fixed)."""
        import torch

        dist, W = self.dist, self.world
        owner = self.local_targets // self.t_chunk  # local_targets is ascending -> grouped by owner
        send_counts = np.bincount(owner, minlength=W).astype(np.int64)
        cnt_in = torch.as_tensor(send_counts)
        cnt_out = torch.empty(W, dtype=torch.int64)
        dev = self._local_targets_dev.device
        if dist.get_backend(self.group) == "nccl":
            cnt_in, cnt_out = cnt_in.to(dev), cnt_out.to(dev)
        dist.all_to_all_single(cnt_out, cnt_in, group=self.group)
        self._send_counts = [int(c) for c in send_counts]
        self._recv_counts = [int(c) for c in cnt_out.cpu()]
        ids_in = self.backend.to_device((self.local_targets - owner * self.t_chunk).astype(np.int64))
        ids_out = torch.empty(sum(self._recv_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(ids_out, ids_in, output_split_sizes=self._recv_counts,
                               input_split_sizes=self._send_counts, group=self.group)
        self._recv_ids = ids_out  # positions inside my slice, grouped by sender
        # per owned target: the received rows that belong to it, in sender order (stable sort of the positions)
        ids_host = ids_out.cpu()
        order = torch.argsort(ids_host, stable=True)
        counts = torch.bincount(ids_host, minlength=self.t_chunk)
        indptr = torch.zeros(self.t_chunk + 1, dtype=torch.int64)
        indptr[1:] = torch.cumsum(counts, 0)
        self._recv_order = order.to(dev)
        self._recv_indptr = indptr.to(dev)

    def rebuild(self):
        self.weights = self.backend.rebuild_weights()

    # ---- persistence of the sharded weights: one file per rank, no gather
    @staticmethod
    def shard_path(prefix, rank, world):
        return f"{prefix}.rank{rank}of{world}.npz"

    def to_file(self, prefix) -> str:
        """Every rank writes its shard -- the local CSR (rows = its touched targets, columns = its source faces)
        under the variable names of regridder.py:264-271, plus the two global id lists that place the shard in
        the full matrix -- to ``<prefix>.rank<r>of<W>.npz``.  No communication."""
        data, indices, indptr, n, m = self.backend.download_weights(self.weights)
        path = self.shard_path(prefix, self.rank, self.world)
        np.savez(
            path, __regrid_data=data, __regrid_indices=indices, __regrid_indptr=indptr, __regrid_n=n, __regrid_m=m,
            __regrid_nnz=data.size, __shard_source_faces=self.local_faces, __shard_target_faces=self.local_targets,
            __shard_rank=self.rank, __shard_world=self.world, __n_source=self.n_source, __n_target=self.n_target,
            __shard_exchange=self.exchange,
        )
        return path

    @classmethod
    def from_file(cls, prefix, backend, group=None, exchange=None):
        """Counterpart of ``Regridder.from_weights`` for sharded weights: every rank reads its own file (written
        by a job of the SAME world size) and the exchange lists are set up again; no mesh, no weight construction."""
        import torch.distributed as dist

        self = cls.__new__(cls)
        self.dist, self.group, self.backend = dist, group, backend
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        with np.load(cls.shard_path(prefix, self.rank, self.world)) as f:
            if int(f["__shard_world"]) != self.world or int(f["__shard_rank"]) != self.rank:
                raise ValueError("sharded weights were written by a job of a different shape")
            self.exchange = exchange or str(f["__shard_exchange"])
            self.n_source, self.n_target = int(f["__n_source"]), int(f["__n_target"])
            self.local_faces = f["__shard_source_faces"].astype(np.int64)
            self.local_targets = f["__shard_target_faces"].astype(np.int64)
            n, m = int(f["__regrid_n"]), int(f["__regrid_m"])
            if n != self.local_targets.size or m != self.local_faces.size:
                raise ValueError("shard id lists do not match the stored matrix")
            self.weights = backend.upload_weights(f["__regrid_data"], f["__regrid_indices"], f["__regrid_indptr"], n, m)
        if self.exchange not in ("sparse", "dense"):
            raise ValueError(f"unknown exchange mode {self.exchange!r}")
        self.t_chunk = -(-self.n_target // self.world)
        self._local_targets_dev = backend.to_device(self.local_targets)
        self._setup_sparse_exchange()
        return self

    def local_source(self, data):
        """(K, S) global source data -> this rank's (K, S_local) columns, on the device."""
        data = np.asarray(data)
        if data.ndim == 1:
            data = data[None, :]
        if data.dtype not in (np.float32, np.float64):
            if data.dtype.kind not in "biuf":
                raise TypeError(f"unsupported source dtype {data.dtype}")
            data = data.astype(np.float64)  # as the single-GPU path (engine._source_2d)
        return self.backend.to_device(data[:, self.local_faces])

    def regrid_local(self, local_source):
        """local (K, S_local) device tensor -> this rank's (K, t_chunk) slice of the result."""
        import torch

        K = local_source.shape[0]
        if self.exchange == "sparse":
            # one row of 2K values per touched target, rows grouped by owner rank
            send = self.backend.partial_mean_rows(self.weights, local_source)  # (T_local, 2K)
            recv = torch.empty((sum(self._recv_counts), 2 * K), dtype=send.dtype, device=send.device)
            self.dist.all_to_all_single(recv, send, output_split_sizes=self._recv_counts,
                                        input_split_sizes=self._send_counts, group=self.group)
            if hasattr(self.backend, "reduce_mean_rows"):
                return self.backend.reduce_mean_rows(recv, self._recv_indptr, self._recv_order, self.t_chunk, K)
            acc = torch.zeros((self.t_chunk, 2 * K), dtype=send.dtype, device=send.device)
            start = 0
            for cnt in self._recv_counts:  # sender by sender: ids are unique within a sender
                if cnt:
                    self.backend.accumulate_rows(acc, self._recv_ids[start:start + cnt], recv[start:start + cnt])
                start += cnt
            return self.backend.finalize_mean_rows(acc, K)
        part = self.backend.partial_mean(self.weights, local_source)  # (2, K, T_local)
        t_pad = self.t_chunk * self.world
        nd = torch.zeros((2, K, t_pad), dtype=part.dtype, device=part.device)
        nd.index_copy_(2, self._local_targets_dev, part)  # dense exchange buffer, zeros elsewhere
        # (2, K, world, chunk) -> (world, 2, K, chunk): slice w of the target axis goes to rank w
        send = nd.view(2, K, self.world, self.t_chunk).permute(2, 0, 1, 3).contiguous()
        mine = _reduce_scatter_sum(self.dist, send, self.world, self.group)  # (2, K, chunk)
        return self.backend.finalize_mean(mine[0].contiguous(), mine[1].contiguous())

    def regrid(self, data, gather=True):
        """(K, S) or (S,) global source data -> (K, T) float64 on every rank (gather=True)."""
        import torch

        squeeze = np.asarray(data).ndim == 1
        local = self.regrid_local(self.local_source(data))  # (K, chunk)
        if not gather:
            return local
        parts = [torch.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(parts, local, group=self.group)
        out = torch.cat(parts, dim=1)[:, : self.n_target].cpu().numpy()
        return out[0] if squeeze else out


class TargetPartitionedRegridder:
    """
    ``OverlapRegridder(source, target, method)`` for reducers that need whole rows: rank r owns the targets
    ``[r * chunk, (r + 1) * chunk)`` and the source faces that can overlap them.  No data-path collective; results
    are exactly those of a single GPU.  ``method``: a name of reduce.ABSOLUTE_OVERLAP_METHODS /
    RELATIVE_OVERLAP_METHODS or a ``Method`` from ``create_percentile_method``.
    """

    def __init__(self, source_xy, source_faces, target_xy, target_faces, backend, method="mean", group=None):
        import torch.distributed as dist

        from .reduce import ABSOLUTE_OVERLAP_METHODS, RELATIVE_OVERLAP_METHODS, Method

        if isinstance(method, Method):
            self.method = method
        else:
            table = {**ABSOLUTE_OVERLAP_METHODS, **RELATIVE_OVERLAP_METHODS}
            if method not in table:
                raise ValueError("Invalid regridding method. Available methods are: {}".format(table.keys()))
            self.method = table[method]
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend
        source_faces = np.asarray(source_faces)
        target_faces = np.asarray(target_faces)
        sxy = np.asarray(source_xy, dtype=np.float64)
        txy = np.asarray(target_xy, dtype=np.float64)
        self.n_source, self.n_target = source_faces.shape[0], target_faces.shape[0]
        self.t_chunk = -(-self.n_target // self.world)
        lo = min(self.rank * self.t_chunk, self.n_target)
        hi = min(lo + self.t_chunk, self.n_target)
        self.local_targets = np.arange(lo, hi)
        # sources that can overlap the owned targets (conservative filter; complete rows guaranteed)
        self.local_faces = _targets_near_shard(txy, target_faces[lo:hi], sxy, source_faces)
        self.weights = backend.build_weights(sxy, source_faces[self.local_faces], txy, target_faces[lo:hi])

    def local_source(self, data):
        data = np.asarray(data)
        if data.ndim == 1:
            data = data[None, :]
        if data.dtype not in (np.float32, np.float64):
            if data.dtype.kind not in "biuf":
                raise TypeError(f"unsupported source dtype {data.dtype}")
            data = data.astype(np.float64)  # as the single-GPU path (engine._source_2d)
        return self.backend.to_device(data[:, self.local_faces])

    def regrid_local(self, local_source):
        """local (K, S_local) device tensor -> this rank's (K, owned targets) slice."""
        return self.backend.apply(self.weights, local_source, self.method.method_id, self.method.percentile)

    def regrid(self, data, gather=True):
        import torch

        squeeze = np.asarray(data).ndim == 1
        local = self.regrid_local(self.local_source(data))
        if not gather:
            return local
        K = local.shape[0]
        padded = torch.full((K, self.t_chunk), float("nan"), dtype=local.dtype, device=local.device)
        padded[:, : local.shape[1]] = local
        parts = [torch.empty_like(padded) for _ in range(self.world)]
        self.dist.all_gather(parts, padded, group=self.group)
        out = torch.cat(parts, dim=1)[:, : self.n_target].cpu().numpy()
        return out[0] if squeeze else out


def init_process_group_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend)
    return dist
