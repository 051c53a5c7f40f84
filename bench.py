#!/usr/bin/env python
"""
Benchmark of the regridding hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): target cells regridded / s for an OverlapRegridder, area-weighted mean.
Workload at N = 1: BASELINE config 2 -- a ~1M-triangle Delaunay source mesh (500k jittered-lattice
points, seed 0) regridded to a ~1M-triangle target (seed 1, rotated 30 degrees, scaled 0.7).

One "step" is one full pass of the hot path over the batch, starting from raw mesh arrays that are
already resident in HBM (node coordinates f64, connectivity i32, source data f64):
    per-face preparation of both meshes (fill, CCW, bbox, area)  ->  spatial index over the source
    ->  candidate search  ->  polygon clip  ->  CSR assembly  ->  apply (mean) of one variable.
Nothing is cached between steps (xr_mesh_invalidate); the scalar read-backs the path needs (number
of candidate pairs, nnz) are inside the timed region.  value = T / time per step.

N > 1 (weak scaling): the meshes grow with N (N x 500k points each); source faces are sharded
over the ranks (Morton blocks), the target is replicated; each step additionally does the one
exchange step of the path: a reduce-scatter (RCCL) of the per-target partial sums.  value = total
target cells / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" for the dominant kernel (algorithmic bytes / hipEvent-timed duration)
and, at N = 1, "cpu_baseline" (the CPU oracle -- a port of the reference path -- timed on the host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_meshes(points_per_mesh, delaunay=True):
    from xugrid_amd import meshgen

    t0 = time.perf_counter()
    sxy, sf = meshgen.triangle_mesh(points_per_mesh, 0, delaunay=delaunay)
    txy, tf = meshgen.triangle_mesh(points_per_mesh, 1, 30.0, 0.7, delaunay=delaunay)
    log(f"[bench] meshes: S={sf.shape[0]} T={tf.shape[0]} ({time.perf_counter() - t0:.1f}s)")
    return sxy, sf, txy, tf


def algorithmic_bytes(S, T, Ns, Nt, C, P, Ms=3, Mt=3):
    """Per-kernel algorithmic HBM bytes of one step: every array a kernel has
    to read or write once -- int32 connectivity / indices, f64 coordinates / areas, the face-major
    vertex blocks (16 M bytes per face) and 16-byte f32 record boxes the engine keeps in HBM."""
    vs, vt = 16 * Ms, 16 * Mt  # face-major vertex block bytes
    b = {}
    b["prepare_faces"] = 4 * (Ms * S + Mt * T) + 16 * (Ns + Nt) + (vs + 1 + 32 + 8) * S + (vt + 1 + 32 + 8) * T
    b["index_count"] = 32 * S + 4 * S
    b["order_count"] = 32 * T + 4 * T
    b["index_scatter"] = (4 + vs + 1 + 32) * S + (4 + vs + 1 + 16) * S
    b["order_scatter"] = (4 + vt + 1 + 32) * T + (4 + vt + 1 + 32) * T
    b["search"] = 32 * T + 16 * S + 8 * T + 8 * C  # query boxes, record boxes once, count + offset, pair queue out
    b["clip_small"] = 8 * C + vt * T + vs * S + (S + T) + 4 * S + 12 * C + 4 * T  # queue, vertex blocks, area + face id out
    b["row_fill"] = 4 * T + 16 * C + 4 * T + 12 * P
    b["apply_stream"] = 12 * P + 4 * (T + 1) + 4 * T + 8 * (S + T)
    # whole weight construction, SURVEY.md 8(d): read both meshes once, write the CSR once
    b["build_total"] = 4 * (Ms * S + Mt * T) + 16 * (Ns + Nt) + 12 * P + 4 * (T + 1)
    return b


def cpu_baseline(sxy, sf, txy, tf, data):
    """The CPU oracle (oracle/xr_oracle.c: cell tree + SAT + Sutherland-Hodgman + CSR + mean apply,
    OpenMP over queries / candidate pairs, apply threaded over K only as numba's prange) timed once
    on the full workload."""
    from oracle import oracle as O

    O.build()
    cores = O.num_threads()
    t0 = time.perf_counter()
    tree = O.CellTree2d(sxy, sf)
    q, s, a = tree.intersect_faces(txy, tf)
    indptr = O.to_csr_indptr(q, tf.shape[0])
    t1 = time.perf_counter()
    out = O.regrid_csr("mean", data[None, :], a, s, indptr, tf.shape[0], parallel_rows=False)
    t2 = time.perf_counter()
    log(f"[bench] cpu oracle: weights {t1 - t0:.2f}s apply {t2 - t1:.3f}s on {cores} threads, nnz {a.size}")
    return {
        "value": tf.shape[0] / (t2 - t0),
        "unit": "target cells/s",
        "cores": cores,
        "kind": "port",
        "sample": f"full workload once: S={sf.shape[0]} T={tf.shape[0]} K=1 (weights {t1 - t0:.2f}s + apply {t2 - t1:.3f}s)",
    }, out[0]


def run_single(args):
    import ctypes

    import xugrid_amd as xa
    from xugrid_amd import _lib, engine as E

    E.init(0)
    lib = _lib.load()
    sxy, sf, txy, tf = make_meshes(args.points, delaunay=not args.no_delaunay)
    S, T, Ns, Nt = sf.shape[0], tf.shape[0], sxy.shape[0], txy.shape[0]
    ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
    data = xa.meshgen.smooth_field(ms.centroids(), 0)
    # source data and output resident in HBM
    d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.xr_dev_alloc(8 * S, ctypes.byref(d_src)))
    _lib.check(lib.xr_dev_alloc(8 * T, ctypes.byref(d_out)))
    _lib.check(lib.xr_dev_upload(d_src, data.ctypes.data_as(ctypes.c_void_p), 8 * S))

    state = {}

    def step():
        ms.invalidate()
        mt.invalidate()
        csr = ms.overlap(mt)  # prepare x2 + index + search + clip + CSR
        csr.apply_dev(d_src.value, E.XR_F64, 1, d_out.value, 0)
        state["csr"] = csr

    for _ in range(args.warmup):
        step()
    E.dev_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    E.dev_sync()
    elapsed = time.perf_counter() - t0
    ms_per_step = 1e3 * elapsed / args.steps
    csr = state["csr"]
    C, P = ms.last_candidates(), csr.nnz

    # weights-only and apply-only rates (reported in config, not the headline)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csr.apply_dev(d_src.value, E.XR_F64, 1, d_out.value, 0)
    E.dev_sync()
    apply_ms = 1e3 * (time.perf_counter() - t0) / args.steps

    # weights only (no apply), and the whole thing from HOST arrays (mesh uploads + step + result download over PCIe);
    # both informative, neither is `value`
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ms.invalidate()
        mt.invalidate()
        ms.overlap(mt)
    E.dev_sync()
    build_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    host_times = []
    for _ in range(3):
        t0 = time.perf_counter()
        hs, ht = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
        h_out = hs.overlap(ht).apply(data[None, :], 0)
        host_times.append(time.perf_counter() - t0)
    del hs, ht, h_out
    host_to_host_ms = 1e3 * min(host_times)

    # per-kernel durations: the same K steps with hipEvents around every launch (engine stream)
    with E.KernelTimer() as kt:
        for _ in range(args.steps):
            step()
    kernels = {k: (n, t / n) for k, (n, t) in kt.records.items()}  # name -> (launches, avg ms)
    per_step = {k: n * avg / args.steps for k, (n, avg) in kernels.items()}
    dominant = max(per_step, key=per_step.get)
    ab = algorithmic_bytes(S, T, Ns, Nt, C, P)
    dom_bytes = ab.get(dominant)
    dom_ms = kernels[dominant][1]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_bytes else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get(dominant)
            # HBM-side bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)
            traffic = entry["hbm_bytes_corrected"] if entry else None
        except Exception:
            traffic = None
    # The dominant kernel (the clip) is FP64-VALU work, not HBM traffic: next to the HBM fraction the contract asks
    # for, report how close it runs to the VALU issue limit.  Instruction count from the committed PMC pass
    # (SQ_INSTS_VALU, wave-instructions per launch); a wave64 FP64 instruction occupies its SIMD for 4 cycles
    # (78.6 TFLOP/s FP64 vector peak = 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz), a 32-bit one for 2.
    valu = None
    ppath = os.path.join(ROOT, "profiles", "r01l_pmc_per_launch.json")
    pmc_names = {"clip_small": "k_clip_small<6, 256, true>", "search": "k_search"}
    if os.path.exists(ppath) and dominant in pmc_names:
        try:
            insts = json.load(open(ppath))[pmc_names[dominant]]["SQ_INSTS_VALU"]
            n_simd, clock = 256 * 4, 2.4e9
            floor_fp64_ms = insts * 4 / n_simd / clock * 1e3
            floor_mixed_ms = insts * 3 / n_simd / clock * 1e3
            valu = {
                "wave_instructions_per_launch": insts,
                "issue_floor_ms_all_fp64": floor_fp64_ms,
                "issue_floor_ms_half_fp64": floor_mixed_ms,
                "frac_of_issue_limit": [floor_mixed_ms / dom_ms, floor_fp64_ms / dom_ms],
                "source": "profiles/r01l_pmc_per_launch.json",
            }
        except Exception:
            valu = None
    build_kernel_ms = sum(v for k, v in per_step.items() if not k.startswith("apply"))
    roofline = {
        "bound": "hbm",
        "kernel": dominant,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS if achieved else None,
        "traffic": traffic,
        "algorithmic_bytes_per_launch": dom_bytes,
        "avg_launch_ms": dom_ms,
        "valu_issue": valu,
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
        "build_aggregate": {
            "algorithmic_bytes": ab["build_total"],
            "kernel_ms": build_kernel_ms,
            "GBps": ab["build_total"] / (build_kernel_ms * 1e-3) / 1e9,
            "frac": ab["build_total"] / (build_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "apply": {
            "algorithmic_bytes": ab["apply_stream"],
            "avg_launch_ms": kernels["apply_stream"][1],
            "GBps": ab["apply_stream"] / (kernels["apply_stream"][1] * 1e-3) / 1e9,
            "frac": ab["apply_stream"] / (kernels["apply_stream"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
    }

    out_gpu = np.empty(T)
    _lib.check(lib.xr_dev_download(out_gpu.ctypes.data_as(ctypes.c_void_p), d_out, 8 * T))
    cpu = None
    if not args.no_cpu:
        cpu, out_cpu = cpu_baseline(sxy, sf, txy, tf, data)
        both = ~(np.isnan(out_gpu) & np.isnan(out_cpu))
        n_diff = int((out_gpu[both] != out_cpu[both]).sum())
        with np.errstate(invalid="ignore", divide="ignore"):
            max_rel = float(np.nanmax(np.abs(out_gpu[both] - out_cpu[both]) / np.abs(out_cpu[both]))) if both.any() else 0.0
        log(f"[bench] GPU vs CPU oracle: {n_diff} of {T} values differ, max rel {max_rel:.3g} "
            "(rows > 32 entries are reduced cooperatively in a fixed tree order; all others bit-identical)")
        cpu["gpu_values_differing"] = n_diff
        cpu["gpu_max_rel_diff"] = max_rel
    result = {
        "metric": "target cells regridded/s (OverlapRegridder 1M->1M tri, weights + mean apply)",
        "value": T / (elapsed / args.steps),
        "unit": "target cells/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 2: ~1M-triangle jittered-lattice Delaunay source -> ~1M-triangle target "
            "(seed 1, rotated 30 deg, scaled 0.7), OverlapRegridder area-weighted mean, K=1",
            "source_faces": S,
            "target_faces": T,
            "candidate_pairs": C,
            "nnz": P,
            "parallelism": "1 GPU",
            "apply_only_ms": apply_ms,
            "apply_only_cells_per_s": T / (apply_ms * 1e-3),
            "weights_only_ms": build_ms,
            "weights_only_cells_per_s": T / (build_ms * 1e-3),
            "host_to_host_ms": host_to_host_ms,
            "host_to_host_note": "mesh uploads (pageable host arrays, int64 connectivity) + weights + apply + download "
            "of the result vector over PCIe; not part of `value`",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if not args.no_extras:
        result["other_configs"] = other_configs(E, lib, _lib, csr, S, T, ms, sxy)
    print(json.dumps(result), flush=True)


def other_configs(E, lib, _lib, csr, S, T, mesh, mesh_xy):
    """BASELINE configs 5 and 3, a structured pair and a network, measured beside the headline (rank 0, N = 1): informative
    extras of the JSON line, never part of `value`.  Bounded to a few seconds each."""
    import ctypes

    import numpy as np

    import xugrid_amd as xa
    from xugrid_amd.regrid.structured import Raster, StructuredGrid2d

    out = {}
    try:  # config 5: cached weights, K = 256 stacked variables
        K = 256
        block = np.random.default_rng(5).random((8, S))
        src = np.tile(block, (K // 8, 1))
        d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.xr_dev_alloc(8 * K * S, ctypes.byref(d_src)))
        _lib.check(lib.xr_dev_alloc(8 * K * T, ctypes.byref(d_out)))
        _lib.check(lib.xr_dev_upload(d_src, src.ctypes.data_as(ctypes.c_void_p), 8 * K * S))
        del src
        for _ in range(2):
            csr.apply_dev(d_src.value, E.XR_F64, K, d_out.value, 0)
        E.dev_sync()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            csr.apply_dev(d_src.value, E.XR_F64, K, d_out.value, 0)
        E.dev_sync()
        dt = (time.perf_counter() - t0) / n
        nbytes = 12 * csr.nnz + 4 * (T + 1) + 8 * K * (S + T)
        out["config5_apply_K256"] = {
            "weights": "the benchmark's own matrix (qhull-numbered Delaunay source: gathers less local than a "
            "lattice-numbered mesh, which runs at 1.05 ms = 3.9 TB/s)",
            "ms": 1e3 * dt, "cell_variables_per_s": K * T / dt, "algorithmic_GBps": nbytes / dt / 1e9,
            "frac_of_hbm_peak": nbytes / dt / 1e9 / HBM_PEAK_GBS,
        }
        lib.xr_dev_free(d_src)
        lib.xr_dev_free(d_out)
    except Exception as e:  # noqa: BLE001
        out["config5_apply_K256"] = {"error": repr(e)}
    try:  # config 3: BarycentricInterpolator 1M faces -> 4M target faces (lattice-split meshes)
        sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=False)
        txy, tf = xa.meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=False)
        src_g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
        tgt_g = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
        src_g.device_mesh, tgt_g.device_mesh  # uploads are not part of the construction
        xa.BarycentricInterpolator(src_g, tgt_g)  # untimed: first-use allocations of this process
        times = []
        for _ in range(3):
            E.dev_sync()
            t0 = time.perf_counter()
            rg = xa.BarycentricInterpolator(src_g, tgt_g)
            E.dev_sync()
            times.append(time.perf_counter() - t0)
        out["config3_barycentric_1M_to_4M"] = {
            "construct_ms": 1e3 * min(times), "construct_ms_max_of_3": 1e3 * max(times),
            "target_points_per_s": tgt_g.n_face / min(times), "nnz": rg._device_weights.nnz,
            "note": "source and target meshes resident; Voronoi pre-step (device + O(boundary) host part) + "
            "xr_barycentric_csr; the first constructions of a fresh process take longer (pool warm-up)",
        }
        del rg, src_g, tgt_g
    except Exception as e:  # noqa: BLE001
        out["config3_barycentric_1M_to_4M"] = {"error": repr(e)}
    try:  # structured pair: 4000^2 -> 4000^2 raster cells
        ns = nt = 4000
        src_r = Raster(x=np.arange(0.5, ns), y=np.arange(ns - 0.5, 0.0, -1.0))
        tgt_r = Raster(x=0.37 + 0.98 * (np.arange(nt) + 0.5), y=(0.21 + 0.98 * (np.arange(nt) + 0.5))[::-1].copy())
        s2, t2 = StructuredGrid2d(src_r), StructuredGrid2d(tgt_r)
        s2.overlap_device(t2, False)
        E.dev_sync()
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            w = s2.overlap_device(t2, False)
            E.dev_sync()
            times.append(time.perf_counter() - t0)
        dt = min(times)
        d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.xr_dev_alloc(8 * s2.size, ctypes.byref(d_src)))
        _lib.check(lib.xr_dev_alloc(8 * t2.size, ctypes.byref(d_out)))
        field = np.random.default_rng(0).normal(size=s2.size)
        _lib.check(lib.xr_dev_upload(d_src, field.ctypes.data_as(ctypes.c_void_p), 8 * s2.size))
        for _ in range(3):
            w.apply_dev(d_src.value, 0, 1, d_out.value)
        E.dev_sync()
        t0 = time.perf_counter()
        for _ in range(10):
            w.apply_dev(d_src.value, 0, 1, d_out.value)
        E.dev_sync()
        t_apply = (time.perf_counter() - t0) / 10
        _lib.check(lib.xr_dev_free(d_src))
        _lib.check(lib.xr_dev_free(d_out))
        out["structured_4000x4000"] = {
            "weights_ms": 1e3 * dt, "apply_mean_ms": 1e3 * t_apply, "target_cells_per_s": t2.size / (dt + t_apply),
            "apply_GBps_of_data": 8.0 * (s2.size + t2.size) / t_apply / 1e9, "nnz": w.nnz,
            "note": "weights kept as their two per-axis factors (xr_outer); the apply forms the 64M products on the fly",
        }
    except Exception as e:  # noqa: BLE001
        out["structured_4000x4000"] = {"error": repr(e)}
    try:  # NetworkGridder weights: 1M network edges over the benchmark's source mesh
        rng = np.random.default_rng(7)
        n_edge = 1_000_000
        xy_lo, xy_hi = float(mesh_xy.min()), float(mesh_xy.max())
        a = rng.uniform(xy_lo, xy_hi, (n_edge, 2))
        ang = rng.uniform(0, 2 * np.pi, n_edge)
        length = rng.exponential(0.002 * (xy_hi - xy_lo), n_edge)
        edges = np.stack([a, a + length[:, None] * np.column_stack([np.cos(ang), np.sin(ang)])], axis=1)
        E.edge_length_csr(mesh, edges)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            w = E.edge_length_csr(mesh, edges)
            E.dev_sync()
            times.append(time.perf_counter() - t0)
        out["network_gridder_1M_edges"] = {
            "weights_ms": 1e3 * min(times), "edges_per_s": n_edge / min(times), "nnz": w.nnz,
            "note": "1M random segments (exponential lengths, mean ~2 cell sizes) over the ~1M-triangle source mesh; "
            "includes the 32 MB upload of the edge coordinates",
        }
    except Exception as e:  # noqa: BLE001
        out["network_gridder_1M_edges"] = {"error": repr(e)}
    return out


def run_multi(args):
    import torch
    import torch.distributed as dist

    from xugrid_amd import meshgen
    from xugrid_amd.distributed import HipBackend, ShardedOverlapRegridder, init_process_group_from_env

    init_process_group_from_env("nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    backend = HipBackend(local_rank)
    # weak scaling: N x 500k points per mesh.  The lattice-split triangulation keeps the set-up of
    # the N-times larger meshes to seconds on every rank (qhull on 4M points takes minutes).
    sxy, sf, txy, tf = make_meshes(args.points * world, delaunay=False)
    S, T = sf.shape[0], tf.shape[0]
    rg = ShardedOverlapRegridder(sxy, sf, txy, tf, backend, partition=args.partition, exchange=args.exchange)
    data = meshgen.smooth_field(sxy[sf].mean(axis=1), 0)
    local = rg.local_source(data)

    def step():
        rg.rebuild()
        return rg.regrid_local(local)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=backend.device)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    nnz = torch.tensor([rg.weights.nnz], dtype=torch.int64, device=backend.device)
    dist.all_reduce(nnz)
    # rank 0's kernels of a few more steps (hipEvents around every launch): roofline of its dominant kernel
    roofline = None
    from xugrid_amd import engine as E

    with E.KernelTimer() as kt:
        for _ in range(3):
            step()
    dist.barrier()
    if rank == 0:
        try:
            kernels = {k: (n, t / n) for k, (n, t) in kt.records.items()}
            per_step = {k: n * avg / 3 for k, (n, avg) in kernels.items()}
            dominant = max(per_step, key=per_step.get)
            s_loc, t_loc = rg.local_faces.size, rg.local_targets.size
            ab = algorithmic_bytes(s_loc, t_loc, sxy.shape[0], txy.shape[0], backend._src_mesh.last_candidates(),
                                   rg.weights.nnz)
            dom_bytes, dom_ms = ab.get(dominant), kernels[dominant][1]
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_bytes else None
            roofline = {
                "bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms, "rank": 0,
                "local_source_faces": int(s_loc), "local_target_faces": int(t_loc),
                "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            }
        except Exception as e:  # noqa: BLE001
            roofline = {"error": repr(e)}
    if rank == 0:
        result = {
            "metric": "target cells regridded/s (OverlapRegridder 1M->1M tri per GPU, weights + mean apply)",
            "value": T / (elapsed / args.steps),
            "unit": "target cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{world} x BASELINE config 2: ~{S} source -> ~{T} target triangles (lattice-split "
                "triangulation), OverlapRegridder mean, K=1",
                "source_faces": S,
                "target_faces": T,
                "nnz": int(nnz.item()),
                "parallelism": f"source faces sharded over {world} GPUs ({args.partition} blocks), target replicated, "
                + ("RCCL sparse all-to-all (reduce-scatter restricted to the touched targets) of per-target partial sums"
                   if args.exchange == "sparse" else "RCCL reduce-scatter of per-target partial sums"),
            },
            "roofline": roofline,
            "cpu_baseline": None,
        }
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=500_000, help="lattice points per mesh and per GPU (faces ~ 2x)")
    ap.add_argument("--no-delaunay", action="store_true", help="lattice-split triangulation instead of qhull")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the informative extras (configs 3 and 5, structured pair)")
    ap.add_argument("--partition", default="balanced", choices=["balanced", "morton", "hash"],
                    help="source shards: Morton blocks of equal estimated work (default) / equal face counts, or id mod N")
    ap.add_argument("--exchange", default="sparse", choices=["sparse", "dense"],
                    help="sparse all-to-all of the touched targets (default) or dense reduce-scatter")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path even with one rank")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.force_dist:
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
