#!/usr/bin/env python3
"""Headline benchmark: MLUPs/s of the lattice-Boltzmann time step on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (the configuration the headline metric is quoted on): periodic Taylor-Green vortex, D3Q19 SRT, FP32 DDFs, 512^3 cells per GPU,
no UPDATE_FIELDS (153 algorithmic bytes per cell and step).  With N GPUs the lattice is 512 x 512 x (512 N), split
into N z-slabs (weak scaling); the only communication is the per-step DDF halo exchange with the two z-neighbours.
A step = one stream_collide pass over the whole lattice (+ halo exchange).  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PUBLISHED_MLUPS = {1: 30380.0, 8: 188742.0}  # D3Q19 SRT FP32 on MI355X, reference OpenCL code (FluidX3D-master/README.md:1079, 1615)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(threads):
    """The same workload on the host cores: the reference's own kernels compiled for the CPU (oracle/_ref) when that
    library is present, else the scalar C restatement.  Bounded sample: a 256^3 (reference) or 128^3 (port) lattice."""
    from oracle import harness as H
    from cfd_amd import setups
    if H.RefEngine.available("bench_tgv256"):
        n, kind, cores = 256, "reference", threads
        engine = lambda local, D: H.RefEngine("bench_tgv256")
    else:
        n, kind, cores = 128, "port", 1
        engine = None
    lbm = H.OracleLBM((n, n, n), 0.01, engine_factory=engine)
    setups.taylor_green(lbm)
    lbm.initialize()
    t0 = time.time(); lbm.run(1, threads=cores); first = time.time() - t0
    steps = int(max(2, min(400, 15.0 / max(first, 1e-3))))
    t0 = time.time(); lbm.run(steps, threads=cores); dt = time.time() - t0
    return {"value": round(n ** 3 * steps / dt / 1e6, 2), "unit": "MLUPs/s", "cores": cores, "kind": kind,
            "sample": "taylor_green %d^3 D3Q19 SRT FP32, %d steps in %.1f s" % (n, steps, dt)}


def measured_traffic(workload):
    """HBM bytes per stream_collide launch from the committed rocprofv3 PMC passes of this workload (or None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


TUNE_STEPS = 400


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512, help="cells per axis and GPU")
    ap.add_argument("--storage", default="FP32", choices=["FP32", "FP16S", "FP16C"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cfd_amd
    from cfd_amd import setups
    from cfd_amd.distributed import DistributedLBM
    from cfd_amd.lbm import check

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d; launch N>1 through torch.distributed.run" % (a.gpus, world))
    device = local_rank % torch.cuda.device_count()  # ranks share a GPU only in a gloo rehearsal
    torch.cuda.set_device(device)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(a.backend)

    n = a.size
    lbm = DistributedLBM((n, n, n * world), 0.01, Q=19, storage=a.storage, collision="SRT", rank=rank, world=world, overlap=not a.no_overlap)
    setups.taylor_green(lbm)
    lbm.initialize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: the library tries its launch geometries on the first full-lattice launches and keeps the fastest
    tuned = C.c_int(0)
    for _ in range(TUNE_STEPS // 20):
        lbm.run(20)
        check(cfd_amd.lib().lbmhip_get_tuning(lbm.ops.dom.h, b"tuned", C.byref(tuned)))
        if tuned.value:
            break
    lbm.run(a.warmup)
    barrier()
    lib = cfd_amd.lib()
    ms_events = C.c_float()
    t0 = time.perf_counter()
    check(lib.lbmhip_timer_start(lbm.ops.dom.h))          # HIP events on the stream the kernels are launched on
    for _ in range(a.steps):
        lbm.do_time_step()
    check(lib.lbmhip_timer_stop(lbm.ops.dom.h, C.byref(ms_events)))
    barrier()
    seconds = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())

    cells_global = n * n * n * world
    value = cells_global * a.steps / seconds / 1e6
    B = cfd_amd.lib().lbmhip_bandwidth_bytes_per_cell(19, cfd_amd.STORAGE[a.storage], 0)
    if rank == 0:
        workload = "taylor_green_%d^3_per_gpu_d3q19_srt_%s" % (n, a.storage.lower())
        kernel_ms = ms_events.value / a.steps               # one rank's launches per step (N=1: exactly one stream_collide launch)
        achieved = B * n ** 3 / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "MLUPs/s", "value": round(value, 1), "unit": "MLUPs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(seconds / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / PUBLISHED_MLUPS[world], 4) if (world in PUBLISHED_MLUPS and a.storage == "FP32" and n == 512) else None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "lattice": [n, n, n * world], "velocity_set": "D3Q19", "collision": "SRT", "ddf_storage": a.storage,
                       "decomposition": "z-slabs x%d" % world, "bytes_per_cell_step": B, "halo_overlap": bool(lbm.overlap)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(workload), "kernel": "stream_collide", "kernel_ms": round(kernel_ms, 4)},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    lbm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
