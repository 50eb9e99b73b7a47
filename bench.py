#!/usr/bin/env python3
"""Headline benchmark: MLUPs/s of the lattice-Boltzmann time step on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config tgv|channel|q27_force] [--scaling weak|strong]

N > 1 runs one rank per GPU over RCCL.  Started under torch.distributed.run (WORLD_SIZE set) it is one of those ranks;
started plainly it launches the N ranks itself (before anything touches the GPU), forwards rank 0's JSON line and exits
with their status.

Configurations (a step = one stream_collide pass over the whole lattice + the DDF halo exchange between z-slabs; on one
GPU the K timed steps go to the library in one lbmhip_stream_collide_steps call, which runs a single periodic domain of the
headline kind two steps per launch - the same bits as K single launches):
  tgv        periodic Taylor-Green vortex, D3Q19 SRT, 512^3 cells per GPU, weak scaling - the configuration the headline
             metric is quoted on (153 algorithmic bytes per cell and step with FP32 DDFs)
  channel    force-driven channel with extruded pillars (seed 42), walls in y and z, D3Q19 TRT + SUBGRID + VOLUME_FORCE,
             512^3 cells per GPU, weak scaling
  q27_force  the same geometry, D3Q27 FP16C + FORCE_FIELD + VOLUME_FORCE, strong scaling: a fixed global lattice
             (2048^3, or the largest 2048 x 2048 x Nz that fits the GPUs in use) split into N z-slabs
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line; for the default configuration it also carries
the same lattice measured with FP16S DDFs ("fp16s": the metric is quoted for both), and for N > 1 where every rank sat ("config.ranks").
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PUBLISHED_MLUPS = {1: 30380.0, 8: 188742.0}  # D3Q19 SRT FP32 on MI355X, reference OpenCL code (FluidX3D-master/README.md:1079, 1615)
HBM_PEAK_GBS = 8000.0
TUNE_STEPS = 400
SHORT_REGION_S = 0.5

CONFIGS = {
    "tgv": dict(Q=19, collision="SRT", features=(), storage="FP32", scaling="weak", f=(0.0, 0.0, 0.0), label="taylor_green"),
    "channel": dict(Q=19, collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), storage="FP32", scaling="weak", f=(2e-4, 0.0, 0.0), label="channel_pillars"),
    "q27_force": dict(Q=27, collision="SRT", features=("FORCE_FIELD", "VOLUME_FORCE"), storage="FP16C", scaling="strong", f=(1e-5, 0.0, 0.0),
                      label="channel_pillars", lattice=(2048, 2048, 2048)),
}


def cpu_baseline(threads):
    """The headline workload on the host cores: the reference's own kernels compiled for the CPU (oracle/_ref) when that
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
    steps = int(max(2, min(400, 12.0 / max(first, 1e-3))))
    t0 = time.time(); lbm.run(steps, threads=cores); dt = time.time() - t0
    return {"value": round(n ** 3 * steps / dt / 1e6, 2), "unit": "MLUPs/s", "cores": cores, "kind": kind,
            "sample": "taylor_green %d^3 D3Q19 SRT FP32, %d steps in %.1f s" % (n, steps, dt)}


def cpu_baseline_cavity(threads):
    """Second reported baseline: the lid-driven cavity the reference runs on a CPU device (128^3, D3Q19 SRT FP32,
    MOVING_BOUNDARIES, setup.cpp.bak:236-249), a bounded number of its 1000 steps."""
    from oracle import harness as H
    from cfd_amd import setups
    tag = "long_cavity128_fp32"
    if not H.RefEngine.available(tag):
        return None
    lbm = H.OracleLBM((128, 128, 128), 0.0126, features=("MOVING_BOUNDARIES",), engine_factory=lambda local, D: H.RefEngine(tag))
    setups.lid_cavity(lbm)
    lbm.initialize()
    t0 = time.time(); lbm.run(5, threads=threads); first = (time.time() - t0) / 5
    steps = int(max(10, min(1000, 10.0 / max(first, 1e-4))))
    t0 = time.time(); lbm.run(steps, threads=threads); dt = time.time() - t0
    return {"value": round(128 ** 3 * steps / dt / 1e6, 2), "unit": "MLUPs/s", "cores": threads, "kind": "reference",
            "sample": "lid_cavity 128^3 D3Q19 SRT FP32 MOVING_BOUNDARIES, %d of 1000 steps in %.1f s" % (steps, dt)}


def measured_traffic(workload):
    """HBM bytes per stream_collide launch from the committed rocprofv3 PMC passes of this workload, with the profile they
    come from (None, None when there is no such profile): the counters are not read in this run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    entry = table.get(workload)
    if isinstance(entry, dict):
        return entry.get("bytes_per_launch"), entry.get("source")
    return entry, ("profiles/traffic.json" if entry is not None else None)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(n):
    """Start the N ranks as children of this process, which has not touched the GPU, and exit with their status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def choose_lattice(a, cfg, world, lib, device):
    """Global lattice and whether it had to be reduced to fit.  Weak: `size`^3 cells per GPU stacked along z.  Strong: the
    configured global lattice, its z extent cut down to what the GPUs in use can hold."""
    if a.lattice:
        return tuple(int(v) for v in a.lattice.split(",")), False
    if a.scaling == "weak":
        return (a.size, a.size, a.size * world), False
    Nx, Ny, Nz = cfg.get("lattice") or (a.size, a.size, a.size)
    free, total = C.c_uint64(), C.c_uint64()
    from cfd_amd.lbm import check, STORAGE, feature_mask
    check(lib.lbmhip_device_memory(device, C.byref(free), C.byref(total)))
    per_cell = lib.lbmhip_bytes_per_cell_device(cfg["Q"], STORAGE[a.storage], feature_mask(cfg["features"]))
    planes = int(0.92 * free.value / (per_cell * Nx * Ny)) - 2   # per GPU, besides the two halo planes
    fit = max(64, (planes // 64) * 64) * world
    return (Nx, Ny, min(Nz, fit)), fit < Nz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="tgv", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="default: the configuration's own (tgv, channel: weak; q27_force: strong)")
    ap.add_argument("--size", type=int, default=512, help="weak scaling: cells per axis and GPU")
    ap.add_argument("--lattice", default=None, help="global lattice Nx,Ny,Nz (overrides --size and the configuration's own)")
    ap.add_argument("--storage", default=None, choices=["FP32", "FP16S", "FP16C"], help="DDF storage (default: the configuration's own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-fp16s", action="store_true", help="skip the second measurement (the same lattice with FP16S DDFs) of the default configuration")
    ap.add_argument("--no-fuse", action="store_true", help="one launch per step also where the library would run two steps per launch")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    import torch
    import torch.distributed as dist
    import cfd_amd
    from cfd_amd import setups
    from cfd_amd.distributed import DistributedLBM
    from cfd_amd.lbm import check, STORAGE, feature_mask

    cfg = CONFIGS[a.config]
    a.storage = a.storage or cfg["storage"]
    a.scaling = a.scaling or cfg["scaling"]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (a.gpus, world))
    device = local_rank % torch.cuda.device_count()  # ranks share a GPU only in a gloo rehearsal
    torch.cuda.set_device(device)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(a.backend)
    lib = cfd_amd.lib()

    N, reduced = choose_lattice(a, cfg, world, lib, device)
    if world > 1:  # every rank must run the same lattice: rank 0 decides
        box = [N, reduced]
        dist.broadcast_object_list(box, src=0)
        N, reduced = tuple(box[0]), box[1]
    lean = a.config != "tgv"  # slabs of up to 10^9 cells: no host copy of the fields, flags go up plane by plane

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(storage):
        """The K timed steps of this configuration with `storage` DDFs: (lattice, seconds of the slowest rank, seconds per rank,
        milliseconds between HIP events on the launch stream, launch geometry, overlap flag)."""
        lbm = DistributedLBM(N, 0.01, Q=cfg["Q"], storage=storage, collision=cfg["collision"], features=cfg["features"], f=cfg["f"],
                             rank=rank, world=world, overlap=not a.no_overlap, host_fields=not lean)
        n = lbm.N
        if lean:
            mask = setups.obstacle_mask(n[0], n[1], seed=42)
            lbm.fill_planes(lambda z: {"flags": setups.channel_flag_plane(n[0], n[1], n[2], z, 42, mask)})
        else:
            setups.taylor_green(lbm)
        lbm.initialize()
        h = lbm.ops.dom.h
        if a.no_fuse:
            check(lib.lbmhip_set_tuning(h, b"fuse", 0))
            lbm.pair_steps = False
        # untimed: the library tries its launch geometries on the first full-lattice launches and keeps the fastest
        # (a lattice whose steps are paired needs no search: the odd step of a call at most runs the single-step kernel)
        tuned, fused = C.c_int(0), C.c_int(0)
        for _ in range(TUNE_STEPS // 20):
            lbm.run(20)
            check(lib.lbmhip_get_tuning(h, b"tuned", C.byref(tuned)))
            check(lib.lbmhip_get_tuning(h, b"fused", C.byref(fused)))
            if tuned.value or fused.value:
                break
        lbm.run(a.warmup)
        barrier()
        ms_events = C.c_float()
        t0 = time.perf_counter()
        check(lib.lbmhip_timer_start(h))          # HIP events on the stream the kernels are launched on
        lbm.run(a.steps)   # one rank: one lbmhip_stream_collide_steps call; slabs: the steps two at a time around the halo exchanges (DistributedLBM.run)
        check(lib.lbmhip_timer_stop(h, C.byref(ms_events)))
        barrier()
        own_seconds = time.perf_counter() - t0
        seconds, per_rank = own_seconds, [own_seconds]
        if world > 1:
            t = torch.tensor([own_seconds], dtype=torch.float64, device="cuda")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank = [float(v.item()) for v in every]
            seconds = max(per_rank)
        geometry = {}
        check(lib.lbmhip_get_tuning(h, b"fused", C.byref(fused)))  # of the timed call
        keys = {0: ("tuned", "cells_per_thread", "cells_together", "blocks_per_cu", "block", "jit"), 1: ("pair_lag", "pair_blocks_per_cu"),
                2: ("band_rows", "band_lag", "band_blocks_per_cu", "band_given_up")}[fused.value]
        for key in keys:
            v = C.c_int(); check(lib.lbmhip_get_tuning(h, key.encode(), C.byref(v))); geometry[key] = v.value
        geometry["steps_per_launch"] = 2 if fused.value else 1
        geometry["kernel"] = {0: "stream_collide", 1: "stream_collide, two steps per launch (k_collide_pair)",
                              2: "stream_collide, two steps per launch (k_collide_band + k_collide_seam)"}[fused.value]
        overlap = bool(lbm.overlap)
        lbm.close()
        return n, seconds, per_rank, ms_events.value, geometry, overlap

    def roofline(storage, n, ms_events, geometry, workload):
        """Algorithmic bytes of a launch (the reference's bytes per cell and step, lbm.cpp:51-82, x cells x steps per launch) over the time of a launch between HIP events."""
        B = lib.lbmhip_bandwidth_bytes_per_cell(cfg["Q"], STORAGE[storage], feature_mask(cfg["features"]))
        cells_rank = n[0] * n[1] * (n[2] // world)
        spl = geometry["steps_per_launch"]
        kernel_ms = ms_events / a.steps * spl   # per launch (two steps per launch: the launch that pairs them and, band by band, the short one that finishes its block edges)
        achieved = B * cells_rank * spl / (kernel_ms * 1e-3) / 1e9
        suffix = {1: "", 2: "_2steps"}[spl] + ("_band" if "band_rows" in geometry else "")
        traffic, traffic_source = measured_traffic(workload + suffix) if world == 1 else (None, None)
        r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
             "traffic": traffic, "traffic_source": traffic_source, "kernel": geometry["kernel"],
             "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": B * cells_rank * spl}
        if spl > 1:
            r["note"] = ("achieved = algorithmic bytes of both steps / launch time: an effective rate against the HBM peak, not HBM traffic - "
                         "the second step reads what the first wrote from the XCD's L2 or the Infinity Cache")
        return B, r

    def workload_name(storage, n):
        if a.config == "tgv" and n[0] == n[1] == n[2] // world:
            return "taylor_green_%d^3_per_gpu_d3q19_srt_%s" % (n[0], storage.lower())
        return "%s_%s_per_gpu_d3q%d_%s_%s%s" % (cfg["label"], "%dx%dx%d" % (n[0], n[1], n[2] // world), cfg["Q"], cfg["collision"].lower(), storage.lower(),
                                                 "".join("_" + f.lower() for f in cfg["features"]))

    N, seconds, per_rank, ms_events, geometry, overlap = measure(a.storage)
    # the metric is quoted for FP32 and FP16S DDFs: the default run measures the same lattice again with FP16S storage
    second = measure("FP16S") if (a.config == "tgv" and a.storage == "FP32" and not a.no_fp16s) else None
    ranks = None
    if world > 1:  # where every rank sat
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": local_rank, "device": device, "name": props.name, "host": socket.gethostname(), "pid": os.getpid(),
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
                "uuid": str(getattr(props, "uuid", "")), "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    cells_global = N[0] * N[1] * N[2]
    cells_rank = N[0] * N[1] * (N[2] // world)
    value = cells_global * a.steps / seconds / 1e6
    if rank == 0:
        workload = workload_name(a.storage, N)
        B, roof = roofline(a.storage, N, ms_events, geometry, workload)
        headline = a.config == "tgv" and a.storage == "FP32" and N[0] == N[1] == 512 and N[2] == 512 * world
        out = {
            "metric": "MLUPs/s", "value": round(value, 1), "unit": "MLUPs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(seconds / a.steps * 1e3, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": round(value / PUBLISHED_MLUPS[world], 4) if (headline and world in PUBLISHED_MLUPS) else None,
            "dtype": "f32" if a.storage == "FP32" else "f32 arithmetic, %s DDF storage" % a.storage.lower(), "data": "synthetic",
            "config": {"workload": workload, "lattice": list(N), "velocity_set": "D3Q%d" % cfg["Q"], "collision": cfg["collision"], "extensions": list(cfg["features"]),
                       "ddf_storage": a.storage, "decomposition": "z-slabs x%d" % world, "bytes_per_cell_step": B, "halo_overlap": overlap,
                       "backend": ("rccl" if a.backend == "nccl" else a.backend) if world > 1 else None, "lattice_reduced_to_fit": bool(reduced),
                       "launch_geometry": geometry, "tuning_steps_untimed": "up to %d, until the geometry search finished" % TUNE_STEPS, "ranks": ranks},
            "timed_region_s": round(seconds, 4),
            "per_rank_mlups": [round(cells_rank * a.steps / s / 1e6, 1) for s in per_rank],
            "roofline": roof,
        }
        if second:
            n2, seconds2, per_rank2, ms2, geometry2, _ = second
            B2, roof2 = roofline("FP16S", n2, ms2, geometry2, workload_name("FP16S", n2))
            out["fp16s"] = {"value": round(cells_global * a.steps / seconds2 / 1e6, 1), "unit": "MLUPs/s", "ms_per_step": round(seconds2 / a.steps * 1e3, 4),
                            "dtype": "f32 arithmetic, fp16s DDF storage", "bytes_per_cell_step": B2, "timed_region_s": round(seconds2, 4),
                            "per_rank_mlups": [round(cells_rank * a.steps / s / 1e6, 1) for s in per_rank2], "roofline": roof2, "launch_geometry": geometry2}
        if seconds < SHORT_REGION_S:
            out["short_timed_region"] = True
            print("bench.py: the timed region lasted %.3f s; --steps %d or more gives at least %.1f s" % (
                seconds, int(a.steps * SHORT_REGION_S / seconds) + 1, SHORT_REGION_S), file=sys.stderr)
        if world == 1 and not a.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(threads)
            cavity = cpu_baseline_cavity(threads)
            if cavity:
                out["cpu_baseline_cavity128"] = cavity
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
