"""Diagnostic: single-GPU stream_collide rate of the five benchmark configurations (GPU needed).  The lattices start at
rest (the rate does not depend on the field data); solids are an extruded pillar mask uploaded plane by plane, so no
host copy of the big lattices is ever made.
usage: python tools/config_bench.py [--sweep] [--geometry=V,P,blocks_per_cu,block] [--steps=n] [--fuse=-1|1|2] [--band=rows,lag,blocks_per_cu[,lanes per workgroup]] [--together=1|2] [--moving] [names... | fp32_features_<mask>]
(--sweep also tries every launch geometry; --geometry fixes it and skips the library's search, for profiler passes that must see one kernel;
--fuse hands the timed steps to lbmhip_stream_collide_steps in one call with that "fuse" setting: two steps per launch where the library has an
instance, --band fixes k_collide_band's geometry; --moving starts from a vortex field instead of a lattice at rest)"""
import ctypes as C
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cfd_amd
from cfd_amd import lbm as L
from cfd_amd import setups

F = L.FEATURES
CONFIGS = {
    "cavity128_fp32_mb": dict(N=(128, 128, 128), Q=19, st="FP32", coll="SRT", feat=F["MOVING_BOUNDARIES"], walls=True),
    "tgv512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "tgv512_fp16s": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=0),
    "tgv512_fp16c": dict(N=(512, 512, 512), Q=19, st="FP16C", coll="SRT", feat=0),
    "q27_384_fp16c": dict(N=(384, 384, 384), Q=27, st="FP16C", coll="SRT", feat=0),
    "aero512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["EQUILIBRIUM_BOUNDARIES"] | F["SUBGRID"], mask=True),
    "aero512_fp16s": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=F["EQUILIBRIUM_BOUNDARIES"] | F["SUBGRID"], mask=True),
    "vf512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["VOLUME_FORCE"], mask=True, f=(2e-4, 0.0, 0.0)),
    "vf_sgs512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["VOLUME_FORCE"] | F["SUBGRID"], mask=True, f=(2e-4, 0.0, 0.0)),
    "eq512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["EQUILIBRIUM_BOUNDARIES"], mask=True),
    "mb_eq_sgs512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["MOVING_BOUNDARIES"] | F["EQUILIBRIUM_BOUNDARIES"] | F["SUBGRID"], mask=True),
    "mb512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["MOVING_BOUNDARIES"], mask=True),
    "ff_eq512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["FORCE_FIELD"] | F["EQUILIBRIUM_BOUNDARIES"], mask=True),
    "ff512_fp32": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["FORCE_FIELD"], mask=True),
    "q27_384_fp32": dict(N=(384, 384, 384), Q=27, st="FP32", coll="SRT", feat=0),
    "q27_384_fp16s": dict(N=(384, 384, 384), Q=27, st="FP16S", coll="SRT", feat=0),
    "q27_512_fp32": dict(N=(512, 512, 512), Q=27, st="FP32", coll="SRT", feat=0),
    "q27_512_fp16s": dict(N=(512, 512, 512), Q=27, st="FP16S", coll="SRT", feat=0),
    "tgv512_fp32_trt": dict(N=(512, 512, 512), Q=19, st="FP32", coll="TRT", feat=0),
    "tgv512_fp32_sgs": dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=F["SUBGRID"]),
    "gui512_fp16s": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=F["UPDATE_FIELDS"] | F["VOLUME_FORCE"] | F["EQUILIBRIUM_BOUNDARIES"] | F["SUBGRID"], mask=True, f=(0.0, 0.0, 1e-5)),
    "chan_srt_fp16s": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=F["VOLUME_FORCE"] | F["SUBGRID"], mask=True, f=(2e-4, 0.0, 0.0)),
    "chan_srt_fp16s_nomask": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=F["VOLUME_FORCE"] | F["SUBGRID"], f=(2e-4, 0.0, 0.0)),
    "gui512_fp16s_nomask": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=F["UPDATE_FIELDS"] | F["VOLUME_FORCE"] | F["EQUILIBRIUM_BOUNDARIES"] | F["SUBGRID"], f=(0.0, 0.0, 1e-5)),
    "fp16s_1600x512x512": dict(N=(1600, 512, 512), Q=19, st="FP16S", coll="SRT", feat=0),
    "fp16s_1024": dict(N=(1024, 1024, 1024), Q=19, st="FP16S", coll="SRT", feat=0),
    "fp16s_768": dict(N=(768, 768, 768), Q=19, st="FP16S", coll="SRT", feat=0),
    "fp16s_512x1600x512": dict(N=(512, 1600, 512), Q=19, st="FP16S", coll="SRT", feat=0),
    "fp32_1024x512x512": dict(N=(1024, 512, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_512x1024x512": dict(N=(512, 1024, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_512x512x1024": dict(N=(512, 512, 1024), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_768x512x512": dict(N=(768, 512, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_576x512x512": dict(N=(576, 512, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_512x520x512": dict(N=(512, 520, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_512x512x520": dict(N=(512, 512, 520), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_1024": dict(N=(1024, 1024, 1024), Q=19, st="FP32", coll="SRT", feat=0),
    "fp32_1600x512x512": dict(N=(1600, 512, 512), Q=19, st="FP32", coll="SRT", feat=0),
    "capacity1600_fp16s": dict(N=(1600, 1600, 1600), Q=19, st="FP16S", coll="SRT", feat=0),
    "channel512_trt_sgs": dict(N=(512, 512, 512), Q=19, st="FP32", coll="TRT", feat=F["VOLUME_FORCE"] | F["SUBGRID"], mask=True, f=(2e-4, 0.0, 0.0)),
    "channel512_trt_sgs_fp16s": dict(N=(512, 512, 512), Q=19, st="FP16S", coll="TRT", feat=F["VOLUME_FORCE"] | F["SUBGRID"], mask=True, f=(2e-4, 0.0, 0.0)),
    "q27_fp16c_force_slab": dict(N=(2048, 2048, 256), Q=27, st="FP16C", coll="SRT", feat=F["FORCE_FIELD"] | F["VOLUME_FORCE"], mask=True, f=(1e-5, 0.0, 0.0)),
}


def run(name, c):
    lib = cfd_amd.lib()
    Nx, Ny, Nz = c["N"]
    cfg = L.Config(Nx, Ny, Nz, 1, 1, 1, 0, 0, 0, c["Q"], L.STORAGE[c["st"]], L.COLLISION[c["coll"]], c["feat"], 0.01, 0)
    dom = L.Domain(cfg)
    try:
        if c.get("mask") or c.get("walls"):
            plane = np.zeros((Ny, Nx), dtype=np.uint8)
            if c.get("mask"):
                plane[...] = setups.obstacle_mask(Nx, Ny, seed=42)
                plane[0] = plane[-1] = 1
            else:
                plane[0] = plane[-1] = 1; plane[:, 0] = plane[:, -1] = 1
            for z in range(Nz):
                p = np.ones_like(plane) if (c.get("walls") and z in (0, Nz - 1)) else plane
                dom.upload(L.FIELD_FLAGS, np.ascontiguousarray(p), offset=z * Nx * Ny)
            if c.get("walls"):  # moving lid
                lid = np.full(Nx * Ny, 0.1, dtype=np.float32)
                dom.upload(L.FIELD_U, lid, offset=(dom.N + (Nz - 1) * Nx * Ny) * 4)
        if MOVING:  # a velocity field that varies in all three directions, uploaded plane by plane (a lattice at rest holds zeros only, and
            # the two-step launches then run several per cent faster than with real data)
            xs = (np.arange(Nx, dtype=np.float32) + 0.5) * np.float32(2.0 * np.pi / Nx)
            ys = (np.arange(Ny, dtype=np.float32) + 0.5) * np.float32(2.0 * np.pi / Ny)
            cs, sn = np.cos(xs)[None, :], np.sin(xs)[None, :]
            cy, sy = np.cos(ys)[:, None], np.sin(ys)[:, None]
            base = [np.ascontiguousarray(0.05 * cs * sy, dtype=np.float32), np.ascontiguousarray(-0.05 * sn * cy, dtype=np.float32), np.ascontiguousarray(0.02 * sn * sy, dtype=np.float32)]
            for z in range(Nz):
                cz = 2.0 * np.pi * (z + 0.5) / Nz
                for comp, scale in enumerate((np.sin(cz), np.sin(cz), np.cos(cz))):
                    dom.upload(L.FIELD_U, base[comp] * np.float32(scale), offset=(comp * dom.N + z * Nx * Ny) * 4)
        L.check(lib.lbmhip_initialize(dom.h))
        f = c.get("f", (0.0, 0.0, 0.0))
        step = lambda t: L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2])))
        t = [0]
        B = lib.lbmhip_bandwidth_bytes_per_cell(c["Q"], L.STORAGE[c["st"]], c["feat"])

        def measure(label, warm=6):
            for _ in range(warm):
                step(t[0]); t[0] += 1
            n = STEPS or max(10, min(400, int(4e9 / dom.N)))
            ms = C.c_float()
            L.check(lib.lbmhip_timer_start(dom.h))
            for _ in range(n):
                step(t[0]); t[0] += 1
            L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
            per = ms.value / n
            return per, "%-26s %5dx%4dx%4d  %-22s %.3f ms/step  %7.0f MLUPs/s  %4d B/cell  %5.0f GB/s (%.1f%% of 8 TB/s)" % (
                name, Nx, Ny, Nz, label, per, dom.N / per / 1e3, B, B * dom.N / per / 1e6, B * dom.N / per / 1e6 / 80)

        if FUSE is not None:
            L.check(lib.lbmhip_set_tuning(dom.h, b"fuse", FUSE))
            L.check(lib.lbmhip_set_tuning(dom.h, b"pair_cells_together", TOGETHER))
            for key, val in zip(("band_rows", "band_lag", "band_blocks_per_cu", "pair_block"), BAND or ()):
                L.check(lib.lbmhip_set_tuning(dom.h, key.encode(), val))

            def measure(label, warm=6):  # noqa: F811 - the steps of a measurement in one call
                call = lambda n: L.check(lib.lbmhip_stream_collide_steps(dom.h, C.c_uint64(t[0]), n, C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2])))
                call(warm); t[0] += warm
                n = STEPS or max(10, min(400, int(4e9 / dom.N)))
                n += n % 2
                ms = C.c_float()
                L.check(lib.lbmhip_timer_start(dom.h))
                call(n); t[0] += n
                L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
                per = ms.value / n
                v = C.c_int(); L.check(lib.lbmhip_get_tuning(dom.h, b"fused", C.byref(v)))
                return per, "%-26s %5dx%4dx%4d  %-22s %.3f ms/step  %7.0f MLUPs/s  %4d B/cell  %5.0f GB/s (%.1f%% of 8 TB/s)" % (
                    name, Nx, Ny, Nz, "fuse %d -> form %d" % (FUSE, v.value), per, dom.N / per / 1e3, B, B * dom.N / per / 1e6, B * dom.N / per / 1e6 / 80)
        if GEOMETRY:
            for key, val in zip(("cells_per_thread", "cells_together", "blocks_per_cu", "block"), GEOMETRY):
                L.check(lib.lbmhip_set_tuning(dom.h, key.encode(), val))
        if not GEOMETRY:  # let the library's geometry search finish
            tuned = C.c_int(0)
            for _ in range(400):
                step(t[0]); t[0] += 1
                L.check(lib.lbmhip_get_tuning(dom.h, b"tuned", C.byref(tuned)))
                if tuned.value:
                    break
        per, line = measure("fixed" if GEOMETRY else "default", warm=6)
        got = {}
        for key in ("tuned", "cells_per_thread", "cells_together", "blocks_per_cu", "block"):
            v = C.c_int(); L.check(lib.lbmhip_get_tuning(dom.h, key.encode(), C.byref(v))); got[key] = v.value
        print(line + "  [tuned=%(tuned)d V=%(cells_per_thread)d P=%(cells_together)d blocks/CU<=%(blocks_per_cu)d block=%(block)d]" % got, flush=True)
        if SWEEP:
            res = []
            for b in BLOCKS:
                for v, p in ((1, 1), (2, 1), (4, 1), (2, 2), (4, 2)):
                    for k in (0, 2, 3, 4, 5, 6, 7, 8):
                        if p == 2 and k > 4:
                            continue
                        for key, val in (("cells_per_thread", v), ("cells_together", p), ("blocks_per_cu", k), ("block", b)):
                            L.check(lib.lbmhip_set_tuning(dom.h, key.encode(), val))
                        res.append(measure("V=%d P=%d blocks/CU<=%d block=%d" % (v, p, k, b or 256)))
            for per, line in sorted(res)[:8]:
                print("   " + line, flush=True)
    finally:
        dom.close()


SWEEP = "--sweep" in sys.argv or "--sweep-blocks" in sys.argv
BLOCKS = (0, 192, 128, 64) if "--sweep-blocks" in sys.argv else (0,)
GEOMETRY = next((tuple(int(v) for v in a.split("=")[1].split(",")) for a in sys.argv[1:] if a.startswith("--geometry=")), None)
STEPS = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--steps=")), 0)
FUSE = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--fuse=")), None)
MOVING = "--moving" in sys.argv
TOGETHER = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--together=")), 1)
BAND = next((tuple(int(v) for v in a.split("=")[1].split(",")) for a in sys.argv[1:] if a.startswith("--band=")), None)
names = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in (names or list(CONFIGS)):
    if name not in CONFIGS and name.startswith("fp32_trt_features_"):
        m = int(name.rsplit("_", 1)[1])
        CONFIGS[name] = dict(N=(512, 512, 512), Q=19, st="FP32", coll="TRT", feat=m, mask=True, f=(1e-4, 0.0, 0.0) if m & F["VOLUME_FORCE"] else (0.0, 0.0, 0.0))
    if name not in CONFIGS and name.startswith("fp16s_features_"):
        m = int(name.rsplit("_", 1)[1])
        CONFIGS[name] = dict(N=(512, 512, 512), Q=19, st="FP16S", coll="SRT", feat=m, mask=True, f=(1e-4, 0.0, 0.0) if m & F["VOLUME_FORCE"] else (0.0, 0.0, 0.0))
    if name not in CONFIGS and name.startswith("fp32_features_"):  # any feature mask: fp32_features_<decimal mask>, D3Q19 SRT FP32 512^3 with the pillar mask
        m = int(name.rsplit("_", 1)[1])
        CONFIGS[name] = dict(N=(512, 512, 512), Q=19, st="FP32", coll="SRT", feat=m, mask=True, f=(1e-4, 0.0, 0.0) if m & F["VOLUME_FORCE"] else (0.0, 0.0, 0.0))
    run(name, CONFIGS[name])
