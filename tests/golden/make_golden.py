#!/usr/bin/env python3
"""Generates tests/golden/<case>.npz from the reference's own kernels compiled for the host (oracle/_ref).

Run in the build container (needs /root/reference through oracle/ref/ref_variants.py):
    make -C oracle && python tests/golden/make_golden.py
Each fixture holds, for one parity case of tests/cases.py after its `steps` time steps:
  crc_*      CRC-32 of the raw bytes of flags, rho, u, (F,) and of every domain's DDF buffer (bit-exact pins),
  sum_*      float64 sums of rho and |u| (size-independent sanity),
  flags      the complete flag lattice (uint8),
  rho_sub, u_sub   every 2nd (long cases: 4th) cell per axis of rho and u, for the tolerance-based comparison,
  pins["object"]  (FORCE_FIELD cases) the object_center_of_mass / object_force / object_torque kernel sums per domain
                   and the host totals, groups added in ascending order,
plus the generating configuration.  Only data is stored; no reference source.
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import harness as H  # noqa: E402
from tests.cases import CASES, LONG_CASES  # noqa: E402
from tests.voxel_cases import VOXEL_CASES, voxelize_with_engine  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def summarize(lbm):
    """The pins of one finished run (works for OracleLBM and cfd_amd.LBM alike)."""
    out = {"crc_flags": crc(lbm.flags), "crc_rho": crc(lbm.rho), "crc_u": crc(lbm.u),
           "sum_rho": float(lbm.rho.sum(dtype=np.float64)), "sum_absu": float(np.abs(lbm.u).sum(dtype=np.float64))}
    if lbm.F is not None:
        out["crc_F"] = crc(lbm.F)
    return out


OBJECT_CENTER = (1.5, -2.0, 0.5)


def object_marker(lbm):
    """The flag value of the case's solid body: TYPE_S|TYPE_X where the setup marks one, else TYPE_S."""
    return 65 if (lbm.flags == 65).any() else 1


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]
    for name, case in sorted(list(CASES.items()) + list(LONG_CASES.items())):
        if only and not any(name.startswith(o) for o in only):
            continue
        kw = dict(case["kw"])
        ref = H.OracleLBM(engine_factory=lambda local, D, n=name: H.RefEngine(n), **kw)
        case["setup"](ref)
        ref.run(case["steps"], threads=os.cpu_count() if name in LONG_CASES else 1)  # cells are independent within a step
        ref.read_from_device()
        if ref.F is not None:
            ref.update_force_field()
            ref.read_from_device()
        pins = summarize(ref)
        pins["crc_fi"] = [crc(d.fi) for d in ref.domains]
        if ref.F is not None:  # the object_* reduction kernels, per domain as the kernels define them, and the host totals
            marker = object_marker(ref)
            pins["object"] = {"marker": marker, "center": list(OBJECT_CENTER),
                              "com": [float(v) for v in ref.object_center_of_mass(marker)],
                              "force": [float(v) for v in ref.object_force(marker)],
                              "torque": [float(v) for v in ref.object_torque(OBJECT_CENTER, marker)],
                              "domain_force": [[float(v) for v in d.engine.object_sum(1, d.F, d.flags, marker)[:3]] for d in ref.domains],
                              "domain_torque": [[float(v) for v in d.engine.object_sum(2, d.F, d.flags, marker, OBJECT_CENTER)[:3]] for d in ref.domains],
                              "domain_com": [[float(v) for v in d.engine.object_sum(0, d.F, d.flags, marker)[:3]] for d in ref.domains],
                              "domain_cells": [int(d.engine.object_sum(0, d.F, d.flags, marker)[3:4].view(np.uint32)[0]) for d in ref.domains]}
        meta = {"case": name, "steps": case["steps"], "kw": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                "generator": "oracle/_ref (reference OpenCL kernels, clang -x cl, x86-64), tests/golden/make_golden.py"}
        k = 4 if name in LONG_CASES else 2  # sub-sampling stride of the stored rho/u
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), pins=json.dumps(pins), flags=ref.flags,
                            rho_sub=ref.rho[::k, ::k, ::k], u_sub=ref.u[:, ::k, ::k, ::k])
        print(name, pins["crc_rho"], pins["crc_u"], "%.6f" % pins["sum_rho"])
    # voxelizer fixtures: flags (complete) and the CRC of u from the reference's voxelize_mesh kernel
    for name, case in sorted(VOXEL_CASES.items()):
        if only and not any(("voxel_" + name).startswith(o) for o in only):
            continue
        flags, u = voxelize_with_engine(case, lambda local, D, d, n=name: H.RefEngine("voxel_%s_d%d" % (n, d)))
        np.savez_compressed(os.path.join(out_dir, "voxel_" + name + ".npz"), flags=flags, crc_u=np.uint32(crc(u)),
                            meta=json.dumps({"case": name, "generator": "oracle/_ref voxelize_mesh, tests/golden/make_golden.py"}))
        print("voxel", name, int(flags.sum()), crc(u))


if __name__ == "__main__":
    main()
