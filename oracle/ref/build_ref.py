#!/usr/bin/env python3
"""Build recipe for oracle/_ref: the reference's own OpenCL C kernels compiled for the host CPU.

TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is imported, linked or executed by the product
(cfd_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

What this does (needs /root/reference, so it only runs in the build container):
  1. g++ compiles the reference's kernel.cpp where it lies (FluidX3D-master/src/kernel.cpp) and links
     it with dump_cl.cpp (ours), which prints the string returned by the reference's
     get_opencl_c_code() (FluidX3D-master/src/kernel.hpp:7-18).
  2. A prelude with the constants LBM_Domain::device_defines() would emit for one configuration
     (FluidX3D-master/src/lbm.cpp:424-665) plus cl_workgroup_size (FluidX3D-master/src/opencl.hpp:290)
     is generated here from the configuration.
  3. /opt/rocm/lib/llvm/bin/clang -x cl compiles prelude+kernels for x86-64 with the reference's own
     build options (FluidX3D-master/src/opencl.hpp:311).
  4. The object is linked with cl_host.cpp (ours): definitions of the OpenCL C *language* built-ins the
     kernels call (get_global_id, fma, clamp, ... as the OpenCL C standard specifies them; no reference
     header, library or generated code is replaced) and a driver that loops the NDRange.

All intermediate files (the kernel text, the per-configuration .cl, objects) live in a temporary
directory OUTSIDE the repository and are deleted afterwards; the only output is the shared library
oracle/_ref/libref_<tag>.so (git-ignored binary, shipped to the GPU box by gpurun like any built .so).
No reference source text is ever written into the repository tree.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.normpath(os.path.join(HERE, "..", "_ref"))
REF_SRC = "/root/reference/FluidX3D-master/src"
CLANG = "/opt/rocm/lib/llvm/bin/clang"

VELOCITY_SETS = {  # Q: (dimensions, transfers)   FluidX3D-master/src/lbm.cpp:5-21
    9: (2, 3), 15: (3, 5), 19: (3, 5), 27: (3, 9)}
WEIGHTS = {  # FluidX3D-master/src/lbm.cpp:512-529
    9: {"def_w0": "(1.0f/2.25f)", "def_ws": "(1.0f/9.0f)", "def_we": "(1.0f/36.0f)"},
    15: {"def_w0": "(1.0f/4.5f)", "def_ws": "(1.0f/9.0f)", "def_wc": "(1.0f/72.0f)"},
    19: {"def_w0": "(1.0f/3.0f)", "def_ws": "(1.0f/18.0f)", "def_we": "(1.0f/36.0f)"},
    27: {"def_w0": "(1.0f/3.375f)", "def_ws": "(1.0f/13.5f)", "def_we": "(1.0f/54.0f)", "def_wc": "(1.0f/216.0f)"},
}
FEATURES = ("UPDATE_FIELDS", "VOLUME_FORCE", "MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "FORCE_FIELD", "SUBGRID")


def available():
    return os.path.isfile(os.path.join(REF_SRC, "kernel.cpp")) and os.path.isfile(CLANG)


def f32_literal(x):
    """9 significant digits round-trip a float exactly (FluidX3D-master/src/utilities.hpp:2745-2754)."""
    s = "%.9g" % float(np.float32(x))
    return s if any(c in s for c in ".eE") else s + ".0"


def relaxation_rate(nu):
    """def_w = 1/tau, tau = 3*nu+0.5, evaluated in float (FluidX3D-master/src/lbm.hpp:132, lbm.cpp:509-511)."""
    nu = np.float32(nu)
    tau = np.float32(np.float32(3.0) * nu) + np.float32(0.5)
    return np.float32(np.float32(1.0) / np.float32(tau))


def prelude(cfg):
    Nx, Ny, Nz = cfg["N"]
    Dx, Dy, Dz = cfg["D"]
    Ox, Oy, Oz = cfg["O"]
    Q = cfg["Q"]
    dims, transfers = VELOCITY_SETS[Q]
    N = Nx * Ny * Nz
    d = ["#define cl_workgroup_size 64u"]
    d += ["#define def_Nx %du" % Nx, "#define def_Ny %du" % Ny, "#define def_Nz %du" % Nz, "#define def_N %dul" % N]
    d.append("#define uxx %s" % ("uint" if N <= 0xFFFFFFFF else "ulong"))
    d += ["#define def_GNx %du" % ((Nx - 2 * (Dx > 1)) * Dx), "#define def_GNy %du" % ((Ny - 2 * (Dy > 1)) * Dy),
          "#define def_GNz %du" % ((Nz - 2 * (Dz > 1)) * Dz)]
    d += ["#define def_Dx %du" % Dx, "#define def_Dy %du" % Dy, "#define def_Dz %du" % Dz]
    d += ["#define def_Ox %d" % Ox, "#define def_Oy %d" % Oy, "#define def_Oz %d" % Oz]
    d += ["#define def_Ax %du" % (Ny * Nz), "#define def_Ay %du" % (Nz * Nx), "#define def_Az %du" % (Nx * Ny)]
    for name, n_, o_, dd in (("x", Nx, Ox, Dx), ("y", Ny, Oy, Dy), ("z", Nz, Oz, Dz)):
        off = 0.5 * float(n_ + 2 * o_ + dd * (2 * (dd > 1) - n_))
        d.append("#define def_domain_offset_%s %sf" % (name, f32_literal(off)))
    d.append("#define D%dQ%d" % (dims, Q))
    d += ["#define def_velocity_set %du" % Q, "#define def_dimensions %du" % dims, "#define def_transfers %du" % transfers]
    d.append("#define def_c 0.57735027f")
    d.append("#define def_w %sf" % f32_literal(relaxation_rate(cfg["nu"])))
    for k, v in WEIGHTS[Q].items():
        d.append("#define %s %s" % (k, v))
    d.append("#define %s" % cfg["collision"])
    for name, val in (("TYPE_S", 1), ("TYPE_E", 2), ("TYPE_T", 4), ("TYPE_F", 8), ("TYPE_I", 16), ("TYPE_G", 32),
                      ("TYPE_X", 64), ("TYPE_Y", 128), ("TYPE_MS", 3), ("TYPE_BO", 3), ("TYPE_IF", 0x18),
                      ("TYPE_IG", 0x30), ("TYPE_GI", 0x38), ("TYPE_SU", 0x38)):
        d.append("#define %s 0x%02X" % (name, val))
    st = cfg["storage"]
    if st == "FP16S":
        d += ["#define fpxx half", "#define fpxx_copy ushort", "#define load(p,o) vload_half(o,p)*3.0517578E-5f",
              "#define store(p,o,x) vstore_half_rte((x)*32768.0f,o,p)"]
    elif st == "FP16C":
        d += ["#define fpxx ushort", "#define fpxx_copy ushort", "#define load(p,o) half_to_float_custom(p[o])",
              "#define store(p,o,x) p[o]=float_to_half_custom(x)"]
    else:
        d += ["#define fpxx float", "#define fpxx_copy float", "#define load(p,o) p[o]", "#define store(p,o,x) p[o]=x"]
    for f in cfg["features"]:
        assert f in FEATURES, f
        d.append("#define %s" % f)
    return "\n".join(d) + "\n"


def run(cmd, capture=False):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT if not capture else subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + (r.stdout or "") + (r.stderr or "" if capture else ""))
        raise SystemExit(r.returncode)
    return r.stdout


def build_all(cfgs, force=False):
    """Build oracle/_ref/libref_<tag>.so for every configuration; returns the list of paths."""
    os.makedirs(OUT, exist_ok=True)
    todo = [c for c in cfgs if force or not os.path.exists(os.path.join(OUT, "libref_%s.so" % c["tag"]))]
    outs = [os.path.join(OUT, "libref_%s.so" % c["tag"]) for c in cfgs]
    if not todo:
        return outs
    tmp = tempfile.mkdtemp(prefix="lbm_ref_build_")  # outside the repository on purpose
    try:
        obj = os.path.join(tmp, "kernel_host.o")
        exe = os.path.join(tmp, "dump_cl")
        run(["g++", "-std=c++17", "-O0", "-w", "-c", os.path.join(REF_SRC, "kernel.cpp"), "-o", obj])
        run(["g++", "-std=c++17", os.path.join(HERE, "dump_cl.cpp"), obj, "-o", exe, "-pthread"])
        body = run([exe], capture=True)
        for cfg in todo:
            tag = cfg["tag"]
            full = os.path.join(tmp, "ref_%s.cl" % tag)
            with open(full, "w") as f:
                f.write(prelude(cfg) + body)
            clobj = os.path.join(tmp, "ref_%s.o" % tag)
            run([CLANG, "-x", "cl", "-cl-std=CL2.0", "-Xclang", "-finclude-default-header", "-target",
                 "x86_64-unknown-linux-gnu", "-O2", "-fPIC", "-cl-finite-math-only", "-cl-no-signed-zeros",
                 "-cl-mad-enable", "-fvisibility=hidden", "-w", "-c", full, "-o", clobj])
            Nx, Ny, Nz = cfg["N"]
            Dx, Dy, Dz = cfg["D"]
            _, transfers = VELOCITY_SETS[cfg["Q"]]
            defs = ["-DREF_NX=%d" % Nx, "-DREF_NY=%d" % Ny, "-DREF_NZ=%d" % Nz, "-DREF_DX=%d" % Dx,
                    "-DREF_DY=%d" % Dy, "-DREF_DZ=%d" % Dz, "-DREF_Q=%d" % cfg["Q"],
                    "-DREF_TRANSFERS=%d" % transfers, "-DREF_FPXX_BYTES=%d" % (4 if cfg["storage"] == "FP32" else 2)]
            defs += ["-DREF_%s" % f for f in cfg["features"]]
            so = os.path.join(OUT, "libref_%s.so" % tag)
            run([CLANG + "++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-pthread"] + defs +
                [os.path.join(HERE, "cl_host.cpp"), clobj, "-o", so])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return outs


def parse(spec):
    """tag:Nx,Ny,Nz:Dx,Dy,Dz:Ox,Oy,Oz:Q:storage:collision:nu:FEATURE+FEATURE"""
    p = spec.split(":")
    ints = lambda s: tuple(int(v) for v in s.split(","))
    return {"tag": p[0], "N": ints(p[1]), "D": ints(p[2]), "O": ints(p[3]), "Q": int(p[4]), "storage": p[5],
            "collision": p[6], "nu": float(p[7]), "features": [f for f in p[8].split("+") if f] if len(p) > 8 else []}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    for path in build_all([parse(s) for s in a.specs], force=a.force):
        print(path)
