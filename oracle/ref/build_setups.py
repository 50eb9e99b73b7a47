#!/usr/bin/env python3
"""Test infrastructure: builds a few of the reference's OWN setups, unchanged, against the C++ host of this repository.

The `void main_setup()` blocks named below are cut out of FluidX3D-master/src/setup.cpp.bak into a temporary directory
outside the repository, compiled together with cfd_amd/host/lbm.cpp and main.cpp under the -D switches their header
comment asks for, and linked against cfd_amd/liblbmhip.so.  Only the executables land in oracle/_ref/ (git-ignored):
    oracle/_ref/setup_benchmark, setup_poiseuille, setup_stokes_drag
They travel to the GPU box like the other oracle/_ref files; tests/test_reference_setups_gpu.py runs them there.
Needs /root/reference; without it nothing is built."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
SRC = "/root/reference/FluidX3D-master/src/setup.cpp.bak"
OUT = os.path.join(ROOT, "oracle", "_ref")
HOST = os.path.join(ROOT, "cfd_amd", "host")
# name -> (start of the block's title comment, -D switches)
SETUPS = {
    "benchmark": ("benchmark", ["-DD3Q19", "-DSRT", "-DBENCHMARK"]),
    "poiseuille": ("Poiseuille flow validation", ["-DD3Q19", "-DSRT", "-DVOLUME_FORCE"]),
    "stokes_drag": ("Stokes drag validation", ["-DD3Q19", "-DSRT", "-DFORCE_FIELD", "-DEQUILIBRIUM_BOUNDARIES"]),
}


def block(text, title):
    m = re.search(r"void main_setup\(\) \{ // " + re.escape(title), text)
    depth, i = 1, text.index("\n", m.start())
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.start():i]


def build_all():
    if not os.path.exists(SRC):
        return []
    text = open(SRC, errors="replace").read()
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ref_setups_")
    built = []
    for name, (title, defs) in SETUPS.items():
        path = os.path.join(tmp, name + ".cpp")
        with open(path, "w") as f:
            f.write('#include "setup.hpp"\n#include "info.hpp"\n' + block(text, title) + "\n")
        exe = os.path.join(OUT, "setup_" + name)
        subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-w", "-I", HOST] + defs + [path, os.path.join(HOST, "lbm.cpp"), os.path.join(HOST, "main.cpp"),
                        "-o", exe, "-L" + os.path.join(ROOT, "cfd_amd"), "-llbmhip", "-Wl,-rpath,$ORIGIN/../../cfd_amd"], check=True)
        built.append(exe)
    return built


if __name__ == "__main__":
    for p in build_all():
        print(p)
    sys.exit(0)
