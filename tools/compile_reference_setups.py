#!/usr/bin/env python3
"""Checks how much of the reference's setup collection compiles UNCHANGED against cfd_amd/host (syntax check only, CPU).
Every `void main_setup() { ... }` block of FluidX3D-master/src/setup.cpp.bak is extracted into a temporary directory
outside the repository, the extensions named in its first comment become -D switches, and g++ -fsyntax-only is run.
Needs /root/reference; prints one line per setup.  usage: tools/compile_reference_setups.py [-v]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/FluidX3D-master/src/setup.cpp.bak"
HAVE = {"D2Q9", "D3Q15", "D3Q19", "D3Q27", "SRT", "TRT", "FP16S", "FP16C", "UPDATE_FIELDS", "VOLUME_FORCE", "FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES",
        "MOVING_BOUNDARIES", "SUBGRID", "BENCHMARK", "GRAPHICS", "INTERACTIVE_GRAPHICS", "INTERACTIVE_GRAPHICS_ASCII"}


def blocks(text):
    for m in re.finditer(r"void main_setup\(\) \{ // ([^\n]*)\n", text):
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        yield m.group(1), text[m.start():i]


def main():
    verbose = "-v" in sys.argv
    if not os.path.exists(SRC):
        print("reference not present")
        return 0
    text = open(SRC, errors="replace").read()
    tmp = tempfile.mkdtemp(prefix="setups_")
    ok = skipped = failed = 0
    for k, (title, body) in enumerate(blocks(text)):
        m = re.search(r"required extensions in defines\.hpp: (.*)$", title)
        names = re.findall(r"[A-Z][A-Z0-9_]+", m.group(1)) if m else []
        missing = sorted(set(n for n in names if n not in HAVE))
        label = title.split(";")[0][:60]
        if missing:
            skipped += 1
            print("skip  %-62s needs %s" % (label, ", ".join(missing)))
            continue
        defs = ["-D" + n for n in names if n not in ("GRAPHICS", "INTERACTIVE_GRAPHICS", "INTERACTIVE_GRAPHICS_ASCII", "BENCHMARK")]
        if not any(d in defs for d in ("-DD2Q9", "-DD3Q15", "-DD3Q19", "-DD3Q27")):
            defs.append("-DD3Q19")
        if "-DTRT" not in defs:
            defs.append("-DSRT")
        if any(n.startswith("INTERACTIVE_GRAPHICS") or n == "GRAPHICS" for n in names):
            defs.append("-DGRAPHICS")
        path = os.path.join(tmp, "setup_%02d.cpp" % k)
        with open(path, "w") as f:
            f.write('#include "setup.hpp"\n#include "info.hpp"\n' + body + "\n")
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", os.path.join(ROOT, "cfd_amd", "host")] + defs + [path],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode == 0:
            ok += 1
            print("ok    %-62s %s" % (label, " ".join(defs)))
        else:
            failed += 1
            errs = [l for l in r.stdout.splitlines() if "error" in l]
            print("FAIL  %-62s %s" % (label, errs[0].split("error:")[-1].strip()[:110] if errs else "?"))
            if verbose:
                print("\n".join("        " + l for l in errs[:8]))
    print("compiled %d, failed %d, skipped (extensions not provided) %d" % (ok, failed, skipped))
    return 0


if __name__ == "__main__":
    sys.exit(main())
