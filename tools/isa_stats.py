#!/usr/bin/env python3
"""Diagnostic: compile one collide_*.hip with --save-temps into a temp dir and print, per kernel instance, the number of
vector, scalar, branch and memory instructions and the register use.  usage: tools/isa_stats.py q19_fp16s [filter]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    name = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    tmp = tempfile.mkdtemp(prefix="isa_")
    src = os.path.join(ROOT, "cfd_amd", "csrc", "collide_%s.hip" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--save-temps=obj",
                    "-c", src, "-o", os.path.join(tmp, "k.o")], check=True, stderr=subprocess.DEVNULL, cwd=tmp)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    cur, stats = None, collections.OrderedDict()
    for line in open(os.path.join(tmp, asm)):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            stats[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        t = line.strip()
        if line.startswith("\t") and t and not t.startswith((".", ";")):
            op = t.split()[0]
            c = stats[cur]
            c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "load" if "_load" in op else "store" if "_store" in op else "other"] += 1
            if op.startswith("s_cbranch"):
                c["branch"] += 1
            if op.startswith("v_pk_"):
                c["packed"] += 1
        m = re.match(r"; (NumVgprs|Occupancy|ScratchSize): (\d+)", t)
        if m:
            stats[cur][m.group(1)] = int(m.group(2))
    for k, c in stats.items():
        m = re.match(r"_ZN3lbm9k_collideILi(\d+)ELi(\d+)ELi(\d+)ELi(n?\d+)ELb(\d)", k)
        w = re.match(r"_ZN3lbm14k_collide_wideILi(\d+)ELi(\d+)ELi(\d+)ELi(n?\d+)ELi(\d+)ELi(\d+)E", k)
        if not (m or w) or c["valu"] < 50:
            continue
        tag = "Q%s st%s coll%s feat%s V1 fields%s" % m.groups() if m else "Q%s st%s coll%s feat%s V%s P%s wide" % w.groups()
        if flt in tag:
            print("%-44s valu %4d (packed %4d) salu %4d branch %3d load %3d store %3d vgpr %3d occ %d scratch %d" % (
                tag, c["valu"], c["packed"], c["salu"], c["branch"], c["load"], c["store"], c["NumVgprs"], c["Occupancy"], c["ScratchSize"]))


if __name__ == "__main__":
    main()
