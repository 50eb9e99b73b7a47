#!/usr/bin/env python3
"""Diagnostic: instruction histogram of one stream_collide instance.
usage: tools/isa_hist.py <unit, e.g. q27_fp16c> <Q> <ST> <COLL> <FEAT> <V> <P>   (V=1: the one-cell-per-lane kernel)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, Q, ST, COLL, FEAT, V, P = sys.argv[1:8]
tmp = tempfile.mkdtemp(prefix="isa_")
src = os.path.join(ROOT, "cfd_amd", "csrc", "collide_%s.hip" % unit)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--save-temps=obj", "-c", src, "-o", os.path.join(tmp, "k.o")],
               check=True, stderr=subprocess.DEVNULL, cwd=tmp)
asm = open(os.path.join(tmp, [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0])).read()
feat = FEAT.replace("-", "n")
name = ("_ZN3lbm9k_collideILi%sELi%sELi%sELi%sELb0EEEvNS_4ArgsE" % (Q, ST, COLL, feat)) if V == "1" else ("_ZN3lbm14k_collide_wideILi%sELi%sELi%sELi%sELi%sELi%sELb0EEEvNS_4ArgsE" % (Q, ST, COLL, feat, V, P))
m = re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(name), asm, re.S | re.M)
if not m:
    raise SystemExit("kernel %s not found" % name)
c = collections.Counter()
for line in m.group(1).split("\n"):
    t = line.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    c[t.split()[0]] += 1
print("%s: %d instructions, %d vector" % (name, sum(c.values()), sum(v for k, v in c.items() if k.startswith("v_"))))
for k, v in c.most_common(60):
    print("%5d %s" % (v, k))
if len(sys.argv) > 8:
    open(sys.argv[8], "w").write(m.group(1))
