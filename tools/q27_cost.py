import sys
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "none"]
src = open("tools/config_bench.py").read().split("SWEEP = ")[0]
ns = {}
exec(compile(src, "cb", "exec"), ns)
ns["SWEEP"] = False; ns["BLOCKS"] = (0,)
for st in ("FP16C", "FP16S", "FP32"):
    for feat, label in ((0, "plain"), (2, "VF"), (16, "FF"), (18, "VF+FF")):
        for mask in (False, True):
            c = dict(N=(384, 384, 384), Q=27, st=st, coll="SRT", feat=feat, f=(1e-5, 0.0, 0.0) if feat & 2 else (0.0, 0.0, 0.0))
            if mask: c["mask"] = True
            ns["run"]("q27 %s %s%s" % (st, label, "+mask" if mask else ""), c)
