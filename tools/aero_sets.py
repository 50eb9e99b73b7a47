import sys
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "none"]
src = open("tools/config_bench.py").read().split("SWEEP = ")[0]
ns = {}
exec(compile(src, "cb", "exec"), ns)
ns["SWEEP"] = False; ns["BLOCKS"] = (0,)
for st in ("FP16S", "FP16C", "FP32"):
    for feat, label in ((40, "EQ+SGS"), (44, "MB+EQ+SGS"), (42, "VF+EQ+SGS (generic kernel)")):
        c = dict(N=(512, 512, 512), Q=19, st=st, coll="SRT", feat=feat, mask=True, f=(1e-5, 0.0, 0.0) if feat & 2 else (0.0, 0.0, 0.0))
        ns["run"]("%s %s" % (st, label), c)
