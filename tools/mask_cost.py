import sys
sys.path.insert(0, ".")
_args = list(sys.argv); sys.argv = [sys.argv[0], "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("cb", "tools/config_bench.py")
src = open("tools/config_bench.py").read().split("SWEEP = ")[0]
ns = {}
exec(compile(src, "cb", "exec"), ns)
ns["SWEEP"] = False; ns["BLOCKS"] = (0,)
F = ns["F"]
ST = _args[1] if len(_args) > 1 and _args[1] in ("FP32", "FP16S", "FP16C") else "FP16S"
for feat, label in ((0, "plain"), (1, "UF"), (8, "EQ"), (34, "VF+SGS"), (35, "UF+VF+SGS"), (42, "VF+EQ+SGS"), (43, "GUI")):
    for mask in (False, True):
        c = dict(N=(512, 512, 512), Q=19, st=ST, coll="SRT", feat=feat, f=(1e-5, 0.0, 0.0) if feat & 2 else (0.0, 0.0, 0.0))
        if mask: c["mask"] = True
        ns["run"]("%s%s" % (label, "+mask" if mask else ""), c)
