"""Diagnostic: upload and download rate of a 1.6 GB field from and to pageable and page-locked host memory (GPU needed)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L
lib = cfd_amd.lib()
n = 512
dom = L.Domain(L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, 0, 0, 0, 0.01, 0))
N = dom.N
pageable = np.empty(3 * N, dtype=np.float32)
pinned_t = torch.empty(3 * N, dtype=torch.float32).pin_memory()
pinned = pinned_t.numpy()
for name, a in (("pageable", pageable), ("pinned", pinned), ("pageable", pageable), ("pinned", pinned)):
    for direction in ("download", "upload"):
        ts = []
        for r in range(3):
            t0 = time.perf_counter()
            (dom.download if direction == "download" else dom.upload)(L.FIELD_U, a)
            ts.append(time.perf_counter() - t0)
        print("%-9s %-8s %s GB/s" % (name, direction, " ".join("%.1f" % (a.nbytes / t / 1e9) for t in ts)), flush=True)
dom.close()
