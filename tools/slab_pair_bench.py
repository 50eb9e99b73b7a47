"""Diagnostic: per-GPU step rate of the slab runner on one GPU, the rank being its own z-neighbour over RCCL, with the steps
taken two at a time (inner planes of both steps in one launch) and one at a time.  usage: [SLAB_ONLY_PAIRED=1] python tools/slab_pair_bench.py [storage]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import cfd_amd
from cfd_amd import setups
from cfd_amd.distributed import DistributedLBM
from cfd_amd.lbm import check

import numpy as np

storage = sys.argv[1] if len(sys.argv) > 1 else "FP32"
lib = cfd_amd.lib()
n = 512
# a vortex field, plane by plane (a lattice at rest holds zeros only, and the two-step launches then run several per cent faster than with real data)
ang = (np.arange(n, dtype=np.float32) + 0.5) * np.float32(2.0 * np.pi / n)
cx, sx, cy, sy = np.cos(ang)[None, :], np.sin(ang)[None, :], np.cos(ang)[:, None], np.sin(ang)[:, None]
base = np.stack([0.05 * cx * sy, -0.05 * sx * cy, 0.02 * sx * sy]).astype(np.float32)


def plane(z):
    c = 2.0 * np.pi * (z + 0.5) / n
    return {"u": base * np.array([np.sin(c), np.sin(c), np.cos(c)], dtype=np.float32)[:, None, None]}


for paired in ((True,) if os.environ.get("SLAB_ONLY_PAIRED") else (True, False, True, False)):
    lbm = DistributedLBM((n, n, 2 * n), 0.01, storage=storage, rank=0, world=2, overlap=True, host_fields=False)
    lbm.next = lbm.prev = 0
    lbm.pair_steps = paired
    if not paired:
        check(lib.lbmhip_set_tuning(lbm.ops.dom.h, b"fuse", 0))
    lbm.fill_planes(plane)
    lbm.initialize()
    tuned = C.c_int(0)
    for _ in range(20):
        lbm.run(20)
        check(lib.lbmhip_get_tuning(lbm.ops.dom.h, b"tuned", C.byref(tuned)))
        fused = C.c_int(0); check(lib.lbmhip_get_tuning(lbm.ops.dom.h, b"fused", C.byref(fused)))
        if tuned.value or fused.value:
            break
    lbm.run(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lbm.run(100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s 512^3 per rank, steps %s: %.3f ms/step %7.0f MLUPs/s per GPU (fused launches: %d)" % (storage, "two at a time" if paired else "one at a time", dt * 10, n ** 3 * 100 / dt / 1e6, fused.value), flush=True)
    lbm.close()
dist.destroy_process_group()
