"""Diagnostic: step time of the one-process-per-GPU runner (cfd_amd.distributed, backend nccl = RCCL) on ONE GPU: a single
rank plays slab 0 of 2 and is its own neighbour on both sides, so the whole production path runs (boundary planes, pack,
RCCL send/recv, unpack, overlapped interior).  usage: python tools/self_exchange_bench.py [N] [storage] [overlap 0|1]"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

import cfd_amd
from cfd_amd import setups
from cfd_amd.distributed import DistributedLBM
from cfd_amd.lbm import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
storage = sys.argv[2] if len(sys.argv) > 2 else "FP32"
overlap = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lbm = DistributedLBM((n, n, 2 * n), 0.01, storage=storage, rank=0, world=2, overlap=overlap)
lbm.next = lbm.prev = 0
setups.taylor_green(lbm)
lbm.run(120)
lib = cfd_amd.lib()
for rep in range(3):
    ms = C.c_float()
    torch.cuda.synchronize()
    check(lib.lbmhip_timer_start(lbm.ops.dom.h))
    with lbm.ops.scope():
        for _ in range(100):
            lbm._do_time_step()
    check(lib.lbmhip_timer_stop(lbm.ops.dom.h, C.byref(ms)))
    torch.cuda.synchronize()
    print("%s %d^3 per rank, overlap=%s: %.3f ms/step  %.0f MLUPs/s per GPU" % (storage, n, overlap, ms.value / 100, n ** 3 / (ms.value / 100) / 1e3), flush=True)
lbm.close()
dist.destroy_process_group()
