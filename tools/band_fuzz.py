"""Diagnostic: random lattices, feature sets and launch geometries through lbmhip_stream_collide_steps with two to four steps per
launch (fuse = 2, or 1 for the form without bands; one domain, or two or three slabs in one process), compared bit for bit with
one launch per step.  (GPU needed)
usage: python tools/band_fuzz.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfd_amd
from cfd_amd import setups

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SETS = [  # what collide_pairs.hip holds
    dict(Q=19, storage="FP32"), dict(Q=19, storage="FP32", collision="TRT"), dict(Q=27, storage="FP32"), dict(Q=19, storage="FP16S"), dict(Q=19, storage="FP16C"),
    dict(Q=27, storage="FP16S"), dict(Q=27, storage="FP16C"), dict(Q=19, storage="FP16S", features=("EQUILIBRIUM_BOUNDARIES",)), dict(Q=19, storage="FP16S", features=("VOLUME_FORCE",), f=(1e-4, 0.0, 0.0)),
    dict(Q=19, storage="FP32", features=("SUBGRID",)), dict(Q=19, storage="FP32", features=("VOLUME_FORCE",), f=(1e-4, 2e-5, -3e-5)),
    dict(Q=19, storage="FP32", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)), dict(Q=19, storage="FP32", features=("EQUILIBRIUM_BOUNDARIES",)),
    dict(Q=19, storage="FP32", features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")), dict(Q=19, storage="FP32", features=("FORCE_FIELD",)),
    dict(Q=19, storage="FP32", features=("FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES")), dict(Q=19, storage="FP32", features=("VOLUME_FORCE", "EQUILIBRIUM_BOUNDARIES"), f=(1e-4, 0.0, 2e-5)),
    dict(Q=19, storage="FP32", features=("VOLUME_FORCE", "EQUILIBRIUM_BOUNDARIES", "SUBGRID"), f=(1e-4, 0.0, 0.0)), dict(Q=19, storage="FP32", features=("FORCE_FIELD", "SUBGRID")),
    dict(Q=19, storage="FP32", features=("FORCE_FIELD", "VOLUME_FORCE"), f=(1e-4, 0.0, 0.0)), dict(Q=19, storage="FP32", features=("FORCE_FIELD", "VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)),
    dict(Q=19, storage="FP32", features=("FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), dict(Q=19, storage="FP32", features=("FORCE_FIELD", "VOLUME_FORCE", "EQUILIBRIUM_BOUNDARIES"), f=(1e-4, 0.0, 0.0)),
    dict(Q=19, storage="FP32", features=("FORCE_FIELD", "VOLUME_FORCE", "EQUILIBRIUM_BOUNDARIES", "SUBGRID"), f=(1e-4, 0.0, 0.0)),
    dict(Q=19, storage="FP32", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)), dict(Q=19, storage="FP32", collision="TRT", features=("VOLUME_FORCE",), f=(1e-4, 0.0, 0.0)),
    dict(Q=19, storage="FP32", collision="TRT", features=("SUBGRID",)), dict(Q=19, storage="FP32", collision="TRT", features=("EQUILIBRIUM_BOUNDARIES",)),
    dict(Q=19, storage="FP32", collision="TRT", features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")),
    dict(Q=19, storage="FP16S", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)),
    dict(Q=27, storage="FP16C", features=("FORCE_FIELD", "VOLUME_FORCE"), f=(1e-5, 0.0, 0.0)),
]
bad = 0
for case in range(cases):
    kw = dict(SETS[rng.randint(len(SETS))])
    half = kw["storage"] != "FP32"
    Nx = int(rng.choice([rng.randint(2, 40), rng.randint(40, 300), rng.randint(300, 700), 64, 128, 256, 512]))
    if half:
        Nx += Nx % 2
    Dz = int(rng.choice([1, 1, 1, 2, 3]))  # slabs in one process: lbmhip_stream_collide_pair_planes between the exchanges
    N = (max(Nx, 2), int(rng.randint(25, 96)), int(rng.randint(8, 40)) if Dz == 1 else int(rng.randint(10, 30))*Dz)
    if Dz > 1:
        kw["D"] = (1, 1, Dz)
    feats = kw.get("features", ())
    setup = setups.inflow_box if "EQUILIBRIUM_BOUNDARIES" in feats else setups.channel if "VOLUME_FORCE" in feats else setups.taylor_green
    fuse = 1 if rng.rand() < 0.15 else 2
    tune = dict(band_rows=int(rng.choice([0, 2, 3, 5, 8, 64])), band_lag=int(rng.choice([0, 2, 3, 7])), band_steps=int(rng.choice([0, 2, 3, 4])),
                band_blocks_per_cu=int(rng.choice([0, 0, 2, 5])), pair_block=int(rng.choice([0, 0, 64, 128])), pair_cells_together=int(rng.choice([1, 2])) if half and not feats else 1,
                pair_lag=int(rng.choice([0, 3, 5])))
    steps = int(rng.randint(2, 10))
    a = cfd_amd.LBM(N=N, nu=0.02, **kw); b = cfd_amd.LBM(N=N, nu=0.02, **kw)
    try:
        a.set_tuning("fuse", fuse); b.set_tuning("fuse", 0)
        for k, v in tune.items():
            a.set_tuning(k, v)
        setup(a); setup(b)
        if "FORCE_FIELD" in feats:  # a force on every cell (the kernels read it for fluid cells)
            F = np.random.RandomState(case).uniform(-1e-5, 1e-5, a.F.shape).astype(np.float32)
            a.F[...] = F; b.F[...] = F
        a.run(steps); b.run(steps)
        same = all(bool(np.array_equal(da.download_fi().view(np.uint8), db.download_fi().view(np.uint8))) for da, db in zip(a.domains, b.domains))
        form, gave = a.get_tuning("fused"), a.get_tuning("band_given_up")
        if not same or gave:
            bad += 1
        print("%s case %3d: N %s %s fuse %d -> form %d, %d steps, %s, given up %d" % ("ok  " if same and not gave else "BAD ", case, N, kw, fuse, form, steps, tune, gave), flush=True)
    finally:
        a.close(); b.close()
print("%d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
