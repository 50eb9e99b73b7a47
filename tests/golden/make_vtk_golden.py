#!/usr/bin/env python3
"""Expected export files of cfd_amd/host/examples/setup_vtk_fixture.cpp, worked out from the reference's file format alone
(no code of this repository's writers is involved): tests/golden/vtk_fixture/{rho,u,flags,F}-000000000.vtk and
status-000000000.txt.  Lines of the status file that depend on the clock hold a regular expression between <<< and >>>.

Format sources: FluidX3D-master/src/lbm.hpp:298-340 (header lines, big-endian payload, components interleaved, SI scaling),
lbm.cpp:104-118 (name-%09t.vtk), lbm.cpp:1599-1643 (status keys), units.hpp:9-13, 58-71 (unit factors),
utilities.hpp:2599-2630, 2669-2676, 2745-2754 (how the reference prints a float: d.dddddddd[E±x])."""
import os
import struct

import numpy as np

f32 = np.float32


def ref_float_string(x):
    """to_string(float) of the reference (utilities.hpp:2745-2754 with split_float :2599-2630), in float32 arithmetic."""
    x = f32(x)
    s = ""
    if x < 0:
        s, x = "-", f32(-x)
    exponent = 0
    if x >= f32(10.0):
        for lim, mul, e in ((1e32, 1e-32, 32), (1e16, 1e-16, 16), (1e8, 1e-8, 8), (1e4, 1e-4, 4), (1e2, 1e-2, 2), (1e1, 1e-1, 1)):
            if x >= f32(lim):
                x = f32(x * f32(mul)); exponent += e
    if x > 0 and x <= f32(1.0):
        for lim, mul, e in ((1e-31, 1e32, 32), (1e-15, 1e16, 16), (1e-7, 1e8, 8), (1e-3, 1e4, 4), (1e-1, 1e2, 2), (1e0, 1e1, 1)):
            if x < f32(lim):
                x = f32(x * f32(mul)); exponent -= e
    integral = int(x)
    remainder = f32(f32(x - f32(integral)) * f32(1e8))
    decimal = int(remainder)
    if f32(remainder - f32(decimal)) >= f32(0.5):
        decimal += 1
        if decimal >= 100000000:
            decimal = 0; integral += 1
            if integral >= 10:
                integral = 1; exponent += 1
    return "%s%d.%08d%s" % (s, integral, decimal, ("E%d" % exponent) if exponent else "")


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vtk_fixture")
    os.makedirs(out, exist_ok=True)
    Nx, Ny, Nz = 4, 3, 2
    N = Nx * Ny * Nz
    n = np.arange(N, dtype=np.float32)
    # units.set_m_kg_s(1, 1, 1, 2, 4, 8): m = 2, kg = 8 * 2^3 = 64, s = 1/4 * 2 = 1/2
    m, kg, s = 2.0, 64.0, 0.5
    factor = {"rho": kg / m ** 3, "u": m / s, "flags": 1, "F": kg * m / s ** 2}   # 8, 4, 1, 512
    fields = {"rho": (1.0 + n / 64.0)[:, None], "u": np.stack([n / 128.0, -n / 256.0, n / 512.0], axis=1),
              "flags": ((np.arange(N) * 7) % 256).astype(np.uint8)[:, None], "F": np.stack([n / 16.0, -n / 32.0, n / 8.0], axis=1)}
    for name, values in fields.items():
        file = "%s-%09d.vtk" % (name, 0)
        header = ("# vtk DataFile Version 3.0\nFluidX3D %s\nBINARY\nDATASET STRUCTURED_POINTS\nDIMENSIONS 4 3 2\n"
                  "ORIGIN -3.00000000 -2.00000000 -1.00000000\nSPACING 2.00000000 2.00000000 2.00000000\nPOINT_DATA 24\n"
                  "SCALARS data %s %d\nLOOKUP_TABLE default\n") % (file, "unsigned_char" if name == "flags" else "float", values.shape[1])
        payload = b""
        for cell in values:           # cell after cell, the components of a cell side by side, most significant byte first
            for v in cell:
                payload += struct.pack("B", int(v)) if name == "flags" else struct.pack(">f", float(v) * factor[name])
        with open(os.path.join(out, file), "wb") as f:
            f.write(header.encode() + payload)
    nu = f32(0.125)
    re_max = f32(f32(f32(0.57735027) * np.sqrt(f32(Nx * Nx + Ny * Ny + Nz * Nz))) / nu)
    clock = r"<<<(?:\d+[ydhm])*\d+s>>>"  # print_time, utilities.hpp:2876-2888
    status = ("Grid Resolution = 4 x 3 x 2 = 24\nGrid Domains = 1 x 1 x 1 = 1\nLBM Type = D3Q19 SRT (FP32/FP32)\n"
              "Memory Usage = CPU 0 MB, GPU 1x 0 MB\nMaximum Allocation Size = 0 MB\nTime Steps = 0 / infinite\n"
              "Runtime = %s (total) = %s (LBM) + %s (rendering and data evaluation)\nAverage MLUPs/s = <<<\\d+>>>\n"
              "Kinematic Viscosity = %s\nRelaxation Time = %s\nMaximum Reynolds Number = %s\nVolume Force = (%s, %s, %s)\n") % (
        clock, clock, clock, ref_float_string(nu), ref_float_string(f32(f32(3.0) * nu + f32(0.5))), ref_float_string(re_max),
        ref_float_string(0.0), ref_float_string(0.0), ref_float_string(0.0009765625))
    with open(os.path.join(out, "status-%09d.txt" % 0), "w") as f:
        f.write(status)
    print(status)


if __name__ == "__main__":
    main()
