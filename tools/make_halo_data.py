#!/usr/bin/env python3
"""Re-emit the reference's two halo-orbit DATA files as package data (run in the build container).

L2_Anderson_1.txt / L2_Anderson_2.txt (6 rows x 100 columns, 15 significant digits) are the only
fixtures the reference ships: two closed Earth-Moon L2 halo orbits sampled at equal time spacing
(CRTBP_Multishoot_indirect_demo.jl:66-70).  They are benchmark/test INPUT data,
not source code.  /root/reference does not exist on the GPU box, so the numbers are stored once as
lowthrustopt_amd/data/halo_L2_{1,2}.txt with full round-trip precision.
"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for k in (1, 2):
    a = np.loadtxt("/root/reference/L2_Anderson_%d.txt" % k)
    assert a.shape == (6, 100)
    out = os.path.join(ROOT, "lowthrustopt_amd", "data", "halo_L2_%d.txt" % k)
    np.savetxt(out, a, fmt="%.17g")
    b = np.loadtxt(out)
    assert np.array_equal(a, b)
    print("wrote", out)
