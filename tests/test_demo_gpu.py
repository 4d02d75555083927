"""GPU end-to-end: the indirect part of the reference demo (CRTBP_Multishoot_indirect_demo.jl) -- 30-node L2 halo ->
halo transfer, p = 2 then p = 1 at 0.05 N, rho continuation -- converges with every defect / Jacobian / Newton solve
on the device."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_halo_transfer_demo_converges():
    spec = importlib.util.spec_from_file_location("halo_demo", os.path.join(ROOT, "examples", "halo_transfer_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(seed=0, verbose=False, rho_target=1e-2)
    assert res["p2"][0] == 0 and res["p2"][1] <= 1e-10          # indirect.jl:280 convergence threshold
    assert res["p1"][0] == 0 and res["p1"][1] <= 1e-10
    assert res["rho"][0] == 0 and res["rho"][1] <= 1e-10
