"""GPU: the driver's smoke() path, through the C ABI, against the oracle."""
import pytest


@pytest.mark.gpu
def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
