"""GPU: a short fixed-seed slice of scripts/fuzz_parity.py — random ragged shapes for every kernel, each compared
with its oracle at the tolerances of the dedicated tests (bit-exact for Gray-Scott, LBM, masks and SPH cell indices)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_shapes(eng, oracle_built, seed):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    msgs = []
    it, bad = m.sweep(seed=seed, seconds=90.0, max_iter=90, log=msgs.append)
    assert it == 90 and bad == 0, "\n".join(msgs)
