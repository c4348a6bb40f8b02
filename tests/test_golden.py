"""The oracle reproduces the committed golden fixtures bit for bit (regression pin of the checker)."""
import numpy as np
import pytest

import tests.oracle_binding as ob
from tests.golden_util import golden_cases, load_golden


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_golden(name):
    d, film, counters = load_golden(name)
    o = ob.OracleScene(d)
    out = o.render(threads=2)          # thread count must not matter: blocks are independent
    st = o.last_stats
    two_pass = "2pass" in name         # film += pass results: order of the two additions is commutative -> still exact
    assert np.array_equal(out, film) or (two_pass and np.allclose(out, film, rtol=1e-6))
    assert [st["n_iter"], st["n_lookup"], st["n_nee_step"], st["samples"]] == list(counters)


def test_golden_weights_and_alpha():
    for name in golden_cases():
        _, film, counters = load_golden(name)
        spp = counters[3] / (film.shape[0] * film.shape[1])
        assert np.all(film[..., 4] == spp)                 # W channel: one unit per sample (box filter)
        assert np.all(film[..., 3] <= spp) and np.isfinite(film).all()
