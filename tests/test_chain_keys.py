"""The work-list rule of chain launches (Engine::launch): every dependency of a reconstruction group has a smaller key.  tools/chain_keys.py restates what
the device code waits for, macroblock by macroblock, and checks the rule by brute force; round 4 found two violations with it after chain launches of
4 / 8 streams had given up on the GPU (profiles/r04_ab6_first_giveup.json).  No GPU needed."""
import os

import pytest

from tools import chain_keys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_model_uses_the_constants_of_the_sources():
    c = chain_keys.constants_in_sources(ROOT)
    mine = dict(BR=chain_keys.BR, DEPTH=chain_keys.DEPTH, PUB=chain_keys.PUB, KPUBLAG=chain_keys.KPUBLAG, ROW_LAG=chain_keys.ROW_LAG,
                K_BAND_LAG=chain_keys.K_BAND_LAG, KEY_SLACK=chain_keys.KEY_SLACK, INTRA_EXTRA=chain_keys.INTRA_EXTRA, CHAIN_LAG=chain_keys.CHAIN_LAG)
    assert c == mine


@pytest.mark.parametrize("size", [(120, 68), (240, 135), (22, 18), (45, 30), (8, 40)])
@pytest.mark.parametrize("p_intra,g_intra", [(False, False), (False, True), (True, False), (True, True)])
def test_every_dependency_has_a_smaller_key(size, p_intra, g_intra):
    assert chain_keys.check(size[0], size[1], p_intra, g_intra) < 0


def test_the_model_sees_the_two_violations_round_4_fixed(monkeypatch):
    """(a) no extra room behind a picture with the intra role; (b) the x + 2y slope for a deblock-only picture whose bands move one row per step."""
    monkeypatch.setattr(chain_keys, "INTRA_EXTRA", -68)            # (a): successor starts lag + slack behind, as rounds 2-3 had it
    assert chain_keys.check(120, 68, True, False) >= 0
    monkeypatch.undo()
    # (b): keys of slope 2 for both pictures = "successor intra-role" keys with P's own keys at slope 2, geometry one row per step
    import types
    src = open(chain_keys.__file__).read().replace("slope_p, slope_g = (2 if p_intra else L), (2 if g_intra else L)", "slope_p, slope_g = 2, 2")
    mod = types.ModuleType("chain_keys_old"); exec(compile(src, "chain_keys_old", "exec"), mod.__dict__)
    assert mod.check(120, 68, False, False) >= 0
