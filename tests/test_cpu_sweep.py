"""tools/cpu_sweep.py under test: a few of its random configurations per codec (generator reconstruction = oracle output; the product's host parser
digest = the oracle's syntax digest).  The tool itself runs thousands (profiles/r06_cpu_sweep.txt)."""
import pytest

from tools import cpu_sweep


@pytest.mark.parametrize("codec", [0, 1])
def test_a_few_random_configurations(codec):
    for i in range(4):
        assert cpu_sweep.one((codec, 77, i, False)) is None
