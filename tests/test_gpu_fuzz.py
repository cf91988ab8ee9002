"""Fixed-seed cases of the randomised GPU-vs-oracle sweep (tests/_fuzz.py): random levels, species counts,
densities (one to three 32-neighbour tiles), ragged subset lists, grade calls, all three LDS layouts.  Tolerance: 1e-9
relative on forces, energy, virial and grades (fp64 re-association only)."""
import numpy as np
import pytest

from _fuzz import fuzz_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [2026, 7, 99, 31337])
def test_fuzz_parity(seed, tmp_path):
    rng = np.random.default_rng(seed)
    for case in range(4):
        desc, err = fuzz_case(rng, tmp_path, "s%d_%d" % (seed, case))
        for k, v in err.items():
            assert v < 1e-9, "seed %d case %d (%s): relative %s error %.2e" % (seed, case, desc, k, v)
