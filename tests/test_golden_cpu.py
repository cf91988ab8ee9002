"""The committed golden vectors must be reproduced bit-for-bit by the oracle (they were
written by it: a regression pin, see tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle.pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_reproduces_golden_vectors():
    gdir = os.path.join(ROOT, "tests", "golden")
    names = sorted(f for f in os.listdir(gdir) if f.endswith(".npz"))
    assert len(names) >= 5
    for n in names:
        g = np.load(os.path.join(gdir, n))
        ext = "grades" in g.files
        o = Oracle(os.path.join(ROOT, "potentials", str(g["potential"])), selection=ext)
        r = o.compute(g["x"], g["types"], g["ilist"], g["first"], g["neigh"], extrapolation=ext,
                      natoms=int(g["nlocal"]))
        np.testing.assert_allclose(r["f"], g["f"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(r["eatom"], g["eatom"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(r["virial"], g["virial"], rtol=1e-12, atol=1e-12)
        if ext:
            np.testing.assert_allclose(r["grades"], g["grades"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(r["coeff_ders"], g["coeff_ders"], rtol=1e-12, atol=1e-12)
