"""Writes the committed golden vectors (inputs + expected outputs).  The expected values
come from the CPU oracle (oracle/mtp_oracle.c) -- NOT from a run of the reference, which
cannot be built in this image ("Oracle"); they pin the oracle against
regressions and give the GPU tests fixed vectors that need no generator run.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lammps_mtp_kokkos_amd import mtpgen  # noqa: E402
from lammps_mtp_kokkos_amd.driver import periodic_system  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

CASES = [("W_L8_54", "W_L8.mtp", (3, 3, 3), 1, False), ("W_L16_54", "W_L16.mtp", (3, 3, 3), 1, False),
         ("WRe_L20_16", "WRe_L20.mtp", (2, 2, 2), 2, False), ("W_L16_nbh_16", "W_L16_nbh.almtp", (2, 2, 2), 1, True),
         ("WRe_L10_cfg_16", "WRe_L10_cfg.almtp", (2, 2, 2), 2, True)]

if __name__ == "__main__":
    for name, potf, ncell, species, ext in CASES:
        pos, box = mtpgen.bcc_lattice(*ncell, seed=2024)
        types = np.random.default_rng(11).integers(1, species + 1, size=len(pos)).astype(np.int32)
        s = periodic_system(pos, box, types, 7.0)
        o = Oracle(os.path.join(ROOT, "potentials", potf), selection=ext)
        r = o.compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=ext, natoms=s.nlocal)
        extra = {}
        if ext:
            extra = dict(grades=r["grades"], max_grade=r["max_grade"], coeff_ders=r["coeff_ders"])
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), potential=potf, x=s.x,
                            types=s.types, ilist=s.ilist, first=s.first, neigh=s.neigh, nlocal=s.nlocal,
                            owner=s.owner, f=r["f"], eatom=r["eatom"], energy=r["energy"], virial=r["virial"],
                            vatom=r["vatom"], **extra)
        print(name, s.nlocal, s.nall, "E=%.12f" % r["energy"])
