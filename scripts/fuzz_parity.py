#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (not part of the test suite): random levels, species counts, densities
(one to three 32-neighbour tiles), ragged lists, flags and both LDS plans.
  python scripts/fuzz_parity.py [cases] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lammps_mtp_kokkos_amd import capi, mtpgen  # noqa: E402
from lammps_mtp_kokkos_amd.driver import periodic_system  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
tmp = tempfile.mkdtemp()
worst = 0.0
for case in range(ncase):
    level = int(rng.choice([6, 8, 10, 12, 14, 16, 18]))
    species = int(rng.integers(1, 4))
    a = float(rng.choice([3.165, 2.9, 2.6, 2.3]))
    cells = tuple(int(v) for v in rng.integers(2, 5, 3))
    grade = bool(rng.integers(0, 2)) and species <= 2
    plan = int(rng.integers(0, 2))
    tab = mtpgen.level8_template() if level == 8 else mtpgen.build_table(level)
    pot_d = mtpgen.random_potential(tab, species, int(rng.integers(1, 10 ** 6)))
    if grade:
        mtpgen.add_selection_state(pot_d, "nbh", seed=int(rng.integers(1, 1000)))
    path = os.path.join(tmp, "p%d.almtp" % case)
    mtpgen.write_mtp(pot_d, path)
    pos, box = mtpgen.bcc_lattice(*cells, a=a, jitter=0.08, seed=int(rng.integers(1, 10 ** 6)))
    types = rng.integers(1, species + 1, len(pos)).astype(np.int32)
    s = periodic_system(pos, box, types, 6.5)
    # ragged: drop a random subset of rows (subset ilist), keep full lists for the kept atoms
    keep = np.sort(rng.choice(s.nlocal, max(1, int(s.nlocal * rng.uniform(0.5, 1.0))), replace=False)).astype(np.int32)
    first = np.zeros(len(keep) + 1, np.int32)
    first[1:] = np.cumsum(s.first[keep + 1] - s.first[keep])
    neigh = np.concatenate([s.neigh[s.first[i]:s.first[i + 1]] for i in keep]) if len(keep) else np.zeros(0, np.int32)
    os.environ["MTP_REBUILD_TABLES"] = str(plan)
    pot = capi.Potential(path, selection=grade)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(keep, first, neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4, grade=grade)
    want = Oracle(path, selection=grade).compute(s.x, s.types, keep, first, neigh, extrapolation=grade,
                                                 natoms=len(keep))
    scale = max(1.0, np.abs(want["f"]).max())
    ef = np.abs(got["f"] - want["f"]).max() / scale
    ee = abs(got["energy"] - want["energy"]) / max(1.0, abs(want["energy"]))
    ev = np.abs(got["virial"] - want["virial"]).max() / max(1.0, np.abs(want["virial"]).max())
    eg = 0.0
    if grade:
        eg = np.abs(got["grades"][keep] - want["grades"][keep]).max() / max(1.0, np.abs(want["grades"][keep]).max())
    mx = int(np.diff(first).max()) if len(keep) else 0
    worst = max(worst, ef, ee, ev, eg)
    print("case %2d level %2d species %d a %.3f cells %s rows %4d maxrow %3d grade %d plan %d  rel err F %.1e E %.1e V %.1e G %.1e"
          % (case, level, species, a, cells, len(keep), mx, grade, plan, ef, ee, ev, eg), flush=True)
print("fuzz_parity: worst relative error %.2e" % worst)
sys.exit(0 if worst < 1e-9 else 1)
