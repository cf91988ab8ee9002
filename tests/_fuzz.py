"""One randomised GPU-vs-oracle parity case: random level, species count, density (one to three 32-neighbour
tiles), ragged subset lists, grade calls and all three LDS layouts.  Shared by tests/test_gpu_fuzz.py (fixed seeds, part
of `pytest -m gpu`) and scripts/fuzz_parity.py (open-ended sweeps)."""
import os

import numpy as np

from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system


def fuzz_case(rng, tmpdir, tag="p"):
    """Runs one case; returns (description, dict of relative errors F / E / V / G)."""
    from oracle.pyoracle import Oracle
    level = int(rng.choice([6, 8, 10, 12, 14, 16, 18]))
    species = int(rng.integers(1, 4))
    a = float(rng.choice([3.165, 2.9, 2.6, 2.3]))
    cells = tuple(int(v) for v in rng.integers(2, 5, 3))
    grade = bool(rng.integers(0, 2)) and species <= 2
    plan = str(rng.choice(["keep", "lean", "rebuild"]))
    tab = mtpgen.level8_template() if level == 8 else mtpgen.build_table(level)
    pot_d = mtpgen.random_potential(tab, species, int(rng.integers(1, 10 ** 6)))
    if grade:
        mtpgen.add_selection_state(pot_d, "nbh", seed=int(rng.integers(1, 1000)))
    path = os.path.join(str(tmpdir), "%s.almtp" % tag)
    mtpgen.write_mtp(pot_d, path)
    pos, box = mtpgen.bcc_lattice(*cells, a=a, jitter=0.08, seed=int(rng.integers(1, 10 ** 6)))
    types = rng.integers(1, species + 1, len(pos)).astype(np.int32)
    s = periodic_system(pos, box, types, 6.5)
    # ragged: a random subset of the owned atoms as ilist, their full rows
    keep = np.sort(rng.choice(s.nlocal, max(1, int(s.nlocal * rng.uniform(0.5, 1.0))), replace=False)).astype(np.int32)
    first = np.zeros(len(keep) + 1, np.int32)
    first[1:] = np.cumsum(s.first[keep + 1] - s.first[keep])
    neigh = np.concatenate([s.neigh[s.first[i]:s.first[i + 1]] for i in keep]) if len(keep) else np.zeros(0, np.int32)
    wps = str(rng.choice([2, 3]))          # register build (3 only exists for the narrow lane grids: ignored elsewhere)
    old, old_w = os.environ.get("MTP_LAYOUT"), os.environ.get("MTP_WPS")
    os.environ["MTP_LAYOUT"] = plan
    os.environ["MTP_WPS"] = wps
    try:
        pot = capi.Potential(path, selection=grade)
        ctx = capi.Context(pot, 0)
        ctx.set_neighbors(keep, first, neigh, s.nall)
        got = ctx.compute(s.x, s.types, eflag=3, vflag=4, grade=grade)
    finally:
        for k, v in (("MTP_LAYOUT", old), ("MTP_WPS", old_w)):
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    want = Oracle(path, selection=grade).compute(s.x, s.types, keep, first, neigh, extrapolation=grade,
                                                 natoms=len(keep))
    scale = max(1.0, np.abs(want["f"]).max())
    err = dict(F=np.abs(got["f"] - want["f"]).max() / scale,
               E=abs(got["energy"] - want["energy"]) / max(1.0, abs(want["energy"])),
               V=np.abs(got["virial"] - want["virial"]).max() / max(1.0, np.abs(want["virial"]).max()),
               G=0.0)
    if grade:
        err["G"] = np.abs(got["grades"][keep] - want["grades"][keep]).max() / max(1.0, np.abs(want["grades"][keep]).max())
    mx = int(np.diff(first).max()) if len(keep) else 0
    desc = "level %2d species %d a %.3f cells %s rows %4d maxrow %3d grade %d layout %s wps %s" % (
        level, species, a, cells, len(keep), mx, grade, plan, wps)
    return desc, err
