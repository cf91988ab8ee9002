"""A reader of committed MLIP-3 text potentials for the tests of the independent net (tests/test_oracle.py): plain
string handling, sharing no code with the oracle's or the product's parser.  Returns an mtpgen.Potential whose table is
the generator's level-L table, after checking that the file's alpha tables ARE that table (so the generator's
contraction graphs describe the file's scalars, in order)."""
import re

import numpy as np

from lammps_mtp_kokkos_amd import mtpgen


def _ints(txt, key):
    m = re.search(r"^%s\s*=\s*(.*)$" % key, txt, flags=re.M)
    return [int(v) for v in re.findall(r"-?\d+", m.group(1))]


def _floats(line):
    return [float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", line)]


def read_committed(path, level):
    raw = open(path, "rb").read()
    cut = raw.find(b"#MVS_v1.1")
    txt = (raw if cut < 0 else raw[:cut]).decode()
    Sp = _ints(txt, "species_count")[0]
    R = _ints(txt, r"\s*radial_basis_size")[0]
    Mu = _ints(txt, r"\s*radial_funcs_count")[0]
    rmin = float(re.search(r"min_dist\s*=\s*(\S+)", txt).group(1))
    rmax = float(re.search(r"max_dist\s*=\s*(\S+)", txt).group(1))
    sc = re.search(r"^scaling\s*=\s*(\S+)", txt, flags=re.M)
    lines = txt.split("\n")
    rc = np.zeros((Sp * Sp, Mu, R))
    k = next(i for i, l in enumerate(lines) if l.strip() == "radial_coeffs") + 1
    for _ in range(Sp * Sp):
        t1, t2 = (int(v) for v in lines[k].strip().split("-"))
        for mu in range(Mu):
            rc[t1 * Sp + t2, mu] = _floats(lines[k + 1 + mu])
        k += 1 + Mu
    table = mtpgen.build_table(level)
    basic = np.array(_ints(txt, "alpha_index_basic")).reshape(-1, 4)
    times = np.array(_ints(txt, "alpha_index_times")).reshape(-1, 4)
    mapping = _ints(txt, "alpha_moment_mapping")
    assert [tuple(b) for b in basic] == list(table.basic), "the file's basics are not the generator's level table"
    assert [tuple(r) for r in times] == list(table.times) and mapping == list(table.mapping)
    assert _ints(txt, "alpha_moments_count")[0] == table.nmoments
    spc = np.array(_floats(next(l for l in lines if l.startswith("species_coeffs"))))
    mc = np.array(_floats(next(l for l in lines if l.startswith("moment_coeffs"))))
    assert len(spc) == Sp and len(mc) == len(mapping)
    pot = mtpgen.Potential(table, Sp, rmin, rmax, R, float(sc.group(1)) if sc else 1.0, rc, spc, mc)
    if cut >= 0:
        tail = raw[cut:]
        h = tail.index(b"#", 1)
        C = pot.coeff_count
        mats = np.frombuffer(tail[h + 1: h + 1 + 2 * C * C * 8], dtype="<f8").reshape(2, C, C)
        pot.active_set, pot.inverse_active_set = mats[0].copy(), mats[1].copy()
        pot.mvs_mode = "cfg" if re.search(rb"energy_weight\s*=\s*1", tail[:h]) else "nbh"
    return pot
