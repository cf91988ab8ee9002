"""The BASELINE.json configurations at their real sizes and table shapes, HIP path (through the C ABI) against the
CPU oracle.  Where the oracle cannot walk every atom in seconds (65,536 atoms), forces are pinned on *balls*:
for every atom i of a ball S, the total force on i only involves pairs (k, j) with k within the cutoff of i, so the
oracle run on ilist' = {atoms within r0 + rc of the centre} reproduces the GPU's whole-system force on S exactly
(same pairs, /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:236-254).  Grades and site energies depend on the atom's own
row only (pair_mtp_extrapolation.cpp:332-358), so they are sampled at random.

Tolerances as in test_gpu_parity.py: |dF| <= 1e-9 eV/A + 1e-10 |F|max; grades 1e-9 relative.
"""
import os

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")
RC = 5.0


def _oracle(path, selection=False):
    from oracle.pyoracle import Oracle
    return Oracle(path, selection=selection)


def _close(got, want, what, atol=1e-9, rtol=1e-10):
    scale = max(1.0, float(np.abs(want).max())) if np.size(want) else 1.0
    err = float(np.abs(np.asarray(got) - np.asarray(want)).max()) if np.size(want) else 0.0
    assert err <= atol + rtol * scale, "%s: max abs err %.3e (scale %.3e)" % (what, err, scale)
    return err


def _sub_list(s, rows):
    rows = np.asarray(rows, dtype=np.int64)
    first = np.zeros(len(rows) + 1, np.int32)
    first[1:] = np.cumsum(s.first[rows + 1] - s.first[rows])
    neigh = np.concatenate([s.neigh[s.first[i]:s.first[i + 1]] for i in rows]) if len(rows) else np.zeros(0, np.int32)
    return rows.astype(np.int32), first, neigh


def _ball(s, centre, r0):
    """(S, ilist'): owned atoms within r0 / within r0 + rc (+ margin) of the centre, minimum image."""
    d = s.x[: s.nlocal] - np.asarray(centre)
    d -= s.box * np.round(d / s.box)
    r = np.sqrt((d * d).sum(1))
    return np.nonzero(r < r0)[0], np.nonzero(r < r0 + RC + 1e-6)[0]


def _ball_force_parity(s, got_f, orc, centres, r0, extrapolation=False, what="forces"):
    """got_f: whole-system GPU forces [nall,3] (ghost rows included)."""
    F_gpu = s.fold_forces(got_f)
    worst = 0.0
    for c in centres:
        S, sup = _ball(s, c, r0)
        assert len(S) > 20
        il, first, neigh = _sub_list(s, sup)
        want = orc.compute(s.x, s.types, il, first, neigh, eflag=0, vflag=0, extrapolation=extrapolation,
                           natoms=len(il))
        F_ref = s.fold_forces(want["f"])
        worst = max(worst, _close(F_gpu[S], F_ref[S], "%s on the ball at %s" % (what, c)))
    return worst


def _append_selection(src, dst, C, mode="nbh", seed=99):
    """Appends a synthetic #MVS_v1.1 block (well-conditioned active set A = 2 I + 0.05 U(-1,1) and its inverse,
    SURVEY.md 8d) to an MLIP-3 text file; the layout pair_mtp_extrapolation.cpp:550-611 reads."""
    rng = np.random.default_rng(seed)
    A = 2.0 * np.eye(C) + 0.05 * rng.uniform(-1, 1, size=(C, C))
    cfg = mode == "cfg"
    tail = ["#MVS_v1.1", "energy_weight = %d" % (1 if cfg else 0), "force_weight = 0", "stress_weight = 0",
            "site_en_weight = %d" % (0 if cfg else 1), "weight_scaling = 1"]
    with open(src, "rb") as fh:
        data = fh.read()
    data += ("\n".join(tail) + "\n").encode() + b"#" + A.astype("<f8").tobytes() + np.linalg.inv(A).astype("<f8").tobytes()
    with open(dst, "wb") as fh:
        fh.write(data)
    return dst


def _full_compare(path, s, variant=None):
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    if variant is not None:
        ctx.set_variant(variant)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4)
    want = _oracle(path).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    _close(got["f"], want["f"], "forces")
    n = len(s.ilist)
    assert abs(got["energy"] - want["energy"]) / n <= 1e-10 * max(1.0, abs(want["energy"]) / n)
    _close(got["eatom"], want["eatom"], "eatom", atol=1e-10)
    _close(got["virial"], want["virial"], "virial", atol=1e-8, rtol=1e-10)
    _close(got["vatom"], want["vatom"], "vatom")
    return got, want, ctx


# ---- config 3: 2,048-atom W, level 16, block-parallel ("small") variant -------------------------------------

def test_config3_2048_atoms_small_variant_full_parity():
    pos, box = mtpgen.bcc_lattice(8, 8, 16)
    s = periodic_system(pos, box, None, 7.0)
    assert s.nlocal == 2048
    got, want, ctx = _full_compare(os.path.join(POT, "W_L16.mtp"), s, variant=capi.VARIANT_SMALL)
    info = ctx.launch_info()
    assert info["waves_per_block"] * info["grid_blocks"] >= 2048      # every atom has its own wavefront
    # ... and the thread-parallel ("large") variant gives the same numbers
    got2, _, _ = _full_compare(os.path.join(POT, "W_L16.mtp"), s, variant=capi.VARIANT_LARGE)
    _close(got2["f"], got["f"], "large vs small variant", atol=1e-10)


# ---- config 2: 65,536-atom W, level 16: forces on balls, site energies sampled ----------------------------------

@pytest.fixture(scope="module")
def w64k():
    pos, box = mtpgen.bcc_lattice(32, 32, 32)
    s = periodic_system(pos, box, None, 7.0)
    assert s.nlocal == 65536
    return s


def test_config2_64k_force_parity_on_balls(w64k):
    s = w64k
    path = os.path.join(POT, "W_L16.mtp")
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4)
    orc = _oracle(path)
    # one ball in the bulk, one across a periodic corner (ghost forces folded), one across a face
    centres = [s.box * 0.5, np.zeros(3), np.array([0.0, 0.5, 0.3]) * s.box]
    _ball_force_parity(s, got["f"], orc, centres, r0=9.0)
    # site energies and per-atom virials of 1,024 random atoms
    pick = np.random.default_rng(1).choice(s.nlocal, 1024, replace=False)
    il, first, neigh = _sub_list(s, pick)
    want = orc.compute(s.x, s.types, il, first, neigh, eflag=3, vflag=4)
    _close(got["eatom"][pick], want["eatom"][pick], "sampled eatom", atol=1e-10)
    _close(got["vatom"][pick], want["vatom"][pick], "sampled vatom")
    assert abs(got["energy"] - got["eatom"][: s.nlocal].sum()) < 1e-6
    assert np.abs(s.fold_forces(got["f"]).sum(0)).max() < 1e-7


def _threads():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 32))


def _full_parity_threaded(path, s):
    """EVERY force, site energy, per-atom virial and the totals of a 65,536-atom call against the threaded oracle
    (oracle/mtp_oracle_mt.c, pinned to the serial restatement in tests/test_oracle.py)."""
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4)
    want = _oracle(path).compute_mt(_threads(), s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    _close(got["f"], want["f"], "all forces (ghost rows included)")
    _close(got["eatom"], want["eatom"], "all site energies", atol=1e-10)
    _close(got["vatom"], want["vatom"], "all per-atom virials")
    n = s.nlocal
    assert abs(got["energy"] - want["energy"]) / n <= 1e-10 * max(1.0, abs(want["energy"]) / n)
    _close(got["virial"], want["virial"], "virial", atol=1e-7, rtol=1e-10)


def test_config2_64k_every_force_against_the_threaded_oracle(w64k):
    _full_parity_threaded(os.path.join(POT, "W_L16.mtp"), w64k)


def test_config4_shard_64k_level20_every_force_against_the_threaded_oracle():
    """the per-GPU shard of the 512k-atom W-Re case: 65,536 atoms, level 20, two species"""
    s = _wre((32, 32, 32))
    assert s.nlocal == 65536
    _full_parity_threaded(os.path.join(POT, "WRe_L20.mtp"), s)


# ---- config 5: 65,536-atom W, neighbourhood grades every step -----------------------------------------------------

def test_config5_64k_grades_and_forces(w64k):
    s = w64k
    path = os.path.join(POT, "W_L16_nbh.almtp")
    pot = capi.Potential(path, selection=True)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=0, grade=True)
    orc = _oracle(path, selection=True)
    pick = np.random.default_rng(2).choice(s.nlocal, 2048, replace=False)
    il, first, neigh = _sub_list(s, pick)
    want = orc.compute(s.x, s.types, il, first, neigh, eflag=3, vflag=0, extrapolation=True, natoms=len(il))
    _close(got["grades"][pick], want["grades"][pick], "sampled grades", atol=1e-9, rtol=1e-9)
    _close(got["eatom"][pick], want["eatom"][pick], "sampled eatom (grade call)", atol=1e-10)
    assert not got["grades"][s.nlocal:].any()
    # the global maximum is attained somewhere: it bounds the sample and equals max over the owned atoms
    assert abs(got["max_grade"] - got["grades"][: s.nlocal].max()) <= 1e-12 * max(1.0, got["max_grade"])
    assert got["max_grade"] >= want["grades"][pick].max() * (1 - 1e-9)
    _ball_force_parity(s, got["f"], orc, [s.box * 0.5, np.zeros(3)], r0=8.0, extrapolation=True,
                       what="forces (grade call)")


def test_config5_64k_every_grade_against_the_serial_oracle(w64k):
    """all 65,536 neighbourhood grades, forces and site energies of one grade call (the serial oracle walks them in a
    few seconds)"""
    s = w64k
    path = os.path.join(POT, "W_L16_nbh.almtp")
    pot = capi.Potential(path, selection=True)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=0, grade=True)
    want = _oracle(path, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=0,
                                                 extrapolation=True, natoms=s.nlocal)
    _close(got["grades"], want["grades"], "all grades", atol=1e-9, rtol=1e-9)
    _close(got["f"], want["f"], "all forces (grade call)")
    _close(got["eatom"], want["eatom"], "all site energies (grade call)", atol=1e-10)
    assert abs(got["max_grade"] - want["grades"].max()) <= 1e-9 * max(1.0, want["grades"].max())


# ---- config 4: W-Re level 20 (C = 622: the grade GEMM with operands from L2, mtp_grade_kernel<0>) ---------------

def _wre(ncell, seed=4242, frac=0.10):
    pos, box = mtpgen.bcc_lattice(*ncell)
    types = (np.random.default_rng(seed).random(len(pos)) < frac).astype(np.int32) + 1     # W + 10 % Re
    return periodic_system(pos, box, types, 7.0)


def test_config4_level20_two_species_8k_atoms_force_parity_on_balls():
    s = _wre((16, 16, 16))
    assert s.nlocal == 8192
    path = os.path.join(POT, "WRe_L20.mtp")
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4)
    orc = _oracle(path)
    _ball_force_parity(s, got["f"], orc, [s.box * 0.5, np.zeros(3)], r0=7.0)
    pick = np.random.default_rng(3).choice(s.nlocal, 256, replace=False)
    il, first, neigh = _sub_list(s, pick)
    want = orc.compute(s.x, s.types, il, first, neigh, eflag=3, vflag=4)
    _close(got["eatom"][pick], want["eatom"][pick], "sampled eatom", atol=1e-10)
    _close(got["vatom"][pick], want["vatom"][pick], "sampled vatom")


def test_level20_two_species_neighbourhood_grades_c622(tmp_path):
    """C = 622 > 160: candidate vectors through mtp_cvec_kernel (Mu = 5 is outside the fused shape) and the
    grade GEMM with both operands from L2 (mtp_grade_kernel<0>)."""
    src = os.path.join(POT, "WRe_L20.mtp")
    C = capi.Potential(src).info.coeff_count
    assert C == 622
    path = _append_selection(src, str(tmp_path / "WRe_L20_nbh.almtp"), C, "nbh", seed=17)
    s = _wre((4, 4, 4), frac=0.3)
    pot = capi.Potential(path, selection=True)
    assert pot.info.has_selection and not pot.info.configuration_mode
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4, grade=True)
    want = _oracle(path, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4,
                                                 extrapolation=True, natoms=s.nlocal)
    _close(got["f"], want["f"], "forces (grade call)")
    _close(got["eatom"], want["eatom"], "eatom (grade call)", atol=1e-10)
    _close(got["grades"][s.ilist], want["grades"][s.ilist], "grades C=622", atol=1e-9, rtol=1e-9)
    assert abs(got["max_grade"] - want["max_grade"]) <= 1e-9 * max(1.0, want["max_grade"])


def test_level20_two_species_configuration_mode_c622(tmp_path):
    src = os.path.join(POT, "WRe_L20.mtp")
    path = _append_selection(src, str(tmp_path / "WRe_L20_cfg.almtp"), 622, "cfg", seed=18)
    s = _wre((3, 3, 3), frac=0.3)
    pot = capi.Potential(path, selection=True)
    assert pot.info.configuration_mode
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    got = ctx.compute(s.x, s.types, eflag=3, vflag=4, grade=True)
    want = _oracle(path, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4,
                                                 extrapolation=True, natoms=s.nlocal)
    _close(got["coeff_ders"], want["coeff_ders"], "sum_i dE_i/dtheta", atol=1e-9, rtol=1e-10)
    g = pot.cfg_grade(got["coeff_ders"]) / s.nlocal
    assert abs(g - want["max_grade"]) <= 1e-9 * max(1.0, want["max_grade"])
