"""Pins the CPU oracle (oracle/mtp_oracle.c) and the table generator with checks that
share no code with either: closed forms, an einsum evaluation from the mathematical
definition, finite differences and symmetry invariants.  (The reference has no tests or
fixtures of its own to borrow -- SURVEY.md section 4 -- and cannot be compiled in this image;
the oracle header says "parity unpinned".)
"""
import numpy as np
import pytest

from lammps_mtp_kokkos_amd import mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system
from oracle.pyoracle import Oracle

import _definition as defn


def _make(tmp, level, species=1, mvs=None, seed=12345, name=None, damp=0.25, template8=False):
    tab = mtpgen.level8_template() if template8 else mtpgen.build_table(level)
    pot = mtpgen.random_potential(tab, species, seed, damp=damp)
    if mvs:
        mtpgen.add_selection_state(pot, mvs)
    path = str(tmp / (name or "L%d_s%d_%s.mtp" % (level, species, mvs)))
    mtpgen.write_mtp(pot, path)
    return pot, path


def _system(ncell=(3, 3, 3), species=1, seed=777, jitter=0.05, list_cutoff=7.0):
    pos, box = mtpgen.bcc_lattice(*ncell, jitter=jitter, seed=seed)
    rng = np.random.default_rng(5)
    types = rng.integers(1, species + 1, size=len(pos)).astype(np.int32)
    return periodic_system(pos, box, types, list_cutoff)


def test_generator_sizes_match_survey_enumeration():
    # SURVEY.md App. C table (S complete, B) -- independent brute-force count there.
    expect = {2: (1, 1), 4: (2, 1), 6: (5, 5), 8: (9, 11), 10: (17, 25), 12: (32, 46), 14: (61, 81),
              16: (116, 130)}
    for L, (S, B) in expect.items():
        t = mtpgen.build_table(L)
        assert (len(t.mapping), len(t.basic)) == (S, B), L


def test_generator_level8_is_known_answer():
    """Same basics, sizes and scalar order as the level-8 template of SURVEY.md App. A; the
    rows agree except where the generator picks a shallower (balanced) product for M00^4, so
    the two tables are compared as polynomials on random moments."""
    t = mtpgen.build_table(8)
    k = mtpgen.LEVEL8_KNOWN_ANSWER
    assert t.basic == k["basic"]
    assert t.mapping == k["mapping"] and t.nmoments == k["nmoments"] and len(t.times) == len(k["times"])
    assert len(set(t.times) ^ set(k["times"])) <= 2
    rng = np.random.default_rng(0)

    def scalars(times):
        m = np.zeros(k["nmoments"])
        m[:11] = rng_vals
        for a0, a1, mu, a3 in times:
            m[a3] += mu * m[a0] * m[a1]
        return m[k["mapping"]]

    for _ in range(3):
        rng_vals = rng.uniform(-1, 1, 11)
        np.testing.assert_allclose(scalars(t.times), scalars(k["times"]), rtol=1e-14)


def test_parser_roundtrip(tmp_path):
    pot, path = _make(tmp_path, 10, species=2, mvs="nbh")
    o = Oracle(path, selection=True)
    s = o.sizes
    t = pot.table
    assert (s["B"], s["T"], s["S"], s["A"], s["Mu"], s["R"], s["Sp"]) == \
        (len(t.basic), len(t.times), len(t.mapping), t.nmoments, t.radial_funcs, 8, 2)
    assert s["C"] == pot.coeff_count
    np.testing.assert_array_equal(o.arr("alpha_index_basic", 4 * s["B"], np.int64).reshape(-1, 4), np.array(t.basic))
    np.testing.assert_array_equal(o.arr("alpha_index_times", 4 * s["T"], np.int64).reshape(-1, 4), np.array(t.times))
    np.testing.assert_allclose(o.arr("radial_basis_coeffs", pot.radial_coeffs.size), pot.radial_coeffs.ravel(), rtol=1e-15)
    np.testing.assert_allclose(o.arr("linear_coeffs", s["S"]), pot.moment_coeffs, rtol=1e-15)
    np.testing.assert_array_equal(o.arr("inverse_active_set", s["C"] ** 2), pot.inverse_active_set.ravel())
    assert o.m.configuration_mode == 0


def test_parser_rejects_bad_files(tmp_path):
    pot, path = _make(tmp_path, 8)
    txt = open(path).read()
    for bad, what in [(txt.replace("MTP\n", "XTP\n", 1), "Only MTP"),
                      (txt.replace("version = 1.1.0", "version = 1.0.0"), "version"),
                      (txt.replace("RBChebyshev", "RBFoo"), "radial basis"),
                      (txt.replace("{1, 0, 0, 0}}", "{0, 0, 0, 0}}"), "Wrong number of radial")]:
        p = tmp_path / "bad.mtp"
        p.write_text(bad)
        with pytest.raises(RuntimeError, match=what):
            Oracle(str(p))
    with pytest.raises(RuntimeError, match="No selection state"):
        Oracle(path, selection=True)


def test_chebyshev_closed_form(tmp_path):
    pot, path = _make(tmp_path, 8)
    o = Oracle(path)
    h = 1e-6
    for r in np.linspace(2.0, 5.0, 13):
        v, d = o.radial_basis(r)
        xi = (2 * r - 7.0) / 3.0
        want = np.array([defn.chebyshev_T(k, xi) for k in range(8)]) * (r - 5.0) ** 2
        np.testing.assert_allclose(v, want, rtol=1e-12, atol=1e-13)
        vp, _ = o.radial_basis(r + h)
        vm, _ = o.radial_basis(r - h)
        np.testing.assert_allclose(d, (vp - vm) / (2 * h), rtol=1e-6, atol=1e-7)
    v, d = o.radial_basis(5.0)          # value and slope vanish at the cutoff
    assert np.abs(v).max() == 0 and np.abs(d).max() == 0


def test_level8_energy_from_hand_written_definition(tmp_path):
    pot, path = _make(tmp_path, 8, species=2, template8=True, name="L8tmpl.mtp")
    sysm = _system(species=2)
    res = Oracle(path).compute(sysm.x, sysm.types, sysm.ilist, sysm.first, sysm.neigh)
    need = [(0, 0), (1, 0), (0, 1), (0, 2)]
    for i in range(sysm.nlocal):
        js = sysm.neigh[sysm.first[i]:sysm.first[i + 1]]
        rv = sysm.x[js] - sysm.x[i]
        m = (rv ** 2).sum(1) <= 25.0
        zi = sysm.types[i] - 1
        ten = defn.site_tensors(pot, rv[m], zi, sysm.types[js[m]] - 1, need)
        e = pot.species_coeffs[zi] + defn.level8_basis(ten) @ pot.moment_coeffs
        assert abs(e - res["eatom"][i]) < 1e-11 * max(1.0, abs(e))


@pytest.mark.parametrize("level,species", [(10, 2), (12, 1), (16, 1)])
def test_generated_tables_match_einsum_definition(tmp_path, level, species):
    pot, path = _make(tmp_path, level, species=species)
    sysm = _system(ncell=(2, 2, 2) if level == 16 else (3, 3, 3), species=species)
    res = Oracle(path).compute(sysm.x, sysm.types, sysm.ilist, sysm.first, sysm.neigh)
    E, _ = defn.site_energies(pot, pot.table.graphs, sysm)
    np.testing.assert_allclose(res["eatom"][:sysm.nlocal], E, rtol=2e-11, atol=1e-11)
    assert abs(res["energy"] - E.sum()) < 1e-10 * max(1.0, abs(E.sum()))


@pytest.mark.parametrize("level,species", [(8, 2), (12, 1)])
def test_forces_are_minus_gradient_and_sum_to_zero(tmp_path, level, species):
    pot, path = _make(tmp_path, level, species=species)
    o = Oracle(path)
    pos, box = mtpgen.bcc_lattice(3, 3, 3)
    rng = np.random.default_rng(5)
    types = rng.integers(1, species + 1, size=len(pos)).astype(np.int32)

    def energy_forces(p):
        s = periodic_system(p, box, types, 7.0)
        r = o.compute(s.x, s.types, s.ilist, s.first, s.neigh)
        return r["energy"], s.fold_forces(r["f"]), r, s

    E, F, r, s = energy_forces(pos)
    assert abs(E - r["eatom"][:s.nlocal].sum()) < 1e-10 * max(1, abs(E))
    assert np.abs(F.sum(0)).max() < 1e-10 * max(1.0, np.abs(F).max())
    h = 1e-5
    for (a, c) in [(0, 0), (7, 1), (20, 2), (53, 0)]:
        pp = pos.copy(); pp[a, c] += h
        pm = pos.copy(); pm[a, c] -= h
        fd = -(energy_forces(pp)[0] - energy_forces(pm)[0]) / (2 * h)
        assert abs(fd - F[a, c]) < 2e-7 * max(1.0, np.abs(F).max())


def test_virial_is_strain_derivative(tmp_path):
    pot, path = _make(tmp_path, 8)
    o = Oracle(path)
    pos, box = mtpgen.bcc_lattice(3, 3, 3)

    def run(eps):
        s = periodic_system(pos * (1 + eps), box * (1 + eps), None, 7.0)
        return o.compute(s.x, s.types, s.ilist, s.first, s.neigh), s

    r0, s0 = run(0.0)
    h = 1e-6
    dE = (run(h)[0]["energy"] - run(-h)[0]["energy"]) / (2 * h)
    # LAMMPS sign: virial_ab = sum r_a f_b  ->  dE/d(eps) = -(vxx+vyy+vzz)
    assert abs(dE + r0["virial"][:3].sum()) < 1e-6 * max(1.0, abs(dE))
    np.testing.assert_allclose(r0["vatom"][:s0.nlocal].sum(0), r0["virial"], rtol=1e-12, atol=1e-12)


def test_rotation_translation_permutation_invariance(tmp_path):
    pot, path = _make(tmp_path, 10)
    o = Oracle(path)
    s = _system()
    r0 = o.compute(s.x, s.types, s.ilist, s.first, s.neigh)
    # random rotation about the origin + translation of every coordinate (list is index-based)
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    r1 = o.compute(s.x @ q.T + np.array([0.3, -1.2, 4.0]), s.types, s.ilist, s.first, s.neigh)
    np.testing.assert_allclose(r1["eatom"], r0["eatom"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(r1["f"], r0["f"] @ q.T, rtol=1e-10, atol=1e-11)
    # shuffle the order of each atom's neighbours
    neigh = s.neigh.copy()
    for i in range(s.nlocal):
        rng.shuffle(neigh[s.first[i]:s.first[i + 1]])
    r2 = o.compute(s.x, s.types, s.ilist, s.first, neigh)
    np.testing.assert_allclose(r2["f"], r0["f"], rtol=1e-11, atol=1e-12)


def test_species_outside_potential_is_an_error(tmp_path):
    pot, path = _make(tmp_path, 8)
    s = _system()
    t = s.types.copy()
    t[3] = 2
    with pytest.raises(RuntimeError):
        Oracle(path).compute(s.x, t, s.ilist, s.first, s.neigh)


def test_extrapolation_candidate_vector_is_dE_dtheta(tmp_path):
    """c = dE_i/dtheta (pair_mtp_extrapolation.cpp:235-252, 323-329): linear part =
    basis values; radial part checked by finite differences on the coefficient file;
    E, F identical to the plain path; grade = max|A^-1 c|."""
    pot, path = _make(tmp_path, 10, species=2, mvs="nbh")
    s = _system(ncell=(2, 2, 2), species=2)
    o = Oracle(path, selection=True)
    plain = o.compute(s.x, s.types, s.ilist, s.first, s.neigh)
    ext = o.compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    np.testing.assert_array_equal(ext["f"], plain["f"])
    assert ext["energy"] == plain["energy"]
    # per-atom candidate vector: run one atom at a time
    _, Bv = defn.site_energies(pot, pot.table.graphs, s)
    nrad = 4 * pot.table.radial_funcs * 8
    Ainv = pot.inverse_active_set
    for i in [0, 5, 11]:
        one = o.compute(s.x, s.types, s.ilist[i:i + 1], np.array([0, s.first[i + 1] - s.first[i]], dtype=np.int32),
                        s.neigh[s.first[i]:s.first[i + 1]], extrapolation=True)
        c = one["coeff_ders"]
        np.testing.assert_allclose(c[nrad + 2:], Bv[i], rtol=1e-10, atol=1e-11)
        sp = np.zeros(2); sp[s.types[i] - 1] = 1
        np.testing.assert_array_equal(c[nrad:nrad + 2], sp)
        assert abs(np.abs(Ainv @ c).max() - ext["grades"][i]) < 1e-12 * max(1, ext["grades"][i])
        # radial block by finite differences of E_i w.r.t. two coefficients
        for flat in [3, nrad - 5, (s.types[i] - 1) * 2 * pot.table.radial_funcs * 8 + 9]:
            h = 1e-6
            es = []
            for sign in (+1, -1):
                p2 = mtpgen.random_potential(pot.table, 2, 12345)
                p2.radial_coeffs = pot.radial_coeffs.copy()
                p2.radial_coeffs.reshape(-1)[flat] += sign * h
                pth = str(tmp_path / "pert.mtp")
                mtpgen.write_mtp(p2, pth)
                es.append(Oracle(pth).compute(s.x, s.types, s.ilist[i:i + 1],
                                              np.array([0, s.first[i + 1] - s.first[i]], dtype=np.int32),
                                              s.neigh[s.first[i]:s.first[i + 1]])["eatom"][i])
            fd = (es[0] - es[1]) / (2 * h)
            assert abs(fd - c[flat]) < 1e-6 * max(1.0, abs(c[flat]))
    assert ext["max_grade"] == ext["grades"][:s.nlocal].max()


def test_configuration_mode_grade(tmp_path):
    pot, path = _make(tmp_path, 8, mvs="cfg")
    s = _system(ncell=(2, 2, 2))
    o = Oracle(path, selection=True)
    assert o.m.configuration_mode == 1
    r = o.compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True, natoms=s.nlocal)
    c = r["coeff_ders"]
    assert c[2 * 8] == s.nlocal                    # species slot counts atoms
    want = np.abs(pot.inverse_active_set @ c).max() / s.nlocal
    assert abs(want - r["max_grade"]) < 1e-12 * want


@pytest.mark.parametrize("level,species,threads", [(16, 1, 4), (12, 2, 3)])
def test_threaded_oracle_matches_serial_oracle(tmp_path, level, species, threads):
    """oracle/mtp_oracle_mt.c (threads over atoms, atomic adds into the one force array: the all-cores CPU baseline and
    the checker of the full 65,536-atom GPU parity tests) against the serial restatement: same per-atom arithmetic, so
    only the order of the force / virial sums differs."""
    _, path = _make(tmp_path, level, species)
    s = _system((4, 4, 4), species)
    o = Oracle(path)
    a = o.compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    b = o.compute_mt(threads, s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    assert np.abs(a["f"] - b["f"]).max() <= 1e-12 * max(1.0, np.abs(a["f"]).max())
    assert np.array_equal(a["eatom"], b["eatom"])          # one thread per atom: bit-identical
    assert abs(a["energy"] - b["energy"]) <= 1e-11 * max(1.0, abs(a["energy"]))
    assert np.abs(a["virial"] - b["virial"]).max() <= 1e-10 * max(1.0, np.abs(a["virial"]).max())
    assert np.abs(a["vatom"] - b["vatom"]).max() <= 1e-12 * max(1.0, np.abs(a["vatom"]).max())


# ---- the same independent checks on the COMMITTED tables the BASELINE configs run on ---------------------------------
# (potentials/WRe_L20.mtp: config 4; W_L16.mtp: configs 2, 3; W_L16_nbh.almtp: config 5).  The oracle is unpinned with
# respect to a running reference, so these -- definition, finite differences, strain derivative -- are what stands
# under it at exactly the table shapes that are benchmarked.

import os

import _mtpfile

_POT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "potentials")


def test_committed_level20_two_species_table_matches_einsum_definition():
    """site energies of 16 atoms (2 x 2 x 2 cells, W + Re at random) with potentials/WRe_L20.mtp: oracle
    (pair_mtp.cpp:154-212 restated) against full Cartesian tensors contracted graph by graph -- 460 scalars of up to
    ten tensors, ranks up to 8."""
    path = os.path.join(_POT, "WRe_L20.mtp")
    pot = _mtpfile.read_committed(path, 20)
    assert (pot.species_count, len(pot.table.graphs), pot.table.radial_funcs) == (2, 460, 5)
    s = _system(ncell=(2, 2, 2), species=2)
    assert s.nlocal == 16 and set(s.types[:16]) == {1, 2}
    res = Oracle(path).compute(s.x, s.types, s.ilist, s.first, s.neigh)
    E, _ = defn.site_energies(pot, pot.table.graphs, s)
    np.testing.assert_allclose(res["eatom"][: s.nlocal], E, rtol=5e-11, atol=1e-11)
    assert abs(res["energy"] - E.sum()) < 1e-10 * max(1.0, abs(E.sum()))


def _fd_setup(fname, ncell=(3, 3, 3), species=1):
    path = os.path.join(_POT, fname)
    pos, box = mtpgen.bcc_lattice(*ncell)
    types = np.random.default_rng(5).integers(1, species + 1, size=len(pos)).astype(np.int32)
    return path, pos, box, types


def test_committed_level16_forces_are_central_differences_of_the_energy():
    """F = -dE/dx (pair_mtp.cpp:196-254: products, reverse mode, scatter onto i and j) at level 16 on W_L16.mtp"""
    path, pos, box, types = _fd_setup("W_L16.mtp")
    o = Oracle(path)

    def energy(p):
        s = periodic_system(p, box, types, 7.0)
        return o.compute(s.x, s.types, s.ilist, s.first, s.neigh), s

    r, s = energy(pos)
    F = s.fold_forces(r["f"])
    assert np.abs(F.sum(0)).max() < 1e-10 * max(1.0, np.abs(F).max())
    assert abs(r["energy"] - r["eatom"][: s.nlocal].sum()) < 1e-10 * max(1.0, abs(r["energy"]))
    h = 1e-5
    for (a, c) in [(0, 0), (11, 1), (29, 2), (53, 0), (40, 1)]:
        pp = pos.copy(); pp[a, c] += h
        pm = pos.copy(); pm[a, c] -= h
        fd = -(energy(pp)[0]["energy"] - energy(pm)[0]["energy"]) / (2 * h)
        assert abs(fd - F[a, c]) < 2e-7 * max(1.0, np.abs(F).max()), (a, c, fd, F[a, c])


def test_committed_level20_two_species_forces_are_central_differences():
    path, pos, box, types = _fd_setup("WRe_L20.mtp", species=2)
    o = Oracle(path)

    def energy(p):
        s = periodic_system(p, box, types, 7.0)
        return o.compute(s.x, s.types, s.ilist, s.first, s.neigh), s

    r, s = energy(pos)
    F = s.fold_forces(r["f"])
    assert np.abs(F.sum(0)).max() < 1e-10 * max(1.0, np.abs(F).max())
    h = 1e-5
    for (a, c) in [(3, 0), (17, 2), (44, 1)]:
        pp = pos.copy(); pp[a, c] += h
        pm = pos.copy(); pm[a, c] -= h
        fd = -(energy(pp)[0]["energy"] - energy(pm)[0]["energy"]) / (2 * h)
        assert abs(fd - F[a, c]) < 2e-7 * max(1.0, np.abs(F).max()), (a, c, fd, F[a, c])


def test_committed_level16_virial_is_the_strain_derivative():
    """isotropic AND shear strain: dE/d(eps_ab) = -virial_ab (LAMMPS sign, virial_ab = sum r_a f_b; pair_mtp.cpp:257-277
    symmetrises the off-diagonals), and sum_i vatom_i = virial"""
    path, pos, box, types = _fd_setup("W_L16.mtp")
    o = Oracle(path)
    s0 = periodic_system(pos, box, types, 7.0)
    r0 = o.compute(s0.x, s0.types, s0.ilist, s0.first, s0.neigh)

    def energy_strained(eps):
        # the ghost images are images of the UNSTRAINED cell mapped through the same strain (a sheared box is not
        # orthogonal, so the images are strained rather than regenerated); the list is index-based
        x = s0.x @ (np.eye(3) + eps).T
        return o.compute(x, s0.types, s0.ilist, s0.first, s0.neigh)["energy"]

    h = 1e-6
    comps = [((0, 0), 0), ((1, 1), 1), ((2, 2), 2), ((0, 1), 3), ((0, 2), 4), ((1, 2), 5)]
    for (a, b), v in comps:
        e = np.zeros((3, 3))
        e[a, b] = e[b, a] = h if a != b else h
        # symmetric strain: for a != b both eps_ab and eps_ba are switched on, dE = -(v_ab + v_ba) h = -2 v_ab h
        dE = (energy_strained(e) - energy_strained(-e)) / (2 * h)
        want = -r0["virial"][v] * (2.0 if a != b else 1.0)
        assert abs(dE - want) < 2e-6 * max(1.0, abs(want)), ((a, b), dE, want)
    np.testing.assert_allclose(r0["vatom"][: s0.nlocal].sum(0), r0["virial"], rtol=1e-12, atol=1e-11)


def test_committed_level16_nbh_candidate_vector_by_finite_differences(tmp_path):
    """config 5's file: c = dE_i/dtheta (pair_mtp_extrapolation.cpp:193-198, 235-252, 323-329).  Linear block = the
    basis values from the definition; species block = indicator; radial block = central differences of the site energy
    with respect to single radial coefficients (rewritten files); grade = max |A^-1 c| with the file's own inverse."""
    path = os.path.join(_POT, "W_L16_nbh.almtp")
    pot = _mtpfile.read_committed(path, 16)
    assert pot.mvs_mode == "nbh" and pot.inverse_active_set.shape == (149, 149)
    s = _system(ncell=(2, 2, 2))
    o = Oracle(path, selection=True)
    ext = o.compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    plain = Oracle(os.path.join(_POT, "W_L16.mtp")).compute(s.x, s.types, s.ilist, s.first, s.neigh)
    np.testing.assert_array_equal(ext["f"], plain["f"])          # the same coefficients: W_L16.mtp + the selection state
    _, Bv = defn.site_energies(pot, pot.table.graphs, s)
    Mu = pot.table.radial_funcs
    nrad = Mu * 8
    assert nrad + 1 + Bv.shape[1] == 149
    for i in [0, 7, 12]:
        row = (s.ilist[i:i + 1], np.array([0, s.first[i + 1] - s.first[i]], dtype=np.int32), s.neigh[s.first[i]:s.first[i + 1]])
        c = o.compute(s.x, s.types, *row, extrapolation=True)["coeff_ders"]
        np.testing.assert_allclose(c[nrad + 1:], Bv[i], rtol=1e-10, atol=1e-11)
        assert c[nrad] == 1.0
        assert abs(np.abs(pot.inverse_active_set @ c).max() - ext["grades"][i]) < 1e-12 * max(1.0, ext["grades"][i])
        for flat in [0, 5, 8 + 3, 2 * 8 + 7, 3 * 8 + 1, nrad - 1]:          # every radial function mu is touched
            h = 1e-6
            es = []
            for sign in (+1, -1):
                p2 = mtpgen.Potential(pot.table, 1, pot.min_dist, pot.max_dist, 8, pot.scaling, pot.radial_coeffs.copy(),
                                      pot.species_coeffs, pot.moment_coeffs)
                p2.radial_coeffs.reshape(-1)[flat] += sign * h
                pth = str(tmp_path / "pert.mtp")
                mtpgen.write_mtp(p2, pth)
                es.append(Oracle(pth).compute(s.x, s.types, *row)["eatom"][s.ilist[i]])
            fd = (es[0] - es[1]) / (2 * h)
            assert abs(fd - c[flat]) < 1e-6 * max(1.0, abs(c[flat])), (i, flat, fd, c[flat])
    assert ext["max_grade"] == ext["grades"][: s.nlocal].max()
