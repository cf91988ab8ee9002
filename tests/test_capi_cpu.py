"""CPU-side checks of the product library: it loads, exports every symbol the header
declares, parses potentials exactly like the oracle's restatement of the reference
parser, reports the reference's error cases, and refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import capi, mtpgen
from oracle.pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mtp_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mtp_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 18
    L = capi.lib()
    for n in sorted(names):
        assert hasattr(L, n), n
    assert names == set(capi.EXPORTS)


@pytest.mark.parametrize("fname,sel", [("W_L8.mtp", False), ("W_L16.mtp", False), ("W_L16_nbh.almtp", True),
                                       ("WRe_L20.mtp", False), ("WRe_L10_cfg.almtp", True)])
def test_parser_matches_oracle_parser(fname, sel):
    p = capi.Potential(os.path.join(POT, fname), selection=sel)
    o = Oracle(os.path.join(POT, fname), selection=sel)
    so, sp = o.sizes, p.sizes
    for k in ("Sp", "R", "Mu", "A", "B", "T", "S", "P", "C"):
        assert so[k] == sp[k], k
    t = p.tables()
    np.testing.assert_array_equal(t["alpha_index_basic"].ravel(), o.arr("alpha_index_basic", 4 * so["B"], np.int32))
    np.testing.assert_array_equal(t["alpha_index_times"].ravel(), o.arr("alpha_index_times", 4 * so["T"], np.int32))
    np.testing.assert_array_equal(t["alpha_moment_mapping"], o.arr("alpha_moment_mapping", so["S"], np.int32))
    np.testing.assert_array_equal(t["radial_coeffs"], o.arr("radial_basis_coeffs", t["radial_coeffs"].size))
    np.testing.assert_array_equal(t["moment_coeffs"], o.arr("linear_coeffs", so["S"]))
    np.testing.assert_array_equal(t["species_coeffs"], o.arr("species_coeffs", so["Sp"]))
    assert p.info.max_cutoff == o.m.max_cutoff and p.info.min_cutoff == o.m.min_cutoff
    assert bool(p.info.configuration_mode) == bool(o.m.configuration_mode)
    if sel:
        np.testing.assert_array_equal(t["inverse_active_set"].ravel(), o.arr("inverse_active_set", so["C"] ** 2))
    sizes = __import__("json").load(open(os.path.join(POT, "SIZES.json")))[fname]
    assert (sizes["B"], sizes["T"], sizes["S"], sizes["A"]) == (sp["B"], sp["T"], sp["S"], sp["A"])


def test_parser_error_codes(tmp_path):
    txt = open(os.path.join(POT, "W_L8.mtp")).read()
    cases = [(txt.replace("MTP\n", "XTP\n", 1), -4, "Only MTP"),
             (txt.replace("version = 1.1.0", "version = 1.1.1"), -4, "version"),
             (txt.replace("RBChebyshev", "RBShapeev"), -6, "radial basis set type"),
             (txt.replace("\tradial_coeffs", "\tmagnetic_basis_type = x"), -6, "Magnetic"),
             (txt.replace("{1, 0, 0, 0}}", "{0, 0, 0, 0}}"), -7, "Wrong number of radial"),
             (txt.replace("species_count = 1", "species_kount = 1"), -5, "Species count"),
             (txt[:txt.index("species_coeffs")], -3, "end of MTP file")]
    for bad, code, what in cases:
        p = tmp_path / "bad.mtp"
        p.write_text(bad)
        with pytest.raises(capi.MtpError, match=what) as ei:
            capi.Potential(str(p))
        assert ei.value.code == code
    with pytest.raises(capi.MtpError, match="No selection state") as ei:
        capi.Potential(os.path.join(POT, "W_L8.mtp"), selection=True)
    assert ei.value.code == -8
    with pytest.raises(capi.MtpError) as ei:
        capi.Potential(str(tmp_path / "nope.mtp"))
    assert ei.value.code == -2


def test_optional_header_lines_and_comments(tmp_path):
    """potential_name / scaling / potential_tag are optional; '#' starts a comment; the
    radial block may carry its own scaling line that the top-level value supersedes
    (pair_mtp.cpp:364-381, 398-406, 416; mtp_radial_basis.cpp:70-76)."""
    txt = open(os.path.join(POT, "W_L8.mtp")).read()
    a = txt.replace("potential_name = W_L8_synthetic\n", "scaling = 2.5\n").replace("potential_tag = \n", "")
    a = a.replace("\tmin_dist", "\tscaling = 7.0\n\tmin_dist").replace("species_count = 1", "species_count = 1 # one")
    p = tmp_path / "a.mtp"
    p.write_text(a)
    pp = capi.Potential(str(p))
    oo = Oracle(str(p))
    assert pp.info.scaling == 2.5 == oo.m.scaling
    b = txt.replace("min_dist", "min_val").replace("max_dist", "max_val")
    p.write_text(b)
    assert capi.Potential(str(p)).info.max_cutoff == 5.0


def test_level_schedule_respects_sequential_semantics():
    """Rows grouped by dependency level must reproduce the reference's in-order
    execution (pair_mtp.cpp:196-201): emulate both on random moments."""
    for fname in ("W_L8.mtp", "W_L16.mtp", "WRe_L20.mtp"):
        p = capi.Potential(os.path.join(POT, fname))
        t = p.tables()
        A, B = p.info.alpha_moment_count, p.info.alpha_index_basic_count
        rng = np.random.default_rng(1)
        m0 = np.zeros(A)
        m0[:B] = rng.uniform(-1, 1, B)
        seq = m0.copy()
        for a0, a1, mu, a3 in t["alpha_index_times"]:
            seq[a3] += mu * seq[a0] * seq[a1]
        assert 1 <= p.info.product_levels <= 12
        # the generator's tables have all rows of one target adjacent and operands complete
        done = set(range(B))
        rows = t["alpha_index_times"]
        targets = [r[3] for r in rows]
        for k, (a0, a1, mu, a3) in enumerate(rows):
            assert a0 in done and a1 in done
            if k + 1 == len(rows) or targets[k + 1] != a3:
                assert a3 not in targets[k + 1:]
                done.add(a3)
        assert np.isfinite(seq).all()


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    p = capi.Potential(os.path.join(POT, "W_L8.mtp"))
    with pytest.raises(capi.MtpError) as ei:
        capi.Context(p, 0)
    assert ei.value.code == -21
