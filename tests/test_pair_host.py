"""The C++ host mirror of the reference's Pair interface (lammps_mtp_kokkos_amd/host), driven
through the call sequence LAMMPS uses: settings -> coeff -> init_style -> init_one -> compute,
plus extract / extract_peratom / pvector for the extrapolation style."""
import os
import subprocess

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")
EXE = os.path.join(ROOT, "tests", "cpp", "test_pair_host")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lammps_mtp_kokkos_amd", "host")])


def _write_system(path, s):
    with open(path, "w") as fh:
        fh.write("%d %d %.17g %.17g %.17g\n" % (s.nlocal, s.nall, *s.box))
        for (x, y, z), t in zip(s.x, s.types):
            fh.write("%.17g %.17g %.17g %d\n" % (x, y, z, t))
        for i in range(s.nlocal):
            row = s.neigh[s.first[i]:s.first[i + 1]]
            fh.write("%d %s\n" % (len(row), " ".join(map(str, row))))


def test_argument_grammar_and_errors():
    _build()
    out = subprocess.run([EXE, "args", os.path.join(POT, "W_L8.mtp"), os.path.join(POT, "W_L16_nbh.almtp")],
                         capture_output=True, text=True, cwd=str(ROOT))
    assert out.returncode == 0 and "ARGS OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("cfg_mode", [0, 1])
def test_cfg_writer_across_ranks_is_byte_identical_to_one_rank(tmp_path, cfg_mode):
    """pair_mtp_extrapolation.cpp:401-479: offsets by MPI_Scan, atom lines gathered on rank 0 in rank order.  The file
    a 2- or 3-rank job writes (uneven shards, one empty) equals the 1-rank file, which equals the format spelled out
    here line by line."""
    _build()
    pos, box = mtpgen.bcc_lattice(2, 2, 3)
    rng = np.random.default_rng(8)
    types = rng.integers(1, 3, len(pos)).astype(np.int32)
    s = periodic_system(pos, box, types, 4.0)
    sysf = str(tmp_path / "sys.txt")
    _write_system(sysf, s)
    r = subprocess.run([EXE, "cfg", sysf, str(tmp_path), str(cfg_mode)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    one = open(tmp_path / "cfg_1.cfg", "rb").read()
    assert one == open(tmp_path / "cfg_2.cfg", "rb").read()
    assert one == open(tmp_path / "cfg_3.cfg", "rb").read()
    n = s.nlocal
    grades = [0.37 * i + 1.0 / (i + 3.0) for i in range(n)]
    want = ["BEGIN_CFG", "Size", "%d" % n, "Supercell", "%.6f %.6f %.6f" % (s.box[0], 0, 0),
            "%.6f %.6f %.6f" % (0.25, s.box[1], 0), "%.6f %.6f %.6f" % (-0.5, 0.125, s.box[2])]
    if cfg_mode:
        want.append("AtomData:  id type       cartes_x      cartes_y      cartes_z")
        want += ["%d\t%d\t%.6f\t%.6f\t%.6f" % (i + 1, s.types[i] - 1, *s.x[i]) for i in range(n)]
    else:
        want.append("AtomData:  id type       cartes_x      cartes_y      cartes_z       nbh_grades")
        want += ["%d\t%d\t%.6f\t%.6f\t%.6f\t%.5f" % (i + 1, s.types[i] - 1, *s.x[i], grades[i]) for i in range(n)]
    want += ["Feature   MV_grade\t%.6f" % 3.14159265, "END_CFG", "", ""]
    assert one.decode() == "\n".join(want)
    # the three utils::logmesg lines of the reference (pair_mtp.cpp:383, 389; pair_mtp_extrapolation.cpp:508-517)
    assert r.stdout == ("The scaling is : 1.00e+00.\nThere are 2 species.\n"
                        "Extrapolation Scheme: Neighborhood mode, with a selection threshold of 2 and break threshold of 10.5.\n"
                        "Extrapolation Mode: Configuration mode.\n")


@pytest.mark.gpu
@pytest.mark.parametrize("style,extra", [("mtp", []), ("mtp/kk", ["chunksize", "32768"]),
                                         ("mtp/small/kk", ["chunksize", "4096"])])
def test_pair_styles_match_oracle(tmp_path, style, extra):
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(4, 4, 4)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16.mtp")
    r = subprocess.run([EXE, "run", style, sysf, outf, potf] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(outf).read().split("\n")
    e, cut = map(float, lines[0].split())
    vir = np.array(lines[1].split(), float)
    arr = np.array([l.split() for l in lines[2:2 + s.nall]], float)
    want = Oracle(potf).compute(s.x, s.types, s.ilist, s.first, s.neigh)
    assert cut == 5.0
    assert abs(e - want["energy"]) < 1e-9
    assert np.abs(arr[:, :3] - want["f"]).max() < 1e-9
    assert np.abs(arr[:, 3] - want["eatom"]).max() < 1e-10
    assert np.abs(vir - want["virial"]).max() < 1e-8


@pytest.mark.gpu
def test_extrapolation_style_fix_pair_protocol_and_cfg_file(tmp_path):
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(3, 3, 3)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf, cfgf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt"), str(tmp_path / "sel.cfg")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16_nbh.almtp")
    want = Oracle(potf, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    mg = want["max_grade"]
    # select threshold below the max grade -> configuration written; break threshold above -> keeps running
    r = subprocess.run([EXE, "runext", "mtp/extrapolation", sysf, outf, potf, cfgf, "%.6f" % (0.5 * mg), "%.6f" % (2 * mg)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(outf).read().split("\n")
    e, e_plain, pv, stopped = lines[0].split()
    assert abs(float(e) - want["energy"]) < 1e-9 and abs(float(e_plain) - want["energy"]) < 1e-9
    assert abs(float(pv) - mg) < 1e-9 * max(1, mg) and stopped == "0"
    g = np.array(lines[1:1 + s.nlocal], float)
    assert np.abs(g - want["grades"][: s.nlocal]).max() < 1e-9 * max(1, mg)
    cfg = open(cfgf).read()
    # MLIP-3 style grades every call: two compute calls -> two records (pair_mtp_extrapolation.cpp:71, 341)
    assert cfg.count("BEGIN_CFG") == 2 and cfg.count("END_CFG") == 2
    rec = cfg.split("END_CFG")[0].split("\n")
    assert rec[0] == "BEGIN_CFG" and rec[1] == "Size" and int(rec[2]) == s.nlocal and rec[3] == "Supercell"
    assert rec[4] == "%.6f %.6f %.6f" % (s.box[0], 0, 0)
    assert rec[7].startswith("AtomData:  id type       cartes_x") and rec[7].endswith("nbh_grades")
    first = rec[8].split("\t")
    assert first[0] == "1" and first[1] == "0" and first[5] == "%.5f" % want["grades"][0]
    assert any(l.startswith("Feature   MV_grade\t%.6f" % mg) for l in rec)
    # break threshold below the max grade -> the run is terminated with the reference's message
    r = subprocess.run([EXE, "runext", "mtp/extrapolation", sysf, outf, potf, cfgf, "%.6f" % (0.5 * mg), "%.6f" % (0.9 * mg)],
                       capture_output=True, text=True)
    assert "Exceeded Break Threshold" in r.stdout + r.stderr


# ---- the /kk styles' data path: x, f, type and the padded 2-D neighbour view resident on the device --------------------

@pytest.mark.gpu
@pytest.mark.parametrize("style,extra", [("mtp/kk", ["chunksize", "32768"]), ("mtp/small/kk", ["chunksize", "4096"])])
def test_kk_styles_device_resident_step_with_fdotr_vflag(tmp_path, style, extra):
    """DeviceAtomView + DeviceNeighListView (what LAMMPS-KOKKOS hands the reference's /kk styles,
    KOKKOS/pair_mtp_kokkos.cpp:231-240): list compacted on the device from the LayoutLeft view, one stream for the whole
    step, eatom / vatom copied back because the step asked, and the global virial tallied although vflag = VIRIAL_FDOTR
    | VIRIAL_ATOM (the pair style tallies on the raw flag, pair_mtp.cpp:257)."""
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(4, 4, 4)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16.mtp")
    r = subprocess.run([EXE, "rundev", style, sysf, outf, potf] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(outf).read().split("\n")
    e, cut = map(float, lines[0].split())
    vir = np.array(lines[1].split(), float)
    arr = np.array([l.split() for l in lines[2:2 + s.nall]], float)
    want = Oracle(potf).compute(s.x, s.types, s.ilist, s.first, s.neigh)
    assert cut == 5.0 and abs(e - want["energy"]) < 1e-9
    assert np.abs(arr[:, :3] - want["f"]).max() < 1e-9
    assert np.abs(arr[:, 3] - want["eatom"]).max() < 1e-10
    assert np.abs(vir).max() > 1e-3 and np.abs(vir - want["virial"]).max() < 1e-8        # not dropped under VIRIAL_FDOTR


@pytest.mark.gpu
def test_extrapolation_kk_style_keeps_grades_on_the_device_until_asked(tmp_path):
    """mtp/extrapolation/kk through the device-resident path: grade steps run on the device views too (the reference
    copies grades to the host only when asked, KOKKOS/pair_mtp_extrapolation_kokkos.cpp:223-243): pvector[0], the
    grades behind extract_peratom and the .cfg record (positions copied only for the record) against the oracle."""
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(3, 3, 3)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf, cfgf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt"), str(tmp_path / "sel.cfg")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16_nbh.almtp")
    want = Oracle(potf, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    mg = want["max_grade"]
    r = subprocess.run([EXE, "runextdev", "mtp/extrapolation/kk", sysf, outf, potf, cfgf, "%.6f" % (0.5 * mg),
                        "%.6f" % (2 * mg), "chunksize", "1024"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(outf).read().split("\n")
    e, e_plain, pv, stopped = lines[0].split()
    assert abs(float(e) - want["energy"]) < 1e-9 and abs(float(e_plain) - want["energy"]) < 1e-9
    assert abs(float(pv) - mg) < 1e-9 * max(1, mg) and stopped == "0"
    g = np.array(lines[1:1 + s.nlocal], float)
    assert np.abs(g - want["grades"][: s.nlocal]).max() < 1e-9 * max(1, mg)
    cfg = open(cfgf).read()
    assert cfg.count("BEGIN_CFG") == 2 and cfg.count("END_CFG") == 2
    rec = cfg.split("END_CFG")[0].split("\n")
    first = rec[8].split("\t")
    assert first[0] == "1" and first[1] == "0" and first[5] == "%.5f" % want["grades"][0]
    assert first[2:5] == ["%.6f" % v for v in s.x[0]]
    # plain 1-argument form driven by `fix pair` (extrapolation_flag through extract): no file, grades on request
    r = subprocess.run([EXE, "runextdev", "mtp/extrapolation/small/kk", sysf, outf, potf], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(outf).read().split("\n")
    assert abs(float(lines[0].split()[2]) - mg) < 1e-9 * max(1, mg)
    g = np.array(lines[1:1 + s.nlocal], float)
    assert np.abs(g - want["grades"][: s.nlocal]).max() < 1e-9 * max(1, mg)


# ---- the LAMMPS plugin adapter, compiled against the mock of the LAMMPS API (tests/cpp/lammps_mock) ------------------

PLUGIN_EXE = os.path.join(ROOT, "tests", "cpp", "test_plugin_mock")


def test_plugin_adapter_compiles_and_registers_the_six_reference_styles():
    """lammpsplugin_init -> six `pair` registrations with the reference's style names (pair_mtp.h:18-21,
    pair_mtp_extrapolation.h:18-21, KOKKOS/pair_mtp*_kokkos.h:18-23).  The mock is test scaffolding: this pins that the
    adapter compiles and registers, not compatibility with a LAMMPS binary."""
    _build()
    r = subprocess.run([PLUGIN_EXE, "list"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    names = r.stdout.split("\n")[:6]
    assert sorted(names) == sorted(["mtp", "mtp/kk", "mtp/small/kk", "mtp/extrapolation", "mtp/extrapolation/kk",
                                    "mtp/extrapolation/small/kk"])


@pytest.mark.gpu
@pytest.mark.parametrize("style,extra", [("mtp", []), ("mtp/kk", ["chunksize", "32768"])])
def test_plugin_adapter_walks_the_lammps_call_sequence(tmp_path, style, extra):
    """creator -> settings -> coeff -> init_style (REQ_FULL request) -> init_one -> compute(ENERGY_GLOBAL | ENERGY_ATOM,
    VIRIAL_FDOTR | VIRIAL_ATOM) through the adapter: forces, eng_vdwl, eatom, virial (tallied although LAMMPS asked for
    FDOTR: no_virial_fdotr_compute is set, pair_mtp.cpp:257) and vatom against the oracle."""
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(4, 4, 4)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16.mtp")
    r = subprocess.run([PLUGIN_EXE, "run", style, sysf, outf, potf] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "The scaling is : 1.00e+00." in r.stdout and "There are 1 species." in r.stdout      # utils::logmesg lines
    lines = open(outf).read().split("\n")
    e, cut, flags = lines[0].split()
    assert float(cut) == 5.0 and flags == "11100"     # no_virial_fdotr_compute, manybody, one_coeff set; single, restart off
    vir = np.array(lines[1].split(), float)
    arr = np.array([l.split() for l in lines[2:2 + s.nall]], float)
    want = Oracle(potf).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    assert abs(float(e) - want["energy"]) < 1e-9
    assert np.abs(arr[:, :3] - want["f"]).max() < 1e-9
    assert np.abs(arr[:, 3] - want["eatom"]).max() < 1e-10
    assert np.abs(arr[:, 4] - want["vatom"][:, 0]).max() < 1e-9
    assert np.abs(vir - want["virial"]).max() < 1e-8 and np.abs(vir).max() > 1e-3


@pytest.mark.gpu
def test_plugin_adapter_extrapolation_style_fix_pair_protocol(tmp_path):
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(3, 3, 3)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16_nbh.almtp")
    want = Oracle(potf, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    r = subprocess.run([PLUGIN_EXE, "run", "mtp/extrapolation", sysf, outf, potf], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Extrapolation Mode: Neighborhood mode." in r.stdout
    lines = open(outf).read().split("\n")
    e, e_first, pv, nextra = lines[0].split()
    mg = want["max_grade"]
    assert abs(float(e) - want["energy"]) < 1e-9 and abs(float(e_first) - want["energy"]) < 1e-9 and nextra == "1"
    assert abs(float(pv) - mg) < 1e-9 * max(1, mg)
    g = np.array(lines[1:1 + s.nlocal], float)
    assert np.abs(g - want["grades"][: s.nlocal]).max() < 1e-9 * max(1, mg)


PLUGIN_KK_EXE = os.path.join(ROOT, "tests", "cpp", "test_plugin_mock_kk")


@pytest.mark.gpu
@pytest.mark.parametrize("style,layout,extra", [("mtp/kk", "left", ["chunksize", "32768"]),
                                                ("mtp/kk", "right", ["chunksize", "4096"]),
                                                ("mtp/small/kk", "left", ["chunksize", "32768"])])
def test_plugin_adapter_kokkos_branch_runs_on_device_resident_atoms_and_list(tmp_path, style, layout, extra):
    """The adapter's LMP_KOKKOS branch, compiled against the mock of the KOKKOS package's data classes
    (tests/cpp/lammps_mock/kokkos_mock.h) and run on device memory: x / f / type come from the dual views' device side
    (the host copy of x holds NaN), the list from NeighListKokkos' padded d_neighbors(i, jj) in either Kokkos layout with
    LAMMPS' special bits in the ids, everything on the execution space's stream (KOKKOS/pair_mtp_kokkos.cpp:231-240);
    forces land in the device f, eng_vdwl / virial / eatom / vatom in the host arrays LAMMPS reads.  Also the data-manager
    protocol: sync(Device, X|F|TYPE) before the force call, modified(Device, F) after it."""
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(4, 4, 4)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16.mtp")
    r = subprocess.run([PLUGIN_KK_EXE, "runkk", style, sysf, outf, layout, potf] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(outf).read().split("\n")
    e, cut, flags = lines[0].split()
    assert float(cut) == 5.0 and flags == "11100"
    vir = np.array(lines[1].split(), float)
    arr = np.array([l.split() for l in lines[2:2 + s.nall]], float)
    want = Oracle(potf).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=4)
    assert abs(float(e) - want["energy"]) < 1e-9
    assert np.abs(arr[:, :3] - want["f"]).max() < 1e-9
    assert np.abs(arr[:, 3] - want["eatom"]).max() < 1e-10
    assert np.abs(arr[:, 4] - want["vatom"][:, 0]).max() < 1e-9
    assert np.abs(vir - want["virial"]).max() < 1e-8 and np.abs(vir).max() > 1e-3
    tag, synced, modified, sync_call, modified_call = lines[2 + s.nall].split()
    assert tag == "kokkos" and int(synced) == (0x1 | 0x4 | 0x10) and int(modified) == 0x4      # X | F | TYPE, then F
    assert 0 <= int(sync_call) < int(modified_call)


@pytest.mark.gpu
def test_plugin_adapter_kokkos_branch_extrapolation_style(tmp_path):
    """mtp/extrapolation/kk through the KOKKOS branch: grades stay on the device until extract_peratom asks for them
    (KOKKOS/pair_mtp_extrapolation_kokkos.cpp:223-243)."""
    from oracle.pyoracle import Oracle
    _build()
    pos, box = mtpgen.bcc_lattice(3, 3, 3)
    s = periodic_system(pos, box, None, 7.0)
    sysf, outf = str(tmp_path / "sys.txt"), str(tmp_path / "out.txt")
    _write_system(sysf, s)
    potf = os.path.join(POT, "W_L16_nbh.almtp")
    want = Oracle(potf, selection=True).compute(s.x, s.types, s.ilist, s.first, s.neigh, extrapolation=True)
    r = subprocess.run([PLUGIN_KK_EXE, "runkk", "mtp/extrapolation/kk", sysf, outf, "left", potf, "chunksize", "32768"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(outf).read().split("\n")
    e, e_first, pv, nextra = lines[0].split()
    mg = want["max_grade"]
    assert abs(float(e) - want["energy"]) < 1e-9 and abs(float(e_first) - want["energy"]) < 1e-9 and nextra == "1"
    assert abs(float(pv) - mg) < 1e-9 * max(1, mg)
    g = np.array(lines[1:1 + s.nlocal], float)
    assert np.abs(g - want["grades"][: s.nlocal]).max() < 1e-9 * max(1, mg)
