"""ASan + UBSan run of the host-only code of the product library: the MLIP-3 parser and the
native schedule builder on every committed potential, and on truncated / corrupted variants that must end in an
error code, never in a memory error (the sanitizers abort the process on the first finding)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")
EXE = os.path.join(ROOT, "tests", "cpp", "test_parser_san")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lammps_mtp_kokkos_amd", "host"), "san"])
    return EXE


def _run(exe, path, sel):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, path, str(int(sel))], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout.strip()


@pytest.mark.parametrize("fname,sel", [("W_L8.mtp", 0), ("W_L16.mtp", 0), ("W_L16_nbh.almtp", 1), ("WRe_L20.mtp", 0),
                                       ("WRe_L10_cfg.almtp", 1)])
def test_committed_potentials_parse_clean_under_sanitizers(exe, fname, sel):
    import json
    out = _run(exe, os.path.join(POT, fname), sel)
    assert out.startswith("OK "), out
    sizes = json.load(open(os.path.join(POT, "SIZES.json")))[fname]
    b, t, s, a = map(int, out.split()[1:5])
    assert (b, t, s, a) == (sizes["B"], sizes["T"], sizes["S"], sizes["A"])
    # the gather programs of the product passes reproduce the reference's sequential products and adjoints
    assert float(out.split()[8]) < 1e-12, out


@pytest.mark.parametrize("fname", ["W_L16.mtp", "WRe_L10_cfg.almtp"])
def test_schedule_corner_cases_follow_the_in_order_semantics(exe, tmp_path, fname):
    """A late writer of a factor of never-read scalars (their rows may then not be deferred as leaf rows) and two
    coefficients on one never-read scalar: the driver replays both forms of the native schedule, leaf rows included,
    against the file-order loops of pair_mtp.cpp:196-233."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _mutate import mutate_mtp
    src = os.path.join(POT, fname)
    text_only = str(tmp_path / "plain.mtp")
    data = open(src, "rb").read()
    open(text_only, "wb").write(data[: data.index(b"#MVS_v1.1")] if b"#MVS_v1.1" in data else data)
    dst = str(tmp_path / "mutated.mtp")
    info = mutate_mtp(text_only, dst)
    assert info["leaves"] > 0
    base, out = _run(exe, text_only, 0), _run(exe, dst, 0)
    assert base.startswith("OK ") and out.startswith("OK "), (base, out)
    assert float(out.split()[8]) < 1e-12, out
    # the late writer costs leaves their special treatment: the checksum (which includes the stored-moment count) moves
    assert out.split()[7] != base.split()[7]


def test_truncated_and_corrupted_files_fail_cleanly(exe, tmp_path):
    data = open(os.path.join(POT, "W_L16_nbh.almtp"), "rb").read()
    text_end = data.index(b"#MVS_v1.1")
    cases = {}
    for frac in (0.05, 0.2, 0.5, 0.8, 0.97):                       # text cut anywhere
        cases["cut%.2f" % frac] = data[: int(text_end * frac)]
    cases["short_binary"] = data[: text_end + 200]                  # selection state cut inside the matrices
    cases["no_binary"] = data[: data.index(b"#", text_end + 1) + 1]
    cases["huge_counts"] = data.replace(b"alpha_index_times_count = ", b"alpha_index_times_count = 9", 1)
    cases["negative_index"] = data.replace(b"alpha_index_basic = {{0, 0, 0, 0}", b"alpha_index_basic = {{0, -3, 0, 0}", 1)
    cases["moment_out_of_range"] = data.replace(b"alpha_moment_mapping = {0,", b"alpha_moment_mapping = {99999,", 1)
    cases["garbage"] = bytes(range(256)) * 40
    for name, blob in cases.items():
        p = tmp_path / (name + ".almtp")
        p.write_bytes(blob)
        out = _run(exe, str(p), 1)
        assert out.startswith("ERR ") or out.startswith("OK "), (name, out)
        if name in ("short_binary", "no_binary", "garbage", "cut0.50", "negative_index", "moment_out_of_range"):
            assert out.startswith("ERR "), (name, out)
