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
