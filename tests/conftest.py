import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library spends up to ~10 s per potential load on its LDS-bank search for the product rows (csrc/mtp_potential.cpp;
# a performance matter only: every numbering gives the same results).  The suites load hundreds of potentials, so they run
# with two rounds of it unless the caller set the knobs; tests/test_gpu_parity.py::test_production_bank_search_effort runs
# the shipped default.
os.environ.setdefault("MTP_BANK_ROUNDS", "2")
os.environ.setdefault("MTP_BANK_SCALE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_code_is_built():
    """The suites need libmtp_mi355x.so (hipcc cross-compiles it without a GPU) and the oracle; build
    them once if a fresh checkout has not run __graft_entry__.build() yet."""
    lib = os.path.join(ROOT, "lammps_mtp_kokkos_amd", "libmtp_mi355x.so")
    host = os.path.join(ROOT, "lammps_mtp_kokkos_amd", "libpair_mtp_mi355x.so")
    orc = os.path.join(ROOT, "oracle", "libmtp_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(host) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def tmp_pot_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("pots")
