#!/usr/bin/env python3
"""Open-ended randomised GPU-vs-oracle parity sweep (the fixed-seed cases of tests/test_gpu_fuzz.py run the same
function under `pytest -m gpu`).
  python scripts/fuzz_parity.py [cases] [seed]"""
import os
import sys
import tempfile

import numpy as np

os.environ.setdefault("MTP_BANK_ROUNDS", "2")   # (search effort of the LDS-bank numbering: a load-time / speed knob only)
os.environ.setdefault("MTP_BANK_SCALE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests._fuzz import fuzz_case  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
tmp = tempfile.mkdtemp()
worst = 0.0
for case in range(ncase):
    desc, err = fuzz_case(rng, tmp, "p%d" % case)
    worst = max(worst, *err.values())
    print("case %2d %s  rel err F %.1e E %.1e V %.1e G %.1e" % (case, desc, err["F"], err["E"], err["V"], err["G"]),
          flush=True)
print("fuzz_parity: worst relative error %.2e" % worst)
sys.exit(0 if worst < 1e-9 else 1)
