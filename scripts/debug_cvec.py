#!/usr/bin/env python3
"""Diagnostic: configuration-mode candidate vector, fused force-kernel path vs mtp_cvec_kernel path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system

path = sys.argv[1] if len(sys.argv) > 1 else "potentials/WRe_L10_cfg.almtp"
pos, box = mtpgen.bcc_lattice(3, 3, 3)
rng = np.random.default_rng(1)
pos = pos + rng.normal(0, 0.05, pos.shape)
types = rng.integers(1, 3, len(pos)).astype(np.int32)
pot = capi.Potential(path, selection=True)
info = pot.info
if info.species_count == 1:
    types[:] = 1
s = periodic_system(pos, box, types, 7.0)
out = {}
for tag, env in (("fused", None), ("general", "1")):
    if env:
        os.environ["MTP_GRADE_UNFUSED"] = env
    else:
        os.environ.pop("MTP_GRADE_UNFUSED", None)
    ctx = capi.Context(pot, 0)
    ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
    r = ctx.compute(s.x, s.types, grade=True)
    out[tag] = r["coeff_ders"] if info.configuration_mode else r["grades"]
a, b = out["fused"], out["general"]
print("Sp", info.species_count, "Mu", info.radial_func_count, "R", info.radial_basis_size, "len", len(a))
d = np.abs(a - b)
print("max diff", d.max(), "at", int(d.argmax()))
nrad = info.species_count ** 2 * info.radial_func_count * info.radial_basis_size
if info.configuration_mode:
    np.set_printoptions(linewidth=200, precision=4)
    print("radial block fused  :", a[:nrad][:40])
    print("radial block general:", b[:nrad][:40])
    print("ratio:", (a[:nrad] / np.where(b[:nrad] == 0, 1, b[:nrad]))[:40])
