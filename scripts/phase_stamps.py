#!/usr/bin/env python3
"""Diagnostic: per-phase share of wave cycles from the stamped build (make -C csrc stamps).
Run with MTP_LIB=lammps_mtp_kokkos_amd/ab/libmtp_mi355x_stamps.so on the GPU box."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_mtp_kokkos_amd import capi, mtpgen  # noqa: E402
from lammps_mtp_kokkos_amd.driver import periodic_system  # noqa: E402

cells = [int(c) for c in sys.argv[1].split("x")] if len(sys.argv) > 1 else [32]   # "32" or "8x8x16"
cells = cells * 3 if len(cells) == 1 else cells
potf = sys.argv[2] if len(sys.argv) > 2 else "potentials/W_L16.mtp"
variant = sys.argv[3] if len(sys.argv) > 3 else None                               # "small" | "large"
pos, box = mtpgen.bcc_lattice(*cells)
s = periodic_system(pos, box, None, 7.0)
pot = capi.Potential(potf)
ctx = capi.Context(pot, 0)
if variant:
    ctx.set_variant(dict(auto=0, large=1, small=2)[variant])
dev = torch.device("cuda:0")
il, fi, ne = (torch.from_numpy(a).to(dev) for a in (s.ilist, s.first, s.neigh))
ctx.set_neighbors_device(il, fi, ne, s.nall, int(np.diff(s.first).max()))
x = torch.from_numpy(s.x).to(dev)
ty = torch.from_numpy(s.types).to(dev)
f = torch.zeros((s.nall, 3), dtype=torch.float64, device=dev)
ev = torch.zeros(8, dtype=torch.float64, device=dev)
L = capi.lib()
buf = (C.c_ulonglong * 16)()
import time
ctx.set_timing(True)
for it in range(12):
    ctx.compute_device(x, ty, f, eflag=1, vflag=1, ev_t=ev)
    ctx.synchronize()
    L.mtp_debug_read_stamps(ctx.h, buf)
print("kernel ms (HIP events, stamped build): %.4f" % ctx.last_kernel_ms())
names = ["loop head", "compaction", "tile tables", "basic moments", "products fwd", "energy+seeds", "products bwd",
         "forces", "totals", "coef blocks"]
v = np.array(list(buf)[:10], dtype=float)
print("launch", ctx.launch_info())
for n, c in zip(names, v):
    print("%-14s %6.2f %%   %8.0f cycles/atom" % (n, 100 * c / v.sum(), c / s.nlocal))
print("sum %.0f cycles/atom/wave" % (v.sum() / s.nlocal))
nw = max(1, int(buf[12]))
print("per wavefront (%d wavefronts): prologue %.0f cycles, entry -> end of the atom loop %.0f cycles, atoms %.2f" % (
    nw, buf[10] / nw, buf[11] / nw, s.nlocal / nw))
