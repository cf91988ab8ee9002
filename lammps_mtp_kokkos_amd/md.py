"""Device-resident velocity-Verlet (NVE) driver around the MTP force call -- the standalone counterpart of what
LAMMPS' Verlet / Comm / Neighbor classes do around `Pair::compute`: positions,
velocities, forces, ghosts and the neighbour list all stay in HBM between steps.

  per step      ghosts <- owners + periodic shift (index_select), force call (mtp_compute_device), ghost forces
                folded onto their owners (index_add_: newton_pair on, /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:252-254,
                315), two half kicks and a drift (metal units)
  every `every` steps (or when an atom moved more than half the skin, checked on the device)
                ghost set rebuilt from the wrapped positions, full list rebuilt on the GPU
                (mtp_build_neighbors_device)

torch is plumbing here (index maps, axpy); the force call and the list build are the library's HIP kernels.
"""
from __future__ import annotations

import numpy as np

from .driver import make_ghosts

MVV2E = 1.0364269e-4          # (g/mol)(A/ps)^2 -> eV     (LAMMPS metal units)
FTM2V = 1.0 / MVV2E           # eV/A / (g/mol) -> A/ps^2


class DeviceNVE:
    def __init__(self, ctx, pos, box, rc, types=None, mass=183.84, list_cutoff=7.0, device=None, every=10):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = device or torch.device("cuda:0")
        # torch's default stream is the null stream, which the library maps to its own non-blocking stream: use one
        # real stream for the index maps, the axpys and the library's kernels so that they are ordered
        if torch.cuda.current_stream(self.dev).cuda_stream == 0:
            from .capi import use_private_torch_stream
            use_private_torch_stream(self.dev)
        self.box_np = np.asarray(box, dtype=np.float64)
        self.box = torch.from_numpy(self.box_np).to(self.dev)
        self.n = len(pos)
        self.types_np = np.ones(self.n, dtype=np.int32) if types is None else np.asarray(types, dtype=np.int32)
        self.mass = float(mass)
        self.cut = float(list_cutoff)
        self.rc = float(rc)                # potential cutoff: skin = list_cutoff - rc
        self.every = int(every)
        self.x = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float64)).to(self.dev)   # owned, unwrapped
        self.v = torch.zeros_like(self.x)
        self.ev = torch.zeros(8, dtype=torch.float64, device=self.dev)
        self.steps_since_build = 0
        self.builds = 0
        self.energy = 0.0
        self._reneighbor()
        self._forces()

    # ---- ghosts + list (re-neighbouring) ------------------------------------------------------------
    def _reneighbor(self):
        torch = self.torch
        pos = self.x.cpu().numpy()
        wrapped = pos - np.floor(pos / self.box_np) * self.box_np
        xall, owner = make_ghosts(wrapped, self.box_np, self.cut)
        self.nall = len(xall)
        self.owner = torch.from_numpy(owner.astype(np.int64)).to(self.dev)
        # ghost k sits at x[owner] + shift; owned atoms get their wrap shift, so xall = x[owner] + shift exactly
        shift = xall - pos[owner]
        self.shift = torch.from_numpy(shift).to(self.dev)
        self.types_all = torch.from_numpy(self.types_np[owner]).to(self.dev)
        self.xall = torch.empty((self.nall, 3), dtype=torch.float64, device=self.dev)
        self.fall = torch.zeros((self.nall, 3), dtype=torch.float64, device=self.dev)
        self._update_ghosts()
        lo = -self.cut - 1.0
        hi = self.box_np + self.cut + 1.0
        self.entries, self.max_row = self.ctx.build_neighbors_device(self.xall, self.n, self.nall, self.cut,
                                                                     [lo, lo, lo], hi,
                                                                     stream=torch.cuda.current_stream().cuda_stream)
        self.x_at_build = self.x.clone()
        self.steps_since_build = 0
        self.builds += 1

    def _update_ghosts(self):
        self.torch.index_select(self.x, 0, self.owner, out=self.xall)
        self.xall += self.shift

    def _forces(self):
        torch = self.torch
        self._update_ghosts()
        self.fall.zero_()
        self.ev.zero_()
        self.ctx.compute_device(self.xall, self.types_all, self.fall, eflag=1, vflag=0, ev_t=self.ev,
                                stream=torch.cuda.current_stream().cuda_stream)
        self.f = torch.zeros((self.n, 3), dtype=torch.float64, device=self.dev)
        self.f.index_add_(0, self.owner, self.fall)          # reverse communication

    # ---- one velocity-Verlet step ---------------------------------------------------------------------
    def step(self, dt):
        k = 0.5 * dt * FTM2V / self.mass
        self.v.add_(self.f, alpha=k)
        self.x.add_(self.v, alpha=dt)
        self.steps_since_build += 1
        need = self.steps_since_build >= self.every
        if not need and self.steps_since_build % 4 == 0:   # half-skin criterion, on the device
            d2 = ((self.x - self.x_at_build) ** 2).sum(1).max()
            need = bool(d2 > (0.5 * (self.cut - self.rc)) ** 2)
        if need:
            self._reneighbor()
        self._forces()
        self.v.add_(self.f, alpha=k)

    def total_energy(self):
        ke = 0.5 * MVV2E * self.mass * float((self.v ** 2).sum().item())
        return float(self.ev[0].item()) + ke
