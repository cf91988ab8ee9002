"""Device-resident velocity-Verlet (NVE) driver around the MTP force call -- the standalone counterpart of what
LAMMPS' Verlet / Comm / Neighbor classes do around `Pair::compute`.  Positions,
velocities, forces, ghost maps and the neighbour list stay in HBM between steps; every piece of the step is one of
the library's HIP kernels (include/mtp_mi355x.h, "standalone MD support"):

  per step      mtp_nve_initial (kick + drift) -> mtp_ghosts_forward (ghosts follow their owners) -> force call
                (mtp_compute_device) -> mtp_ghosts_reverse (ghost forces onto owners: newton_pair on,
                /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:252-254, 315) -> mtp_nve_final (kick)
  every `every` steps (or when an atom moved more than half the skin: mtp_nve_monitor, one 16-byte read-back)
                mtp_ghosts_build (wrap + periodic images, on the device) and mtp_build_neighbors_device

torch only allocates the arrays and provides the stream; the host sees the ghost count and the list size at a
re-neighbouring (they size arrays) and nothing else.
"""
from __future__ import annotations

import numpy as np

from . import capi

MVV2E = 1.0364269e-4          # (g/mol)(A/ps)^2 -> eV     (LAMMPS metal units)
FTM2V = 1.0 / MVV2E           # eV/A / (g/mol) -> A/ps^2


class DeviceNVE:
    def __init__(self, ctx, pos, box, rc, types=None, mass=183.84, list_cutoff=7.0, device=None, every=10,
                 check_every=4, vflag=0):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = device or torch.device("cuda:0")
        # torch's default stream is the null stream, which the library maps to its own non-blocking stream: use one
        # real stream for the allocations' fills and the library's kernels so that they are ordered
        if torch.cuda.current_stream(self.dev).cuda_stream == 0:
            capi.use_private_torch_stream(self.dev)
        self.st = torch.cuda.current_stream(self.dev).cuda_stream
        self.box_np = np.asarray(box, dtype=np.float64)
        self.n = len(pos)
        types_np = np.ones(self.n, dtype=np.int32) if types is None else np.asarray(types, dtype=np.int32)
        masses = np.atleast_1d(np.asarray(mass, dtype=np.float64))
        if len(masses) < int(types_np.max()):
            masses = np.full(int(types_np.max()), float(masses[0]))
        self.mass_t = torch.from_numpy(masses).to(self.dev)
        self.inv_mass_t = torch.from_numpy(1.0 / masses).to(self.dev)
        self.cut = float(list_cutoff)
        self.rc = float(rc)                # potential cutoff: skin = list_cutoff - rc
        self.every = int(every)
        self.check_every = int(check_every)
        self.vflag = int(vflag)          # 1: the global virial is tallied too (as in the headline benchmark)
        self.ghosts = capi.Ghosts(self.dev.index or 0)
        self.cap = 0
        self._alloc(int(self.n * 1.6) + 1024, pos, types_np)
        self.v = torch.zeros((self.n, 3), dtype=torch.float64, device=self.dev)
        self.mon = torch.zeros(2, dtype=torch.float64, device=self.dev)
        self.x_ref = torch.empty((self.n, 3), dtype=torch.float64, device=self.dev)
        self.steps_since_build = 0
        self.builds = 0
        self._reneighbor()
        self._forces()

    @property
    def x(self):
        """owned atoms (wrapped into the box at the last re-neighbouring)"""
        return self.xall[: self.n]

    @property
    def f(self):
        return self.fall[: self.n]

    def _alloc(self, cap, pos=None, types_np=None):
        torch = self.torch
        xall = torch.zeros((cap, 3), dtype=torch.float64, device=self.dev)
        tall = torch.ones(cap, dtype=torch.int32, device=self.dev)
        if pos is not None:
            xall[: self.n] = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float64)).to(self.dev)
            tall[: self.n] = torch.from_numpy(types_np).to(self.dev)
        else:
            xall[: self.n] = self.xall[: self.n]
            tall[: self.n] = self.types_all[: self.n]
        self.xall, self.types_all = xall, tall
        # forces and the energy / virial totals share one allocation, so that one launch zeroes both every step
        self.fbuf = torch.zeros(3 * cap + 8, dtype=torch.float64, device=self.dev)
        self.fall = self.fbuf[: 3 * cap].view(cap, 3)
        self.ev = self.fbuf[3 * cap:]
        self.cap = cap

    # ---- ghosts + list (re-neighbouring), all on the device ----------------------------------------------------
    def _reneighbor(self):
        try:
            self.nall = self.ghosts.build(self.xall, self.n, self.box_np, self.cut, stream=self.st)
        except capi.MtpError as e:
            if e.code != -24 or self.ghosts.nall <= self.cap:
                raise
            self._alloc(int(self.ghosts.nall * 1.2) + 1024)
            self.nall = self.ghosts.build(self.xall, self.n, self.box_np, self.cut, stream=self.st)
        self.ghosts.types(self.types_all, stream=self.st)
        lo = [-self.cut - 1.0] * 3
        hi = self.box_np + self.cut + 1.0
        self.entries, self.max_row = self.ctx.build_neighbors_device(self.xall, self.n, self.nall, self.cut, lo, hi,
                                                                     stream=self.st)
        self.x_ref.copy_(self.xall[: self.n])
        self.steps_since_build = 0
        self.builds += 1

    def _forces(self):
        self.ghosts.forward(self.xall, stream=self.st)
        capi.zero_async(self.fbuf, stream=self.st)      # f (all rows of the allocation) and ev in one launch
        # (finish_tallies=False: the energy / virial fold rides in the launch that folds the ghost forces)
        self.ctx.compute_device_rows(0, self.n, False, self.xall, self.types_all, self.fall, eflag=1, vflag=self.vflag,
                                     ev_t=self.ev, stream=self.st)
        self.ghosts.reverse_finish(self.ctx, self.fall, self.ev, eflag=1, vflag=self.vflag, stream=self.st)

    # ---- one velocity-Verlet step ------------------------------------------------------------------------------
    def step(self, dt):
        dtf = 0.5 * dt * FTM2V
        capi.nve_initial(self.n, self.xall, self.v, self.fall, self.types_all, self.inv_mass_t, dtf, dt, stream=self.st)
        self.steps_since_build += 1
        need = self.steps_since_build >= self.every
        if not need and self.check_every and self.steps_since_build % self.check_every == 0:   # half-skin criterion
            capi.nve_monitor(self.n, self.xall, self.x_ref, self.v, self.types_all, self.mass_t, self.mon, stream=self.st)
            need = bool(self.mon[0].item() > (0.5 * (self.cut - self.rc)) ** 2)
        if need:
            self._reneighbor()
        self._forces()
        capi.nve_final(self.n, self.v, self.fall, self.types_all, self.inv_mass_t, dtf, stream=self.st)

    def total_energy(self):
        capi.nve_monitor(self.n, self.xall, self.x_ref, self.v, self.types_all, self.mass_t, self.mon, stream=self.st)
        m = self.mon.cpu().numpy()
        return float(self.ev[0].item()) + 0.5 * MVV2E * float(m[1])
