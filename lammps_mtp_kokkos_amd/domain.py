"""Spatial domain decomposition with ghost-atom halo exchange -- the native counterpart of
what LAMMPS `Comm` does around the pair style: atoms are sharded
by sub-box, each rank holds the ghosts within the list cutoff of its box, and every
force call is bracketed by a forward halo (ghost positions) and a reverse halo (ghost
forces summed into their owners: `newton_pair on`,
/root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:252-254, 315).

One process per GPU.  This module holds the *plan* (who owns what, which ghosts, index
lists) and a torch twin of the exchange, `HaloExchange`, used by the CPU tests (`gloo`) and
the several-ranks-on-one-GPU rehearsals.  On MI355X nodes the exchange itself is the
library's: `capi.Halo` -> `mtp_halo_*` (csrc/mtp_halo.hip: device pack / unpack kernels and
ONE grouped RCCL send/recv per direction, so on a fully connected xGMI node all 7 links carry
traffic concurrently and there is a single latency stage instead of LAMMPS' x -> y -> z
staging).  Energies/virials stay rank-local sums until the caller all-reduces them, as
LAMMPS does at thermo output.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .driver import full_neighbor_list

GRIDS = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2), 3: (1, 1, 3), 6: (1, 2, 3)}


def rank_grid(nranks):
    if nranks in GRIDS:
        return GRIDS[nranks]
    return (1, 1, nranks)


@dataclass
class HaloPlan:
    """Everything one rank needs; index arrays are numpy on creation (to_torch moves them)."""
    rank: int
    nranks: int
    grid: tuple
    nlocal: int
    nghost: int
    owned_global: np.ndarray            # [nlocal] global atom ids
    x0: np.ndarray                      # [nall,3] initial coordinates (owned then ghosts)
    types: np.ndarray                   # [nall] 1-based
    ilist: np.ndarray
    first: np.ndarray
    neigh: np.ndarray
    # forward halo: for each peer q (in rank order) the owned atoms q needs and their shifts
    send_idx: np.ndarray                # [nsend] local indices, grouped by destination rank
    send_shift: np.ndarray              # [nsend,3] periodic shift to add (in the receiver's frame)
    send_counts: list                   # per destination rank
    recv_counts: list                   # per source rank (sum = nghost); ghosts are stored in this order
    ghost_global: np.ndarray = field(default=None)   # [nghost] global id of each ghost (tests)

    @property
    def nall(self):
        return self.nlocal + self.nghost


def _owner_of(pos, box, grid):
    g = np.asarray(grid)
    cell = np.floor(pos / box * g).astype(int)
    cell = np.minimum(np.maximum(cell, 0), g - 1)
    return (cell[:, 0] * g[1] + cell[:, 1]) * g[2] + cell[:, 2]


def _ghost_images(pos, box, lo, hi, rghost, owner, rank):
    """All periodic images (global id, shift) lying within rghost of [lo,hi) that are not
    this rank's own unshifted atoms.  Deterministic order: by owner rank, then shift, then id."""
    nimg = np.ceil(rghost / box).astype(int)
    ids, shifts = [], []
    for sx in range(-nimg[0], nimg[0] + 1):
        for sy in range(-nimg[1], nimg[1] + 1):
            for sz in range(-nimg[2], nimg[2] + 1):
                sh = np.array([sx, sy, sz], dtype=np.float64) * box
                p = pos + sh
                m = np.all((p >= lo - rghost) & (p < hi + rghost), axis=1)
                if sx == sy == sz == 0:
                    m &= owner != rank
                k = np.nonzero(m)[0]
                if len(k):
                    ids.append(k)
                    shifts.append(np.broadcast_to(np.array([sx, sy, sz]), (len(k), 3)))
    if not ids:
        return np.zeros(0, dtype=np.int64), np.zeros((0, 3), dtype=np.int64)
    ids = np.concatenate(ids)
    shifts = np.concatenate(shifts)
    key = np.lexsort((ids, shifts[:, 2], shifts[:, 1], shifts[:, 0], owner[ids]))
    return ids[key], shifts[key]


def decompose(pos, box, types, nranks, rank, list_cutoff=7.0, with_lists=True):
    """Build rank `rank`'s shard of a periodic orthogonal system.  Every rank runs this on
    the same global arrays (a static decomposition, valid until atoms migrate -- the
    situation between two LAMMPS re-neighbourings)."""
    return decompose_all(pos, box, types, nranks, list_cutoff, with_lists, ranks=[rank])[0]


def decompose_all(pos, box, types, nranks, list_cutoff=7.0, with_lists=True, ranks=None):
    """The shards of `ranks` (default: all) of one decomposition; the ghost images of every sub-box are found once and
    shared by the plans (a rank's receive list is its own images, its send lists are its atoms among the peers')."""
    pos = np.asarray(pos, dtype=np.float64)
    box = np.asarray(box, dtype=np.float64)
    types = np.ones(len(pos), dtype=np.int32) if types is None else np.asarray(types, dtype=np.int32)
    grid = rank_grid(nranks)
    g = np.asarray(grid)
    pos = pos - np.floor(pos / box) * box                     # wrap into the box
    owner = _owner_of(pos, box, grid)

    def sub_box(r):
        c = np.array([r // (g[1] * g[2]), (r // g[2]) % g[1], r % g[2]])
        return c * box / g, (c + 1) * box / g

    images = []                                               # per rank q: (global ids, integer shifts) of its ghosts
    for q in range(nranks):
        qlo, qhi = sub_box(q)
        images.append(_ghost_images(pos, box, qlo, qhi, list_cutoff, owner, q))
    plans = []
    for rank in (range(nranks) if ranks is None else ranks):
        owned = np.nonzero(owner == rank)[0]
        local_of = -np.ones(len(pos), dtype=np.int64)
        local_of[owned] = np.arange(len(owned))
        gid, gsh = images[rank]
        recv_counts = [int((owner[gid] == q).sum()) for q in range(nranks)]
        x_ghost = pos[gid] + gsh * box
        x0 = np.concatenate([pos[owned], x_ghost])
        ty = np.concatenate([types[owned], types[gid]])
        # what every peer needs from me, in the order the peer stores it
        send_idx, send_shift, send_counts = [], [], []
        for q in range(nranks):
            qid, qsh = images[q]
            m = owner[qid] == rank
            send_idx.append(local_of[qid[m]])
            send_shift.append(qsh[m] * box)
            send_counts.append(int(m.sum()))
        nlocal = len(owned)
        if with_lists:
            first, neigh = full_neighbor_list(x0, nlocal, list_cutoff)
        else:
            first, neigh = np.zeros(nlocal + 1, np.int32), np.zeros(0, np.int32)
        plans.append(HaloPlan(rank=rank, nranks=nranks, grid=grid, nlocal=nlocal, nghost=len(gid), owned_global=owned,
                              x0=x0, types=ty, ilist=np.arange(nlocal, dtype=np.int32), first=first, neigh=neigh,
                              send_idx=np.concatenate(send_idx) if send_idx else np.zeros(0, np.int64),
                              send_shift=np.concatenate(send_shift) if send_shift else np.zeros((0, 3)),
                              send_counts=send_counts, recv_counts=recv_counts, ghost_global=gid))
    return plans


class HaloExchange:
    """Runtime side: device-resident index lists and the two all-to-all exchanges."""

    def __init__(self, plan: HaloPlan, device, group=None):
        import torch
        self.torch = torch
        self.plan = plan
        self.device = device
        self.group = group
        self.send_idx = torch.from_numpy(np.ascontiguousarray(plan.send_idx, dtype=np.int64)).to(device)
        self.send_shift = torch.from_numpy(np.ascontiguousarray(plan.send_shift, dtype=np.float64)).to(device)
        self.nsend = int(self.send_idx.numel())
        self.sendbuf = torch.empty((self.nsend, 3), dtype=torch.float64, device=device)
        self.recvbuf = torch.empty((plan.nghost, 3), dtype=torch.float64, device=device)
        self.frecv = torch.empty((self.nsend, 3), dtype=torch.float64, device=device)
        self.single = plan.nranks == 1

    def _a2a(self, out, inp, out_splits, in_splits):
        import torch.distributed as dist
        if self.single:
            out.copy_(inp)
        elif inp.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks on one GPU): gloo has no device all-to-all, stage through the host
            o = self.torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)

    def forward(self, x):
        """x [nall,3]: refresh the ghost rows from their owners' current positions."""
        p = self.plan
        self.torch.index_select(x[: p.nlocal], 0, self.send_idx, out=self.sendbuf)
        self.sendbuf += self.send_shift
        self._a2a(x[p.nlocal:], self.sendbuf, p.recv_counts, p.send_counts)

    def forward_begin(self, x):
        """Start the forward halo and return a handle for forward_end(); force work that needs no
        ghost positions (interior atoms) can be queued in between and overlaps the exchange."""
        import torch.distributed as dist
        p = self.plan
        self.torch.index_select(x[: p.nlocal], 0, self.send_idx, out=self.sendbuf)
        self.sendbuf += self.send_shift
        if self.single or (self.sendbuf.is_cuda and dist.get_backend(self.group) == "gloo"):
            self._a2a(x[p.nlocal:], self.sendbuf, p.recv_counts, p.send_counts)
            return None
        return dist.all_to_all_single(x[p.nlocal:], self.sendbuf, output_split_sizes=p.recv_counts,
                                      input_split_sizes=p.send_counts, group=self.group, async_op=True)

    @staticmethod
    def forward_end(handle):
        if handle is not None:
            handle.wait()      # the current stream now waits for the exchange

    def reverse(self, f):
        """f [nall,3]: add the ghost rows' forces into their owners (on the owning ranks)."""
        p = self.plan
        self._a2a(self.frecv, f[p.nlocal:], p.send_counts, p.recv_counts)
        f[: p.nlocal].index_add_(0, self.send_idx, self.frecv)


def split_interior(plan: HaloPlan):
    """Indices (into plan.ilist) of owned atoms whose list holds no ghost (interior: computable
    before the forward halo lands) and of those that do (boundary)."""
    has_ghost = np.zeros(plan.nlocal, dtype=bool)
    if plan.nlocal:
        ghost_entry = plan.neigh >= plan.nlocal
        counts = np.diff(plan.first)
        row = np.repeat(np.arange(plan.nlocal), counts)
        np.logical_or.at(has_ghost, row, ghost_entry)
    interior = np.nonzero(~has_ghost)[0]
    boundary = np.nonzero(has_ghost)[0]
    return interior, boundary


def sub_list(plan: HaloPlan, rows):
    """CSR neighbour list restricted to the given owned atoms."""
    rows = np.asarray(rows, dtype=np.int64)
    counts = (plan.first[rows + 1] - plan.first[rows]).astype(np.int64)
    first = np.zeros(len(rows) + 1, dtype=np.int32)
    np.cumsum(counts, out=first[1:])
    if len(rows):
        idx = np.concatenate([np.arange(plan.first[r], plan.first[r + 1]) for r in rows]) if len(rows) < 4096 else \
            (np.repeat(plan.first[rows].astype(np.int64) - first[:-1], counts) + np.arange(first[-1]))
        neigh = plan.neigh[idx]
    else:
        neigh = np.zeros(0, dtype=np.int32)
    return plan.ilist[rows].astype(np.int32), first, neigh.astype(np.int32)


def overlap_order(plan: HaloPlan, round_atoms=3072, align_rounds=True):
    """Neighbour list of the rank reordered for an overlapped step: rows [0, nA) and the last nC rows are interior
    atoms (no ghost in their list); the rows in between hold the boundary atoms and whatever interior atoms are left.
    The step then runs forward-halo || rows A, middle rows, reverse-halo || rows C (mtp_halo_force_step).  A and C
    only have to outlast one exchange, so they get at most `round_atoms` atoms each (one wavefront per atom: 256 CUs
    x 12 wavefronts fill the GPU once) -- every extra launch costs a partially filled last round of wavefronts.
    Returns (ilist, first, neigh, (nA, nB, nC))."""
    interior, boundary = split_interior(plan)
    R = int(round_atoms)
    # the middle launch in whole rounds of wavefronts where the interior atoms allow it (a launch of 1.5 rounds takes as
    # long as one of 2); A and C share what is left, at most one round each
    n_mid = -(-len(boundary) // R) * R
    spare = plan.nlocal - n_mid
    na = min(spare // 2, R) if (align_rounds and spare >= 1024) else min(len(interior) // 2, R)
    na = max(0, min(na, len(interior) // 2))
    nc = na
    order = np.concatenate([interior[:na], boundary, interior[na:len(interior) - nc],
                            interior[len(interior) - nc:]]).astype(np.int64)
    ilist, first, neigh = sub_list(plan, order)
    return ilist, first, neigh, (na, len(order) - na - nc, nc)
