"""Host-side stand-ins for the LAMMPS core pieces either side of the pair style
(Neighbor, Comm ghosts) so the hot path can be exercised
without LAMMPS: periodic ghost images, a *full* neighbour list (the reference requests
REQ_FULL, /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:318) in CSR form, and the
reverse-communication fold of ghost forces onto their owners (newton_pair on,
pair_mtp.cpp:252-254, 315).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class System:
    x: np.ndarray          # [nall, 3] owned atoms first, then ghosts
    types: np.ndarray      # [nall] int32, 1-based (LAMMPS)
    nlocal: int
    owner: np.ndarray      # [nall] index of the owned atom each entry images (identity for owned)
    box: np.ndarray        # [3]
    ilist: np.ndarray      # [nlocal] int32
    first: np.ndarray      # [nlocal+1] int32 CSR offsets
    neigh: np.ndarray      # [first[-1]] int32 neighbour indices into x
    cutoff: float

    @property
    def nall(self):
        return self.x.shape[0]

    def fold_forces(self, f):
        """Reverse communication: add ghost forces to their owners; returns [nlocal,3]."""
        out = np.zeros((self.nlocal, 3))
        np.add.at(out, self.owner, f)
        return out


def make_ghosts(pos, box, rghost, lo=None):
    """Periodic images within `rghost` of the box [lo, lo+box) (orthogonal cell)."""
    pos = np.asarray(pos, dtype=np.float64)
    box = np.asarray(box, dtype=np.float64)
    lo = np.zeros(3) if lo is None else np.asarray(lo, dtype=np.float64)
    n = pos.shape[0]
    nimg = np.ceil(rghost / box).astype(int)
    xs = [pos]
    owners = [np.arange(n)]
    for sx in range(-nimg[0], nimg[0] + 1):
        for sy in range(-nimg[1], nimg[1] + 1):
            for sz in range(-nimg[2], nimg[2] + 1):
                if sx == sy == sz == 0:
                    continue
                sh = np.array([sx, sy, sz]) * box
                p = pos + sh
                m = np.all((p >= lo - rghost) & (p < lo + box + rghost), axis=1)
                if m.any():
                    xs.append(p[m])
                    owners.append(np.nonzero(m)[0])
    return np.concatenate(xs), np.concatenate(owners)


def full_neighbor_list(x, nlocal, cutoff, chunk=65536):
    """CSR full list over the first nlocal atoms: every j != i with |x_j - x_i| <= cutoff.  Rows are queried in
    chunks, so the Python lists of a 500k-atom system never exist all at once."""
    from scipy.spatial import cKDTree

    tree = cKDTree(x)
    counts = np.zeros(nlocal, dtype=np.int64)
    parts = []
    for c0 in range(0, nlocal, chunk):
        c1 = min(nlocal, c0 + chunk)
        lists = tree.query_ball_point(x[c0:c1], cutoff, workers=-1, return_sorted=True)
        lens = np.fromiter((len(l) for l in lists), dtype=np.int64, count=c1 - c0)
        flat = np.fromiter((j for l in lists for j in l), dtype=np.int32, count=int(lens.sum()))
        rows = np.repeat(np.arange(c0, c1, dtype=np.int64), lens)
        keep = flat != rows                                   # the atom itself
        parts.append(flat[keep])
        counts[c0:c1] = lens - 1
    first = np.zeros(nlocal + 1, dtype=np.int64)
    np.cumsum(counts, out=first[1:])
    neigh = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int32)
    assert first[-1] < 2 ** 31 and len(neigh) == first[-1]
    return first.astype(np.int32), neigh


def periodic_system(pos, box, types=None, list_cutoff=7.0):
    """Owned atoms + ghost images + full neighbour list with cutoff `list_cutoff`
    (LAMMPS builds the list with cutoff + skin; 5 A + 2 A metal-units skin = 7 A in the
    BASELINE configs)."""
    pos = np.asarray(pos, dtype=np.float64)
    n = pos.shape[0]
    if types is None:
        types = np.ones(n, dtype=np.int32)
    x, owner = make_ghosts(pos, box, list_cutoff)
    first, neigh = full_neighbor_list(x, n, list_cutoff)
    return System(x=x, types=np.asarray(types, dtype=np.int32)[owner], nlocal=n, owner=owner,
                  box=np.asarray(box, dtype=np.float64), ilist=np.arange(n, dtype=np.int32),
                  first=first, neigh=neigh, cutoff=float(list_cutoff))
