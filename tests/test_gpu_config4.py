"""BASELINE.json config 4 at its real shape, on the one GPU of the box: 524,288-atom W-Re (10 % Re), level-20
two-species potential, decomposed 2 x 2 x 2 -- every rank's ghosts come from SEVEN different peers (3 face, 3 edge,
1 corner partner; no self images).

The eight halo objects of the decomposition live on device 0 (created without a communicator); each rank's shard runs
in turn through the library's own pack kernel, mtp_halo_local_exchange (device copies addressed with the per-peer
send / receive offset tables the RCCL group uses, csrc/mtp_halo.hip `exchange`), mtp_compute_device with its own
context and neighbour list, the reverse exchange and the library's unpack kernel.  EVERY owned-atom force and site
energy of the 524,288 atoms is then compared with the threaded CPU oracle run on the single periodic domain
(/root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:72-280; forces on ghosts folded onto their owners: newton_pair on,
pair_mtp.cpp:248-254, 315).  Tolerance as everywhere: |dF| <= 1e-9 eV/A + 1e-10 |F|max.
"""
import os

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.domain import decompose_all
from lammps_mtp_kokkos_amd.driver import periodic_system

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")


def _threads():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 32))


def _run_decomposed(path, pos, box, types, world, grade=False):
    """All `world` shards on device 0.  Returns (F [N,3], eatom [N], energy, virial[6], plans, ghost-row check)."""
    import torch
    dev = torch.device("cuda:0")
    stream = capi.use_private_torch_stream(dev)
    st = stream.cuda_stream
    plans = decompose_all(pos, box, types, world, 7.0)
    pot = capi.Potential(path)
    halos, ctxs, xs, fs, tys, eas, evs = [], [], [], [], [], [], []
    for p in plans:
        halos.append(capi.Halo(p, 0, None))                     # no communicator: pack / unpack / layout only
        lay = halos[-1].layout()
        assert lay["nsend"] == len(p.send_idx) and list(lay["send_counts"]) == list(p.send_counts)
        ctx = capi.Context(pot, 0)
        il, fi, ne = (torch.from_numpy(a).to(dev) for a in (p.ilist, p.first, p.neigh))
        ctx.set_neighbors_device(il, fi, ne, p.nall, int(np.diff(p.first).max()))
        ctxs.append(ctx)
        x = np.concatenate([p.x0[: p.nlocal], np.full((p.nghost, 3), np.nan)])    # ghosts unknown until the halo lands
        xs.append(torch.from_numpy(x).to(dev))
        fs.append(torch.zeros((p.nall, 3), dtype=torch.float64, device=dev))
        tys.append(torch.from_numpy(p.types).to(dev))
        eas.append(torch.zeros(p.nall, dtype=torch.float64, device=dev))
        evs.append(torch.zeros(8, dtype=torch.float64, device=dev))
    # forward: pack on every rank, then every (source, destination) segment
    for h, x in zip(halos, xs):
        h.pack_forward(x, st)
    capi.Halo.local_exchange(halos, 0, xs, st)
    stream.synchronize()
    for p, x in zip(plans, xs):
        got = x.cpu().numpy()
        assert np.array_equal(got[p.nlocal:], p.x0[p.nlocal:]), "ghost positions of rank %d" % p.rank     # copies + one add
    for p, c, x, ty, f, ea, ev in zip(plans, ctxs, xs, tys, fs, eas, evs):
        c.compute_device(x, ty, f, eflag=3, vflag=1, eatom_t=ea, ev_t=ev, stream=st)
    for c in ctxs:
        c.synchronize(st)
    # reverse: ghost forces back onto their owners through the owners' receive buffers
    capi.Halo.local_exchange(halos, 1, fs, st)
    for h, f in zip(halos, fs):
        h.unpack_reverse(f, st)
    stream.synchronize()
    n = len(pos)
    F = np.full((n, 3), np.nan)
    E = np.full(n, np.nan)
    energy, virial = 0.0, np.zeros(6)
    for p, f, ea, ev in zip(plans, fs, eas, evs):
        F[p.owned_global] = f.cpu().numpy()[: p.nlocal]
        E[p.owned_global] = ea.cpu().numpy()[: p.nlocal]
        e = ev.cpu().numpy()
        energy += e[0]
        virial += e[1:7]
    assert not np.isnan(F).any() and not np.isnan(E).any()        # every atom is owned by exactly one rank
    return F, E, energy, virial, plans


def _check(F, E, energy, virial, s, want):
    F_ref = s.fold_forces(want["f"])
    scale = max(1.0, float(np.abs(F_ref).max()))
    err = float(np.abs(F - F_ref).max())
    assert err <= 1e-9 + 1e-10 * scale, "forces: max abs err %.3e (scale %.3e)" % (err, scale)
    assert float(np.abs(E - want["eatom"][: s.nlocal]).max()) <= 1e-10
    n = s.nlocal
    assert abs(energy - want["energy"]) / n <= 1e-10 * max(1.0, abs(want["energy"]) / n)
    assert float(np.abs(virial - want["virial"]).max()) <= 1e-6 + 1e-10 * float(np.abs(want["virial"]).max())
    return err


def _system(ncell, frac=0.10, seed=4242):
    pos, box = mtpgen.bcc_lattice(*ncell)
    types = (np.random.default_rng(seed).random(len(pos)) < frac).astype(np.int32) + 1     # W + 10 % Re
    return pos, box, types


def test_eight_shards_with_seven_distinct_peers_small():
    """the same machinery on 16^3 cells (8,192 atoms; sub-boxes of 25.3 A > 2 x 7 A), serial oracle"""
    from oracle.pyoracle import Oracle
    pos, box, types = _system((16, 16, 16), frac=0.3)
    path = os.path.join(POT, "WRe_L20.mtp")
    F, E, energy, virial, plans = _run_decomposed(path, pos, box, types, 8)
    for p in plans:
        assert p.send_counts[p.rank] == 0 and p.recv_counts[p.rank] == 0            # no self images in a 2x2x2 grid
        assert all(c > 0 for q, c in enumerate(p.recv_counts) if q != p.rank)       # ghosts from all seven peers
    s = periodic_system(pos, box, types, 7.0)
    want = Oracle(path).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=1)
    _check(F, E, energy, virial, s, want)


def test_config4_524288_atoms_2x2x2_every_force_against_the_threaded_oracle():
    from oracle.pyoracle import Oracle
    pos, box, types = _system((64, 64, 64))
    assert len(pos) == 524288
    path = os.path.join(POT, "WRe_L20.mtp")
    F, E, energy, virial, plans = _run_decomposed(path, pos, box, types, 8)
    assert sum(p.nlocal for p in plans) == 524288 and all(abs(p.nlocal - 65536) < 656 for p in plans)
    for p in plans:
        assert p.send_counts[p.rank] == 0 and p.recv_counts[p.rank] == 0
        assert all(c > 0 for q, c in enumerate(p.recv_counts) if q != p.rank)
    s = periodic_system(pos, box, types, 7.0)
    want = Oracle(path).compute_mt(_threads(), s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=1)
    err = _check(F, E, energy, virial, s, want)
    print("config 4, 524,288 atoms over 8 shards: max |dF| = %.3e eV/A" % err)
