"""Domain decomposition + halo exchange on CPU: world_size-2 and -4 `gloo` runs whose
sharded forces/energies must equal the single-domain result (the oracle does the
per-shard force call here; on the GPU box the HIP path does)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lammps_mtp_kokkos_amd import mtpgen
from lammps_mtp_kokkos_amd.domain import HaloExchange, decompose, rank_grid
from lammps_mtp_kokkos_amd.driver import periodic_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POTF = os.path.join(ROOT, "potentials", "WRe_L10_cfg.almtp")


def _global_system():
    pos, box = mtpgen.bcc_lattice(3, 3, 4, seed=31)
    types = np.random.default_rng(2).integers(1, 3, size=len(pos)).astype(np.int32)
    return pos, box, types


def _worker(rank, world, port, out):
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos, box, types = _global_system()
        plan = decompose(pos, box, types, world, rank, 7.0)
        halo = HaloExchange(plan, torch.device("cpu"))
        x = torch.from_numpy(plan.x0.copy())
        x[plan.nlocal:] = 0.0                      # ghosts must come from the forward halo
        halo.forward(x)
        assert np.abs(x.numpy() - plan.x0).max() < 1e-12
        o = Oracle(POTF)
        r = o.compute(x.numpy(), plan.types, plan.ilist, plan.first, plan.neigh, eflag=3, vflag=1)
        f = torch.from_numpy(r["f"].copy())
        halo.reverse(f)
        ev = torch.tensor([r["energy"]] + list(r["virial"]))
        dist.all_reduce(ev)
        gathered = [None] * world
        dist.all_gather_object(gathered, (plan.owned_global, f[: plan.nlocal].numpy(), r["eatom"][: plan.nlocal]))
        if rank == 0:
            n = len(pos)
            F = np.zeros((n, 3))
            E = np.zeros(n)
            seen = np.zeros(n, int)
            for ids, ff, ee in gathered:
                F[ids] = ff
                E[ids] = ee
                seen[ids] += 1
            assert (seen == 1).all()
            np.savez(out, F=F, E=E, ev=ev.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_equals_single_domain(tmp_path, world):
    from oracle.pyoracle import Oracle
    out = str(tmp_path / "res.npz")
    port = 29600 + world + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    pos, box, types = _global_system()
    pos = pos - np.floor(pos / box) * box
    s = periodic_system(pos, box, types, 7.0)
    r = Oracle(POTF).compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=3, vflag=1)
    F = s.fold_forces(r["f"])
    np.testing.assert_allclose(got["F"], F, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(got["E"], r["eatom"][: s.nlocal], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(got["ev"][0], r["energy"], rtol=1e-11)
    np.testing.assert_allclose(got["ev"][1:], r["virial"], rtol=1e-9, atol=1e-9)


def test_plan_bookkeeping():
    pos, box, types = _global_system()
    for world in (1, 2, 4, 8):
        plans = [decompose(pos, box, types, world, r, 7.0, with_lists=False) for r in range(world)]
        assert sum(p.nlocal for p in plans) == len(pos)
        assert np.prod(rank_grid(world)) == world
        for r, p in enumerate(plans):
            assert sum(p.recv_counts) == p.nghost and sum(p.send_counts) == len(p.send_idx)
            for q, pq in enumerate(plans):
                assert p.send_counts[q] == pq.recv_counts[r]
            assert (p.send_idx >= 0).all() and (p.send_idx < p.nlocal).all()


def test_overlap_order_partitions_rows_into_interior_boundary_interior():
    from lammps_mtp_kokkos_amd.domain import overlap_order
    pos, box = mtpgen.bcc_lattice(8, 8, 8)
    plan = decompose(pos, box, None, 2, 1, 7.0)
    for cap in (3072, 40):
        ilist, first, neigh, (na, nb, nc) = overlap_order(plan, cap)
        assert na + nb + nc == plan.nlocal and sorted(ilist.tolist()) == list(range(plan.nlocal))
        assert na == nc and 0 < na <= cap
        for r, i in enumerate(ilist):
            row = neigh[first[r]:first[r + 1]]
            want = plan.neigh[plan.first[i]:plan.first[i + 1]]
            assert np.array_equal(row, want)
            has_ghost = bool((row >= plan.nlocal).any())
            if r < na or r >= na + nb:
                assert not has_ghost          # the overlapped ranges never read a ghost position


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c_side_halo_tables_equal_the_plan(world):
    """mtp_halo_layout (host only; the tables mtp_halo_create hands to ncclSend / ncclRecv, csrc/mtp_halo.hip
    `exchange`) against domain.py's plan: offsets are the running sums of the counts, the two sides of every
    (source, destination) segment agree on its length, and the atoms a rank sends to a peer are exactly the ghosts the
    peer expects from it, in the peer's storage order."""
    from lammps_mtp_kokkos_amd import capi
    from lammps_mtp_kokkos_amd.domain import decompose_all
    pos, box = mtpgen.bcc_lattice(10, 10, 10)
    types = (np.random.default_rng(5).random(len(pos)) < 0.2).astype(np.int32) + 1
    plans = decompose_all(pos, box, types, world, 7.0, with_lists=False)
    lay = [capi.halo_layout(p) for p in plans]
    for p, (so, ro) in zip(plans, lay):
        assert list(so) == [0] + list(np.cumsum(p.send_counts)) and list(ro) == [0] + list(np.cumsum(p.recv_counts))
        assert so[-1] == len(p.send_idx) and ro[-1] == p.nghost
    for r, pr in enumerate(plans):
        for q, pq in enumerate(plans):
            assert pr.recv_counts[q] == pq.send_counts[r]
            sent = pq.owned_global[pq.send_idx[lay[q][0][r]: lay[q][0][r + 1]]]          # what q packs for r
            expected = pr.ghost_global[lay[r][1][q]: lay[r][1][q + 1]]                   # r's ghosts owned by q
            assert np.array_equal(sent, expected)
            shifted = pq.x0[pq.send_idx[lay[q][0][r]: lay[q][0][r + 1]]] + pq.send_shift[lay[q][0][r]: lay[q][0][r + 1]]
            assert np.array_equal(shifted, pr.x0[pr.nlocal + lay[r][1][q]: pr.nlocal + lay[r][1][q + 1]])


def test_c_side_halo_layout_rejects_bad_plans():
    from lammps_mtp_kokkos_amd import capi
    from lammps_mtp_kokkos_amd.domain import decompose
    pos, box = mtpgen.bcc_lattice(8, 8, 8)
    p = decompose(pos, box, None, 2, 0, 7.0, with_lists=False)
    p.recv_counts = [p.recv_counts[0] + 1, p.recv_counts[1]]
    with pytest.raises(capi.MtpError) as e:
        capi.halo_layout(p)
    assert e.value.code == -20 and "add up" in str(e.value)
    p = decompose(pos, box, None, 2, 0, 7.0, with_lists=False)
    p.send_idx = p.send_idx.copy()
    p.send_idx[0] = p.nlocal
    with pytest.raises(capi.MtpError) as e:
        capi.halo_layout(p)
    assert "outside the owned atoms" in str(e.value)


def test_shipped_library_carries_no_experiment_switches():
    from lammps_mtp_kokkos_amd import capi
    assert capi.build_flags() == ""
