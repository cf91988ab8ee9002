"""The library's RCCL halo (mtp_halo_*, csrc/mtp_halo.hip) on the one GPU of the box: a single rank whose peers
are its own periodic images, so every ghost travels through ncclSend / ncclRecv (to and from rank 0) inside one
group per direction -- the same code path N ranks take, minus the other ranks.  Checked against the index
arithmetic of the plan (forward), against the single-shot force call + host fold (reverse), and against the
oracle; the overlapped step uses mtp_compute_device_rows on the interior / boundary ordering."""
import os

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.domain import decompose, overlap_order

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT = os.path.join(ROOT, "potentials")


@pytest.fixture(scope="module")
def rank0():
    import torch
    pos, box = mtpgen.bcc_lattice(8, 8, 8)          # 1,024 atoms, box 25.3 A > 2 x 7 A
    plan = decompose(pos, box, None, 1, 0, 7.0)
    assert plan.nghost > 0 and plan.send_counts == [plan.nghost] and plan.recv_counts == [plan.nghost]
    dev = torch.device("cuda:0")
    stream = capi.use_private_torch_stream(dev)
    halo = capi.Halo(plan, 0, capi.halo_unique_id())
    return plan, halo, dev, stream


def test_rccl_reports_one_rank(rank0):
    plan, halo, dev, stream = rank0
    cc = halo.comm_count()
    assert cc["nranks"] == 1 and cc["rank"] == 0 and cc["rccl_version"] >= 20000


def test_forward_halo_moves_ghosts_with_their_owners(rank0):
    import torch
    plan, halo, dev, stream = rank0
    rng = np.random.default_rng(3)
    x_own = plan.x0[: plan.nlocal] + rng.normal(0, 0.05, (plan.nlocal, 3))
    x = torch.from_numpy(np.concatenate([x_own, np.full((plan.nghost, 3), np.nan)])).to(dev)
    halo.forward(x, stream.cuda_stream)
    stream.synchronize()
    want = x_own[plan.send_idx] + plan.send_shift          # ghosts are stored in the order they are sent (one peer)
    got = x.cpu().numpy()
    assert np.array_equal(got[plan.nlocal:], want)          # copies and one add each: bit-exact
    assert np.array_equal(got[: plan.nlocal], x_own)


def test_reverse_halo_folds_ghost_forces_onto_owners(rank0):
    import torch
    plan, halo, dev, stream = rank0
    rng = np.random.default_rng(4)
    f_np = rng.normal(size=(plan.nall, 3))
    f = torch.from_numpy(f_np.copy()).to(dev)
    halo.reverse(f, stream.cuda_stream)
    stream.synchronize()
    want = f_np[: plan.nlocal].copy()
    np.add.at(want, plan.send_idx, f_np[plan.nlocal:])
    got = f.cpu().numpy()
    np.testing.assert_allclose(got[: plan.nlocal], want, rtol=0, atol=1e-13)      # atomic order only
    assert np.array_equal(got[plan.nlocal:], f_np[plan.nlocal:])                     # ghost rows are left as they were


def test_allreduce_sum_and_max_single_rank(rank0):
    import torch
    plan, halo, dev, stream = rank0
    a = torch.arange(7, dtype=torch.float64, device=dev) - 3.0
    b = a.clone()
    halo.allreduce(b, capi.REDUCE_SUM, stream.cuda_stream)
    halo.allreduce(b, capi.REDUCE_MAX, stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("potential,grade", [("W_L16.mtp", False), ("W_L16_nbh.almtp", True)])
def test_decomposed_step_matches_single_shot_call_and_oracle(rank0, potential, grade, overlap):
    """mtp_halo_force_step through the RCCL self-exchange, both schedules: one stream (pack, forward exchange, all rows,
    reverse exchange, unpack -- the default) and overlapped (forward halo || interior rows, boundary rows, reverse halo
    || interior rows)."""
    import torch
    from oracle.pyoracle import Oracle
    plan, halo, dev, stream = rank0
    halo.set_overlap(overlap)
    assert halo.overlap == overlap
    path = os.path.join(POT, potential)
    pot = capi.Potential(path, selection=grade)
    ctx = capi.Context(pot, 0)
    ilist, first, neigh, (na, nb, nc) = overlap_order(plan)
    assert na > 0 and nb > 0 and nc > 0 and na + nb + nc == plan.nlocal
    il, fi, ne = (torch.from_numpy(a).to(dev) for a in (ilist, first, neigh))
    ctx.set_neighbors_device(il, fi, ne, plan.nall, int(np.diff(first).max()))
    rng = np.random.default_rng(5)
    x_own = plan.x0[: plan.nlocal] + rng.normal(0, 0.03, (plan.nlocal, 3))
    x = torch.from_numpy(np.concatenate([x_own, plan.x0[plan.nlocal:]])).to(dev)     # ghosts stale until the halo lands
    ty = torch.from_numpy(plan.types).to(dev)
    f = torch.zeros((plan.nall, 3), dtype=torch.float64, device=dev)
    ev = torch.zeros(8, dtype=torch.float64, device=dev)
    ea = torch.zeros(plan.nall, dtype=torch.float64, device=dev)
    gr = torch.zeros(plan.nall, dtype=torch.float64, device=dev) if grade else None
    mg = torch.zeros(1, dtype=torch.float64, device=dev) if grade else None
    st = stream.cuda_stream
    kw = dict(eflag=3, vflag=1, grade=grade, eatom_t=ea, grades_t=gr, maxg_t=mg, stream=st)
    for _ in range(2):                                   # twice: events and buffers are reused across steps
        f.zero_()
        ev.zero_()
        if grade:
            mg.zero_()
        halo.forward_begin(x, st)
        ctx.compute_device_rows(0, na, False, x, ty, f, **kw)
        halo.forward_end(st)
        ctx.compute_device_rows(na, nb, False, x, ty, f, **kw)
        halo.reverse_begin(f, st)
        ctx.compute_device_rows(na + nb, nc, True, x, ty, f, ev_t=ev, **kw)
        halo.reverse_end(f, st)
    ctx.synchronize(st)
    f_steps = f.clone()
    ev_steps = ev.clone()
    # ... and the same step through the single entry point (mtp_halo_force_step)
    ev.zero_()
    if grade:
        mg.zero_()
    halo.force_step(ctx, (na, nb, nc), x, ty, f, ev_t=ev, **kw)
    ctx.synchronize(st)
    assert (f - f_steps).abs().max().item() < 1e-11 and (ev - ev_steps).abs().max().item() < 1e-8
    x_all = np.concatenate([x_own, x_own[plan.send_idx] + plan.send_shift])
    o = Oracle(path, selection=grade)
    want = o.compute(x_all, plan.types, plan.ilist, plan.first, plan.neigh, eflag=3, vflag=1, extrapolation=grade,
                     natoms=plan.nlocal)
    F_ref = want["f"][: plan.nlocal].copy()
    np.add.at(F_ref, plan.send_idx, want["f"][plan.nlocal:])
    got_f = f.cpu().numpy()
    scale = max(1.0, np.abs(F_ref).max())
    assert np.abs(got_f[: plan.nlocal] - F_ref).max() <= 1e-9 + 1e-10 * scale
    evh = ev.cpu().numpy()
    assert abs(evh[0] - want["energy"]) <= 1e-10 * plan.nlocal * max(1.0, abs(want["energy"]) / plan.nlocal)
    assert np.abs(evh[1:7] - want["virial"]).max() <= 1e-8 + 1e-10 * np.abs(want["virial"]).max()
    assert np.abs(ea.cpu().numpy() - want["eatom"]).max() <= 1e-10
    if grade:
        g = gr.cpu().numpy()
        assert np.abs(g[: plan.nlocal] - want["grades"][: plan.nlocal]).max() <= 1e-9 * max(1.0, want["max_grade"])
        assert abs(float(mg.item()) - want["max_grade"]) <= 1e-9 * max(1.0, want["max_grade"])
