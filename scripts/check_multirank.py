#!/usr/bin/env python3
"""Rehearsal of the N > 1 path on whatever GPUs are visible (several ranks may share one GPU with
MTP_BENCH_BACKEND=gloo): shard a periodic crystal, forward halo, HIP force call per shard, reverse
halo, gather; rank 0 compares with the single-domain HIP result.

  MTP_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/check_multirank.py
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lammps_mtp_kokkos_amd import capi, mtpgen  # noqa: E402
from lammps_mtp_kokkos_amd.domain import HaloExchange, decompose  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("MTP_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    capi.use_private_torch_stream(dev)   # one real stream for torch ops, collectives and the library
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    pos, box = mtpgen.bcc_lattice(6, 6, 8, seed=91)
    types = (np.random.default_rng(3).random(len(pos)) < 0.1).astype(np.int32) + 1
    potf = os.path.join(ROOT, "potentials", "WRe_L20.mtp")
    pot = capi.Potential(potf)

    def forces(plan, split=False):
        """split: interior atoms are computed between forward_begin/forward_end (bench.py's overlap path)"""
        st = torch.cuda.current_stream().cuda_stream
        max_nn = int(np.diff(plan.first).max())
        x = torch.from_numpy(plan.x0).to(dev)
        x[plan.nlocal:] = 0.0
        halo = HaloExchange(plan, dev)
        ty = torch.from_numpy(plan.types).to(dev)
        f = torch.zeros((plan.nall, 3), dtype=torch.float64, device=dev)
        ev = torch.zeros(8, dtype=torch.float64, device=dev)
        keep = []
        if split:   # bench.py's overlapped step: rows interior | boundary | interior of ONE context
            from lammps_mtp_kokkos_amd.domain import overlap_order
            ilist, first, neigh, (na, nb, nc) = overlap_order(plan)
            ctx = capi.Context(pot, dev.index)
            t3 = [torch.from_numpy(a).to(dev) for a in (ilist, first, neigh)]
            keep.append(t3)
            ctx.set_neighbors_device(*t3, plan.nall, max_nn)
            h = halo.forward_begin(x)
            ctx.compute_device_rows(0, na, False, x, ty, f, eflag=1, vflag=1, stream=st)
            halo.forward_end(h)
            ctx.compute_device_rows(na, nb, False, x, ty, f, eflag=1, vflag=1, stream=st)
            ctx.compute_device_rows(na + nb, nc, True, x, ty, f, eflag=1, vflag=1, ev_t=ev, stream=st)
            ctx.synchronize(st)
        else:
            ctx = capi.Context(pot, dev.index)
            il, fi, ne = (torch.from_numpy(a).to(dev) for a in (plan.ilist, plan.first, plan.neigh))
            ctx.set_neighbors_device(il, fi, ne, plan.nall, max_nn)
            halo.forward(x)
            ctx.compute_device(x, ty, f, eflag=1, vflag=1, ev_t=ev, stream=st)
            ctx.synchronize(st)
        halo.reverse(f)
        return f[: plan.nlocal].cpu().numpy(), ev

    plan = decompose(pos, box, types, world, rank, 7.0)
    f, ev = forces(plan, split=os.environ.get("MTP_CHECK_SPLIT", "1") != "0")
    dist.all_reduce(ev)
    parts = [None] * world
    dist.all_gather_object(parts, (plan.owned_global, f))
    ok = True
    if rank == 0:
        F = np.zeros((len(pos), 3))
        for ids, ff in parts:
            F[ids] = ff
        dist.destroy_process_group()
        # single-domain result on this rank alone (world-1 plan needs no collective)
        one = decompose(pos, box, types, 1, 0, 7.0)
        f1, ev1 = forces(one)
        F1 = np.zeros((len(pos), 3))
        F1[one.owned_global] = f1
        err = np.abs(F - F1).max()
        de = abs(float(ev[0]) - float(ev1[0]))
        dv = np.abs(ev[1:7].cpu().numpy() - ev1[1:7].cpu().numpy()).max()
        print("multirank check: world=%d grid=%s max|dF|=%.3e |dE|=%.3e max|dvirial|=%.3e" % (world, plan.grid, err, de, dv))
        ok = err < 1e-9 and de < 1e-8 and dv < 1e-7
    else:
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
