#!/usr/bin/env python3
"""compile_grades across ranks (pair_mtp_extrapolation.cpp:363-382) on the sharded path: configuration mode
sums the candidate vector over ranks (all-reduce SUM of C doubles) before max|A^-1 c| / natoms; neighbourhood mode
takes the maximum of the per-rank maxima (all-reduce MAX).  Rank 0 compares with the single-domain HIP result.

  MTP_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/check_multirank_grades.py
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


def grade_call(pot, plan, dev):
    st = torch.cuda.current_stream().cuda_stream
    ctx = capi.Context(pot, dev.index)
    il, fi, ne = (torch.from_numpy(a).to(dev) for a in (plan.ilist, plan.first, plan.neigh))
    ctx.set_neighbors_device(il, fi, ne, plan.nall, int(np.diff(plan.first).max()))
    x = torch.from_numpy(plan.x0).to(dev)
    x[plan.nlocal:] = 0.0
    halo = HaloExchange(plan, dev)
    halo.forward(x)
    ty = torch.from_numpy(plan.types).to(dev)
    f = torch.zeros((plan.nall, 3), dtype=torch.float64, device=dev)
    ev = torch.zeros(8, dtype=torch.float64, device=dev)
    cfg = bool(pot.info.configuration_mode)
    coeff = torch.zeros(pot.info.coeff_count, dtype=torch.float64, device=dev) if cfg else None
    grades = None if cfg else torch.zeros(plan.nall, dtype=torch.float64, device=dev)
    maxg = None if cfg else torch.zeros(1, dtype=torch.float64, device=dev)
    ctx.compute_device(x, ty, f, eflag=1, vflag=0, ev_t=ev, grade=True, grades_t=grades, maxg_t=maxg, coeff_t=coeff,
                       stream=st)
    ctx.synchronize(st)
    return coeff, grades, maxg


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("MTP_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    capi.use_private_torch_stream(dev)   # one real stream for torch ops, collectives and the library
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    pos, box = mtpgen.bcc_lattice(6, 6, 8, seed=91)
    natoms = len(pos)
    results = {}
    for name, species in (("WRe_L10_cfg.almtp", 2), ("W_L16_nbh.almtp", 1)):
        types = (np.random.default_rng(3).random(natoms) < 0.1).astype(np.int32) + 1 if species == 2 else None
        pot = capi.Potential(os.path.join(ROOT, "potentials", name), selection=True)
        plan = decompose(pos, box, types, world, rank, 7.0)
        coeff, grades, maxg = grade_call(pot, plan, dev)
        if pot.info.configuration_mode:
            c = coeff.cpu() if backend == "gloo" else coeff
            dist.all_reduce(c, op=dist.ReduceOp.SUM)          # the cross-GPU sum of C doubles
            g = pot.cfg_grade(c.cpu().numpy()) / natoms        # :369-376
        else:
            m = maxg.cpu() if backend == "gloo" else maxg
            dist.all_reduce(m, op=dist.ReduceOp.MAX)           # :378
            g = float(m.item())
        results[name] = (pot, types, g)
    ok = True
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        for name, (pot, types, g) in results.items():
            one = decompose(pos, box, types, 1, 0, 7.0)
            coeff, grades, maxg = grade_call(pot, one, dev)
            g1 = pot.cfg_grade(coeff.cpu().numpy()) / natoms if pot.info.configuration_mode else float(maxg.item())
            err = abs(g - g1) / max(1.0, abs(g1))
            print("multirank grades: %s world=%d grade %.12g single-domain %.12g rel err %.2e" % (name, world, g, g1, err))
            ok = ok and err < 1e-9
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
