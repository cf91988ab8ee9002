#!/usr/bin/env python3
"""Headline benchmark: atom-steps/s of the MTP force call on a 64k-atom BCC W crystal with
the level-16 potential (BASELINE.json configs[1]) on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one force call on positions already resident in HBM: zero the force array,
(N > 1: forward halo of ghost positions), the fused MTP kernel with global energy and virial
tallies, (N > 1: reverse halo of ghost forces).  N > 1 shards the SAME 65,536 atoms by
spatial domain decomposition (strong scaling, as BASELINE.json's metric asks); the halo is
the library's own (mtp_halo_*: device pack / unpack + one grouped RCCL send/recv per
direction on its own stream) and the owned atoms run as interior | boundary | interior row
ranges so both exchanges overlap force work -- no torch collective in the timed loop.
Rank 0 prints one JSON line; `roofline` prices the dominant kernel against the bound that governs
it (fp64 vector issue: the reference algorithm's flop count F_alg of SURVEY.md section 8d over the
kernel time measured here), with the HBM side (algorithmic bytes, counter traffic) in its `hbm`
sub-block, and `cpu_baseline` times the CPU oracle on this box's host cores in the same run.
N > 1 with a failing library halo exits non-zero (no silent second path; MTP_BENCH_HALO=torch
runs the torch twin on purpose).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vendor vector fp64 peak


def algorithmic_bytes(nlist_entries, nall, nlocal):
    """compulsory traffic of one force call."""
    return 4 * nlist_entries + 76 * nall + 8 * nlocal


def algorithmic_flops_reference(sizes, jc_total, nlocal):
    """SURVEY.md section 8(d) F_alg: flops the REFERENCE algorithm spends (the native kernel does
    fewer: it never forms the per-pair Jacobian)."""
    R, P, Mu, B, T, S = sizes["R"], sizes["P"], sizes["Mu"], sizes["B"], sizes["T"], sizes["S"]
    return jc_total * (9 + 8 * R + 4 * P + 4 * Mu * R + 34 * B) + nlocal * (9 * T + 2 * S)


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota where one is set (a GPU box
    hands each lease a share of its host cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def info_launch_waves(ctx):
    i = ctx.launch_info()
    return i["waves_per_block"] * max(1, i["grid_blocks"] // 256)


def main():
    # stdout carries exactly ONE line, the JSON record: everything else that libraries print there (RCCL's version
    # banner at communicator creation, for one) goes to stderr.  The original stdout is kept for the record.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=32, help="BCC cells per edge (32 -> 65,536 atoms)")
    ap.add_argument("--potential", default=os.path.join(ROOT, "potentials", "W_L16.mtp"))
    ap.add_argument("--variant", default="auto", choices=["auto", "large", "small"])
    ap.add_argument("--workload", default="w16", choices=["w16", "small2k", "wre20", "grades"],
                    help="w16: BASELINE configs[1] (default); small2k: configs[2]; wre20: level-20 W-Re shard of "
                         "configs[3]; grades: configs[4] (MaxVol neighbourhood grades every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: capture one step (zero f, force call, tally fold) in a HIP graph and time its replays "
                         "-- for launch-bound sizes (a caller like LAMMPS can do the same around mtp_compute_device)")
    ap.add_argument("--no-whole-step", action="store_true", help="skip the device-resident MD loop (whole-step time)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lammps_mtp_kokkos_amd import capi, mtpgen
    from lammps_mtp_kokkos_amd.domain import HaloExchange, decompose, overlap_order

    # BASELINE.json configs other than the headline one (the default is untouched by these)
    cells3 = (args.cells,) * 3
    species, grade, metric_name = 1, False, "atom-steps/s (64k-atom W, level-16 MTP)"
    if args.workload == "small2k":
        cells3 = (8, 8, 16)
        args.variant = "small" if args.variant == "auto" else args.variant
        metric_name = "atom-steps/s (2,048-atom W, level-16 MTP, small variant)"
    elif args.workload == "wre20":
        args.potential = os.path.join(ROOT, "potentials", "WRe_L20.mtp")
        species = 2
        metric_name = "atom-steps/s (W-Re 10%% Re, level-20 MTP, %d^3 cells)" % args.cells
    elif args.workload == "grades":
        args.potential = os.path.join(ROOT, "potentials", "W_L16_nbh.almtp")
        grade = True
        metric_name = "atom-steps/s (64k-atom W, level-16 MTP, MaxVol neighbourhood grades every step)"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    backend = os.environ.get("MTP_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of N > 1 on a single GPU
    devidx = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(devidx)
    dev = torch.device("cuda", devidx)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- synthetic workload --------------------------------------------------
    pos, box = mtpgen.bcc_lattice(*cells3, a=3.165, jitter=0.05, seed=777)
    natoms = len(pos)
    list_cutoff = 7.0
    gtypes = None
    if species == 2:   # W + 10 % Re, seeded
        gtypes = (np.random.default_rng(4242).random(natoms) < 0.10).astype(np.int32) + 1
    plan = decompose(pos, box, gtypes, world, rank, list_cutoff)
    pot = capi.Potential(args.potential, selection=grade)
    sizes = pot.sizes
    ctx = capi.Context(pot, devidx)
    ctx.set_variant(dict(auto=0, large=1, small=2)[args.variant])
    # N > 1: the rows are ordered interior | boundary | interior (atoms whose list holds no ghost are "interior")
    halo_kind = os.environ.get("MTP_BENCH_HALO", "native" if backend == "nccl" else "torch")
    # MTP_BENCH_SELF_HALO=1 (rehearsal on one GPU): the N > 1 step -- library halo, three row ranges -- with this rank's
    # own periodic images as its only peer, i.e. the per-step structure and overheads of a rank of a larger job
    self_halo = world == 1 and os.environ.get("MTP_BENCH_SELF_HALO", "0") == "1"
    decomposed = world > 1 or self_halo
    use_rows = decomposed and os.environ.get("MTP_BENCH_OVERLAP", "1") != "0"
    if use_rows:
        ilist_np, first_np, neigh_np, (n_a, n_b, n_c) = overlap_order(
            plan, align_rounds=os.environ.get("MTP_BENCH_ALIGN_ROUNDS", "1") != "0")
    else:
        ilist_np, first_np, neigh_np = plan.ilist, plan.first, plan.neigh
        n_a, n_b, n_c = 0, plan.nlocal, 0
    il = torch.from_numpy(ilist_np).to(dev)
    fi = torch.from_numpy(first_np).to(dev)
    ne = torch.from_numpy(neigh_np).to(dev)
    max_nn = int(np.diff(first_np).max()) if plan.nlocal else 0
    ctx.set_neighbors_device(il, fi, ne, plan.nall, max_nn)
    x = torch.from_numpy(plan.x0).to(dev)
    ty = torch.from_numpy(plan.types).to(dev)
    f = torch.zeros((plan.nall, 3), dtype=torch.float64, device=dev)
    ev = torch.zeros(8, dtype=torch.float64, device=dev)
    stream = capi.use_private_torch_stream(dev).cuda_stream   # torch ops and the library's kernels on ONE stream
    EFLAG, VFLAG = 1, 1
    grades_t = torch.zeros(plan.nall, dtype=torch.float64, device=dev) if grade else None
    maxg_t = torch.zeros(1, dtype=torch.float64, device=dev) if grade else None

    # the halo: the library's RCCL exchange (production), or the torch twin (gloo rehearsals on one GPU)
    halo, halo_info, halo_note = None, None, None
    if self_halo:
        halo_kind = "native"
        halo = capi.Halo(plan, devidx, capi.halo_unique_id())
        halo_info = halo.comm_count()
        if os.environ.get("MTP_BENCH_HALO_OVERLAP", "0") != "0":
            halo.set_overlap(True)
    elif world > 1:
        if halo_kind == "native":
            # every rank must end up on the same path: the outcome of the (collective) creation is agreed on below
            ok = 1
            try:
                store = dist.distributed_c10d._get_default_store()     # rendezvous plumbing only: 128 bytes, once
                if rank == 0:
                    store.set("mtp_halo_unique_id", capi.halo_unique_id())
                uid = bytes(store.get("mtp_halo_unique_id"))
                halo = capi.Halo(plan, devidx, uid)
                halo_info = halo.comm_count()
                if os.environ.get("MTP_BENCH_HALO_OVERLAP", "0") != "0":
                    halo.set_overlap(True)
                assert halo_info["nranks"] == world and halo_info["rank"] == rank
            except Exception as exc:
                ok, halo_note = 0, "library halo unavailable (%s: %s)" % (type(exc).__name__, exc)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                # No silent second code path: a scaling number must come from the RCCL halo it claims to measure.  The
                # torch all-to-all twin runs only when MTP_BENCH_HALO=torch asked for it explicitly.
                print("bench.py rank %d: %s -- refusing to fall back to the torch all-to-all halo "
                      "(set MTP_BENCH_HALO=torch to run that path on purpose)"
                      % (rank, halo_note or "library halo unavailable on another rank"), file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
        if halo_kind != "native":
            halo = HaloExchange(plan, dev)

    kw = dict(eflag=EFLAG, vflag=VFLAG, grade=grade, grades_t=grades_t, maxg_t=maxg_t, stream=stream)

    def step():
        if decomposed and halo_kind == "native":     # zero f, forward halo || rows A, rows B, reverse halo || rows C, fold
            halo.force_step(ctx, (n_a, n_b, n_c), x, ty, f, ev_t=ev, **kw)
            return
        capi.zero_async(f, stream)
        if world == 1:
            ctx.compute_device(x, ty, f, ev_t=ev, **kw)
        else:
            h = halo.forward_begin(x)
            if n_a:
                ctx.compute_device_rows(0, n_a, False, x, ty, f, **kw)
            halo.forward_end(h)
            ctx.compute_device_rows(n_a, n_b + n_c, True, x, ty, f, ev_t=ev, **kw)
            halo.reverse(f)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize(stream)
    # Which schedule of the decomposed step (one stream | exchanges overlapped with interior rows) is faster depends on
    # how long an exchange takes on the fabric at hand: unless MTP_BENCH_HALO_OVERLAP pins it, both are timed in the
    # warm-up (max over ranks, so every rank decides alike) and the faster one runs the timed steps.
    schedule_probe = None
    if decomposed and halo_kind == "native" and use_rows and "MTP_BENCH_HALO_OVERLAP" not in os.environ:
        probe = {}
        for mode in (False, True):
            halo.set_overlap(mode)
            for _ in range(5):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            fence()
            tp = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            probe[mode] = float(tp.item())
        halo.set_overlap(probe[True] < probe[False])
        schedule_probe = {"one_stream_ms": probe[False] * 1e3, "overlapped_ms": probe[True] * 1e3}
    run_step = step
    if args.graph and world == 1 and not decomposed:
        # the library's device path only launches kernels on the caller's stream, so a step can be captured and replayed
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=torch.cuda.current_stream(dev)):
            step()
        run_step = graph.replay
        for _ in range(3):
            run_step()
        ctx.synchronize(stream)
    ev.zero_()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    t_issue = time.perf_counter() - t0      # the host's share: when this approaches dt the GPU waits for launches
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if halo_kind == "native":
            halo.allreduce(ev, capi.REDUCE_SUM, stream)      # ncclAllReduce of {E, virial} on the library's communicator
            torch.cuda.synchronize()
        else:
            dist.all_reduce(ev)
    dt = float(tmax.item())
    ctx.synchronize(stream)
    energy_per_atom = float(ev[0].item()) / args.steps / natoms

    # ---- roofline of the dominant kernel: HIP events on the launch stream, separate pass -----------
    ctx.set_timing(True)
    kms = []
    for _ in range(min(args.steps, 50)):
        f.zero_()
        ctx.compute_device(x, ty, f, eflag=EFLAG, vflag=VFLAG, ev_t=ev, stream=stream, grade=grade,
                           grades_t=grades_t, maxg_t=maxg_t)
        kms.append(ctx.last_kernel_ms())
    ctx.set_timing(False)
    kernel_ms = float(np.mean(kms))

    # in-cutoff pair count of this rank (for the reference flop model), on the device
    with torch.no_grad():
        cnt = torch.diff(fi.long())
        row = torch.repeat_interleave(il.long(), cnt)
        d = x[ne.long()] - x[row]
        jc_total = int(((d * d).sum(1) <= pot.info.max_cutoff ** 2).sum().item())
    bytes_alg = algorithmic_bytes(int(plan.first[-1]), plan.nall, plan.nlocal)
    flops_ref = algorithmic_flops_reference(sizes, jc_total, plan.nlocal)
    achieved_gbs = bytes_alg / (kernel_ms * 1e-3) / 1e9
    # rocprofv3 PMC counters of the dominant kernel (scripts/gpu_pmc.sh -> profiles/r02_pmc_counters.json): only quoted when
    # they were collected from THIS build of the kernels (source hash) on this workload; per launch, like `achieved`
    traffic, pmc_block = None, None
    suffix = "" if args.workload == "w16" else "_" + args.workload
    cpath = next((c for c in (os.path.join(ROOT, "profiles", "r%02d_pmc_counters%s.json" % (r, suffix)) for r in (3, 2))
                  if os.path.exists(c)), "")
    if cpath and world == 1:
        try:
            pc = json.load(open(cpath))
            if pc.get("source_hash") == capi.kernel_source_hash() and pc.get("workload") == args.workload \
                    and pc.get("cells", 32) == args.cells:
                cn = pc["counters"]
                traffic = pc.get("hbm_bytes_per_launch")
                # kernel duration in shader cycles from the same counters: the wavefronts of the persistent grid live for the
                # whole launch, SQ_WAVE_CYCLES counts quad-cycles (MI355X_MICROARCH.md)
                ncu = 256
                cyc = cn["SQ_WAVE_CYCLES"] * 4.0 / cn["SQ_WAVES"]
                pmc_block = {
                    "replayed": True,   # NOT measured in this run: read from the committed rocprofv3 collection below
                    "source": "profiles/%s (same kernel sources: %s)" % (os.path.basename(cpath), pc["source_hash"][:12]),
                    "lds_busy": cn["SQ_LDS_IDX_ACTIVE"] / ncu / cyc,                       # LDS pipe cycles / kernel cycles, per CU
                    "lds_bank_conflict_share": cn["SQ_LDS_BANK_CONFLICT"] / max(cn["SQ_LDS_IDX_ACTIVE"], 1.0),
                    "valu_busy": cn["SQ_ACTIVE_INST_VALU"] * 4.0 / (4 * ncu) / cyc,         # quad-cycles of VALU issue per SIMD
                    "kernel_cycles": cyc, "clock_ghz_under_profiler": cyc / (kernel_ms * 1e-3) / 1e9,
                    "lds_wave_instructions": cn["SQ_INSTS_LDS"], "valu_wave_instructions": cn["SQ_INSTS_VALU"],
                    "waves_per_cu": cn["SQ_WAVES"] / ncu,
                }
        except Exception:
            traffic, pmc_block = None, None

    # ---- CPU baseline: the oracle (a port of the reference CPU path), rank 0, N = 1 -----------------
    # leg (i) one thread; leg (ii) every host core: threads over atoms adding into one force array (oracle/mtp_oracle_mt.c)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.pyoracle import Oracle
        o = Oracle(args.potential, selection=grade)
        xs = plan.x0
        nsub = min(plan.nlocal, 16384)
        sub_first = plan.first[: nsub + 1]
        sub_neigh = plan.neigh[: sub_first[-1]]
        passes, tcpu = 0, 0.0
        while tcpu < args.cpu_seconds / 2 and passes < 50:
            c0 = time.perf_counter()
            rc = o.compute(xs, plan.types, plan.ilist[:nsub], sub_first, sub_neigh, eflag=EFLAG, vflag=VFLAG,
                           extrapolation=grade)
            tcpu += time.perf_counter() - c0
            passes += 1
        one = nsub * passes / tcpu
        cpu = {"value": one, "unit": "atom-steps/s", "cores": 1, "kind": "port",
               "sample": "%d passes over the first %d of the %d atoms (same lattice, potential, list, flags), "
                         "serial C oracle, %.1f s" % (passes, nsub, natoms, tcpu),
               "host_cpus": os.cpu_count()}
        nthr = usable_cpus()
        if nthr > 1 and not grade:
            mp_, tmt = 0, 0.0
            while tmt < args.cpu_seconds / 2 and mp_ < 200:
                c0 = time.perf_counter()
                o.compute_mt(nthr, xs, plan.types, plan.ilist, plan.first, plan.neigh, eflag=EFLAG, vflag=VFLAG)
                tmt += time.perf_counter() - c0
                mp_ += 1
            cpu.update({"value": plan.nlocal * mp_ / tmt, "cores": nthr, "value_1_thread": one,
                        "sample": "%d passes over all %d atoms, %d threads over atoms adding into one force array "
                                  "(oracle/mtp_oracle_mt.c), %.1f s; 1 thread: %s" % (mp_, plan.nlocal, nthr, tmt, cpu["sample"])})
        # parity of the timed configuration, sampled: site energies of the sub-list
        ea = torch.zeros(plan.nall, dtype=torch.float64, device=dev)
        f.zero_()
        ctx.compute_device(x, ty, f, eflag=3, vflag=0, eatom_t=ea, ev_t=ev, stream=stream)
        ctx.synchronize(stream)
        nchk = min(nsub, 512)
        rchk = o.compute(xs, plan.types, plan.ilist[:nchk], plan.first[: nchk + 1], plan.neigh[: plan.first[nchk]],
                         eflag=3, vflag=0)
        cpu["max_abs_dE_site_eV"] = float(np.abs(ea.cpu().numpy()[:nchk] - rchk["eatom"][:nchk]).max())

    # device-resident neighbour-list build, informational: same atoms, same cutoff, own context
    list_build_ms = None
    if rank == 0 and world == 1:
        cnb = capi.Context(pot, devidx)
        xh = plan.x0
        lo, hi = xh.min(0) - 1e-9, xh.max(0) + 1e-9
        tot, _ = cnb.build_neighbors_device(x, plan.nlocal, plan.nall, list_cutoff, lo, hi, stream=stream)
        assert tot == int(plan.first[-1]), "device-built list differs in size from the host list"
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(5):
            cnb.build_neighbors_device(x, plan.nlocal, plan.nall, list_cutoff, lo, hi, stream=stream)
        torch.cuda.synchronize()
        list_build_ms = (time.perf_counter() - c0) / 5 * 1e3
        del cnb

    # ---- whole MD step ("force-only and whole-step"), N = 1: velocity-Verlet with everything in HBM --
    # kick + drift, ghost refresh, force call (same flags as above), ghost-force fold, kick; every 10th step the
    # ghost images and the neighbour list are rebuilt on the device (lammps_mtp_kokkos_amd/md.py)
    whole = None
    if rank == 0 and world == 1 and not grade and not args.no_whole_step:
        from lammps_mtp_kokkos_amd.md import DeviceNVE, MVV2E
        ctx_md = capi.Context(pot, devidx)
        ctx_md.set_variant(dict(auto=0, large=1, small=2)[args.variant])
        md = DeviceNVE(ctx_md, pos, box, rc=pot.info.max_cutoff, types=gtypes, mass=183.84, list_cutoff=list_cutoff,
                       device=dev, every=10, check_every=0, vflag=VFLAG)
        rng = np.random.default_rng(300)
        vel = rng.normal(size=pos.shape) * np.sqrt(8.617343e-5 * 30.0 / (183.84 * MVV2E))   # 30 K
        md.v.copy_(torch.from_numpy(vel - vel.mean(0)).to(dev))
        dt_ps, nmd = 2.5e-4, 60      # the synthetic potential is stiff: 0.25 fs (the time does not depend on dt)
        for _ in range(10):
            md.step(dt_ps)
        torch.cuda.synchronize()
        e0, b0 = md.total_energy(), md.builds
        c0 = time.perf_counter()
        for _ in range(nmd):
            md.step(dt_ps)
        torch.cuda.synchronize()
        wdt = time.perf_counter() - c0
        whole = {"ms_per_step": wdt / nmd * 1e3, "atom_steps_per_s": natoms * nmd / wdt, "steps": nmd,
                 "reneighbor_every": 10, "rebuilds_in_timed_steps": md.builds - b0, "dt_fs": dt_ps * 1e3,
                 "ghosts": md.nall - md.n, "list_entries": int(md.entries),
                 "energy_drift_eV_per_atom": (md.total_energy() - e0) / natoms,
                 "what": "velocity-Verlet, all arrays resident in HBM: kick+drift, ghost refresh, force call (eflag=1 "
                         "vflag=%d), ghost-force fold, kick; ghost images + full list rebuilt on the device every 10 steps" % VFLAG}
        del md, ctx_md

    if rank == 0:
        value = natoms * args.steps / dt
        info = ctx.launch_info()
        info.update(ctx.plan_info())
        line = {
            "metric": metric_name, "value": value, "unit": "atom-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d-atom BCC %s (a=3.165 A, +-0.05 A jitter, seed 777), %s "
                                   "(B=%d T=%d S=%d A=%d R=%d Mu=%d C=%d), rc=5 A, full list 7 A, eflag=1 vflag=1%s"
                                   % (natoms, "W" if species == 1 else "W-10%Re", os.path.basename(args.potential),
                                      sizes["B"], sizes["T"], sizes["S"], sizes["A"], sizes["R"], sizes["Mu"], sizes["C"],
                                      ", neighbourhood grades every step" if grade else ""),
                       "atoms": natoms, "potential": os.path.basename(args.potential),
                       "parallelism": ("domain decomposition %s, %s%s" % (
                           "x".join(map(str, plan.grid)),
                           "library halo: grouped RCCL send/recv per direction (communicator of %d ranks, RCCL %d)"
                           % (halo_info["nranks"], halo_info["rccl_version"]) if halo_info else "torch all-to-all halo (%s)%s" % (
                               backend, "; " + halo_note if halo_note else ""),
                           (", rows interior|boundary|interior = %d|%d|%d overlap both exchanges" % (n_a, n_b, n_c)
                            if (use_rows and (halo_kind != "native" or halo.overlap)) else
                            ", one stream: pack, forward exchange, all rows in one launch, reverse exchange, unpack")))
                       if decomposed else ("single GPU, step replayed from a HIP graph" if run_step is not step else "single GPU"),
                       "atoms_rank0": plan.nlocal, "ghosts_rank0": plan.nghost, "list_entries_rank0": int(plan.first[-1]),
                       "in_cutoff_pairs_rank0": jc_total, "launch": info, "halo_schedule_probe": schedule_probe,
                       "device_list_build_ms": list_build_ms,
                       # host time to ENQUEUE a step (python -> ctypes -> HIP launches), rank 0: the GPU waits for the
                       # host when this approaches ms_per_step
                       "host_issue_ms_per_step": t_issue / args.steps * 1e3},
            # The governing bound of the fused kernel is fp64 vector issue, so that is what `achieved` /
            # `peak` / `frac` price: the REFERENCE algorithm's flop count F_alg per launch over the kernel time measured
            # here with HIP events.  The HBM side (the contract's default bound) is the `hbm` sub-block: algorithmic bytes
            # over the same time, and the counter traffic beside it.
            "roofline": {"bound": "fp64_valu", "achieved": flops_ref / (kernel_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops_ref / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": "mtp_wave_kernel", "kernel_ms": kernel_ms, "reference_flops_per_launch": flops_ref,
                         "note": "achieved = SURVEY.md 8d F_alg (flops of the reference algorithm, not instructions executed: the "
                                 "native kernel never forms the per-pair Jacobian) / kernel time by HIP events in this run; "
                                 "traffic = HBM bytes per launch from rocprofv3 counters (replayed from profiles/, see pmc)",
                         "hbm": {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes": bytes_alg, "traffic": traffic,
                                 "traffic_ratio": (traffic / bytes_alg) if traffic else None},
                         "pmc": pmc_block},
            "cpu_baseline": cpu,
            "whole_step": whole,
            "energy_per_atom_eV": energy_per_atom,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    halo = None          # the RCCL communicator goes before the process group and the HIP runtime do
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
