"""BASELINE.json configs[0] as a plumbing check of the whole loop: 256-atom BCC W, level-8 MTP, 10
velocity-Verlet NVE steps (dt = 1 fs, metal units) driven through the C ABI on the GPU, against the
same integrator driven by the CPU oracle.  Neighbour list rebuilt every step by the host stand-in."""
import os

import numpy as np
import pytest

from lammps_mtp_kokkos_amd import capi, mtpgen
from lammps_mtp_kokkos_amd.driver import periodic_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MVV2E = 1.0364269e-4          # (g/mol)(A/ps)^2 -> eV
FTM2V = 1.0 / MVV2E           # eV/A / (g/mol) -> A/ps^2
KB = 8.617343e-5
MASS = 183.84


def _wrapped_diff(a, b, box):
    """max |a - b| modulo the box (the device driver wraps owned atoms into the box at every re-neighbouring)"""
    d = a - b
    d -= np.asarray(box) * np.round(d / np.asarray(box))
    return float(np.abs(d).max())


def _run(force_fn, pos, box, vel, nsteps, dt):
    e_tot = []
    f, e = force_fn(pos)
    for _ in range(nsteps):
        vel = vel + 0.5 * dt * FTM2V * f / MASS
        pos = pos + dt * vel
        f, e = force_fn(pos)
        vel = vel + 0.5 * dt * FTM2V * f / MASS
        e_tot.append(e + 0.5 * MVV2E * MASS * (vel ** 2).sum())
    return pos, vel, np.array(e_tot)


@pytest.mark.gpu
def test_config1_ten_nve_steps_match_oracle_and_conserve_energy():
    from oracle.pyoracle import Oracle
    path = os.path.join(ROOT, "potentials", "W_L8.mtp")
    pos0, box = mtpgen.bcc_lattice(4, 4, 8)
    assert len(pos0) == 256
    rng = np.random.default_rng(300)
    vel0 = rng.normal(size=pos0.shape) * np.sqrt(KB * 300.0 / (MASS * MVV2E))
    vel0 -= vel0.mean(0)
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    orc = Oracle(path)

    def gpu_force(p):
        s = periodic_system(p, box, None, 7.0)
        ctx.set_neighbors(s.ilist, s.first, s.neigh, s.nall)
        r = ctx.compute(s.x, s.types, eflag=1, vflag=0)
        return s.fold_forces(r["f"]), r["energy"]

    def cpu_force(p):
        s = periodic_system(p, box, None, 7.0)
        r = orc.compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=1, vflag=0)
        return s.fold_forces(r["f"]), r["energy"]

    pg, vg, eg = _run(gpu_force, pos0.copy(), box, vel0.copy(), 10, 1e-3)
    pc, vc, ec = _run(cpu_force, pos0.copy(), box, vel0.copy(), 10, 1e-3)
    assert np.abs(pg - pc).max() < 1e-10 and np.abs(vg - vc).max() < 1e-9
    assert np.abs(eg - ec).max() < 1e-8
    assert np.abs(eg - eg[0]).max() < 2e-4 * 256      # NVE drift over 10 fs, eV


@pytest.mark.gpu
def test_device_resident_nve_matches_host_driven_loop():
    """the same ten steps with positions, velocities, ghosts and the neighbour list kept in HBM
    (list built by mtp_build_neighbors_device, ghost update / force fold as device index maps) follow the
    oracle-driven trajectory."""
    import torch
    from oracle.pyoracle import Oracle
    from lammps_mtp_kokkos_amd.md import DeviceNVE
    path = os.path.join(ROOT, "potentials", "W_L8.mtp")
    pos0, box = mtpgen.bcc_lattice(4, 4, 8)
    rng = np.random.default_rng(300)
    vel0 = rng.normal(size=pos0.shape) * np.sqrt(KB * 300.0 / (MASS * MVV2E))
    vel0 -= vel0.mean(0)
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    orc = Oracle(path)

    def cpu_force(p):
        s = periodic_system(p, box, None, 7.0)
        r = orc.compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=1, vflag=0)
        return s.fold_forces(r["f"]), r["energy"]

    md = DeviceNVE(ctx, pos0.copy(), box, rc=pot.info.max_cutoff, mass=MASS, list_cutoff=7.0, every=3)
    md.v.copy_(torch.from_numpy(vel0))
    eg = []
    for _ in range(10):
        md.step(1e-3)
        eg.append(md.total_energy())
    pc, vc, ec = _run(cpu_force, pos0.copy(), box, vel0.copy(), 10, 1e-3)
    assert md.builds >= 4                                   # re-neighboured on the device along the way
    assert _wrapped_diff(md.x.cpu().numpy(), pc, box) < 1e-10 and np.abs(md.v.cpu().numpy() - vc).max() < 1e-9
    assert np.abs(np.array(eg) - ec).max() < 1e-8


@pytest.mark.gpu
def test_device_resident_nve_level16_follows_oracle():
    """The device-resident loop at level 16 (the headline kernel shape): 432 atoms, 30 steps of 0.25 fs from 30 K,
    list rebuilt on the GPU every 5 steps, against the oracle-driven integrator.  (The synthetic random-coefficient
    potential is stiff and not bounded like a fitted one: a BCC lattice is not its minimum, so the run is kept short
    and cold; trajectory agreement, not long-time energy conservation, is the check.)"""
    import torch
    from oracle.pyoracle import Oracle
    from lammps_mtp_kokkos_amd.md import DeviceNVE
    path = os.path.join(ROOT, "potentials", "W_L16.mtp")
    pos0, box = mtpgen.bcc_lattice(6, 6, 6)
    rng = np.random.default_rng(17)
    vel0 = rng.normal(size=pos0.shape) * np.sqrt(KB * 30.0 / (MASS * MVV2E))
    vel0 -= vel0.mean(0)
    pot = capi.Potential(path)
    ctx = capi.Context(pot, 0)
    orc = Oracle(path)

    def cpu_force(p):
        s = periodic_system(p, box, None, 7.0)
        r = orc.compute(s.x, s.types, s.ilist, s.first, s.neigh, eflag=1, vflag=0)
        return s.fold_forces(r["f"]), r["energy"]

    md = DeviceNVE(ctx, pos0.copy(), box, rc=pot.info.max_cutoff, mass=MASS, list_cutoff=7.0, every=5)
    md.v.copy_(torch.from_numpy(vel0))
    eg = []
    for _ in range(30):
        md.step(2.5e-4)
        eg.append(md.total_energy())
    pc, vc, ec = _run(cpu_force, pos0.copy(), box, vel0.copy(), 30, 2.5e-4)
    assert md.builds >= 6
    assert _wrapped_diff(md.x.cpu().numpy(), pc, box) < 1e-9 and np.abs(md.v.cpu().numpy() - vc).max() < 1e-7
    assert np.abs(np.array(eg) - ec).max() < 1e-7
    assert abs(float(md.v.mean())) < 1e-9            # momentum conserved (forces sum to zero)


@pytest.mark.gpu
@pytest.mark.parametrize("ncell,cut", [((4, 4, 4), 7.0), ((3, 5, 4), 5.5), ((8, 8, 8), 7.0)])
def test_device_ghost_images_match_host_construction(ncell, cut):
    """mtp_ghosts_build against driver.make_ghosts: the same set of (owner, image position) pairs, the owned atoms
    wrapped into the box; forward refresh and reverse fold against numpy."""
    import torch
    from lammps_mtp_kokkos_amd.driver import make_ghosts
    pos, box = mtpgen.bcc_lattice(*ncell)
    rng = np.random.default_rng(21)
    pos = pos + rng.normal(0, 0.3, pos.shape) + np.array([40.0, -13.0, 0.2]) * (rng.random((len(pos), 1)) < 0.1)
    n = len(pos)
    dev = torch.device("cuda:0")
    st = capi.use_private_torch_stream(dev).cuda_stream
    g = capi.Ghosts(0)
    x = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    x.copy_(torch.from_numpy(pos))
    with pytest.raises(capi.MtpError) as ei:       # no room for ghosts: the size needed is reported
        g.build(x, n, box, cut, stream=st)
    assert ei.value.code == -24 and g.nall > n
    xa = torch.zeros((g.nall, 3), dtype=torch.float64, device=dev)
    xa[:n] = torch.from_numpy(pos)
    nall = g.build(xa, n, box, cut, stream=st)
    got = xa.cpu().numpy()
    wrapped = pos - np.floor(pos / box) * box
    assert np.abs(got[:n] - wrapped).max() < 1e-12 and (got[:n] >= 0).all() and (got[:n] < box).all()
    want_x, want_owner = make_ghosts(wrapped, box, cut)
    assert nall == len(want_x)
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert np.abs(key(np.round(got[n:], 9)) - key(np.round(want_x[n:], 9))).max() < 1e-9
    # forward: ghosts follow their owners; reverse: ghost rows add onto owner rows
    ty = torch.zeros(nall, dtype=torch.int32, device=dev)
    ty[:n] = torch.from_numpy(rng.integers(1, 4, n).astype(np.int32))
    g.types(ty, stream=st)
    moved = got.copy()
    moved[:n] += rng.normal(0, 0.05, (n, 3))
    xa[:n] = torch.from_numpy(moved[:n])
    g.forward(xa, stream=st)
    f_np = rng.normal(size=(nall, 3))
    f = torch.from_numpy(f_np.copy()).to(dev)
    g.reverse(f, stream=st)
    torch.cuda.synchronize()
    new = xa.cpu().numpy()
    d = new[n:] - got[n:]                      # every ghost moved exactly as its owner did
    tyh = ty.cpu().numpy()
    # recover owners from the types + displacement: each ghost's displacement equals one owned atom's displacement
    disp = moved[:n] - got[:n]
    tree_owner = np.array([int(np.argmin(np.abs(disp - dk).sum(1))) for dk in d])
    assert np.abs(disp[tree_owner] - d).max() < 1e-12 and np.array_equal(tyh[n:], tyh[:n][tree_owner])
    want_f = f_np[:n].copy()
    np.add.at(want_f, tree_owner, f_np[n:])
    assert np.abs(f.cpu().numpy()[:n] - want_f).max() < 1e-12
