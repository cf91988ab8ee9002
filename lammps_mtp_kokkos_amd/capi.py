"""ctypes binding of libmtp_mi355x.so (include/mtp_mi355x.h) for tests, bench and the
multi-GPU driver.  There is no fallback: a missing library or a missing gfx950 device
raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTP_LIB", os.path.join(_HERE, "libmtp_mi355x.so"))   # MTP_LIB: diagnostic builds

MTP_OK = 0
ERR_NAMES = {-2: "IO", -3: "EOF", -4: "FORMAT", -5: "PARSE", -6: "UNSUPPORTED", -7: "TABLE",
             -8: "SELECTION", -9: "MODE", -20: "ARG", -21: "DEVICE", -22: "SPECIES", -23: "STATE",
             -24: "LIMIT"}
VARIANT_AUTO, VARIANT_LARGE, VARIANT_SMALL = 0, 1, 2

EXPORTS = [
    "mtp_potential_load", "mtp_potential_free", "mtp_potential_get_info", "mtp_potential_get_tables",
    "mtp_context_create", "mtp_context_destroy", "mtp_last_error", "mtp_context_set_variant",
    "mtp_set_neighbors", "mtp_set_neighbors_csr", "mtp_set_neighbors_device", "mtp_compute",
    "mtp_compute_device", "mtp_synchronize", "mtp_cfg_grade", "mtp_context_launch_info",
    "mtp_context_set_timing", "mtp_context_last_kernel_ms", "mtp_build_neighbors_device",
    "mtp_copy_neighbors_to_host", "mtp_compute_device_rows", "mtp_context_plan_info",
    "mtp_halo_get_unique_id", "mtp_halo_create", "mtp_halo_destroy", "mtp_halo_last_error", "mtp_halo_comm_count",
    "mtp_halo_forward_begin", "mtp_halo_forward_end", "mtp_halo_forward", "mtp_halo_reverse_begin",
    "mtp_halo_reverse_end", "mtp_halo_reverse", "mtp_halo_allreduce", "mtp_halo_force_step", "mtp_halo_set_overlap",
    "mtp_halo_get_overlap", "mtp_halo_layout", "mtp_halo_pack_forward", "mtp_halo_unpack_reverse", "mtp_halo_get_layout",
    "mtp_halo_local_exchange", "mtp_set_neighbors_device_2d", "mtp_build_flags", "mtp_compute_resident",
    "mtp_resident_totals", "mtp_resident_peratom_device", "mtp_resident_peratom_host", "mtp_copy_to_host",
    "mtp_ghosts_create", "mtp_ghosts_destroy", "mtp_ghosts_last_error", "mtp_ghosts_build", "mtp_ghosts_forward",
    "mtp_ghosts_reverse", "mtp_ghosts_reverse_finish", "mtp_ghosts_types", "mtp_nve_initial", "mtp_nve_final", "mtp_nve_monitor",
    "mtp_context_set_deterministic", "mtp_zero_async",
]
HALO_ID_BYTES = 128
REDUCE_SUM, REDUCE_MAX = 0, 1


class MtpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmtp_mi355x: %s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class PotentialInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "species_count", "radial_basis_size", "radial_func_count", "alpha_moment_count",
        "alpha_index_basic_count", "alpha_index_times_count", "alpha_scalar_count",
        "max_alpha_index_basic", "coeff_count", "has_selection", "configuration_mode",
        "product_levels")] + [("scaling", C.c_double), ("min_cutoff", C.c_double), ("max_cutoff", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; importing it first makes this library bind to
        # that same HIP runtime (one runtime per process) instead of loading /opt/rocm's beside it
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for n in EXPORTS:
            getattr(L, n)           # AttributeError if the ABI drifted
        for n in EXPORTS:
            getattr(L, n).restype = C.c_int
        L.mtp_last_error.restype = C.c_char_p
        L.mtp_build_flags.restype = C.c_char_p
        L.mtp_halo_last_error.restype = C.c_char_p
        L.mtp_ghosts_last_error.restype = C.c_char_p
        L.mtp_ghosts_destroy.restype = None
        L.mtp_potential_free.restype = None
        L.mtp_context_destroy.restype = None
        L.mtp_halo_destroy.restype = None
        _lib = L
    return _lib


def kernel_source_hash():
    """sha256 over the sources of the force / grade kernels, their planner and the table builder: ties committed
    rocprofv3 counters to the build they came from.  The neighbour-list, halo and integrator kernels are other launches
    (their sources do not enter the counted kernels) and are left out, so that work on them does not orphan the counters."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    other_launches = ("mtp_neighbor_kernels.hip", "mtp_halo.hip", "mtp_md.hip")
    for n in sorted(os.listdir(src)):
        if n.endswith((".hip", ".hpp", ".cpp")) and n not in other_launches:
            h.update(n.encode())
            h.update(open(os.path.join(src, n), "rb").read())
    return h.hexdigest()


def build_flags():
    """compile-time switches of the loaded library that differ from the shipped defaults ("" for a release build)"""
    return lib().mtp_build_flags().decode()


def halo_layout(plan):
    """(send_off, recv_off) as the C side derives them from a plan's counts (host only: mtp_halo_layout)."""
    idx = np.ascontiguousarray(plan.send_idx, np.int32)
    sc = np.ascontiguousarray(plan.send_counts, np.int32)
    rc_ = np.ascontiguousarray(plan.recv_counts, np.int32)
    so = np.zeros(plan.nranks + 1, np.int32)
    ro = np.zeros(plan.nranks + 1, np.int32)
    err = C.create_string_buffer(512)
    rc = lib().mtp_halo_layout(int(plan.nranks), int(plan.nlocal), int(plan.nghost), _np(idx, C.c_int), _np(sc, C.c_int),
                               _np(rc_, C.c_int), _np(so, C.c_int), _np(ro, C.c_int), err, 512)
    if rc:
        raise MtpError(rc, err.value.decode())
    return so, ro


def use_private_torch_stream(device):
    """PyTorch's default stream is the legacy null stream (handle 0), and a null handle asks this library for the
    context's own NON-BLOCKING stream -- torch work and the library's kernels would then not be ordered with respect
    to each other.  Drivers that mix both call this once: it makes a real stream current for torch and returns its
    handle for the library's `stream` arguments (collectives issued from torch follow the current stream too)."""
    import torch
    s = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(s)
    return s


def _np(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def _ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


class Potential:
    def __init__(self, path, selection=False):
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().mtp_potential_load(os.fsencode(path), int(selection), C.byref(self.h), err, 512)
        if rc:
            raise MtpError(rc, err.value.decode())
        self.info = PotentialInfo()
        lib().mtp_potential_get_info(self.h, C.byref(self.info))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mtp_potential_free(self.h)
            self.h = None

    @property
    def sizes(self):
        i = self.info
        return dict(Sp=i.species_count, R=i.radial_basis_size, Mu=i.radial_func_count, A=i.alpha_moment_count,
                    B=i.alpha_index_basic_count, T=i.alpha_index_times_count, S=i.alpha_scalar_count,
                    P=i.max_alpha_index_basic, C=i.coeff_count, levels=i.product_levels)

    def tables(self):
        i = self.info
        out = dict(
            alpha_index_basic=np.zeros((i.alpha_index_basic_count, 4), np.int32),
            alpha_index_times=np.zeros((i.alpha_index_times_count, 4), np.int32),
            alpha_moment_mapping=np.zeros(i.alpha_scalar_count, np.int32),
            radial_coeffs=np.zeros(i.species_count ** 2 * i.radial_func_count * i.radial_basis_size),
            species_coeffs=np.zeros(i.species_count), moment_coeffs=np.zeros(i.alpha_scalar_count))
        inv = np.zeros((i.coeff_count, i.coeff_count)) if i.has_selection else None
        rc = lib().mtp_potential_get_tables(
            self.h, _np(out["alpha_index_basic"], C.c_int32), _np(out["alpha_index_times"], C.c_int32),
            _np(out["alpha_moment_mapping"], C.c_int32), _np(out["radial_coeffs"], C.c_double),
            _np(out["species_coeffs"], C.c_double), _np(out["moment_coeffs"], C.c_double), _np(inv, C.c_double))
        if rc:
            raise MtpError(rc, "get_tables")
        if inv is not None:
            out["inverse_active_set"] = inv
        return out

    def cfg_grade(self, coeff_ders):
        g = C.c_double(0)
        c = np.ascontiguousarray(coeff_ders, dtype=np.float64)
        rc = lib().mtp_cfg_grade(self.h, _np(c, C.c_double), C.byref(g))
        if rc:
            raise MtpError(rc, "cfg_grade")
        return g.value


class Context:
    """One GPU context.  Host-array path: set_neighbors + compute.  Device path:
    set_neighbors_device + compute_device with torch tensors."""

    def __init__(self, pot: Potential, device=0):
        self.pot = pot
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().mtp_context_create(pot.h, int(device), C.byref(self.h), err, 512)
        if rc:
            raise MtpError(rc, err.value.decode())
        self._keep = []

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mtp_context_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc:
            raise MtpError(rc, lib().mtp_last_error(self.h).decode())

    def set_variant(self, v):
        self._check(lib().mtp_context_set_variant(self.h, int(v)))

    def set_neighbors(self, ilist, first, neigh, nall):
        ilist = np.ascontiguousarray(ilist, np.int32)
        first = np.ascontiguousarray(first, np.int32)
        neigh = np.ascontiguousarray(neigh, np.int32)
        self.nall = int(nall)
        self._check(lib().mtp_set_neighbors_csr(self.h, len(ilist), _np(ilist, C.c_int), _np(first, C.c_int),
                                                _np(neigh, C.c_int), int(nall)))

    def set_neighbors_lammps(self, ilist, numneigh, rows, nall):
        """LAMMPS form: numneigh[i], firstneigh[i] indexed by atom id (rows: list of int32 arrays)."""
        ilist = np.ascontiguousarray(ilist, np.int32)
        numneigh = np.ascontiguousarray(numneigh, np.int32)
        rows = [np.ascontiguousarray(r, np.int32) for r in rows]
        arr = (C.POINTER(C.c_int) * len(rows))(*[_np(r, C.c_int) for r in rows])
        self.nall = int(nall)
        self._check(lib().mtp_set_neighbors(self.h, len(ilist), _np(ilist, C.c_int), _np(numneigh, C.c_int), arr,
                                            int(nall)))

    def set_neighbors_device_2d(self, ilist_t, numneigh_t, neighbors_t, stride_i, stride_jj, max_neighs, nall, stream=None):
        """LAMMPS-KOKKOS form: d_ilist(ii), d_numneigh(i), padded 2-D view d_neighbors(i, jj) with the given strides."""
        self._keep = [ilist_t, numneigh_t, neighbors_t]
        self.nall = int(nall)
        self._inum = int(ilist_t.numel())
        self._check(lib().mtp_set_neighbors_device_2d(self.h, C.c_void_p(stream) if stream else None, int(ilist_t.numel()),
                                                      _ptr(ilist_t), _ptr(numneigh_t), _ptr(neighbors_t),
                                                      C.c_longlong(stride_i), C.c_longlong(stride_jj), int(max_neighs),
                                                      int(nall)))

    def set_neighbors_device(self, ilist_t, first_t, neigh_t, nall, max_numneigh):
        self._keep = [ilist_t, first_t, neigh_t]
        self.nall = int(nall)
        self._check(lib().mtp_set_neighbors_device(self.h, int(ilist_t.numel()), _ptr(ilist_t), _ptr(first_t),
                                                   _ptr(neigh_t), int(nall), int(max_numneigh)))

    def compute(self, x, types, eflag=3, vflag=4, grade=False):
        x = np.ascontiguousarray(x, np.float64)
        types = np.ascontiguousarray(types, np.int32)
        nall = x.shape[0]
        assert nall == self.nall
        f = np.zeros((nall, 3))
        eatom = np.zeros(nall)
        vatom = np.zeros((nall, 6))
        virial = np.zeros(6)
        e = C.c_double(0)
        mg = C.c_double(0)
        grades = np.zeros(nall) if grade else None
        cd = np.zeros(self.pot.info.coeff_count) if grade else None
        self._check(lib().mtp_compute(self.h, _np(x, C.c_double), _np(types, C.c_int), int(eflag), int(vflag),
                                      int(bool(grade)), _np(f, C.c_double), _np(eatom, C.c_double),
                                      _np(vatom, C.c_double), C.byref(e), _np(virial, C.c_double),
                                      _np(grades, C.c_double), C.byref(mg), _np(cd, C.c_double)))
        out = dict(energy=e.value, eatom=eatom, f=f, virial=virial, vatom=vatom)
        if grade:
            out.update(grades=grades, max_grade=mg.value, coeff_ders=cd)
        return out

    def build_neighbors_device(self, x_t, inum, nall, list_cutoff, lo, hi, stream=None):
        """Full neighbour list built on the GPU from device-resident positions; returns
        (entries, longest row).  The list stays in the context."""
        lo3 = (C.c_double * 3)(*[float(v) for v in lo])
        hi3 = (C.c_double * 3)(*[float(v) for v in hi])
        total, mx = C.c_longlong(0), C.c_int32(0)
        st = C.c_void_p(stream) if stream else None
        self._check(lib().mtp_build_neighbors_device(self.h, st, _ptr(x_t), int(inum), int(nall), C.c_double(list_cutoff),
                                                     lo3, hi3, None, None, C.byref(total), C.byref(mx)))
        self._inum = int(inum)
        self.nall = int(nall)
        return total.value, mx.value

    def neighbors_to_host(self, inum=None, total=None):
        """(first, neigh) of the list the context owns, as numpy arrays."""
        inum = self._inum if inum is None else inum
        first = np.zeros(inum + 1, dtype=np.int32)
        self._check(lib().mtp_copy_neighbors_to_host(self.h, _np(first, C.c_int32), None))
        neigh = np.zeros(max(int(first[-1]), 1), dtype=np.int32)
        self._check(lib().mtp_copy_neighbors_to_host(self.h, _np(first, C.c_int32), _np(neigh, C.c_int32)))
        return first, neigh[: int(first[-1])]

    def compute_device(self, x_t, type_t, f_t, eflag=0, vflag=0, grade=False, eatom_t=None, vatom_t=None,
                       ev_t=None, grades_t=None, maxg_t=None, coeff_t=None, stream=None):
        st = C.c_void_p(stream) if stream else None
        self._check(lib().mtp_compute_device(self.h, st, _ptr(x_t), _ptr(type_t), int(eflag), int(vflag),
                                             int(bool(grade)), _ptr(f_t), _ptr(eatom_t), _ptr(vatom_t),
                                             _ptr(ev_t), _ptr(grades_t), _ptr(maxg_t), _ptr(coeff_t)))

    def compute_device_rows(self, row_begin, row_count, finish, x_t, type_t, f_t, eflag=0, vflag=0, grade=False,
                            eatom_t=None, vatom_t=None, ev_t=None, grades_t=None, maxg_t=None, coeff_t=None, stream=None):
        """Rows [row_begin, row_begin + row_count) of the installed list; `finish` folds the energy / virial tallies."""
        st = C.c_void_p(stream) if stream else None
        self._check(lib().mtp_compute_device_rows(self.h, st, int(row_begin), int(row_count), int(bool(finish)),
                                                  _ptr(x_t), _ptr(type_t), int(eflag), int(vflag), int(bool(grade)),
                                                  _ptr(f_t), _ptr(eatom_t), _ptr(vatom_t), _ptr(ev_t),
                                                  _ptr(grades_t), _ptr(maxg_t), _ptr(coeff_t)))

    def synchronize(self, stream=None):
        self._check(lib().mtp_synchronize(self.h, C.c_void_p(stream) if stream else None))

    def plan_info(self):
        a, b = C.c_int32(), C.c_int32()
        self._check(lib().mtp_context_plan_info(self.h, C.byref(a), C.byref(b)))
        return dict(waves_per_simd=a.value, rebuild_tables=b.value)

    def launch_info(self):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(lib().mtp_context_launch_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(lds_bytes_per_wave=a.value, waves_per_block=b.value, grid_blocks=c.value,
                    neighbor_tile=d.value)

    def set_deterministic(self, on=True):
        self._check(lib().mtp_context_set_deterministic(self.h, int(on)))

    def set_timing(self, on=True):
        self._check(lib().mtp_context_set_timing(self.h, int(on)))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(lib().mtp_context_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


def zero_async(t, stream=None):
    """t[...] = 0 on `stream` in one kernel launch (fp64 tensor)"""
    rc = lib().mtp_zero_async(C.c_void_p(stream) if stream else None, _ptr(t), C.c_longlong(t.numel()))
    if rc:
        raise MtpError(rc, "mtp_zero_async")


def halo_unique_id():
    """ncclGetUniqueId (one rank calls this and hands the bytes to every rank)."""
    buf = C.create_string_buffer(HALO_ID_BYTES)
    rc = lib().mtp_halo_get_unique_id(buf)
    if rc:
        raise MtpError(rc, "ncclGetUniqueId failed")
    return buf.raw


class Halo:
    """The library's RCCL halo (include/mtp_mi355x.h, "multi-GPU halo") for one rank of a decomposition
    (domain.HaloPlan).  Creation is collective over all ranks -- unless unique_id is None: the halo then has no
    communicator (pack / unpack / layout only; segments move through Halo.local_exchange)."""

    def __init__(self, plan, device, unique_id):
        self.plan = plan
        self.h = C.c_void_p()
        idx = np.ascontiguousarray(plan.send_idx, np.int32)
        shift = np.ascontiguousarray(plan.send_shift, np.float64).reshape(-1, 3)
        sc = np.ascontiguousarray(plan.send_counts, np.int32)
        rc_ = np.ascontiguousarray(plan.recv_counts, np.int32)
        assert (unique_id is None or len(unique_id) == HALO_ID_BYTES) and len(sc) == plan.nranks == len(rc_)
        err = C.create_string_buffer(512)
        rc = lib().mtp_halo_create(int(device), int(plan.nranks), int(plan.rank), unique_id, int(plan.nlocal),
                                   int(plan.nghost), _np(idx, C.c_int), _np(shift, C.c_double), _np(sc, C.c_int),
                                   _np(rc_, C.c_int), C.byref(self.h), err, 512)
        if rc:
            raise MtpError(rc, err.value.decode())

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mtp_halo_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc:
            raise MtpError(rc, lib().mtp_halo_last_error(self.h).decode())

    def comm_count(self):
        n, r, v = C.c_int(), C.c_int(), C.c_int()
        self._check(lib().mtp_halo_comm_count(self.h, C.byref(n), C.byref(r), C.byref(v)))
        return dict(nranks=n.value, rank=r.value, rccl_version=v.value)

    @staticmethod
    def _st(stream):
        return C.c_void_p(stream) if stream else None

    def forward_begin(self, x_t, stream=None):
        self._check(lib().mtp_halo_forward_begin(self.h, self._st(stream), _ptr(x_t)))

    def forward_end(self, stream=None):
        self._check(lib().mtp_halo_forward_end(self.h, self._st(stream)))

    def forward(self, x_t, stream=None):
        self._check(lib().mtp_halo_forward(self.h, self._st(stream), _ptr(x_t)))

    def reverse_begin(self, f_t, stream=None):
        self._check(lib().mtp_halo_reverse_begin(self.h, self._st(stream), _ptr(f_t)))

    def reverse_end(self, f_t, stream=None):
        self._check(lib().mtp_halo_reverse_end(self.h, self._st(stream), _ptr(f_t)))

    def reverse(self, f_t, stream=None):
        self._check(lib().mtp_halo_reverse(self.h, self._st(stream), _ptr(f_t)))

    def force_step(self, ctx, rows, x_t, type_t, f_t, eflag=0, vflag=0, grade=False, eatom_t=None, vatom_t=None, ev_t=None,
                   grades_t=None, maxg_t=None, coeff_t=None, stream=None):
        """One decomposed force call (one C call).  Default: zero f + pack, forward exchange, all rows in one launch,
        reverse exchange, unpack, on one stream; after set_overlap(True): forward halo || interior rows, boundary rows,
        reverse halo || interior rows (rows = (nA, nB, nC) of domain.overlap_order)."""
        na, nb, nc = rows
        rc = lib().mtp_halo_force_step(self.h, ctx.h, self._st(stream), int(na), int(nb), int(nc), _ptr(x_t), _ptr(type_t),
                                       int(eflag), int(vflag), int(bool(grade)), _ptr(f_t), _ptr(eatom_t), _ptr(vatom_t),
                                       _ptr(ev_t), _ptr(grades_t), _ptr(maxg_t), _ptr(coeff_t))
        if rc:
            msg = lib().mtp_halo_last_error(self.h).decode() or lib().mtp_last_error(ctx.h).decode()
            raise MtpError(rc, msg)

    def pack_forward(self, x_t, stream):
        self._check(lib().mtp_halo_pack_forward(self.h, self._st(stream), _ptr(x_t)))

    def unpack_reverse(self, f_t, stream):
        self._check(lib().mtp_halo_unpack_reverse(self.h, self._st(stream), _ptr(f_t)))

    def layout(self):
        n = self.plan.nranks
        ns = C.c_int()
        so, sc, ro, rc_ = (np.zeros(n + 1, np.int32), np.zeros(n, np.int32), np.zeros(n + 1, np.int32), np.zeros(n, np.int32))
        self._check(lib().mtp_halo_get_layout(self.h, C.byref(ns), _np(so, C.c_int), _np(sc, C.c_int), _np(ro, C.c_int),
                                              _np(rc_, C.c_int)))
        return dict(nsend=ns.value, send_off=so, send_counts=sc, recv_off=ro, recv_counts=rc_)

    @staticmethod
    def local_exchange(halos, direction, arrays, stream):
        """All ranks' exchange of one direction (0 forward: arrays = positions, 1 reverse: arrays = forces) between the
        halo objects of one decomposition living in this process (mtp_halo_local_exchange)."""
        n = len(halos)
        hs = (C.c_void_p * n)(*[h.h for h in halos])
        ps = (C.c_void_p * n)(*[t.data_ptr() for t in arrays])
        rc = lib().mtp_halo_local_exchange(hs, n, C.c_void_p(stream), int(direction), ps)
        if rc:
            raise MtpError(rc, lib().mtp_halo_last_error(halos[0].h).decode())

    def set_overlap(self, enable):
        self._check(lib().mtp_halo_set_overlap(self.h, int(bool(enable))))

    @property
    def overlap(self):
        return bool(lib().mtp_halo_get_overlap(self.h))

    def allreduce(self, buf_t, op=REDUCE_SUM, stream=None):
        self._check(lib().mtp_halo_allreduce(self.h, self._st(stream), _ptr(buf_t), int(buf_t.numel()), int(op)))


class Ghosts:
    """Periodic ghost images of one GPU's own atoms, kept on the device (mtp_ghosts_*)."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().mtp_ghosts_create(int(device), C.byref(self.h))
        if rc:
            raise MtpError(rc, "mtp_ghosts_create")
        self.nall = 0

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mtp_ghosts_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc:
            raise MtpError(rc, lib().mtp_ghosts_last_error(self.h).decode())

    def build(self, x_t, nlocal, box, rghost, stream=None):
        """Wraps x_t[:nlocal] and writes the ghost positions behind them; returns nall.  x_t must have room:
        raises MtpError(LIMIT) otherwise, with self.nall set to the size needed."""
        b3 = (C.c_double * 3)(*[float(v) for v in box])
        nall = C.c_int(0)
        rc = lib().mtp_ghosts_build(self.h, C.c_void_p(stream) if stream else None, _ptr(x_t), int(nlocal),
                                    int(x_t.shape[0]), b3, C.c_double(rghost), C.byref(nall))
        self.nall = nall.value
        self._check(rc)
        return nall.value

    def forward(self, x_t, stream=None):
        self._check(lib().mtp_ghosts_forward(self.h, C.c_void_p(stream) if stream else None, _ptr(x_t)))

    def reverse(self, f_t, stream=None):
        self._check(lib().mtp_ghosts_reverse(self.h, C.c_void_p(stream) if stream else None, _ptr(f_t)))

    def reverse_finish(self, ctx, f_t, ev_t, eflag=0, vflag=0, stream=None):
        """ghost forces onto their owners + the tally fold of a force call made with finish_tallies=False, one launch"""
        self._check(lib().mtp_ghosts_reverse_finish(self.h, ctx.h, C.c_void_p(stream) if stream else None, int(eflag), int(vflag),
                                                    _ptr(f_t), _ptr(ev_t)))

    def types(self, type_t, stream=None):
        self._check(lib().mtp_ghosts_types(self.h, C.c_void_p(stream) if stream else None, _ptr(type_t)))


def nve_initial(nlocal, x_t, v_t, f_t, type_t, inv_mass_t, dtf, dt, stream=None):
    rc = lib().mtp_nve_initial(C.c_void_p(stream) if stream else None, int(nlocal), _ptr(x_t), _ptr(v_t), _ptr(f_t),
                               _ptr(type_t), _ptr(inv_mass_t), C.c_double(dtf), C.c_double(dt))
    if rc:
        raise MtpError(rc, "mtp_nve_initial")


def nve_final(nlocal, v_t, f_t, type_t, inv_mass_t, dtf, stream=None):
    rc = lib().mtp_nve_final(C.c_void_p(stream) if stream else None, int(nlocal), _ptr(v_t), _ptr(f_t), _ptr(type_t),
                             _ptr(inv_mass_t), C.c_double(dtf))
    if rc:
        raise MtpError(rc, "mtp_nve_final")


def nve_monitor(nlocal, x_t, x_ref_t, v_t, type_t, mass_t, out2_t, stream=None):
    rc = lib().mtp_nve_monitor(C.c_void_p(stream) if stream else None, int(nlocal), _ptr(x_t), _ptr(x_ref_t), _ptr(v_t),
                               _ptr(type_t), _ptr(mass_t), _ptr(out2_t))
    if rc:
        raise MtpError(rc, "mtp_nve_monitor")
