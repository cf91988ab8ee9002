"""ctypes loader for the CPU oracle (oracle/libmtp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under lammps_mtp_kokkos_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Model(C.Structure):
    _fields_ = [
        ("scaling", C.c_double), ("min_cutoff", C.c_double), ("max_cutoff", C.c_double),
        ("species_count", C.c_int), ("radial_basis_size", C.c_int), ("radial_func_count", C.c_int),
        ("alpha_moment_count", C.c_int), ("alpha_index_basic_count", C.c_int),
        ("alpha_index_times_count", C.c_int), ("alpha_scalar_count", C.c_int),
        ("max_alpha_index_basic", C.c_int),
        ("alpha_index_basic", C.POINTER(C.c_int)), ("alpha_index_times", C.POINTER(C.c_int)),
        ("alpha_moment_mapping", C.POINTER(C.c_int)),
        ("radial_basis_coeffs", C.POINTER(C.c_double)), ("linear_coeffs", C.POINTER(C.c_double)),
        ("species_coeffs", C.POINTER(C.c_double)),
        ("has_selection", C.c_int), ("configuration_mode", C.c_int), ("coeff_count", C.c_int),
        ("active_set", C.POINTER(C.c_double)), ("inverse_active_set", C.POINTER(C.c_double)),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmtp_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.mtp_oracle_read_file.restype = C.c_int
        L.mtp_oracle_compute.restype = C.c_int
        L.mtp_oracle_compute_extrapolation.restype = C.c_int
        L.mtp_oracle_compute_mt.restype = C.c_int
        L.mtp_oracle_grade.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class Oracle:
    """One parsed potential + the two compute entry points."""

    def __init__(self, path, selection=False):
        self.m = Model()
        err = C.create_string_buffer(512)
        rc = lib().mtp_oracle_read_file(os.fsencode(path), int(selection), C.byref(self.m), err, 512)
        if rc:
            raise RuntimeError("oracle read_file rc=%d: %s" % (rc, err.value.decode()))

    def __del__(self):
        try:
            lib().mtp_oracle_free(C.byref(self.m))
        except Exception:
            pass

    # table views ------------------------------------------------------------------
    def arr(self, name, n, ty=np.float64):
        ptr = getattr(self.m, name)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(ty).copy()

    @property
    def sizes(self):
        m = self.m
        return dict(Sp=m.species_count, R=m.radial_basis_size, Mu=m.radial_func_count,
                    A=m.alpha_moment_count, B=m.alpha_index_basic_count, T=m.alpha_index_times_count,
                    S=m.alpha_scalar_count, P=m.max_alpha_index_basic, C=m.coeff_count)

    def radial_basis(self, dist):
        R = self.m.radial_basis_size
        v = np.zeros(R)
        d = np.zeros(R)
        lib().mtp_oracle_radial_basis(C.byref(self.m), C.c_double(dist), _p(v, C.c_double), _p(d, C.c_double))
        return v, d

    def compute_mt(self, nthreads, x, types, ilist, first, neigh, eflag=3, vflag=4):
        """compute() with threads over atoms (atomic adds into the one force array): the cpu_baseline leg that
        uses every host core."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        types = np.ascontiguousarray(types, dtype=np.int32)
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        first = np.ascontiguousarray(first, dtype=np.int32)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        nall = x.shape[0]
        f = np.zeros((nall, 3))
        eatom = np.zeros(nall)
        vatom = np.zeros((nall, 6))
        virial = np.zeros(6)
        e = C.c_double(0.0)
        rc = lib().mtp_oracle_compute_mt(C.byref(self.m), int(nthreads), nall, len(ilist), _p(ilist, C.c_int),
                                         _p(first, C.c_int), _p(neigh, C.c_int), _p(x, C.c_double), _p(types, C.c_int),
                                         eflag, vflag, _p(f, C.c_double), C.byref(e), _p(eatom, C.c_double),
                                         _p(virial, C.c_double), _p(vatom, C.c_double))
        if rc:
            raise RuntimeError("oracle compute_mt rc=%d" % rc)
        return dict(energy=e.value, eatom=eatom, f=f, virial=virial, vatom=vatom)

    def compute(self, x, types, ilist, first, neigh, eflag=3, vflag=4, extrapolation=False, natoms=0):
        """x [nall,3] f64, types [nall] i32 (1-based), CSR neighbour list over ilist.
        Returns dict(energy, eatom, f, virial, vatom[, grades, max_grade, coeff_ders])."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        types = np.ascontiguousarray(types, dtype=np.int32)
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        first = np.ascontiguousarray(first, dtype=np.int32)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        nall = x.shape[0]
        f = np.zeros((nall, 3))
        eatom = np.zeros(nall)
        vatom = np.zeros((nall, 6))
        virial = np.zeros(6)
        e = C.c_double(0.0)
        out = {}
        if not extrapolation:
            rc = lib().mtp_oracle_compute(C.byref(self.m), len(ilist), _p(ilist, C.c_int), _p(first, C.c_int),
                                          _p(neigh, C.c_int), _p(x, C.c_double), _p(types, C.c_int),
                                          eflag, vflag, _p(f, C.c_double), C.byref(e), _p(eatom, C.c_double),
                                          _p(virial, C.c_double), _p(vatom, C.c_double))
        else:
            grades = np.zeros(nall)
            mg = C.c_double(0.0)
            cd = np.zeros(self.m.coeff_count)
            rc = lib().mtp_oracle_compute_extrapolation(
                C.byref(self.m), len(ilist), _p(ilist, C.c_int), _p(first, C.c_int), _p(neigh, C.c_int),
                _p(x, C.c_double), _p(types, C.c_int), eflag, vflag, _p(f, C.c_double), C.byref(e),
                _p(eatom, C.c_double), _p(virial, C.c_double), _p(vatom, C.c_double),
                _p(grades, C.c_double), C.byref(mg), _p(cd, C.c_double), C.c_long(natoms))
            out.update(grades=grades, max_grade=mg.value, coeff_ders=cd)
        if rc:
            raise RuntimeError("oracle compute rc=%d" % rc)
        out.update(energy=e.value, eatom=eatom, f=f, virial=virial, vatom=vatom)
        return out
