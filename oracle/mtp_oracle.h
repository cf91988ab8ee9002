/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C, fp64, serial) of the reference's
 * `pair_style mtp` / `mtp/extrapolation` algorithm.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call this; the product (libmtp_mi355x.so) never
 * links or loads it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or potential files
 *  and its sources need LAMMPS headers (pair.h, memory.h,
 * text_file_reader.h, ...) that are absent from this image, so it cannot be compiled
 * here without writing stand-in headers, which this build's rules forbid.  The
 * restatement is therefore checked against (a) the closed form of the Chebyshev basis,
 * (b) an independent tensor-contraction (einsum) evaluation of the level-8 basis
 * functions from their mathematical definition, (c) F = -dE/dx by central differences,
 * (d) sum F = 0, E = sum eatom, rotation/translation/permutation invariance and the
 * strain derivative of E for the virial (tests/test_oracle.py).
 *
 * Each function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef MTP_ORACLE_H
#define MTP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define MTP_ORACLE_NEIGHMASK 0x1FFFFFFF /* LAMMPS NEIGHMASK, pair_mtp.cpp:114 */

typedef struct mtp_oracle_model {
  /* LAMMPS/ML-MTP/pair_mtp.h:47-83 (model state) */
  double scaling;
  double min_cutoff, max_cutoff;
  int species_count;
  int radial_basis_size;       /* R  */
  int radial_func_count;       /* Mu */
  int alpha_moment_count;      /* A  */
  int alpha_index_basic_count; /* B  */
  int alpha_index_times_count; /* T  */
  int alpha_scalar_count;      /* S  */
  int max_alpha_index_basic;   /* P = max(a+b+c)+1 */
  int *alpha_index_basic;      /* [B][4] {mu,a,b,c} */
  int *alpha_index_times;      /* [T][4] {a0,a1,mult,a3} */
  int *alpha_moment_mapping;   /* [S] */
  double *radial_basis_coeffs; /* [(t1*Sp+t2)*Mu*R + mu*R + ri] */
  double *linear_coeffs;       /* [S] */
  double *species_coeffs;      /* [Sp] */
  /* LAMMPS/ML-MTP/pair_mtp_extrapolation.h (selection state) */
  int has_selection;           /* 1 when a #MVS_v1.1 block was read */
  int configuration_mode;      /* energy_weight == 1 */
  int coeff_count;             /* C = Sp^2 Mu R + Sp + S */
  double *active_set;          /* [C][C] */
  double *inverse_active_set;  /* [C][C] */
} mtp_oracle_model;

/* pair_mtp.cpp:335-570 (+ mtp_radial_basis.cpp:59-102); with want_selection != 0 also
 * pair_mtp_extrapolation.cpp:528-612.  Returns 0 or a negative code with `err` filled. */
int mtp_oracle_read_file(const char *path, int want_selection, mtp_oracle_model *m, char *err,
                         int errlen);
void mtp_oracle_free(mtp_oracle_model *m);

/* mtp_rb_chevbyshev_basis.cpp:29-54 */
void mtp_oracle_radial_basis(const mtp_oracle_model *m, double dist, double *vals, double *ders);

/* pair_mtp.cpp:72-280.  Neighbours of ilist[ii] are neigh[first[ii] .. first[ii+1]) (the
 * reference's firstneigh[i] rows laid end to end; entries are masked with NEIGHMASK).
 * type is 1-based (LAMMPS).  f, virial, vatom accumulate (the caller zeroes, as LAMMPS
 * does); eatom[i] is assigned; *eng_vdwl accumulates.  eflag: bit0 global, bit1 per-atom;
 * vflag: nonzero = global virial, bit2 (4) also per-atom.  Returns 0, or -1 on a species
 * index outside the potential (pair_mtp.cpp:91-93,116-118). */
int mtp_oracle_compute(const mtp_oracle_model *m, int inum, const int *ilist, const int *first,
                       const int *neigh, const double *x, const int *type, int eflag, int vflag,
                       double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);

/* (internal to the threaded driver: this thread's compute calls add into a force array shared with other threads) */
void mtp_oracle_share_force_array(int on);

/* The same, threads over atoms (mtp_oracle_mt.c): nthreads slices of ilist adding into the one force array
 * with atomic adds.  bench.py's cpu_baseline leg (ii); nall = rows of x / f. */
int mtp_oracle_compute_mt(const mtp_oracle_model *m, int nthreads, int nall, int inum, const int *ilist,
                          const int *first, const int *neigh, const double *x, const int *type, int eflag,
                          int vflag, double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);

/* pair_mtp_extrapolation.cpp:68-382.  Neighbourhood mode: grades[i] for i in ilist,
 * *max_grade = max.  Configuration mode: coeff_ders[C] = sum_i dE_i/dtheta (before any
 * cross-rank reduction), *max_grade = max|A^-1 c| / natoms when natoms > 0
 * (compile_grades, :363-377).  coeff_ders may be NULL. */
int mtp_oracle_compute_extrapolation(const mtp_oracle_model *m, int inum, const int *ilist,
                                     const int *first, const int *neigh, const double *x,
                                     const int *type, int eflag, int vflag, double *f,
                                     double *eng_vdwl, double *eatom, double *virial,
                                     double *vatom, double *grades, double *max_grade,
                                     double *coeff_ders, long natoms);

/* pair_mtp_extrapolation.cpp:347-358 */
double mtp_oracle_grade(const mtp_oracle_model *m, const double *coeff_ders);

#ifdef __cplusplus
}
#endif
#endif
