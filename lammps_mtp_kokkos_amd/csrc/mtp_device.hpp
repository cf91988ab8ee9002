// Kernel parameter block shared by the host context and the HIP kernels (internal).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#define MTP_EV_SLOTS 4096   // per-wave energy/virial tally slots (8 doubles each); >= wavefronts of a launch (256 CUs x 12), so
                            // every slot has one writer and the folded sums do not depend on timing
#define MTP_MAX_WPB 12      // wavefronts per workgroup: 8 (512 threads) in the 2-per-SIMD build, 12 in the 3-per-SIMD build
#define MTP_PITCH 33        // doubles per row of the per-wavefront LDS tables (32 neighbour columns + 1: odd pitch)
#define MTP_PSTRIDE 12      // slot ids per mu in the LDS blob (nu = 0..11, -1 padded)

// A times row packed in 8 bytes: lo = 8 a0 | 8 a1 << 16, hi = 8 a3 | (mult & 0xffff) << 16 -- BYTE offsets of the moments
// in the per-atom LDS image (moment indices below 8192)
struct alignas(8) MtpRow8 {   // 8-byte aligned: one ds_read_b64 / global_load_dwordx2 per row, not two dword reads
  uint32_t lo, hi;
};

struct MtpDevParams {
  // potential sizes
  int Sp, R, Mu, P, A, B, T, S, C;
  int nslot, nlevels, nseed;
  // leaf moments (mtp_potential.hpp) have no LDS slot in force calls: Ad = stored moments (adjoint extent), Am = moment
  // extent of this launch (Ad; A in grade calls, which put the leaves' values into the candidate vector), Se = scalars
  // of stored moments (the energy tables map / lin and the seeds list only those); nlevels = dependency levels, the
  // leaf rows are entry [nlevels] of the level table, their constants leaf_cf / leaf_cb (HBM / L2) one per padded row
  int Am, Ad, Se;
  const double *leaf_cf, *leaf_cb;
  const int *g_map_all;    // int[S]: every scalar's moment (grade calls)
  // slots are numbered by tensor rank: rank d owns slots [deg_first[d], deg_first[d+1]); the force phase keeps
  // one block of derivative-polynomial coefficients per slot (1 double for rank 0, 3*d*(d+1)/2 for rank d:
  // x-, y-, z-derivative, each over the monomials of degree d-1), rank d's blocks starting at deg_coef[d]
  int deg_first[MTP_PSTRIDE + 2], deg_coef[MTP_PSTRIDE + 2];
  int coef_total;          // doubles of all coefficient blocks
  int coef_dense;          // every coefficient has a source basic (no zero fill)
  double rmin, rmax, scaling, cutsq, inv_span;   // inv_span = 1 / (rmax - rmin)
  double inv_rmax;
  // Read-mostly tables, one contiguous blob in HBM that every workgroup copies into the
  // head of its LDS once; offsets in bytes from the blob start (all 8-byte aligned).
  const unsigned char *blob;
  int blob_bytes;          // multiple of 16
  int off_rows;            // MtpRow8[T]   (only when rows_in_lds)
  int off_level;           // int[nlevels+2]
  int off_slot;            // int[Mu][MTP_PSTRIDE], -1 padded, 16-byte aligned
  int off_radial;          // double[Sp*Sp*Mu*R]
  int off_seed_idx;        // int[nseed]
  int off_seed_val;        // double[nseed]
  int off_map;             // int[Se]
  int off_lin;             // double[Se]
  int off_pack;            // int[B] slot | a<<8 | b<<12 | c<<16 | mu<<20
  int off_fwd;             // int[nfb][8] head x tail blocks of the basic-moment pass (mtp_potential.hpp)
  int nfb;                 // number of those blocks
  int off_smu;             // int[nslot] radial function index mu of each slot
  int off_coef;            // int2[B] scatter targets of each basic's adjoint: {tx | ty << 16, tz | fa << 16 | fb << 20 | fc << 24}
  int off_leaf_cf, off_leaf_cb;   // double[leaf rows], behind the rows (same prefix): one array when the two agree
  int rows_in_lds;
  int tgt_in_lds;          // the scatter targets (off_coef) are in the blob prefix of this launch; else read g_tgt (HBM / L2)
  const int *g_tgt;
  // scalar map / linear coefficients / adjoint seeds: in the blob when small, else read from HBM/L2 once per atom
  // (at level 20 they are 11 KB, the difference between 3 and 4 wavefronts per CU)
  int scalars_in_lds;
  const int *g_map, *g_seed_idx;
  const double *g_lin, *g_seed_val;
  const MtpRow8 *rows;     // [T] by level, in HBM (always valid)
  // gather programs of the product passes (mtp_potential.hpp): operations in HBM / L2, segment tables in the blob
  const MtpRow8 *prog_fwd, *prog_bwd;
  int off_seg_fwd, off_seg_bwd;   // int[nlevels][4] {first block, groups, chunk size, 0}
  const double *species_coeffs;
  // system
  int inum, nall;
  const int *ilist, *first, *neigh;
  const double *x;         // [nall][3]
  const int *type;         // [nall] 1-based
  // outputs
  double *f;               // [nall][3] accumulated
  long long *fq;           // deterministic mode: [nall][3] fixed-point force accumulators (else null)
  double *eatom;           // [nall] or null
  double *vatom;           // [nall][6] or null
  double *ev_slots;        // [8][MTP_EV_SLOTS], quantity-major
  double *cvec;            // [inum][cpad] candidate vectors dE_i/dtheta (grade calls only)
  int cpad;                // row stride of cvec and of the padded inverse active set (multiple of 16)
  double *dbasic;          // [inum][dpad] adjoints of the basics, zero padded (grade calls only)
  int dpad;                // = KL*KB of the lane grid
  int *err_flag;
  unsigned long long *stamps;   // [16] diagnostic build only (MTP_STAMPS), else unused
  int eflag, vflag, grade_flag;
  int xcd_map;             // contiguous eighths of ilist per XCD (workgroup b belongs to XCD b % 8)
  int grade_fused;         // grade calls: the force kernel also writes the radial block of cvec (R = 8, Mu <= 4, Sp <= 2)
  // launch geometry
  int NT;                  // neighbours per LDS tile: 32 (table row pitch MTP_PITCH doubles)
  int tab_rows;            // table rows = 2*nslot + 3*P (candidate-vector kernel: 4*P + R)
  // per-atom LDS image (mtp_kernels.hip, WaveLds): offsets in doubles from the start of the wavefront's region
  int dg_mode;             // bit 0 "nodg": no dg rows -- Mu rows f'_mu written ahead of the force phase from radial derivatives
                           // parked in registers (Mu <= 4); bit 1 "rebuild": g (and dg) rows built a second time
  int fp_row;              // nodg: table row of f'_0 (rows fp_row .. fp_row + Mu - 1), behind the coefficient blocks
  int rebuild_tables;      // dg_mode & 2 (kept for the launch-info report)
  int pow_row;             // first coordinate-power row of the table: 2*nslot (keep) or nslot
  int dg_off;              // from a g row to its dg row
  int w_m, w_d, w_coef, w_nb;   // moments, adjoints, derivative-polynomial coefficients, neighbour arrays
  int wps;                 // register build: 2 or 3 wavefronts per SIMD (mtp_wave_kernel's WPS)
  int row0;                // first row of ilist / first this launch works on (inum = rows of this launch)
  int ov_doubles;          // force kernel: doubles of the overlay = max(3*P*MTP_PITCH, m_doubles + d_doubles)
  int cj_cap;              // capacity of the compacted id list
  int wave_doubles;        // LDS doubles per wavefront
  int m_doubles;           // doubles of the moment region = max(A, coef_total, 16)
  int d_doubles;           // doubles of the adjoint region = A
  float inv_mu;            // 1 / Mu
};

// lane-grid shape of the candidate-vector kernel for B basics: KL k-lanes x KB basics per lane; -1 when B is too large
int mtp_pick_shape(int B, int *KL, int *KB);
// lane-grid shape of the force kernel's basic-moment pass: KL lanes x NB 3x3 blocks per lane; -1 beyond 256 blocks
int mtp_pick_fwd_shape(int nblk, int *KL, int *NB);
hipError_t mtp_launch_wave_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st);
bool mtp_wave_kernel_has_wps3(int nfb, int P);
hipError_t mtp_launch_ev_finish(double *ev_slots, double *ev, hipStream_t st);
hipError_t mtp_launch_ev_finish_unpack(double *ev_slots, double *ev, int fold, double *f, const int *idx, const double *frecv,
                                       int n3, hipStream_t st);
// internal to the library (mtp_halo.hip): the tally fold of a step whose last force launch ran with finish_tallies = 0,
// together with the fold of the received ghost forces, in one launch
struct mtp_context;
int mtp_internal_finish_unpack(mtp_context *c, void *stream, int eflag, int vflag, double *d_ev, double *d_f, const int *d_idx,
                               const double *d_frecv, int n3);
// NULL -> the context's own stream; anything else unchanged (the one place the C ABI's NULL-stream rule is resolved)
void *mtp_internal_resolve_stream(mtp_context *c, void *stream);
hipError_t mtp_launch_fixed_to_force(long long *fq, double *f, int nall, hipStream_t st);
hipError_t mtp_launch_zero(double *p, size_t n, hipStream_t st);
// non-default compile-time tunables of mtp_kernels.hip, "NAME=value " each ("" for the shipped build)
const char *mtp_kernel_build_flags();
// radial block of cvec from dbasic (grade calls, after the force kernel)
hipError_t mtp_launch_cvec_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st);
// grades[ilist[ii]] = max_r |sum_c cvec[ii][c] Ainv[r][c]| (f64 MFMA), running maximum into max_grade;
// ainv_tiled = Ainv in MFMA operand order [cpad/16][cpad/4][64] (used when cpad <= 160), else ainv_pad [cpad][cpad]
hipError_t mtp_launch_grade_kernel(const double *cvec, const double *ainv_pad, const double *ainv_tiled, int cpad,
                                   int C, int inum, const int *ilist, double *grades, double *max_grade, hipStream_t st);
// coeff_ders[c] += sum_ii cvec[ii][c]
hipError_t mtp_launch_colsum_kernel(const double *cvec, int cpad, int C, int inum, double *coeff_ders, hipStream_t st);
// device neighbour-list build (mtp_neighbor_kernels.hip): stage 1 (neigh == nullptr) bins, counts and scans and
// leaves {entries, longest row} in d_info[2]; stage 2 fills neigh[]
hipError_t mtp_launch_neighbor_build(const double *x, int inum, int nall, double cutoff, const double lo[3],
                                     const int ncell3[3], int *scratch, double *xs, void *cub_tmp, size_t cub_bytes, int *ilist,
                                     int *first, int *neigh, int *d_info, hipStream_t st);
size_t mtp_neighbor_scan_bytes(int n, int nall);
// LAMMPS-KOKKOS 2-D list view -> CSR (mtp_neighbor_kernels.hip): stage 1 (neigh == nullptr) counts and scans and
// leaves {entries, longest row, bad-row flag} in d_info[3]; stage 2 fills neigh[]
hipError_t mtp_launch_list_from_2d(int inum, const int *d_ilist, const int *d_numneigh, const int *d_neighbors,
                                   long long stride_i, long long stride_jj, int cap, int *counts, void *cub_tmp,
                                   size_t cub_bytes, int *first, int *neigh, int *d_info, hipStream_t st);
