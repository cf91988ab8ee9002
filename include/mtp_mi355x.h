/* libmtp_mi355x -- C ABI of the MI355X-native MTP pair-style compute path.
 *
 * This is the drop-in boundary: plain C types, opaque handles, int status
 * returns, no exceptions across the boundary.  Each entry point names the reference
 * interface (file:line relative to /root/reference) it stands in for; INTEGRATION.md
 * shows the LAMMPS `Pair` subclass / plugin stub that binds them.
 *
 * Threading: one context per rank/GPU; calls on one context must be serialised by the
 * caller (the reference is not re-entrant either: member scratch buffers,
 * LAMMPS/ML-MTP/pair_mtp.h:70-83).
 *
 * Streams.  Device work is ordered on the `stream` (a hipStream_t) given to an entry point.  ONE rule for NULL:
 *   - entry points that take a context (mtp_compute_device[_rows], mtp_build_neighbors_device,
 *     mtp_set_neighbors_device_2d, mtp_synchronize, mtp_halo_force_step, mtp_ghosts_reverse_finish): NULL means the
 *     context's own stream, resolved once per call -- every launch and RCCL group of that call runs on it;
 *   - entry points without a context (the other mtp_halo_*, mtp_ghosts_*, mtp_nve_* calls, mtp_zero_async): NULL is
 *     rejected with MTP_ERR_ARG -- there is no stream to map it to, and the legacy null stream is never used.
 * The context's stream is created NON-BLOCKING: it does not synchronise with the legacy default stream, so a caller
 * whose other GPU work runs on the default stream (PyTorch's default) must pass a real stream handle that its own
 * work is ordered on, or synchronise around the calls.
 */
#ifndef MTP_MI355X_H
#define MTP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTP_MI355X_ABI_VERSION 3

/* status codes (the reference aborts through error->one/all, pair_mtp.cpp:92,354-358;
 * the adapter turns a non-zero status + mtp_last_error() into error->all) */
enum {
  MTP_OK = 0,
  MTP_ERR_IO = -2,          /* cannot open / short read */
  MTP_ERR_EOF = -3,
  MTP_ERR_FORMAT = -4,      /* not "MTP" / wrong version */
  MTP_ERR_PARSE = -5,       /* keyword or number missing */
  MTP_ERR_UNSUPPORTED = -6, /* magnetic basis, unknown radial basis */
  MTP_ERR_TABLE = -7,       /* inconsistent alpha tables */
  MTP_ERR_SELECTION = -8,   /* #MVS block missing or malformed */
  MTP_ERR_MODE = -9,        /* energy_weight + site_en_weight > 1 */
  MTP_ERR_ARG = -20,
  MTP_ERR_DEVICE = -21,     /* HIP runtime failure, no device, wrong architecture */
  MTP_ERR_SPECIES = -22,    /* atom type outside the potential (pair_mtp.cpp:91-93,116-118) */
  MTP_ERR_STATE = -23,      /* e.g. compute before set_neighbors, grades without #MVS */
  MTP_ERR_LIMIT = -24       /* table or neighbour count beyond what one wave's LDS holds */
};

/* eflag / vflag bits follow LAMMPS (pair.h: ENERGY_GLOBAL 1, ENERGY_ATOM 2,
 * VIRIAL_PAIR 1, VIRIAL_FDOTR 2, VIRIAL_ATOM 4).  PairMTP::compute tests the raw vflag
 * (pair_mtp.cpp:257): any non-zero vflag tallies the global virial, bit 4 also vatom. */
#define MTP_ENERGY_GLOBAL 1
#define MTP_ENERGY_ATOM 2
#define MTP_VIRIAL_ATOM 4

/* kernel variant: the reference exposes two GPU styles, `mtp/kk` (thread-parallel,
 * KOKKOS/pair_mtp_kokkos.h:20-22) and `mtp/small/kk` (block-parallel,
 * KOKKOS/pair_mtps_kokkos.h:20-22).  Both map onto one wavefront-per-atom design here;
 * the variant only picks how many wavefronts share a workgroup. */
enum { MTP_VARIANT_AUTO = 0, MTP_VARIANT_LARGE = 1, MTP_VARIANT_SMALL = 2 };

typedef struct mtp_potential mtp_potential; /* parsed MLIP-3 file: PairMTP model state, pair_mtp.h:47-83 */
typedef struct mtp_context mtp_context;     /* one GPU: device tables, neighbour list, workspaces */

typedef struct mtp_potential_info {
  int32_t species_count;
  int32_t radial_basis_size;       /* R  */
  int32_t radial_func_count;       /* Mu */
  int32_t alpha_moment_count;      /* A  */
  int32_t alpha_index_basic_count; /* B  */
  int32_t alpha_index_times_count; /* T  */
  int32_t alpha_scalar_count;      /* S  */
  int32_t max_alpha_index_basic;   /* P  */
  int32_t coeff_count;             /* C = Sp^2 Mu R + Sp + S (pair_mtp_extrapolation.cpp:533) */
  int32_t has_selection;           /* a #MVS_v1.1 block was read */
  int32_t configuration_mode;      /* energy_weight == 1 (pair_mtp_extrapolation.cpp:605) */
  int32_t product_levels;          /* dependency levels of the times table (native schedule) */
  double scaling;
  double min_cutoff;
  double max_cutoff;               /* what PairMTP::init_one returns, pair_mtp.cpp:325-330 */
} mtp_potential_info;

/* PairMTP::read_file (pair_mtp.cpp:335-570) + RadialMTPBasis::ReadBasisProperties
 * (mtp_radial_basis.cpp:59-102); with want_selection != 0 also
 * PairMTPExtrapolation::read_file (pair_mtp_extrapolation.cpp:528-612).  Host only. */
int mtp_potential_load(const char *path, int want_selection, mtp_potential **out, char *err, int errlen);
void mtp_potential_free(mtp_potential *pot);
int mtp_potential_get_info(const mtp_potential *pot, mtp_potential_info *info);
/* copies of the parsed tables for inspection (sizes from get_info); any pointer may be NULL */
int mtp_potential_get_tables(const mtp_potential *pot, int32_t *alpha_index_basic /*[B][4]*/,
                             int32_t *alpha_index_times /*[T][4]*/, int32_t *alpha_moment_mapping /*[S]*/,
                             double *radial_coeffs /*[Sp*Sp*Mu*R]*/, double *species_coeffs /*[Sp]*/,
                             double *moment_coeffs /*[S]*/, double *inverse_active_set /*[C*C]*/);

/* The device copies made in PairMTPKokkos::settings (KOKKOS/pair_mtp_kokkos.cpp:108-174).
 * Fails with MTP_ERR_DEVICE when no gfx950 device is usable: there is no CPU fallback. */
int mtp_context_create(const mtp_potential *pot, int device_id, mtp_context **out, char *err, int errlen);
void mtp_context_destroy(mtp_context *ctx);
const char *mtp_last_error(const mtp_context *ctx);
int mtp_context_set_variant(mtp_context *ctx, int variant);

/* The neighbour list PairMTP::compute reads (pair_mtp.cpp:81-85): call after every
 * LAMMPS re-neighbouring.  `firstneigh[i]` is indexed by atom index i = ilist[ii], as in
 * LAMMPS; entries are masked with NEIGHMASK on the device. */
int mtp_set_neighbors(mtp_context *ctx, int inum, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, int nall);
/* Same list in CSR form: neighbours of ilist[ii] are neigh[first[ii] .. first[ii+1]). */
int mtp_set_neighbors_csr(mtp_context *ctx, int inum, const int *ilist, const int *first, const int *neigh,
                          int nall);
/* CSR arrays already resident in HBM (no copy; must stay valid until replaced). */
int mtp_set_neighbors_device(mtp_context *ctx, int inum, const int *d_ilist, const int *d_first,
                             const int *d_neigh, int nall, int max_numneigh);
/* The list as LAMMPS-KOKKOS holds it on the device: the `/kk` styles of the reference read k_list->d_ilist(ii),
 * d_numneigh(i) and the padded 2-D view d_neighbors(i, jj) (KOKKOS/pair_mtp_kokkos.cpp:236-239, pair_mtp_kokkos.h:115;
 * `FindMaxNumNeighs`, pair_mtp_kokkos.cpp:177-191, 254-256).  Element (i, jj) is d_neighbors[i * stride_i + jj *
 * stride_jj] -- LayoutLeft (the GPU default): stride_i = 1, stride_jj = extent(0); LayoutRight: stride_i = extent(1),
 * stride_jj = 1 -- rows are indexed by atom index i = d_ilist[ii], max_neighs = extent(1).  Two device kernels and a
 * scan compact the view into the context's CSR arrays (special-bond bits are kept and masked with NEIGHMASK in the
 * force kernel, pair_mtp.cpp:114); d_ilist is used in place and must stay valid until the next list is installed.
 * Nothing passes through the host except one 12-byte read-back (entry count, longest row), which synchronises
 * `stream` once per re-neighbouring. */
int mtp_set_neighbors_device_2d(mtp_context *ctx, void *stream, int inum, const int *d_ilist, const int *d_numneigh,
                                const int *d_neighbors, long long stride_i, long long stride_jj, int max_neighs,
                                int nall);
/* Builds that list on the GPU from positions resident in HBM (what LAMMPS' Neighbor class
 * does ahead of the pair style, REQ_FULL at pair_mtp.cpp:317-318): a full list for atoms [0, inum) over all
 * nall atoms (owned first, then explicit ghosts -- no periodic images are invented), entries j != i with
 * |x_j - x_i|^2 <= list_cutoff^2.  lo / hi bound the positions of all nall atoms.  The list stays in the
 * context (ilist = 0..inum-1) and is installed as by mtp_set_neighbors_device; d_first_out / d_neigh_out (may be
 * null) receive device pointers to the CSR arrays, total_out / max_numneigh_out their sizes.  Synchronises the
 * stream once (to size the entry array). */
int mtp_build_neighbors_device(mtp_context *ctx, void *stream, const double *d_x, int inum, int nall,
                               double list_cutoff, const double lo[3], const double hi[3],
                               const int **d_first_out, const int **d_neigh_out, long long *total_out,
                               int *max_numneigh_out);
/* Copies the CSR arrays of the list the context owns (uploaded by mtp_set_neighbors[_csr] or built by
 * mtp_build_neighbors_device) back to the host: first[inum + 1], neigh[first[inum]].  MTP_ERR_STATE when the
 * current list lives in caller memory (mtp_set_neighbors_device). */
int mtp_copy_neighbors_to_host(mtp_context *ctx, int *first, int *neigh);

/* PairMTP::compute (pair_mtp.cpp:72-280) and, with grade_flag != 0,
 * PairMTPExtrapolation::compute (pair_mtp_extrapolation.cpp:68-382), on host arrays laid
 * out as LAMMPS lays them out: x, f [nall][3]; type [nall] 1-based; eatom [nall];
 * vatom [nall][6].  f, virial and vatom ACCUMULATE, eatom[i] is assigned for i in ilist,
 * *energy accumulates (eng_vdwl), exactly as in the reference.  grades[i] (neighbourhood
 * mode) is assigned for i in ilist; *max_grade is this rank's maximum (neighbourhood) or
 * is left to mtp_cfg_grade (configuration mode, where coeff_ders[C] receives this rank's
 * sum_i dE_i/dtheta for the caller to all-reduce, pair_mtp_extrapolation.cpp:369).
 * Unused outputs may be NULL.  Copies go over PCIe; use the _device form to avoid them. */
int mtp_compute(mtp_context *ctx, const double *x, const int *type, int eflag, int vflag, int grade_flag,
                double *f, double *eatom, double *vatom, double *energy, double *virial /*[6]*/,
                double *grades, double *max_grade, double *coeff_ders);

/* Same, device pointers, asynchronous on `stream` (a hipStream_t, NULL = context stream).
 * d_ev[7] accumulates {energy, virial xx,yy,zz,xy,xz,yz}; d_max_grade[1] is max-updated
 * (caller zeroes); d_coeff_ders[C] accumulates.  The atom-type error is reported by the next
 * mtp_synchronize(). */
int mtp_compute_device(mtp_context *ctx, void *stream, const double *d_x, const int *d_type, int eflag,
                       int vflag, int grade_flag, double *d_f, double *d_eatom, double *d_vatom,
                       double *d_ev, double *d_grades, double *d_max_grade, double *d_coeff_ders);
/* The same on rows [row_begin, row_begin + row_count) of the installed list only (rows = positions in ilist).  A
 * domain-decomposed step splits its owned atoms into those whose list holds no ghost and the rest, so that the
 * former are computed while the ghost positions are still in flight (what LAMMPS-KOKKOS does with its
 * interior / boundary kernels; reference anchor for the semantics: pair_mtp.cpp:252-254, 315).  Energy and virial
 * keep accumulating in the context's tally slots over the launches of a step; the launch with finish_tallies != 0
 * folds them into d_ev (the others may pass d_ev = NULL).  Grades / candidate vectors of the rows are produced
 * per launch. */
int mtp_compute_device_rows(mtp_context *ctx, void *stream, int row_begin, int row_count, int finish_tallies,
                            const double *d_x, const int *d_type, int eflag, int vflag, int grade_flag,
                            double *d_f, double *d_eatom, double *d_vatom, double *d_ev, double *d_grades,
                            double *d_max_grade, double *d_coeff_ders);
int mtp_synchronize(mtp_context *ctx, void *stream);

/* ---- device-resident step: what the reference's /kk styles do around their kernels --------------------------------
 * PairMTPKokkos::compute (KOKKOS/pair_mtp_kokkos.cpp:197-399) and PairMTPExtrapolationKokkos::compute
 * (KOKKOS/pair_mtp_extrapolation_kokkos.cpp:274-610) read x / type and write f through LAMMPS-KOKKOS device views
 * (:231-240) and keep every other output -- eng_vdwl / virial (`ev`), d_eatom, d_vatom, the grades -- in device views
 * that are copied to the host only when something asks (k_eatom / k_vatom sync :379-388; grades :223-243).  Here those
 * views belong to the context:
 *   mtp_compute_resident   zeroes the totals and the requested per-atom arrays on `stream`, then runs the force call of
 *                          mtp_compute_device (same flags, same accumulate / assign semantics) into them; nothing
 *                          crosses PCIe and nothing waits;
 *   mtp_resident_totals    the ONE wait of a step: ev7 = {energy, virial xx,yy,zz,xy,xz,yz} of that call, this rank's
 *                          maximum grade (neighbourhood mode) and, in configuration mode, sum_i dE_i/dtheta (C doubles,
 *                          may be NULL); reports the atom-type error like mtp_synchronize;
 *   mtp_resident_peratom_device / _host   the per-atom arrays of that call -- eatom [nall], vatom [nall][6], grades
 *                          [nall] -- as a device pointer (valid until the next call that grows them) or copied to the
 *                          host on request (what `fix pair` / `dump` trigger through extract_peratom,
 *                          pair_mtp_extrapolation.cpp:641-652). */
enum { MTP_PERATOM_EATOM = 0, MTP_PERATOM_VATOM = 1, MTP_PERATOM_GRADES = 2 };
int mtp_compute_resident(mtp_context *ctx, void *stream, const double *d_x, const int *d_type, double *d_f, int eflag,
                         int vflag, int grade_flag);
int mtp_resident_totals(mtp_context *ctx, void *stream, double *ev7, double *max_grade, double *coeff_ders);
int mtp_resident_peratom_device(mtp_context *ctx, int what, const double **d_ptr, int *ncol);
int mtp_resident_peratom_host(mtp_context *ctx, void *stream, int what, double *host);
/* `bytes` from a device array to the host, ordered on `stream` and waited for (atomKK->sync(Host, ...) for the few
 * rows a .cfg record needs, pair_mtp_extrapolation.cpp:401-479, when positions live on the device) */
int mtp_copy_to_host(mtp_context *ctx, void *stream, void *host, const void *d_src, size_t bytes);

/* PairMTPExtrapolation::calculate_extrapolation_grade (pair_mtp_extrapolation.cpp:347-358)
 * for configuration mode: max_i |sum_j coeff_ders[j] A^-1[i][j]| on the host (C^2 flops,
 * once per step after the cross-rank sum). */
int mtp_cfg_grade(const mtp_potential *pot, const double *coeff_ders, double *grade);

/* Compile-time switches of this build that differ from the shipped defaults, space separated ("" for a release
 * build: tests assert that the library they load carries none -- diagnostic variants are never shipped). */
const char *mtp_build_flags(void);
/* introspection for benchmarks: LDS bytes per wavefront, wavefronts per workgroup, grid */
int mtp_context_launch_info(const mtp_context *ctx, int32_t *lds_bytes_per_wave, int32_t *waves_per_block,
                            int32_t *grid_blocks, int32_t *neighbor_tile);
/* d_p[0, n) = 0.0 in one kernel launch on `stream` (d_p 16-byte aligned): the "zero the force array" that precedes
 * every force call (LAMMPS: Verlet::force_clear) without hipMemsetAsync's two fill kernels. */
int mtp_zero_async(void *stream, double *d_p, long long n);
/* Deterministic force sums (tests, reproducible goldens; SURVEY.md section 5 "deterministic-reduction mode"): the
 * scatter f_j -= F_ij and the per-atom totals are accumulated as 64-bit fixed-point integers (2^-40 eV/A, |f| < 2^23)
 * and converted once, so two calls on the same input return the same bits; energy and virial are folded in a fixed
 * order in either mode.  Default off: native fp64 HBM atomics, whose sums depend on arrival order in the last bits
 * (as the reference's own Kokkos atomics do, KOKKOS/pair_mtp_kokkos.cpp:602-605). */
int mtp_context_set_deterministic(mtp_context *ctx, int enable);
/* register build the planner chose (2 or 3 wavefronts per SIMD) and whether the per-atom LDS image uses the
 * "rebuild" layout (radial tables built twice, moments overlaying them) */
int mtp_context_plan_info(const mtp_context *ctx, int32_t *waves_per_simd, int32_t *rebuild_tables);
/* last kernel time of the dominant kernel in ms, measured with HIP events on the launch
 * stream (enable with mtp_context_set_timing(ctx, 1); costs one event pair per call) */
int mtp_context_set_timing(mtp_context *ctx, int enable);
int mtp_context_last_kernel_ms(mtp_context *ctx, float *ms);

/* ---- multi-GPU halo: spatial domain decomposition, one process per GPU, RCCL over xGMI ------------------------
 *
 * The reference leaves the ghost-atom exchange to LAMMPS' Comm class (forward_comm of x before Pair::compute,
 * reverse_comm of f after it); it only relies on it: forces are written onto ghosts (pair_mtp.cpp:252-254) and
 * newton_pair must be on (pair_mtp.cpp:315).  A standalone driver (or a KOKKOS-resident LAMMPS that hands over
 * device views) gets the same two exchanges from the library: device pack / unpack kernels around ONE grouped
 * RCCL exchange per direction (ncclGroupStart, ncclSend / ncclRecv to every peer, ncclGroupEnd) on the halo's own
 * stream, ordered with the caller's stream by events, so interior force work overlaps the exchange.
 *
 * Layout contract: atoms [0, nlocal) are owned, ghosts follow, grouped by the rank that owns them, in rank order
 * (recv_counts[q] ghosts from rank q).  send_idx lists, grouped by destination rank in rank order (send_counts[q]
 * entries for rank q), the owned atoms each peer holds as ghosts, in the order that peer stores them; send_shift
 * is the periodic shift added to their coordinates on the way.  A rank may be its own peer (periodic images).
 */
typedef struct mtp_halo mtp_halo;
#define MTP_HALO_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
enum { MTP_REDUCE_SUM = 0, MTP_REDUCE_MAX = 1 };

/* ncclGetUniqueId: called on one rank; the caller passes the bytes to every rank (MPI_Bcast, a TCP store, a file) */
int mtp_halo_get_unique_id(void *id_out /*[MTP_HALO_ID_BYTES]*/);
/* Host only, no device: the per-peer segment tables mtp_halo_create derives from the layout contract above -- peer q's
 * segment starts at atom send_off[q] of the packed send buffer and at ghost recv_off[q]; arrays of nranks + 1 entries
 * (last = totals) -- with the same checks (counts add up, send_idx inside the owned atoms).  These are the offsets the
 * grouped ncclSend / ncclRecv of a direction use (the counterpart of LAMMPS' Comm sendlist / firstrecv bookkeeping
 * the reference relies on, pair_mtp.cpp:252-254, 315). */
int mtp_halo_layout(int nranks, int nlocal, int nghost, const int *send_idx, const int *send_counts,
                    const int *recv_counts, int *send_off /*[nranks+1]*/, int *recv_off /*[nranks+1]*/, char *err,
                    int errlen);
/* ncclCommInitRank + device copies of the index lists: collective over all nranks processes.  unique_id == NULL
 * creates the halo WITHOUT a communicator (no collective call): it packs, unpacks and answers mtp_halo_get_layout,
 * and its segments are moved by mtp_halo_local_exchange or by the caller; the RCCL entry points then fail. */
int mtp_halo_create(int device_id, int nranks, int rank, const void *unique_id, int nlocal, int nghost,
                    const int *send_idx /*[sum send_counts]*/, const double *send_shift /*[sum send_counts][3]*/,
                    const int *send_counts /*[nranks]*/, const int *recv_counts /*[nranks]*/, mtp_halo **out,
                    char *err, int errlen);
void mtp_halo_destroy(mtp_halo *halo);
const char *mtp_halo_last_error(const mtp_halo *halo);
/* what RCCL itself reports for the communicator (ncclCommCount, ncclCommUserRank, ncclGetVersion) */
int mtp_halo_comm_count(const mtp_halo *halo, int *nranks, int *rank, int *rccl_version);
/* Comm::forward_comm: d_x[nlocal + k] <- owner's x + shift.  begin: pack on `stream`, exchange on the halo's
 * stream; end: `stream` waits for the exchange.  Work queued on `stream` in between overlaps it. */
int mtp_halo_forward_begin(mtp_halo *halo, void *stream, double *d_x /*[nall][3]*/);
int mtp_halo_forward_end(mtp_halo *halo, void *stream);
int mtp_halo_forward(mtp_halo *halo, void *stream, double *d_x);
/* Comm::reverse_comm: ghost rows of d_f are sent back and added onto their owners (fp64 atomics).  begin: the
 * exchange starts once `stream` has reached this point (every launch that writes ghost forces must precede it);
 * end: `stream` waits, then adds the received rows into d_f[0, nlocal). */
int mtp_halo_reverse_begin(mtp_halo *halo, void *stream, const double *d_f /*[nall][3]*/);
int mtp_halo_reverse_end(mtp_halo *halo, void *stream, double *d_f);
int mtp_halo_reverse(mtp_halo *halo, void *stream, double *d_f);
/* One domain-decomposed force call (Comm::forward_comm, Pair::compute, Comm::reverse_comm of a LAMMPS step,
 * pair_mtp.cpp:252-254, 315): zero d_f, ghost positions in, forces of the rows_a + rows_b + rows_c = inum rows, ghost
 * forces back onto their owners, tallies folded into d_ev by the last force launch.  Default schedule: everything
 * on `stream` -- pack, forward group, one launch over all rows, reverse group, unpack (measured faster on MI355X than
 * the overlapped one at every domain size tried).  mtp_halo_set_overlap(halo, 1): both exchanges overlapped -- the
 * installed list must then be ordered interior | boundary | interior (interior = no ghost in the atom's list): forward
 * halo || rows [0, rows_a), boundary rows, reverse halo || the last rows_c rows, on `stream` and the halo's stream. */
int mtp_halo_force_step(mtp_halo *halo, mtp_context *ctx, void *stream, int rows_a, int rows_b, int rows_c,
                        double *d_x, const int *d_type, int eflag, int vflag, int grade_flag, double *d_f,
                        double *d_eatom, double *d_vatom, double *d_ev, double *d_grades, double *d_max_grade,
                        double *d_coeff_ders);
int mtp_halo_set_overlap(mtp_halo *halo, int enable);
int mtp_halo_get_overlap(const mtp_halo *halo);
/* The kernels either side of an exchange on their own: sendbuf[k] = d_x[send_idx[k]] + send_shift[k] (what
 * mtp_halo_forward_begin launches ahead of its group) and d_f[send_idx[k]] += frecv[k] (what mtp_halo_reverse_end
 * launches behind its group). */
int mtp_halo_pack_forward(mtp_halo *halo, void *stream, const double *d_x);
int mtp_halo_unpack_reverse(mtp_halo *halo, void *stream, double *d_f);
/* the tables of this halo as the exchange uses them (arrays of nranks + 1 / nranks entries; any may be NULL) */
int mtp_halo_get_layout(const mtp_halo *halo, int *nsend, int *send_off, int *send_counts, int *recv_off,
                        int *recv_counts);
/* Single-process rehearsal of an n-rank exchange: halos[r] = rank r of ONE n-rank decomposition, all on one device
 * (created with or without a communicator).  direction 0 = forward: after mtp_halo_pack_forward on every rank, copies
 * every (source q, destination r) segment into the ghost rows of d_arrays[r] (= rank r's positions [nall_r][3]);
 * direction 1 = reverse: copies the ghost rows of d_arrays[r] (= rank r's forces) into the owners' receive buffers,
 * to be folded by mtp_halo_unpack_reverse.  Device-to-device copies on `stream`, addressed with the same per-peer
 * offset tables the RCCL groups use; fails when the two sides of a segment disagree on its length. */
int mtp_halo_local_exchange(mtp_halo *const *halos, int n, void *stream, int direction, double *const *d_arrays);
/* in-place ncclAllReduce of `count` doubles: energy / virial and the configuration-mode candidate vector (SUM,
 * pair_mtp_extrapolation.cpp:369), the neighbourhood-mode maximum grade (MAX, :379) */
int mtp_halo_allreduce(mtp_halo *halo, void *stream, double *d_buf, int count, int op);

/* ---- standalone MD support: LAMMPS-core work either side of Pair::compute, on the device ------
 *
 * For drivers that keep the whole step in HBM (bench.py's whole-step number, lammps_mtp_kokkos_amd/md.py): the
 * periodic ghost images of ONE GPU's own atoms (Comm::borders / forward_comm / reverse_comm of a single rank; the
 * pair style needs them because it writes forces onto ghosts, pair_mtp.cpp:252-254, 315) and the two halves of a
 * velocity-Verlet step (fix nve).  Orthogonal box [0, box), every edge >= rghost.
 */
typedef struct mtp_ghosts mtp_ghosts;
int mtp_ghosts_create(int device_id, mtp_ghosts **out);
void mtp_ghosts_destroy(mtp_ghosts *g);
const char *mtp_ghosts_last_error(const mtp_ghosts *g);
/* Re-neighbouring: wraps d_x[0, nlocal) into the box, finds every periodic image within rghost of the box (atom
 * order, then lexicographic shift order: deterministic) and writes their positions behind the owned atoms.
 * *nall_out = nlocal + ghosts; MTP_ERR_LIMIT (nothing written beyond the wrap) when that exceeds `capacity` rows.
 * Synchronises the stream once (the ghost count sizes the caller's arrays and the neighbour list). */
int mtp_ghosts_build(mtp_ghosts *g, void *stream, double *d_x /*[capacity][3]*/, int nlocal, int capacity,
                     const double box[3], double rghost, int *nall_out);
int mtp_ghosts_forward(mtp_ghosts *g, void *stream, double *d_x);   /* ghost rows <- owner + shift            */
int mtp_ghosts_reverse(mtp_ghosts *g, void *stream, double *d_f);   /* owner rows += ghost rows (fp64 atomics) */
/* The same together with the energy / virial fold of a force call made through mtp_compute_device_rows(...,
 * finish_tallies = 0, ...): one launch instead of two (d_ev as in mtp_compute_device; eflag / vflag of that call). */
int mtp_ghosts_reverse_finish(mtp_ghosts *g, mtp_context *ctx, void *stream, int eflag, int vflag, double *d_f, double *d_ev);
int mtp_ghosts_types(mtp_ghosts *g, void *stream, int *d_type);     /* ghost types <- owner types              */
/* fix nve (metal units: dtf = 0.5 dt ftm2v): v += dtf f / m; x += dt v   and   v += dtf f / m; masses per type */
int mtp_nve_initial(void *stream, int nlocal, double *d_x, double *d_v, const double *d_f, const int *d_type,
                    const double *d_inv_mass, double dtf, double dt);
int mtp_nve_final(void *stream, int nlocal, double *d_v, const double *d_f, const int *d_type,
                  const double *d_inv_mass, double dtf);
/* d_out2[0] = max_i |x_i - x_ref_i|^2 (the half-skin re-neighbouring test), d_out2[1] = sum_i m_i v_i^2 */
int mtp_nve_monitor(void *stream, int nlocal, const double *d_x, const double *d_x_ref, const double *d_v,
                    const int *d_type, const double *d_mass, double *d_out2);

#ifdef __cplusplus
}
#endif
#endif
