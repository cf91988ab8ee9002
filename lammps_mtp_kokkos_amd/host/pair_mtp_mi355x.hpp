// Host-side mirror of the reference's pair-style interface, above the C ABI
// (include/mtp_mi355x.h), for callers that are not LAMMPS: same method names, argument
// grammar and error text as LAMMPS_NS::PairMTP / PairMTPExtrapolation
// (/root/reference/LAMMPS/ML-MTP/pair_mtp.h:34-40, pair_mtp_extrapolation.h:35-38).
// Where LAMMPS would call error->all/one, these throw mtp_mi355x::Error with the same
// message.  The LAMMPS plugin adapter (lammps_plugin/) is the same logic behind a real
// `Pair` subclass.
#pragma once

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mtp_mi355x.h"
#include "mtp_cfg_writer.hpp"

namespace mtp_mi355x {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// What the pair style reads from LAMMPS each step (pair_mtp.cpp:77-85): plain pointers, no ownership.
struct AtomView {
  const double *x = nullptr;   // [nall][3]
  double *f = nullptr;         // [nall][3], accumulated
  const int *type = nullptr;   // [nall], 1-based
  int nlocal = 0, nall = 0;
  long natoms = 0;             // global atom count (atom->natoms)
};
struct NeighListView {
  int inum = 0;
  const int *ilist = nullptr;
  const int *numneigh = nullptr;
  const int *const *firstneigh = nullptr;
};
// What a KOKKOS-resident LAMMPS hands the /kk styles instead (KOKKOS/pair_mtp_kokkos.cpp:231-240: atomKK->sync,
// x = atomKK->k_x.view<DeviceType>() ...; :236-239: k_list->d_numneigh, d_neighbors, d_ilist): device pointers and the
// stream of the execution space.  ALL device work of a step -- the zeroing of the tallies, the force call, the copies
// behind extract_peratom -- is ordered on `stream` (no second stream, no legacy null stream).
struct DeviceAtomView {
  const double *d_x = nullptr;   // [nall][3]
  double *d_f = nullptr;         // [nall][3], accumulated
  const int *d_type = nullptr;   // [nall], 1-based
  int nlocal = 0, nall = 0;
  long natoms = 0;
  void *stream = nullptr;        // hipStream_t of the execution space (NULL: the context's own stream)
  // host arrays LAMMPS reads per-atom results from (eatom / vatom of Pair); filled only when the flags ask, as the
  // reference syncs k_eatom / k_vatom to the host only then (pair_mtp_kokkos.cpp:379-388)
};
struct DeviceNeighListView {     // padded 2-D view: element (i, jj) at d_neighbors[i * stride_i + jj * stride_jj]
  int inum = 0;
  const int *d_ilist = nullptr, *d_numneigh = nullptr, *d_neighbors = nullptr;
  long long stride_i = 0, stride_jj = 0;
  int maxneighs = 0;             // extent(1) of d_neighbors
};
using BoxView = CfgBox;   // domain->xprd ... for the .cfg writer (pair_mtp_extrapolation.cpp:449-451)
// cross-rank reductions (LAMMPS: MPI_Allreduce on `world`); identity when unset (one rank)
struct Reductions {
  void (*sum)(double *buf, int n, void *ctx) = nullptr;
  void (*max)(double *buf, int n, void *ctx) = nullptr;
  void *ctx = nullptr;
  int me = 0, nprocs = 1;
  // the .cfg writer's exchanges (mtp_cfg_writer.hpp: MPI_Scan, MPI_Send / MPI_Recv to rank 0); me / nprocs / ctx are
  // taken from above
  int (*scan_sum)(int value, void *ctx) = nullptr;
  void (*send_to_root)(const char *buf, size_t n, void *ctx) = nullptr;
  void (*recv_on_root)(int src, std::string &out, void *ctx) = nullptr;
};
// utils::logmesg stand-in: the reference reports the scaling, the species count and the extrapolation mode from rank 0
// (pair_mtp.cpp:383, 389; pair_mtp_extrapolation.cpp:508-517).  Default: stdout.
struct LogSink {
  void (*write)(const char *msg, void *ctx) = nullptr;
  void *ctx = nullptr;
};

// pair_style mtp <file>            (pair_mtp.cpp:285-297)
// pair_style mtp/kk <file> chunksize <N>, mtp/small/kk ... (KOKKOS/pair_mtp_kokkos.cpp:108-117):
// the chunk size only bounded the reference's spilled Jacobian; it is parsed and ignored.
class PairMTP {
 public:
  enum Style { MTP, MTP_KK, MTP_SMALL_KK };
  explicit PairMTP(Style style = MTP, int device = 0);
  virtual ~PairMTP();
  PairMTP(const PairMTP &) = delete;
  PairMTP &operator=(const PairMTP &) = delete;

  virtual void settings(int narg, char **arg);
  void coeff(int narg, char **arg);     // only "pair_coeff * *" (pair_mtp.cpp:303-307)
  void init_style(int newton_pair);     // needs newton_pair on, full list (pair_mtp.cpp:313-319)
  double init_one(int i, int j);        // returns the cutoff (pair_mtp.cpp:325-330)
  virtual void compute(int eflag, int vflag);

  // bindings in place of the LAMMPS pointers
  void bind(const AtomView &a)
  {
    atom = a;
    resident_ = false;
  }
  void set_neighbor_list(const NeighListView &l);   // call after every re-neighbouring
  // the /kk styles' data path: positions, forces, types and the neighbour list stay on the device
  void bind_device(const DeviceAtomView &a);
  void set_neighbor_list_device(const DeviceNeighListView &l);
  bool resident() const { return resident_; }
  void set_log(const LogSink &l) { log_ = l; }
  void set_rank(int me) { me_ = me; }               // comm->me: only rank 0 logs

  // what LAMMPS reads back (pair.h)
  double eng_vdwl = 0.0, virial[6] = {0, 0, 0, 0, 0, 0};
  std::vector<double> eatom, vatom;   // [nall], [nall][6]; filled when the flags ask
  int single_enable = 0, restartinfo = 0, one_coeff = 1, manybody_flag = 1;   // pair_mtp.cpp:37-40
  mtp_potential_info info{};

 protected:
  void ev_setup(int eflag, int vflag);
  void require(int rc, const char *what);
  void load(const char *file, bool selection);
  void logmesg(const std::string &msg) const;
  LogSink log_;
  int me_ = 0;
  Style style_;
  int device_;
  mtp_potential *pot_ = nullptr;
  mtp_context *ctx_ = nullptr;
  AtomView atom;
  DeviceAtomView datom;
  bool resident_ = false;   // bound to device views: compute() runs mtp_compute_resident
  void compute_resident(int eflag, int vflag, int grade, double *max_grade, double *coeff_ders);
  bool list_set_ = false;
  int eflag_either = 0, eflag_global = 0, eflag_atom = 0, vflag_either = 0, vflag_global = 0, vflag_atom = 0;
};

// pair_style mtp/extrapolation <file> [<out> <select> <break>] [chunksize <N>]
// (pair_mtp_extrapolation.cpp:485-523)
class PairMTPExtrapolation : public PairMTP {
 public:
  explicit PairMTPExtrapolation(Style style = MTP, int device = 0);
  ~PairMTPExtrapolation() override;
  void settings(int narg, char **arg) override;
  void compute(int eflag, int vflag) override;
  void *extract(const char *str, int &dim);           // "extrapolation_flag" (:624-631)
  void *extract_peratom(const char *str, int &ncol);  // "extrapolation" (:641-652)
  void set_reductions(const Reductions &r)
  {
    red = r;
    me_ = r.me;
  }
  void set_box(const BoxView &b) { box = b; }

  int nextra = 1;
  double pvector[1] = {0.0};    // max grade, rank 0 only (:381)
  int extrapolation_flag = 0;   // set by `fix pair` through extract()
  double max_grade = 0.0;

 private:
  void compile_grades();     // :363-382
  void evaluate_grades();    // :387-397
  void write_config();       // :401-479
  bool mlip3_style = false, configuration_mode = false;
  double select_threshold = 0, break_threshold = 0;
  std::FILE *preselected_file = nullptr;
  std::vector<double> nbh_extrapolation_grades, energy_ders_wrt_coeffs;
  bool grades_on_device_ = false;   // the last grade call left the grades in the context's HBM: copied on extract_peratom
  Reductions red;
  BoxView box;
};

}   // namespace mtp_mi355x
