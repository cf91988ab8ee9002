// LAMMPS plugin adapter: registers pair styles `mtp`, `mtp/kk`, `mtp/small/kk`,
// `mtp/extrapolation`, `mtp/extrapolation/kk`, `mtp/extrapolation/small/kk` at run time
// (`plugin load libmtp_mi355x_lammps.so`) and forwards the Pair virtuals the reference overrides
// (/root/reference/LAMMPS/ML-MTP/pair_mtp.h:34-40, pair_mtp_extrapolation.h:35-38).
//
// Thin by construction: every decision -- argument grammar, thresholds, grade steps, the .cfg record, the
// device-resident /kk data path, which flags tally what -- lives in the host mirror (../host/pair_mtp_mi355x.{hpp,cpp}),
// which IS compiled and exercised on the GPU by tests/cpp/test_pair_host.cpp.  This file only moves LAMMPS' pointers
// into the mirror's views and its results back, and turns mtp_mi355x::Error into error->all / error->one.
//
// COMPILE-GATED: needs a LAMMPS source tree (not present in this repository's build image):
//   hipcc/g++ -std=c++17 -fPIC -shared -DLAMMPS_SOURCE_DIR_AVAILABLE [-DLMP_KOKKOS] -I$LAMMPS_SOURCE_DIR/src
//       [-I$LAMMPS_SOURCE_DIR/src/KOKKOS -I<kokkos include dirs>] -I$REPO/include pair_mtp_mi355x_plugin.cpp
//       -L$REPO/lammps_mtp_kokkos_amd -lpair_mtp_mi355x -lmtp_mi355x -o libmtp_mi355x_lammps.so
// It has not been compiled against LAMMPS here; INTEGRATION.md lists what to check first.  What IS done here: the
// adapter is compiled against a mock of the API surface it touches (tests/cpp/lammps_mock/, test scaffolding written for
// this purpose) and driven through LAMMPS' call sequence on the GPU (tests/cpp/test_plugin_mock.cpp) -- the LMP_KOKKOS
// branch too, against a mock of the KOKKOS package's data classes on device memory (lammps_mock/kokkos_mock.h).
#ifdef LAMMPS_SOURCE_DIR_AVAILABLE

#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "lammpsplugin.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "pair.h"
#include "utils.h"
#include "version.h"
#ifdef LMP_KOKKOS
#include "atom_kokkos.h"
#include "atom_masks.h"
#include "kokkos.h"
#include "neigh_list_kokkos.h"
#endif

#include <cstring>
#include <memory>
#include <mpi.h>
#include <string>

#include "../host/pair_mtp_mi355x.hpp"

namespace LAMMPS_NS {

class PairMTPMI355X : public Pair {
  using Mirror = mtp_mi355x::PairMTP;
  using MirrorExt = mtp_mi355x::PairMTPExtrapolation;

 public:
  // kk: one of the /kk styles (argument grammar of KOKKOS/pair_mtp_kokkos.cpp:113-117 and, with LMP_KOKKOS,
  // device-resident x / f / type / neighbour list)
  PairMTPMI355X(LAMMPS *lmp, Mirror::Style style, bool ext, bool kk) : Pair(lmp), style_(style), ext_(ext), kk_(kk)
  {
    single_enable = 0;   // pair_mtp.cpp:37-40
    restartinfo = 0;
    one_coeff = 1;
    manybody_flag = 1;
    // The pair style tallies its own virial on the raw vflag (pair_mtp.cpp:257) and never calls
    // virial_fdotr_compute(); say so, as the reference's /kk styles do (KOKKOS/pair_mtp_kokkos.cpp:210), so that
    // LAMMPS passes VIRIAL_PAIR instead of VIRIAL_FDOTR.  (The mirror tallies on either.)
    no_virial_fdotr_compute = 1;
    if (ext_) {          // pair_mtp_extrapolation.cpp:42-44
      nextra = 1;
      pvector = new double[1];
      pvector[0] = 0.0;
    }
  }
  ~PairMTPMI355X() override
  {
    if (allocated) {
      memory->destroy(setflag);
      memory->destroy(cutsq);
    }
    if (ext_) delete[] pvector;
  }

  void settings(int narg, char **arg) override
  {
    // one rank per GPU: node-local rank -> device
    int ndev = 1, local = 0;
    MPI_Comm node;
    MPI_Comm_split_type(world, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node);
    MPI_Comm_rank(node, &local);
    MPI_Comm_free(&node);
    if (const char *e = getenv("MTP_MI355X_GPUS_PER_NODE")) ndev = atoi(e);
    const int device = local % (ndev > 0 ? ndev : 1);
    if (ext_) {
      auto *p = new MirrorExt(style_, device);
      impl_.reset(p);
      mtp_mi355x::Reductions red;   // compile_grades / write_config over `world` (pair_mtp_extrapolation.cpp:363-382, 401-479)
      red.ctx = this;
      red.me = comm->me;
      red.nprocs = comm->nprocs;
      red.sum = [](double *b, int n, void *c) { MPI_Allreduce(MPI_IN_PLACE, b, n, MPI_DOUBLE, MPI_SUM, ((PairMTPMI355X *) c)->world); };
      red.max = [](double *b, int n, void *c) { MPI_Allreduce(MPI_IN_PLACE, b, n, MPI_DOUBLE, MPI_MAX, ((PairMTPMI355X *) c)->world); };
      red.scan_sum = [](int v, void *c) {
        int out = 0;
        MPI_Scan(&v, &out, 1, MPI_INT, MPI_SUM, ((PairMTPMI355X *) c)->world);
        return out;
      };
      red.send_to_root = [](const char *buf, size_t n, void *c) { MPI_Send(buf, (int) n, MPI_CHAR, 0, 0, ((PairMTPMI355X *) c)->world); };
      red.recv_on_root = [](int src, std::string &out, void *c) {
        MPI_Comm w = ((PairMTPMI355X *) c)->world;
        MPI_Status st;
        int n = 0;
        MPI_Probe(src, 0, w, &st);
        MPI_Get_count(&st, MPI_CHAR, &n);
        out.resize((size_t) n);
        MPI_Recv(out.empty() ? nullptr : &out[0], n, MPI_CHAR, src, 0, w, &st);
      };
      p->set_reductions(red);
    } else {
      impl_.reset(new Mirror(style_, device));
      impl_->set_rank(comm->me);
    }
    mtp_mi355x::LogSink log;   // utils::logmesg (pair_mtp.cpp:383, 389; pair_mtp_extrapolation.cpp:508-517)
    log.ctx = lmp;
    log.write = [](const char *msg, void *c) { utils::logmesg((LAMMPS *) c, msg); };
    impl_->set_log(log);
    // the reference opens the file through utils::open_potential (pair_mtp.cpp:293, pair_mtp_extrapolation.cpp:504):
    // resolve the name the same way (LAMMPS_POTENTIALS search path) before the library opens it
    std::string path = utils::get_potential_file_path(arg[0]);
    if (path.empty()) error->all(FLERR, "Cannot open MTP potential file {}", arg[0]);
    std::vector<char *> av(arg, arg + narg);
    av[0] = &path[0];
    guard([&] { impl_->settings(narg, av.data()); });
    const int np1 = impl_->info.species_count + 1;   // pair_mtp.cpp:391-393, 455-456
    memory->create(setflag, np1, np1, "pair:setflag");
    memory->create(cutsq, np1, np1, "pair:cutsq");
    for (int i = 1; i < np1; i++)
      for (int j = 1; j < np1; j++) {
        setflag[i][j] = 1;
        cutsq[i][j] = impl_->info.max_cutoff * impl_->info.max_cutoff;
      }
    allocated = 1;
  }
  void coeff(int narg, char **arg) override
  {
    guard([&] { impl_->coeff(narg, arg); });
  }
  void init_style() override
  {
    guard([&] { impl_->init_style(force->newton_pair); });
    neighbor->add_request(this, NeighConst::REQ_FULL);   // pair_mtp.cpp:317-318
  }
  double init_one(int i, int j) override
  {
    if (setflag[i][j] == 0) error->all(FLERR, "Not all pair coeffs are set. See types {}-{}.", i, j);
    return impl_->info.max_cutoff;
  }

  void compute(int eflag, int vflag) override
  {
    ev_init(eflag, vflag);
    const int nall = atom->nlocal + atom->nghost;
    const bool relist = neighbor->ago == 0 || !list_sent_;   // the list was rebuilt this step
    if (ext_) {
      auto *p = static_cast<MirrorExt *>(impl_.get());
      p->extrapolation_flag = extrapolation_flag_;
      mtp_mi355x::BoxView b;
      b.xprd = domain->xprd;
      b.yprd = domain->yprd;
      b.zprd = domain->zprd;
      b.xy = domain->xy;
      b.xz = domain->xz;
      b.yz = domain->yz;
      p->set_box(b);
    }
#ifdef LMP_KOKKOS
    // KOKKOS-resident atoms AND list (KOKKOS/pair_mtp_kokkos.cpp:231-240): nothing crosses PCIe except the totals --
    // also on grade steps and per-atom-tally steps (the mirror copies eatom / vatom only when the flags ask and the
    // grades only on extract_peratom, as pair_mtp_extrapolation_kokkos.cpp:223-243 does).  One stream for the whole
    // step: the execution space's, so the zeroing, atomKK's syncs and the force kernel are ordered.
    if (kk_ && lmp->kokkos) {
      auto *akk = (AtomKokkos *) atom;
      akk->sync(Device, X_MASK | F_MASK | TYPE_MASK);
      mtp_mi355x::DeviceAtomView dv;
      dv.d_x = akk->k_x.d_view.data();
      dv.d_f = akk->k_f.d_view.data();
      dv.d_type = akk->k_type.d_view.data();
      dv.nlocal = atom->nlocal;
      dv.nall = nall;
      dv.natoms = (long) atom->natoms;
      dv.stream = (void *) Kokkos::HIP().hip_stream();
      impl_->bind_device(dv);
      if (relist) {
        auto *kl = static_cast<NeighListKokkos<LMPDeviceType> *>(list);
        mtp_mi355x::DeviceNeighListView lv;
        lv.inum = list->inum;
        lv.d_ilist = kl->d_ilist.data();
        lv.d_numneigh = kl->d_numneigh.data();
        lv.d_neighbors = kl->d_neighbors.data();
        lv.stride_i = (long long) kl->d_neighbors.stride(0);
        lv.stride_jj = (long long) kl->d_neighbors.stride(1);
        lv.maxneighs = (int) kl->d_neighbors.extent(1);
        guard([&] { impl_->set_neighbor_list_device(lv); });
        list_sent_ = true;
      }
      guard_one([&] { impl_->compute(eflag, vflag); });
      akk->modified(Device, F_MASK);
      collect(eflag, vflag, nall);
      return;
    }
#endif
    mtp_mi355x::AtomView av;
    av.x = &atom->x[0][0];
    av.f = &atom->f[0][0];
    av.type = atom->type;
    av.nlocal = atom->nlocal;
    av.nall = nall;
    av.natoms = (long) atom->natoms;
    impl_->bind(av);
    if (relist) {
      mtp_mi355x::NeighListView lv{list->inum, list->ilist, list->numneigh, list->firstneigh};
      guard([&] { impl_->set_neighbor_list(lv); });
      list_sent_ = true;
    }
    guard_one([&] { impl_->compute(eflag, vflag); });
    collect(eflag, vflag, nall);
  }

  void *extract(const char *str, int &dim) override
  {
    dim = 0;
    if (ext_ && strcmp(str, "extrapolation_flag") == 0) return (void *) &extrapolation_flag_;   // pair_mtp_extrapolation.cpp:624-631
    return nullptr;
  }
  void *extract_peratom(const char *str, int &ncol) override
  {
    if (!ext_) return nullptr;
    void *p = nullptr;
    guard_one([&] { p = static_cast<MirrorExt *>(impl_.get())->extract_peratom(str, ncol); });   // :641-652
    return p;
  }

 private:
  // results of the mirror into the arrays LAMMPS reads (pair.h): eng_vdwl / virial accumulate, eatom / vatom were
  // zeroed by ev_init and receive the call's values, pvector[0] on rank 0 only (pair_mtp_extrapolation.cpp:381)
  void collect(int, int, int nall)
  {
    eng_vdwl += impl_->eng_vdwl;
    for (int q = 0; q < 6; q++) virial[q] += impl_->virial[q];
    if (eflag_atom)
      for (int i = 0; i < nall; i++) eatom[i] += impl_->eatom[i];
    if (vflag_atom)
      for (int i = 0; i < nall; i++)
        for (int q = 0; q < 6; q++) vatom[i][q] += impl_->vatom[6 * (size_t) i + q];
    if (ext_ && comm->me == 0) pvector[0] = static_cast<MirrorExt *>(impl_.get())->pvector[0];
  }
  template <class F> void guard(F &&fn)   // errors every rank meets: error->all
  {
    try {
      fn();
    } catch (const mtp_mi355x::Error &e) {
      error->all(FLERR, e.what());
    }
  }
  template <class F> void guard_one(F &&fn)   // per-rank errors (device failures, the break threshold on rank 0): error->one
  {
    try {
      fn();
    } catch (const mtp_mi355x::Error &e) {
      error->one(FLERR, e.what());
    }
  }
  Mirror::Style style_;
  bool ext_, kk_, list_sent_ = false;
  int extrapolation_flag_ = 0;   // set by `fix pair` through extract()
  std::unique_ptr<Mirror> impl_;
};

}   // namespace LAMMPS_NS

using namespace LAMMPS_NS;
#define MTP_CREATOR(fn, style, ext, kk) \
  static Pair *fn(LAMMPS *lmp) { return new PairMTPMI355X(lmp, mtp_mi355x::PairMTP::style, ext, kk); }
MTP_CREATOR(make_mtp, MTP, false, false)
MTP_CREATOR(make_mtp_kk, MTP_KK, false, true)
MTP_CREATOR(make_mtp_small, MTP_SMALL_KK, false, true)
MTP_CREATOR(make_ext, MTP, true, false)
MTP_CREATOR(make_ext_kk, MTP_KK, true, true)
MTP_CREATOR(make_ext_small, MTP_SMALL_KK, true, true)

extern "C" void lammpsplugin_init(void *lmp, void *handle, void *regfunc)
{
  lammpsplugin_t plugin;
  auto register_plugin = (lammpsplugin_regfunc) regfunc;
  plugin.version = LAMMPS_VERSION;
  plugin.style = "pair";
  plugin.info = "Moment Tensor Potential pair styles on MI355X (libmtp_mi355x)";
  plugin.author = "mtp-mi355x";
  plugin.handle = handle;
  struct {
    const char *name;
    Pair *(*fn)(LAMMPS *);
  } styles[] = {{"mtp", make_mtp},         {"mtp/kk", make_mtp_kk},         {"mtp/small/kk", make_mtp_small},
                {"mtp/extrapolation", make_ext}, {"mtp/extrapolation/kk", make_ext_kk},
                {"mtp/extrapolation/small/kk", make_ext_small}};
  for (auto &s : styles) {
    plugin.name = s.name;
    plugin.creator.v1 = (lammpsplugin_factory1 *) s.fn;
    (*register_plugin)(&plugin, lmp);
  }
}

#endif   // LAMMPS_SOURCE_DIR_AVAILABLE
