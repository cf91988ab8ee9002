// LAMMPS plugin adapter: registers pair styles `mtp`, `mtp/kk`, `mtp/small/kk`,
// `mtp/extrapolation`, `mtp/extrapolation/kk`, `mtp/extrapolation/small/kk` at run time
// (`plugin load libmtp_mi355x_lammps.so`) and forwards the Pair virtuals the reference overrides
// (/root/reference/LAMMPS/ML-MTP/pair_mtp.h:34-40, pair_mtp_extrapolation.h:35-38) to the C ABI
// of libmtp_mi355x (include/mtp_mi355x.h).
//
// COMPILE-GATED: needs a LAMMPS source tree (not present in this repository's build image):
//   hipcc/g++ -std=c++17 -fPIC -shared -I$LAMMPS_SOURCE_DIR/src -I$REPO/include \
//       pair_mtp_mi355x_plugin.cpp -L$REPO/lammps_mtp_kokkos_amd -lmtp_mi355x -o libmtp_mi355x_lammps.so
// It has not been compiled against LAMMPS here; INTEGRATION.md lists what to check first.
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
#endif

#include <cstring>
#include <mpi.h>
#include <string>
#include <vector>

#include "../host/mtp_cfg_writer.hpp"   // the .cfg record and the logmesg lines, shared with the host mirror
#include "mtp_mi355x.h"

namespace LAMMPS_NS {

class PairMTPMI355X : public Pair {
 public:
  // kk: one of the /kk styles (argument grammar of KOKKOS/pair_mtp_kokkos.cpp:113-117 and, with LMP_KOKKOS, device-resident x / f)
  PairMTPMI355X(LAMMPS *lmp, int variant, bool ext, bool kk) : Pair(lmp), variant_(variant), ext_(ext), kk_(kk)
  {
    single_enable = 0;   // pair_mtp.cpp:37-40
    restartinfo = 0;
    one_coeff = 1;
    manybody_flag = 1;
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
    if (preselected_file_) fclose(preselected_file_);
    if (ctx_) mtp_context_destroy(ctx_);
    if (pot_) mtp_potential_free(pot_);
  }

  void settings(int narg, char **arg) override
  {
    // the reference's grammars, argument count for argument count:
    //   mtp                        <file> [ignored ...]                      pair_mtp.cpp:285-297
    //   mtp/kk, mtp/small/kk       <file> chunksize <N>   (exactly 3)         KOKKOS/pair_mtp_kokkos.cpp:113-117
    //   mtp/extrapolation[...]     <file> [<out> <sel> <brk>] [chunksize <N>] pair_mtp_extrapolation.cpp:488-502
    // chunksize only bounded the reference's spilled Jacobian: parsed, checked, ignored
    int n = narg;
    if (!ext_) {
      if (!kk_) {
        if (n < 1) error->all(FLERR, "Pair mtp only accepts 1 argument, the MTP potential file");
        if (n > 1 && comm->me == 0)
          utils::logmesg(lmp, "Pair mtp only accepts 1 argument, the MTP potential file. Ignoring excessive arguments!\n");
      } else {
        if (n != 3 || utils::lowercase(arg[1]) != "chunksize")
          error->all(FLERR, "Pair mtp/kk requires 3 arguments {{potential_file} \"chunksize\" {{chunksize}}.");
        (void) utils::inumeric(FLERR, arg[2], true, lmp);
      }
    } else {
      if ((n == 3 && utils::lowercase(arg[1]) == "chunksize") || (n == 6 && utils::lowercase(arg[4]) == "chunksize")) {
        if (comm->me == 0) utils::logmesg(lmp, "Ignoring chunksize settings!\n");
        n -= 2;
      } else if (n != 1 && n != 4) {
        error->all(FLERR, "Pair mtp/extrapolation only accepts 1 argument: {{potential_file}}. Or 4 arguments: "
                          "{{potential_file}} {{output_file}}. {{selection_threshold}} {{break_threshold}}.");
      }
      if (n == 4) {
        mlip3_style_ = true;
        select_ = utils::numeric(FLERR, arg[2], true, lmp);
        break_ = utils::numeric(FLERR, arg[3], true, lmp);
      }
    }
    char err[512] = "";
    if (mtp_potential_load(arg[0], ext_ ? 1 : 0, &pot_, err, sizeof(err)) != MTP_OK) error->all(FLERR, err);
    mtp_potential_get_info(pot_, &info_);
    if (comm->me == 0) {   // the reference parses, hence logs, on rank 0 (pair_mtp.cpp:343, 383, 389)
      utils::logmesg(lmp, mtp_mi355x::log_scaling(info_.scaling));
      utils::logmesg(lmp, mtp_mi355x::log_species(info_.species_count));
      if (ext_)            // pair_mtp_extrapolation.cpp:508-517
        utils::logmesg(lmp, mtp_mi355x::log_extrapolation_mode(mlip3_style_, info_.configuration_mode != 0, select_, break_));
    }
    if (mlip3_style_ && comm->me == 0) {   // pair_mtp_extrapolation.cpp:519-521
      preselected_file_ = fopen(arg[1], "w");
      if (!preselected_file_) error->one(FLERR, "Cannot open {}", arg[1]);
    }
    const int np1 = info_.species_count + 1;   // pair_mtp.cpp:391-393, 455-456
    memory->create(setflag, np1, np1, "pair:setflag");
    memory->create(cutsq, np1, np1, "pair:cutsq");
    for (int i = 1; i < np1; i++)
      for (int j = 1; j < np1; j++) {
        setflag[i][j] = 1;
        cutsq[i][j] = info_.max_cutoff * info_.max_cutoff;
      }
    allocated = 1;
  }
  void coeff(int narg, char **) override
  {
    if (narg != 2) error->all(FLERR, "Only \"pair_coeff * *\" is permitted");
  }
  void init_style() override
  {
    if (force->newton_pair != 1) error->all(FLERR, "Pair style MTP requires Newton Pair on");
    neighbor->add_request(this, NeighConst::REQ_FULL);
    if (!ctx_) {
      char err[512] = "";
      int ndev = 1;   // one rank per GPU: local rank -> device
      MPI_Comm node;
      MPI_Comm_split_type(world, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node);
      int local = 0;
      MPI_Comm_rank(node, &local);
      MPI_Comm_free(&node);
      if (const char *e = getenv("MTP_MI355X_GPUS_PER_NODE")) ndev = atoi(e);
      if (mtp_context_create(pot_, local % (ndev > 0 ? ndev : 1), &ctx_, err, sizeof(err)) != MTP_OK)
        error->all(FLERR, err);
      mtp_context_set_variant(ctx_, variant_);
    }
  }
  double init_one(int i, int j) override
  {
    if (setflag[i][j] == 0) error->all(FLERR, "Not all pair coeffs are set. See types {}-{}.", i, j);
    return info_.max_cutoff;
  }
  void compute(int eflag, int vflag) override
  {
    ev_init(eflag, vflag);
    const int nall = atom->nlocal + atom->nghost;
    if (neighbor->ago == 0 || !list_sent_) {   // list was rebuilt this step
      check(mtp_set_neighbors(ctx_, list->inum, list->ilist, list->numneigh, list->firstneigh, nall));
      list_sent_ = true;
    }
    const int grade = ext_ && (extrapolation_flag_ || mlip3_style_);   // pair_mtp_extrapolation.cpp:71
    const bool cfg = info_.configuration_mode != 0;
    if (grade && !cfg && (int) grades_.size() < nall) grades_.resize(nall, 0.0);   // :91-94
    if (grade && cfg) cders_.assign(info_.coeff_count, 0.0);                        // :97-98
    double mg = 0.0;
#ifdef LMP_KOKKOS
    // KOKKOS-resident positions and forces (KOKKOS/pair_mtp_kokkos.cpp:231-240: atomKK->sync / modified): no PCIe
    // copy of x and f.  Per-atom outputs and grades still go through the host arrays below, so this path is taken
    // when the step asks for global tallies only -- every step of a production run.
    if (kk_ && lmp->kokkos && !eflag_atom && !vflag_atom && !grade) {
      auto *akk = (AtomKokkos *) atom;
      akk->sync(Device, X_MASK | F_MASK | TYPE_MASK);
      if (!d_ev_) check(hipMalloc((void **) &d_ev_, 8 * sizeof(double)) == hipSuccess ? MTP_OK : MTP_ERR_DEVICE);
      hipMemsetAsync(d_ev_, 0, 8 * sizeof(double), nullptr);
      check(mtp_compute_device(ctx_, nullptr, akk->k_x.d_view.data(), akk->k_type.d_view.data(), eflag, vflag, 0,
                               akk->k_f.d_view.data(), nullptr, nullptr, d_ev_, nullptr, nullptr, nullptr));
      check(mtp_synchronize(ctx_, nullptr));
      double ev[8];
      hipMemcpy(ev, d_ev_, sizeof(ev), hipMemcpyDeviceToHost);
      if (eflag_global) eng_vdwl += ev[0];
      if (vflag_either)
        for (int q = 0; q < 6; q++) virial[q] += ev[1 + q];
      akk->modified(Device, F_MASK);
      return;
    }
#endif
    check(mtp_compute(ctx_, &atom->x[0][0], atom->type, eflag, vflag, grade, &atom->f[0][0],
                      eflag_atom ? eatom : nullptr, vflag_atom ? &vatom[0][0] : nullptr, &eng_vdwl, virial,
                      grade && !cfg ? grades_.data() : nullptr, &mg, grade && cfg ? cders_.data() : nullptr));
    if (!grade) return;
    // compile_grades, pair_mtp_extrapolation.cpp:363-382
    if (cfg) {
      MPI_Allreduce(MPI_IN_PLACE, cders_.data(), info_.coeff_count, MPI_DOUBLE, MPI_SUM, world);
      mtp_cfg_grade(pot_, cders_.data(), &mg);
      mg = atom->natoms > 0 ? mg / atom->natoms : 0.0;
    } else {
      MPI_Allreduce(MPI_IN_PLACE, &mg, 1, MPI_DOUBLE, MPI_MAX, world);
    }
    if (comm->me == 0) pvector[0] = mg;
    if (!mlip3_style_) return;
    // evaluate_grades, :387-397
    if (mg >= select_) write_config(mg);
    if (mg >= break_ && comm->me == 0) {
      if (preselected_file_) {
        fflush(preselected_file_);
        fclose(preselected_file_);
        preselected_file_ = nullptr;
      }
      error->one(FLERR, "Exceeded Break Threshold: {:.5f}. Terminating simulation.\n", mg);
    }
  }
  void *extract(const char *str, int &dim) override
  {
    dim = 0;
    if (ext_ && strcmp(str, "extrapolation_flag") == 0) return (void *) &extrapolation_flag_;
    return nullptr;
  }
  void *extract_peratom(const char *str, int &ncol) override
  {
    if (ext_ && strcmp(str, "extrapolation") == 0) {
      if (info_.configuration_mode)
        error->one(FLERR, "Please use the MLIP-3 style extrapolation for configuration mode MTPs!");
      ncol = 0;
      return (void *) grades_.data();
    }
    return nullptr;
  }

 private:
  // write_config, pair_mtp_extrapolation.cpp:401-479: the shared writer with MPI behind its three exchanges
  struct MpiCtx {
    MPI_Comm world;
  };
  static int scan_sum(int v, void *c)
  {
    int out = 0;
    MPI_Scan(&v, &out, 1, MPI_INT, MPI_SUM, ((MpiCtx *) c)->world);
    return out;
  }
  static void send_to_root(const char *buf, size_t n, void *c)
  {
    MPI_Send(buf, (int) n, MPI_CHAR, 0, 0, ((MpiCtx *) c)->world);
  }
  static void recv_on_root(int src, std::string &out, void *c)
  {
    MPI_Status st;
    int n = 0;
    MPI_Probe(src, 0, ((MpiCtx *) c)->world, &st);
    MPI_Get_count(&st, MPI_CHAR, &n);
    out.resize((size_t) n);
    MPI_Recv(out.empty() ? nullptr : &out[0], n, MPI_CHAR, src, 0, ((MpiCtx *) c)->world, &st);
  }
  void write_config(double mg)
  {
    MpiCtx mc{world};
    mtp_mi355x::CfgComm cc;
    cc.me = comm->me;
    cc.nprocs = comm->nprocs;
    cc.ctx = &mc;
    cc.scan_sum = scan_sum;
    cc.send_to_root = send_to_root;
    cc.recv_on_root = recv_on_root;
    mtp_mi355x::CfgBox b;
    b.xprd = domain->xprd;
    b.yprd = domain->yprd;
    b.zprd = domain->zprd;
    b.xy = domain->xy;
    b.xz = domain->xz;
    b.yz = domain->yz;
    mtp_mi355x::cfg_write_record(preselected_file_, cc, (long) atom->natoms, b, info_.configuration_mode != 0, list->inum,
                                 atom->type, &atom->x[0][0], info_.configuration_mode ? nullptr : grades_.data(), mg);
  }
  void check(int rc)
  {
    if (rc != MTP_OK) error->one(FLERR, "libmtp_mi355x: {}", ctx_ ? mtp_last_error(ctx_) : "device error");
  }
  int variant_;
  bool ext_, kk_, mlip3_style_ = false, list_sent_ = false;
  int extrapolation_flag_ = 0;
  double select_ = 0, break_ = 0;
  FILE *preselected_file_ = nullptr;
  double *d_ev_ = nullptr;
  mtp_potential *pot_ = nullptr;
  mtp_context *ctx_ = nullptr;
  mtp_potential_info info_{};
  std::vector<double> grades_, cders_;
};

}   // namespace LAMMPS_NS

using namespace LAMMPS_NS;
#define MTP_CREATOR(fn, variant, ext, kk) \
  static Pair *fn(LAMMPS *lmp) { return new PairMTPMI355X(lmp, variant, ext, kk); }
MTP_CREATOR(make_mtp, MTP_VARIANT_AUTO, false, false)
MTP_CREATOR(make_mtp_kk, MTP_VARIANT_LARGE, false, true)
MTP_CREATOR(make_mtp_small, MTP_VARIANT_SMALL, false, true)
MTP_CREATOR(make_ext, MTP_VARIANT_AUTO, true, false)
MTP_CREATOR(make_ext_kk, MTP_VARIANT_LARGE, true, true)
MTP_CREATOR(make_ext_small, MTP_VARIANT_SMALL, true, true)

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
