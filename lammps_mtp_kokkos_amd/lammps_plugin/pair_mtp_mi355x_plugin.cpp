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
#include "error.h"
#include "force.h"
#include "lammpsplugin.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "pair.h"
#include "utils.h"
#include "version.h"

#include <cstring>
#include <mpi.h>
#include <vector>

#include "mtp_mi355x.h"

namespace LAMMPS_NS {

class PairMTPMI355X : public Pair {
 public:
  PairMTPMI355X(LAMMPS *lmp, int variant, bool ext) : Pair(lmp), variant_(variant), ext_(ext)
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
    if (ctx_) mtp_context_destroy(ctx_);
    if (pot_) mtp_potential_free(pot_);
  }

  void settings(int narg, char **arg) override
  {
    // same grammar as the reference (pair_mtp.cpp:285-297, pair_mtp_kokkos.cpp:113-117,
    // pair_mtp_extrapolation.cpp:488-502); chunksize is accepted and ignored
    int n = narg;
    if (n >= 3 && utils::lowercase(arg[n - 2]) == "chunksize") n -= 2;
    if (!ext_ && n < 1) error->all(FLERR, "Pair mtp only accepts 1 argument, the MTP potential file");
    if (ext_ && n != 1 && n != 4)
      error->all(FLERR, "Pair mtp/extrapolation only accepts 1 argument: {{potential_file}}. Or 4 arguments.");
    if (ext_ && n == 4) {
      mlip3_style_ = true;
      select_ = utils::numeric(FLERR, arg[2], true, lmp);
      break_ = utils::numeric(FLERR, arg[3], true, lmp);
    }
    char err[512] = "";
    if (mtp_potential_load(arg[0], ext_ ? 1 : 0, &pot_, err, sizeof(err)) != MTP_OK) error->all(FLERR, err);
    mtp_potential_get_info(pot_, &info_);
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
    if (neighbor->ago == 0 || !list_sent_) {   // list was rebuilt this step
      check(mtp_set_neighbors(ctx_, list->inum, list->ilist, list->numneigh, list->firstneigh,
                              atom->nlocal + atom->nghost));
      list_sent_ = true;
    }
    const int grade = ext_ && (extrapolation_flag_ || mlip3_style_);
    const int nall = atom->nlocal + atom->nghost;
    if (grade && !info_.configuration_mode && (int) grades_.size() < nall) grades_.resize(nall, 0.0);
    if (grade && info_.configuration_mode) cders_.assign(info_.coeff_count, 0.0);
    double mg = 0.0;
    check(mtp_compute(ctx_, &atom->x[0][0], atom->type, eflag, vflag, grade, &atom->f[0][0],
                      eflag_atom ? eatom : nullptr, vflag_atom ? &vatom[0][0] : nullptr, &eng_vdwl, virial,
                      grade && !info_.configuration_mode ? grades_.data() : nullptr, &mg,
                      grade && info_.configuration_mode ? cders_.data() : nullptr));
    if (grade) {   // compile_grades, pair_mtp_extrapolation.cpp:363-382
      if (info_.configuration_mode) {
        MPI_Allreduce(MPI_IN_PLACE, cders_.data(), info_.coeff_count, MPI_DOUBLE, MPI_SUM, world);
        mtp_cfg_grade(pot_, cders_.data(), &mg);
        mg = atom->natoms > 0 ? mg / atom->natoms : 0.0;
      } else {
        MPI_Allreduce(MPI_IN_PLACE, &mg, 1, MPI_DOUBLE, MPI_MAX, world);
      }
      if (comm->me == 0) pvector[0] = mg;
      if (mlip3_style_ && mg >= break_ && comm->me == 0)
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
  void check(int rc)
  {
    if (rc != MTP_OK) error->one(FLERR, "libmtp_mi355x: {}", mtp_last_error(ctx_));
  }
  int variant_;
  bool ext_, mlip3_style_ = false, list_sent_ = false;
  int extrapolation_flag_ = 0;
  double select_ = 0, break_ = 0;
  mtp_potential *pot_ = nullptr;
  mtp_context *ctx_ = nullptr;
  mtp_potential_info info_{};
  std::vector<double> grades_, cders_;
};

}   // namespace LAMMPS_NS

using namespace LAMMPS_NS;
#define MTP_CREATOR(fn, variant, ext) \
  static Pair *fn(LAMMPS *lmp) { return new PairMTPMI355X(lmp, variant, ext); }
MTP_CREATOR(make_mtp, MTP_VARIANT_AUTO, false)
MTP_CREATOR(make_mtp_kk, MTP_VARIANT_LARGE, false)
MTP_CREATOR(make_mtp_small, MTP_VARIANT_SMALL, false)
MTP_CREATOR(make_ext, MTP_VARIANT_AUTO, true)
MTP_CREATOR(make_ext_kk, MTP_VARIANT_LARGE, true)
MTP_CREATOR(make_ext_small, MTP_VARIANT_SMALL, true)

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
