// See pair_mtp_mi355x.hpp.  Reference line numbers are relative to /root/reference/LAMMPS/ML-MTP/.
#include "pair_mtp_mi355x.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>

namespace mtp_mi355x {

namespace {
std::string lower(const char *s)
{
  std::string r(s ? s : "");
  std::transform(r.begin(), r.end(), r.begin(), [](unsigned char c) { return (char) std::tolower(c); });
  return r;
}
double numeric(const char *s)   // utils::numeric: the whole token must be a number
{
  char *e = nullptr;
  double v = std::strtod(s, &e);
  if (!s || !*s || *e) throw Error(MTP_ERR_ARG, std::string("Expected floating point parameter instead of '") + (s ? s : "") + "' in input script or data file");
  return v;
}
}   // namespace

PairMTP::PairMTP(Style style, int device) : style_(style), device_(device) {}

PairMTP::~PairMTP()
{
  if (ctx_) mtp_context_destroy(ctx_);
  if (pot_) mtp_potential_free(pot_);
}

void PairMTP::require(int rc, const char *what)
{
  if (rc == MTP_OK) return;
  std::string msg = ctx_ ? mtp_last_error(ctx_) : "";
  throw Error(rc, msg.empty() ? std::string(what) : msg);
}

void PairMTP::load(const char *file, bool selection)
{
  char err[512] = "";
  if (pot_) {
    if (ctx_) mtp_context_destroy(ctx_);
    mtp_potential_free(pot_);
    pot_ = nullptr;
    ctx_ = nullptr;
  }
  int rc = mtp_potential_load(file, selection ? 1 : 0, &pot_, err, (int) sizeof(err));
  if (rc != MTP_OK) throw Error(rc, err);
  mtp_potential_get_info(pot_, &info);
  logmesg(log_scaling(info.scaling));          // pair_mtp.cpp:383
  logmesg(log_species(info.species_count));    // pair_mtp.cpp:389
}

void PairMTP::logmesg(const std::string &msg) const
{
  if (me_ != 0) return;   // the reference parses, hence logs, on rank 0 only (pair_mtp.cpp:343)
  if (log_.write) log_.write(msg.c_str(), log_.ctx);
  else std::fputs(msg.c_str(), stdout);
}

void PairMTP::settings(int narg, char **arg)
{
  if (style_ == MTP) {   // pair_mtp.cpp:285-297: one argument, extra ones are ignored with a notice
    if (narg < 1) throw Error(MTP_ERR_ARG, "Pair mtp only accepts 1 argument, the MTP potential file");
    if (narg > 1)
      std::fprintf(stderr, "Pair mtp only accepts 1 argument, the MTP potential file. Ignoring excessive arguments!\n");
  } else {               // KOKKOS/pair_mtp_kokkos.cpp:113-117: exactly <file> chunksize <N>
    if (narg != 3 || lower(arg[1]) != "chunksize")
      throw Error(MTP_ERR_ARG, "Pair mtp/kk requires 3 arguments {{potential_file} \"chunksize\" {chunksize}}.");
    char *e = nullptr;
    (void) std::strtol(arg[2], &e, 10);
    if (*e) throw Error(MTP_ERR_ARG, std::string("Expected integer parameter instead of '") + arg[2] + "' in input script or data file");
  }
  load(arg[0], false);
}

void PairMTP::coeff(int narg, char **)
{
  if (narg != 2) throw Error(MTP_ERR_ARG, "Only \"pair_coeff * *\" is permitted");
}

void PairMTP::init_style(int newton_pair)
{
  if (newton_pair != 1) throw Error(MTP_ERR_STATE, "Pair style MTP requires Newton Pair on");
  if (!pot_) throw Error(MTP_ERR_STATE, "pair_style settings were not given");
  if (!ctx_) {   // device tables (the reference copies them in PairMTPKokkos::settings)
    char err[512] = "";
    int rc = mtp_context_create(pot_, device_, &ctx_, err, (int) sizeof(err));
    if (rc != MTP_OK) throw Error(rc, err);
    mtp_context_set_variant(ctx_, style_ == MTP_SMALL_KK ? MTP_VARIANT_SMALL : (style_ == MTP_KK ? MTP_VARIANT_LARGE : MTP_VARIANT_AUTO));
  }
}

double PairMTP::init_one(int i, int j)
{
  // every species pair of the file is set in read_file (pair_mtp.cpp:455); a type beyond the file is not
  if (!pot_ || i < 1 || j < 1 || i > info.species_count || j > info.species_count)
    throw Error(MTP_ERR_STATE, "Not all pair coeffs are set. See types " + std::to_string(i) + "-" + std::to_string(j) + ".");
  return info.max_cutoff;
}

void PairMTP::set_neighbor_list(const NeighListView &l)
{
  if (!ctx_) throw Error(MTP_ERR_STATE, "init_style() must run before the neighbour list is handed over");
  require(mtp_set_neighbors(ctx_, l.inum, l.ilist, l.numneigh, l.firstneigh, atom.nall), "mtp_set_neighbors");
  list_set_ = true;
}

void PairMTP::bind_device(const DeviceAtomView &a)
{
  datom = a;
  resident_ = true;
  atom.nlocal = a.nlocal;   // sizes of eatom / vatom / grades and the .cfg writer's atom count
  atom.nall = a.nall;
  atom.natoms = a.natoms;
}

void PairMTP::set_neighbor_list_device(const DeviceNeighListView &l)
{
  if (!ctx_) throw Error(MTP_ERR_STATE, "init_style() must run before the neighbour list is handed over");
  if (!resident_) throw Error(MTP_ERR_STATE, "bind_device() must precede set_neighbor_list_device()");
  require(mtp_set_neighbors_device_2d(ctx_, datom.stream, l.inum, l.d_ilist, l.d_numneigh, l.d_neighbors, l.stride_i,
                                      l.stride_jj, l.maxneighs, datom.nall),
          "mtp_set_neighbors_device_2d");
  list_set_ = true;
}

// One device-resident step (KOKKOS/pair_mtp_kokkos.cpp:197-399): everything on datom.stream, one wait for the
// totals.  The global virial is tallied on the RAW vflag, as PairMTP::compute does (pair_mtp.cpp:257 `if (vflag)`;
// the reference's /kk styles set no_virial_fdotr_compute and read ev.v the same way, pair_mtp_kokkos.cpp:210, 367-375).
void PairMTP::compute_resident(int eflag, int vflag, int grade, double *max_grade, double *coeff_ders)
{
  require(mtp_compute_resident(ctx_, datom.stream, datom.d_x, datom.d_type, datom.d_f, eflag, vflag, grade), "mtp_compute_resident");
  double ev[7];
  require(mtp_resident_totals(ctx_, datom.stream, ev, max_grade, coeff_ders), "mtp_resident_totals");
  if (eflag_global) eng_vdwl += ev[0];
  if (vflag)
    for (int q = 0; q < 6; q++) virial[q] += ev[1 + q];
  // per-atom tallies reach the host only when the step asked for them (k_eatom / k_vatom sync, :379-388)
  if (eflag_atom) require(mtp_resident_peratom_host(ctx_, datom.stream, MTP_PERATOM_EATOM, eatom.data()), "eatom copy");
  if (vflag_atom) require(mtp_resident_peratom_host(ctx_, datom.stream, MTP_PERATOM_VATOM, vatom.data()), "vatom copy");
}

void PairMTP::ev_setup(int eflag, int vflag)
{
  // LAMMPS Pair::ev_setup bit semantics (pair.h); accumulators are zeroed each call
  eflag_either = eflag;
  eflag_global = eflag & MTP_ENERGY_GLOBAL;
  eflag_atom = eflag & MTP_ENERGY_ATOM;
  vflag_either = vflag;
  vflag_global = vflag & 3;
  vflag_atom = vflag & MTP_VIRIAL_ATOM;
  eng_vdwl = 0.0;
  std::fill(virial, virial + 6, 0.0);
  if (eflag_atom) eatom.assign((size_t) atom.nall, 0.0);
  if (vflag_atom) vatom.assign((size_t) atom.nall * 6, 0.0);
}

void PairMTP::compute(int eflag, int vflag)
{
  if (!ctx_ || !list_set_) throw Error(MTP_ERR_STATE, "compute() before init_style()/set_neighbor_list()");
  ev_setup(eflag, vflag);
  if (resident_) {
    compute_resident(eflag, vflag, 0, nullptr, nullptr);
    return;
  }
  require(mtp_compute(ctx_, atom.x, atom.type, eflag, vflag, 0, atom.f, eflag_atom ? eatom.data() : nullptr,
                      vflag_atom ? vatom.data() : nullptr, &eng_vdwl, virial, nullptr, nullptr, nullptr),
          "mtp_compute");
}

// ---------------------------------------------------------------------------------------------------

PairMTPExtrapolation::PairMTPExtrapolation(Style style, int device) : PairMTP(style, device) {}

PairMTPExtrapolation::~PairMTPExtrapolation()
{
  if (preselected_file) std::fclose(preselected_file);
}

void PairMTPExtrapolation::settings(int narg, char **arg)
{
  // pair_mtp_extrapolation.cpp:488-502
  if ((narg == 3 && lower(arg[1]) == "chunksize") || (narg == 6 && lower(arg[4]) == "chunksize")) {
    if (red.me == 0) std::fprintf(stderr, "Ignoring chunksize settings!\n");
    narg -= 2;
  } else if (narg != 1 && narg != 4) {
    throw Error(MTP_ERR_ARG,
                "Pair mtp/extrapolation only accepts 1 argument: {potential_file}. Or 4 arguments: {potential_file} "
                "{output_file}. {selection_threshold} {break_threshold}.");
  }
  if (narg == 4) {
    mlip3_style = true;
    select_threshold = numeric(arg[2]);
    break_threshold = numeric(arg[3]);
  }
  load(arg[0], true);
  configuration_mode = info.configuration_mode != 0;
  logmesg(log_extrapolation_mode(mlip3_style, configuration_mode, select_threshold, break_threshold));   // :508-517
  energy_ders_wrt_coeffs.assign((size_t) info.coeff_count, 0.0);
  if (mlip3_style && red.me == 0) {
    preselected_file = std::fopen(arg[1], "w");
    if (!preselected_file) throw Error(MTP_ERR_IO, std::string("cannot open ") + arg[1]);
  }
}

void PairMTPExtrapolation::compute(int eflag, int vflag)
{
  if (!extrapolation_flag && !mlip3_style) {   // :71-74
    PairMTP::compute(eflag, vflag);
    return;
  }
  if (!ctx_ || !list_set_) throw Error(MTP_ERR_STATE, "compute() before init_style()/set_neighbor_list()");
  max_grade = 0.0;
  ev_setup(eflag, vflag);
  if (!configuration_mode && (int) nbh_extrapolation_grades.size() < atom.nall)
    nbh_extrapolation_grades.resize((size_t) atom.nall, 0.0);   // :91-94 (grown, never shrunk)
  if (configuration_mode) std::fill(energy_ders_wrt_coeffs.begin(), energy_ders_wrt_coeffs.end(), 0.0);   // :97-98
  if (resident_) {
    // grades stay in HBM (KOKKOS/pair_mtp_extrapolation_kokkos.cpp:223-243 copies them only when asked); the maximum
    // and, in configuration mode, the C-double candidate vector come back with the totals
    compute_resident(eflag, vflag, 1, &max_grade, configuration_mode ? energy_ders_wrt_coeffs.data() : nullptr);
    grades_on_device_ = !configuration_mode;
    compile_grades();
    if (mlip3_style) evaluate_grades();
    return;
  }
  grades_on_device_ = false;
  require(mtp_compute(ctx_, atom.x, atom.type, eflag, vflag, 1, atom.f, eflag_atom ? eatom.data() : nullptr,
                      vflag_atom ? vatom.data() : nullptr, &eng_vdwl, virial,
                      configuration_mode ? nullptr : nbh_extrapolation_grades.data(), &max_grade,
                      configuration_mode ? energy_ders_wrt_coeffs.data() : nullptr),
          "mtp_compute");
  compile_grades();
  if (mlip3_style) evaluate_grades();
}

void PairMTPExtrapolation::compile_grades()
{
  if (configuration_mode) {   // :366-376
    if (red.sum) red.sum(energy_ders_wrt_coeffs.data(), (int) energy_ders_wrt_coeffs.size(), red.ctx);
    double g = 0.0;
    require(mtp_cfg_grade(pot_, energy_ders_wrt_coeffs.data(), &g), "mtp_cfg_grade");
    max_grade = atom.natoms > 0 ? g / (double) atom.natoms : 0.0;
  } else {                    // :378-380
    if (red.max) red.max(&max_grade, 1, red.ctx);
  }
  if (red.me == 0) pvector[0] = max_grade;   // :381
}

void PairMTPExtrapolation::evaluate_grades()
{
  if (max_grade >= select_threshold) write_config();   // :389
  if (max_grade >= break_threshold && red.me == 0) {   // :390-396
    if (preselected_file) {
      std::fflush(preselected_file);
      std::fclose(preselected_file);
      preselected_file = nullptr;
    }
    char msg[128];
    std::snprintf(msg, sizeof(msg), "Exceeded Break Threshold: %.5f. Terminating simulation.\n", max_grade);
    throw Error(MTP_ERR_STATE, msg);
  }
}

// MLIP-3 .cfg record of the current configuration (:401-479): every rank formats its atom lines, rank 0 writes the
// header, its own lines, then the other ranks' in rank order (mtp_cfg_writer.hpp; the exchanges go through `red`).
void PairMTPExtrapolation::write_config()
{
  int ncol = 0;
  if (!configuration_mode) (void) extract_peratom("extrapolation", ncol);   // grades to the host when they are not there yet
  std::vector<double> xh;
  const double *x_host = atom.x;
  if (resident_) {   // the record lists positions: one copy of the owned rows, only on the (rare) steps that write
    xh.resize(3 * (size_t) atom.nlocal);
    require(mtp_copy_to_host(ctx_, datom.stream, xh.data(), datom.d_x, xh.size() * sizeof(double)), "position copy");
    x_host = xh.data();
  }
  std::vector<int> th;
  const int *t_host = atom.type;
  if (resident_) {
    th.resize((size_t) atom.nlocal);
    require(mtp_copy_to_host(ctx_, datom.stream, th.data(), datom.d_type, th.size() * sizeof(int)), "type copy");
    t_host = th.data();
  }
  CfgComm cc;
  cc.me = red.me;
  cc.nprocs = red.nprocs;
  cc.ctx = red.ctx;
  cc.scan_sum = red.scan_sum;
  cc.send_to_root = red.send_to_root;
  cc.recv_on_root = red.recv_on_root;
  cfg_write_record(preselected_file, cc, atom.natoms, box, configuration_mode, atom.nlocal, t_host, x_host,
                   configuration_mode ? nullptr : nbh_extrapolation_grades.data(), max_grade);
}

void *PairMTPExtrapolation::extract(const char *str, int &dim)
{
  dim = 0;
  if (std::strcmp(str, "extrapolation_flag") == 0) return (void *) &extrapolation_flag;
  return nullptr;
}

void *PairMTPExtrapolation::extract_peratom(const char *str, int &ncol)
{
  if (std::strcmp(str, "extrapolation") == 0) {
    if (configuration_mode)
      throw Error(MTP_ERR_STATE, "Please use the MLIP-3 style extrapolation for configuration mode MTPs!");
    ncol = 0;
    if (grades_on_device_) {   // first request after a device-resident grade call: one copy, then the host array is current
      if ((int) nbh_extrapolation_grades.size() < atom.nall) nbh_extrapolation_grades.resize((size_t) atom.nall, 0.0);
      require(mtp_resident_peratom_host(ctx_, datom.stream, MTP_PERATOM_GRADES, nbh_extrapolation_grades.data()), "grades copy");
      grades_on_device_ = false;
    }
    return (void *) nbh_extrapolation_grades.data();
  }
  return nullptr;
}

}   // namespace mtp_mi355x
