// Drives lammps_plugin/pair_mtp_mi355x_plugin.cpp -- compiled against the MOCK of the LAMMPS API in
// tests/cpp/lammps_mock/ (test scaffolding, not LAMMPS) -- through the call sequence LAMMPS makes on a pair style
// loaded as a plugin: lammpsplugin_init (six registrations) -> creator -> settings -> coeff -> init_style -> init_one
// -> compute (-> extract / extract_peratom / pvector for the extrapolation styles).  What this pins: the adapter
// compiles, registers the reference's six style names, moves LAMMPS' pointers into the host mirror's views and the
// results back with LAMMPS' accumulate / assign semantics.  What it cannot pin: binary compatibility with a real LAMMPS.
//
//   test_plugin_mock list
//   test_plugin_mock run <style> <system> <out> <pair_style args...>
//   test_plugin_mock_kk runkk <style> <system> <out> <left|right> <pair_style args...>     (built with -DLMP_KOKKOS against
//       lammps_mock/kokkos_mock.h: the adapter's KOKKOS branch on device-resident x / f / type and a padded 2-D list in
//       either Kokkos layout; the host copies of x hold NaN, so a host path would be noticed)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "lammps_mock/lammps_mock.h"
#ifdef LMP_KOKKOS
#include <cmath>
#include <limits>

#include "lammps_mock/kokkos_mock.h"
#define HIP_OK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
template <class T> static T *to_device(const std::vector<T> &v)
{
  T *d = nullptr;
  HIP_OK(hipMalloc((void **) &d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
#endif

using namespace LAMMPS_NS;

static std::map<std::string, lammpsplugin_factory1 *> g_styles;
static std::string g_info;
static void regfunc(void *pv, void *)
{
  auto *p = (lammpsplugin_t *) pv;
  if (std::strcmp(p->style, "pair") == 0) g_styles[p->name] = p->creator.v1;
  g_info = std::string(p->version) + "|" + p->info + "|" + p->author;
}

struct System {
  int nlocal = 0, nall = 0;
  std::vector<double> x, f;
  std::vector<double *> xr, fr;
  std::vector<int> type, ilist, numneigh;
  std::vector<std::vector<int>> rows;
  std::vector<int *> firstneigh;
  double box[3] = {0, 0, 0};
  void read(const char *path)
  {
    std::ifstream in(path);
    if (!in) throw std::runtime_error(std::string("cannot open ") + path);
    in >> nlocal >> nall >> box[0] >> box[1] >> box[2];
    x.resize(3 * (size_t) nall);
    type.resize(nall);
    for (int i = 0; i < nall; i++) in >> x[3 * i] >> x[3 * i + 1] >> x[3 * i + 2] >> type[i];
    numneigh.assign(nall, 0);
    rows.assign(nall, {});
    for (int i = 0; i < nlocal; i++) {
      int n;
      in >> n;
      numneigh[i] = n;
      rows[i].resize(n);
      for (int k = 0; k < n; k++) in >> rows[i][k];
      ilist.push_back(i);
    }
    firstneigh.resize(nall);
    for (int i = 0; i < nall; i++) firstneigh[i] = rows[i].data();
    f.assign(3 * (size_t) nall, 0.0);
    xr.resize(nall);
    fr.resize(nall);
    for (int i = 0; i < nall; i++) {
      xr[i] = &x[3 * (size_t) i];
      fr[i] = &f[3 * (size_t) i];
    }
  }
};

int main(int argc, char **argv)
{
  try {
    LAMMPS lmp;
    lammpsplugin_init(&lmp, nullptr, (void *) regfunc);
    if (argc >= 2 && !std::strcmp(argv[1], "list")) {
      for (auto &kv : g_styles) std::printf("%s\n", kv.first.c_str());
      std::printf("%s\n", g_info.c_str());
      return g_styles.size() == 6 ? 0 : 1;
    }
#ifdef LMP_KOKKOS
    const bool kk = argc >= 7 && !std::strcmp(argv[1], "runkk");
    if (!kk) {
      std::fprintf(stderr, "usage: see the header of this file\n");
      return 2;
    }
    const bool left = !std::strcmp(argv[5], "left");
    const int a0 = 6;   // first pair_style argument
    delete lmp.atom;
    auto *akk = new AtomKokkos;
    lmp.atom = akk;
    lmp.kokkos = (void *) &lmp;
    HIP_OK(hipStreamCreateWithFlags(&Kokkos::mock_stream(), hipStreamNonBlocking));
#else
    if (argc < 6 || std::strcmp(argv[1], "run")) {
      std::fprintf(stderr, "usage: see the header of this file\n");
      return 2;
    }
    const int a0 = 5;
#endif
    const std::string style = argv[2];
    if (!g_styles.count(style)) throw std::runtime_error("style not registered: " + style);
    System s;
    s.read(argv[3]);
    lmp.atom->x = s.xr.data();
    lmp.atom->f = s.fr.data();
    lmp.atom->type = s.type.data();
    lmp.atom->nlocal = s.nlocal;
    lmp.atom->nghost = s.nall - s.nlocal;
    lmp.atom->natoms = s.nlocal;
    lmp.domain->xprd = s.box[0];
    lmp.domain->yprd = s.box[1];
    lmp.domain->zprd = s.box[2];
#ifdef LMP_KOKKOS
    // device-resident atoms; the host copy of x is poisoned (the KOKKOS branch must not read it)
    double *d_x = to_device(s.x), *d_f = to_device(s.f);
    int *d_type = to_device(s.type);
    for (double &v : s.x) v = std::numeric_limits<double>::quiet_NaN();
    akk->k_x.d_view.ptr = d_x;
    akk->k_f.d_view.ptr = d_f;
    akk->k_type.d_view.ptr = d_type;
    // the list as NeighListKokkos holds it: d_neighbors(i, jj) padded to maxneighs, LAMMPS' special bits in the ids
    NeighListKokkos<LMPDeviceType> list;
    int maxneighs = 0;
    for (int i = 0; i < s.nlocal; i++) maxneighs = std::max(maxneighs, s.numneigh[i]);
    maxneighs += 3;
    const size_t nrow = (size_t) s.nall;
    std::vector<int> nb(nrow * (size_t) maxneighs, 0x7fffffff);
    const size_t st_i = left ? 1 : (size_t) maxneighs, st_jj = left ? nrow : 1;
    for (int i = 0; i < s.nlocal; i++)
      for (int k = 0; k < s.numneigh[i]; k++)
        nb[(size_t) i * st_i + (size_t) k * st_jj] = (int) ((unsigned) s.rows[i][k] | ((unsigned) ((i + k) & 3) << 30));
    list.inum = s.nlocal;
    list.d_ilist.ptr = to_device(s.ilist);
    list.d_numneigh.ptr = to_device(s.numneigh);
    list.d_neighbors.ptr = to_device(nb);
    list.d_neighbors.ext[0] = nrow;
    list.d_neighbors.ext[1] = (size_t) maxneighs;
    list.d_neighbors.str[0] = st_i;
    list.d_neighbors.str[1] = st_jj;
    auto fetch_forces = [&]() {
      HIP_OK(hipStreamSynchronize(Kokkos::mock_stream()));
      HIP_OK(hipMemcpy(s.f.data(), d_f, s.f.size() * sizeof(double), hipMemcpyDeviceToHost));
    };
    auto clear_forces = [&]() { HIP_OK(hipMemsetAsync(d_f, 0, s.f.size() * sizeof(double), Kokkos::mock_stream())); };   // LAMMPS' force_clear
#else
    NeighList list;
    list.inum = s.nlocal;
    list.ilist = s.ilist.data();
    list.numneigh = s.numneigh.data();
    list.firstneigh = s.firstneigh.data();
    auto fetch_forces = [&]() {};
    auto clear_forces = [&]() { std::fill(s.f.begin(), s.f.end(), 0.0); };
#endif
    Pair *p = (Pair *) g_styles[style]((void *) &lmp);
    p->settings(argc - a0, argv + a0);
    char star[] = "*";
    char *cf[2] = {star, star};
    p->coeff(2, cf);
    p->init_style();
    if (lmp.neighbor->last_request_flags != NeighConst::REQ_FULL) throw std::runtime_error("no REQ_FULL list request");
    const double cut = p->init_one(1, 1);
    p->list = &list;
    lmp.neighbor->ago = 0;
    std::ofstream out(argv[4]);
    out.precision(17);
    const bool ext = style.find("extrapolation") != std::string::npos;
    const int eflag = 1 | 2, vflag = 2 | 4;   // ENERGY_GLOBAL | ENERGY_ATOM, VIRIAL_FDOTR | VIRIAL_ATOM
    p->compute(eflag, vflag);
    const double e_first = p->eng_vdwl;
    if (ext) {
      int dim = -1, ncol = -1;
      int *flag = (int *) p->extract("extrapolation_flag", dim);
      if (!flag || dim != 0) throw std::runtime_error("extract(extrapolation_flag)");
      *flag = 1;   // what `fix pair` does on its steps
      clear_forces();
      lmp.neighbor->ago = 1;   // same list: must not be handed over again
      p->compute(eflag, vflag);
      out << p->eng_vdwl << " " << e_first << " " << p->pvector[0] << " " << p->nextra << "\n";
      const double *g = (const double *) p->extract_peratom("extrapolation", ncol);
      if (!g || ncol != 0) throw std::runtime_error("extract_peratom(extrapolation)");
      for (int i = 0; i < s.nlocal; i++) out << g[i] << "\n";
    } else {
      fetch_forces();
      out << p->eng_vdwl << " " << cut << " " << p->no_virial_fdotr_compute << p->manybody_flag << p->one_coeff
          << p->single_enable << p->restartinfo << "\n";
      for (int q = 0; q < 6; q++) out << p->virial[q] << (q == 5 ? "\n" : " ");
      for (int i = 0; i < s.nall; i++)
        out << s.f[3 * i] << " " << s.f[3 * i + 1] << " " << s.f[3 * i + 2] << " " << p->eatom[i] << " " << p->vatom[i][0] << "\n";
    }
#ifdef LMP_KOKKOS
    // the data-manager protocol: x, f, type synced to the device BEFORE the force call, f marked modified there AFTER
    out << "kokkos " << akk->synced_device << " " << akk->modified_device << " " << akk->sync_call << " " << akk->modified_call << "\n";
#endif
    delete p;
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
