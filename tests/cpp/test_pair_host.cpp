// Driver for the C++ host mirror (lammps_mtp_kokkos_amd/host): the call sequence LAMMPS makes
// on a pair style -- settings, coeff, init_style, init_one, compute -- on a system read from a
// text file written by tests/test_pair_host.py.
//
//   test_pair_host args                                   argument-grammar checks (no GPU needed)
//   test_pair_host cfg <system> <outdir> <0|1>            .cfg records of 1-, 2- and 3-rank jobs (no GPU needed)
//   test_pair_host run    <style> <system> <out> <pair_style args...>
//   test_pair_host runext <style> <system> <out> <pair_style args...>
//   test_pair_host rundev / runextdev ...                 the same through the /kk styles' data path: x, f, type and
//                                                         the padded 2-D neighbour view resident on the device
//                                                         (DeviceAtomView / DeviceNeighListView), vflag = VIRIAL_FDOTR
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>   // host API only: staging of the device views a KOKKOS-resident LAMMPS would own

#include "../../lammps_mtp_kokkos_amd/host/pair_mtp_mi355x.hpp"

using namespace mtp_mi355x;

struct System {
  int nlocal = 0, nall = 0;
  std::vector<double> x, f;
  std::vector<int> type, ilist, numneigh;
  std::vector<std::vector<int>> rows;
  std::vector<const int *> firstneigh;
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
  }
};


// The device views of a KOKKOS-resident LAMMPS, staged by hand: x / f / type, d_ilist, d_numneigh (indexed by atom) and
// the padded LayoutLeft view d_neighbors(i, jj) = base[i + jj * nall] with LAMMPS' special-bond bits in the entries and
// garbage in the padding; everything on one non-blocking stream (the "execution space instance").
struct DeviceSystem {
  double *d_x = nullptr, *d_f = nullptr;
  int *d_type = nullptr, *d_ilist = nullptr, *d_numneigh = nullptr, *d_neighbors = nullptr;
  hipStream_t stream = nullptr;
  int maxneighs = 0;
  static void ok(hipError_t e, const char *what)
  {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  template <class T> static T *up(const std::vector<T> &v)
  {
    T *d = nullptr;
    ok(hipMalloc((void **) &d, std::max<size_t>(v.size(), 1) * sizeof(T)), "hipMalloc");
    if (!v.empty()) ok(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy");
    return d;
  }
  void stage(const System &s)
  {
    ok(hipSetDevice(0), "hipSetDevice");
    ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    d_x = up(s.x);
    d_f = up(s.f);
    d_type = up(s.type);
    d_ilist = up(s.ilist);
    d_numneigh = up(s.numneigh);
    for (int i = 0; i < s.nlocal; i++) maxneighs = std::max(maxneighs, s.numneigh[i]);
    maxneighs += 3;
    std::vector<int> view((size_t) s.nall * maxneighs);
    for (size_t k = 0; k < view.size(); k++) view[k] = (int) ((k * 2654435761u) % (unsigned) s.nall);   // padding: garbage ids
    for (int i = 0; i < s.nlocal; i++)
      for (int jj = 0; jj < s.numneigh[i]; jj++)
        view[(size_t) i + (size_t) jj * s.nall] = s.rows[i][jj] | (int) ((unsigned) ((i + jj) & 3) << 30);   // special bits
    d_neighbors = up(view);
  }
  DeviceAtomView atoms(const System &s) const
  {
    DeviceAtomView a;
    a.d_x = d_x;
    a.d_f = d_f;
    a.d_type = d_type;
    a.nlocal = s.nlocal;
    a.nall = s.nall;
    a.natoms = s.nlocal;
    a.stream = stream;
    return a;
  }
  DeviceNeighListView list(const System &s) const
  {
    DeviceNeighListView l;
    l.inum = s.nlocal;
    l.d_ilist = d_ilist;
    l.d_numneigh = d_numneigh;
    l.d_neighbors = d_neighbors;
    l.stride_i = 1;
    l.stride_jj = s.nall;
    l.maxneighs = maxneighs;
    return l;
  }
  void forces_to(System &s) const
  {
    ok(hipStreamSynchronize(stream), "sync");
    ok(hipMemcpy(s.f.data(), d_f, s.f.size() * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy f");
  }
  void zero_forces(const System &s) const { ok(hipMemsetAsync(d_f, 0, s.f.size() * sizeof(double), stream), "memset f"); }
};

static PairMTP::Style style_of(const std::string &s)
{
  if (s == "mtp" || s == "mtp/extrapolation") return PairMTP::MTP;
  if (s == "mtp/kk" || s == "mtp/extrapolation/kk") return PairMTP::MTP_KK;
  if (s == "mtp/small/kk" || s == "mtp/extrapolation/small/kk") return PairMTP::MTP_SMALL_KK;
  throw std::runtime_error("unknown style " + s);
}

template <class F> static bool throws(F &&fn, const char *needle)
{
  try {
    fn();
  } catch (const Error &e) {
    if (std::strstr(e.what(), needle)) return true;
    std::printf("unexpected message: %s\n", e.what());
    return false;
  }
  std::printf("no exception, expected: %s\n", needle);
  return false;
}

static int check_args(const char *l8, const char *nbh)
{
  int bad = 0;
  char a0[512], a1[] = "chunksize", a2[] = "32768", a3[] = "oops", a4[] = "out.cfg", a5[] = "2.0", a6[] = "10.0";
  std::snprintf(a0, sizeof(a0), "%s", l8);
  {   // pair_style mtp: one argument; extras are ignored (pair_mtp.cpp:285-297)
    PairMTP p;
    char *none[1] = {nullptr};
    bad += !throws([&] { p.settings(0, none); }, "only accepts 1 argument");
    char *two[2] = {a0, a3};
    p.settings(2, two);
    bad += p.info.alpha_index_basic_count != 11;
    char *c2[2] = {a3, a3};
    p.coeff(2, c2);
    bad += !throws([&] { p.coeff(3, c2); }, "Only \"pair_coeff * *\" is permitted");
    bad += p.init_one(1, 1) != 5.0;
    bad += !throws([&] { p.init_one(1, 2); }, "Not all pair coeffs are set. See types 1-2.");
    bad += !throws([&] { p.init_style(0); }, "requires Newton Pair on");
  }
  {   // mtp/kk: exactly <file> chunksize <N> (KOKKOS/pair_mtp_kokkos.cpp:113-117)
    PairMTP p(PairMTP::MTP_KK);
    char *one[1] = {a0};
    bad += !throws([&] { p.settings(1, one); }, "requires 3 arguments");
    char *wrong[3] = {a0, a3, a2};
    bad += !throws([&] { p.settings(3, wrong); }, "requires 3 arguments");
    char *ok[3] = {a0, a1, a2};
    p.settings(3, ok);
  }
  {   // mtp/extrapolation: 1 or 4 arguments, optional trailing chunksize pair (pair_mtp_extrapolation.cpp:488-502)
    std::snprintf(a0, sizeof(a0), "%s", nbh);
    PairMTPExtrapolation p;
    char *two[2] = {a0, a4};
    bad += !throws([&] { p.settings(2, two); }, "only accepts 1 argument");
    char *three[3] = {a0, a1, a2};
    p.settings(3, three);
    char *six[6] = {a0, a4, a5, a6, a1, a2};
    p.settings(6, six);
    int dim = -1, ncol = -1;
    bad += p.extract("extrapolation_flag", dim) != (void *) &p.extrapolation_flag || dim != 0;
    bad += p.extract("nope", dim) != nullptr;
    bad += p.extract_peratom("nope", ncol) != nullptr;
    char *badnum[4] = {a0, a4, a3, a6};
    bad += !throws([&] { p.settings(4, badnum); }, "Expected floating point parameter");
    std::snprintf(a0, sizeof(a0), "%s", l8);   // a plain .mtp has no selection state
    char *one[1] = {a0};
    bad += !throws([&] { p.settings(1, one); }, "No selection state found");
  }
  std::remove("out.cfg");
  std::printf(bad ? "ARGS FAILED %d\n" : "ARGS OK\n", bad);
  return bad;
}

// ---- the .cfg writer over several ranks (no GPU): the ranks of one "job" run one after the other in this process and
// talk through an in-memory mailbox that stands for MPI_Scan / MPI_Send / MPI_Recv; the file of a 1-, 2- and 3-rank
// job over the same atoms must be the same, byte for byte (pair_mtp_extrapolation.cpp:401-479)
struct Mailbox {
  std::vector<int> counts;              // inum of every rank
  std::vector<std::string> sent;        // what rank r handed to rank 0
  int me = 0;
};
static int mb_scan(int value, void *ctx)
{
  Mailbox *m = (Mailbox *) ctx;
  int s = 0;
  for (int r = 0; r <= m->me; r++) s += m->counts[r];
  return m->counts[m->me] == value ? s : -1000000;
}
static void mb_send(const char *buf, size_t n, void *ctx)
{
  Mailbox *m = (Mailbox *) ctx;
  m->sent[m->me].assign(buf, n);
}
static void mb_recv(int src, std::string &out, void *ctx) { out = ((Mailbox *) ctx)->sent[src]; }

static int write_cfg_jobs(const char *sysfile, const char *outdir, int mode_cfg)
{
  System s;
  s.read(sysfile);
  std::vector<double> grades(s.nlocal);
  for (int i = 0; i < s.nlocal; i++) grades[i] = 0.37 * i + 1.0 / (i + 3.0);
  CfgBox box;
  box.xprd = s.box[0];
  box.yprd = s.box[1];
  box.zprd = s.box[2];
  box.xy = 0.25;
  box.xz = -0.5;
  box.yz = 0.125;
  for (int nprocs = 1; nprocs <= 3; nprocs++) {
    Mailbox mb;
    mb.counts.resize(nprocs);
    mb.sent.resize(nprocs);
    std::vector<int> start(nprocs + 1, 0);
    for (int r = 0; r < nprocs; r++) {   // uneven shards, one of them possibly empty
      start[r + 1] = r + 1 == nprocs ? s.nlocal : (nprocs == 3 && r == 1 ? start[r] : (r + 1) * s.nlocal / (nprocs + 1));
      mb.counts[r] = start[r + 1] - start[r];
    }
    const std::string path = std::string(outdir) + "/cfg_" + std::to_string(nprocs) + ".cfg";
    std::FILE *fp = std::fopen(path.c_str(), "w");
    if (!fp) return 1;
    for (int pass = 0; pass < 2; pass++)     // senders first, then the root: a valid schedule of the message passing
      for (int r = nprocs - 1; r >= 0; r--) {
        if ((pass == 0) == (r == 0)) continue;
        mb.me = r;
        CfgComm cc;
        cc.me = r;
        cc.nprocs = nprocs;
        cc.ctx = &mb;
        cc.scan_sum = mb_scan;
        cc.send_to_root = mb_send;
        cc.recv_on_root = mb_recv;
        cfg_write_record(r == 0 ? fp : nullptr, cc, s.nlocal, box, mode_cfg != 0, mb.counts[r], s.type.data() + start[r],
                         s.x.data() + 3 * (size_t) start[r], grades.data() + start[r], 3.14159265);
      }
    std::fclose(fp);
  }
  std::printf("%s%s%s", log_scaling(1.0).c_str(), log_species(2).c_str(),
              (log_extrapolation_mode(true, false, 2.0, 10.5) + log_extrapolation_mode(false, true, 0, 0)).c_str());
  return 0;
}

int main(int argc, char **argv)
{
  try {
    if (argc >= 4 && !std::strcmp(argv[1], "args")) return check_args(argv[2], argv[3]);
    if (argc >= 5 && !std::strcmp(argv[1], "cfg")) return write_cfg_jobs(argv[2], argv[3], std::atoi(argv[4]));
    if (argc < 6) {
      std::fprintf(stderr, "usage: see the header of this file\n");
      return 2;
    }
    const bool dev = !std::strcmp(argv[1], "rundev") || !std::strcmp(argv[1], "runextdev");
    const bool ext = !std::strcmp(argv[1], "runext") || !std::strcmp(argv[1], "runextdev");
    System s;
    s.read(argv[3]);
    DeviceSystem ds;
    if (dev) ds.stage(s);
    // VIRIAL_FDOTR (2): what LAMMPS passes with newton_pair on unless the style opts out; the pair style must tally
    // its own virial on the raw flag (pair_mtp.cpp:257)
    const int vflag_run = dev ? (2 | 4) : 4;
    AtomView av;
    av.x = s.x.data();
    av.f = s.f.data();
    av.type = s.type.data();
    av.nlocal = s.nlocal;
    av.nall = s.nall;
    av.natoms = s.nlocal;
    NeighListView lv{s.nlocal, s.ilist.data(), s.numneigh.data(), s.firstneigh.data()};
    char star[] = "*";
    char *cf[2] = {star, star};
    std::ofstream out(argv[4]);
    out.precision(17);
    if (!ext) {
      PairMTP p(style_of(argv[2]));
      p.settings(argc - 5, argv + 5);
      p.coeff(2, cf);
      p.init_style(1);
      double cut = p.init_one(1, 1);
      if (dev) {
        p.bind_device(ds.atoms(s));
        p.set_neighbor_list_device(ds.list(s));
      } else {
        p.bind(av);
        p.set_neighbor_list(lv);
      }
      p.compute(3, vflag_run);
      if (dev) ds.forces_to(s);
      out << p.eng_vdwl << " " << cut << "\n";
      for (int q = 0; q < 6; q++) out << p.virial[q] << (q == 5 ? "\n" : " ");
      for (int i = 0; i < s.nall; i++)
        out << s.f[3 * i] << " " << s.f[3 * i + 1] << " " << s.f[3 * i + 2] << " " << p.eatom[i] << "\n";
    } else {
      PairMTPExtrapolation p(style_of(argv[2]));
      p.settings(argc - 5, argv + 5);
      p.coeff(2, cf);
      BoxView b;
      b.xprd = s.box[0];
      b.yprd = s.box[1];
      b.zprd = s.box[2];
      p.set_box(b);
      p.init_style(1);
      if (dev) {
        p.bind_device(ds.atoms(s));
        p.set_neighbor_list_device(ds.list(s));
      } else {
        p.bind(av);
        p.set_neighbor_list(lv);
      }
      int dim = 0, ncol = 0;
      // no grade requested: plain forces (pair_mtp_extrapolation.cpp:71-74) unless MLIP-3 style
      p.compute(3, 0);
      const double e_plain = p.eng_vdwl;
      std::fill(s.f.begin(), s.f.end(), 0.0);
      if (dev) ds.zero_forces(s);
      *(int *) p.extract("extrapolation_flag", dim) = 1;   // what `fix pair` does
      bool stopped = false;
      std::string why;
      try {
        p.compute(3, 0);
      } catch (const Error &e) {
        stopped = true;
        why = e.what();
      }
      out << p.eng_vdwl << " " << e_plain << " " << p.pvector[0] << " " << (stopped ? 1 : 0) << "\n";
      if (!p.info.configuration_mode) {
        const double *g = (const double *) p.extract_peratom("extrapolation", ncol);
        for (int i = 0; i < s.nlocal; i++) out << g[i] << "\n";
      }
      if (stopped) std::printf("stopped: %s", why.c_str());
    }
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "ERROR: %s\n", e.what());
    return 1;
  }
}
