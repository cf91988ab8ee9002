// MOCK of the slice of the LAMMPS C++ API that lammps_plugin/pair_mtp_mi355x_plugin.cpp touches -- written from the
// public LAMMPS developer documentation for ONE purpose: to compile the adapter and drive it through the call sequence
// LAMMPS makes on a pair style (tests/cpp/test_plugin_mock.cpp).  It is test scaffolding of THIS repository, not
// LAMMPS, not the reference, and proves nothing about binary compatibility with a real LAMMPS build.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpi.h"

#define FLERR __FILE__, __LINE__
#define LAMMPS_VERSION "mock"

namespace LAMMPS_NS {

typedef int64_t bigint;

class LAMMPS;
class Pair;

struct MockAbort : std::runtime_error {
  explicit MockAbort(const std::string &m) : std::runtime_error(m) {}
};

class Memory {
 public:
  template <class T> T **create(T **&array, int n1, int n2, const char *)
  {
    T *data = (T *) std::calloc((size_t) n1 * n2, sizeof(T));
    array = (T **) std::malloc(sizeof(T *) * (size_t) n1);
    for (int i = 0; i < n1; i++) array[i] = data + (size_t) i * n2;
    return array;
  }
  template <class T> void destroy(T **&array)
  {
    if (!array) return;
    std::free(array[0]);
    std::free(array);
    array = nullptr;
  }
};

class Error {
 public:
  // (fmt-style arguments are appended unformatted: enough for a test driver)
  template <class... A> [[noreturn]] void all(const std::string &, int, const std::string &msg, A &&...)
  {
    throw MockAbort("ERROR: " + msg);
  }
  template <class... A> [[noreturn]] void one(const std::string &, int, const std::string &msg, A &&...)
  {
    throw MockAbort("ERROR on proc 0: " + msg);
  }
};

class Atom {
 public:
  double **x = nullptr, **f = nullptr;
  int *type = nullptr;
  int nlocal = 0, nghost = 0;
  bigint natoms = 0;
};
class Comm {
 public:
  int me = 0, nprocs = 1;
};
class Force {
 public:
  int newton_pair = 1;
};
class Domain {
 public:
  double xprd = 0, yprd = 0, zprd = 0, xy = 0, xz = 0, yz = 0;
};
class NeighList {
 public:
  int inum = 0;
  int *ilist = nullptr, *numneigh = nullptr;
  int **firstneigh = nullptr;
};
namespace NeighConst {
enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0 };
}
class NeighRequest {};
class Neighbor {
 public:
  int ago = 0;
  int last_request_flags = -1;
  NeighRequest req;
  NeighRequest *add_request(Pair *, int flags = 0)
  {
    last_request_flags = flags;
    return &req;
  }
};

class LAMMPS {
 public:
  Memory *memory = new Memory;
  Error *error = new Error;
  Atom *atom = new Atom;
  Comm *comm = new Comm;
  Force *force = new Force;
  Domain *domain = new Domain;
  Neighbor *neighbor = new Neighbor;
  MPI_Comm world = 0;
  void *kokkos = nullptr;
  std::string log;
};

class Pointers {
 public:
  explicit Pointers(LAMMPS *ptr)
      : lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), comm(ptr->comm), force(ptr->force),
        domain(ptr->domain), neighbor(ptr->neighbor), world(ptr->world)
  {
  }
  virtual ~Pointers() = default;

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Comm *&comm;
  Force *&force;
  Domain *&domain;
  Neighbor *&neighbor;
  MPI_Comm &world;
};

namespace utils {
inline void logmesg(LAMMPS *lmp, const std::string &mesg)
{
  lmp->log += mesg;
  std::fputs(mesg.c_str(), stdout);
}
inline std::string get_potential_file_path(const std::string &path)
{
  if (FILE *fp = std::fopen(path.c_str(), "r")) {
    std::fclose(fp);
    return path;
  }
  if (const char *dir = std::getenv("LAMMPS_POTENTIALS")) {
    const std::string alt = std::string(dir) + "/" + path;
    if (FILE *fp = std::fopen(alt.c_str(), "r")) {
      std::fclose(fp);
      return alt;
    }
  }
  return "";
}
}   // namespace utils

// pair.h: the members and the ev_init semantics the adapter relies on (energy / virial bit flags as in pair.h:
// ENERGY_GLOBAL 1, ENERGY_ATOM 2; VIRIAL_PAIR 1, VIRIAL_FDOTR 2, VIRIAL_ATOM 4)
class Pair : protected Pointers {
 public:
  explicit Pair(LAMMPS *lmp) : Pointers(lmp) {}
  ~Pair() override
  {
    std::free(eatom);
    if (vatom) {
      std::free(vatom[0]);
      std::free(vatom);
    }
  }
  double eng_vdwl = 0.0, eng_coul = 0.0, virial[6] = {0, 0, 0, 0, 0, 0};
  double *eatom = nullptr, **vatom = nullptr;
  int single_enable = 1, restartinfo = 1, one_coeff = 0, manybody_flag = 0, no_virial_fdotr_compute = 0;
  int nextra = 0;
  double *pvector = nullptr;
  int allocated = 0;
  int **setflag = nullptr;
  double **cutsq = nullptr;
  NeighList *list = nullptr;
  int evflag = 0, eflag_either = 0, eflag_global = 0, eflag_atom = 0, vflag_either = 0, vflag_global = 0, vflag_atom = 0,
      vflag_fdotr = 0;
  int maxeatom = 0, maxvatom = 0;

  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style() {}
  virtual double init_one(int, int) { return 0.0; }
  virtual void *extract(const char *, int &) { return nullptr; }
  virtual void *extract_peratom(const char *, int &) { return nullptr; }

  void ev_init(int eflag, int vflag, int = 1)
  {
    evflag = 1;
    eflag_either = eflag;
    eflag_global = eflag & 1;
    eflag_atom = eflag & 2;
    vflag_global = vflag & 3;
    if (vflag_global == 2 && no_virial_fdotr_compute == 1) vflag_global = 1;
    vflag_fdotr = 0;
    if (vflag_global == 2) {
      vflag_fdotr = 1;
      vflag_global = 0;
    }
    vflag_atom = vflag & 4;
    vflag_either = vflag_global || vflag_atom;
    const int nall = atom->nlocal + atom->nghost;
    if (eflag_atom && nall > maxeatom) {
      std::free(eatom);
      eatom = (double *) std::calloc((size_t) nall, sizeof(double));
      maxeatom = nall;
    }
    if (vflag_atom && nall > maxvatom) {
      if (vatom) {
        std::free(vatom[0]);
        std::free(vatom);
      }
      double *d = (double *) std::calloc((size_t) nall * 6, sizeof(double));
      vatom = (double **) std::malloc(sizeof(double *) * (size_t) nall);
      for (int i = 0; i < nall; i++) vatom[i] = d + 6 * (size_t) i;
      maxvatom = nall;
    }
    eng_vdwl = eng_coul = 0.0;
    for (double &v : virial) v = 0.0;
    if (eflag_atom)
      for (int i = 0; i < nall; i++) eatom[i] = 0.0;
    if (vflag_atom)
      for (int i = 0; i < nall; i++)
        for (int q = 0; q < 6; q++) vatom[i][q] = 0.0;
  }
};

}   // namespace LAMMPS_NS

// lammpsplugin.h
extern "C" {
typedef void *(lammpsplugin_factory1)(void *);
typedef void *(lammpsplugin_factory2)(void *, int, char **);
typedef struct {
  const char *version;
  const char *style;
  const char *name;
  const char *info;
  const char *author;
  union {
    lammpsplugin_factory1 *v1;
    lammpsplugin_factory2 *v2;
  } creator;
  void *handle;
} lammpsplugin_t;
typedef void (*lammpsplugin_regfunc)(void *, void *);
void lammpsplugin_init(void *, void *, void *);
}
