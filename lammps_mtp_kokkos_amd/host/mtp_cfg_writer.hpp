// MLIP-3 `.cfg` record of the current configuration, as `pair_style mtp/extrapolation F OUT SEL BRK` writes it when
// the grade reaches the selection threshold (/root/reference/LAMMPS/ML-MTP/pair_mtp_extrapolation.cpp:401-479),
// written transport-free: the three exchanges the reference makes with MPI (MPI_Scan of the atom counts :415-416,
// MPI_Reduce of the buffer sizes :438, MPI_Send / MPI_Recv of the per-rank atom lines to rank 0 :461-474) go
// through callbacks, so the same code serves the host mirror (any transport, or none on one rank) and the LAMMPS
// plugin (MPI).  Header-only, no GPU or LAMMPS dependency.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace mtp_mi355x {

struct CfgBox {   // domain->xprd ... (:449-451)
  double xprd = 0, yprd = 0, zprd = 0, xy = 0, xz = 0, yz = 0;
};

struct CfgComm {
  int me = 0, nprocs = 1;
  void *ctx = nullptr;
  // inclusive prefix sum of `value` over the ranks (MPI_Scan, MPI_SUM); unset = one rank
  int (*scan_sum)(int value, void *ctx) = nullptr;
  // ranks != 0: hand this rank's atom lines to rank 0 (MPI_Send)
  void (*send_to_root)(const char *buf, size_t n, void *ctx) = nullptr;
  // rank 0: the lines of rank `src` (MPI_Recv + MPI_Get_count); called for src = 1 .. nprocs-1, in that order
  void (*recv_on_root)(int src, std::string &out, void *ctx) = nullptr;
};

// one line per owned atom (:418-433): global id (1-based, offset by the ranks before this one), 0-based type,
// position, and in neighbourhood mode the atom's grade.  Like the reference, atoms are taken as 0 .. inum-1.
inline void cfg_atom_lines(std::string &out, int inum, const int *type, const double *x /*[.][3]*/,
                           const double *grades /*null: configuration mode*/, int index_offset)
{
  char line[192];
  for (int i = 0; i < inum; i++) {
    const double *xi = x + 3 * (size_t) i;
    int n;
    if (grades)
      n = std::snprintf(line, sizeof(line), "%d\t%d\t%.6f\t%.6f\t%.6f\t%.5f\n", i + index_offset + 1, type[i] - 1, xi[0],
                        xi[1], xi[2], grades[i]);
    else
      n = std::snprintf(line, sizeof(line), "%d\t%d\t%.6f\t%.6f\t%.6f\n", i + index_offset + 1, type[i] - 1, xi[0], xi[1],
                        xi[2]);
    if (n > 0) out.append(line, (size_t) (n < (int) sizeof(line) ? n : (int) sizeof(line) - 1));
  }
}

// The whole record; collective over the ranks of `comm` (every rank calls it, rank 0 holds `fp`).
inline void cfg_write_record(std::FILE *fp, const CfgComm &comm, long natoms, const CfgBox &box, bool configuration_mode,
                             int inum, const int *type, const double *x, const double *grades, double max_grade)
{
  int index_offset = 0;
  if (comm.scan_sum) index_offset = comm.scan_sum(inum, comm.ctx) - inum;   // :415-416
  std::string mine;
  cfg_atom_lines(mine, inum, type, x, configuration_mode ? nullptr : grades, index_offset);
  if (comm.me != 0) {
    if (comm.send_to_root) comm.send_to_root(mine.data(), mine.size(), comm.ctx);   // :461-462
    return;
  }
  // rank 0 always receives what the other ranks sent, even without a file to write to (a closed or failed file must
  // not leave their MPI_Send unmatched)
  std::vector<std::string> theirs(comm.nprocs > 1 ? (size_t) comm.nprocs - 1 : 0);
  if (comm.recv_on_root)
    for (int src = 1; src < comm.nprocs; src++) comm.recv_on_root(src, theirs[(size_t) src - 1], comm.ctx);   // :464-472
  if (!fp) return;
  std::fprintf(fp, "BEGIN_CFG\n");   // :444-459
  std::fprintf(fp, "Size\n");
  std::fprintf(fp, "%ld\n", natoms);
  std::fprintf(fp, "Supercell\n");
  std::fprintf(fp, "%.6f %.6f %.6f\n", box.xprd, 0.0, 0.0);
  std::fprintf(fp, "%.6f %.6f %.6f\n", box.xy, box.yprd, 0.0);
  std::fprintf(fp, "%.6f %.6f %.6f\n", box.xz, box.yz, box.zprd);
  if (!configuration_mode)
    std::fprintf(fp, "AtomData:  id type       cartes_x      cartes_y      cartes_z       nbh_grades\n");
  else
    std::fprintf(fp, "AtomData:  id type       cartes_x      cartes_y      cartes_z\n");
  std::fwrite(mine.data(), 1, mine.size(), fp);
  for (const std::string &t : theirs) std::fwrite(t.data(), 1, t.size(), fp);   // rank order, so ids ascend through the file
  std::fprintf(fp, "Feature   MV_grade\t%.6f\n", max_grade);   // :474-477
  std::fprintf(fp, "END_CFG\n\n");
  std::fflush(fp);
}

// utils::logmesg lines of the reference (pair_mtp.cpp:383, 389; pair_mtp_extrapolation.cpp:508-517), rank 0 only there
inline std::string log_scaling(double scaling)
{
  char b[96];
  std::snprintf(b, sizeof(b), "The scaling is : %.2e.\n", scaling);
  return b;
}
inline std::string log_species(int species_count)
{
  char b[96];
  std::snprintf(b, sizeof(b), "There are %d species.\n", species_count);
  return b;
}
// fmt's "{}" of a double: the shortest decimal string that reads back as the same double ("2", "0.5", "1e-05")
inline std::string cfg_shortest(double v)
{
  char b[40];
  for (int p = 1; p <= 17; p++) {
    std::snprintf(b, sizeof(b), "%.*g", p, v);
    if (std::strtod(b, nullptr) == v) break;
  }
  return b;
}
inline std::string log_extrapolation_mode(bool mlip3_style, bool configuration_mode, double select_threshold,
                                          double break_threshold)
{
  const char *mode = configuration_mode ? "Configuration" : "Neighborhood";
  std::string s;
  if (mlip3_style)
    s = std::string("Extrapolation Scheme: ") + mode + " mode, with a selection threshold of " +
        cfg_shortest(select_threshold) + " and break threshold of " + cfg_shortest(break_threshold) + ".\n";
  else
    s = std::string("Extrapolation Mode: ") + mode + " mode.\n";
  return s;
}

}   // namespace mtp_mi355x
