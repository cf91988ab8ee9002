// Sanitizer driver for the host-only parts of libmtp_mi355x ("ASan/UBSan on host code"): the
// MLIP-3 parser and the native schedule builder (csrc/mtp_potential.cpp: parse, finalize -- levels, bank optimiser,
// head x tail blocks), the .cfg writer and log lines (host/mtp_cfg_writer.hpp).  Built with
// -fsanitize=address,undefined by `make -C lammps_mtp_kokkos_amd/host san`; tests/test_sanitizers_cpu.py feeds it
// every committed potential plus truncated and corrupted variants.  No GPU, no HIP runtime.
//
//   test_parser_san <file> <want_selection 0|1>      prints "OK B T S A C levels" or "ERR <code> <message>"
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../lammps_mtp_kokkos_amd/csrc/mtp_potential.hpp"
#include "../../lammps_mtp_kokkos_amd/host/mtp_cfg_writer.hpp"

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  mtp_potential pot;
  std::string err;
  const int rc = mtp_parse_file(argv[1], std::atoi(argv[2]) != 0, pot, err);
  if (rc != 0) {
    std::printf("ERR %d %s\n", rc, err.c_str());
    return 0;
  }
  // touch what finalize built (reads under ASan)
  long long sum = 0;
  for (const MtpRow &r : pot.rows_by_level) sum += r.a0 + r.a1 + r.a3 + r.mult;
  for (int v : pot.fwd_blocks) sum += v;
  for (int v : pot.basic_tgt) sum += v;
  for (int v : pot.moment_perm) sum += v;
  std::string lines;
  const int type[3] = {1, 2, 1};
  const double x[9] = {0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4}, g[3] = {0.1, 0.2, 0.3};
  mtp_mi355x::cfg_atom_lines(lines, 3, type, x, g, 5);
  sum += (long long) lines.size() + (long long) mtp_mi355x::log_extrapolation_mode(true, false, 2.0, 1e-5).size();
  std::printf("OK %d %d %d %d %d %d %lld\n", pot.alpha_index_basic_count, pot.alpha_index_times_count, pot.alpha_scalar_count,
              pot.alpha_moment_count, pot.coeff_count, (int) pot.level_offset.size() - 1, sum);
  return 0;
}
