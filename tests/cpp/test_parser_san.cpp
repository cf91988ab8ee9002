// Sanitizer driver for the host-only parts of libmtp_mi355x ("ASan/UBSan on host code"): the
// MLIP-3 parser and the native schedule builder (csrc/mtp_potential.cpp: parse, finalize -- levels, bank optimiser,
// head x tail blocks), the .cfg writer and log lines (host/mtp_cfg_writer.hpp).  Built with
// -fsanitize=address,undefined by `make -C lammps_mtp_kokkos_amd/host san`; tests/test_sanitizers_cpu.py feeds it
// every committed potential plus truncated and corrupted variants.  No GPU, no HIP runtime.
//
//   test_parser_san <file> <want_selection 0|1>      prints "OK B T S A C levels" or "ERR <code> <message>"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

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
  // the gather programs of the product passes against the reference's sequential semantics (pair_mtp.cpp:196-201,
  // 221-233) on pseudo-random moments: forward values and adjoints must agree to rounding
  double prog_err = 0.0;
  {
    const int A = pot.alpha_moment_count, B = pot.alpha_index_basic_count, T = pot.alpha_index_times_count;
    std::vector<double> m_ref((size_t) A, 0.0), d_ref((size_t) A, 0.0), m_g((size_t) A, 0.0), d_g((size_t) A, 0.0);
    unsigned long long rng = 12345;
    auto next = [&]() {
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      return (double) (rng >> 11) / 9007199254740992.0 - 0.5;
    };
    for (int k = 0; k < B; k++) m_ref[(size_t) k] = next();
    for (int k = 0; k < T; k++) {   // file order, file numbering
      const int32_t *q = &pot.alpha_index_times[4 * (size_t) k];
      m_ref[(size_t) q[3]] += q[2] * m_ref[(size_t) q[0]] * m_ref[(size_t) q[1]];
    }
    std::vector<double> seed((size_t) A, 0.0);
    for (int i = 0; i < pot.alpha_scalar_count; i++) seed[(size_t) pot.alpha_moment_mapping[(size_t) i]] = 0.1 + 0.01 * i;
    d_ref = seed;
    for (int k = T - 1; k >= 0; k--) {
      const int32_t *q = &pot.alpha_index_times[4 * (size_t) k];
      const double d3 = d_ref[(size_t) q[3]] * q[2];
      d_ref[(size_t) q[1]] += d3 * m_ref[(size_t) q[0]];
      d_ref[(size_t) q[0]] += d3 * m_ref[(size_t) q[1]];
    }
    // gather programs, LDS numbering (moment_perm[file index] = LDS index)
    for (int k = 0; k < B; k++) m_g[(size_t) pot.moment_perm[(size_t) k]] = m_ref[(size_t) k];
    for (int k = 0; k < A; k++) d_g[(size_t) pot.moment_perm[(size_t) k]] = seed[(size_t) k];
    auto run = [&](const std::vector<MtpRow> &prog, const std::vector<int32_t> &seg, const std::vector<double> &X,
                   const std::vector<double> &Y, std::vector<double> &Tt) {
      for (size_t l = 0; 4 * l < seg.size(); l++) {
        const int first = seg[4 * l], groups = seg[4 * l + 1], cs = seg[4 * l + 2];
        std::vector<double> add((size_t) A, 0.0);   // a level reads the state before the level (rows of a level commute)
        for (int g = 0; g < groups; g++)
          for (int lane = 0; lane < 64; lane++) {
            double acc = 0.0;
            int tgt = -1;
            for (int u = 0; u < cs; u++) {
              const MtpRow &o = prog[((size_t) first + (size_t) g * cs + u) * 64 + lane];
              if (u == 0) tgt = o.a3;
              acc += (double) o.mult * X[(size_t) o.a0] * Y[(size_t) o.a1];
            }
            add[(size_t) tgt] += acc;
          }
        for (int k = 0; k < A; k++) Tt[(size_t) k] += add[(size_t) k];
      }
    };
    {
      std::vector<double> snapshot;
      // forward: X = Y = T = moments; run level by level on the live array (targets of a level are not read in it)
      for (size_t l = 0; 4 * l < pot.seg_fwd.size(); l++) {
        std::vector<int32_t> one(pot.seg_fwd.begin() + 4 * (long) l, pot.seg_fwd.begin() + 4 * (long) l + 4);
        snapshot = m_g;
        run(pot.prog_fwd, one, snapshot, snapshot, m_g);
      }
      for (size_t l = 0; 4 * l < pot.seg_bwd.size(); l++) {
        std::vector<int32_t> one(pot.seg_bwd.begin() + 4 * (long) l, pot.seg_bwd.begin() + 4 * (long) l + 4);
        snapshot = d_g;
        run(pot.prog_bwd, one, snapshot, m_g, d_g);
      }
    }
    for (int k = 0; k < A; k++) {
      const double sm = std::fabs(m_ref[(size_t) k]) + 1.0, sd = std::fabs(d_ref[(size_t) k]) + 1.0;
      prog_err = std::fmax(prog_err, std::fabs(m_g[(size_t) pot.moment_perm[(size_t) k]] - m_ref[(size_t) k]) / sm);
      prog_err = std::fmax(prog_err, std::fabs(d_g[(size_t) pot.moment_perm[(size_t) k]] - d_ref[(size_t) k]) / sd);
    }
  }
  std::string lines;
  const int type[3] = {1, 2, 1};
  const double x[9] = {0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4}, g[3] = {0.1, 0.2, 0.3};
  mtp_mi355x::cfg_atom_lines(lines, 3, type, x, g, 5);
  sum += (long long) lines.size() + (long long) mtp_mi355x::log_extrapolation_mode(true, false, 2.0, 1e-5).size();
  std::printf("OK %d %d %d %d %d %d %lld %.3e\n", pot.alpha_index_basic_count, pot.alpha_index_times_count, pot.alpha_scalar_count,
              pot.alpha_moment_count, pot.coeff_count, (int) pot.level_offset.size() - 1, sum, prog_err);
  return 0;
}
