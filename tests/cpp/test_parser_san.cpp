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
  // the native schedule of the product passes -- row-per-lane levels and gather programs, each followed by the leaf
  // rows -- against the reference's sequential semantics (pair_mtp.cpp:196-233) on pseudo-random basic moments: the
  // stored moments, their adjoints and the energy sum must agree to rounding
  double prog_err = 0.0;
  {
    const int A = pot.alpha_moment_count, B = pot.alpha_index_basic_count, T = pot.alpha_index_times_count;
    const int S = pot.alpha_scalar_count, Ast = pot.stored_moment_count, NL = pot.normal_levels;
    if (Ast < B || Ast > A || NL + 2 != (int) pot.level_offset.size() || pot.seg_fwd.size() != (size_t) 4 * NL ||
        pot.leaf_cf.size() != pot.rows_by_level.size() - (size_t) pot.level_offset[(size_t) NL] ||
        pot.leaf_cb.size() != pot.leaf_cf.size() || pot.e_map.size() != pot.e_lin.size()) {
      std::printf("ERR -1 inconsistent native schedule\n");
      return 0;
    }
    std::vector<double> m_ref((size_t) A, 0.0), d_ref((size_t) A, 0.0);
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
    double e_ref = 0.0;
    for (int i = 0; i < S; i++) {
      e_ref += pot.linear_coeffs[(size_t) i] * m_ref[(size_t) pot.alpha_moment_mapping[(size_t) i]];
      d_ref[(size_t) pot.alpha_moment_mapping[(size_t) i]] = pot.linear_coeffs[(size_t) i];   // assignment: the last one wins
    }
    for (int k = T - 1; k >= 0; k--) {
      const int32_t *q = &pot.alpha_index_times[4 * (size_t) k];
      const double d3 = d_ref[(size_t) q[3]] * q[2];
      d_ref[(size_t) q[1]] += d3 * m_ref[(size_t) q[0]];
      d_ref[(size_t) q[0]] += d3 * m_ref[(size_t) q[1]];
    }
    // one level of a gather program: reads the state before the level (its operations commute)
    auto run = [&](const std::vector<MtpRow> &prog, const int32_t *seg, const std::vector<double> &X,
                   const std::vector<double> &Y, std::vector<double> &Tt) {
      const int first = seg[0], groups = seg[1], cs = seg[2];
      std::vector<double> add((size_t) Ast, 0.0);
      for (int g = 0; g < groups; g++)
        for (int lane = 0; lane < 64; lane++) {
          double acc = 0.0;
          int tgt = -1;
          for (int u = 0; u < cs; u++) {
            const MtpRow &o = prog[((size_t) first + (size_t) g * cs + u) * 64 + lane];
            if (u == 0) tgt = o.a3;
            acc += (double) o.mult * X.at((size_t) o.a0) * Y.at((size_t) o.a1);
          }
          add.at((size_t) tgt) += acc;
        }
      for (int k = 0; k < Ast; k++) Tt[(size_t) k] += add[(size_t) k];
    };
    const size_t leaf0 = (size_t) pot.level_offset[(size_t) NL];
    for (int form = 0; form < 2; form++) {   // 0: row per lane, 1: gather programs; arrays hold the STORED moments only
      std::vector<double> m_g((size_t) Ast, 0.0), d_g((size_t) Ast, 0.0), snapshot;
      for (int k = 0; k < B; k++) m_g[(size_t) pot.moment_perm[(size_t) k]] = m_ref[(size_t) k];
      for (int l = 0; l < NL; l++) {
        snapshot = m_g;
        if (form == 1) run(pot.prog_fwd, &pot.seg_fwd[4 * (size_t) l], snapshot, snapshot, m_g);
        else
          for (int r = pot.level_offset[(size_t) l]; r < pot.level_offset[(size_t) l + 1]; r++) {
            const MtpRow &o = pot.rows_by_level[(size_t) r];
            m_g.at((size_t) o.a3) += (double) o.mult * snapshot.at((size_t) o.a0) * snapshot.at((size_t) o.a1);
          }
      }
      double e_g = 0.0;
      for (size_t r = 0; r < pot.leaf_cf.size(); r++) {
        const MtpRow &o = pot.rows_by_level[leaf0 + r];
        e_g += pot.leaf_cf[r] * m_g.at((size_t) o.a0) * m_g.at((size_t) o.a1);
      }
      for (size_t k = 0; k < pot.e_map.size(); k++) e_g += pot.e_lin[k] * m_g.at((size_t) pot.e_map[k]);
      for (size_t k = 0; k < pot.seed_idx.size(); k++) d_g.at((size_t) pot.seed_idx[k]) = pot.seed_val[k];
      for (size_t r = 0; r < pot.leaf_cb.size(); r++) {
        const MtpRow &o = pot.rows_by_level[leaf0 + r];
        d_g.at((size_t) o.a1) += pot.leaf_cb[r] * m_g[(size_t) o.a0];
        d_g.at((size_t) o.a0) += pot.leaf_cb[r] * m_g[(size_t) o.a1];
      }
      for (int l = NL - 1; l >= 0; l--) {
        snapshot = d_g;
        if (form == 1) run(pot.prog_bwd, &pot.seg_bwd[4 * (size_t) (NL - 1 - l)], snapshot, m_g, d_g);
        else
          for (int r = pot.level_offset[(size_t) l]; r < pot.level_offset[(size_t) l + 1]; r++) {
            const MtpRow &o = pot.rows_by_level[(size_t) r];
            const double d3 = snapshot.at((size_t) o.a3) * o.mult;
            d_g.at((size_t) o.a1) += d3 * m_g[(size_t) o.a0];
            d_g.at((size_t) o.a0) += d3 * m_g[(size_t) o.a1];
          }
      }
      int nstored = 0;
      for (int k = 0; k < A; k++) {
        const int kl = pot.moment_perm[(size_t) k];
        if (kl >= Ast) continue;   // leaf: no slot
        nstored++;
        const double sm = std::fabs(m_ref[(size_t) k]) + 1.0, sd = std::fabs(d_ref[(size_t) k]) + 1.0;
        prog_err = std::fmax(prog_err, std::fabs(m_g[(size_t) kl] - m_ref[(size_t) k]) / sm);
        prog_err = std::fmax(prog_err, std::fabs(d_g[(size_t) kl] - d_ref[(size_t) k]) / sd);
      }
      if (nstored != Ast) prog_err = 1.0;
      prog_err = std::fmax(prog_err, std::fabs(e_g - e_ref) / (std::fabs(e_ref) + 1.0));
    }
    sum += Ast;
  }
  std::string lines;
  const int type[3] = {1, 2, 1};
  const double x[9] = {0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4}, g[3] = {0.1, 0.2, 0.3};
  mtp_mi355x::cfg_atom_lines(lines, 3, type, x, g, 5);
  sum += (long long) lines.size() + (long long) mtp_mi355x::log_extrapolation_mode(true, false, 2.0, 1e-5).size();
  std::printf("OK %d %d %d %d %d %d %lld %.3e\n", pot.alpha_index_basic_count, pot.alpha_index_times_count, pot.alpha_scalar_count,
              pot.alpha_moment_count, pot.coeff_count, pot.normal_levels, sum, prog_err);
  return 0;
}
