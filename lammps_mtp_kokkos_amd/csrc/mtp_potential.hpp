// Host model of one MLIP-3 potential: the state PairMTP keeps after read_file
// (/root/reference/LAMMPS/ML-MTP/pair_mtp.h:47-83, pair_mtp_extrapolation.h:44-66) plus the
// native execution schedule derived from it (dependency levels of the times table,
// radial slots, packed basic descriptors).  Internal to libmtp_mi355x.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

struct MtpRow {
  int32_t a0, a1, mult, a3;
};

struct mtp_potential {
  // --- as read -------------------------------------------------------------------------
  std::string potential_name, potential_tag;
  double scaling = 1.0, min_cutoff = 0.0, max_cutoff = 0.0;
  int species_count = 0, radial_basis_size = 0, radial_func_count = 0;
  int alpha_moment_count = 0, alpha_index_basic_count = 0, alpha_index_times_count = 0;
  int alpha_scalar_count = 0, max_alpha_index_basic = 0;
  std::vector<int32_t> alpha_index_basic;    // [B][4]
  std::vector<int32_t> alpha_index_times;    // [T][4]
  std::vector<int32_t> alpha_moment_mapping; // [S]
  std::vector<double> radial_basis_coeffs;   // [(t1*Sp+t2)*Mu*R + mu*R + ri]
  std::vector<double> linear_coeffs, species_coeffs;
  std::vector<int32_t> setflag;              // [(Sp+1)^2], pair_mtp.cpp:455
  bool has_selection = false, configuration_mode = false;
  int coeff_count = 0;
  std::vector<double> active_set, inverse_active_set;   // [C][C]

  // --- native schedule (built by finalize) -------------------------------------------------
  // times rows stably sorted by dependency level; level_offset[l]..level_offset[l+1]
  std::vector<MtpRow> rows_by_level;
  std::vector<int32_t> level_offset;
  // LDS numbering of the moments (moment_perm[file index] = LDS index; identity on the basics) chosen to spread
  // the product passes over the LDS banks; rows_by_level and seed_idx are already in LDS numbering,
  // mapping_lds = alpha_moment_mapping in LDS numbering
  std::vector<int32_t> moment_perm, mapping_lds, basic_pack_lds;
  // distinct (mu, nu) pairs used by the basics -> slot; slot_of[mu*P+nu] or -1
  std::vector<int32_t> slot_of;
  int slot_count = 0;
  // slots sorted by tensor rank nu (then mu): rank d owns slots [deg_first[d], deg_first[d+1]);
  // slot_coef_off[s] = first double of the slot's derivative-polynomial coefficient block,
  // deg_coef[d] = that of rank d's first slot, coef_total = doubles of all blocks
  std::vector<int32_t> slot_coef_off, slot_mu;
  int deg_first[14] = {0}, deg_coef[14] = {0};
  int coef_total = 0;
  // per basic: {tx | ty << 16, tz | fa << 16 | fb << 20 | fc << 24}: coefficient entries (0xffff = none) that
  // receive fa*D, fb*D, fc*D; coef_dense = every entry has a source (no zero fill needed)
  std::vector<int32_t> basic_tgt;
  int coef_dense = 0;
  // Basic-moment pass in 3 x 3 register blocks: a basic (slot s; a, b, c) is head (s, a) x tail (b, c) with
  // g_s x^a as the head value and y^b z^c as the tail value; all heads of a block share j = nu_s - a = b + c.
  // 8 ints per block: {s0 | s1<<8 | s2<<16, a0 | a1<<4 | a2<<8 | b0<<12 | b1<<16 | b2<<20, c0 | c1<<4 | c2<<8,
  // then nine int16 basic indices (head-major, -1 = no such basic) in 4.5 ints, padded}
  std::vector<int32_t> fwd_blocks;
  int fwd_block_count = 0;
  // per basic: slot | a<<8 | b<<12 | c<<16 | mu<<20 (what a lane needs per k)
  std::vector<int32_t> basic_pack;
  // adjoint seeds: D[idx] = val (last mapping entry wins, pair_mtp.cpp:217-218)
  std::vector<int32_t> seed_idx;
  std::vector<double> seed_val;

  // gather programs of the product passes (finalize): operations {a0 = X index, a1 = Y index, mult, a3 = target} in
  // LDS numbering, 64 per block, blocks in execution order; seg_* = per executed level {first block, groups, chunk
  // size, 0} (a group = chunk-size consecutive blocks: lane l's chunk is operation l of each of them)
  std::vector<MtpRow> prog_fwd, prog_bwd;
  std::vector<int32_t> seg_fwd, seg_bwd;

  // Leaf moments: products that times rows write but never read (always scalars of the basis: pair_mtp.cpp:204-233
  // uses them in the energy sum and as adjoint seeds only).  Neither their value nor their adjoint needs an LDS slot:
  // a row into leaf t adds linear_coeff(t) mult M[a0] M[a1] to the site energy and its reverse terms are
  // D[a0] += seed(t) mult M[a1], D[a1] += seed(t) mult M[a0] with a constant seed.  The leaf rows are the LAST entry of
  // level_offset (after the normal_levels dependency levels; possibly empty), leaf_cf / leaf_cb hold the two constants
  // per row of that block (energy: sum of the coefficients mapped to t; adjoint: the last one, :217-218), and the
  // leaves are numbered [stored_moment_count, A) in LDS numbering (only grade calls keep their values: candidate vector).
  // e_map / e_lin / seed_* list the scalars of stored moments only; mapping_lds keeps all of them.
  int stored_moment_count = 0, normal_levels = 0;
  std::vector<double> leaf_cf, leaf_cb;
  std::vector<int32_t> e_map;
  std::vector<double> e_lin;

  int finalize(std::string &err);
};

int mtp_parse_file(const char *path, bool want_selection, mtp_potential &pot, std::string &err);
