// MLIP-3 potential file reader and native schedule builder (host only).
//
// Grammar and error behaviour follow PairMTP::read_file
// (/root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:335-570), RadialMTPBasis::ReadBasisProperties
// (mtp_radial_basis.cpp:59-102) and PairMTPExtrapolation::read_file
// (pair_mtp_extrapolation.cpp:528-612); SURVEY.md App. A is the condensed grammar.
#include "mtp_potential.hpp"

#include "../../include/mtp_mi355x.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace {

struct ParseError {
  int code;
  std::string msg;
};

// LAMMPS TextFileReader semantics the reference depends on: fgets into a buffer whose
// size the caller may change (pair_mtp.cpp:489-492, 525-528), '#' starts a comment when
// ignore_comments is set, lines without words are skipped.
class LineReader {
 public:
  explicit LineReader(FILE *fp) : fp_(fp), buf_(1024) {}
  bool ignore_comments = true;
  void set_bufsize(size_t n) { buf_.assign(std::max<size_t>(n, 2), '\0'); }
  // returns false at end of file
  bool next(std::string &out)
  {
    while (fgets(buf_.data(), (int) buf_.size(), fp_)) {
      out.assign(buf_.data());
      if (ignore_comments) {
        auto p = out.find('#');
        if (p != std::string::npos) out.erase(p);
      }
      if (out.find_first_not_of(" \t\r\n\f") != std::string::npos) return true;
    }
    return false;
  }

 private:
  FILE *fp_;
  std::vector<char> buf_;
};

// ValueTokenizer: split on a separator set; typed getters fail loudly.
class Tokens {
 public:
  Tokens() = default;
  Tokens(const std::string &line, const char *seps) : s_(line), seps_(seps) {}
  bool next(std::string &w)
  {
    size_t b = s_.find_first_not_of(seps_, pos_);
    if (b == std::string::npos) return false;
    size_t e = s_.find_first_of(seps_, b);
    if (e == std::string::npos) e = s_.size();
    w = s_.substr(b, e - b);
    pos_ = e;
    return true;
  }
  std::string word()
  {
    std::string w;
    if (!next(w)) throw ParseError{MTP_ERR_PARSE, "Not enough tokens"};
    return w;
  }
  int integer()
  {
    std::string w = word();
    char *e = nullptr;
    errno = 0;
    long v = std::strtol(w.c_str(), &e, 10);
    if (*e || errno) throw ParseError{MTP_ERR_PARSE, "Not a valid integer number: '" + w + "'"};
    return (int) v;
  }
  double real()
  {
    std::string w = word();
    char *e = nullptr;
    double v = std::strtod(w.c_str(), &e);
    if (*e) throw ParseError{MTP_ERR_PARSE, "Not a valid floating-point number: '" + w + "'"};
    return v;
  }

 private:
  std::string s_;
  std::string seps_;
  size_t pos_ = 0;
};

const char *kSeps = " \t\r\n\f=, ";
const char *kSepsDash = " \t\r\n\f=, -";
const char *kSepsBrace = " \t\r\n\f=, {},";

struct Cursor {
  LineReader &rd;
  std::string line, key;
  Tokens tok;
  void advance(const char *seps)
  {
    if (!rd.next(line)) throw ParseError{MTP_ERR_EOF, "Unexpected end of MTP file."};
    tok = Tokens(line, seps);
    if (!tok.next(key)) key.clear();
  }
  void expect(const char *kw, const char *msg)
  {
    if (key != kw) throw ParseError{MTP_ERR_PARSE, msg};
  }
};

void parse_text(FILE *fp, bool want_selection, mtp_potential &p)
{
  LineReader rd(fp);
  Cursor c{rd, {}, {}, {}};

  c.advance(kSeps);
  if (c.key != "MTP") throw ParseError{MTP_ERR_FORMAT, "Only MTP potential files are accepted."};
  {
    std::string ver;
    if (!rd.next(ver) || ver != "version = 1.1.0\n")   // exact, newline included (:357)
      throw ParseError{MTP_ERR_FORMAT, "MTP file must have version \"1.1.0\""};
  }
  c.advance(kSeps);
  if (c.key == "potential_name") {
    std::string w;
    p.potential_name = c.tok.next(w) ? w : "";
    c.advance(kSeps);
  }
  p.scaling = 1.0;
  if (c.key == "scaling") {
    p.scaling = c.tok.real();
    c.advance(kSeps);
  }
  c.expect("species_count", "Error reading MTP file. Species count not found.");
  p.species_count = c.tok.integer();
  if (p.species_count < 1) throw ParseError{MTP_ERR_PARSE, "species_count must be positive"};

  c.advance(kSeps);
  if (c.key == "potential_tag") {
    std::string w;
    p.potential_tag = c.tok.next(w) ? w : "";
    c.advance(kSeps);
  }
  c.expect("radial_basis_type", "Error reading MTP file. No radial basis set type is specified.");
  {
    std::string ty = c.tok.word();
    if (ty != "RBChebyshev")
      throw ParseError{MTP_ERR_UNSUPPORTED,
                       "Error reading MTP file. The specified radial basis set type, " + ty + ", was not found.."};
  }
  // radial basis block (mtp_radial_basis.cpp:59-102)
  c.advance(kSeps);
  if (c.key == "scaling") {   // parsed, then superseded by the top-level value (pair_mtp.cpp:416)
    (void) c.tok.real();
    c.advance(kSeps);
  }
  if (c.key != "min_val" && c.key != "min_dist")
    throw ParseError{MTP_ERR_PARSE, "Error in reading MTP file. Cannot read lower cutoff."};
  p.min_cutoff = c.tok.real();
  c.advance(kSeps);
  if (c.key != "max_val" && c.key != "max_dist")
    throw ParseError{MTP_ERR_PARSE, "Error in reading MTP file. Cannot read upper cutoff."};
  p.max_cutoff = c.tok.real();
  c.advance(kSeps);
  c.expect("radial_basis_size", "Error in reading MTP file. Cannot read radial basis set size.");
  p.radial_basis_size = c.tok.integer();

  c.advance(kSeps);
  c.expect("radial_funcs_count", "Error in reading MTP file. Cannot read radial function count.");
  p.radial_func_count = c.tok.integer();
  c.advance(kSeps);
  if (c.key != "radial_coeffs") {
    if (c.key == "magnetic_basis_type")
      throw ParseError{MTP_ERR_UNSUPPORTED, "Magnetic basis is currently not supported."};
    throw ParseError{MTP_ERR_PARSE, "Error in reading MTP file. Cannot read radial coeffs count."};
  }
  const int Sp = p.species_count, R = p.radial_basis_size, Mu = p.radial_func_count;
  if (R < 1 || Mu < 1) throw ParseError{MTP_ERR_PARSE, "radial basis sizes must be positive"};
  p.radial_basis_coeffs.assign((size_t) Sp * Sp * Mu * R, 0.0);
  p.setflag.assign((size_t) (Sp + 1) * (Sp + 1), 0);
  for (int blk = 0; blk < Sp * Sp; blk++) {   // any order, 0-based species (:450-469)
    if (!rd.next(c.line)) throw ParseError{MTP_ERR_EOF, "Unexpected end of MTP file."};
    Tokens hdr(c.line, kSepsDash);
    int t1 = hdr.integer(), t2 = hdr.integer();
    if (t1 < 0 || t2 < 0 || t1 >= Sp || t2 >= Sp)
      throw ParseError{MTP_ERR_PARSE, "radial_coeffs block names a species outside species_count"};
    p.setflag[(size_t) (t1 + 1) * (Sp + 1) + t2 + 1] = 1;
    double *dst = &p.radial_basis_coeffs[(size_t) (t1 * Sp + t2) * Mu * R];
    for (int mu = 0; mu < Mu; mu++) {
      if (!rd.next(c.line)) throw ParseError{MTP_ERR_EOF, "Unexpected end of MTP file."};
      Tokens row(c.line, kSepsBrace);
      for (int ri = 0; ri < R; ri++) dst[mu * R + ri] = row.real();
    }
  }
  c.advance(kSeps);
  c.expect("alpha_moments_count", "Error reading MTP file. Alpha moment count not found.");
  p.alpha_moment_count = c.tok.integer();
  c.advance(kSeps);
  c.expect("alpha_index_basic_count", "Error reading MTP file. Alpha moment count not found.");
  p.alpha_index_basic_count = c.tok.integer();
  const int B = p.alpha_index_basic_count;
  if (B < 1) throw ParseError{MTP_ERR_TABLE, "alpha_index_basic_count must be positive"};

  rd.set_bufsize((size_t) B * 20 + 20);
  c.advance(kSepsBrace);
  c.expect("alpha_index_basic", "Error reading MTP file. Alpha index basic not found.");
  p.alpha_index_basic.resize((size_t) B * 4);
  int mu_max = 0, rank_max = 0;
  for (int i = 0; i < B; i++) {
    int32_t *q = &p.alpha_index_basic[4 * (size_t) i];
    for (int j = 0; j < 4; j++) q[j] = c.tok.integer();
    mu_max = std::max(mu_max, (int) q[0]);
    rank_max = std::max(rank_max, (int) (q[1] + q[2] + q[3]));
  }
  if (mu_max != Mu - 1) throw ParseError{MTP_ERR_TABLE, "Wrong number of radial functions specified!"};
  p.max_alpha_index_basic = rank_max + 1;

  c.advance(kSeps);
  c.expect("alpha_index_times_count", "Error reading MTP file. Alpha index times count not found.");
  p.alpha_index_times_count = c.tok.integer();
  const int T = p.alpha_index_times_count;
  rd.set_bufsize((size_t) std::max(T, 0) * 32 + 20);
  c.advance(kSepsBrace);
  c.expect("alpha_index_times", "Error reading MTP file. Alpha index times not found.");
  p.alpha_index_times.resize((size_t) std::max(T, 0) * 4);
  for (size_t i = 0; i < p.alpha_index_times.size(); i++) p.alpha_index_times[i] = c.tok.integer();

  c.advance(kSeps);
  c.expect("alpha_scalar_moments", "Error reading MTP file. Alpha scalar moment count not found.");
  p.alpha_scalar_count = c.tok.integer();
  const int S = p.alpha_scalar_count;
  c.advance(kSepsBrace);
  c.expect("alpha_moment_mapping", "Error reading MTP file. Alpha moment mappings not found.");
  p.alpha_moment_mapping.resize((size_t) S);
  for (int i = 0; i < S; i++) p.alpha_moment_mapping[i] = c.tok.integer();
  c.advance(kSepsBrace);
  c.expect("species_coeffs", "Error reading MTP file. Species coefficients not found.");
  p.species_coeffs.resize((size_t) Sp);
  for (int i = 0; i < Sp; i++) p.species_coeffs[i] = c.tok.real();
  c.advance(kSepsBrace);
  c.expect("moment_coeffs", "Error reading MTP file. Moment coefficients not found.");
  p.linear_coeffs.resize((size_t) S);
  for (int i = 0; i < S; i++) p.linear_coeffs[i] = c.tok.real();

  p.coeff_count = Sp * Sp * Mu * R + Sp + S;

  if (!want_selection) return;
  // selection state (pair_mtp_extrapolation.cpp:545-612)
  rd.ignore_comments = false;
  std::string ln;
  if (!rd.next(ln))
    throw ParseError{MTP_ERR_SELECTION,
                     "No selection state found! Consider training/retraining or disabling extrapolation!"};
  {
    Tokens t(ln, kSeps);
    std::string w;
    if (!t.next(w) || w != "#MVS_v1.1")
      throw ParseError{MTP_ERR_SELECTION,
                       "Error in reading MTP file selection state. Please verify MVS version is #MVS_v1.1!"};
  }
  rd.ignore_comments = true;
  int energy_weight = 0, site_en_weight = 0;
  static const char *names[5] = {"energy_weight", "force_weight", "stress_weight", "site_en_weight",
                                 "weight_scaling"};
  for (int w = 0; w < 5; w++) {
    c.advance(kSeps);
    if (c.key != names[w])
      throw ParseError{MTP_ERR_SELECTION, std::string("Error in reading MTP file, ") + names[w]};
    if (w == 0) energy_weight = (int) c.tok.real();
    if (w == 3) site_en_weight = (int) c.tok.real();
  }
  if (energy_weight + site_en_weight > 1)
    throw ParseError{MTP_ERR_MODE,
                     "Error, the MTP currently only supports configuration mode (energy_weight=1) or "
                     "neighbourhood mode (site_en_weight=1). Please retrain the MTP with the correct modes!"};
  p.configuration_mode = (energy_weight == 1);
  const size_t n = (size_t) p.coeff_count * p.coeff_count;
  p.active_set.resize(n);
  p.inverse_active_set.resize(n);
  fgetc(fp);   // the '#' in front of the raw fp64 block (:607)
  if (fread(p.active_set.data(), sizeof(double), n, fp) != n ||
      fread(p.inverse_active_set.data(), sizeof(double), n, fp) != n)
    throw ParseError{MTP_ERR_IO, "Unexpected end of file while reading the active set"};
  p.has_selection = true;
}

}   // namespace

int mtp_parse_file(const char *path, bool want_selection, mtp_potential &pot, std::string &err)
{
  FILE *fp = std::fopen(path, "rb");
  if (!fp) {
    err = std::string("Cannot open potential file ") + path + ": " + std::strerror(errno);
    return MTP_ERR_IO;
  }
  int rc = MTP_OK;
  try {
    parse_text(fp, want_selection, pot);
  } catch (const ParseError &e) {
    err = e.msg;
    rc = e.code;
  } catch (const std::exception &e) {
    err = e.what();
    rc = MTP_ERR_PARSE;
  }
  std::fclose(fp);
  if (rc == MTP_OK) rc = pot.finalize(err);
  return rc;
}

// Build the native schedule.  The reference executes the times rows strictly in file
// order (pair_mtp.cpp:196-201) and in reverse for the adjoint (:221-233).  Rows are
// assigned to dependency levels so that any two rows in one level commute under that
// sequential semantics (read-after-write and write-after-read on the moment array are
// both respected); a level is then executed by all lanes at once.
int mtp_potential::finalize(std::string &err)
{
  const int A = alpha_moment_count, B = alpha_index_basic_count, T = alpha_index_times_count;
  const int S = alpha_scalar_count, P = max_alpha_index_basic, Mu = radial_func_count;
  if (A < B) {
    err = "alpha_moments_count is smaller than alpha_index_basic_count";
    return MTP_ERR_TABLE;
  }
  for (int i = 0; i < B; i++) {
    const int32_t *q = &alpha_index_basic[4 * (size_t) i];
    if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[3] < 0) {
      err = "negative entry in alpha_index_basic";
      return MTP_ERR_TABLE;
    }
    if (q[1] > 15 || q[2] > 15 || q[3] > 15) {
      err = "alpha_index_basic exponent above 15 is not supported";
      return MTP_ERR_LIMIT;
    }
  }
  if (max_alpha_index_basic > 12) {
    err = "tensor rank above 11 is not supported by this build";
    return MTP_ERR_LIMIT;
  }
  for (int k = 0; k < T; k++) {
    const int32_t *q = &alpha_index_times[4 * (size_t) k];
    for (int j : {0, 1, 3})
      if (q[j] < 0 || q[j] >= A) {
        err = "alpha_index_times refers to a moment outside alpha_moments_count";
        return MTP_ERR_TABLE;
      }
  }
  // leaf moments (mtp_potential.hpp): written by rows, never read by one, not a basic.  Their rows are deferred to the
  // end of the forward pass, which keeps the reference's in-order semantics only if no later row still adds to one of
  // the row's factors: a leaf with such a row stays an ordinary stored moment.
  std::vector<char> leaf((size_t) A, 0);
  if (!std::getenv("MTP_NO_LEAF")) {
    std::vector<char> is_factor((size_t) A, 0), is_target((size_t) A, 0);
    std::vector<int> last_write((size_t) A, -1);
    for (int k = 0; k < T; k++) {
      const int32_t *q = &alpha_index_times[4 * (size_t) k];
      is_factor[q[0]] = is_factor[q[1]] = 1;
      is_target[q[3]] = 1;
      last_write[q[3]] = k;
    }
    for (int m = B; m < A; m++) leaf[m] = is_target[m] && !is_factor[m];
    for (int k = 0; k < T; k++) {
      const int32_t *q = &alpha_index_times[4 * (size_t) k];
      if (leaf[q[3]] && (last_write[q[0]] > k || last_write[q[1]] > k)) leaf[q[3]] = 0;
    }
  }
  std::vector<int> wlevel((size_t) A, 0), rlevel((size_t) A, 0);
  std::vector<int> lvl((size_t) T, 0);
  int nlev = 0;
  for (int k = 0; k < T; k++) {
    const int32_t *q = &alpha_index_times[4 * (size_t) k];
    if (leaf[q[3]]) continue;
    int l = std::max(wlevel[q[0]], wlevel[q[1]]) + 1;   // operands complete
    l = std::max(l, rlevel[q[3]] + 1);                  // earlier readers of a3 come first
    lvl[k] = l;
    wlevel[q[3]] = std::max(wlevel[q[3]], l);
    rlevel[q[0]] = std::max(rlevel[q[0]], l);
    rlevel[q[1]] = std::max(rlevel[q[1]], l);
    nlev = std::max(nlev, l);
  }
  normal_levels = nlev;
  nlev++;   // the leaf rows: one more "level" behind the others (possibly empty)
  for (int k = 0; k < T; k++)
    if (leaf[alpha_index_times[4 * (size_t) k + 3]]) lvl[k] = nlev;
  level_offset.assign((size_t) nlev + 1, 0);
  for (int k = 0; k < T; k++) level_offset[lvl[k]]++;        // counts at [1..nlev]
  for (int l = 1; l <= nlev; l++) level_offset[l] += level_offset[l - 1];
  // level_offset[l] now = end of level l; shift to starts
  std::vector<int32_t> start((size_t) nlev + 1, 0);
  for (int l = 1; l <= nlev; l++) start[l] = level_offset[l - 1];
  rows_by_level.assign((size_t) T, MtpRow{0, 0, 0, 0});
  {
    std::vector<int32_t> cur(start);
    for (int k = 0; k < T; k++) {
      const int32_t *q = &alpha_index_times[4 * (size_t) k];
      rows_by_level[cur[lvl[k]]++] = MtpRow{q[0], q[1], q[2], q[3]};
    }
  }
  // level l (1-based) spans [level_offset[l-1], level_offset[l])
  // Rows of one level commute, so order them for the LDS: a wave instruction touches 64 consecutive
  // rows, served in lane groups of 32 (reads) / 16 (ds_add_f64).  Greedy: fill each group of 16 with
  // the rows whose operand and target moments fall on banks not yet used by a different moment of the
  // group (same moment = broadcast for reads, but serialised for the atomic adds).
  for (int l = 1; l <= nlev; l++) {
    const int b = level_offset[l - 1], e = level_offset[l];
    const int n = e - b;
    if (n <= 16) continue;
    std::vector<MtpRow> pool(rows_by_level.begin() + b, rows_by_level.begin() + e), out;
    std::vector<char> used((size_t) n, 0);
    out.reserve((size_t) n);
    int remaining = n, scan_from = 0;
    while (remaining > 0) {
      int rd0[32], rd1[32], rd3[32];          // moment occupying each read bank in the current 32-group (-1 free)
      for (int h = 0; h < 2 && remaining > 0; h++) {   // two 16-lane halves share the 32-lane read group
        if (h == 0)
          for (int q = 0; q < 32; q++) rd0[q] = rd1[q] = rd3[q] = -1;
        int at0[16], at1[16], at3[16];        // atomic-add banks of this 16-group
        for (int q = 0; q < 16; q++) at0[q] = at1[q] = at3[q] = -1;
        for (int slot = 0; slot < 16 && remaining > 0; slot++) {
          int best = -1, best_cost = 1 << 30;
          int seen = 0;
          for (int k = scan_from; k < n && seen < 256; k++) {   // bounded look-ahead keeps this O(n * 256)
            if (used[k]) continue;
            seen++;
            const MtpRow &r = pool[k];
            int cost = 0;
            cost += (rd0[r.a0 & 31] >= 0 && rd0[r.a0 & 31] != r.a0);
            cost += (rd1[r.a1 & 31] >= 0 && rd1[r.a1 & 31] != r.a1);
            if (l < nlev) {   // (leaf rows neither read D[a3] nor add into M[a3])
              cost += (rd3[r.a3 & 31] >= 0 && rd3[r.a3 & 31] != r.a3);
              cost += 2 * (at3[r.a3 & 15] >= 0);            // forward ds_add target
            }
            cost += (at0[r.a0 & 15] >= 0) + (at1[r.a1 & 15] >= 0);   // backward ds_add targets
            if (cost < best_cost) {
              best_cost = cost;
              best = k;
              if (cost == 0) break;
            }
          }
          const MtpRow &r = pool[best];
          used[best] = 1;
          remaining--;
          while (scan_from < n && used[scan_from]) scan_from++;
          rd0[r.a0 & 31] = r.a0;
          rd1[r.a1 & 31] = r.a1;
          rd3[r.a3 & 31] = r.a3;
          at0[r.a0 & 15] = r.a0;
          at1[r.a1 & 15] = r.a1;
          at3[r.a3 & 15] = r.a3;
          out.push_back(r);
        }
      }
    }
    std::copy(out.begin(), out.end(), rows_by_level.begin() + b);
  }
  // Pad every level to whole 64-row blocks with neutral rows (multiplicity 0, operands = target, a
  // different moment in every lane): the product kernels then run without bounds checks or lane masks.
  {
    std::vector<int> stored;   // (file numbering; the leaves have no LDS slot a padding row could touch)
    for (int m = 0; m < A; m++)
      if (!leaf[m]) stored.push_back(m);
    if (stored.empty()) stored.push_back(0);
    std::vector<MtpRow> padded;
    std::vector<int32_t> off((size_t) nlev + 1, 0);
    for (int l = 1; l <= nlev; l++) {
      const int b = level_offset[l - 1], e = level_offset[l];
      padded.insert(padded.end(), rows_by_level.begin() + b, rows_by_level.begin() + e);
      while ((int) padded.size() % 64 != 0) {
        const int t = stored[(size_t) ((int) padded.size() % 64) % stored.size()];
        padded.push_back(MtpRow{t, t, 0, t});
      }
      off[l] = (int32_t) padded.size();
    }
    rows_by_level.swap(padded);
    level_offset.swap(off);
  }
  for (int i = 0; i < S; i++)
    if (alpha_moment_mapping[i] < 0 || alpha_moment_mapping[i] >= A) {
      err = "alpha_moment_mapping refers to a moment outside alpha_moments_count";
      return MTP_ERR_TABLE;
    }
  // adjoint seeds: assignment, so the last scalar mapped to a moment wins (:217-218)
  {
    std::vector<int> last((size_t) A, -1);
    for (int i = 0; i < S; i++) last[alpha_moment_mapping[i]] = i;
    seed_idx.clear();
    seed_val.clear();
    for (int m = 0; m < A; m++)
      if (last[m] >= 0 && !leaf[m]) {
        seed_idx.push_back(m);
        seed_val.push_back(linear_coeffs[last[m]]);
      }
  }
  // LDS numbering of the moments.  The product passes read M[a0], M[a1], D[a3] (ds_read_b64: the two 32-lane halves
  // of a wave instruction are banked separately over 32 eight-byte banks) and add into M[a3], D[a0], D[a1]
  // (ds_add_f64, banked like ds_write_b64: four 16-lane groups over 16 eight-byte banks); distinct moments of one
  // group on one bank serialise (MI355X_MICROARCH.md, LDS table).  The row order is fixed by now: renumber the
  // moments -- basics among [0, B), products among [B, A), so the zero-fill and the k < B loops of the kernel keep
  // working -- by pairwise swaps that lower the modelled extra cycles.  Deterministic (fixed-seed LCG).
  // moment_perm[file index] = LDS index; every device table below is written in LDS numbering.
  // The leaves take the numbers behind the stored moments (only grade calls give them LDS slots).
  moment_perm.resize((size_t) A);
  std::vector<int> cls_members[3];   // 0 basics, 1 stored products, 2 leaves (file indices)
  {
    int nstored = B;
    for (int m = B; m < A; m++) nstored += !leaf[m];
    stored_moment_count = nstored;
    int next_stored = B, next_leaf = nstored;
    for (int m = 0; m < A; m++) {
      moment_perm[m] = m < B ? m : (leaf[m] ? next_leaf++ : next_stored++);
      cls_members[m < B ? 0 : (leaf[m] ? 2 : 1)].push_back(m);
    }
  }
  const int leaf_row0 = level_offset[(size_t) nlev - 1];   // first (padded) row of the leaf block
  int blocks_ahead = -1;   // head x tail blocks of the basic-moment pass, counted before they are built (search effort)
  if (A >= 2 && !rows_by_level.empty() && !std::getenv("MTP_NO_RENUMBER")) {
    struct Access {
      int nbk, w;
    };
    const int w_read[3] = {2, 2, 1}, w_add[3] = {1, 1, 1};   // a0, a1 are read in both passes, D[a3] in the reverse one
    auto pen = [](int n) { return n > 1 ? n - 1 : 0; };
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      return (uint32_t) (rng >> 33);
    };
    // extra cycles of one group of rows [r0, r0 + grp) under the current numbering
    // reads of one address broadcast (count distinct moments); adds to one address serialise (count rows)
    auto group_cost = [&](int r0, int grp, int nbk, const int w3[3], bool distinct) {
      int c = 0;
      for (int st = 0; st < (r0 >= leaf_row0 ? 2 : 3); st++) {   // (leaf rows: no access to their target)
        int seen[32], ns = 0;
        uint8_t h[32] = {0};
        for (int r = r0; r < r0 + grp; r++) {
          const MtpRow &row = rows_by_level[(size_t) r];
          const int m = moment_perm[st == 0 ? row.a0 : (st == 1 ? row.a1 : row.a3)];
          bool dup = false;
          if (distinct)
            for (int q = 0; q < ns; q++) dup |= seen[q] == m;
          if (!dup) {
            seen[ns++] = m;
            h[m % nbk]++;
          }
        }
        for (int b = 0; b < nbk; b++) c += w3[st] * pen(h[b]);
      }
      return c;
    };
    auto total_cost = [&]() {
      long long c = 0;
      for (size_t r0 = 0; r0 < rows_by_level.size(); r0 += 32) c += group_cost((int) r0, 32, 32, w_read, true);
      for (size_t r0 = 0; r0 < rows_by_level.size(); r0 += 16) c += group_cost((int) r0, 16, 16, w_add, false);
      return c;
    };
    const long long cost_before = total_cost();
    // Search effort.  Potentials whose product passes run row per lane (the narrow lane grids: up to level 16) pay every
    // modelled collision in ds_add_f64 cycles, the busiest pipe of their kernel: eight rounds of four times the proposals
    // (about 10 s at level 16, once per potential load) take the model from 403 to 335 extra cycles per atom and the force
    // call from 0.4322 to 0.4272 ms (same box, alternating runs).  The wide grids run the gather programs, which have
    // their own refinement below: two rounds.  MTP_BANK_ROUNDS / MTP_BANK_SCALE override (tests use two rounds).
    // (row per lane <=> at most 32 head x tail blocks in the basic-moment pass, mtp_pick_fwd_shape; the blocks are built
    // further down, from the numbering found here, so their number is counted ahead: per tail degree j the slots with
    // nu >= j in threes times the (b, c) pairs in threes, blocks without a basic left out)
    blocks_ahead = 0;
    {
      std::vector<uint8_t> have((size_t) 16 * 16 * 16 * 16, 0), slot_used((size_t) 16 * 16, 0);
      int Pmax = 0;
      for (int i = 0; i < B; i++) {
        const int32_t *q = &alpha_index_basic[4 * (size_t) i];
        if (q[0] > 15 || q[1] > 15 || q[2] > 15 || q[3] > 15 || q[1] + q[2] + q[3] > 15) continue;   // (refused further down)
        have[(((size_t) q[0] * 16 + q[1]) * 16 + q[2]) * 16 + q[3]] = 1;
        slot_used[(size_t) q[0] * 16 + (q[1] + q[2] + q[3])] = 1;
        Pmax = std::max(Pmax, q[1] + q[2] + q[3] + 1);
      }
      std::vector<std::pair<int, int>> slots;   // (mu, nu) in the order they are numbered below: nu ascending, then mu
      for (int nu = 0; nu < Pmax; nu++)
        for (int mu = 0; mu < 16; mu++)
          if (slot_used[(size_t) mu * 16 + nu]) slots.push_back({mu, nu});
      for (int j = 0; j < Pmax; j++) {
        std::vector<std::pair<int, int>> heads;   // (slot index, a)
        for (size_t sidx = 0; sidx < slots.size(); sidx++)
          if (slots[sidx].second >= j) heads.push_back({(int) sidx, slots[sidx].second - j});
        for (size_t h0 = 0; h0 < heads.size(); h0 += 3)
          for (int t0 = 0; t0 <= j; t0 += 3) {
            bool any = false;
            for (size_t h = h0; h < h0 + 3 && h < heads.size(); h++)
              for (int c = t0; c < t0 + 3 && c <= j; c++)
                any |= have[(((size_t) slots[(size_t) heads[h].first].first * 16 + heads[h].second) * 16 + (j - c)) * 16 + c] != 0;
            blocks_ahead += any;
          }
      }
    }
    const bool row_per_lane = blocks_ahead <= 32;
    int bank_rounds = row_per_lane ? 8 : 2, bank_scale = row_per_lane ? 4 : 1;
    if (const char *e = std::getenv("MTP_BANK_ROUNDS")) bank_rounds = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("MTP_BANK_SCALE")) bank_scale = std::max(1, std::atoi(e));
    for (int round = 0; round < bank_rounds; round++) {
      // ---- (a) renumber moments, rows fixed ----------------------------------------------------------
      std::vector<Access> acc;
      // moment (file index) -> (access, weight of the moment in it: 1 for reads, its row count for adds), ascending
      std::vector<std::vector<std::pair<int, int>>> occ((size_t) A);
      std::vector<uint8_t> hist;                       // [access][32]: load per bank
      auto add_accesses = [&](int grp, int nbk, const int w3[3], bool distinct) {
        const int ngroups = (int) rows_by_level.size() / grp;
        for (int g = 0; g < ngroups; g++)
          for (int st = 0; st < (grp * g >= leaf_row0 ? 2 : 3); st++) {
            const int id = (int) acc.size();
            acc.push_back({nbk, w3[st]});
            hist.resize(hist.size() + 32, 0);
            for (int r = grp * g; r < grp * g + grp; r++) {
              const MtpRow &row = rows_by_level[(size_t) r];
              const int m = st == 0 ? row.a0 : (st == 1 ? row.a1 : row.a3);
              auto &o = occ[(size_t) m];
              if (!o.empty() && o.back().first == id) {
                if (distinct) continue;
                o.back().second++;
              } else {
                o.push_back({id, 1});
              }
              hist[(size_t) id * 32 + (moment_perm[m] % nbk)]++;
            }
          }
      };
      add_accesses(32, 32, w_read, true);
      add_accesses(16, 16, w_add, false);
      auto mult_in = [&](int m, int id) {   // rows (adds) / 1 (reads) of moment m in access id, 0 if absent
        const auto &o = occ[(size_t) m];
        auto it = std::lower_bound(o.begin(), o.end(), std::make_pair(id, 0));
        return it != o.end() && it->first == id ? it->second : 0;
      };
      // cost change of swapping the numbers of m (at `from`) and other (at `to`), counted over m's accesses; accesses
      // holding both are counted once, from the smaller moment
      auto move_delta = [&](int m, int from, int to, int other) {
        int d = 0;
        for (const auto &e : occ[(size_t) m]) {
          const int id = e.first, k = e.second, k2 = mult_in(other, id);
          if (k2 > 0 && m > other) continue;
          const int f = from % acc[(size_t) id].nbk, t = to % acc[(size_t) id].nbk;
          if (f == t) continue;
          const uint8_t *h = &hist[(size_t) id * 32];
          d += acc[(size_t) id].w * (pen(h[f] - k + k2) + pen(h[t] + k - k2) - pen(h[f]) - pen(h[t]));
        }
        return d;
      };
      auto apply_move = [&](int m, int from, int to, int other) {
        for (const auto &e : occ[(size_t) m]) {
          const int id = e.first, k = e.second, k2 = mult_in(other, id);
          if (k2 > 0 && m > other) continue;
          const int f = from % acc[(size_t) id].nbk, t = to % acc[(size_t) id].nbk;
          if (f == t) continue;
          hist[(size_t) id * 32 + f] = (uint8_t) (hist[(size_t) id * 32 + f] - k + k2);
          hist[(size_t) id * 32 + t] = (uint8_t) (hist[(size_t) id * 32 + t] + k - k2);
        }
      };
      const long long trials = bank_scale * std::min<long long>(200ll * A, 300000ll);
      for (long long t = 0; t < trials; t++) {
        // three proposals in four start from a moment some row uses (weighted by use: the often-used moments are the ones
        // that collide), the rest from any moment
        int m1 = (int) (next() % (uint32_t) A);
        if ((next() & 3) != 0) {
          const MtpRow &pr = rows_by_level[next() % (uint32_t) rows_by_level.size()];
          const uint32_t st = next() % 3u;
          m1 = st == 0 ? pr.a0 : (st == 1 ? pr.a1 : pr.a3);
        }
        const std::vector<int> &cls = cls_members[m1 < B ? 0 : (leaf[m1] ? 2 : 1)];   // numbers swap inside a class only
        if (cls.size() < 2) continue;
        const int m2 = cls[next() % (uint32_t) cls.size()];
        if (m1 == m2) continue;
        const int p1 = moment_perm[m1], p2 = moment_perm[m2];
        if (move_delta(m1, p1, p2, m2) + move_delta(m2, p2, p1, m1) >= 0) continue;
        apply_move(m1, p1, p2, m2);
        apply_move(m2, p2, p1, m1);
        std::swap(moment_perm[m1], moment_perm[m2]);
      }
      // ---- (b) swap rows inside a level (they commute), numbering fixed -----------------------------------
      const int nlev2 = (int) level_offset.size() - 1;
      const long long rtrials = bank_scale * std::min<long long>(100ll * (long long) rows_by_level.size(), 250000ll);
      for (long long t = 0; t < rtrials; t++) {
        const int l = (int) (next() % (uint32_t) nlev2);
        const int b = level_offset[l], n = level_offset[l + 1] - b;
        if (n < 2) continue;
        // the first row comes from a 16-row group that has a collision (four tries), its partner from anywhere in the level
        int r1 = b + (int) (next() % (uint32_t) n);
        for (int tries = 0; tries < 4 && group_cost(r1 / 16 * 16, 16, 16, w_add, false) == 0; tries++)
          r1 = b + (int) (next() % (uint32_t) n);
        const int r2 = b + (int) (next() % (uint32_t) n);
        if (r1 / 16 == r2 / 16) continue;   // same add group (hence same read group): nothing changes
        auto local = [&]() {
          int c = group_cost(r1 / 16 * 16, 16, 16, w_add, false) + group_cost(r2 / 16 * 16, 16, 16, w_add, false);
          c += group_cost(r1 / 32 * 32, 32, 32, w_read, true);
          if (r1 / 32 != r2 / 32) c += group_cost(r2 / 32 * 32, 32, 32, w_read, true);
          return c;
        };
        const int c0 = local();
        std::swap(rows_by_level[(size_t) r1], rows_by_level[(size_t) r2]);
        if (local() >= c0) std::swap(rows_by_level[(size_t) r1], rows_by_level[(size_t) r2]);
      }
    }
    if (std::getenv("MTP_DEBUG_BANKS")) {
      long long cr = 0, ca = 0;
      for (size_t r0 = 0; r0 < rows_by_level.size(); r0 += 32) cr += group_cost((int) r0, 32, 32, w_read, true);
      for (size_t r0 = 0; r0 < rows_by_level.size(); r0 += 16) ca += group_cost((int) r0, 16, 16, w_add, false);
      // adds split by stream (a0, a1: reverse pass; a3: forward) and into same-address / same-bank shares
      long long same_addr[3] = {0, 0, 0}, same_bank[3] = {0, 0, 0};
      for (size_t r0 = 0; r0 < rows_by_level.size(); r0 += 16)
        for (int st = 0; st < ((int) r0 >= leaf_row0 ? 2 : 3); st++) {
          int cnt_addr[16][16], ids[16][16], nid[16] = {0};
          int h[16] = {0};
          for (size_t r = r0; r < r0 + 16; r++) {
            const MtpRow &row = rows_by_level[r];
            const int m = moment_perm[st == 0 ? row.a0 : (st == 1 ? row.a1 : row.a3)], b = m % 16;
            h[b]++;
            int q = 0;
            for (; q < nid[b]; q++)
              if (ids[b][q] == m) break;
            if (q == nid[b]) {
              ids[b][q] = m;
              cnt_addr[b][q] = 0;
              nid[b]++;
            }
            cnt_addr[b][q]++;
          }
          for (int b = 0; b < 16; b++) {
            int sa = 0;
            for (int q = 0; q < nid[b]; q++) sa += cnt_addr[b][q] - 1;
            same_addr[st] += sa;
            same_bank[st] += (h[b] > 1 ? h[b] - 1 : 0) - sa;
          }
        }
      std::fprintf(stderr, "mtp: LDS bank model of the product passes: %lld -> %lld extra cycles per atom (reads %lld, adds %lld); "
                           "%d head x tail blocks, %d rounds x %d\n", cost_before, total_cost(), cr, ca, blocks_ahead, bank_rounds, bank_scale);
      std::fprintf(stderr, "mtp:   adds, same address / other address on the bank: D[a0] %lld / %lld, D[a1] %lld / %lld, M[a3] %lld / %lld\n",
                   same_addr[0], same_bank[0], same_addr[1], same_bank[1], same_addr[2], same_bank[2]);
    }
  }
  for (MtpRow &row : rows_by_level) {
    row.a0 = moment_perm[row.a0];
    row.a1 = moment_perm[row.a1];
    row.a3 = moment_perm[row.a3];
  }
  for (int32_t &m : seed_idx) m = moment_perm[m];
  {   // constants of the leaf rows and the energy tables of the stored scalars
    std::vector<double> c_energy((size_t) A, 0.0), c_seed((size_t) A, 0.0);   // by LDS number
    e_map.clear();
    e_lin.clear();
    for (int i = 0; i < S; i++) {
      const int m = alpha_moment_mapping[i], ml = moment_perm[m];
      if (leaf[m]) {
        c_energy[(size_t) ml] += linear_coeffs[i];
        c_seed[(size_t) ml] = linear_coeffs[i];   // the last one wins (:217-218)
      } else {
        e_map.push_back(ml);
        e_lin.push_back(linear_coeffs[i]);
      }
    }
    const size_t nleaf_rows = rows_by_level.size() - (size_t) leaf_row0;
    leaf_cf.assign(nleaf_rows, 0.0);
    leaf_cb.assign(nleaf_rows, 0.0);
    for (size_t r = 0; r < nleaf_rows; r++) {
      const MtpRow &row = rows_by_level[(size_t) leaf_row0 + r];
      if (row.mult == 0) continue;   // padding
      leaf_cf[r] = c_energy[(size_t) row.a3] * row.mult;
      leaf_cb[r] = c_seed[(size_t) row.a3] * row.mult;
    }
  }
  mapping_lds.resize((size_t) S);
  for (int i = 0; i < S; i++) mapping_lds[i] = moment_perm[alpha_moment_mapping[i]];

  // radial slots = distinct (mu, nu) of the basics, numbered by tensor rank nu, then mu
  slot_of.assign((size_t) Mu * P, -1);
  for (int i = 0; i < B; i++) {
    const int32_t *q = &alpha_index_basic[4 * (size_t) i];
    if (q[0] > 15) {
      err = "radial function index above 15 is not supported";
      return MTP_ERR_LIMIT;
    }
    slot_of[(size_t) q[0] * P + (q[1] + q[2] + q[3])] = -2;   // used, not yet numbered
  }
  slot_count = 0;
  slot_coef_off.clear();
  slot_mu.clear();
  coef_total = 0;
  for (int nu = 0; nu < 14; nu++) {
    deg_first[nu] = slot_count;
    deg_coef[nu] = coef_total;
    if (nu >= P) continue;
    for (int mu = 0; mu < Mu; mu++)
      if (slot_of[(size_t) mu * P + nu] == -2) {
        slot_of[(size_t) mu * P + nu] = slot_count++;
        slot_coef_off.push_back(coef_total);
        slot_mu.push_back(mu);
        coef_total += nu == 0 ? 1 : 3 * (nu * (nu + 1) / 2);
      }
  }
  if (slot_count > 256) {
    err = "more than 256 distinct (mu, nu) radial slots";
    return MTP_ERR_LIMIT;
  }
  basic_pack.resize((size_t) B);
  for (int i = 0; i < B; i++) {
    const int32_t *q = &alpha_index_basic[4 * (size_t) i];
    const int s = slot_of[(size_t) q[0] * P + (q[1] + q[2] + q[3])];
    basic_pack[i] = s | (q[1] << 8) | (q[2] << 12) | (q[3] << 16) | (q[0] << 20);
  }
  // where each basic's adjoint goes in the derivative-polynomial coefficient blocks: basic (s; a, b, c)
  // contributes a*D to the d/dx coefficient of x^(a-1) y^b z^c, b*D and c*D alike (monomials of degree
  // nu-1 ordered a descending, then b descending: index j(j+1)/2 + c with j = b + c); rank 0: D itself
  if (coef_total > 65534) {
    err = "derivative-polynomial coefficient blocks above 65534 entries are not supported by this build";
    return MTP_ERR_LIMIT;
  }
  basic_tgt.assign((size_t) 2 * B, 0);
  {
    std::vector<int> hits((size_t) coef_total, 0);
    for (int i = 0; i < B; i++) {
      const int32_t *q = &alpha_index_basic[4 * (size_t) i];
      const int a = q[1], b = q[2], c = q[3], j = b + c, nu = a + j, C = nu * (nu + 1) / 2;
      const int base = slot_coef_off[(size_t) (basic_pack[i] & 255)];
      uint32_t tx = 0xffffu, ty = 0xffffu, tz = 0xffffu, fa = (uint32_t) a;
      if (nu == 0) {
        tx = (uint32_t) base;
        fa = 1;
      }
      if (a > 0) tx = (uint32_t) (base + j * (j + 1) / 2 + c);
      if (b > 0) ty = (uint32_t) (base + C + (j - 1) * j / 2 + c);
      if (c > 0) tz = (uint32_t) (base + 2 * C + (j - 1) * j / 2 + c - 1);
      for (uint32_t t : {tx, ty, tz})
        if (t != 0xffffu) hits[t]++;
      // stored at the basic's LDS number: the kernel walks D[k], tgt[k] with k in LDS numbering
      basic_tgt[2 * (size_t) moment_perm[i]] = (int32_t) (tx | (ty << 16));
      basic_tgt[2 * (size_t) moment_perm[i] + 1] = (int32_t) (tz | (fa << 16) | ((uint32_t) b << 20) | ((uint32_t) c << 24));
    }
    // head x tail blocks of the basic-moment pass
    {
      std::vector<int> basic_of((size_t) slot_count * 16 * 16 * 16, -1);   // (slot, a, b, c) -> k
      for (int i = 0; i < B; i++) {
        const int32_t *q = &alpha_index_basic[4 * (size_t) i];
        basic_of[(((size_t) (basic_pack[i] & 255) * 16 + q[1]) * 16 + q[2]) * 16 + q[3]] = i;
      }
      std::vector<int> slot_nu((size_t) slot_count, 0);
      for (int nu = 0; nu < P; nu++)
        for (int sidx = deg_first[nu]; sidx < deg_first[nu + 1]; sidx++) slot_nu[sidx] = nu;
      fwd_blocks.clear();
      fwd_block_count = 0;
      std::vector<int> covered((size_t) B, 0);
      for (int j = 0; j < P; j++) {
        std::vector<std::pair<int, int>> heads, tails;   // (slot, a), (b, c)
        for (int sidx = 0; sidx < slot_count; sidx++)
          if (slot_nu[sidx] >= j) heads.push_back({sidx, slot_nu[sidx] - j});
        for (int c = 0; c <= j; c++) tails.push_back({j - c, c});
        for (size_t h0 = 0; h0 < heads.size(); h0 += 3)
          for (size_t t0 = 0; t0 < tails.size(); t0 += 3) {
            int32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int16_t kk[10];
            for (int e = 0; e < 10; e++) kk[e] = -1;
            bool any = false;
            for (int h = 0; h < 3 && h0 + h < heads.size(); h++) {
              w[0] |= heads[h0 + h].first << (8 * h);
              w[1] |= heads[h0 + h].second << (4 * h);
            }
            for (int t = 0; t < 3 && t0 + t < tails.size(); t++) {
              w[1] |= tails[t0 + t].first << (12 + 4 * t);
              w[2] |= tails[t0 + t].second << (4 * t);
            }
            for (int h = 0; h < 3 && h0 + h < heads.size(); h++)
              for (int t = 0; t < 3 && t0 + t < tails.size(); t++) {
                const int k = basic_of[(((size_t) heads[h0 + h].first * 16 + heads[h0 + h].second) * 16 +
                                        tails[t0 + t].first) * 16 + tails[t0 + t].second];
                if (k >= 0) {
                  kk[3 * h + t] = (int16_t) moment_perm[k];   // LDS numbering
                  covered[k]++;
                  any = true;
                }
              }
            if (!any) continue;
            std::memcpy(&w[3], kk, sizeof(int16_t) * 10);
            fwd_blocks.insert(fwd_blocks.end(), w, w + 8);
            fwd_block_count++;
          }
      }
      for (int i = 0; i < B; i++)
        if (covered[i] != 1) {
          err = "internal: basic moment not covered exactly once by the head x tail blocks";
          return MTP_ERR_TABLE;
        }
      if (blocks_ahead >= 0 && blocks_ahead != fwd_block_count) {
        err = "internal: the head x tail blocks counted ahead of the renumbering differ from the blocks built";
        return MTP_ERR_TABLE;
      }
    }
    // packed basic descriptors in LDS numbering (mtp_cvec_kernel pairs them with dbasic[k] = D[k])
    basic_pack_lds.assign((size_t) B, 0);
    for (int i = 0; i < B; i++) basic_pack_lds[(size_t) moment_perm[i]] = basic_pack[i];
    coef_dense = 1;
    for (int t = 0; t < coef_total; t++) {
      if (hits[t] > 1) {
        err = "alpha_index_basic lists the same (mu, a, b, c) twice";
        return MTP_ERR_TABLE;
      }
      if (hits[t] == 0) coef_dense = 0;
    }
  }
  // ---- gather programs of the product passes ---------------------------------------------------------------------
  // The product passes as the kernel runs them (mtp_kernels.hip, gather_pass): per level a list of *chunks*; a chunk
  // holds up to cs operations acc += mult * X[o0] * Y[o1] that share one target, lane l of a group of 64 lanes runs one
  // chunk and ends it with ONE atomic add T[tgt] += acc.  Forward pass (pair_mtp.cpp:196-201): X = Y = T = moments,
  // chunks = the rows of a target.  Reverse pass (:221-233): X = adjoints, Y = moments, T = adjoints, chunks = the
  // terms D[a3] mult M[other] of one destination moment -- so the reverse pass needs two atomics per FOUR-TO-EIGHT
  // rows instead of two per row.  The chunk size cs in {1, 2, 4, 8} is chosen per level and pass by modelled LDS cycles
  // (2 per read, 15 per atomic add, padding included); chunks are dealt to lanes greedily so that the operands of one
  // wave instruction spread over the LDS banks (reads: 32 lanes over 32 eight-byte banks, adds: 16 lanes over 16).
  {
    const int nlev2 = normal_levels;   // (the leaf rows keep the row-per-lane form: no target to share)
    const int A_st = stored_moment_count;
    // Local search on top of the greedy deal (deterministic, fixed-seed LCG): swap the chunks of two lanes of the level,
    // or two operations inside a chunk (their sum does not depend on the order), whenever the modelled extra cycles --
    // per wave instruction and 32-lane half the largest number of distinct addresses on one read bank, per 16-lane
    // group the largest number of adds on one bank -- do not grow.  Padding operations (mult 0) are wildcards: they
    // end up on an address another lane of their half reads anyway (a broadcast).  Measured on the level-20 programs:
    // average bank load of the reads 2.1 -> 1.3.
    auto refine = [&](std::vector<MtpRow> &prog, size_t base, int G, int cs) {
      // (the kernel runs the programs in its 64-lane block grids only -- more than 32 head x tail blocks --, unless
      // it was built with -DMTP_GATHER_ALL; MTP_REFINE_PROGRAMS=0 / 1 overrides)
      bool wanted = fwd_block_count > 32;
      if (const char *e = std::getenv("MTP_REFINE_PROGRAMS")) wanted = std::atoi(e) != 0;
      if (G * 64 < 2 || !wanted) return;
      auto op = [&](int g, int u, int lane) -> MtpRow & { return prog[base + ((size_t) g * cs + u) * 64 + lane]; };
      int hot = 0;   // a lane on the most loaded bank of the last read_cost call
      auto read_cost = [&](int g, int u, int half, int which) {
        int first[32], extra[32][7], mx = 0;
        uint8_t n[32] = {0};
        for (int lane = 32 * half; lane < 32 * half + 32; lane++) {
          const MtpRow &o = op(g, u, lane);
          if (o.mult == 0) continue;
          const int a = which ? o.a1 : o.a0, b = a & 31;
          bool dup = false;
          if (n[b] > 0) {
            dup = first[b] == a;
            for (int k = 0; k + 1 < n[b] && k < 7 && !dup; k++) dup = extra[b][k] == a;
          }
          if (!dup) {
            if (n[b] == 0) first[b] = a;
            else if (n[b] - 1 < 7) extra[b][n[b] - 1] = a;
            n[b]++;
            if (n[b] > mx) {
              mx = n[b];
              hot = lane;
            }
          }
        }
        return mx > 1 ? mx - 1 : 0;
      };
      auto real_chunk = [&](int g, int lane) {
        for (int u = 0; u < cs; u++)
          if (op(g, u, lane).mult != 0) return true;
        return false;
      };
      auto add_cost = [&](int g, int q) {
        uint8_t h[16] = {0};
        int mx = 0;
        for (int lane = 16 * q; lane < 16 * q + 16; lane++)
          if (real_chunk(g, lane) && ++h[op(g, 0, lane).a3 & 15] > mx) {
            mx = h[op(g, 0, lane).a3 & 15];
            hot = lane;
          }
        return mx > 1 ? 2 * (mx - 1) : 0;
      };
      std::vector<int> rc((size_t) G * cs * 4), ac((size_t) G * 4);
      for (int g = 0; g < G; g++) {
        for (int u = 0; u < cs; u++)
          for (int hw = 0; hw < 4; hw++) rc[((size_t) g * cs + u) * 4 + hw] = read_cost(g, u, hw >> 1, hw & 1);
        for (int q = 0; q < 4; q++) ac[(size_t) g * 4 + q] = add_cost(g, q);
      }
      uint64_t rng = 0xD1B54A32D192ED03ull;
      auto next = [&]() {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t) (rng >> 33);
      };
      const long long trials = std::min<long long>(300ll * G * 64, 300000ll);
      for (long long t = 0; t < trials; t++) {
        // start from a read (or, one time in four, an add) that has a conflict: a lane on its most loaded bank moves
        const int g0 = (int) (next() % (uint32_t) G), u0 = (int) (next() % (uint32_t) cs), hw0 = (int) (next() & 3);
        const bool from_add = (next() & 3) == 0;
        if (from_add) {
          if (ac[(size_t) g0 * 4 + hw0] == 0) continue;
          (void) add_cost(g0, hw0);
        } else {
          if (rc[((size_t) g0 * cs + u0) * 4 + hw0] == 0) continue;
          (void) read_cost(g0, u0, hw0 >> 1, hw0 & 1);
        }
        const int lane0 = hot;
        if (!from_add && cs > 1 && (next() & 1) == 0) {   // two operations of one chunk
          const int g = g0, lane = lane0, half = lane >> 5;
          const int u1 = u0, u2 = (int) (next() % (uint32_t) cs);
          if (u1 == u2) continue;
          int before = 0, after = 0;
          for (int w = 0; w < 2; w++) before += rc[((size_t) g * cs + u1) * 4 + 2 * half + w] + rc[((size_t) g * cs + u2) * 4 + 2 * half + w];
          std::swap(op(g, u1, lane), op(g, u2, lane));
          int nc[4];
          for (int w = 0; w < 2; w++) {
            nc[w] = read_cost(g, u1, half, w);
            nc[2 + w] = read_cost(g, u2, half, w);
            after += nc[w] + nc[2 + w];
          }
          if (after > before) {
            std::swap(op(g, u1, lane), op(g, u2, lane));
            continue;
          }
          for (int w = 0; w < 2; w++) {
            rc[((size_t) g * cs + u1) * 4 + 2 * half + w] = nc[w];
            rc[((size_t) g * cs + u2) * 4 + 2 * half + w] = nc[2 + w];
          }
        } else {   // the chunks of two lanes
          const int g1 = g0, l1 = lane0;
          const int g2 = (int) (next() % (uint32_t) G), l2 = (int) (next() & 63);
          const int h1 = l1 >> 5, h2 = l2 >> 5, q1 = l1 >> 4, q2 = l2 >> 4;
          if (g1 == g2 && q1 == q2) continue;   // same add group, hence same read half: nothing changes
          const bool same_half = g1 == g2 && h1 == h2;
          int before = ac[(size_t) g1 * 4 + q1] + ac[(size_t) g2 * 4 + q2], after = 0;
          for (int u = 0; u < cs; u++)
            for (int w = 0; w < 2; w++) {
              before += rc[((size_t) g1 * cs + u) * 4 + 2 * h1 + w];
              if (!same_half) before += rc[((size_t) g2 * cs + u) * 4 + 2 * h2 + w];
            }
          for (int u = 0; u < cs; u++) std::swap(op(g1, u, l1), op(g2, u, l2));
          int n1[16], n2[16];
          for (int u = 0; u < cs; u++)
            for (int w = 0; w < 2; w++) {
              n1[2 * u + w] = read_cost(g1, u, h1, w);
              after += n1[2 * u + w];
              if (!same_half) {
                n2[2 * u + w] = read_cost(g2, u, h2, w);
                after += n2[2 * u + w];
              }
            }
          const int a1 = add_cost(g1, q1), a2 = add_cost(g2, q2);
          after += a1 + a2;
          if (after > before) {
            for (int u = 0; u < cs; u++) std::swap(op(g1, u, l1), op(g2, u, l2));
            continue;
          }
          for (int u = 0; u < cs; u++)
            for (int w = 0; w < 2; w++) {
              rc[((size_t) g1 * cs + u) * 4 + 2 * h1 + w] = n1[2 * u + w];
              if (!same_half) rc[((size_t) g2 * cs + u) * 4 + 2 * h2 + w] = n2[2 * u + w];
            }
          ac[(size_t) g1 * 4 + q1] = a1;
          ac[(size_t) g2 * 4 + q2] = a2;
        }
      }
      // wildcards: read what another lane of the half reads (broadcast); padding chunks add 0.0 on a free add bank
      for (int g = 0; g < G; g++) {
        for (int u = 0; u < cs; u++)
          for (int half = 0; half < 2; half++) {
            int a0 = 0, a1 = 0;
            for (int lane = 32 * half; lane < 32 * half + 32; lane++)
              if (op(g, u, lane).mult != 0) {
                a0 = op(g, u, lane).a0;
                a1 = op(g, u, lane).a1;
                break;
              }
            for (int lane = 32 * half; lane < 32 * half + 32; lane++)
              if (op(g, u, lane).mult == 0) {
                op(g, u, lane).a0 = a0;
                op(g, u, lane).a1 = a1;
              }
          }
        for (int q = 0; q < 4; q++) {
          bool busy[16] = {false};
          for (int lane = 16 * q; lane < 16 * q + 16; lane++)
            if (real_chunk(g, lane)) busy[op(g, 0, lane).a3 & 15] = true;
          for (int lane = 16 * q; lane < 16 * q + 16; lane++) {
            if (real_chunk(g, lane)) continue;
            int t = op(g, 0, lane).a3;
            for (int m = 0; m < A_st; m++)
              if (!busy[m & 15]) {
                t = m;
                break;
              }
            busy[t & 15] = true;
            for (int u = 0; u < cs; u++) op(g, u, lane).a3 = t;
          }
        }
      }
    };
    auto build = [&](bool reverse, std::vector<MtpRow> &prog, std::vector<int32_t> &seg) {
      prog.clear();
      seg.clear();
      for (int li = 0; li < nlev2; li++) {
        const int l = reverse ? nlev2 - 1 - li : li;
        // operations of the level, keyed by target
        std::vector<std::vector<MtpRow>> by_tgt((size_t) A);
        for (int r = level_offset[l]; r < level_offset[l + 1]; r++) {
          const MtpRow &row = rows_by_level[(size_t) r];
          if (row.mult == 0) continue;   // neutral padding rows of the old layout
          if (!reverse) {
            by_tgt[(size_t) row.a3].push_back(row);
          } else if (row.a0 == row.a1 && 2 * row.mult <= 32767 && 2 * row.mult >= -32768) {
            by_tgt[(size_t) row.a0].push_back(MtpRow{row.a3, row.a0, 2 * row.mult, row.a0});   // both terms in one
          } else {
            by_tgt[(size_t) row.a1].push_back(MtpRow{row.a3, row.a0, row.mult, row.a1});       // D[a1] += D[a3] mult M[a0]
            by_tgt[(size_t) row.a0].push_back(MtpRow{row.a3, row.a1, row.mult, row.a0});       // D[a0] += D[a3] mult M[a1]
          }
        }
        // chunk size by modelled LDS cycles
        int best_cs = 1;
        long long best_cost = -1;
        for (int cs : {1, 2, 4, 8}) {
          long long chunks = 0;
          for (const auto &v : by_tgt) chunks += ((long long) v.size() + cs - 1) / cs;
          const long long groups = (chunks + 63) / 64, cost = groups * cs * 6 + groups * 15;
          if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_cs = cs;
          }
        }
        const int cs = best_cs;
        struct Chunk {
          int tgt;
          MtpRow op[8];
        };
        std::vector<Chunk> chunks;
        for (int t = 0; t < A; t++) {
          const auto &v = by_tgt[(size_t) t];
          for (size_t b = 0; b < v.size(); b += (size_t) cs) {
            Chunk c;
            c.tgt = t;
            for (int u = 0; u < cs; u++) c.op[u] = b + u < v.size() ? v[b + u] : MtpRow{t, t, 0, t};
            chunks.push_back(c);
          }
        }
        // longest-first would not matter (all chunks are cs long after padding); keep file order, pad to whole groups
        const int ngroups = (int) ((chunks.size() + 63) / 64);
        const int first_block = (int) (prog.size() / 64);
        std::vector<char> used(chunks.size(), 0);
        size_t scan_from = 0;
        for (int g = 0; g < ngroups; g++) {
          int occx[8][2][32], occy[8][2][32], occt[4][16];
          for (auto &a : occx)
            for (auto &b : a)
              for (int &c : b) c = -1;
          for (auto &a : occy)
            for (auto &b : a)
              for (int &c : b) c = -1;
          for (auto &a : occt)
            for (int &c : a) c = -1;
          std::vector<MtpRow> blk((size_t) 64 * cs);
          for (int lane = 0; lane < 64; lane++) {
            const int half = lane >> 5, q16 = lane >> 4;
            int best = -1, best_rot = 0, bcost = 1 << 30, seen = 0;
            for (size_t k = scan_from; k < chunks.size() && seen < 128; k++) {
              if (used[k]) continue;
              seen++;
              const Chunk &c = chunks[k];
              const int tcost = 3 * (occt[q16][c.tgt & 15] >= 0);
              for (int rot = 0; rot < cs; rot++) {
                int cost = tcost;
                for (int u = 0; u < cs; u++) {
                  const MtpRow &o = c.op[(u + rot) % cs];
                  const int bx = occx[u][half][o.a0 & 31], by = occy[u][half][o.a1 & 31];
                  cost += (bx >= 0 && bx != o.a0) + (by >= 0 && by != o.a1);
                }
                if (cost < bcost) {
                  bcost = cost;
                  best = (int) k;
                  best_rot = rot;
                }
                if (cost == 0) break;
              }
              if (bcost == 0) break;
            }
            Chunk c;
            if (best >= 0) {
              c = chunks[(size_t) best];
              used[(size_t) best] = 1;
              while (scan_from < chunks.size() && used[scan_from]) scan_from++;
            } else {   // padding chunk: adds 0.0 to a moment whose add bank is still free in this 16-lane group
              int t = 0;
              for (int m = 0; m < A_st; m++)
                if (occt[q16][m & 15] < 0) {
                  t = m;
                  break;
                }
              c.tgt = t;
              for (int u = 0; u < cs; u++) c.op[u] = MtpRow{t, t, 0, t};
              best_rot = 0;
            }
            occt[q16][c.tgt & 15] = c.tgt;
            for (int u = 0; u < cs; u++) {
              MtpRow o = c.op[(u + best_rot) % cs];
              o.a3 = c.tgt;
              occx[u][half][o.a0 & 31] = o.a0;
              occy[u][half][o.a1 & 31] = o.a1;
              blk[(size_t) u * 64 + lane] = o;
            }
          }
          prog.insert(prog.end(), blk.begin(), blk.end());
        }
        refine(prog, (size_t) first_block * 64, ngroups, cs);
        seg.push_back(first_block);
        seg.push_back(ngroups);
        seg.push_back(cs);
        seg.push_back(0);
      }
    };
    build(false, prog_fwd, seg_fwd);
    build(true, prog_bwd, seg_bwd);
  }
  return MTP_OK;
}
