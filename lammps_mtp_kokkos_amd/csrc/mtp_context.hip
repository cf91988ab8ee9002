// Context (one GPU) and the extern "C" entry points of libmtp_mi355x (include/mtp_mi355x.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mtp_mi355x.h"
#include "mtp_device.hpp"
#include "mtp_potential.hpp"

namespace {

struct HipFail {
  hipError_t e;
  const char *what;
};
#define HIP_CHECK(call)                                   \
  do {                                                    \
    hipError_t _e = (call);                               \
    if (_e != hipSuccess) throw HipFail{_e, #call};       \
  } while (0)

template <class T> struct DevBuf {
  T *ptr = nullptr;
  size_t cap = 0;
  void reserve(size_t n)
  {
    if (n <= cap) return;
    if (ptr) (void) hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), std::max<size_t>(n, 1) * sizeof(T)));
    cap = n;
  }
  void upload(const T *src, size_t n, hipStream_t st)
  {
    reserve(n);
    if (n) HIP_CHECK(hipMemcpyAsync(ptr, src, n * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void upload(const std::vector<T> &v, hipStream_t st) { upload(v.data(), v.size(), st); }
  ~DevBuf()
  {
    if (ptr) (void) hipFree(ptr);
  }
};

void copy_err(const std::string &s, char *err, int errlen)
{
  if (err && errlen > 0) std::snprintf(err, (size_t) errlen, "%s", s.c_str());
}

}   // namespace

struct mtp_context {
  const mtp_potential *pot = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  int variant = MTP_VARIANT_AUTO;
  int num_cus = 256;
  // LDS table blob: the pieces a launch plan may leave in HBM / L2 are its tail -- [core | adjoint scatter targets |
  // basic descriptors (candidate-vector kernel) | packed times rows]; a plan copies one of these four prefixes
  int blob_bytes_core = 0, blob_bytes_tgt = 0, blob_bytes_norows = 0, blob_bytes_rows = 0;
  bool xcd_map = true;   // MTP_XCD_MAP=0 (tuning override) turns the XCD-aware atom map off
  // potential tables
  DevBuf<double> d_species;
  DevBuf<MtpRow8> d_rows, d_prog_fwd, d_prog_bwd;
  DevBuf<unsigned char> d_blob;
  DevBuf<int32_t> d_seed_idx, d_map, d_map_all, d_tgt;
  DevBuf<double> d_seed_val, d_lin, d_leaf_cf, d_leaf_cb;
  // neighbour list
  DevBuf<int> d_ilist, d_first, d_neigh;
  DevBuf<int> d_nb_scratch, d_nb_info;   // device neighbour-list build
  hipStream_t list_stream = nullptr;     // stream the context-owned list was last written on
  DevBuf<unsigned char> d_nb_tmp;
  DevBuf<double> d_nb_xs;
  const int *ilist = nullptr, *first = nullptr, *neigh = nullptr;   // active (owned or caller's)
  int inum = 0, nall = 0, max_numneigh = 0;
  bool have_list = false;
  // host-path staging
  DevBuf<double> d_x, d_f, d_eatom, d_vatom, d_grades, d_coeff;
  DevBuf<int> d_type;
  std::vector<double> h_tmp;
  // workspaces
  DevBuf<double> d_ev_slots, d_ev, d_maxg;
  DevBuf<long long> d_fq;      // deterministic mode: fixed-point force accumulators, kept zeroed between calls
  bool deterministic = false;
  DevBuf<int> d_err;
  DevBuf<unsigned long long> d_stamps;
  // launch geometry
  struct Layout {   // per-atom LDS image, offsets in doubles (MtpDevParams: dg_mode, pow_row, dg_off, off_*)
    int mode = 0, pow_row = 0, dg_off = 0, fp_row = 0, off_m = 0, off_d = 0, off_coef = 0, off_nb = 0, m_doubles = 0;
  };
  struct LaunchPlan {
    Layout layout;
    int wpb = 1, grid = 1, wave_doubles = 0, tab_rows = 0, g_doubles = 0, m_doubles = 0, ov_doubles = 0;
    bool rebuild = false;
    int wps = 2;
    bool rows_lds = false, tgt_lds = true;
    int blob_bytes = 0;   // the blob prefix this plan copies
    size_t lds_bytes = 0;
  } lp[3];   // [0] force calls (wavefront per atom), [1] candidate-vector kernel of grade calls, [2] the fused kernel's
             // grade instantiation (its image also holds the leaf moments' values)
  DevBuf<double> d_cvec, d_ainv_pad, d_ainv_tiled, d_dbasic;
  int cpad = 0, dpad = 0;
  // device-resident outputs of mtp_compute_resident (the /kk styles' DualViews): which of them the last call filled
  bool res_valid = false;
  int res_eflag = 0, res_vflag = 0, res_grade = 0;
  DevBuf<double> d_res_tot;   // [8 + 1 + C]: ev[8] | max grade | coeff_ders[C], zeroed by one launch per step
  // timing
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;

  MtpDevParams base{};

  void plan();
};

// Choose the workgroup shape from the LDS budget (160 KiB / CU): every workgroup carries one copy
// of the table blob plus one private region per wavefront.  Grade calls need extra table rows
// (r^-nu, Q_ri) and scratch, so they get their own plan.
void mtp_context::plan()
{
  const mtp_potential &p = *pot;
  const int A = p.alpha_moment_count, P = p.max_alpha_index_basic;
  int KL = 16, KB = 1;
  (void) mtp_pick_shape(p.alpha_index_basic_count, &KL, &KB);
  const int cap = std::max(64, (max_numneigh + 31) / 32 * 32);
  const size_t LDS = 160 * 1024;
  const int nt = 32;
  // One plan for a given table-blob prefix; returns the wavefronts per CU it reaches (0: does not fit).
  auto plan_one = [&](int which, size_t blob, LaunchPlan &L) -> int {
    L.tab_rows = 2 * p.slot_count + 3 * P;
    L.g_doubles = 0;
    // leaf moments have no LDS slot in force calls; grade calls keep their values (candidate vector), not their adjoints
    const int d_doubles = p.stored_moment_count, Am = which == 2 ? A : p.stored_moment_count;
    const size_t ints = (size_t) 2 * nt + cap;
    const int grows = p.slot_count * MTP_PITCH;   // g rows; the dg rows take as much again
    const int trows = 2 * grows;
    const size_t tail = 5 * (size_t) nt * 8 + ints * 4;   // neighbour arrays behind the tables
    // Three layouts of the per-atom LDS image (mtp_kernels.hip, WaveLds), sizes in doubles:
    //   keep     [g rows | dg rows | overlay]; coordinate-power rows and moments / adjoints share the overlay
    //            (never live together); the derivative-polynomial coefficients later take the moments' place
    const int m_keep = std::max(std::max(Am, p.coef_total), 16);
    Layout keep;
    keep.mode = 0;
    keep.pow_row = 2 * p.slot_count;
    keep.dg_off = grows;
    keep.off_m = trows;
    keep.off_d = trows + m_keep;
    keep.off_coef = trows;
    keep.m_doubles = m_keep;
    keep.off_nb = trows + std::max(3 * P * MTP_PITCH, d_doubles + m_keep);
    //   nodg     [g rows | overlay] (Mu <= 4): no dg rows; Mu rows f'_mu (written ahead of the force phase from radial
    //            derivatives the tile build parked in registers) sit behind the coefficient blocks: 11.7 instead of
    //            15.7 KB per atom at level 16, and neither the 16 dg rows nor a second evaluation of the radial functions
    const int Mu_ = p.radial_func_count;
    const bool nodg_ok = Mu_ <= 4;
    Layout lean = keep;
    lean.mode = 1;
    lean.pow_row = p.slot_count;
    lean.dg_off = 0;
    lean.off_m = grows;
    lean.off_d = grows + m_keep;
    lean.off_coef = grows;
    lean.fp_row = (grows + p.coef_total + MTP_PITCH - 1) / MTP_PITCH;
    lean.off_nb = std::max(grows + std::max(3 * P * MTP_PITCH, d_doubles + m_keep), (lean.fp_row + Mu_) * MTP_PITCH);
    //   rebuild  everything overlays everything (many moments): first table build = g rows and power rows only;
    //            moments and adjoints then take the front of the region; ahead of the force phase the g and dg rows
    //            are built again (coefficient blocks behind them, D[0, B) in front).  One more pass over the
    //            neighbours' radial functions buys LDS: level 20 goes from 37 to 24 KB per atom.
    const int m_reb = std::max(Am, 16);
    Layout reb;
    reb.mode = 2;
    reb.pow_row = p.slot_count;
    reb.dg_off = grows;
    reb.off_d = 0;
    reb.off_m = d_doubles;
    reb.off_coef = trows;
    reb.m_doubles = m_reb;
    reb.off_nb = std::max(std::max(grows + 3 * P * MTP_PITCH, trows + p.coef_total), d_doubles + m_reb);
    const bool reb_ok = trows >= p.alpha_index_basic_count;
    //   rebuild without dg rows (Mu <= 4): the second build writes the g rows only, the f' rows follow the coefficients
    Layout rebn = reb;
    rebn.mode = 3;
    rebn.dg_off = 0;
    rebn.off_coef = grows;
    rebn.fp_row = (grows + p.coef_total + MTP_PITCH - 1) / MTP_PITCH;
    rebn.off_nb = std::max(std::max(grows + 3 * P * MTP_PITCH, (rebn.fp_row + Mu_) * MTP_PITCH), d_doubles + m_reb);
    const bool rebn_ok = nodg_ok && grows >= p.alpha_index_basic_count;
    auto bytes_of = [&](const Layout &y) { return ((size_t) y.off_nb * 8 + tail + 15) / 16 * 16; };
    // registers: 8 wavefronts per CU (2 per SIMD at <= 256 VGPRs) in workgroups of up to 8, or -- for the table
    // shapes that have the 168-VGPR build -- 12 (3 per SIMD).  Measured on MI355X: a workgroup is only admitted when
    // every SIMD it lands on has room, and workgroups of 5..7 wavefronts load the SIMDs unevenly (two 6-wavefront
    // workgroups never shared a CU at 3 per SIMD); so the 3-per-SIMD plan uses one workgroup of 12 or three of 4.
    int wave_cap = 32;
    if (const char *e = std::getenv("MTP_MAX_WAVES")) wave_cap = std::max(1, std::min(16, std::atoi(e)));
    auto waves2 = [&](int w, size_t wbytes) {   // 2-per-SIMD build, w wavefronts per workgroup
      const size_t blk = blob + w * wbytes;
      if (w > 8 || blk > LDS) return 0;
      return std::min<int>(std::min(wave_cap, 8), (int) (LDS / blk) * w);
    };
    auto best2 = [&](size_t wbytes) {
      int v = 0;
      for (int w = 1; w <= 8; w++) v = std::max(v, waves2(w, wbytes));
      return v;
    };
    auto shape3 = [&](size_t wbytes) {   // 3-per-SIMD build: wavefronts per workgroup that reach 12 per CU, or 0
      if (wave_cap < 12) return 0;
      if (blob + 12 * wbytes <= LDS) return 12;
      if (3 * (blob + 4 * wbytes) <= LDS) return 4;
      return 0;
    };
    const bool fine = variant == MTP_VARIANT_SMALL || (variant == MTP_VARIANT_AUTO && inum < num_cus * 16);
    // candidates in order of preference at equal occupancy: nodg (least work), keep, then the rebuilding ones
    std::vector<const Layout *> cands;
    if (nodg_ok) cands.push_back(&lean);
    cands.push_back(&keep);
    if (rebn_ok) cands.push_back(&rebn);
    if (reb_ok) cands.push_back(&reb);
    if (const char *e = std::getenv("MTP_LAYOUT")) {   // tuning override (benchmarks, tests): keep | nodg | rebuild | rebuild-nodg
      const std::string v(e);
      if (v == "keep") cands = {&keep};
      else if ((v == "nodg" || v == "lean") && nodg_ok) cands = {&lean};
      else if (v == "rebuild" && reb_ok) cands = {&reb};
      else if (v == "rebuild-nodg" && rebn_ok) cands = {&rebn};
      else if (v == "rebuild" && rebn_ok) cands = {&rebn};
    }
    // the 3-per-SIMD build pays when there are atoms enough to fill twelve wavefronts per CU
    bool has3 = mtp_wave_kernel_has_wps3(p.fwd_block_count, P);
    if (const char *e = std::getenv("MTP_WPS")) has3 = has3 && std::atoi(e) == 3;   // tuning override: 2 = never, 3 = whenever it fits
    else has3 = has3 && !fine;
    // (the grade instantiation spilled 52 dwords at 168 VGPRs and was 2.6 % slower there until the force totals were
    // reduced per tile: 35 now, and 12 wavefronts per CU make the grade call 8 % faster; MTP_GRADE_WPS3=0 turns it off)
    if (which == 2)
      if (const char *e = std::getenv("MTP_GRADE_WPS3")) has3 = has3 && std::atoi(e) != 0;
    const Layout *pick = nullptr;
    int wps = 2, w3 = 0, pick_waves = 0;
    for (const Layout *y : cands) {
      int v = best2(bytes_of(*y)), vw3 = 0;
      // (the 3-per-SIMD build carries the dg-free force phase only)
      if (has3 && (y->mode & 1) && (vw3 = shape3(bytes_of(*y))) > 0) v = 12;
      if (v > pick_waves) {
        pick = y;
        pick_waves = v;
        wps = vw3 > 0 ? 3 : 2;
        w3 = vw3;
      }
    }
    if (!pick) return 0;
    L.layout = *pick;
    L.rebuild = (pick->mode & 2) != 0;
    L.wps = wps;
    L.m_doubles = pick->m_doubles;
    L.ov_doubles = pick->off_nb;
    const size_t wb = bytes_of(*pick);
    int best_w = 0, best = 0;
    if (wps == 3) {
      best_w = w3;
      best = 12;
    } else {
      // few atoms (or the "small" variant): the finest spread that still reaches the best occupancy;
      // many atoms: as many wavefronts per workgroup as possible (fewer copies of the table blob)
      for (int w = 1; w <= 8; w++) {
        int v = waves2(w, wb);
        if (v > best || (v == best && v > 0 && !fine)) {
          best = v;
          best_w = w;
        }
      }
      if (best == 0) return 0;
      if (fine) {
        // ... every atom its own wavefront, in the WIDEST workgroups that still leave no CU without one (fewer copies of
        // the table blob, fewer workgroups to dispatch: 2,048 atoms in 256 workgroups of 8 wavefronts run 3 % faster than
        // in 1,024 of 2); narrower only when the atoms would not cover the CUs
        int pick_w = 0;
        for (int w = 1; w <= 8; w++)
          if (waves2(w, wb) > 0 && (long long) num_cus * waves2(w, wb) >= inum && (inum + w - 1) / w >= num_cus) pick_w = w;
        if (pick_w == 0)
          for (int w = 1; w < best_w && pick_w == 0; w++)
            if ((long long) num_cus * waves2(w, wb) >= inum) pick_w = w;
        if (pick_w > 0) best_w = pick_w;
      }
      if (const char *e = std::getenv("MTP_WPB")) {   // tuning override (benchmarks only)
        int v = std::atoi(e);
        if (v >= 1 && v <= 8 && waves2(v, wb) > 0) best_w = v;
      }
      best = std::max(1, waves2(best_w, wb));
    }
    const int blocks_per_cu = std::max(1, best / best_w);
    // packed times rows in LDS when the chosen shape still fits with them (or when they are tiny)
    auto fits = [&](int bytes) { return (size_t) blocks_per_cu * ((size_t) bytes + best_w * wb) <= LDS; };
    const bool forced = std::getenv("MTP_BLOB_PREFIX") != nullptr;   // (then exactly the planned prefix is copied)
    L.rows_lds = forced ? (int) blob == blob_bytes_rows : fits(blob_bytes_rows);
    if (const char *e = std::getenv("MTP_ROWS_LDS")) L.rows_lds = L.rows_lds && std::atoi(e) != 0;   // tuning override
    L.tgt_lds = forced ? (int) blob >= blob_bytes_tgt : (L.rows_lds || fits(blob_bytes_tgt));
    if (forced) L.blob_bytes = L.rows_lds ? blob_bytes_rows : std::min((int) blob, blob_bytes_norows);
    else L.blob_bytes = L.rows_lds ? blob_bytes_rows : (fits(blob_bytes_norows) ? blob_bytes_norows : (L.tgt_lds ? blob_bytes_tgt : blob_bytes_core));
    L.wpb = best_w;
    L.wave_doubles = (int) (wb / 8);
    L.lds_bytes = (size_t) L.blob_bytes + wb * best_w;
    const int need = (inum + best_w - 1) / best_w;
    L.grid = std::max(1, std::min(need, num_cus * blocks_per_cu));
    return best;
  };
  // The shape is planned against each prefix of the table blob, longest first: a shorter prefix (tables read from
  // HBM / L2 instead) is taken only when it buys wavefronts per CU (level 20: 8 instead of 7 with the core prefix).
  for (int which : {0, 2}) {   // [0] fused force kernel, [2] its grade instantiation
    int best_waves = 0;
    std::vector<int> prefixes = {blob_bytes_rows, blob_bytes_norows, blob_bytes_tgt, blob_bytes_core};
    if (const char *e = std::getenv("MTP_BLOB_PREFIX")) {   // tuning / test override: plan against one prefix only
      const std::string v(e);
      if (v == "core") prefixes = {blob_bytes_core};
      else if (v == "tgt") prefixes = {blob_bytes_tgt};
      else if (v == "norows") prefixes = {blob_bytes_norows};
      else if (v == "rows") prefixes = {blob_bytes_rows};
    }
    for (int bytes : prefixes) {
      LaunchPlan L;
      const int v = plan_one(which, (size_t) bytes, L);
      if (v > best_waves) {
        best_waves = v;
        lp[which] = L;
      }
    }
    if (best_waves == 0) throw HipFail{hipErrorInvalidValue, "potential + neighbour list exceed one CU's LDS"};
  }
  {   // [1] candidate-vector kernel of grade calls: small table (r^-nu, Q_ri, powers), 8 wavefronts per workgroup
    LaunchPlan &L = lp[1];
    const int Mu = p.radial_func_count, R = p.radial_basis_size, Sp = p.species_count;
    const size_t dbl = (size_t) KL * KB + (size_t) (4 * P + R) * (nt + 2) + 4 * (size_t) nt + (size_t) Mu * nt + (size_t) Sp * Mu * R;
    const size_t ints = (size_t) nt + cap;
    const size_t wb = (dbl * 8 + ints * 4 + 15) / 16 * 16;
    const size_t blob1 = (size_t) blob_bytes_norows;   // (this kernel reads the basic descriptors)
    int w = 8;
    while (w > 1 && blob1 + w * wb > LDS) w--;
    L.wpb = w;
    L.wave_doubles = (int) (wb / 8);
    L.lds_bytes = blob1 + wb * w;
    const int blocks_per_cu = std::max<int>(1, std::min<int>(8 / w, (int) (LDS / L.lds_bytes)));
    L.grid = std::max(1, std::min((inum + w - 1) / w, num_cus * blocks_per_cu));
    L.tab_rows = 4 * P + R;
    L.m_doubles = KL * KB;
    L.g_doubles = 0;
  }
  base.NT = nt;
  base.cj_cap = cap;
  base.d_doubles = p.stored_moment_count;
}

extern "C" {

int mtp_potential_load(const char *path, int want_selection, mtp_potential **out, char *err, int errlen)
{
  if (!path || !out) return MTP_ERR_ARG;
  *out = nullptr;
  mtp_potential *p = new (std::nothrow) mtp_potential();
  if (!p) return MTP_ERR_ARG;
  std::string msg;
  int rc = mtp_parse_file(path, want_selection != 0, *p, msg);
  if (rc != MTP_OK) {
    copy_err(msg, err, errlen);
    delete p;
    return rc;
  }
  *out = p;
  return MTP_OK;
}

void mtp_potential_free(mtp_potential *pot) { delete pot; }

int mtp_potential_get_info(const mtp_potential *p, mtp_potential_info *info)
{
  if (!p || !info) return MTP_ERR_ARG;
  info->species_count = p->species_count;
  info->radial_basis_size = p->radial_basis_size;
  info->radial_func_count = p->radial_func_count;
  info->alpha_moment_count = p->alpha_moment_count;
  info->alpha_index_basic_count = p->alpha_index_basic_count;
  info->alpha_index_times_count = p->alpha_index_times_count;
  info->alpha_scalar_count = p->alpha_scalar_count;
  info->max_alpha_index_basic = p->max_alpha_index_basic;
  info->coeff_count = p->coeff_count;
  info->has_selection = p->has_selection;
  info->configuration_mode = p->configuration_mode;
  info->product_levels = p->normal_levels;
  info->scaling = p->scaling;
  info->min_cutoff = p->min_cutoff;
  info->max_cutoff = p->max_cutoff;
  return MTP_OK;
}

int mtp_potential_get_tables(const mtp_potential *p, int32_t *aib, int32_t *ait, int32_t *map, double *rc,
                             double *sc, double *mc, double *inv)
{
  if (!p) return MTP_ERR_ARG;
  auto cp = [](auto *dst, const auto &v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cp(aib, p->alpha_index_basic);
  cp(ait, p->alpha_index_times);
  cp(map, p->alpha_moment_mapping);
  cp(rc, p->radial_basis_coeffs);
  cp(sc, p->species_coeffs);
  cp(mc, p->linear_coeffs);
  if (inv) {
    if (!p->has_selection) return MTP_ERR_STATE;
    cp(inv, p->inverse_active_set);
  }
  return MTP_OK;
}

int mtp_cfg_grade(const mtp_potential *p, const double *c, double *grade)
{
  if (!p || !c || !grade) return MTP_ERR_ARG;
  if (!p->has_selection) return MTP_ERR_STATE;
  const int C = p->coeff_count;
  double mx = 0.0;
  for (int i = 0; i < C; i++) {
    const double *row = &p->inverse_active_set[(size_t) i * C];
    double g = 0.0;
    for (int j = 0; j < C; j++) g += c[j] * row[j];
    mx = std::max(mx, g < 0 ? -g : g);
  }
  *grade = mx;
  return MTP_OK;
}

int mtp_context_create(const mtp_potential *pot, int device_id, mtp_context **out, char *err, int errlen)
{
  if (!pot || !out) return MTP_ERR_ARG;
  *out = nullptr;
  mtp_context *c = nullptr;
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      copy_err("no HIP device is visible: libmtp_mi355x has no CPU fallback", err, errlen);
      return MTP_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) {
      copy_err("device id out of range", err, errlen);
      return MTP_ERR_ARG;
    }
    HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      copy_err(std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only", err,
               errlen);
      return MTP_ERR_DEVICE;
    }
    c = new mtp_context();
    c->pot = pot;
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    if (const char *e = std::getenv("MTP_XCD_MAP")) c->xcd_map = std::atoi(e) != 0;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    hipStream_t st = c->stream;
    c->d_species.upload(pot->species_coeffs, st);
    // packed rows (8 B each): moment ids in 16 bits, multiplicity in a signed 16 bits
    if (pot->alpha_moment_count > 8191) {   // (packed rows hold 16-bit byte offsets; 8192 moments are 128 KB of LDS anyway)
      copy_err("alpha_moments_count above 8191 is not supported by this build", err, errlen);
      delete c;
      return MTP_ERR_LIMIT;
    }
    std::vector<MtpRow8> rows8(pot->rows_by_level.size());
    for (size_t k = 0; k < rows8.size(); k++) {
      const MtpRow &r = pot->rows_by_level[k];
      if (r.mult > 32767 || r.mult < -32768) {
        copy_err("alpha_index_times multiplicity outside 16 bits is not supported by this build", err, errlen);
        delete c;
        return MTP_ERR_LIMIT;
      }
      rows8[k].lo = (uint32_t) (8 * r.a0) | ((uint32_t) (8 * r.a1) << 16);
      rows8[k].hi = (uint32_t) (8 * r.a3) | (((uint32_t) r.mult & 0xffffu) << 16);
    }
    c->d_rows.upload(rows8.data(), rows8.size(), st);
    auto pack_prog = [&](const std::vector<MtpRow> &prog, DevBuf<MtpRow8> &buf) {
      std::vector<MtpRow8> p8(prog.size());
      for (size_t k = 0; k < prog.size(); k++) {
        const MtpRow &r = prog[k];
        p8[k].lo = (uint32_t) (8 * r.a0) | ((uint32_t) (8 * r.a1) << 16);
        p8[k].hi = (uint32_t) (8 * r.a3) | (((uint32_t) r.mult & 0xffffu) << 16);
      }
      buf.upload(p8.data(), p8.size(), st);
    };
    pack_prog(pot->prog_fwd, c->d_prog_fwd);
    pack_prog(pot->prog_bwd, c->d_prog_bwd);
    // table blob copied into LDS by every workgroup
    MtpDevParams &bb = c->base;
    // The packed rows are the LAST piece of the blob: a launch plan copies them into LDS (blob_bytes_rows) or leaves
    // them in HBM/L2 (blob_bytes_norows) -- measured at level 16: rows in LDS are 1.2 % faster when they fit beside
    // the wavefronts' private regions anyway, and < 1 % slower otherwise (row reads do not depend on data).
    std::vector<unsigned char> blob;
    auto put = [&](const void *src, size_t bytes) {
      size_t off = (blob.size() + 15) / 16 * 16;
      blob.resize(off + bytes, 0);
      if (bytes) std::memcpy(blob.data() + off, src, bytes);
      return (int) off;
    };
    bb.off_level = put(pot->level_offset.data(), pot->level_offset.size() * sizeof(int32_t));
    bb.off_seg_fwd = put(pot->seg_fwd.data(), pot->seg_fwd.size() * sizeof(int32_t));
    bb.off_seg_bwd = put(pot->seg_bwd.data(), pot->seg_bwd.size() * sizeof(int32_t));
    std::vector<int32_t> slot_pad((size_t) pot->radial_func_count * MTP_PSTRIDE, -1);
    for (int mu = 0; mu < pot->radial_func_count; mu++)
      for (int nu = 0; nu < pot->max_alpha_index_basic; nu++)
        slot_pad[(size_t) mu * MTP_PSTRIDE + nu] = pot->slot_of[(size_t) mu * pot->max_alpha_index_basic + nu];
    bb.off_slot = put(slot_pad.data(), slot_pad.size() * sizeof(int32_t));
    bb.off_radial = put(pot->radial_basis_coeffs.data(), pot->radial_basis_coeffs.size() * sizeof(double));
    // scalar-side tables (24 B per basis function): LDS when they are small, HBM/L2 otherwise
    bb.scalars_in_lds = (pot->e_map.size() + pot->seed_idx.size()) * 12 <= 4096;
    if (const char *e = std::getenv("MTP_SCALARS_LDS")) bb.scalars_in_lds = std::atoi(e) != 0;   // tuning override
    if (bb.scalars_in_lds) {
      bb.off_seed_idx = put(pot->seed_idx.data(), pot->seed_idx.size() * sizeof(int32_t));
      bb.off_seed_val = put(pot->seed_val.data(), pot->seed_val.size() * sizeof(double));
      bb.off_map = put(pot->e_map.data(), pot->e_map.size() * sizeof(int32_t));
      bb.off_lin = put(pot->e_lin.data(), pot->e_lin.size() * sizeof(double));
    } else {
      bb.off_seed_idx = bb.off_seed_val = bb.off_map = bb.off_lin = 0;
    }
    c->d_seed_idx.upload(pot->seed_idx, st);
    c->d_seed_val.upload(pot->seed_val, st);
    c->d_map.upload(pot->e_map, st);
    c->d_lin.upload(pot->e_lin, st);
    c->d_map_all.upload(pot->mapping_lds, st);
    c->d_leaf_cf.upload(pot->leaf_cf, st);
    c->d_leaf_cb.upload(pot->leaf_cb, st);
    bb.g_seed_idx = c->d_seed_idx.ptr;
    bb.g_seed_val = c->d_seed_val.ptr;
    bb.g_map = c->d_map.ptr;
    bb.g_lin = c->d_lin.ptr;
    bb.g_map_all = c->d_map_all.ptr;
    bb.leaf_cf = c->d_leaf_cf.ptr;
    bb.leaf_cb = c->d_leaf_cb.ptr;
    bb.off_smu = put(pot->slot_mu.data(), pot->slot_mu.size() * sizeof(int32_t));
    bb.off_fwd = put(pot->fwd_blocks.data(), pot->fwd_blocks.size() * sizeof(int32_t));
    bb.nfb = pot->fwd_block_count;
    blob.resize((blob.size() + 15) / 16 * 16, 0);
    c->blob_bytes_core = (int) blob.size();
    bb.off_coef = put(pot->basic_tgt.data(), pot->basic_tgt.size() * sizeof(int32_t));
    c->d_tgt.upload(pot->basic_tgt, st);
    bb.g_tgt = c->d_tgt.ptr;
    blob.resize((blob.size() + 15) / 16 * 16, 0);
    c->blob_bytes_tgt = (int) blob.size();
    bb.off_pack = put(pot->basic_pack_lds.data(), pot->basic_pack_lds.size() * sizeof(int32_t));
    blob.resize((blob.size() + 15) / 16 * 16, 0);
    c->blob_bytes_norows = (int) blob.size();
    bb.off_rows = put(rows8.data(), rows8.size() * sizeof(MtpRow8));
    bb.off_leaf_cf = put(pot->leaf_cf.data(), pot->leaf_cf.size() * sizeof(double));
    bb.off_leaf_cb = pot->leaf_cb == pot->leaf_cf ? bb.off_leaf_cf : put(pot->leaf_cb.data(), pot->leaf_cb.size() * sizeof(double));
    blob.resize((blob.size() + 15) / 16 * 16, 0);
    c->blob_bytes_rows = (int) blob.size();
    bb.blob_bytes = c->blob_bytes_norows;   // plan() decides per launch plan
    bb.rows_in_lds = 0;
    c->d_blob.upload(blob.data(), blob.size(), st);
    if (pot->has_selection) {   // inverse active set zero padded to a multiple of 16 for the MFMA grade kernel
      const int C = pot->coeff_count;
      c->cpad = (C + 15) / 16 * 16;
      std::vector<double> pad((size_t) c->cpad * c->cpad, 0.0);
      for (int r = 0; r < C; r++)
        std::memcpy(&pad[(size_t) r * c->cpad], &pot->inverse_active_set[(size_t) r * C], (size_t) C * sizeof(double));
      c->d_ainv_pad.upload(pad, st);
      // the same matrix in MFMA operand order for the LDS-staged grade kernel: [tile][k-step][lane],
      // lane (j = l & 15, k = l >> 4) <- Ainv[16 tile + j][4 kstep + k]
      std::vector<double> tiled((size_t) c->cpad * c->cpad, 0.0);
      const int ks_n = c->cpad / 4;
      for (int t = 0; t < c->cpad / 16; t++)
        for (int ks = 0; ks < ks_n; ks++)
          for (int l = 0; l < 64; l++)
            tiled[((size_t) t * ks_n + ks) * 64 + l] = pad[(size_t) (16 * t + (l & 15)) * c->cpad + 4 * ks + (l >> 4)];
      c->d_ainv_tiled.upload(tiled, st);
      HIP_CHECK(hipStreamSynchronize(st));
    }
    c->d_ev_slots.reserve((size_t) MTP_EV_SLOTS * 8);
    HIP_CHECK(hipMemsetAsync(c->d_ev_slots.ptr, 0, (size_t) MTP_EV_SLOTS * 8 * sizeof(double), st));
    c->d_ev.reserve(8);
    c->d_maxg.reserve(1);
    c->d_err.reserve(1);
    HIP_CHECK(hipMemsetAsync(c->d_err.ptr, 0, sizeof(int), st));
    c->d_stamps.reserve(16);
    HIP_CHECK(hipMemsetAsync(c->d_stamps.ptr, 0, 16 * sizeof(unsigned long long), st));
    HIP_CHECK(hipStreamSynchronize(st));

    MtpDevParams &b = c->base;
    b.Sp = pot->species_count;
    b.R = pot->radial_basis_size;
    b.Mu = pot->radial_func_count;
    b.P = pot->max_alpha_index_basic;
    b.A = pot->alpha_moment_count;
    b.B = pot->alpha_index_basic_count;
    b.T = pot->alpha_index_times_count;
    b.S = pot->alpha_scalar_count;
    b.C = pot->coeff_count;
    b.nslot = pot->slot_count;
    b.coef_total = pot->coef_total;
    b.coef_dense = pot->coef_dense;
    for (int d = 0; d <= MTP_PSTRIDE; d++) {
      b.deg_first[d] = pot->deg_first[d];
      b.deg_coef[d] = pot->deg_coef[d];
    }
    b.nlevels = pot->normal_levels;   // (the level table has one more entry: the leaf rows)
    b.nseed = (int) pot->seed_idx.size();
    b.Ad = pot->stored_moment_count;
    b.Am = b.Ad;                      // per launch: the grade instantiation keeps the leaves' values too
    b.Se = (int) pot->e_map.size();
    b.rmin = pot->min_cutoff;
    b.rmax = pot->max_cutoff;
    b.scaling = pot->scaling;
    b.cutsq = pot->max_cutoff * pot->max_cutoff;   // pair_mtp.cpp:449,456
    b.inv_span = 1.0 / (pot->max_cutoff - pot->min_cutoff);
    b.inv_rmax = 1.0 / pot->max_cutoff;
    b.blob = c->d_blob.ptr;
    b.rows = c->d_rows.ptr;
    b.prog_fwd = c->d_prog_fwd.ptr;
    b.prog_bwd = c->d_prog_bwd.ptr;
    b.species_coeffs = c->d_species.ptr;
    b.inv_mu = 1.0f / (float) pot->radial_func_count;
    b.ev_slots = c->d_ev_slots.ptr;
    b.err_flag = c->d_err.ptr;
    b.stamps = c->d_stamps.ptr;
    int kl_ = 0, kb_ = 0;
    if (mtp_pick_fwd_shape(pot->fwd_block_count, &kl_, &kb_) != 0) {
      copy_err("more than 256 head x tail blocks of basic moments are not supported by this build", err, errlen);
      delete c;
      return MTP_ERR_LIMIT;
    }
    if (mtp_pick_shape(pot->alpha_index_basic_count, &kl_, &kb_) != 0) {
      copy_err("alpha_index_basic_count above 640 is not supported by this build", err, errlen);
      delete c;
      return MTP_ERR_LIMIT;
    }
  } catch (const HipFail &f) {
    copy_err(std::string(f.what) + ": " + hipGetErrorString(f.e), err, errlen);
    delete c;
    return MTP_ERR_DEVICE;
  }
  *out = c;
  return MTP_OK;
}

void mtp_context_destroy(mtp_context *c)
{
  if (!c) return;
  (void) hipSetDevice(c->device);
  if (c->ev0) (void) hipEventDestroy(c->ev0);
  if (c->ev1) (void) hipEventDestroy(c->ev1);
  if (c->stream) {
    (void) hipStreamSynchronize(c->stream);
    (void) hipStreamDestroy(c->stream);
  }
  delete c;
}

const char *mtp_last_error(const mtp_context *c) { return c ? c->last_error.c_str() : "null context"; }

int mtp_context_set_variant(mtp_context *c, int variant)
{
  if (!c || variant < MTP_VARIANT_AUTO || variant > MTP_VARIANT_SMALL) return MTP_ERR_ARG;
  c->variant = variant;
  if (c->have_list) {
    try {
      c->plan();
    } catch (const HipFail &f) {
      c->last_error = f.what;
      return MTP_ERR_LIMIT;
    }
  }
  return MTP_OK;
}

static int finish_list(mtp_context *c, int inum, int nall, int max_numneigh)
{
  c->inum = inum;
  c->nall = nall;
  c->max_numneigh = max_numneigh;
  c->have_list = true;
  try {
    c->plan();
    if (c->pot->has_selection && inum > 0) {   // candidate vectors, zero padded rows of cpad doubles
      const size_t n = (size_t) inum * c->cpad;
      if (n > c->d_cvec.cap) {
        c->d_cvec.reserve(n);
        HIP_CHECK(hipMemset(c->d_cvec.ptr, 0, n * sizeof(double)));
      }
      int kl_ = 16, kb_ = 1;
      (void) mtp_pick_shape(c->pot->alpha_index_basic_count, &kl_, &kb_);
      c->dpad = kl_ * kb_;
      c->d_dbasic.reserve((size_t) inum * c->dpad);
    }
  } catch (const HipFail &f) {
    c->last_error = f.what;
    c->have_list = false;
    return f.e == hipErrorInvalidValue ? MTP_ERR_LIMIT : MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_set_neighbors_csr(mtp_context *c, int inum, const int *ilist, const int *first, const int *neigh,
                          int nall)
{
  if (!c || inum < 0 || nall < inum || (inum > 0 && (!ilist || !first))) return MTP_ERR_ARG;
  try {
    HIP_CHECK(hipSetDevice(c->device));
    int mx = 0;
    for (int ii = 0; ii < inum; ii++) mx = std::max(mx, first[ii + 1] - first[ii]);
    const int total = inum > 0 ? first[inum] : 0;
    if (total > 0 && !neigh) return MTP_ERR_ARG;
    c->d_ilist.upload(ilist, (size_t) inum, c->stream);
    std::vector<int> z(1, 0);
    c->d_first.upload(inum > 0 ? first : z.data(), (size_t) inum + 1, c->stream);
    c->d_neigh.upload(neigh, (size_t) total, c->stream);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->ilist = c->d_ilist.ptr;
    c->first = c->d_first.ptr;
    c->neigh = c->d_neigh.ptr;
    return finish_list(c, inum, nall, mx);
  } catch (const HipFail &f) {
    c->last_error = std::string(f.what) + ": " + hipGetErrorString(f.e);
    return MTP_ERR_DEVICE;
  }
}

int mtp_set_neighbors(mtp_context *c, int inum, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, int nall)
{
  if (!c || inum < 0 || (inum > 0 && (!ilist || !numneigh || !firstneigh))) return MTP_ERR_ARG;
  std::vector<int> first((size_t) inum + 1, 0);
  long long total = 0;
  for (int ii = 0; ii < inum; ii++) total += numneigh[ilist[ii]];
  if (total > 2147483647LL) {
    c->last_error = "neighbour list has more than 2^31-1 entries on this rank";
    return MTP_ERR_LIMIT;
  }
  for (int ii = 0; ii < inum; ii++) first[ii + 1] = first[ii] + numneigh[ilist[ii]];
  std::vector<int> neigh((size_t) first[inum]);
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    std::copy(firstneigh[i], firstneigh[i] + numneigh[i], neigh.begin() + first[ii]);
  }
  return mtp_set_neighbors_csr(c, inum, ilist, first.data(), neigh.data(), nall);
}

int mtp_set_neighbors_device(mtp_context *c, int inum, const int *d_ilist, const int *d_first,
                             const int *d_neigh, int nall, int max_numneigh)
{
  if (!c || inum < 0 || nall < inum || max_numneigh < 0 || (inum > 0 && (!d_ilist || !d_first))) return MTP_ERR_ARG;
  (void) hipSetDevice(c->device);
  c->ilist = d_ilist;
  c->first = d_first;
  c->neigh = d_neigh;
  return finish_list(c, inum, nall, max_numneigh);
}

int mtp_build_neighbors_device(mtp_context *c, void *stream, const double *d_x, int inum, int nall,
                               double list_cutoff, const double lo[3], const double hi[3],
                               const int **d_first_out, const int **d_neigh_out, long long *total_out,
                               int *max_numneigh_out)
{
  if (!c || inum < 0 || nall < inum || !lo || !hi || !(list_cutoff > 0.0) || (nall > 0 && !d_x)) return MTP_ERR_ARG;
  if (hipSetDevice(c->device) != hipSuccess) {
    c->last_error = "hipSetDevice failed";
    return MTP_ERR_DEVICE;
  }
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  int n3[3];
  long long ncell = 1;
  for (int a = 0; a < 3; a++) {
    if (!(hi[a] >= lo[a])) return MTP_ERR_ARG;
    n3[a] = std::max(1, (int) std::ceil((hi[a] - lo[a]) / list_cutoff));
    ncell *= n3[a];
  }
  if (ncell > (1ll << 26)) {
    c->last_error = "neighbour build: more than 2^26 cells (box much larger than the atoms it holds?)";
    return MTP_ERR_LIMIT;
  }
  try {
    const size_t scan_n = (size_t) std::max<long long>(ncell, inum) + 1;
    const size_t cub_bytes = mtp_neighbor_scan_bytes((int) scan_n, nall);
    c->d_nb_tmp.reserve(std::max<size_t>(cub_bytes, 16));
    c->d_nb_scratch.reserve((size_t) 4 * nall + 2 * (size_t) ncell + 2 + (size_t) inum + 1);
    c->d_nb_info.reserve(2);
    c->d_nb_xs.reserve((size_t) 3 * std::max(nall, 1));
    c->d_ilist.reserve((size_t) std::max(inum, 1));
    c->d_first.reserve((size_t) inum + 1);
    HIP_CHECK(mtp_launch_neighbor_build(d_x, inum, nall, list_cutoff, lo, n3, c->d_nb_scratch.ptr, c->d_nb_xs.ptr,
                                        c->d_nb_tmp.ptr, c->d_nb_tmp.cap, c->d_ilist.ptr, c->d_first.ptr, nullptr,
                                        c->d_nb_info.ptr, st));
    int info[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(info, c->d_nb_info.ptr, sizeof(info), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (info[0] < 0 || (long long) inum * info[1] > 0x7fffffffll) {
      c->last_error = "neighbour list has more than 2^31-1 entries on this rank";
      return MTP_ERR_LIMIT;
    }
    c->d_neigh.reserve((size_t) std::max(info[0], 1));
    HIP_CHECK(mtp_launch_neighbor_build(d_x, inum, nall, list_cutoff, lo, n3, c->d_nb_scratch.ptr, c->d_nb_xs.ptr,
                                        c->d_nb_tmp.ptr, c->d_nb_tmp.cap, c->d_ilist.ptr, c->d_first.ptr, c->d_neigh.ptr,
                                        c->d_nb_info.ptr, st));
    c->ilist = c->d_ilist.ptr;
    c->first = c->d_first.ptr;
    c->neigh = c->d_neigh.ptr;
    c->list_stream = st;
    if (d_first_out) *d_first_out = c->d_first.ptr;
    if (d_neigh_out) *d_neigh_out = c->d_neigh.ptr;
    if (total_out) *total_out = info[0];
    if (max_numneigh_out) *max_numneigh_out = info[1];
    return finish_list(c, inum, nall, info[1]);
  } catch (const HipFail &f) {
    c->last_error = std::string(f.what) + ": " + hipGetErrorString(f.e);
    return MTP_ERR_DEVICE;
  }
}

int mtp_set_neighbors_device_2d(mtp_context *c, void *stream, int inum, const int *d_ilist, const int *d_numneigh,
                                const int *d_neighbors, long long stride_i, long long stride_jj, int max_neighs, int nall)
{
  if (!c || inum < 0 || nall < inum || max_neighs < 0 || (inum > 0 && (!d_ilist || !d_numneigh)) || stride_i < 0 || stride_jj < 0)
    return MTP_ERR_ARG;
  if (inum > 0 && max_neighs > 0 && (!d_neighbors || (stride_i != 1 && stride_jj != 1))) {
    c->last_error = "mtp_set_neighbors_device_2d: one of the two strides must be 1 (LayoutLeft or LayoutRight view)";
    return MTP_ERR_ARG;
  }
  if (hipSetDevice(c->device) != hipSuccess) {
    c->last_error = "hipSetDevice failed";
    return MTP_ERR_DEVICE;
  }
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  try {
    const size_t cub_bytes = mtp_neighbor_scan_bytes(inum + 1, 1);
    c->d_nb_tmp.reserve(std::max<size_t>(cub_bytes, 16));
    c->d_nb_scratch.reserve((size_t) inum + 1);
    c->d_nb_info.reserve(4);
    c->d_first.reserve((size_t) inum + 1);
    HIP_CHECK(mtp_launch_list_from_2d(inum, d_ilist, d_numneigh, d_neighbors, stride_i, stride_jj, max_neighs,
                                      c->d_nb_scratch.ptr, c->d_nb_tmp.ptr, c->d_nb_tmp.cap, c->d_first.ptr, nullptr,
                                      c->d_nb_info.ptr, st));
    int info[3] = {0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(info, c->d_nb_info.ptr, sizeof(info), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));   // one read-back per re-neighbouring sizes the entry array and the LDS plan
    if (info[2]) {
      c->last_error = "mtp_set_neighbors_device_2d: a d_numneigh entry is negative or exceeds the view's second extent";
      return MTP_ERR_ARG;
    }
    if (info[0] < 0) {
      c->last_error = "neighbour list has more than 2^31-1 entries on this rank";
      return MTP_ERR_LIMIT;
    }
    c->d_neigh.reserve((size_t) std::max(info[0], 1));
    HIP_CHECK(mtp_launch_list_from_2d(inum, d_ilist, d_numneigh, d_neighbors, stride_i, stride_jj, max_neighs,
                                      c->d_nb_scratch.ptr, c->d_nb_tmp.ptr, c->d_nb_tmp.cap, c->d_first.ptr, c->d_neigh.ptr,
                                      c->d_nb_info.ptr, st));
    c->ilist = d_ilist;   // the caller's (KOKKOS') d_ilist stays in use: it must remain valid until the next list
    c->first = c->d_first.ptr;
    c->neigh = c->d_neigh.ptr;
    c->list_stream = st;
    return finish_list(c, inum, nall, info[1]);
  } catch (const HipFail &f) {
    c->last_error = std::string(f.what) + ": " + hipGetErrorString(f.e);
    return MTP_ERR_DEVICE;
  }
}

// ---- device-resident step (include/mtp_mi355x.h, "device-resident step") ------------------------------------------
int mtp_compute_resident(mtp_context *c, void *stream, const double *d_x, const int *d_type, double *d_f, int eflag, int vflag,
                         int grade_flag)
{
  if (!c || !d_x || !d_type || !d_f) return MTP_ERR_ARG;
  if (!c->have_list) {
    c->last_error = "mtp_compute before mtp_set_neighbors";
    return MTP_ERR_STATE;
  }
  if (hipSetDevice(c->device) != hipSuccess) {
    c->last_error = "hipSetDevice failed";
    return MTP_ERR_DEVICE;
  }
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  const size_t nall = (size_t) c->nall, C = (size_t) c->pot->coeff_count;
  const bool want_ea = (eflag & MTP_ENERGY_ATOM) != 0, want_va = (vflag & MTP_VIRIAL_ATOM) != 0;
  const bool cfg = c->pot->configuration_mode != 0;
  c->res_valid = false;
  try {
    // totals: one allocation, padded to an even count, zeroed by ONE launch on the caller's stream
    const size_t ntot = (8 + 1 + C + 1) / 2 * 2;
    c->d_res_tot.reserve(ntot);
    HIP_CHECK(mtp_launch_zero(c->d_res_tot.ptr, ntot, st));
    if (want_ea) {   // eatom[i] is assigned for i in ilist; every other row reads 0 (LAMMPS zeroes eatom in ev_setup)
      c->d_eatom.reserve(nall + (nall & 1));
      HIP_CHECK(mtp_launch_zero(c->d_eatom.ptr, nall + (nall & 1), st));
    }
    if (want_va) {
      c->d_vatom.reserve(6 * nall);
      HIP_CHECK(mtp_launch_zero(c->d_vatom.ptr, 6 * nall, st));
    }
    if (grade_flag && !cfg) {   // grown, never shrunk; rows outside ilist keep their last value (pair_mtp_extrapolation.cpp:91-94)
      if (c->d_grades.cap < nall) {
        c->d_grades.reserve(nall + (nall & 1));
        HIP_CHECK(mtp_launch_zero(c->d_grades.ptr, nall + (nall & 1), st));
      }
    }
  } catch (const HipFail &f) {
    c->last_error = std::string(f.what) + ": " + hipGetErrorString(f.e);
    return MTP_ERR_DEVICE;
  }
  double *tot = c->d_res_tot.ptr;
  const int rc = mtp_compute_device(c, reinterpret_cast<void *>(st), d_x, d_type, eflag, vflag, grade_flag, d_f,
                                    want_ea ? c->d_eatom.ptr : nullptr, want_va ? c->d_vatom.ptr : nullptr, tot,
                                    grade_flag && !cfg ? c->d_grades.ptr : nullptr, grade_flag ? tot + 8 : nullptr,
                                    grade_flag && cfg ? tot + 9 : nullptr);
  if (rc != MTP_OK) return rc;
  c->res_valid = true;
  c->res_eflag = eflag;
  c->res_vflag = vflag;
  c->res_grade = grade_flag ? 1 : 0;
  return MTP_OK;
}

int mtp_resident_totals(mtp_context *c, void *stream, double *ev7, double *max_grade, double *coeff_ders)
{
  if (!c) return MTP_ERR_ARG;
  if (!c->res_valid) {
    c->last_error = "mtp_resident_totals before mtp_compute_resident";
    return MTP_ERR_STATE;
  }
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  const size_t C = (size_t) c->pot->coeff_count, n = 9 + (coeff_ders && c->res_grade ? C : 0);
  c->h_tmp.resize(std::max<size_t>(c->h_tmp.size(), 9 + C));
  if (hipSetDevice(c->device) != hipSuccess ||
      hipMemcpyAsync(c->h_tmp.data(), c->d_res_tot.ptr, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) {
    c->last_error = "resident totals: copy failed";
    return MTP_ERR_DEVICE;
  }
  const int rc = mtp_synchronize(c, reinterpret_cast<void *>(st));   // the one wait of the step; reports the atom-type error
  if (rc != MTP_OK) return rc;
  if (ev7)
    for (int q = 0; q < 7; q++) ev7[q] = c->h_tmp[q];
  if (max_grade) *max_grade = c->res_grade ? c->h_tmp[8] : 0.0;
  if (coeff_ders && c->res_grade)
    for (size_t q = 0; q < C; q++) coeff_ders[q] = c->h_tmp[9 + q];
  return MTP_OK;
}

static int resident_array(mtp_context *c, int what, const double **ptr, int *ncol)
{
  if (!c->res_valid) {
    c->last_error = "per-atom outputs asked for before mtp_compute_resident";
    return MTP_ERR_STATE;
  }
  switch (what) {
    case MTP_PERATOM_EATOM:
      if (!(c->res_eflag & MTP_ENERGY_ATOM)) break;
      *ptr = c->d_eatom.ptr;
      *ncol = 1;
      return MTP_OK;
    case MTP_PERATOM_VATOM:
      if (!(c->res_vflag & MTP_VIRIAL_ATOM)) break;
      *ptr = c->d_vatom.ptr;
      *ncol = 6;
      return MTP_OK;
    case MTP_PERATOM_GRADES:
      if (!c->d_grades.ptr || c->pot->configuration_mode) break;
      *ptr = c->d_grades.ptr;
      *ncol = 1;
      return MTP_OK;
    default:
      return MTP_ERR_ARG;
  }
  c->last_error = "this per-atom output was not produced by the last mtp_compute_resident call";
  return MTP_ERR_STATE;
}

int mtp_resident_peratom_device(mtp_context *c, int what, const double **d_ptr, int *ncol)
{
  if (!c || !d_ptr) return MTP_ERR_ARG;
  int nc = 0;
  const int rc = resident_array(c, what, d_ptr, &nc);
  if (rc == MTP_OK && ncol) *ncol = nc;
  return rc;
}

int mtp_resident_peratom_host(mtp_context *c, void *stream, int what, double *host)
{
  if (!c || !host) return MTP_ERR_ARG;
  const double *src = nullptr;
  int nc = 0;
  const int rc = resident_array(c, what, &src, &nc);
  if (rc != MTP_OK) return rc;
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  if (hipSetDevice(c->device) != hipSuccess ||
      hipMemcpyAsync(host, src, (size_t) c->nall * nc * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    c->last_error = "resident per-atom output: copy failed";
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_copy_to_host(mtp_context *c, void *stream, void *host, const void *d_src, size_t bytes)
{
  if (!c || (bytes > 0 && (!host || !d_src))) return MTP_ERR_ARG;
  if (bytes == 0) return MTP_OK;
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(host, d_src, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    c->last_error = "mtp_copy_to_host failed";
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

const char *mtp_build_flags(void)
{
  // compile-time switches that differ from the shipped defaults; "" for a release build (tests assert exactly that)
  static const std::string flags = [] {
    std::string f;
#ifdef MTP_STAMPS
    f += "MTP_STAMPS ";
#endif
    f += mtp_kernel_build_flags();
    if (!f.empty() && f.back() == ' ') f.pop_back();
    return f;
  }();
  return flags.c_str();
}

int mtp_copy_neighbors_to_host(mtp_context *c, int *first, int *neigh)
{
  if (!c || !first) return MTP_ERR_ARG;
  if (!c->have_list || c->first != c->d_first.ptr || c->neigh != c->d_neigh.ptr) {
    if (c) c->last_error = "the current neighbour list is not owned by the context";
    return MTP_ERR_STATE;
  }
  (void) hipSetDevice(c->device);
  // the list was written on list_stream (a non-blocking stream: the copies below do not wait for it by themselves)
  if (hipStreamSynchronize(c->list_stream ? c->list_stream : c->stream) != hipSuccess) return MTP_ERR_DEVICE;
  if (hipMemcpy(first, c->d_first.ptr, sizeof(int) * ((size_t) c->inum + 1), hipMemcpyDeviceToHost) != hipSuccess)
    return MTP_ERR_DEVICE;
  if (neigh && first[c->inum] > 0 &&
      hipMemcpy(neigh, c->d_neigh.ptr, sizeof(int) * (size_t) first[c->inum], hipMemcpyDeviceToHost) != hipSuccess)
    return MTP_ERR_DEVICE;
  return MTP_OK;
}

int mtp_compute_device_rows(mtp_context *c, void *stream, int row_begin, int row_count, int finish_tallies,
                            const double *d_x, const int *d_type, int eflag, int vflag, int grade_flag, double *d_f,
                            double *d_eatom, double *d_vatom, double *d_ev, double *d_grades, double *d_max_grade,
                            double *d_coeff_ders)
{
  if (!c) return MTP_ERR_ARG;
  if (row_begin < 0 || row_count < 0 || (c->have_list && row_begin + row_count > c->inum)) {
    c->last_error = "row range outside the neighbour list";
    return MTP_ERR_ARG;
  }
  if (!c->have_list) {
    c->last_error = "mtp_compute before mtp_set_neighbors";
    return MTP_ERR_STATE;
  }
  if (!d_x || !d_type || !d_f) return MTP_ERR_ARG;
  if (grade_flag && !c->pot->has_selection) {
    c->last_error = "extrapolation grades requested but the potential has no #MVS_v1.1 selection state";
    return MTP_ERR_STATE;
  }
  const bool cfg = c->pot->configuration_mode;
  if (grade_flag) {
    if (!cfg && !d_grades) {
      c->last_error = "neighbourhood-mode grades need a grades array";
      return MTP_ERR_ARG;
    }
    if (cfg && !d_coeff_ders) {
      c->last_error = "configuration-mode grades need a coeff_ders array";
      return MTP_ERR_ARG;
    }
    if (c->pot->species_count * c->pot->radial_func_count * c->pot->radial_basis_size > 256) {
      c->last_error = "Sp*Mu*R above 256 is not supported by the grade kernels of this build";
      return MTP_ERR_LIMIT;
    }
  }
  if (((eflag & MTP_ENERGY_GLOBAL) || vflag) && finish_tallies && !d_ev) return MTP_ERR_ARG;
  if (c->inum == 0) return MTP_OK;
  if (hipSetDevice(c->device) != hipSuccess) {
    c->last_error = "hipSetDevice failed";
    return MTP_ERR_DEVICE;
  }
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  MtpDevParams p = c->base;
  p.inum = row_count;      // rows of THIS launch: [row0, row0 + inum) of the installed list
  p.row0 = row_begin;
  p.nall = c->nall;
  p.ilist = c->ilist;
  p.first = c->first;
  p.neigh = c->neigh;
  p.x = d_x;
  p.type = d_type;
  p.f = d_f;
  p.fq = nullptr;
  p.eatom = d_eatom;
  p.vatom = d_vatom;
  p.eflag = eflag;
  p.vflag = vflag;
  p.grade_flag = grade_flag ? 1 : 0;
  p.xcd_map = c->xcd_map && row_count >= 8 * 64 ? 1 : 0;
  const mtp_context::LaunchPlan &L = c->lp[grade_flag ? 2 : 0];
  // The plan covers the whole list.  A row range too short to fill it (the interior / boundary pieces of a small
  // domain, strong scaling) runs in smaller workgroups, so that its atoms spread over all CUs instead of filling a
  // few of them: per-atom latency, not throughput, is what such a launch waits for.
  int wpb_launch = L.wpb;
  const int waves_per_cu = L.wpb * std::max(1, L.grid / std::max(1, c->num_cus));
  if (row_count < c->num_cus * waves_per_cu && L.grid >= c->num_cus) {
    const int small = L.wps == 3 ? 4 : 2;   // (multiples of 4 keep the SIMDs balanced at 3 per SIMD)
    if (small < wpb_launch) wpb_launch = small;
  }
  const size_t blob_launch = (size_t) L.blob_bytes;
  const size_t lds_launch = blob_launch + (size_t) wpb_launch * L.wave_doubles * 8;
  auto grid_for = [&](const mtp_context::LaunchPlan &lp_, int wpb_) {
    int g = (row_count + wpb_ - 1) / wpb_;
    if (wpb_ == lp_.wpb) g = std::min(lp_.grid, g);
    else g = std::min(g, c->num_cus * (int) std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds_launch, 1)));
    if (g >= 8) g = (g + 7) / 8 * 8;   // whole rounds of the 8 XCDs for the XCD-aware atom map
    return std::max(1, g);
  };
  p.tab_rows = L.tab_rows;
  p.m_doubles = L.m_doubles;
  p.Am = grade_flag ? c->pot->alpha_moment_count : c->pot->stored_moment_count;
  p.ov_doubles = L.ov_doubles;
  p.rebuild_tables = L.rebuild ? 1 : 0;
  p.rows_in_lds = L.rows_lds ? 1 : 0;
  p.blob_bytes = L.blob_bytes;
  p.tgt_in_lds = L.tgt_lds ? 1 : 0;
  p.dg_mode = L.layout.mode;
  p.pow_row = L.layout.pow_row;
  p.dg_off = L.layout.dg_off;
  p.fp_row = L.layout.fp_row;
  p.w_m = L.layout.off_m;
  p.w_d = L.layout.off_d;
  p.w_coef = L.layout.off_coef;
  p.w_nb = L.layout.off_nb;
  p.wps = L.wps;
  p.wave_doubles = L.wave_doubles;
  p.cvec = grade_flag ? c->d_cvec.ptr : nullptr;
  p.cpad = c->cpad;
  // the force kernel writes the radial block of the candidate vectors itself for the common table shape; other
  // shapes leave the adjoints of the basics in HBM for mtp_cvec_kernel
  const bool fused = grade_flag && c->pot->radial_basis_size == 8 && c->pot->radial_func_count <= 4 &&
      c->pot->species_count <= 2 && std::getenv("MTP_GRADE_UNFUSED") == nullptr;
  p.grade_fused = fused ? 1 : 0;
  p.dbasic = grade_flag && !fused ? c->d_dbasic.ptr : nullptr;
  p.dpad = c->dpad;
  try {
    if (c->timing) {
      if (!c->ev0) {
        HIP_CHECK(hipEventCreate(&c->ev0));
        HIP_CHECK(hipEventCreate(&c->ev1));
      }
      HIP_CHECK(hipEventRecord(c->ev0, st));
    }
    if (c->deterministic && row_count > 0) {
      const size_t n3 = 3 * (size_t) c->nall;
      if (c->d_fq.cap < n3) {
        c->d_fq.reserve(n3);
        HIP_CHECK(hipMemsetAsync(c->d_fq.ptr, 0, n3 * sizeof(long long), st));
      }
      p.fq = c->d_fq.ptr;
    }
    if (row_count > 0) HIP_CHECK(mtp_launch_wave_kernel(p, grid_for(L, wpb_launch), wpb_launch, lds_launch, st));
    if (p.fq) HIP_CHECK(mtp_launch_fixed_to_force(p.fq, d_f, c->nall, st));
    if (c->timing) {
      HIP_CHECK(hipEventRecord(c->ev1, st));
      c->timed = true;
    }
    // the per-wavefront tally slots keep accumulating over the launches of a step; the last one folds them
    if (finish_tallies && ((eflag & MTP_ENERGY_GLOBAL) || vflag)) HIP_CHECK(mtp_launch_ev_finish(c->d_ev_slots.ptr, d_ev, st));
    if (grade_flag && row_count > 0) {
      if (!fused) {
        MtpDevParams pc = p;   // radial block of the candidate vectors from the adjoints left in HBM
        pc.blob_bytes = c->blob_bytes_norows;
        pc.rows_in_lds = 0;
        pc.wave_doubles = c->lp[1].wave_doubles;
        pc.tab_rows = c->lp[1].tab_rows;
        HIP_CHECK(mtp_launch_cvec_kernel(pc, std::max(1, std::min(c->lp[1].grid, (row_count + c->lp[1].wpb - 1) / c->lp[1].wpb)),
                                         c->lp[1].wpb, c->lp[1].lds_bytes, st));
      }
      const double *cv = c->d_cvec.ptr + (size_t) row_begin * c->cpad;
      if (cfg)
        HIP_CHECK(mtp_launch_colsum_kernel(cv, c->cpad, c->pot->coeff_count, row_count, d_coeff_ders, st));
      else
        HIP_CHECK(mtp_launch_grade_kernel(cv, c->d_ainv_pad.ptr, c->d_ainv_tiled.ptr, c->cpad, c->pot->coeff_count, row_count,
                                          c->ilist + row_begin, d_grades, d_max_grade, st));
    }
  } catch (const HipFail &f) {
    c->last_error = std::string(f.what) + ": " + hipGetErrorString(f.e);
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_compute_device(mtp_context *c, void *stream, const double *d_x, const int *d_type, int eflag, int vflag,
                       int grade_flag, double *d_f, double *d_eatom, double *d_vatom, double *d_ev,
                       double *d_grades, double *d_max_grade, double *d_coeff_ders)
{
  if (!c) return MTP_ERR_ARG;
  return mtp_compute_device_rows(c, stream, 0, c->inum, 1, d_x, d_type, eflag, vflag, grade_flag, d_f, d_eatom, d_vatom,
                                 d_ev, d_grades, d_max_grade, d_coeff_ders);
}

int mtp_synchronize(mtp_context *c, void *stream)
{
  if (!c) return MTP_ERR_ARG;
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  int flag = 0;
  hipError_t e = hipMemcpyAsync(&flag, c->d_err.ptr, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    c->last_error = std::string("synchronize: ") + hipGetErrorString(e);
    return MTP_ERR_DEVICE;
  }
  if (flag) {
    (void) hipMemsetAsync(c->d_err.ptr, 0, sizeof(int), st);
    (void) hipStreamSynchronize(st);
    if (flag == 2) {
      c->last_error = "a neighbour list row holds more in-cutoff neighbours than the declared max_numneigh";
      return MTP_ERR_LIMIT;
    }
    c->last_error = "Too few species count in the MTP potential!";   // pair_mtp.cpp:92-93
    return MTP_ERR_SPECIES;
  }
  return MTP_OK;
}

int mtp_compute(mtp_context *c, const double *x, const int *type, int eflag, int vflag, int grade_flag,
                double *f, double *eatom, double *vatom, double *energy, double *virial, double *grades,
                double *max_grade, double *coeff_ders)
{
  if (!c || !x || !type || !f) return MTP_ERR_ARG;
  if (!c->have_list) {
    c->last_error = "mtp_compute before mtp_set_neighbors";
    return MTP_ERR_STATE;
  }
  const size_t nall = (size_t) c->nall;
  try {
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->d_x.upload(x, 3 * nall, st);
    c->d_type.upload(type, nall, st);
    c->d_f.reserve(3 * nall);
    HIP_CHECK(hipMemsetAsync(c->d_f.ptr, 0, 3 * nall * sizeof(double), st));
    const bool want_ea = (eflag & MTP_ENERGY_ATOM) && eatom;
    const bool want_va = (vflag & MTP_VIRIAL_ATOM) && vatom;
    if (want_ea) c->d_eatom.upload(eatom, nall, st);   // keeps entries of atoms outside ilist
    if (want_va) {
      c->d_vatom.reserve(6 * nall);
      HIP_CHECK(hipMemsetAsync(c->d_vatom.ptr, 0, 6 * nall * sizeof(double), st));
    }
    HIP_CHECK(hipMemsetAsync(c->d_ev.ptr, 0, 8 * sizeof(double), st));
    if (grade_flag) {
      c->d_grades.reserve(nall);
      c->d_coeff.reserve((size_t) c->pot->coeff_count);
      HIP_CHECK(hipMemsetAsync(c->d_grades.ptr, 0, nall * sizeof(double), st));
      HIP_CHECK(hipMemsetAsync(c->d_coeff.ptr, 0, (size_t) c->pot->coeff_count * sizeof(double), st));
      HIP_CHECK(hipMemsetAsync(c->d_maxg.ptr, 0, sizeof(double), st));
    }
    int rc = mtp_compute_device(c, st, c->d_x.ptr, c->d_type.ptr, eflag, vflag, grade_flag, c->d_f.ptr,
                                want_ea ? c->d_eatom.ptr : nullptr, want_va ? c->d_vatom.ptr : nullptr,
                                c->d_ev.ptr, grade_flag ? c->d_grades.ptr : nullptr,
                                grade_flag ? c->d_maxg.ptr : nullptr, grade_flag ? c->d_coeff.ptr : nullptr);
    if (rc != MTP_OK) return rc;
    c->h_tmp.resize(std::max<size_t>(6 * nall, 16));
    HIP_CHECK(hipMemcpyAsync(c->h_tmp.data(), c->d_f.ptr, 3 * nall * sizeof(double), hipMemcpyDeviceToHost, st));
    rc = mtp_synchronize(c, st);
    if (rc != MTP_OK) return rc;
    for (size_t q = 0; q < 3 * nall; q++) f[q] += c->h_tmp[q];
    if (want_ea) HIP_CHECK(hipMemcpy(eatom, c->d_eatom.ptr, nall * sizeof(double), hipMemcpyDeviceToHost));
    if (want_va) {
      HIP_CHECK(hipMemcpy(c->h_tmp.data(), c->d_vatom.ptr, 6 * nall * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t q = 0; q < 6 * nall; q++) vatom[q] += c->h_tmp[q];
    }
    double ev[8];
    HIP_CHECK(hipMemcpy(ev, c->d_ev.ptr, 8 * sizeof(double), hipMemcpyDeviceToHost));
    if ((eflag & MTP_ENERGY_GLOBAL) && energy) *energy += ev[0];
    if (vflag && virial)
      for (int q = 0; q < 6; q++) virial[q] += ev[1 + q];
    if (grade_flag) {
      if (grades) HIP_CHECK(hipMemcpy(grades, c->d_grades.ptr, nall * sizeof(double), hipMemcpyDeviceToHost));
      if (max_grade) HIP_CHECK(hipMemcpy(max_grade, c->d_maxg.ptr, sizeof(double), hipMemcpyDeviceToHost));
      if (coeff_ders)
        HIP_CHECK(hipMemcpy(coeff_ders, c->d_coeff.ptr, (size_t) c->pot->coeff_count * sizeof(double),
                            hipMemcpyDeviceToHost));
    }
  } catch (const HipFail &fl) {
    c->last_error = std::string(fl.what) + ": " + hipGetErrorString(fl.e);
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_context_launch_info(const mtp_context *c, int32_t *lds_bytes_per_wave, int32_t *waves_per_block,
                            int32_t *grid_blocks, int32_t *neighbor_tile)
{
  if (!c || !c->have_list) return MTP_ERR_STATE;
  const mtp_context::LaunchPlan &L = c->lp[0];
  if (lds_bytes_per_wave) *lds_bytes_per_wave = L.wave_doubles * 8;   // per atom image
  if (waves_per_block) *waves_per_block = L.wpb;
  if (grid_blocks) *grid_blocks = L.grid;
  if (neighbor_tile) *neighbor_tile = c->base.NT;
  return MTP_OK;
}

int mtp_context_plan_info(const mtp_context *c, int32_t *waves_per_simd, int32_t *rebuild_tables)
{
  if (!c || !c->have_list) return MTP_ERR_STATE;
  if (waves_per_simd) *waves_per_simd = c->lp[0].wps;
  if (rebuild_tables) *rebuild_tables = c->lp[0].rebuild ? 1 : 0;
  return MTP_OK;
}

// diagnostic builds only: read and clear the per-phase cycle sums (not part of the public ABI)
int mtp_debug_read_stamps(mtp_context *c, unsigned long long *out16)
{
  if (!c || !out16) return MTP_ERR_ARG;
  if (hipMemcpy(out16, c->d_stamps.ptr, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return MTP_ERR_DEVICE;
  (void) hipMemset(c->d_stamps.ptr, 0, 16 * sizeof(unsigned long long));
  return MTP_OK;
}

}   // extern "C"

void *mtp_internal_resolve_stream(mtp_context *c, void *stream)
{
  return stream ? stream : (c ? reinterpret_cast<void *>(c->stream) : nullptr);
}

int mtp_internal_finish_unpack(mtp_context *c, void *stream, int eflag, int vflag, double *d_ev, double *d_f, const int *d_idx,
                               const double *d_frecv, int n3)
{
  if (!c || !d_f || n3 < 0) return MTP_ERR_ARG;
  const int fold = ((eflag & MTP_ENERGY_GLOBAL) || vflag) ? 1 : 0;
  if (fold && !d_ev) return MTP_ERR_ARG;
  if (!fold && n3 == 0) return MTP_OK;
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  return mtp_launch_ev_finish_unpack(c->d_ev_slots.ptr, d_ev, fold, d_f, d_idx, d_frecv, n3, st) == hipSuccess ? MTP_OK
                                                                                                                : MTP_ERR_DEVICE;
}

extern "C" {

int mtp_zero_async(void *stream, double *d_p, long long n)
{
  if (n < 0 || (n > 0 && !d_p) || (reinterpret_cast<uintptr_t>(d_p) & 15u)) return MTP_ERR_ARG;
  if (!stream) return MTP_ERR_ARG;   // no context here: NULL is not mapped to anything (include/mtp_mi355x.h, "Streams")
  return mtp_launch_zero(d_p, (size_t) n, reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? MTP_OK : MTP_ERR_DEVICE;
}

int mtp_context_set_deterministic(mtp_context *c, int enable)
{
  if (!c) return MTP_ERR_ARG;
  c->deterministic = enable != 0;
  return MTP_OK;
}

int mtp_context_set_timing(mtp_context *c, int enable)
{
  if (!c) return MTP_ERR_ARG;
  c->timing = enable != 0;
  c->timed = false;
  return MTP_OK;
}

int mtp_context_last_kernel_ms(mtp_context *c, float *ms)
{
  if (!c || !ms) return MTP_ERR_ARG;
  if (!c->timed) return MTP_ERR_STATE;
  hipError_t e = hipEventSynchronize(c->ev1);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, c->ev0, c->ev1);
  if (e != hipSuccess) {
    c->last_error = std::string("event timing: ") + hipGetErrorString(e);
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

}   // extern "C"
