// EXPERIMENTAL (not on the default path; MTP_TEAM=1 selects it; see the note in mtp_context.hip: it
// measured slower than the wavefront-per-atom kernel on MI355X).
// Workgroup-per-atom variant of the fused MTP force kernel (the native counterpart of the
// reference's block-parallel style, KOKKOS/pair_mtps_kokkos.cpp:438-708): WPA = 4 wavefronts share
// one atom's LDS image (tables, moments, adjoints) and split every phase, so the per-atom latency
// chain of the wavefront-per-atom kernel (mtp_kernels.hip) is cut roughly by WPA while the LDS
// footprint per atom stays the same.  Few atoms (or few atoms per GPU after domain decomposition)
// then still fill the chip, and at 64k atoms more wavefronts are resident per CU.
//
// Work split inside the team (team thread id tid = 64 wv + lane):
//   compaction   tid = candidate list entry (up to 256 per round), per-wavefront counts through LDS
//   tile tables  tid = (neighbour, mu) item, then (neighbour, axis) item
//   moments      wavefront wv owns the basics k = kl + KL (wv + WPA tw); same lane grid as the
//                wavefront kernel inside a wavefront (NG neighbour groups x KL k-lanes)
//   products     tid = times row of the current dependency level
//   forces       wavefront wv contracts its own basics; butterfly inside the wavefront, then the four
//                partial (neighbour, component) sums meet in LDS and wavefront 0 scatters
// Phase boundaries are __syncthreads() (workgroup = one team).
#include <hip/hip_runtime.h>

#include "mtp_kernel_common.hpp"

namespace {

template <int PITCH, int WPA> struct TeamLds {
  static constexpr int NT = PITCH - 2;
  double *M, *D, *tab, *nbx, *nby, *nbz, *nbr, *nbi, *red;
  int *nbj, *nbjt, *cj, *cnts;
  unsigned m_addr;   // LDS byte address of M
  __device__ __forceinline__ unsigned addr(const double *ptr) const { return m_addr + 8u * (unsigned) (ptr - M); }
  __device__ __forceinline__ TeamLds(double *base, unsigned base_addr, const MtpDevParams &p)
  {
    M = base;
    m_addr = base_addr;
    D = M + p.m_doubles;
    tab = D + p.d_doubles;
    nbx = tab + (size_t) p.tab_rows * PITCH;
    nby = nbx + NT;
    nbz = nby + NT;
    nbr = nbz + NT;
    nbi = nbr + NT;
    red = nbi + NT;                                   // WPA * 64 doubles
    nbj = reinterpret_cast<int *>(red + WPA * 64);
    nbjt = nbj + NT;
    cnts = nbjt + NT;                                 // 8 ints
    cj = cnts + 8;
  }
};

// tables of one tile, all team threads; columns [0, ntp) are written (see mtp_kernels.hip build_tile)
template <int PITCH, int WPA>
__device__ __forceinline__ void team_build_tile(const MtpDevParams &p, const BlockTables &bt,
                                                const TeamLds<PITCH, WPA> &w, int t0, int cnt, int ntp, bool gather,
                                                double xi0, double xi1, double xi2, int i, int itype, int tid)
{
  constexpr int TEAM = 64 * WPA;
  if (gather) {
    if (tid < ntp) {
      const bool real = t0 + tid < cnt;
      const int j = real ? w.cj[t0 + tid] : i;
      double dx = 0, dy = 0, dz = 0, r = p.rmax;
      if (real) {
        dx = p.x[3 * (size_t) j] - xi0;
        dy = p.x[3 * (size_t) j + 1] - xi1;
        dz = p.x[3 * (size_t) j + 2] - xi2;
        r = sqrt(dx * dx + dy * dy + dz * dz);
      }
      w.nbx[tid] = dx;
      w.nby[tid] = dy;
      w.nbz[tid] = dz;
      w.nbr[tid] = r;
      w.nbi[tid] = 1.0 / r;
      w.nbj[tid] = j;
      w.nbjt[tid] = real ? p.type[j] - 1 : itype;
    }
    __syncthreads();
  }
  const int Mu = p.Mu, P = p.P, R = p.R;
  const double span = p.rmax - p.rmin, mult = 2.0 / span;
  for (int idx = tid; idx < ntp * Mu; idx += TEAM) {
    const int n = __float2int_rz((idx + 0.5f) * p.inv_mu), mu = idx - n * Mu;
    const double r = w.nbr[n], inv = w.nbi[n];
    const int jt = w.nbjt[n];
    const double *c = bt.radial + ((itype * p.Sp + jt) * Mu + mu) * R;
    const double d = r - p.rmax;
    const double ksi = (2.0 * r - (p.rmin + p.rmax)) / span;
    double q0 = p.scaling * (d * d), q1 = p.scaling * (ksi * d * d);
    double e0 = p.scaling * 2.0 * d, e1 = p.scaling * (mult * d * d + 2.0 * ksi * d);
    double val = c[0] * q0, der = c[0] * e0;
    if (R > 1) {
      val += c[1] * q1;
      der += c[1] * e1;
    }
    for (int ri = 2; ri < R; ri++) {
      const double q2 = 2.0 * ksi * q1 - q0;
      const double e2 = 2.0 * (mult * q1 + ksi * e1) - e0;
      val += c[ri] * q2;
      der += c[ri] * e2;
      q0 = q1;
      q1 = q2;
      e0 = e1;
      e1 = e2;
    }
    double *col = w.tab + n;
    const int *sl = bt.slot + mu * MTP_PSTRIDE;
    double rp = 1.0;
    for (int nu = 0; nu < P; nu++) {
      const int s = sl[nu];
      const double g = val * rp;
      if (s >= 0) {
        col[s * PITCH] = g;                                       // f_mu / r^nu
        col[(p.nslot + s) * PITCH] = der * rp - nu * g * inv;     // d/dr (f_mu / r^nu)
      }
      rp *= inv;
    }
  }
  for (int idx = tid; idx < ntp * 3; idx += TEAM) {
    const int n = __float2int_rz((idx + 0.5f) * (1.0f / 3.0f)), ax = idx - 3 * n;
    const double u = ax == 0 ? w.nbx[n] : (ax == 1 ? w.nby[n] : w.nbz[n]);
    double *col = w.tab + (size_t) (2 * p.nslot + ax * P) * PITCH + n;
    double cur = 1.0;
    col[0] = 1.0;
    for (int q = 1; q < P; q++) {
      cur *= u;
      col[q * PITCH] = cur;
    }
  }
  __syncthreads();
}

// products of one atom, rows from `rows` (LDS or HBM: two call sites, never a pointer select)
template <int TEAM>
__device__ __forceinline__ void team_products_forward(const MtpRow8 *rows, const int *level, int nlevels, double *M,
                                                      int tid)
{
  for (int l = 0; l < nlevels; l++) {
    const int end = level[l + 1];
    for (int r0 = level[l] + tid; r0 < end; r0 += 2 * TEAM) {
      MtpRow8 rw[2];
      double v[2];
#pragma unroll
      for (int u = 0; u < 2; u++) rw[u] = rows[min(r0 + TEAM * u, end - 1)];
#pragma unroll
      for (int u = 0; u < 2; u++) v[u] = M[rw[u].lo & 0xffffu] * M[rw[u].lo >> 16];
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (r0 + TEAM * u < end) lds_add(&M[rw[u].hi & 0xffffu], (double) ((int) rw[u].hi >> 16) * v[u]);
    }
    __syncthreads();
  }
}

template <int TEAM>
__device__ __forceinline__ void team_products_backward(const MtpRow8 *rows, const int *level, int nlevels,
                                                       const double *M, double *D, int tid)
{
  for (int l = nlevels - 1; l >= 0; l--) {
    const int end = level[l + 1];
    for (int r0 = level[l] + tid; r0 < end; r0 += 2 * TEAM) {
      MtpRow8 rw[2];
      double d3[2], m0[2], m1[2];
#pragma unroll
      for (int u = 0; u < 2; u++) rw[u] = rows[min(r0 + TEAM * u, end - 1)];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        d3[u] = D[rw[u].hi & 0xffffu] * (double) ((int) rw[u].hi >> 16);
        m0[u] = M[rw[u].lo & 0xffffu];
        m1[u] = M[rw[u].lo >> 16];
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (r0 + TEAM * u < end) {
          lds_add(&D[rw[u].lo >> 16], d3[u] * m0[u]);
          lds_add(&D[rw[u].lo & 0xffffu], d3[u] * m1[u]);
        }
    }
    __syncthreads();
  }
}

template <int KL, int KBW, int WPA, int PITCH>
__global__ void __launch_bounds__(64 * WPA, 3) mtp_team_kernel(const MtpDevParams p)
{
  constexpr int TEAM = 64 * WPA;
  constexpr int NT = PITCH - 2;
  constexpr int NG = 64 / KL;
  constexpr int NPG = NT / NG;
  constexpr int BATCH = KL / 4;
  constexpr int NBATCH = NPG / BATCH;
  constexpr int KP = KL * KBW * WPA;   // padded number of basics

  extern __shared__ double lds[];
  unsigned char *sh = reinterpret_cast<unsigned char *>(lds);
  for (int o = threadIdx.x * 16; o < p.blob_bytes; o += TEAM * 16)
    *reinterpret_cast<uint4 *>(sh + o) = *reinterpret_cast<const uint4 *>(p.blob + o);
  __syncthreads();
  BlockTables bt;
  bt.rows = reinterpret_cast<const MtpRow8 *>(sh + p.off_rows);
  bt.level = reinterpret_cast<const int *>(sh + p.off_level);
  bt.slot = reinterpret_cast<const int *>(sh + p.off_slot);
  bt.radial = reinterpret_cast<const double *>(sh + p.off_radial);
  bt.seed_idx = reinterpret_cast<const int *>(sh + p.off_seed_idx);
  bt.seed_val = reinterpret_cast<const double *>(sh + p.off_seed_val);
  bt.map = reinterpret_cast<const int *>(sh + p.off_map);
  bt.lin = reinterpret_cast<const double *>(sh + p.off_lin);
  bt.pack = reinterpret_cast<const int *>(sh + p.off_pack);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & (KL - 1), q = lane / KL;
  const unsigned team_off = p.blob_bytes >> 3;
  const unsigned lds0 = (unsigned) (size_t) (lds_cdouble *) lds;
  const TeamLds<PITCH, WPA> w(lds + team_off, lds0 + 8u * team_off, p);
  const int ns = p.nslot, P = p.P;
  const int KBtot = (p.B + KL - 1) / KL;   // basics blocks that hold anything

  // descriptors of this wavefront's basics: k = kl + KL (wv + WPA tw)
  unsigned pg[KBW], pd[KBW], px[KBW], py[KBW], pz[KBW];
  bool kval[KBW];
#pragma unroll
  for (int tw = 0; tw < KBW; tw++) {
    const int k = kl + KL * (wv + WPA * tw);
    kval[tw] = k < p.B;
    const int pk = kval[tw] ? bt.pack[k] : 0;
    const int a = (pk >> 8) & 15, b = (pk >> 12) & 15, c = (pk >> 16) & 15;
    pg[tw] = w.addr(w.tab + (size_t) (pk & 255) * PITCH + q);
    pd[tw] = w.addr(w.tab + (size_t) (ns + (pk & 255)) * PITCH + q);
    px[tw] = w.addr(w.tab + (size_t) (2 * ns + a - 1) * PITCH + q);
    py[tw] = w.addr(w.tab + (size_t) (2 * ns + P + b - 1) * PITCH + q);
    pz[tw] = w.addr(w.tab + (size_t) (2 * ns + 2 * P + c - 1) * PITCH + q);
    asm volatile("" : "+v"(pg[tw]), "+v"(pd[tw]), "+v"(px[tw]), "+v"(py[tw]), "+v"(pz[tw]));
  }

  double tally = 0.0;   // wavefront 0, lane 9: energy, lanes 3..8: virial components

  for (int ii = blockIdx.x; ii < p.inum; ii += gridDim.x) {
    const int i = __builtin_amdgcn_readfirstlane(p.ilist[ii]);
    const int itype = __builtin_amdgcn_readfirstlane(p.type[i] - 1);
    if (itype < 0 || itype >= p.Sp) {   // pair_mtp.cpp:91-93 (block-uniform: no barrier is skipped unevenly)
      if (tid == 0) atomicExch(p.err_flag, 1);
      continue;
    }
    const double xi0 = uniform_f64(p.x[3 * (size_t) i]), xi1 = uniform_f64(p.x[3 * (size_t) i + 1]),
                 xi2 = uniform_f64(p.x[3 * (size_t) i + 2]);
    const int jbeg = __builtin_amdgcn_readfirstlane(p.first[ii]);
    const int jnum = __builtin_amdgcn_readfirstlane(p.first[ii + 1]) - jbeg;

    // ---- 1. compaction: up to TEAM candidates per round ------------------------------------------
    int cnt = 0;
    for (int c0 = 0; c0 < jnum; c0 += TEAM) {
      const int jj = c0 + tid;
      bool in = false;
      int j = 0, jt = 0;
      double dx = 0, dy = 0, dz = 0, r2 = 1.0;
      if (jj < jnum) {
        j = p.neigh[jbeg + jj] & MTP_NEIGHMASK;
        jt = p.type[j] - 1;
        if (jt < 0 || jt >= p.Sp) {   // pair_mtp.cpp:116-118
          atomicExch(p.err_flag, 1);
        } else {
          dx = p.x[3 * (size_t) j] - xi0;
          dy = p.x[3 * (size_t) j + 1] - xi1;
          dz = p.x[3 * (size_t) j + 2] - xi2;
          r2 = dx * dx + dy * dy + dz * dz;
          in = !(r2 > p.cutsq);   // pair_mtp.cpp:123
        }
      }
      const unsigned long long m = __ballot(in);
      if (lane == 0) w.cnts[wv] = __popcll(m);
      __syncthreads();
      int base = cnt, tot = 0;
#pragma unroll
      for (int ww = 0; ww < WPA; ww++) {
        const int c = w.cnts[ww];
        if (ww < wv) base += c;
        tot += c;
      }
      if (in) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        w.cj[pos] = j;
        if (pos < NT) {
          const double r = sqrt(r2);
          w.nbx[pos] = dx;
          w.nby[pos] = dy;
          w.nbz[pos] = dz;
          w.nbr[pos] = r;
          w.nbi[pos] = 1.0 / r;
          w.nbj[pos] = j;
          w.nbjt[pos] = jt;
        }
      }
      cnt += __builtin_amdgcn_readfirstlane(tot);
      __syncthreads();   // cnts is reused next round; cj / tile arrays complete
    }
    {   // dummy neighbours pad tile 0 to a multiple of NG
      const int pos = cnt + tid;
      if (cnt < NT && tid < NG && pos < ((min(cnt, NT) + NG - 1) / NG) * NG) {
        w.nbx[pos] = 0.0;
        w.nby[pos] = 0.0;
        w.nbz[pos] = 0.0;
        w.nbr[pos] = p.rmax;
        w.nbi[pos] = 1.0 / p.rmax;
        w.nbj[pos] = i;
        w.nbjt[pos] = itype;
      }
    }
    __syncthreads();

    // ---- 2+3. tiles: tables (whole team), basic moments (each wavefront its own basics) ---------
    double acc[KBW];
#pragma unroll
    for (int tw = 0; tw < KBW; tw++) acc[tw] = 0.0;
    const int ntiles = (cnt + NT - 1) / NT;
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * NT, nt = min(NT, cnt - t0), ntp = ((nt + NG - 1) / NG) * NG;
      team_build_tile<PITCH, WPA>(p, bt, w, t0, cnt, ntp, tile > 0, xi0, xi1, xi2, i, itype, tid);
#pragma unroll
      for (int m = 0; m < NPG; m++) {
        if (m * NG < ntp) {
          double G[KBW], X[KBW], Y[KBW], Z[KBW];
#pragma unroll
          for (int tw = 0; tw < KBW; tw++) {
            if (wv + WPA * tw < KBtot) {   // wave-uniform: blocks past the table cost nothing
              G[tw] = lds_ld(pg[tw], m * NG);
              X[tw] = lds_ld(px[tw], PITCH + m * NG);
              Y[tw] = lds_ld(py[tw], PITCH + m * NG);
              Z[tw] = lds_ld(pz[tw], PITCH + m * NG);
            } else {
              G[tw] = X[tw] = Y[tw] = Z[tw] = 0.0;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tw = 0; tw < KBW; tw++) acc[tw] += G[tw] * (X[tw] * (Y[tw] * Z[tw]));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (ntiles > 1) __syncthreads();
    }
#pragma unroll
    for (int tw = 0; tw < KBW; tw++) {
      if (NG >= 2) acc[tw] += shfl_xor_f64(acc[tw], KL);
      if (NG >= 4) acc[tw] += shfl_xor_f64(acc[tw], 2 * KL);
    }
    for (int m = p.B + tid; m < p.A; m += TEAM) w.M[m] = 0.0;
    for (int m = tid; m < p.A; m += TEAM) w.D[m] = 0.0;
    if (q == 0) {
#pragma unroll
      for (int tw = 0; tw < KBW; tw++)
        if (kval[tw]) w.M[kl + KL * (wv + WPA * tw)] = acc[tw];
    }
    __syncthreads();

    // ---- 4a. products forward, level by level (pair_mtp.cpp:196-201) ------------------------------
    if (p.rows_in_lds) team_products_forward<TEAM>(bt.rows, bt.level, p.nlevels, w.M, tid);
    else team_products_forward<TEAM>(p.rows, bt.level, p.nlevels, w.M, tid);
    // ---- site energy (pair_mtp.cpp:204-212): every wavefront sums a slice, wavefront 0 combines ---
    {
      double e = 0.0;
      for (int k = tid; k < p.S; k += TEAM) e += bt.lin[k] * w.M[bt.map[k]];
      e = wave_sum(e);
      if (lane == 0) w.red[wv] = e;
    }
    // ---- 4b. adjoints (pair_mtp.cpp:217-233) -------------------------------------------------------
    for (int k = tid; k < p.nseed; k += TEAM) w.D[bt.seed_idx[k]] = bt.seed_val[k];
    __syncthreads();
    double e = p.species_coeffs[itype];
#pragma unroll
    for (int ww = 0; ww < WPA; ww++) e += w.red[ww];
    if (p.rows_in_lds) team_products_backward<TEAM>(bt.rows, bt.level, p.nlevels, w.M, w.D, tid);
    else team_products_backward<TEAM>(p.rows, bt.level, p.nlevels, w.M, w.D, tid);

    // ---- 5. forces ---------------------------------------------------------------------------------
    // DK/DA/DB/DC[KP] (adjoints of the basics, plain and times the chain-rule exponents) into the moment region
    for (int k = tid; k < KP; k += TEAM) {
      const bool ok = k < p.B;
      const double d = ok ? w.D[k] : 0.0;
      const int pk = ok ? bt.pack[k] : 0;
      w.M[k] = d;
      w.M[KP + k] = d * (double) ((pk >> 8) & 15);
      w.M[2 * KP + k] = d * (double) ((pk >> 12) & 15);
      w.M[3 * KP + k] = d * (double) ((pk >> 16) & 15);
    }
    unsigned pda = w.addr(w.M + kl + KL * wv);
    asm volatile("" : "+v"(pda));
    __syncthreads();
    double fi0 = 0, fi1 = 0, fi2 = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;   // wavefront 0
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * NT, nt = min(NT, cnt - t0), ntp = ((nt + NG - 1) / NG) * NG;
      if (ntiles > 1) team_build_tile<PITCH, WPA>(p, bt, w, t0, cnt, ntp, true, xi0, xi1, xi2, i, itype, tid);
#pragma unroll
      for (int b = 0; b < NBATCH; b++) {
        if (b * BATCH * NG < ntp) {   // block-uniform
          double part[KL];
#pragma unroll
          for (int u = 0; u < KL; u++) part[u] = 0.0;
#pragma unroll
          for (int tw = 0; tw < KBW; tw++) {
            if (wv + WPA * tw < KBtot) {
              const double Dk = lds_ld(pda, KL * WPA * tw), Da = lds_ld(pda, KP + KL * WPA * tw);
              const double Db = lds_ld(pda, 2 * KP + KL * WPA * tw), Dc = lds_ld(pda, 3 * KP + KL * WPA * tw);
              constexpr int MC = BATCH < 4 ? BATCH : 4;
#pragma unroll
              for (int m0 = 0; m0 < BATCH; m0 += MC) {
                double g[MC], gd[MC], xm[MC], xa[MC], ym[MC], yb[MC], zm[MC], zc[MC];
#pragma unroll
                for (int u = 0; u < MC; u++) {
                  const int o = (b * BATCH + m0 + u) * NG;
                  g[u] = lds_ld(pg[tw], o);
                  gd[u] = lds_ld(pd[tw], o);
                  xm[u] = lds_ld(px[tw], o);
                  xa[u] = lds_ld(px[tw], PITCH + o);
                  ym[u] = lds_ld(py[tw], o);
                  yb[u] = lds_ld(py[tw], PITCH + o);
                  zm[u] = lds_ld(pz[tw], o);
                  zc[u] = lds_ld(pz[tw], PITCH + o);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < MC; u++) {
                  const int mm = m0 + u;
                  const double yz = yb[u] * zc[u], xz = xa[u] * zc[u], xy = xa[u] * yb[u];
                  part[4 * mm + 0] += (Dk * gd[u]) * (xa[u] * yz);
                  part[4 * mm + 1] += (Da * g[u]) * (xm[u] * yz);
                  part[4 * mm + 2] += (Db * g[u]) * (ym[u] * xz);
                  part[4 * mm + 3] += (Dc * g[u]) * (zm[u] * xy);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          Butterfly<KL>::run(part, lane);
          w.red[wv * 64 + lane] = part[0];   // this wavefront's share of value kl of group q
          __syncthreads();
          const int n = q + NG * (b * BATCH + kl);
          if (wv == 0 && kl < BATCH && n < nt) {
            double s4[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
              double s = 0.0;
#pragma unroll
              for (int ww = 0; ww < WPA; ww++) s += w.red[ww * 64 + q * KL + 4 * kl + c];
              s4[c] = s;
            }
            const double sr = s4[0] * w.nbi[n];
            const double rx = w.nbx[n], ry = w.nby[n], rz = w.nbz[n];
            const double Fx = sr * rx + s4[1];
            const double Fy = sr * ry + s4[2];
            const double Fz = sr * rz + s4[3];
            const size_t j = (size_t) w.nbj[n];
            unsafeAtomicAdd(&p.f[3 * j + 0], -Fx);   // pair_mtp.cpp:252-254
            unsafeAtomicAdd(&p.f[3 * j + 1], -Fy);
            unsafeAtomicAdd(&p.f[3 * j + 2], -Fz);
            fi0 += Fx;
            fi1 += Fy;
            fi2 += Fz;
            if (p.vflag) {   // pair_mtp.cpp:257-277
              v0 -= Fx * rx;
              v1 -= Fy * ry;
              v2 -= Fz * rz;
              v3 -= (Fx * ry + Fy * rx) * 0.5;
              v4 -= (Fx * rz + Fz * rx) * 0.5;
              v5 -= (Fy * rz + Fz * ry) * 0.5;
            }
          }
          __syncthreads();   // red is reused by the next batch
        }
      }
    }
    // ---- per-atom totals (wavefront 0): 9 values x 16 lanes through LDS ---------------------------------
    if (wv == 0 && kl < BATCH) {
      const int li = q * BATCH + kl;   // 0..15
      double *fin = w.M;               // the DK.. arrays are dead now (every wavefront passed the last barrier)
      fin[0 * 16 + li] = fi0;
      fin[1 * 16 + li] = fi1;
      fin[2 * 16 + li] = fi2;
      fin[3 * 16 + li] = v0;
      fin[4 * 16 + li] = v1;
      fin[5 * 16 + li] = v2;
      fin[6 * 16 + li] = v3;
      fin[7 * 16 + li] = v4;
      fin[8 * 16 + li] = v5;
    }
    __syncthreads();
    if (wv == 0) {
      if (lane < 9) {
        const double *r = w.M + 16 * lane;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 4) s += (r[u] + r[u + 1]) + (r[u + 2] + r[u + 3]);
        if (lane < 3) {
          unsafeAtomicAdd(&p.f[3 * (size_t) i + lane], s);   // pair_mtp.cpp:248-250
        } else if (p.vflag) {
          tally += s;
          if ((p.vflag & 4) && p.vatom) p.vatom[6 * (size_t) i + (lane - 3)] += s;
        }
      }
      if (lane == 9) {
        if ((p.eflag & 2) && p.eatom) p.eatom[i] = e;
        if (p.eflag & 1) tally += e;
      }
    }
    __syncthreads();   // the moment region is rewritten by the next atom
  }
  if (wv == 0 && lane >= 3 && lane <= 9 && tally != 0.0) {
    double *slot = p.ev_slots + 8 * (size_t) (blockIdx.x % MTP_EV_SLOTS);
    unsafeAtomicAdd(&slot[lane == 9 ? 0 : lane - 2], tally);
  }
}

template <int KL, int KBW> hipError_t launch_team(const MtpDevParams &p, int grid, size_t lds, hipStream_t st)
{
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mtp_team_kernel<KL, KBW, 4, 34>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((mtp_team_kernel<KL, KBW, 4, 34>), dim3(grid), dim3(256), lds, st, p);
  return hipGetLastError();
}

}   // namespace

// Team shape for B basics: KL k-lanes per neighbour group, KBW basics per lane per wavefront (4 wavefronts)
int mtp_pick_team_shape(int B, int *KL, int *KBW)
{
  for (int kl : {16, 32, 64})
    for (int kbw = 1; kbw <= 3; kbw++)
      if (B <= kl * kbw * 4) {
        *KL = kl;
        *KBW = kbw;
        return 0;
      }
  return -1;
}

hipError_t mtp_launch_team_kernel(const MtpDevParams &p, int grid, size_t lds, hipStream_t st)
{
  int KL = 0, KBW = 0;
  if (mtp_pick_team_shape(p.B, &KL, &KBW) != 0 || p.NT != 32 || p.grade_flag) return hipErrorInvalidValue;
#define MTP_TEAM_CASE(kl, kbw) \
  if (KL == kl && KBW == kbw) return launch_team<kl, kbw>(p, grid, lds, st);
  MTP_TEAM_CASE(16, 1)
  MTP_TEAM_CASE(16, 2)
  MTP_TEAM_CASE(16, 3)
  MTP_TEAM_CASE(32, 2)
  MTP_TEAM_CASE(32, 3)
  MTP_TEAM_CASE(64, 2)
  MTP_TEAM_CASE(64, 3)
#undef MTP_TEAM_CASE
  return hipErrorInvalidValue;
}
