// CDNA4 (gfx950) kernels of the MTP pair-style compute path.
//
// One 64-lane wavefront owns one atom (the native counterpart of both reference GPU
// styles, KOKKOS/pair_mtp_kokkos.cpp:404-660 and KOKKOS/pair_mtps_kokkos.cpp:438-708),
// and the per-pair Jacobian the reference spills to HBM
// (d_moment_jacobian[N][J][B][3], pair_mtp_kokkos.cpp:270-282) is never formed:
//
//   0. tables       every workgroup copies the read-mostly potential tables (packed times
//                   rows, level offsets, radial coefficients, slots, seeds) into the head
//                   of its LDS once; a persistent grid-stride loop over atoms follows
//   1. compaction   lanes = list entries: r^2 <= rc^2 test (pair_mtp.cpp:120-127),
//                   ballot/prefix -> in-cutoff neighbours in LDS
//   2. tile tables  per in-cutoff neighbour n, entry-major in LDS (row e, column n):
//                   g[mu,nu](n) = f_mu(r)/r^nu and dg/dr (Chebyshev recurrence + radial
//                   contraction, mtp_rb_chevbyshev_basis.cpp:29-54, pair_mtp.cpp:139-166)
//                   and the coordinate powers x^p (pair_mtp.cpp:133-136)
//   3. basic moments the wavefront is NG neighbour groups x KL block lanes: lane (q, kl) owns NB blocks of
//                   3 heads (g_s x^a) x 3 tails (y^b z^c) = 9 basics (tiled on the host) and the neighbours
//                   n = q + NG m, and accumulates M_k += g x^a y^b z^c in registers (pair_mtp.cpp:154-172)
//                   from 12 LDS reads per block and column; the table row pitch is a compile-time constant
//                   so every LDS read is base register + immediate; NG-way permlane-swap sum at the end
//   4. products     lanes = times rows, one dependency level at a time, moments and
//                   adjoints in LDS with ds_add_f64 (pair_mtp.cpp:196-233)
//   5. forces       per radial slot s = (mu, nu) the adjoints of its basics are the coefficients of a
//                   homogeneous polynomial P_s(x, y, z) = sum_k D_k x^a y^b z^c of degree nu, and
//                   F_ij = sum_s [ dg_s P_s r/|r| + g_s grad P_s ]  (pair_mtp.cpp:174-191, 236-246 summed
//                   over k), with P_s = (r . grad P_s) / nu (Euler).  Lanes = (neighbour, half): every
//                   lane raises the monomials of its neighbour degree by degree in registers and
//                   evaluates the derivative polynomials with coefficients broadcast from LDS (half 0:
//                   d/dx, half 1: d/dz, d/dy split between the halves by slot); 64 lanes then scatter
//                   f_j -= F_ij with fp64 HBM atomics and tally the virial (pair_mtp.cpp:248-277)
//
// Everything is fp64 (the reference's F_FLOAT); indices are int32.
#include <hip/hip_runtime.h>

#include "mtp_device.hpp"

#include "mtp_kernel_common.hpp"

namespace {

#ifndef MTP_PU
#define MTP_PU 2   // times rows in flight per lane in the row-per-lane product passes (4 until the leaf moments and the
                   // per-tile totals changed the balance: re-measured 1 / 2 / 3 / 4 / 5 / 6 rows: 0.473 / 0.473 / 0.477 / 0.482 /
                   // 0.486 / 0.496 ms at level 16)
#endif

// The parameter block is read through the kernarg segment pointer (address space 4: scalar loads from the constant
// cache that the compiler re-issues where a field is needed) instead of a by-value struct, which it kept in SGPRs
// across the whole atom loop and spilled into VGPR lanes (a quarter of the static VALU instructions were
// v_readlane / v_writelane, 255 VGPRs; now 229 and none).  Making the pointer opaque again at every phase boundary
// was measured 1 % slower.
typedef const __attribute__((address_space(4))) MtpDevParams *KP;

// f[idx] += v.  Default: native fp64 HBM atomics (the sum depends on the arrival order in the last bits).
// Deterministic mode (mtp_context_set_deterministic, tests / reproducible goldens): the contributions are added as
// 64-bit fixed-point integers (2^-40 eV/A resolution, |f| < 2^23), which commute exactly, and converted once at the end.
#define MTP_FIXED_SCALE 1099511627776.0   // 2^40
static __device__ __forceinline__ void force_add(KP kp, size_t idx, double v)
{
  if (kp->fq) atomicAdd(reinterpret_cast<unsigned long long *>(kp->fq) + idx, (unsigned long long) __double2ll_rn(v * MTP_FIXED_SCALE));
  else unsafeAtomicAdd(&kp->f[idx], v);
}

template <int PITCH> struct WaveLds {
  static constexpr int NT = 32;   // neighbours per tile; the row pitch (33 doubles: odd, so the 32 lanes of a half-wavefront
                                  // reading 32 different rows of one column hit 32 different 8-byte banks) is the template parameter
  double *M, *D, *coef, *tab, *nbx, *nby, *nbz, *nbr, *nbi;
  int *nbj, *nbjt, *cj;
  unsigned m_addr;   // LDS byte address of M
  __device__ __forceinline__ unsigned addr(const double *ptr) const { return m_addr + 8u * (unsigned) (ptr - M); }
  __device__ __forceinline__ WaveLds(double *base, unsigned base_addr, KP kp)
  {
    // Layouts of the per-atom image, chosen by the host planner (mtp_context.hip, plan()); all of them are
    // [tables | overlay | neighbour arrays] with the regions placed through offsets in the parameter block
    // (dg_mode: bit 0 = no dg rows, bit 1 = rebuild):
    //   keep     [g rows | dg rows | overlay]: the coordinate-power rows live in the overlay from the tile build to the
    //            end of the basic-moment pass, the moments / adjoints (later the derivative-polynomial coefficients)
    //            from there on -- the two are never live together;
    //   nodg     [g rows | overlay] (Mu <= 4): no dg rows.  d/dr (f_mu / r^nu) = f'_mu / r^nu - nu g / r, so
    //            sum_s (dg_s / nu) G_s = sum_s f'_mu(s) (r^-nu / nu) G_s - (1/r) sum_s g_s G_s: the tile build parks the
    //            radial derivatives f'_mu(r) of its neighbour in registers, Mu rows of them (instead of one dg row per
    //            slot) are written behind the coefficient blocks ahead of the force phase, which multiplies them in
    //            per slot and subtracts the g sums it forms anyway;
    //   rebuild  everything overlays everything (potentials with many moments): the moments and adjoints sit on the
    //            g rows, which are built a second time (with the dg rows unless nodg) ahead of the force phase (the
    //            coefficient blocks sit behind the rows, the adjoints of the basics D[0, B) in front: the host checks
    //            that they cannot meet).
    tab = base;
    M = tab + kp->w_m;
    D = tab + kp->w_d;
    coef = tab + kp->w_coef;
    nbx = tab + kp->w_nb;
    m_addr = base_addr + 8u * (unsigned) (M - tab);
    nby = nbx + NT;
    nbz = nby + NT;
    nbr = nbz + NT;
    nbi = nbr + NT;
    nbj = reinterpret_cast<int *>(nbi + NT);
    nbjt = nbj + NT;
    cj = nbjt + NT;
  }
};

// Phase 2: tables of one tile; columns [0, ntp) are written, ntp = nt rounded up to the
// neighbour-group count with dummy neighbours sitting exactly on the cutoff (g = dg = 0).
// with_dg: write the dg rows; do_park (nodg layouts): park[] receives f'_mu(r) of this lane's neighbour for its radial
// functions mu = h, h + 2 (park[0..1]), from which fp_from_parked() writes the f' rows ahead of the force phase.
#define MTP_PARK 2   // radial functions per half-wavefront that can be parked: Mu <= 4
template <int PITCH>
__device__ __forceinline__ void build_tile(KP kp, const BlockTables &bt, const WaveLds<PITCH> &w,
                                           int t0, int cnt, int ntp, bool gather, bool powers, bool with_dg, bool do_park,
                                           double (&park)[MTP_PARK],
                                           double xi0, double xi1, double xi2, int i, int itype, int lane)
{
  if (gather) {
    if (lane < ntp) {
      const bool real = t0 + lane < cnt;
      const int j = real ? w.cj[t0 + lane] : i;
      double dx = 0, dy = 0, dz = 0, r = kp->rmax, inv = kp->inv_rmax;
      if (real) {
        const double *xj = row3(kp->x, j);
        dx = xj[0] - xi0;
        dy = xj[1] - xi1;
        dz = xj[2] - xi2;
        sqrt_and_inverse(dx * dx + dy * dy + dz * dz, r, inv);
      }
      w.nbx[lane] = dx;
      w.nby[lane] = dy;
      w.nbz[lane] = dz;
      w.nbr[lane] = r;
      w.nbi[lane] = inv;
      w.nbj[lane] = j;
      w.nbjt[lane] = real ? kp->type[j] - 1 : itype;
    }
    wave_fence();
  }
  // lanes = (neighbour n, half h): one pass over the tile.  The Chebyshev values q_k(r) and derivatives are
  // shared by all radial functions of the neighbour; half h contracts them for mu = h, h+2, ... and writes
  // the g / dg rows of those mu; the coordinate-power rows are split x,y | z between the halves.
  const int Mu = kp->Mu, P = kp->P, R = kp->R;
  const double mult = 2.0 * kp->inv_span;
  const int n = lane & 31, h = lane >> 5;
  if (n < ntp) {
    const double r = w.nbr[n], inv = w.nbi[n];
    const int jt = w.nbjt[n];
    const double d = r - kp->rmax;
    const double ksi = (2.0 * r - (kp->rmin + kp->rmax)) * kp->inv_span;
    double *col = w.tab + n;
    // radial functions of this half: mu = h, h + 2, ...; the first MTP_PARK of them with static indices, so that the
    // parked derivatives stay in registers (a loop-carried index would put the array into scratch memory)
    auto each_mu = [&](auto &&body) {
      double d0 = 0.0, d1 = 0.0;
      if (h < Mu) d0 = body(h);
      if (h + 2 < Mu) d1 = body(h + 2);
      for (int mu = h + 2 * MTP_PARK; mu < Mu; mu += 2) (void) body(mu);
      if (do_park) {
        park[0] = d0;
        park[1] = d1;
      }
    };
    if (R == 8) {   // the MLIP default: basis in registers, coefficients in bursts of 16-byte reads
      double qv[8], ev[8];
      qv[0] = kp->scaling * (d * d);
      qv[1] = kp->scaling * (ksi * d * d);
      ev[0] = kp->scaling * 2.0 * d;
      ev[1] = kp->scaling * (mult * d * d + 2.0 * ksi * d);
#pragma unroll
      for (int ri = 2; ri < 8; ri++) {   // mtp_rb_chevbyshev_basis.cpp:29-54
        qv[ri] = 2.0 * ksi * qv[ri - 1] - qv[ri - 2];
        ev[ri] = 2.0 * (mult * qv[ri - 1] + ksi * ev[ri - 1]) - ev[ri - 2];
      }
      each_mu([&](int mu) {
        const int4 *sl4 = reinterpret_cast<const int4 *>(bt.slot + mu * MTP_PSTRIDE);
        const int4 sa = sl4[0], sb = sl4[1], sc = sl4[2];
        const int sv[MTP_PSTRIDE] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w, sc.x, sc.y, sc.z, sc.w};
        const double2 *c2 = reinterpret_cast<const double2 *>(bt.radial + (mul24(itype * kp->Sp + jt, Mu) + mu) * 8);
        const double2 c01 = c2[0], c23 = c2[1], c45 = c2[2], c67 = c2[3];
        const double cc[8] = {c01.x, c01.y, c23.x, c23.y, c45.x, c45.y, c67.x, c67.y};
        double val = cc[0] * qv[0], der = cc[0] * ev[0];
#pragma unroll
        for (int ri = 1; ri < 8; ri++) {
          val = fma(cc[ri], qv[ri], val);
          der = fma(cc[ri], ev[ri], der);
        }
        double rp = 1.0;
#pragma unroll
        for (int nu = 0; nu < MTP_PSTRIDE; nu++) {
          if (nu < P) {
            const int sidx = sv[nu];
            const double g = val * rp;
            if (sidx >= 0) {
              double *gp = col + mul24(sidx, PITCH);
              *gp = g;                                                       // f_mu / r^nu
              if (with_dg) gp[kp->dg_off] = der * rp - nu * g * inv;         // d/dr (f_mu / r^nu)
            }
            rp *= inv;
          }
        }
        return der;
      });
    } else {
      each_mu([&](int mu) {
        const int *sl = bt.slot + mu * MTP_PSTRIDE;
        const double *c = bt.radial + mul24(mul24(itype * kp->Sp + jt, Mu) + mu, R);
        double q0 = kp->scaling * (d * d), q1 = kp->scaling * (ksi * d * d);
        double e0 = kp->scaling * 2.0 * d, e1 = kp->scaling * (mult * d * d + 2.0 * ksi * d);
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
        double rp = 1.0;
        for (int nu = 0; nu < P; nu++) {
          const int sidx = sl[nu];
          const double g = val * rp;
          if (sidx >= 0) {
            double *gp = col + mul24(sidx, PITCH);
            *gp = g;
            if (with_dg) gp[kp->dg_off] = der * rp - nu * g * inv;
          }
          rp *= inv;
        }
        return der;
      });
    }
    if (powers) {   // rows of one axis: [q] = u^q
      const double u0 = h == 0 ? w.nbx[n] : w.nbz[n];
      double *pc = col + mul24(kp->pow_row + (h == 0 ? 0 : 2 * P), PITCH);
      double cur = 1.0;
      pc[0] = 1.0;
      for (int q = 1; q < P; q++) {
        cur *= u0;
        pc[q * PITCH] = cur;
      }
      if (h == 0) {
        const double u1 = w.nby[n];
        pc += mul24(P, PITCH);
        cur = 1.0;
        pc[0] = 1.0;
        for (int q = 1; q < P; q++) {
          cur *= u1;
          pc[q * PITCH] = cur;
        }
      }
    }
  }
  wave_fence();
}

// nodg layouts, ahead of the force phase: the Mu rows f'_mu(r_n) of the tile from the derivatives the tile build parked
// in registers (row fp_row + mu; two stores per lane instead of one dg row per slot)
template <int PITCH>
__device__ __forceinline__ void fp_from_parked(KP kp, const WaveLds<PITCH> &w, int ntp, const double (&park)[MTP_PARK], int lane)
{
  const int n = lane & 31, h = lane >> 5, Mu = kp->Mu;
  if (n < ntp) {
    double *col = w.tab + n + (size_t) kp->fp_row * PITCH;
#pragma unroll
    for (int mi = 0; mi < MTP_PARK; mi++) {
      const int mu = 2 * mi + h;
      if (mu < Mu) col[mul24(mu, PITCH)] = park[mi];
    }
  }
  wave_fence();
}

// Packed rows carry BYTE offsets (8 x moment index) in their 16-bit fields, so that an LDS address is one
// v_add_u32_sdwa (base + 16-bit word of the row) instead of a bit-field extract and a shift-add.
static __device__ __forceinline__ double &at8(double *base, unsigned byte_off)
{
  return *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off);
}
static __device__ __forceinline__ const double &at8(const double *base, unsigned byte_off)
{
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}

// Phase 4a: M[a3] += mult * M[a0] * M[a1], one dependency level at a time.  Rows of one level
// never write an operand of the same level, so four rows per lane are in flight before their
// ds_add_f64 issue.  (Two call sites, LDS-resident and HBM-resident rows: a select between the two
// pointers would go through a generic pointer, which hipcc 7.2 miscompiles on gfx950.)
template <int U>
__device__ __forceinline__ void products_forward(const MtpRow8 *rows, const int *level, int nlevels, double *M,
                                                 int lane)
{
  for (int l = 0; l < nlevels; l++) {
    // levels are padded to whole 64-row blocks on the host (neutral rows): no bounds checks, no lane masks
    const int beg = __builtin_amdgcn_readfirstlane(level[l]);
    const int nit = (__builtin_amdgcn_readfirstlane(level[l + 1]) - beg) >> 6;
    const MtpRow8 *rp = rows + beg + lane;
    for (int it = 0; it < nit; it += U) {
      MtpRow8 rw[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) rw[u] = rp[64 * min(it + u, nit - 1)];   // uniform clamp: the tail re-reads the last block
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = at8(M, rw[u].lo & 0xffffu) * at8(M, rw[u].lo >> 16);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (it + u < nit) lds_add(&at8(M, rw[u].hi & 0xffffu), (double) ((int) rw[u].hi >> 16) * v[u]);   // uniform branch
    }
    wave_fence();
  }
}

// Phase 4b: D[a1] += D[a3] mult M[a0]; D[a0] += D[a3] mult M[a1], levels in reverse.
template <int U>
__device__ __forceinline__ void products_backward(const MtpRow8 *rows, const int *level, int nlevels,
                                                  const double *M, double *D, int lane)
{
  for (int l = nlevels - 1; l >= 0; l--) {
    const int beg = __builtin_amdgcn_readfirstlane(level[l]);
    const int nit = (__builtin_amdgcn_readfirstlane(level[l + 1]) - beg) >> 6;
    const MtpRow8 *rp = rows + beg + lane;
    for (int it = 0; it < nit; it += U) {
      MtpRow8 rw[U];
      double d3[U], m0[U], m1[U];
#pragma unroll
      for (int u = 0; u < U; u++) rw[u] = rp[64 * min(it + u, nit - 1)];
#pragma unroll
      for (int u = 0; u < U; u++) {
        d3[u] = at8(D, rw[u].hi & 0xffffu) * (double) ((int) rw[u].hi >> 16);
        m0[u] = at8(M, rw[u].lo & 0xffffu);
        m1[u] = at8(M, rw[u].lo >> 16);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (it + u < nit) {
          lds_add(&at8(D, rw[u].lo >> 16), d3[u] * m0[u]);
          lds_add(&at8(D, rw[u].lo & 0xffffu), d3[u] * m1[u]);
        }
    }
    wave_fence();
  }
}

// Leaf rows (mtp_potential.hpp: products that no row reads, i.e. scalars of the basis; pair_mtp.cpp:204-233).  Their
// moments have no LDS slot in force calls.  Forward: the row's product goes straight into the site energy,
// e += cf M[a0] M[a1] with cf = linear coefficient x mult (grade calls also keep M[a3] += mult M[a0] M[a1]: the
// candidate vector lists the leaves' values).  Reverse: D[a0] += cb M[a1], D[a1] += cb M[a0] with the constant adjoint
// cb = seed(a3) x mult -- no D[a3] read.  Row per lane as above; the constants are lane-contiguous like the rows.
// FAR: rows and constants come from HBM / L2 (wide lane grids, whose rows do not fit in LDS): batches of U rows are
// requested MTP_LD batches ahead of their use, as in the gather passes; otherwise both sit in the LDS blob.
#ifndef MTP_LD
#define MTP_LD 1   // (2: equal, 4: 2 % slower at level 20)
#endif
template <int U, bool STORE, bool FAR>
__device__ __forceinline__ double leaf_forward(const MtpRow8 *rows, const double *cf, int beg, int nit, double *M, int lane)
{
  constexpr int D = FAR ? MTP_LD : 1;
  double e = 0.0;
  const MtpRow8 *rp = rows + beg + lane;
  const double *cp = cf + lane;
  const int nb = (nit + U - 1) / U;
  MtpRow8 q[D][U];
  double qc[D][U];
  auto fetch = [&](int b, MtpRow8 (&r)[U], double (&c)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int o = 64 * min(b * U + u, nit - 1);   // uniform clamp: the tail re-reads the last block
      r[u] = rp[o];
      c[u] = cp[o];
    }
  };
  if (FAR) {
#pragma unroll
    for (int d = 0; d < D; d++)
      if (d < nb) fetch(d, q[d], qc[d]);   // uniform
  }
  for (int b0 = 0; b0 < nb; b0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int b = b0 + d;
      if (b < nb) {   // uniform
        MtpRow8 rw[U];
        double c[U], v[U];
        if (FAR) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            rw[u] = q[d][u];
            c[u] = qc[d][u];
          }
          if (b + D < nb) fetch(b + D, q[d], qc[d]);   // uniform
        } else {
          fetch(b, rw, c);
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = at8(M, rw[u].lo & 0xffffu) * at8(M, rw[u].lo >> 16);
#pragma unroll
        for (int u = 0; u < U; u++)
          if (b * U + u < nit) {   // uniform branch
            e = fma(c[u], v[u], e);
            if (STORE) lds_add(&at8(M, rw[u].hi & 0xffffu), (double) ((int) rw[u].hi >> 16) * v[u]);
          }
      }
    }
  }
  if (STORE) wave_fence();
  return e;
}

template <int U, bool FAR>
__device__ __forceinline__ void leaf_backward(const MtpRow8 *rows, const double *cb, int beg, int nit, const double *M,
                                              double *D_, int lane)
{
  constexpr int D = FAR ? MTP_LD : 1;
  const MtpRow8 *rp = rows + beg + lane;
  const double *cp = cb + lane;
  const int nb = (nit + U - 1) / U;
  MtpRow8 q[D][U];
  double qc[D][U];
  auto fetch = [&](int b, MtpRow8 (&r)[U], double (&c)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int o = 64 * min(b * U + u, nit - 1);
      r[u] = rp[o];
      c[u] = cp[o];
    }
  };
  if (FAR) {
#pragma unroll
    for (int d = 0; d < D; d++)
      if (d < nb) fetch(d, q[d], qc[d]);
  }
  for (int b0 = 0; b0 < nb; b0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int b = b0 + d;
      if (b < nb) {
        MtpRow8 rw[U];
        double c[U], m0[U], m1[U];
        if (FAR) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            rw[u] = q[d][u];
            c[u] = qc[d][u];
          }
          if (b + D < nb) fetch(b + D, q[d], qc[d]);
        } else {
          fetch(b, rw, c);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          m0[u] = at8(M, rw[u].lo & 0xffffu);
          m1[u] = at8(M, rw[u].lo >> 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (b * U + u < nit) {
            lds_add(&at8(D_, rw[u].lo >> 16), c[u] * m0[u]);
            lds_add(&at8(D_, rw[u].lo & 0xffffu), c[u] * m1[u]);
          }
      }
    }
  }
  wave_fence();
}

// Phase 4, gather form (round 2).  A level of a pass is a list of chunks; lane l of a group of 64 lanes runs one chunk:
// acc = sum_u mult_u X[o0_u] Y[o1_u] over its CS operations, then ONE atomic add T[tgt] += acc.  Forward: X = Y = T =
// moments (rows of one target); reverse: X = adjoints, Y = moments, T = adjoints (the terms of one destination), so a
// reverse level issues one ds_add_f64 per CS terms instead of two per row.  Operations (8 bytes, lane-contiguous) come
// from HBM / L2; those of the next trip are requested before the current trip's operands are read (a ring of 2, 4 or
// 8 trips in flight was measured slower at level 20: 1.46 / 1.50 / 1.60 against 1.40 ms).
template <int CS>
__device__ __forceinline__ void gather_groups(const MtpRow8 *rp, int ngroups, const double *X, const double *Y, double *T)
{
  constexpr int U = CS >= 4 ? 4 : CS;        // operations of one chunk in flight
  constexpr int G = CS >= 4 ? 1 : 4 / CS;    // chunks in flight
  constexpr int PARTS = CS / U;              // trips per chunk (CS = 8: two)
  const int ntrip = ((ngroups + G - 1) / G) * PARTS;
  auto fetch = [&](int trip, MtpRow8 (&dst)[G][U]) {
    const int g0 = (trip / PARTS) * G, part = trip % PARTS;
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int u = 0; u < U; u++) dst[j][u] = rp[64 * (min(g0 + j, ngroups - 1) * CS + part * U + u)];
  };
  double acc[G];
#pragma unroll
  for (int j = 0; j < G; j++) acc[j] = 0.0;
  MtpRow8 nxt[G][U];
  fetch(0, nxt);
  for (int trip = 0; trip < ntrip; trip++) {
    MtpRow8 cur[G][U];
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int u = 0; u < U; u++) cur[j][u] = nxt[j][u];
    if (trip + 1 < ntrip) fetch(trip + 1, nxt);   // uniform
    double xv[G][U], yv[G][U];
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        xv[j][u] = at8(X, cur[j][u].lo & 0xffffu);
        yv[j][u] = at8(Y, cur[j][u].lo >> 16);
      }
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int u = 0; u < U; u++) acc[j] = fma((double) ((int) cur[j][u].hi >> 16) * xv[j][u], yv[j][u], acc[j]);
    if (trip % PARTS == PARTS - 1) {
      const int g0 = (trip / PARTS) * G;
#pragma unroll
      for (int j = 0; j < G; j++) {
        if (g0 + j < ngroups) lds_add(&at8(T, cur[j][0].hi & 0xffffu), acc[j]);   // uniform branch
        acc[j] = 0.0;
      }
    }
  }
}

// one pass: the levels in the order the segment table lists them (forward: ascending, reverse: descending)
__device__ __forceinline__ void gather_pass(const MtpRow8 *prog, const int *seg, int nlevels, const double *X,
                                            const double *Y, double *T, int lane)
{
  for (int l = 0; l < nlevels; l++) {
    const int first = __builtin_amdgcn_readfirstlane(seg[4 * l]), ngroups = __builtin_amdgcn_readfirstlane(seg[4 * l + 1]);
    const int cs = __builtin_amdgcn_readfirstlane(seg[4 * l + 2]);
    const MtpRow8 *rp = prog + (size_t) first * 64 + lane;
    if (cs == 4) gather_groups<4>(rp, ngroups, X, Y, T);
    else if (cs == 8) gather_groups<8>(rp, ngroups, X, Y, T);
    else if (cs == 2) gather_groups<2>(rp, ngroups, X, Y, T);
    else gather_groups<1>(rp, ngroups, X, Y, T);
    wave_fence();
  }
}

// ---- phase 5 helpers ------------------------------------------------------------------------
// sum_{i<C} coef[i] * m[i]; the coefficient address is the same in all lanes of a half (LDS broadcast)
#ifndef MTP_POLY_ACC
#define MTP_POLY_ACC 1   // independent accumulation chains of a derivative polynomial (1 | 2 | 4 measured at 65,536 atoms: 0.4367 | 0.4388 | 0.4406 ms; no difference at 2,048)
#endif
template <int C> __device__ __forceinline__ double poly_eval(unsigned coef, const double *m)
{
  constexpr int NA = MTP_POLY_ACC;
  double a[NA];
#pragma unroll
  for (int k = 0; k < NA; k++) a[k] = 0.0;
#ifndef MTP_POLY_CH
#define MTP_POLY_CH 8
#endif
  constexpr int CH = MTP_POLY_CH;   // reads per burst
#pragma unroll
  for (int i0 = 0; i0 < C; i0 += CH) {
    double c[CH];
#pragma unroll
    for (int u = 0; u < CH; u++)
      if (i0 + u < C) c[u] = lds_ld(coef, i0 + u);
#pragma unroll
    for (int u = 0; u < CH; u++)
      if (i0 + u < C) a[(i0 + u) % NA] = fma(c[u], m[i0 + u], a[(i0 + u) % NA]);
  }
  double r = a[0];
#pragma unroll
  for (int k = 1; k < NA; k++) r += a[k];
  return r;
}

// Slots of tensor rank NU: m[] holds the monomials of degree NU-1 of this lane's neighbour, ordered
// (a descending, then b descending): idx(a, b, c) = j (j + 1) / 2 + c with j = b + c.  A slot's coefficient
// block is [d/dx | d/dy | d/dz], each over those monomials.  UA/VA collect sum_s g_s dP_s/dx (half 0) or
// dP_s/dz (half 1) and the same with dg_s / nu; UB/VB the d/dy terms of the slots this half owns.
// NODG: no dg rows -- VA / VB collect sum_s f'_mu(s) (r^-nu / nu) G_s instead (f'_mu(r) of this lane's neighbour from
// row fp_row + mu of the tile, rw = r^-NU on entry) and the caller subtracts (UA, UB) / r at the end:
// dg_s = f'_mu r^-nu - nu g_s / r.
// GRADE (fused candidate vectors): W[mu] collects this lane's share of W_mu(n) = sum_{s in mu} P_s(r_n) / r_n^nu
// (pair_mtp_extrapolation.cpp:193-198), again through P_s = (r . grad P_s) / nu.
template <int NU, int DEG, int PITCH, bool GRADE, bool NODG>
__device__ __forceinline__ void force_degree(KP kp, unsigned pcol, unsigned pcoef, int part, double x,
                                             double y, double z, double *m, double &UA, double &VA, double &UB,
                                             double &VB, const int *smu, double inv, double rw, double *W)
{
  if constexpr (NU <= DEG) {
    constexpr int C = NU * (NU + 1) / 2;   // monomials of degree NU-1
    if (NU < kp->P) {
      const int s0 = kp->deg_first[NU], cnt = kp->deg_first[NU + 1] - s0;
      const double inv_nu = 1.0 / NU;
      const unsigned dgo = 8u * (unsigned) kp->dg_off;
      const unsigned pfp = pcol + 8u * (unsigned) (kp->fp_row * PITCH);   // f' rows of this lane's column (NODG)
      const double rwn = rw * inv_nu;
      const double wa = GRADE ? (part ? z : x) * rwn : 0.0, wb = GRADE ? y * rwn : 0.0;
      {
        unsigned ca = pcoef + 8u * (unsigned) (kp->deg_coef[NU] + part * 2 * C);
        unsigned cg = pcol + 8u * (unsigned) (s0 * PITCH);
        for (int it = 0; it < cnt; it++) {
          // the slot, hence mu, is wave-uniform in this pass
          const int mu = (NODG || GRADE) ? __builtin_amdgcn_readfirstlane(smu[s0 + it]) : 0;
          const double g = lds_ld(cg, 0);
          const double dg = NODG ? lds_ld(pfp + 8u * (unsigned) (mu * PITCH), 0) : lds_ld(cg + dgo, 0);   // NODG: f'_mu (mu: SGPR)
          const double G = poly_eval<C>(ca, m);
          UA = fma(g, G, UA);
          VA = fma(dg * (NODG ? rwn : inv_nu), G, VA);
          if (GRADE) {
            const double val = G * wa;
            if (mu == 0) W[0] += val;
            else if (mu == 1) W[1] += val;
            else if (mu == 2) W[2] += val;
            else W[3] += val;
          }
          ca += 8u * 3 * C;
          cg += 8u * PITCH;
        }
      }
      for (int it = 0; 2 * it < cnt; it++) {
        const int si = 2 * it + part;
        const bool ok = si < cnt;
        const int sc = ok ? si : 0;
        // (per-lane sc: 24-bit multiplies are full rate, 32-bit ones a quarter of it)
        const unsigned cb = pcoef + 8u * (unsigned) (kp->deg_coef[NU] + C) + (unsigned) mul24(sc, 8 * 3 * C);
        const unsigned cg = pcol + (unsigned) mul24(s0 + sc, 8 * PITCH);
        // (the halves hold different slots, hence different mu: a per-lane value here)
        const int mu_raw = (NODG || GRADE) ? smu[s0 + sc] : 0;
        const double g_raw = lds_ld(cg, 0);
        const double dg_raw = NODG ? lds_ld(pfp + (unsigned) mul24(mu_raw, 8 * PITCH), 0) : lds_ld(cg + dgo, 0);
        const double G = poly_eval<C>(cb, m);
        const double g = ok ? g_raw : 0.0, dg = ok ? dg_raw : 0.0;
        UB = fma(g, G, UB);
        VB = fma(dg * (NODG ? rwn : inv_nu), G, VB);
        const int mu = ok ? mu_raw : -1;
        if (GRADE) {
          const double val = G * wb;
#pragma unroll
          for (int v = 0; v < 4; v++) W[v] += mu == v ? val : 0.0;
        }
      }
      if constexpr (NU < DEG) {
        // raise the monomials to degree NU: the new a = 0 tail from the old one, then the head times x
        constexpr int T0 = (NU - 1) * NU / 2;
#pragma unroll
        for (int c = 0; c < NU; c++) m[C + c] = y * m[T0 + c];
        m[C + NU] = z * m[T0 + NU - 1];
#pragma unroll
        for (int i = 0; i < C; i++) m[i] *= x;
        force_degree<NU + 1, DEG, PITCH, GRADE, NODG>(kp, pcol, pcoef, part, x, y, z, m, UA, VA, UB, VB, smu, inv, rw * inv, W);
      }
    }
  }
}

// WPS = wavefronts per SIMD the register budget is sized for: 2 (<= 256 VGPRs, workgroups of up to 8 wavefronts) or
// 3 (<= 168 VGPRs, workgroups of up to 12: one workgroup per CU puts three wavefronts on every SIMD)
template <int KL, int NB, int PITCH, bool GRADE, int DEG, int WPS>
__global__ void __launch_bounds__(WPS == 3 ? 768 : 512, WPS) mtp_wave_kernel(const MtpDevParams p_arg)
{
  constexpr int NT = 32;                 // neighbours per tile
  // Product passes: the wide lane grids (KL = 64: level 18 and up, thousands of times rows that live in HBM / L2 either
  // way) run the gather programs -- measured at level 20: 2.02 -> 1.93 ms; the narrow grids keep the row-per-lane passes
  // with the rows in LDS -- at level 16 the gather programs (27 KB, so in L2) were 2.3 % slower (0.523 vs 0.511 ms).
  constexpr bool GATHER = KL == 64;
  // The 3-per-SIMD build is planned with the dg-free layouts only (its table shapes have Mu <= 4), so the dg paths are
  // compiled out of it; the 2-per-SIMD build takes either (uniform flag).
  constexpr bool NODG_CT = WPS == 3;
  constexpr int NG = 64 / KL;            // neighbour groups in the wavefront
  constexpr int NPG = NT / NG;           // neighbours per group per tile
  static_assert(NT == 32, "the force phase maps lanes to (32 neighbours) x (2 halves)");

  (void) p_arg;   // the only kernel argument: it starts the kernarg segment
  KP kp = (KP) __builtin_amdgcn_kernarg_segment_ptr();
#ifdef MTP_STAMPS
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
  kernarg_touch<(int) sizeof(MtpDevParams)>(kp);
  extern __shared__ double lds[];
  unsigned char *sh = reinterpret_cast<unsigned char *>(lds);
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction: tell the compiler, so per-atom state lives in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;

  // XCD-aware atom map: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b works
  // for XCD b % 8; giving every XCD one contiguous eighth of ilist (callers keep atoms roughly in spatial order:
  // LAMMPS sorts them, the bench lattice is cell-major) keeps the position gathers and the force atomics of a slab in
  // ONE L2 instead of spreading every slab over all eight.
  int ii_beg, ii_end, ii_step;
  if (kp->xcd_map && (gridDim.x & 7) == 0) {
    const int chunk = (kp->inum + 7) >> 3, xcd = blockIdx.x & 7;
    // Rounds of (workgroups x wpb) atoms: when the last round is only partly filled, the wavefronts are numbered
    // wave-major, so that its atoms land on a few wavefronts of EVERY workgroup instead of on all wavefronts of a few
    // (65 536 atoms over 3 072 wavefronts = 21.33 rounds: -2.4 %); with whole rounds the block-major numbering keeps
    // neighbouring atoms on one CU (level 20, 32 rounds exactly: 0.3 % better).
    const int nb8 = gridDim.x >> 3;
    if (chunk % (nb8 * wpb) != 0) ii_beg = kp->row0 + xcd * chunk + wave * nb8 + (blockIdx.x >> 3);
    else ii_beg = kp->row0 + xcd * chunk + (blockIdx.x >> 3) * wpb + wave;
    ii_end = kp->row0 + min(kp->inum, (xcd + 1) * chunk);
    ii_step = (gridDim.x >> 3) * wpb;
  } else {
    ii_step = gridDim.x * wpb;
    if (kp->inum % ii_step != 0) ii_beg = kp->row0 + wave * gridDim.x + blockIdx.x;
    else ii_beg = kp->row0 + blockIdx.x * wpb + wave;
    ii_end = kp->row0 + kp->inum;
  }
  // The head of an atom's list row {ilist, first} is requested one atom ahead and carried in SGPRs: two dependent memory
  // round trips per atom instead of three, and the first atom's ride on the table copy below (what a 2,048-atom call
  // is made of: one atom per wavefront, every load a miss).
  int hd_i = 0, hd_b = 0, hd_e = 0;
  if (ii_beg < ii_end) {   // (uniform)
    hd_i = kp->ilist[ii_beg];
    hd_b = kp->first[ii_beg];
    hd_e = kp->first[ii_beg + 1];
  }
  // ---- 0. workgroup-shared tables ---------------------------------------------------------
  for (int o = threadIdx.x * 16; o < kp->blob_bytes; o += blockDim.x * 16)
    *reinterpret_cast<uint4 *>(sh + o) = *reinterpret_cast<const uint4 *>(kp->blob + o);
  __syncthreads();
  BlockTables bt;
  bt.rows = reinterpret_cast<const MtpRow8 *>(sh + kp->off_rows);
  bt.level = reinterpret_cast<const int *>(sh + kp->off_level);
  bt.seg_fwd = reinterpret_cast<const int *>(sh + kp->off_seg_fwd);
  bt.seg_bwd = reinterpret_cast<const int *>(sh + kp->off_seg_bwd);
  bt.slot = reinterpret_cast<const int *>(sh + kp->off_slot);
  bt.radial = reinterpret_cast<const double *>(sh + kp->off_radial);
  bt.seed_idx = reinterpret_cast<const int *>(sh + kp->off_seed_idx);
  bt.seed_val = reinterpret_cast<const double *>(sh + kp->off_seed_val);
  bt.map = reinterpret_cast<const int *>(sh + kp->off_map);
  bt.lin = reinterpret_cast<const double *>(sh + kp->off_lin);
  bt.pack = reinterpret_cast<const int *>(sh + kp->off_pack);
  bt.coef = reinterpret_cast<const int *>(sh + kp->off_coef);
  bt.smu = reinterpret_cast<const int *>(sh + kp->off_smu);
  bt.fwd = reinterpret_cast<const int *>(sh + kp->off_fwd);
  bt.leaf_cf = reinterpret_cast<const double *>(sh + kp->off_leaf_cf);   // (behind the rows: valid when rows_in_lds)
  bt.leaf_cb = reinterpret_cast<const double *>(sh + kp->off_leaf_cb);
  const bool rows_lds = kp->rows_in_lds != 0;

  const int kl = lane & (KL - 1), q = lane / KL;
  const unsigned wave_off = (kp->blob_bytes >> 3) + wave * kp->wave_doubles;   // doubles
  const unsigned lds0 = (unsigned) (size_t) (lds_cdouble *) lds;            // static cast of the array itself
  const WaveLds<PITCH> w(lds + wave_off, lds0 + 8u * wave_off, kp);
  const int P = kp->P;

  // Basic-moment pass in 3 x 3 register blocks (built on the host, mtp_potential.cpp): lane (q, kl) owns the blocks
  // kl + KL t; a block is 3 heads (slot s, exponent a: head value g_s x^a) times 3 tails (b, c: tail value y^b z^c)
  // with b + c = nu_s - a for all of them, i.e. nine basics from twelve table rows.  Per block: LDS byte addresses of
  // the rows for this lane's neighbour column q.
  unsigned hg[NB][3], hx[NB][3], ty[NB][3], tz[NB][3];
  bool bval[NB];
  // (2-per-SIMD build: formed once per kernel; 3-per-SIMD build: once per atom, so that the twelve registers are free
  // outside the basic-moment pass)
  auto block_addresses = [&](int kl_) {
#pragma unroll
    for (int t = 0; t < NB; t++) {
      const int blk = kl_ + KL * t;
      bval[t] = blk < kp->nfb;
      const int *bd = bt.fwd + 8 * (bval[t] ? blk : 0);
      const unsigned w0 = (unsigned) bd[0], w1 = (unsigned) bd[1], w2 = (unsigned) bd[2];
#pragma unroll
      for (int h = 0; h < 3; h++) {
        const unsigned tq = w.addr(w.tab + q);
        hg[t][h] = tq + (unsigned) mul24((int) ((w0 >> (8 * h)) & 255u), 8 * PITCH);
        hx[t][h] = tq + (unsigned) mul24(kp->pow_row + (int) ((w1 >> (4 * h)) & 15u), 8 * PITCH);
        ty[t][h] = tq + (unsigned) mul24(kp->pow_row + P + (int) ((w1 >> (12 + 4 * h)) & 15u), 8 * PITCH);
        tz[t][h] = tq + (unsigned) mul24(kp->pow_row + 2 * P + (int) ((w2 >> (4 * h)) & 15u), 8 * PITCH);
        // one finished address per register: stops the optimiser from re-splitting them into
        // base + row offset (which costs a v_add per LDS read in the inner loops)
        asm volatile("" : "+v"(hg[t][h]), "+v"(hx[t][h]), "+v"(ty[t][h]), "+v"(tz[t][h]));
      }
    }
  };
  if constexpr (WPS != 3) block_addresses(kl);

  double tally = 0.0;   // lane 9: energy, lanes 3..8: virial components of this wave's atoms
  // Global-only tallies need no per-atom reduction: the per-lane partial sums of the virial (and of the energy) run
  // across the wavefront's atoms and cross the lanes once, after the atom loop.  Per-atom outputs (vatom: vflag & 4,
  // eatom: eflag & 2) keep the per-atom reductions.
  double vacc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, eacc = 0.0;
  // (Only in the 2-per-SIMD build: at 168 VGPRs the seven extra accumulators spill and cost more than the per-atom
  // reductions -- measured 0.514 against 0.500 ms.)
  const bool v_per_atom = WPS == 3 ? kp->vflag != 0 : (kp->vflag & 4) != 0;
  const bool e_per_atom = WPS == 3 ? true : (kp->eflag & 2) != 0;
#ifdef MTP_STAMPS
  unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long st_prologue = st_prev - st_entry;   // argument block, table copy, barrier, first list head
#endif

  int nx_i = __builtin_amdgcn_readfirstlane(hd_i), nx_b = __builtin_amdgcn_readfirstlane(hd_b);
  int nx_n = __builtin_amdgcn_readfirstlane(hd_e) - nx_b;
  for (int ii = ii_beg; ii < ii_end; ii += ii_step) {
    // ii is wave-uniform, so is everything loaded through it: keep it in SGPRs.  Two dependent memory round trips per
    // atom: {type_i, x_i, the row's first 128 neighbour ids; the NEXT atom's ilist, first} -> {x_j, type_j}: every load
    // of a stage is requested before the first wait, and before the type check branches.
    const int i = nx_i, jbeg = nx_b, jnum = nx_n;
    {
      const int iin = min(ii + ii_step, ii_end - 1);   // (the last atom asks for its own row again: no branch)
      hd_i = kp->ilist[iin];
      hd_b = kp->first[iin];
      hd_e = kp->first[iin + 1];
    }
    int jpre[2] = {0, 0};   // neighbour ids of the first chunk
    if (jnum > 0) {         // (uniform)
#pragma unroll
      for (int u = 0; u < 2; u++) jpre[u] = kp->neigh[jbeg + min(64 * u + lane, jnum - 1)];
    }
    const int itype_raw = kp->type[i];
    const double *xi_p = kp->x + 3 * (size_t) i;   // (i in SGPRs: scalar arithmetic)
    const double x0_raw = xi_p[0], x1_raw = xi_p[1], x2_raw = xi_p[2];
    asm volatile("" : "+v"(jpre[0]), "+v"(jpre[1]));   // (pins the first use of the ids behind the requests above)
    const int itype = __builtin_amdgcn_readfirstlane(itype_raw) - 1;
    const double xi0 = uniform_f64(x0_raw), xi1 = uniform_f64(x1_raw), xi2 = uniform_f64(x2_raw);
    nx_i = __builtin_amdgcn_readfirstlane(hd_i);   // (requested ahead of the loads above: here by now)
    nx_b = __builtin_amdgcn_readfirstlane(hd_b);
    nx_n = __builtin_amdgcn_readfirstlane(hd_e) - nx_b;
    if (itype < 0 || itype >= kp->Sp) {   // pair_mtp.cpp:91-93
      if (lane == 0) atomicExch(kp->err_flag, 1);
      continue;
    }

    STAMP(0);   // loop head: ilist/type/x/first loads issue
    // ---- 1. compaction (the first NT survivors go straight into the tile arrays) --------
    int cnt = 0;
    const int cj_last = kp->cj_cap - 1;
    for (int c0 = 0; c0 < jnum; c0 += 128) {
      // two list entries per lane; the loads of both are in flight together (clamped indices, no branches)
      int j2[2], jt2[2];
      double d2[2][3];
      bool ok2[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int jj = c0 + 64 * u + lane;
        ok2[u] = jj < jnum;
        j2[u] = (c0 == 0 ? jpre[u] : kp->neigh[jbeg + min(jj, jnum - 1)]) & MTP_NEIGHMASK;   // (uniform select)
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        jt2[u] = kp->type[j2[u]] - 1;
        const double *xj = row3(kp->x, j2[u]);
        d2[u][0] = xj[0];
        d2[u][1] = xj[1];
        d2[u][2] = xj[2];
      }
      // (no branch on the loaded values ahead of the arithmetic: the compiler otherwise sinks the position loads of the
      // first entry behind its type check -- one more dependent memory round trip per atom)
      bool bad = false;
#pragma unroll
      for (int u = 0; u < 2; u++) bad = bad || (ok2[u] && (jt2[u] < 0 || jt2[u] >= kp->Sp));
      if (__ballot(bad) != 0ull) {   // pair_mtp.cpp:116-118 (uniform, never taken with a valid type array)
        if (bad) atomicExch(kp->err_flag, 1);
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (u == 1 && c0 + 64 >= jnum) break;   // uniform
        const int j = j2[u], jt = jt2[u];
        const double dx = d2[u][0] - xi0, dy = d2[u][1] - xi1, dz = d2[u][2] - xi2;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const bool in = ok2[u] && jt >= 0 && jt < kp->Sp && !(r2 > kp->cutsq);   // pair_mtp.cpp:123
        const unsigned long long m = __ballot(in);
        if (in) {
          const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
          w.cj[min(pos, cj_last)] = j;   // a list longer than the declared max_numneigh is reported below
          if (pos < NT) {
            double r, inv;
            sqrt_and_inverse(r2, r, inv);
            w.nbx[pos] = dx;
            w.nby[pos] = dy;
            w.nbz[pos] = dz;
            w.nbr[pos] = r;
            w.nbi[pos] = inv;
            w.nbj[pos] = j;
            w.nbjt[pos] = jt;
          }
        }
        cnt += __builtin_amdgcn_readfirstlane(__popcll(m));
      }
    }
    if (cnt > kp->cj_cap) {   // the caller's max_numneigh sized the id array: refuse instead of overrunning LDS
      if (lane == 0) atomicExch(kp->err_flag, 2);
      cnt = kp->cj_cap;
    }
    {   // dummy neighbours pad tile 0 to a multiple of NG
      const int pos = cnt + lane;
      if (cnt < NT && lane < NG && pos < ((min(cnt, NT) + NG - 1) / NG) * NG) {
        w.nbx[pos] = 0.0;
        w.nby[pos] = 0.0;
        w.nbz[pos] = 0.0;
        w.nbr[pos] = kp->rmax;
        w.nbi[pos] = kp->inv_rmax;
        w.nbj[pos] = i;
        w.nbjt[pos] = itype;
      }
    }
    wave_fence();

    STAMP(1);   // compaction
    // ---- 2+3. tiles: tables, then basic moments in registers ------------------------------
    double acc[NB][9];
#pragma unroll
    for (int t = 0; t < NB; t++)
#pragma unroll
      for (int e = 0; e < 9; e++) acc[t][e] = 0.0;
    if constexpr (WPS == 3) {
      int kl_o = kl;
      asm volatile("" : "+v"(kl_o));   // opaque per atom: keeps the address arithmetic inside the loop
      block_addresses(kl_o);
    }
    const int ntiles = (cnt + NT - 1) / NT;
    const bool nodg = NODG_CT || (kp->dg_mode & 1) != 0, rebuild = (kp->dg_mode & 2) != 0;
    double park[MTP_PARK] = {0.0, 0.0};   // nodg layouts: f'_mu(r) of this lane's neighbour, mu = half, half + 2
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * NT, nt = min(NT, cnt - t0), ntp = ((nt + NG - 1) / NG) * NG;
      build_tile<PITCH>(kp, bt, w, t0, cnt, ntp, tile > 0, true, !nodg && !rebuild, nodg, park, xi0, xi1, xi2, i, itype, lane);
      STAMP(2);   // tile tables
#pragma unroll
      for (int m = 0; m < NPG; m++) {
        if (m * NG < ntp) {
          // the 12 reads of a block issue back to back (one LDS latency), then 6 products and 9 FMAs; the barriers
          // keep the scheduler from either splitting the burst or hoisting every column's reads (register blow-up)
#pragma unroll
          for (int t = 0; t < NB; t++) {
            double G[3], X[3], Y[3], Z[3];
#pragma unroll
            for (int h = 0; h < 3; h++) {
              G[h] = lds_ld(hg[t][h], m * NG);
              X[h] = lds_ld(hx[t][h], m * NG);
              Y[h] = lds_ld(ty[t][h], m * NG);
              Z[h] = lds_ld(tz[t][h], m * NG);
            }
            __builtin_amdgcn_sched_barrier(0);
            double hd[3], tl[3];
#pragma unroll
            for (int h = 0; h < 3; h++) {
              hd[h] = G[h] * X[h];
              tl[h] = Y[h] * Z[h];
            }
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
              for (int u = 0; u < 3; u++) acc[t][3 * h + u] = fma(hd[h], tl[u], acc[t][3 * h + u]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (ntiles > 1) wave_fence();
    }
    STAMP(3);   // basic moments
    // sum over the neighbour groups, then moments + adjoints into LDS
#pragma unroll
    for (int t = 0; t < NB; t++)
#pragma unroll
      for (int e = 0; e < 9; e++) {
        if (NG >= 4) acc[t][e] = pair_sum16(acc[t][e]);   // KL = 16: groups differ in lane bits 4 and 5
        if (NG >= 2) acc[t][e] = pair_sum32(acc[t][e]);
      }
    for (int m = kp->B + lane; m < kp->Am; m += 64) w.M[m] = 0.0;
    for (int m = lane; m < kp->Ad; m += 64) w.D[m] = 0.0;
    if (q == 0) {
#pragma unroll
      for (int t = 0; t < NB; t++)
        if (bval[t]) {
          // nine int16 basic indices of the block (-1: no such basic), 20 bytes after the three descriptor words
          const short *kk = reinterpret_cast<const short *>(bt.fwd + 8 * (kl + KL * t) + 3);
#pragma unroll
          for (int e = 0; e < 9; e++) {
            const int k = kk[e];
            if (k >= 0) w.M[k] = acc[t][e];
          }
        }
    }
    wave_fence();

    // ---- 4a. products, level by level (pair_mtp.cpp:196-201) -----------------------------
    if constexpr (GATHER) {
      gather_pass(kp->prog_fwd, bt.seg_fwd, kp->nlevels, w.M, w.M, w.M, lane);
    } else {
      if (rows_lds) products_forward<MTP_PU>(bt.rows, bt.level, kp->nlevels, w.M, lane);
      else products_forward<MTP_PU>(kp->rows, bt.level, kp->nlevels, w.M, lane);
    }
    // ---- site energy (pair_mtp.cpp:204-212): the leaf rows' share first ------------------------------------
    double e = 0.0;
    // (two code paths per table home, LDS blob or HBM/L2: no pointer selects between address spaces, see below)
    const int leaf_beg = __builtin_amdgcn_readfirstlane(bt.level[kp->nlevels]);
    const int leaf_nit = (__builtin_amdgcn_readfirstlane(bt.level[kp->nlevels + 1]) - leaf_beg) >> 6;
    if (rows_lds) e = leaf_forward<MTP_PU, GRADE, false>(bt.rows, bt.leaf_cf, leaf_beg, leaf_nit, w.M, lane);
    else e = leaf_forward<MTP_PU, GRADE, true>(kp->rows, kp->leaf_cf, leaf_beg, leaf_nit, w.M, lane);
    STAMP(4);   // products forward
    // ---- candidate vector, species and linear blocks (pair_mtp_extrapolation.cpp:235-252) ----
    if (GRADE) {
      double *crow = kp->cvec + (size_t) ii * kp->cpad + kp->Sp * kp->Sp * kp->Mu * kp->R;
      for (int k = lane; k < kp->Sp; k += 64) crow[k] = k == itype ? 1.0 : 0.0;
      for (int k = lane; k < kp->S; k += 64) crow[kp->Sp + k] = w.M[kp->g_map_all[k]];
    }
    if (kp->scalars_in_lds)
      for (int k = lane; k < kp->Se; k += 64) e += bt.lin[k] * w.M[bt.map[k]];
    else
      for (int k = lane; k < kp->Se; k += 64) e += kp->g_lin[k] * w.M[kp->g_map[k]];
    if (e_per_atom) {
      e = wave_sum(e) + kp->species_coeffs[itype];
      if (lane == 9) {   // (lane 9 carries the energy tally; nothing of e stays live into the force phase)
        if ((kp->eflag & 2) && kp->eatom) kp->eatom[i] = e;
        if (kp->eflag & 1) tally += e;
      }
    } else {
      eacc += e + (lane == 0 ? kp->species_coeffs[itype] : 0.0);
    }
    // ---- 4b. adjoints (pair_mtp.cpp:217-233) ----------------------------------------------
    if (kp->scalars_in_lds)
      for (int k = lane; k < kp->nseed; k += 64) w.D[bt.seed_idx[k]] = bt.seed_val[k];
    else
      for (int k = lane; k < kp->nseed; k += 64) w.D[kp->g_seed_idx[k]] = kp->g_seed_val[k];
    wave_fence();
    STAMP(5);   // energy + seeds
    if (rows_lds) leaf_backward<MTP_PU, false>(bt.rows, bt.leaf_cb, leaf_beg, leaf_nit, w.M, w.D, lane);
    else leaf_backward<MTP_PU, true>(kp->rows, kp->leaf_cb, leaf_beg, leaf_nit, w.M, w.D, lane);
    if constexpr (GATHER) {
      gather_pass(kp->prog_bwd, bt.seg_bwd, kp->nlevels, w.D, w.M, w.D, lane);
    } else {
      if (rows_lds) products_backward<MTP_PU>(bt.rows, bt.level, kp->nlevels, w.M, w.D, lane);
      else products_backward<MTP_PU>(kp->rows, bt.level, kp->nlevels, w.M, w.D, lane);
    }

    STAMP(6);   // products backward
    // ---- 5. forces ---------------------------------------------------------------------------
    // the (now free) moment region receives the coefficient blocks of the derivative polynomials:
    // basic k = (slot s; a, b, c) puts a D_k at the d/dx coefficient of x^(a-1) y^b z^c, b D_k and c D_k alike
    if (!kp->coef_dense) {   // monomials the potential does not list
      for (int k = lane; k < kp->coef_total; k += 64) w.coef[k] = 0.0;
      wave_fence();
    }
    for (int k0 = 0; k0 < (GRADE && kp->dbasic ? max(kp->dpad, kp->B) : kp->B); k0 += 192) {
      constexpr int ROUNDS = 3;   // 192 basics per trip: all reads first, one LDS round trip
      double dd[ROUNDS];
      int2 tg[ROUNDS];
      if (kp->tgt_in_lds) {   // (uniform; two loops, not a pointer select between address spaces)
#pragma unroll
        for (int u = 0; u < ROUNDS; u++) tg[u] = reinterpret_cast<const int2 *>(bt.coef)[min(k0 + lane + 64 * u, kp->B - 1)];
      } else {
#pragma unroll
        for (int u = 0; u < ROUNDS; u++) tg[u] = reinterpret_cast<const int2 *>(kp->g_tgt)[min(k0 + lane + 64 * u, kp->B - 1)];
      }
#pragma unroll
      for (int u = 0; u < ROUNDS; u++) dd[u] = w.D[min(k0 + lane + 64 * u, kp->B - 1)];
#pragma unroll
      for (int u = 0; u < ROUNDS; u++) {
        const int k = k0 + lane + 64 * u;
        const bool ok = k < kp->B;
        if (GRADE && kp->dbasic && k < kp->dpad) kp->dbasic[(size_t) ii * kp->dpad + k] = ok ? dd[u] : 0.0;   // read back by mtp_cvec_kernel
        if (ok) {
          const unsigned t0 = (unsigned) tg[u].x, t1 = (unsigned) tg[u].y;
          const unsigned tx = t0 & 0xffffu, ty_ = t0 >> 16, tz_ = t1 & 0xffffu;
          if (tx != 0xffffu) w.coef[tx] = dd[u] * (double) ((t1 >> 16) & 15u);
          if (ty_ != 0xffffu) w.coef[ty_] = dd[u] * (double) ((t1 >> 20) & 15u);
          if (tz_ != 0xffffu) w.coef[tz_] = dd[u] * (double) ((t1 >> 24) & 15u);
        }
      }
    }
    wave_fence();
    STAMP(9);   // coefficient blocks
    {
      const int n = lane & 31, part = lane >> 5;
      unsigned pcol = w.addr(w.tab + n);
      asm volatile("" : "+v"(pcol));
      double crad = 0.0;
      for (int tile = 0; tile < ntiles; tile++) {
        const int t0 = tile * NT, nt = min(NT, cnt - t0), ntp = ((nt + NG - 1) / NG) * NG;
        if (ntiles > 1 || rebuild)   // (single-tile atoms in the persistent layouts: the g rows and park[] of the tile build stand)
          build_tile<PITCH>(kp, bt, w, t0, cnt, ntp, ntiles > 1, false, !nodg, nodg, park, xi0, xi1, xi2, i, itype, lane);
        if (nodg) fp_from_parked<PITCH>(kp, w, ntp, park, lane);
        // columns past ntp hold stale (finite or not) data: their lanes are masked at the end
        const double x = w.nbx[n], y = w.nby[n], z = w.nbz[n], inv = w.nbi[n];
        double UA = 0.0, VA = 0.0, UB = 0.0, VB = 0.0, S0 = 0.0;
        double Wm[4] = {0.0, 0.0, 0.0, 0.0};
        const bool fused = GRADE && kp->grade_fused;
        {   // rank 0: P_s = D_k, no gradient; dg_s = f'_mu (nodg: its row fp_row + mu; else the slot's dg row)
          const unsigned cg0 = pcol + 8u * (unsigned) (nodg ? kp->fp_row * PITCH : kp->dg_off);
          for (int sidx = 0; sidx < kp->deg_first[1]; sidx++) {
            const double dk = w.coef[kp->deg_coef[0] + sidx];
            const int mu = (nodg || GRADE) ? __builtin_amdgcn_readfirstlane(bt.smu[sidx]) : 0;
            S0 = fma(lds_ld(cg0 + 8u * (unsigned) ((nodg ? mu : sidx) * PITCH), 0), dk, S0);
            if (GRADE) {
#pragma unroll
              for (int v = 0; v < 4; v++) Wm[v] += (mu == v && part == 0) ? dk : 0.0;
            }
          }
        }
        double mono[DEG * (DEG + 1) / 2];
        mono[0] = 1.0;
        if (nodg) {
          force_degree<1, DEG, PITCH, GRADE, true>(kp, pcol, w.addr(w.coef), part, x, y, z, mono, UA, VA, UB, VB, bt.smu, inv, inv, Wm);
          // dg_s / nu = f'_mu r^-nu / nu - g_s / r: the second term of every slot at once
          VA = fma(-inv, UA, VA);
          VB = fma(-inv, UB, VB);
        } else if constexpr (!NODG_CT) {
          force_degree<1, DEG, PITCH, GRADE, false>(kp, pcol, w.addr(w.coef), part, x, y, z, mono, UA, VA, UB, VB, bt.smu, inv, inv, Wm);
        }
        if (fused) {
          // c[jt][mu][ri] += sum_n [type_n = jt] Q_ri(r_n) W_mu(n)  (pair_mtp_extrapolation.cpp:193-198, 323-329):
          // half h of the wavefront reduces the 32 (mu, ri) entries of jt = h over its 32 neighbour lanes
          double qv[8];
          {
            const double r = w.nbr[n], d = r - kp->rmax;
            const double ksi = (2.0 * r - (kp->rmin + kp->rmax)) * kp->inv_span;
            qv[0] = kp->scaling * (d * d);
            qv[1] = kp->scaling * (ksi * d * d);
#pragma unroll
            for (int ri = 2; ri < 8; ri++) qv[ri] = 2.0 * ksi * qv[ri - 1] - qv[ri - 2];
          }
          const bool mine = n < nt && w.nbjt[n] == part;
          double ent[32];
#pragma unroll
          for (int v = 0; v < 4; v++) {
            const double wt = pair_sum32(Wm[v]);   // every lane takes part in the exchange; masked afterwards
#pragma unroll
            for (int ri = 0; ri < 8; ri++) ent[8 * v + ri] = mine ? qv[ri] * wt : 0.0;
          }
          Butterfly<32>::run(ent, lane);
          crad += ent[0];   // lane (h, e): entry e = mu R + ri of block jt = h
        }
        // sum_s dg_s P_s = r . sum_s (dg_s / nu) grad P_s  (+ rank 0), shared by both halves
        double S = (part ? z : x) * VA + y * VB + (part ? 0.0 : S0);
        S = pair_sum32(S);
        UB = pair_sum32(UB);
        const double sr = S * inv;
        const bool valid = n < nt;
        const double Fa = valid ? fma(sr, part ? z : x, UA) : 0.0;   // half 0: F_x, half 1: F_z
        const double Fy = valid && part == 0 ? fma(sr, y, UB) : 0.0;
        const double Fx = part ? 0.0 : Fa, Fz = part ? Fa : 0.0;
        if (valid) {
          const size_t j = (size_t) w.nbj[n];
          unsigned j2 = (unsigned) j << 1;
          asm volatile("" : "+v"(j2));
          const size_t j3 = (size_t) (j2 + (unsigned) j);   // 3 j without a quarter-rate multiply
          force_add(kp, j3 + (part ? 2 : 0), -Fa);   // pair_mtp.cpp:252-254
          if (part == 0) force_add(kp, j3 + 1, -Fy);
        }
        // ---- totals of this tile over the 64 lanes: force on i (3), virial (6); lane v < 9 ends up with value v.
        // Per tile, not per atom: nine running sums carried across the tile loop would be live through the whole
        // force phase (18 VGPRs the 168-VGPR build does not have); atoms with more than 32 neighbours pay one more
        // reduction per extra tile.
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
        if (kp->vflag && valid) {   // pair_mtp.cpp:257-277 (linear in F: each half tallies its components)
          v0 = -Fx * x;
          v1 = -Fy * y;
          v2 = -Fz * z;
          v3 = -(Fx * y + Fy * x) * 0.5;
          v4 = -(Fx * z + Fz * x) * 0.5;
          v5 = -(Fy * z + Fz * y) * 0.5;
        }
        double tot;
        if (kp->vflag && !v_per_atom) {
          vacc[0] += v0;
          vacc[1] += v1;
          vacc[2] += v2;
          vacc[3] += v3;
          vacc[4] += v4;
          vacc[5] += v5;
        }
        if (v_per_atom) {
          double part9[9] = {Fx, Fy, Fz, v0, v1, v2, v3, v4, v5};
          butterfly9(part9, lane);
          tot = part9[0];
        } else {
          // the mirror partners flip the low lane bits too, so they go first (while every lane still holds
          // all entries); the quad butterfly then leaves entry (lane & 3) summed over the row
          double part4[4] = {Fx, Fy, Fz, 0.0};
#pragma unroll
          for (int u = 0; u < 3; u++) {
            part4[u] += partner_f64<8>(part4[u]);
            part4[u] += partner_f64<4>(part4[u]);
          }
          Butterfly<4>::run(part4, lane);
          tot = part4[0];
        }
        tot = pair_sum32(pair_sum16(tot));
        if (lane < 9) {
          if (lane < 3) {
            force_add(kp, 3 * (size_t) i + lane, tot);   // pair_mtp.cpp:248-250
          } else if (v_per_atom) {
            tally += tot;
            if ((kp->vflag & 4) && kp->vatom) kp->vatom[6 * (size_t) i + (lane - 3)] += tot;
          }
        }
        if (ntiles > 1) wave_fence();
      }
      if (GRADE && kp->grade_fused) {   // radial block of the row: block (itype, jt), zeros elsewhere
        const int MuR = kp->Mu * 8, SMR = kp->Sp * MuR;
        double *crow = kp->cvec + (size_t) ii * kp->cpad;
        for (int e = lane; e < kp->Sp * SMR; e += 64)
          if (e / SMR != itype) crow[e] = 0.0;
        if (part < kp->Sp && n < MuR) crow[(itype * kp->Sp + part) * MuR + n] = crad;
      }
    }
    STAMP(7);   // forces
    wave_fence();
    STAMP(8);   // per-atom totals
  }
#ifdef MTP_STAMPS
  if (lane == 0 && kp->stamps)
    for (int k = 0; k < 10; k++) atomicAdd(kp->stamps + k, st_acc[k]);
  if (lane == 0 && kp->stamps) {
    atomicAdd(kp->stamps + 10, st_prologue);
    atomicAdd(kp->stamps + 11, __builtin_amdgcn_s_memtime() - st_entry);   // the wavefront's life up to here
    atomicAdd(kp->stamps + 12, 1ull);                                        // wavefronts
  }
#endif
  if (kp->vflag && !v_per_atom) {   // the deferred virial: one transpose-reduce for all atoms of the wavefront
    double part16[16] = {0.0, 0.0, 0.0, vacc[0], vacc[1], vacc[2], vacc[3], vacc[4], vacc[5], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    Butterfly<16>::run(part16, lane);
    const double tot = pair_sum32(pair_sum16(part16[0]));
    if (lane >= 3 && lane < 9) tally += tot;
  }
  if ((kp->eflag & 1) && !e_per_atom) {
    const double et = wave_sum(eacc);
    if (lane == 9) tally += et;
  }
  if (lane >= 3 && lane <= 9 && tally != 0.0) {
    // quantity-major slots [8][MTP_EV_SLOTS]: the fold reads each quantity's slots as one contiguous run
    double *slot = kp->ev_slots + (size_t) ((blockIdx.x * wpb + wave) % MTP_EV_SLOTS);
    unsafeAtomicAdd(&slot[(size_t) (lane == 9 ? 0 : lane - 2) * MTP_EV_SLOTS], tally);
  }
}

// f[k] += fq[k] 2^-40, fq[k] = 0: the end of a deterministic-mode force call
__global__ void __launch_bounds__(256) mtp_fixed_to_force(long long *__restrict__ fq, double *__restrict__ f, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const long long q = fq[e];
  if (q != 0) {
    f[e] += (double) q * (1.0 / MTP_FIXED_SCALE);
    fq[e] = 0;
  }
}

// folds the per-wave tally slots into ev[7] (accumulating) and clears them: one workgroup of 512 threads per
// quantity, a fixed summation order (lane-strided partial sums, then a fixed tree), so the totals do not depend on timing
__global__ void __launch_bounds__(512) mtp_ev_finish(double *ev_slots, double *ev)
{
  __shared__ double part[8];
  const int q = blockIdx.x;   // 0..6
  double s = 0.0;
  for (int k = threadIdx.x; k < MTP_EV_SLOTS; k += 512) {
    s += ev_slots[(size_t) q * MTP_EV_SLOTS + k];
    ev_slots[(size_t) q * MTP_EV_SLOTS + k] = 0.0;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) t += part[w];
    ev[q] += t;
  }
}

// The tally fold and the fold of the received ghost forces of a decomposed step in one launch (mtp_halo_force_step):
// workgroups [0, 7) are mtp_ev_finish (when `fold`), the others add frecv[e] onto f[3 idx[e / 3] + e % 3] (fp64
// atomics: an owned atom can be a ghost on several peers).
__global__ void __launch_bounds__(512) mtp_ev_finish_unpack(double *ev_slots, double *ev, int fold, double *__restrict__ f,
                                                           const int *__restrict__ idx, const double *__restrict__ frecv,
                                                           int n3)
{
  if (blockIdx.x >= 7) {
    const int e = (blockIdx.x - 7) * 512 + threadIdx.x;
    if (e < n3) {
      const int k = e / 3, c = e - 3 * k;
      unsafeAtomicAdd(&f[3 * (size_t) idx[k] + c], frecv[e]);
    }
    return;
  }
  if (!fold) return;   // (whole workgroup)
  __shared__ double part[8];
  const int q = blockIdx.x;   // 0..6
  double s = 0.0;
  for (int k = threadIdx.x; k < MTP_EV_SLOTS; k += 512) {
    s += ev_slots[(size_t) q * MTP_EV_SLOTS + k];
    ev_slots[(size_t) q * MTP_EV_SLOTS + k] = 0.0;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) t += part[w];
    ev[q] += t;
  }
}

// f[0, n) = 0 in one launch (hipMemsetAsync splits into two fill kernels for sizes that are not multiples of its tile)
__global__ void __launch_bounds__(256) mtp_zero_kernel(double2 *__restrict__ p, size_t n2, double *__restrict__ tail, int ntail)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n2) p[i] = make_double2(0.0, 0.0);
  if (i < (size_t) ntail) tail[i] = 0.0;
}

// ---- MaxVol grade: grades[i] = max_r | sum_c cvec[i][c] Ainv[r][c] |  (pair_mtp_extrapolation.cpp:347-358)
// One inverse active set for every atom: a dense [atoms x C] x [C x C] fp64 contraction, done on
// the matrix cores with v_mfma_f64_16x16x4_f64.  One wavefront owns 16 atoms; M = atoms, N = rows of
// Ainv, K = coefficients.  Operand lane maps (cdna_hip_programming.md section 3): A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], D: col = l&15, row = (l>>4) + 4*reg.  Both arrays are zero padded to cpad
// (multiple of 16), so no bounds checks on c or r.
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int KS_REG>   // > 0: the 16 x cpad block of cvec lives in registers (cpad <= 4*KS_REG)
__global__ void __launch_bounds__(256) mtp_grade_kernel(const double *__restrict__ cvec,
                                                       const double *__restrict__ ainv, int cpad, int inum,
                                                       const int *__restrict__ ilist, double *grades,
                                                       double *max_grade)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int atom0 = (blockIdx.x * 4 + wave) * 16;
  if (atom0 >= inum) return;
  const int li = lane & 15, lk = lane >> 4;
  const int ksteps = cpad >> 2;
  const int arow = min(atom0 + li, inum - 1);   // clamped: rows past the end are computed and dropped
  const double *ap = cvec + (size_t) arow * cpad + lk;
  double areg[KS_REG > 0 ? KS_REG : 1];
  if (KS_REG > 0) {
#pragma unroll
    for (int ks = 0; ks < KS_REG; ks++) areg[ks] = ks < ksteps ? ap[4 * ks] : 0.0;
  }
  double gmax[4] = {0.0, 0.0, 0.0, 0.0};
  for (int n0 = 0; n0 < cpad; n0 += 16) {
    const double *bp = ainv + (size_t) (n0 + li) * cpad + lk;   // B[k][j] = Ainv[n0 + j][k]
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (KS_REG > 0) {
#pragma unroll
      for (int ks = 0; ks < KS_REG; ks++)
        if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[ks], bp[4 * ks], acc, 0, 0, 0);
    } else {
      for (int ks = 0; ks < ksteps; ks++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * ks], bp[4 * ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) gmax[r] = fmax(gmax[r], fabs(acc[r]));
  }
  // max over the 16 lanes (Ainv rows) that share l>>4
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) gmax[r] = fmax(gmax[r], shfl_xor_f64(gmax[r], sft));
  }
  double wmax = 0.0;
  if (li == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int a = atom0 + lk + 4 * r;   // D row = (l>>4) + 4*reg
      if (a < inum) {
        grades[ilist[a]] = gmax[r];   // pair_mtp_extrapolation.cpp:335
        wmax = fmax(wmax, gmax[r]);
      }
    }
  }
#pragma unroll
  for (int sft = 16; sft < 64; sft <<= 1) wmax = fmax(wmax, shfl_xor_f64(wmax, sft));
  // grades are >= 0, so their IEEE bit patterns order like unsigned integers
  if (lane == 0 && max_grade)
    atomicMax(reinterpret_cast<unsigned long long *>(max_grade), (unsigned long long) __double_as_longlong(wmax));
}

// The same contraction for cpad <= 160 with the inverse active set shared through LDS: a workgroup of 8 wavefronts
// (128 atoms, their 16 x cpad blocks of cvec in registers) walks the 16-row tiles of Ainv together; each tile is
// fetched from L2 once per workgroup into a double-buffered LDS stage, already in MFMA operand order (the host
// stores Ainv as [tile][k-step][lane], lane (j = l&15, k = l>>4) holding Ainv[16 tile + j][4 kstep + k]), so a B
// operand is one conflict-free 512-byte ds_read per MFMA.  L2 traffic drops from 205 KB per 16 atoms to per 128.
template <int KS_REG>
__global__ void __launch_bounds__(512) mtp_grade_kernel_lds(const double *__restrict__ cvec,
                                                           const double *__restrict__ ainv_t, int cpad, int inum,
                                                           const int *__restrict__ ilist, double *grades,
                                                           double *max_grade)
{
  extern __shared__ double stage[];   // [2][cpad * 16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int atom0 = (blockIdx.x * 8 + wave) * 16;
  const bool live = atom0 < inum;     // idle wavefronts still take part in the staging and the barriers
  const int li = lane & 15, lk = lane >> 4;
  const int ksteps = cpad >> 2, ntile = cpad >> 4, tile_doubles = cpad * 16;
  const int arow = min(atom0 + li, inum - 1);   // clamped: rows past the end are computed and dropped
  const double *ap = cvec + (size_t) arow * cpad + lk;
  double areg[KS_REG];
#pragma unroll
  for (int ks = 0; ks < KS_REG; ks++) areg[ks] = ks < ksteps ? ap[4 * ks] : 0.0;
  constexpr int PF = (KS_REG * 64 + 511) / 512;   // doubles per thread per tile
  double pf[PF];
#pragma unroll
  for (int u = 0; u < PF; u++) {
    const int e = threadIdx.x + 512 * u;
    if (e < tile_doubles) stage[e] = ainv_t[e];
  }
  __syncthreads();
  double gmax[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < ntile; t++) {
    const double *cur = stage + (size_t) (t & 1) * tile_doubles;
    double *nxt = stage + (size_t) ((t + 1) & 1) * tile_doubles;
    if (t + 1 < ntile) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int e = threadIdx.x + 512 * u;
        pf[u] = e < tile_doubles ? ainv_t[(size_t) (t + 1) * tile_doubles + e] : 0.0;
      }
    }
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS_REG; ks++)
      if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[ks], cur[ks * 64 + lane], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) gmax[r] = fmax(gmax[r], fabs(acc[r]));
    if (t + 1 < ntile) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int e = threadIdx.x + 512 * u;
        if (e < tile_doubles) nxt[e] = pf[u];
      }
    }
    __syncthreads();
  }
  // max over the 16 lanes (Ainv rows) that share l>>4
#pragma unroll
  for (int r = 0; r < 4; r++) {
    gmax[r] = fmax(gmax[r], partner_f64<1>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<2>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<4>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<8>(gmax[r]));
  }
  double wmax = 0.0;
  if (live && li == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int a = atom0 + lk + 4 * r;   // D row = (l>>4) + 4*reg
      if (a < inum) {
        grades[ilist[a]] = gmax[r];   // pair_mtp_extrapolation.cpp:335
        wmax = fmax(wmax, gmax[r]);
      }
    }
  }
  wmax = fmax(wmax, partner_f64<16>(wmax, lane));
  wmax = fmax(wmax, partner_f64<32>(wmax, lane));
  // grades are >= 0, so their IEEE bit patterns order like unsigned integers
  if (live && lane == 0 && max_grade)
    atomicMax(reinterpret_cast<unsigned long long *>(max_grade), (unsigned long long) __double_as_longlong(wmax));
}

// Output-stationary form of the same contraction (round 2): the wavefront keeps the NT = cpad / 16 accumulator tiles of
// its 16 atoms (16 atoms x cpad rows of Ainv) and walks K in slabs of four k-steps; a slab of Ainv (4 k-steps of EVERY
// 16-row tile, 2 KB per tile, contiguous in the tiled layout) is staged through the double-buffered LDS by the
// workgroup and the four A operands of the next slab are fetched while the current slab's 4 NT MFMAs run.  The
// A-stationary kernel above first pulls its whole 16 x cpad block of candidate vectors into registers -- every
// wavefront of the one-round launch at the same time, 84 MB with the matrix pipe idle -- and its MFMAs form one
// dependent chain per tile; here the loads are spread over the K loop and NT independent chains are in flight.
#ifndef MTP_GRADE_TPB
#define MTP_GRADE_TPB 512
#define MTP_GRADE_WPE 2
#endif
template <int NT, int TPB, int WPE>   // TPB threads per workgroup (TPB / 64 wavefronts of 16 atoms), WPE wavefronts per SIMD
__global__ void __launch_bounds__(TPB, WPE) mtp_grade_kernel_os(const double *__restrict__ cvec,
                                                          const double *__restrict__ ainv_t, int inum,
                                                          const int *__restrict__ ilist, double *grades,
                                                          double *max_grade)
{
  extern __shared__ double stage[];   // [2][NT * 256]
  constexpr int cpad = 16 * NT, KS = 4 * NT, NSLAB = NT, slab_doubles = NT * 256;
  constexpr int PF = (slab_doubles + TPB - 1) / TPB;   // doubles per thread per slab
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int atom0 = (blockIdx.x * (TPB / 64) + wave) * 16;
  const bool live = atom0 < inum;     // idle wavefronts still take part in the staging and the barriers
  const int li = lane & 15, lk = lane >> 4;
  const int arow = min(atom0 + li, inum - 1);   // clamped: rows past the end are computed and dropped
  const double *ap = cvec + (size_t) arow * cpad + lk;
  // slab c, element e = tile * 256 + (u * 64 + lane)  <-  ainv_t[(tile * KS + 4 c) * 64 + (u * 64 + lane)]
  auto src = [&](int c, int e) { return ainv_t[((size_t) (e >> 8) * KS + 4 * c) * 64 + (e & 255)]; };
  double pf[PF], a_cur[4], a_nxt[4];
#pragma unroll
  for (int u = 0; u < PF; u++) {
    const int e = threadIdx.x + TPB * u;
    if (e < slab_doubles) stage[e] = src(0, e);
  }
#pragma unroll
  for (int u = 0; u < 4; u++) a_cur[u] = ap[4 * u];
  __syncthreads();
  double4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  for (int c = 0; c < NSLAB; c++) {
    const double *cur = stage + (size_t) (c & 1) * slab_doubles + lane;
    double *nxt = stage + (size_t) ((c + 1) & 1) * slab_doubles;
    if (c + 1 < NSLAB) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int e = threadIdx.x + TPB * u;
        pf[u] = e < slab_doubles ? src(c + 1, e) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) a_nxt[u] = ap[4 * (4 * (c + 1) + u)];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int t = 0; t < NT; t++)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[u], cur[(t * 4 + u) * 64], acc[t], 0, 0, 0);
    if (c + 1 < NSLAB) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int e = threadIdx.x + TPB * u;
        if (e < slab_doubles) nxt[e] = pf[u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) a_cur[u] = a_nxt[u];
    }
    __syncthreads();
  }
  double gmax[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) gmax[r] = fmax(gmax[r], fabs(acc[t][r]));
  // max over the 16 lanes (Ainv rows) that share l>>4
#pragma unroll
  for (int r = 0; r < 4; r++) {
    gmax[r] = fmax(gmax[r], partner_f64<1>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<2>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<4>(gmax[r]));
    gmax[r] = fmax(gmax[r], partner_f64<8>(gmax[r]));
  }
  double wmax = 0.0;
  if (live && li == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int a = atom0 + lk + 4 * r;   // D row = (l>>4) + 4*reg
      if (a < inum) {
        grades[ilist[a]] = gmax[r];   // pair_mtp_extrapolation.cpp:335
        wmax = fmax(wmax, gmax[r]);
      }
    }
  }
  wmax = fmax(wmax, partner_f64<16>(wmax, lane));
  wmax = fmax(wmax, partner_f64<32>(wmax, lane));
  if (live && lane == 0 && max_grade)
    atomicMax(reinterpret_cast<unsigned long long *>(max_grade), (unsigned long long) __double_as_longlong(wmax));
}

// configuration mode: coeff_ders[c] += sum_i cvec[i][c]  (pair_mtp_extrapolation.cpp:97-98, 240-252, 327)
__global__ void __launch_bounds__(256) mtp_colsum_kernel(const double *__restrict__ cvec, int cpad, int C, int inum,
                                                        double *coeff_ders)
{
  const int rows_per_block = 256;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(inum, r0 + rows_per_block);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int r = r0; r < r1; r++) s += cvec[(size_t) r * cpad + c];
  unsafeAtomicAdd(&coeff_ders[c], s);
}

template <int KL, int NB, int PITCH, bool GRADE, int DEG, int WPS>
hipError_t launch_one(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  // the dynamic-LDS limit is a per-device attribute of the function: one bit per device id
  static unsigned long long attr_mask = 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev > 63 || !((attr_mask >> dev) & 1ull)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mtp_wave_kernel<KL, NB, PITCH, GRADE, DEG, WPS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev <= 63) attr_mask |= 1ull << dev;
  }
  hipLaunchKernelGGL((mtp_wave_kernel<KL, NB, PITCH, GRADE, DEG, WPS>), dim3(grid), dim3(64 * wpb), lds, st, p);
  return hipGetLastError();
}

template <int KL, int NB, int DEG, int WPS>
hipError_t launch_grade(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  // only the 32-neighbour tile (pitch MTP_PITCH) is instantiated; the grade variant is its own instantiation so the
  // force-only kernel keeps its register budget
  return p.grade_flag ? launch_one<KL, NB, MTP_PITCH, true, DEG, WPS>(p, grid, wpb, lds, st)
                      : launch_one<KL, NB, MTP_PITCH, false, DEG, WPS>(p, grid, wpb, lds, st);
}

// DEG = highest tensor rank the unrolled force phase covers (monomials up to degree DEG-1 in registers):
// narrow lane grids come with ranks <= 6 in the MLIP level tables, wide ones with ranks <= 8;
// DEG = 11 is the general instantiation (the loader caps the rank at 11).  The 168-VGPR build (three wavefronts per
// SIMD) exists for the narrow grids with ranks <= 6 -- the shapes whose per-atom LDS image lets twelve wavefronts
// share a CU (mtp_wave_kernel_has_wps3() tells the planner).
template <int KL, int NB> hipError_t launch_pitch(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  constexpr int DLOW = KL <= 32 ? 6 : 8;
  if (p.P - 1 <= DLOW) {
    if constexpr (KL <= 32 && NB == 1)
      if (p.wps == 3) return launch_grade<KL, NB, DLOW, 3>(p, grid, wpb, lds, st);
    return launch_grade<KL, NB, DLOW, 2>(p, grid, wpb, lds, st);
  }
  return launch_grade<KL, NB, 11, 2>(p, grid, wpb, lds, st);
}

}   // namespace

// Lane grids of the candidate-vector kernel (KL k-lanes x KB basics per lane); also sizes dbasic rows.
int mtp_pick_shape(int B, int *KL, int *KB)
{
  static const int kb16[] = {2, 3, 5, 7, 9, 10}, kbw[] = {6, 7, 8, 10};
  for (int v : kb16)
    if (B <= 16 * v) {
      *KL = 16;
      *KB = v;
      return 0;
    }
  for (int kl : {32, 64})
    for (int v : kbw)
      if (B <= kl * v) {
        *KL = kl;
        *KB = v;
        return 0;
      }
  return -1;
}

// Lane grids of the force kernel's basic-moment pass: KL block lanes x NB 3x3 blocks per lane (KL * NB >= blocks);
// mtp_pick_fwd_shape() is the single source of truth for the host.
int mtp_pick_fwd_shape(int nblk, int *KL, int *NB)
{
  for (int kl : {16, 32, 64})
    if (nblk <= kl) {
      *KL = kl;
      *NB = 1;
      return 0;
    }
  for (int nb : {2, 3, 4})
    if (nblk <= 64 * nb) {
      *KL = 64;
      *NB = nb;
      return 0;
    }
  return -1;
}

// whether the three-wavefronts-per-SIMD build exists for this table shape (see launch_pitch)
bool mtp_wave_kernel_has_wps3(int nfb, int P)
{
  int KL = 0, NB = 0;
  return mtp_pick_fwd_shape(nfb, &KL, &NB) == 0 && KL <= 32 && NB == 1 && P - 1 <= 6;
}

hipError_t mtp_launch_wave_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  int KL = 0, NB = 0;
  if (mtp_pick_fwd_shape(p.nfb, &KL, &NB) != 0 || p.NT != 32) return hipErrorInvalidValue;
  if (wpb < 1 || wpb > (p.wps == 3 ? 12 : 8)) return hipErrorInvalidValue;
#define MTP_CASE(kl, nb) \
  if (KL == kl && NB == nb) return launch_pitch<kl, nb>(p, grid, wpb, lds, st);
  MTP_CASE(16, 1)
  MTP_CASE(32, 1)
  MTP_CASE(64, 1)
  MTP_CASE(64, 2)
  MTP_CASE(64, 3)
  MTP_CASE(64, 4)
#undef MTP_CASE
  return hipErrorInvalidValue;
}

hipError_t mtp_launch_grade_kernel(const double *cvec, const double *ainv_pad, const double *ainv_tiled, int cpad,
                                   int C, int inum, const int *ilist, double *grades, double *max_grade, hipStream_t st)
{
  (void) C;
  static const bool use_os = !(std::getenv("MTP_GRADE_OS") && std::atoi(std::getenv("MTP_GRADE_OS")) == 0);   // tuning
  if (cpad <= 160 && ainv_tiled && use_os) {
    const size_t lds = (size_t) 2 * cpad * 16 * sizeof(double);
    constexpr int TPB = MTP_GRADE_TPB, WPE = MTP_GRADE_WPE;
    const dim3 grid((inum + TPB / 4 - 1) / (TPB / 4)), block(TPB);
#define MTP_GRADE_OS_CASE(NT)                                                                                         \
  case NT:                                                                                                            \
    hipLaunchKernelGGL((mtp_grade_kernel_os<NT, TPB, WPE>), grid, block, lds, st, cvec, ainv_tiled, inum, ilist, grades,  \
                       max_grade);                                                                                    \
    break;
    switch (cpad / 16) {
      MTP_GRADE_OS_CASE(1) MTP_GRADE_OS_CASE(2) MTP_GRADE_OS_CASE(3) MTP_GRADE_OS_CASE(4) MTP_GRADE_OS_CASE(5)
      MTP_GRADE_OS_CASE(6) MTP_GRADE_OS_CASE(7) MTP_GRADE_OS_CASE(8) MTP_GRADE_OS_CASE(9) MTP_GRADE_OS_CASE(10)
      default: return hipErrorInvalidValue;
    }
#undef MTP_GRADE_OS_CASE
  } else if (cpad <= 160 && ainv_tiled) {
    const size_t lds = (size_t) 2 * cpad * 16 * sizeof(double);
    hipLaunchKernelGGL(mtp_grade_kernel_lds<40>, dim3((inum + 127) / 128), dim3(512), lds, st, cvec, ainv_tiled, cpad,
                       inum, ilist, grades, max_grade);
  } else {
    hipLaunchKernelGGL(mtp_grade_kernel<0>, dim3((inum + 63) / 64), dim3(256), 0, st, cvec, ainv_pad, cpad, inum, ilist,
                       grades, max_grade);
  }
  return hipGetLastError();
}

hipError_t mtp_launch_colsum_kernel(const double *cvec, int cpad, int C, int inum, double *coeff_ders, hipStream_t st)
{
  hipLaunchKernelGGL(mtp_colsum_kernel, dim3((C + 255) / 256, (inum + 255) / 256), dim3(256), 0, st, cvec, cpad, C, inum,
                     coeff_ders);
  return hipGetLastError();
}

hipError_t mtp_launch_fixed_to_force(long long *fq, double *f, int nall, hipStream_t st)
{
  hipLaunchKernelGGL(mtp_fixed_to_force, dim3((3 * nall + 255) / 256), dim3(256), 0, st, fq, f, 3 * nall);
  return hipGetLastError();
}

hipError_t mtp_launch_ev_finish_unpack(double *ev_slots, double *ev, int fold, double *f, const int *idx, const double *frecv,
                                       int n3, hipStream_t st)
{
  hipLaunchKernelGGL(mtp_ev_finish_unpack, dim3(7 + (n3 + 511) / 512), dim3(512), 0, st, ev_slots, ev, fold, f, idx, frecv, n3);
  return hipGetLastError();
}

hipError_t mtp_launch_ev_finish(double *ev_slots, double *ev, hipStream_t st)
{
  hipLaunchKernelGGL(mtp_ev_finish, dim3(7), dim3(512), 0, st, ev_slots, ev);
  return hipGetLastError();
}

const char *mtp_kernel_build_flags()
{
#define MTP_STR2(x) #x
#define MTP_STR(x) MTP_STR2(x)
  return ""
#if MTP_PU != 2
      "MTP_PU=" MTP_STR(MTP_PU) " "
#endif
#if MTP_LD != 1
      "MTP_LD=" MTP_STR(MTP_LD) " "
#endif
#if MTP_POLY_CH != 8
      "MTP_POLY_CH=" MTP_STR(MTP_POLY_CH) " "
#endif
#if MTP_POLY_ACC != 1
      "MTP_POLY_ACC=" MTP_STR(MTP_POLY_ACC) " "
#endif

#if MTP_GRADE_TPB != 512 || MTP_GRADE_WPE != 2
      "MTP_GRADE_TPB=" MTP_STR(MTP_GRADE_TPB) " "
#endif
      ;
}

hipError_t mtp_launch_zero(double *p, size_t n, hipStream_t st)   // p 16-byte aligned (hipMalloc / torch allocations are)
{
  if (n == 0) return hipSuccess;
  const size_t n2 = n / 2;
  hipLaunchKernelGGL(mtp_zero_kernel, dim3((unsigned) ((std::max<size_t>(n2, 1) + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<double2 *>(p), n2, p + 2 * n2, (int) (n - 2 * n2));
  return hipGetLastError();
}
