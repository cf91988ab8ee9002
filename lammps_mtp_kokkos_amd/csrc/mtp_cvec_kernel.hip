// Radial block of the MaxVol candidate vector, grade calls only (pair_mtp_extrapolation.cpp:193-198,
// 323-329):
//
//   c[(it Sp + jt) Mu R + mu R + ri] = sum_k D_k RJ[k][jt][mu R + ri]
//                                    = sum_n [type_n = jt] Q_ri(r_n) W_mu(n),
//   W_mu(n) = sum_{k in mu} D_k x^a y^b z^c / r_n^nu_k
//
// It runs after the fused force kernel of a grade call, which leaves the adjoints of the basics D_k
// in HBM (dbasic[inum][KP], 1 KB per atom at level 16) and has already written the species and linear
// blocks of cvec.  Keeping this out of the force kernel keeps that kernel's register budget (the
// fused version spilled > 300 VGPRs and ran 5x slower).  One wavefront per atom, same lane grid as
// the force kernel (NG neighbour groups x KL k-lanes, butterfly over k), but the LDS table only holds
// r^-nu, Q_ri and the coordinate powers, so twice as many wavefronts fit per CU.
#include <hip/hip_runtime.h>

#include "mtp_kernel_common.hpp"

namespace {

template <int KL, int KB, int PITCH>
__global__ void __launch_bounds__(512, 2) mtp_cvec_kernel(const MtpDevParams p)
{
  constexpr int NT = PITCH - 2;
  constexpr int NG = 64 / KL;
  constexpr int NPG = NT / NG;
  constexpr int BATCH = KL / 4;
  constexpr int NBATCH = NPG / BATCH;
  constexpr int KP = KL * KB;

  extern __shared__ double lds[];
  unsigned char *sh = reinterpret_cast<unsigned char *>(lds);
  for (int o = threadIdx.x * 16; o < p.blob_bytes; o += blockDim.x * 16)
    *reinterpret_cast<uint4 *>(sh + o) = *reinterpret_cast<const uint4 *>(p.blob + o);
  __syncthreads();
  const int *pack = reinterpret_cast<const int *>(sh + p.off_pack);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int kl = lane & (KL - 1), q = lane / KL;
  const int P = p.P, R = p.R, Mu = p.Mu, MuR = Mu * R, SMR = p.Sp * MuR;
  const int rows = 4 * P + R;   // r^-nu [P] | Q_ri [R] | x^q, y^q, z^q [3P]
  // per-wavefront LDS: DK[KP] | tab[rows][PITCH] | nbx,nby,nbz,nbr [NT each] | W[Mu*NT] | stage[SMR] | ints
  const unsigned wave_off = (p.blob_bytes >> 3) + wave * p.wave_doubles;
  double *DK = lds + wave_off;
  double *tab = DK + KP;
  double *nbx = tab + (size_t) rows * PITCH, *nby = nbx + NT, *nbz = nby + NT, *nbr = nbz + NT;
  double *W = nbr + NT;
  double *stage = W + Mu * NT;
  int *nbjt = reinterpret_cast<int *>(stage + SMR);
  int *cj = nbjt + NT;
  const unsigned lds0 = (unsigned) (size_t) (lds_cdouble *) lds;
  auto addr = [&](const double *ptr) { return lds0 + 8u * (unsigned) (ptr - lds); };

  // descriptors of this lane's basics k = kl + KL t: rows of r^-nu, x^a, y^b, z^c for column q
  unsigned pr[KB], px[KB], py[KB], pz[KB];
  int mu_of[KB];
#pragma unroll
  for (int t = 0; t < KB; t++) {
    const int k = kl + KL * t;
    const int pk = k < p.B ? pack[k] : 0;
    const int a = (pk >> 8) & 15, b = (pk >> 12) & 15, c = (pk >> 16) & 15;
    mu_of[t] = k < p.B ? (pk >> 20) & 15 : -1;
    pr[t] = addr(tab + (size_t) (a + b + c) * PITCH + q);
    px[t] = addr(tab + (size_t) (P + R + a) * PITCH + q);
    py[t] = addr(tab + (size_t) (P + R + P + b) * PITCH + q);
    pz[t] = addr(tab + (size_t) (P + R + 2 * P + c) * PITCH + q);
    asm volatile("" : "+v"(pr[t]), "+v"(px[t]), "+v"(py[t]), "+v"(pz[t]));
  }
  unsigned pdk = addr(DK + kl);
  asm volatile("" : "+v"(pdk));

  for (int ii = p.row0 + blockIdx.x * wpb + wave; ii < p.row0 + p.inum; ii += gridDim.x * wpb) {
    const int i = __builtin_amdgcn_readfirstlane(p.ilist[ii]);
    const int itype = __builtin_amdgcn_readfirstlane(p.type[i] - 1);
    if (itype < 0 || itype >= p.Sp) continue;   // reported by the force kernel
    const double xi0 = uniform_f64(p.x[3 * (size_t) i]), xi1 = uniform_f64(p.x[3 * (size_t) i + 1]),
                 xi2 = uniform_f64(p.x[3 * (size_t) i + 2]);
    const int jbeg = __builtin_amdgcn_readfirstlane(p.first[ii]);
    const int jnum = __builtin_amdgcn_readfirstlane(p.first[ii + 1]) - jbeg;
    for (int k = lane; k < KP; k += 64) DK[k] = p.dbasic[(size_t) ii * p.dpad + k];   // zero padded by the producer

    // compaction (as in the force kernel; ids only)
    int cnt = 0;
    for (int c0 = 0; c0 < jnum; c0 += 64) {
      const int jj = c0 + lane;
      bool in = false;
      int j = 0;
      if (jj < jnum) {
        j = p.neigh[jbeg + jj] & MTP_NEIGHMASK;
        const int jt = p.type[j] - 1;
        if (jt >= 0 && jt < p.Sp) {
          const double dx = p.x[3 * (size_t) j] - xi0, dy = p.x[3 * (size_t) j + 1] - xi1,
                       dz = p.x[3 * (size_t) j + 2] - xi2;
          in = !(dx * dx + dy * dy + dz * dz > p.cutsq);
        }
      }
      const unsigned long long m = __ballot(in);
      if (in) cj[cnt + __popcll(m & ((1ull << lane) - 1ull))] = j;
      cnt += __builtin_amdgcn_readfirstlane(__popcll(m));
    }
    wave_fence();

    double crad[4] = {0.0, 0.0, 0.0, 0.0};   // this lane's entries e = lane + 64 ce of c[jt][mu][ri]
    const int ntiles = (cnt + NT - 1) / NT;
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * NT, nt = min(NT, cnt - t0), ntp = ((nt + NG - 1) / NG) * NG;
      // ---- tables: one lane per neighbour (dummies on the cutoff pad to a multiple of NG)
      if (lane < ntp) {
        const bool real = lane < nt;
        const int j = real ? cj[t0 + lane] : i;
        double dx = 0, dy = 0, dz = 0, r = p.rmax;
        if (real) {
          dx = p.x[3 * (size_t) j] - xi0;
          dy = p.x[3 * (size_t) j + 1] - xi1;
          dz = p.x[3 * (size_t) j + 2] - xi2;
          r = sqrt(dx * dx + dy * dy + dz * dz);
        }
        nbjt[lane] = real ? p.type[j] - 1 : -1;   // -1: never matches a species
        nbr[lane] = r;
        nbx[lane] = dx;
        nby[lane] = dy;
        nbz[lane] = dz;
        double *col = tab + lane;
        const double inv = 1.0 / r;
        double rp = 1.0;
        for (int nu = 0; nu < P; nu++) {
          col[nu * PITCH] = rp;   // r^-nu
          rp *= inv;
        }
        // Q_ri(r) (mtp_rb_chevbyshev_basis.cpp:29-38)
        const double span = p.rmax - p.rmin, d = r - p.rmax;
        const double ksi = (2.0 * r - (p.rmin + p.rmax)) / span;
        double q0 = p.scaling * (d * d), q1 = p.scaling * (ksi * d * d);
        col[P * PITCH] = q0;
        if (R > 1) col[(P + 1) * PITCH] = q1;
        for (int ri = 2; ri < R; ri++) {
          const double q2 = 2.0 * ksi * q1 - q0;
          col[(P + ri) * PITCH] = q2;
          q0 = q1;
          q1 = q2;
        }
        const double u3[3] = {dx, dy, dz};
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
          double cur = 1.0;
          double *cp = col + (size_t) (P + R + ax * P) * PITCH;
          cp[0] = 1.0;
          for (int e = 1; e < P; e++) {
            cur *= u3[ax];
            cp[e * PITCH] = cur;
          }
        }
      }
      wave_fence();
      // ---- W_mu(n), four mu per pass
      for (int mu0 = 0; mu0 < Mu; mu0 += 4) {
#pragma unroll
        for (int b = 0; b < NBATCH; b++) {
          if (b * BATCH * NG < ntp) {
            double wp[KL];
#pragma unroll
            for (int u = 0; u < KL; u++) wp[u] = 0.0;
#pragma unroll
            for (int t = 0; t < KB; t++) {
              const double Dk = lds_ld(pdk, KL * t);
              const int ml = mu_of[t] - mu0;
              constexpr int WC = BATCH < 4 ? BATCH : 4;
#pragma unroll
              for (int m0 = 0; m0 < BATCH; m0 += WC) {
                double ri[WC], xa[WC], yb[WC], zc[WC];
#pragma unroll
                for (int u = 0; u < WC; u++) {
                  const int o = (b * BATCH + m0 + u) * NG;
                  ri[u] = lds_ld(pr[t], o);
                  xa[u] = lds_ld(px[t], o);
                  yb[u] = lds_ld(py[t], o);
                  zc[u] = lds_ld(pz[t], o);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < WC; u++) {
                  const double val = (Dk * ri[u]) * (xa[u] * (yb[u] * zc[u]));
#pragma unroll
                  for (int v = 0; v < 4; v++) wp[4 * (m0 + u) + v] += ml == v ? val : 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            Butterfly<KL>::run(wp, lane);
            const int mm = kl >> 2, mu = mu0 + (kl & 3);
            const int n = q + NG * (b * BATCH + mm);
            if (n < nt && mu < Mu) W[mu * NT + n] = wp[0];
          }
        }
      }
      wave_fence();
      // ---- c[jt][mu][ri] += sum_n [type_n = jt] Q_ri(r_n) W_mu(n)
#pragma unroll
      for (int ce = 0; ce < 4; ce++) {
        const int e = lane + 64 * ce;
        if (e < SMR) {
          const int jt = e / MuR, m = e - jt * MuR, mu = m / R, ri = m - mu * R;
          const double *qrow = tab + (size_t) (P + ri) * PITCH;
          double sum = 0.0;
          for (int n = 0; n < nt; n++)
            if (nbjt[n] == jt) sum += qrow[n] * W[mu * NT + n];
          crad[ce] += sum;
        }
      }
      wave_fence();
    }
    // radial part of the row: block (itype, jt) at offset (itype Sp + jt) Mu R, zeros elsewhere
#pragma unroll
    for (int ce = 0; ce < 4; ce++)
      if (lane + 64 * ce < SMR) stage[lane + 64 * ce] = crad[ce];
    wave_fence();
    double *crow = p.cvec + (size_t) ii * p.cpad;
    for (int e = lane; e < p.Sp * SMR; e += 64) {
      const int blk = e / SMR;
      crow[e] = blk == itype ? stage[e - blk * SMR] : 0.0;
    }
    wave_fence();
  }
}

template <int KL, int KB> hipError_t launch_cvec(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mtp_cvec_kernel<KL, KB, 34>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((mtp_cvec_kernel<KL, KB, 34>), dim3(grid), dim3(64 * wpb), lds, st, p);
  return hipGetLastError();
}

}   // namespace

hipError_t mtp_launch_cvec_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  int KL = 0, KB = 0;
  if (mtp_pick_shape(p.B, &KL, &KB) != 0 || p.NT != 32) return hipErrorInvalidValue;
#define MTP_CASE(kl, kb) \
  if (KL == kl && KB == kb) return launch_cvec<kl, kb>(p, grid, wpb, lds, st);
  MTP_CASE(16, 2)
  MTP_CASE(16, 3)
  MTP_CASE(16, 5)
  MTP_CASE(16, 7)
  MTP_CASE(16, 9)
  MTP_CASE(16, 10)
  MTP_CASE(32, 6)
  MTP_CASE(32, 7)
  MTP_CASE(32, 8)
  MTP_CASE(32, 10)
  MTP_CASE(64, 6)
  MTP_CASE(64, 7)
  MTP_CASE(64, 8)
  MTP_CASE(64, 10)
#undef MTP_CASE
  return hipErrorInvalidValue;
}
