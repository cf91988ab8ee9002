// CDNA4 (gfx950) kernels of the MTP pair-style compute path.
//
// One 64-lane wavefront owns one atom (the native counterpart of both reference GPU
// styles, KOKKOS/pair_mtp_kokkos.cpp:404-660 and KOKKOS/pair_mtps_kokkos.cpp:438-708),
// and the per-pair Jacobian the reference spills to HBM
// (d_moment_jacobian[N][J][B][3], pair_mtp_kokkos.cpp:270-282) is never formed:
//
//   1. compaction   lanes = list entries: r^2 <= rc^2 test (pair_mtp.cpp:120-127),
//                   ballot/prefix -> in-cutoff neighbour ids in LDS
//   2. tile tables  per in-cutoff neighbour, in LDS: g[mu,nu] = f_mu(r)/r^nu and
//                   dg/dr (Chebyshev recurrence + radial contraction,
//                   mtp_rb_chevbyshev_basis.cpp:29-54, pair_mtp.cpp:139-166) and the
//                   coordinate powers x^p, p x^(p-1) (pair_mtp.cpp:133-136)
//   3. basic moments lanes = basic index k: M_k += g * x^a y^b z^c over the tile in
//                   registers (pair_mtp.cpp:154-172); no cross-lane reduction
//   4. products     lanes = times rows, one dependency level at a time, moments and
//                   adjoints in LDS (pair_mtp.cpp:196-233)
//   5. forces       lanes = k again: each lane contracts its adjoint D_k with the
//                   analytic d(M_k)/d(r_ij) rebuilt from the LDS tables
//                   (pair_mtp.cpp:174-191, 236-246), a butterfly transpose-reduce sums
//                   over k for 16 neighbours at a time, then lanes = neighbours scatter
//                   f_j -= F_ij with fp64 HBM atomics and tally the virial
//                   (pair_mtp.cpp:248-277)
//
// Everything is fp64 (the reference's F_FLOAT); indices are int32.
#include <hip/hip_runtime.h>

#include "mtp_device.hpp"

#define MTP_NEIGHMASK 0x1FFFFFFF   // LAMMPS NEIGHMASK (pair_mtp.cpp:114)

namespace {

__device__ __forceinline__ void wave_fence()
{
  // LDS operations of one wavefront execute in program order; this only stops the
  // compiler from moving LDS accesses across a phase boundary.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
  return __shfl_xor(v, mask, 64);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += shfl_xor_f64(v, s);
  return v;
}

__device__ __forceinline__ void lds_add(double *p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Butterfly transpose-reduce: every lane enters with N partial sums v[0..N); on exit
// lane l holds in v[0] the wave-wide total of entry (l mod N) ... for N = 64 exactly
// entry l.  N-1 adds and N-1 exchanges per lane instead of N*log2(64).
template <int N> struct Butterfly {
  static __device__ __forceinline__ void run(double *v, int lane)
  {
    constexpr int H = N / 2;
    const bool hi = (lane & H) != 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
      const double keep = hi ? v[i + H] : v[i];
      const double send = hi ? v[i] : v[i + H];
      v[i] = keep + shfl_xor_f64(send, H);
    }
    Butterfly<H>::run(v, lane);
  }
};
template <> struct Butterfly<1> {
  static __device__ __forceinline__ void run(double *, int) {}
};

struct WaveLds {
  double *M, *D, *tab, *nbx, *nby, *nbz, *nbr, *nbi, *red;
  int *nbj, *nbjt, *cj;
};

__device__ __forceinline__ WaveLds carve(double *base, const MtpDevParams &p)
{
  WaveLds w;
  w.M = base;
  w.D = w.M + p.A;
  w.tab = w.D + p.A;
  w.nbx = w.tab + (size_t) p.NT * p.stride;
  w.nby = w.nbx + p.NT;
  w.nbz = w.nby + p.NT;
  w.nbr = w.nbz + p.NT;
  w.nbi = w.nbr + p.NT;
  w.red = w.nbi + p.NT;
  w.nbj = reinterpret_cast<int *>(w.red + 64);
  w.nbjt = w.nbj + p.NT;
  w.cj = w.nbjt + p.NT;
  return w;
}

// Phase 2: tables of one tile of nt <= NT in-cutoff neighbours starting at cj[t0].
__device__ __forceinline__ void build_tile(const MtpDevParams &p, const WaveLds &w, int t0, int nt,
                                           double xi0, double xi1, double xi2, int itype, int lane)
{
  if (lane < nt) {
    const int j = w.cj[t0 + lane];
    const double dx = p.x[3 * (size_t) j] - xi0, dy = p.x[3 * (size_t) j + 1] - xi1,
                 dz = p.x[3 * (size_t) j + 2] - xi2;
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    w.nbx[lane] = dx;
    w.nby[lane] = dy;
    w.nbz[lane] = dz;
    w.nbr[lane] = r;
    w.nbi[lane] = 1.0 / r;
    w.nbj[lane] = j;
    w.nbjt[lane] = p.type[j] - 1;
  }
  wave_fence();
  const int Mu = p.Mu, P = p.P, R = p.R, ns = p.nslot;
  const double span = p.rmax - p.rmin, mult = 2.0 / span;
  for (int idx = lane; idx < nt * Mu; idx += 64) {
    const int n = idx / Mu, mu = idx - n * Mu;
    const double r = w.nbr[n], inv = w.nbi[n];
    const int jt = w.nbjt[n];
    const double *c = p.radial_coeffs + ((size_t) (itype * p.Sp + jt) * Mu + mu) * R;
    // Chebyshev values/derivatives by recurrence, contracted on the fly
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
    double *rec = w.tab + (size_t) n * p.stride;
    double rp = 1.0;
    for (int nu = 0; nu < P; nu++) {
      const int s = p.slot_of[mu * P + nu];
      const double g = val * rp;
      if (s >= 0) {
        rec[s] = g;                                 // f_mu / r^nu
        rec[ns + s] = der * rp - nu * g * inv;      // d/dr (f_mu / r^nu)
      }
      rp *= inv;
    }
  }
  for (int idx = lane; idx < nt * 3; idx += 64) {
    const int n = idx / 3, ax = idx - 3 * n;
    const double u = ax == 0 ? w.nbx[n] : (ax == 1 ? w.nby[n] : w.nbz[n]);
    double *pw = w.tab + (size_t) n * p.stride + 2 * ns + ax * P;
    double *dpw = pw + 3 * P;
    double cur = 1.0;
    pw[0] = 1.0;
    dpw[0] = 0.0;
    for (int q = 1; q < P; q++) {
      dpw[q] = q * cur;   // q u^(q-1)
      cur *= u;
      pw[q] = cur;
    }
  }
  wave_fence();
}

template <int KB> __global__ void __launch_bounds__(256) mtp_wave_kernel(const MtpDevParams p)
{
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const WaveLds w = carve(lds + (size_t) wave * p.wave_doubles, p);
  const int ns = p.nslot, P = p.P;

  // per-lane descriptors of the basics this lane owns: k = lane + 64 t
  int og[KB], oxa[KB], oyb[KB], ozc[KB];
  bool kval[KB];
#pragma unroll
  for (int t = 0; t < KB; t++) {
    const int k = lane + 64 * t;
    kval[t] = k < p.B;
    const int pk = kval[t] ? p.basic_pack[k] : 0;
    og[t] = pk & 255;
    oxa[t] = 2 * ns + ((pk >> 8) & 15);
    oyb[t] = 2 * ns + P + ((pk >> 12) & 15);
    ozc[t] = 2 * ns + 2 * P + ((pk >> 16) & 15);
  }

  double ev_acc[7] = {0, 0, 0, 0, 0, 0, 0};   // lane-0 tallies of this wave: energy + virial

  for (int ii = blockIdx.x * wpb + wave; ii < p.inum; ii += gridDim.x * wpb) {
    const int i = p.ilist[ii];
    const int itype = p.type[i] - 1;
    if (itype < 0 || itype >= p.Sp) {   // pair_mtp.cpp:91-93
      if (lane == 0) atomicExch(p.err_flag, 1);
      continue;
    }
    const double xi0 = p.x[3 * (size_t) i], xi1 = p.x[3 * (size_t) i + 1], xi2 = p.x[3 * (size_t) i + 2];
    const int jbeg = p.first[ii], jnum = p.first[ii + 1] - jbeg;

    // ---- 1. compaction --------------------------------------------------------------------
    int cnt = 0;
    for (int c0 = 0; c0 < jnum; c0 += 64) {
      const int jj = c0 + lane;
      bool in = false;
      int j = 0;
      if (jj < jnum) {
        j = p.neigh[jbeg + jj] & MTP_NEIGHMASK;
        const int jt = p.type[j] - 1;
        if (jt < 0 || jt >= p.Sp) {   // pair_mtp.cpp:116-118
          atomicExch(p.err_flag, 1);
        } else {
          const double dx = p.x[3 * (size_t) j] - xi0, dy = p.x[3 * (size_t) j + 1] - xi1,
                       dz = p.x[3 * (size_t) j + 2] - xi2;
          in = !(dx * dx + dy * dy + dz * dz > p.cutsq);   // pair_mtp.cpp:123
        }
      }
      const unsigned long long m = __ballot(in);
      if (in) w.cj[cnt + __popcll(m & ((1ull << lane) - 1ull))] = j;
      cnt += __popcll(m);
    }
    wave_fence();

    // ---- 2+3. tiles: tables, then basic moments in registers --------------------------------
    double acc[KB];
#pragma unroll
    for (int t = 0; t < KB; t++) acc[t] = 0.0;
    const int ntiles = (cnt + p.NT - 1) / p.NT;
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * p.NT, nt = min(p.NT, cnt - t0);
      build_tile(p, w, t0, nt, xi0, xi1, xi2, itype, lane);
      for (int n = 0; n < nt; n++) {
        const double *rec = w.tab + (size_t) n * p.stride;
#pragma unroll
        for (int t = 0; t < KB; t++) acc[t] += rec[og[t]] * (rec[oxa[t]] * (rec[oyb[t]] * rec[ozc[t]]));
      }
      if (ntiles > 1) wave_fence();
    }
    // moments + adjoints into LDS
    for (int m = p.B + lane; m < p.A; m += 64) w.M[m] = 0.0;
    for (int m = lane; m < p.A; m += 64) w.D[m] = 0.0;
#pragma unroll
    for (int t = 0; t < KB; t++)
      if (kval[t]) w.M[lane + 64 * t] = acc[t];
    wave_fence();

    // ---- 4a. products, level by level (pair_mtp.cpp:196-201) ----------------------------------
    for (int l = 0; l < p.nlevels; l++) {
      for (int r = p.level_off[l] + lane; r < p.level_off[l + 1]; r += 64) {
        const int4 row = p.rows[r];
        lds_add(&w.M[row.w], (double) row.z * w.M[row.x] * w.M[row.y]);
      }
      wave_fence();
    }
    // ---- site energy (pair_mtp.cpp:204-212) ---------------------------------------------------
    double e = 0.0;
    for (int k = lane; k < p.S; k += 64) e += p.lin[k] * w.M[p.map[k]];
    e = wave_sum(e) + p.species_coeffs[itype];
    // ---- 4b. adjoints (pair_mtp.cpp:217-233) ---------------------------------------------------
    for (int k = lane; k < p.nseed; k += 64) w.D[p.seed_idx[k]] = p.seed_val[k];
    wave_fence();
    for (int l = p.nlevels - 1; l >= 0; l--) {
      for (int r = p.level_off[l] + lane; r < p.level_off[l + 1]; r += 64) {
        const int4 row = p.rows[r];
        const double d3 = w.D[row.w] * (double) row.z;
        lds_add(&w.D[row.y], d3 * w.M[row.x]);
        lds_add(&w.D[row.x], d3 * w.M[row.y]);
      }
      wave_fence();
    }

    // ---- 5. forces ---------------------------------------------------------------------------
    double Dk[KB];
#pragma unroll
    for (int t = 0; t < KB; t++) Dk[t] = kval[t] ? w.D[lane + 64 * t] : 0.0;
    double fi0 = 0, fi1 = 0, fi2 = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
    for (int tile = 0; tile < ntiles; tile++) {
      const int t0 = tile * p.NT, nt = min(p.NT, cnt - t0);
      if (ntiles > 1) build_tile(p, w, t0, nt, xi0, xi1, xi2, itype, lane);
      for (int g0 = 0; g0 < nt; g0 += 16) {
        double part[64];
#pragma unroll
        for (int nn = 0; nn < 16; nn++) {
          double sr = 0, sx = 0, sy = 0, sz = 0;
          if (g0 + nn < nt) {
            const double *rec = w.tab + (size_t) (g0 + nn) * p.stride;
#pragma unroll
            for (int t = 0; t < KB; t++) {
              const double g = rec[og[t]], gd = rec[ns + og[t]];
              const double xa = rec[oxa[t]], yb = rec[oyb[t]], zc = rec[ozc[t]];
              const double dxa = rec[oxa[t] + 3 * P], dyb = rec[oyb[t] + 3 * P], dzc = rec[ozc[t] + 3 * P];
              const double yz = yb * zc, Dg = Dk[t] * g;
              sr += (Dk[t] * gd) * (xa * yz);
              sx += Dg * (dxa * yz);
              sy += Dg * (xa * (dyb * zc));
              sz += Dg * (xa * (yb * dzc));
            }
          }
          part[4 * nn + 0] = sr;
          part[4 * nn + 1] = sx;
          part[4 * nn + 2] = sy;
          part[4 * nn + 3] = sz;
        }
        Butterfly<64>::run(part, lane);
        w.red[lane] = part[0];
        wave_fence();
        if (lane < 16 && g0 + lane < nt) {
          const int n = g0 + lane;
          const double sr = w.red[4 * lane] * w.nbi[n];
          const double rx = w.nbx[n], ry = w.nby[n], rz = w.nbz[n];
          const double Fx = sr * rx + w.red[4 * lane + 1];
          const double Fy = sr * ry + w.red[4 * lane + 2];
          const double Fz = sr * rz + w.red[4 * lane + 3];
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
        wave_fence();
      }
    }
    // per-atom totals: lanes 0..15 hold partial sums
    fi0 = wave_sum(fi0);
    fi1 = wave_sum(fi1);
    fi2 = wave_sum(fi2);
    if (p.vflag) {
      v0 = wave_sum(v0);
      v1 = wave_sum(v1);
      v2 = wave_sum(v2);
      v3 = wave_sum(v3);
      v4 = wave_sum(v4);
      v5 = wave_sum(v5);
    }
    if (lane == 0) {
      unsafeAtomicAdd(&p.f[3 * (size_t) i + 0], fi0);   // pair_mtp.cpp:248-250
      unsafeAtomicAdd(&p.f[3 * (size_t) i + 1], fi1);
      unsafeAtomicAdd(&p.f[3 * (size_t) i + 2], fi2);
      if ((p.eflag & 2) && p.eatom) p.eatom[i] = e;
      if (p.eflag & 1) ev_acc[0] += e;
      if (p.vflag) {
        ev_acc[1] += v0;
        ev_acc[2] += v1;
        ev_acc[3] += v2;
        ev_acc[4] += v3;
        ev_acc[5] += v4;
        ev_acc[6] += v5;
        if ((p.vflag & 4) && p.vatom) {
          double *va = p.vatom + 6 * (size_t) i;
          va[0] += v0;
          va[1] += v1;
          va[2] += v2;
          va[3] += v3;
          va[4] += v4;
          va[5] += v5;
        }
      }
    }
    wave_fence();
  }
  if (lane == 0 && ((p.eflag & 1) || p.vflag)) {
    double *slot = p.ev_slots + 8 * (size_t) ((blockIdx.x * wpb + wave) % MTP_EV_SLOTS);
#pragma unroll
    for (int q = 0; q < 7; q++)
      if (ev_acc[q] != 0.0) unsafeAtomicAdd(&slot[q], ev_acc[q]);
  }
}

// folds the per-wave tally slots into ev[7] (accumulating) and clears them
__global__ void mtp_ev_finish(double *ev_slots, double *ev)
{
  const int q = blockIdx.x;   // 0..6
  double s = 0.0;
  for (int k = threadIdx.x; k < MTP_EV_SLOTS; k += 64) {
    s += ev_slots[8 * (size_t) k + q];
    ev_slots[8 * (size_t) k + q] = 0.0;
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) ev[q] += s;
}

}   // namespace

template <int KB> static hipError_t launch_kb(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mtp_wave_kernel<KB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(mtp_wave_kernel<KB>, dim3(grid), dim3(64 * wpb), lds, st, p);
  return hipGetLastError();
}

hipError_t mtp_launch_wave_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st)
{
  const int kb = (p.B + 63) / 64;
  switch (kb) {
    case 1: return launch_kb<1>(p, grid, wpb, lds, st);
    case 2: return launch_kb<2>(p, grid, wpb, lds, st);
    case 3: return launch_kb<3>(p, grid, wpb, lds, st);
    case 4: return launch_kb<4>(p, grid, wpb, lds, st);
    case 5: return launch_kb<5>(p, grid, wpb, lds, st);
    case 6: return launch_kb<6>(p, grid, wpb, lds, st);
    case 7: return launch_kb<7>(p, grid, wpb, lds, st);
    case 8: return launch_kb<8>(p, grid, wpb, lds, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t mtp_launch_ev_finish(double *ev_slots, double *ev, hipStream_t st)
{
  hipLaunchKernelGGL(mtp_ev_finish, dim3(7), dim3(64), 0, st, ev_slots, ev);
  return hipGetLastError();
}
