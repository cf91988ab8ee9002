// Kernel parameter block shared by the host context and the HIP kernels (internal).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#define MTP_EV_SLOTS 1024   // per-wave energy/virial tally slots (8 doubles each)

struct MtpDevParams {
  // potential (device pointers)
  int Sp, R, Mu, P, A, B, T, S, C;
  int nslot, nlevels, nseed;
  double rmin, rmax, scaling, cutsq;
  const double *radial_coeffs;   // [(t1*Sp+t2)*Mu*R + mu*R + ri]
  const int *basic_pack;         // [B] slot | a<<8 | b<<12 | c<<16
  const int *slot_of;            // [Mu*P]
  const int4 *rows;              // [T] {a0,a1,mult,a3} sorted by dependency level
  const int *level_off;          // [nlevels+1]
  const int *seed_idx;           // [nseed]
  const double *seed_val;
  const int *map;                // [S]
  const double *lin;             // [S]
  const double *species_coeffs;  // [Sp]
  const double *inv_active;      // [C][C] or null
  // system
  int inum, nall;
  const int *ilist, *first, *neigh;
  const double *x;               // [nall][3]
  const int *type;               // [nall] 1-based
  // outputs
  double *f;                     // [nall][3] accumulated
  double *eatom;                 // [nall] or null
  double *vatom;                 // [nall][6] or null
  double *ev_slots;              // [MTP_EV_SLOTS][8]
  double *grades;                // [nall] or null
  double *max_grade;             // [1] or null
  double *coeff_ders;            // [C] or null
  int *err_flag;
  int eflag, vflag, grade_flag;
  // launch geometry
  int NT;                        // neighbours per LDS tile (multiple of 16)
  int stride;                    // doubles per neighbour record = 2*nslot + 6*P
  int cj_cap;                    // capacity of the compacted id list
  int wave_doubles;              // LDS doubles per wavefront
};

hipError_t mtp_launch_wave_kernel(const MtpDevParams &p, int grid, int wpb, size_t lds, hipStream_t st);
hipError_t mtp_launch_ev_finish(double *ev_slots, double *ev, hipStream_t st);
