// Device helpers shared by the wavefront-per-atom and the workgroup-per-atom kernels (internal).
#pragma once

#include <hip/hip_runtime.h>

#include "mtp_device.hpp"

#define MTP_NEIGHMASK 0x1FFFFFFF   // LAMMPS NEIGHMASK (pair_mtp.cpp:114)

// Diagnostic build only (make stamps): per-phase s_memtime sums per launch into p.stamps[16].
// The shipped library is built without MTP_STAMPS and executes no stamp.
#ifdef MTP_STAMPS
#define STAMP(k)                                                         \
  do {                                                                   \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
    st_acc[k] += _t - st_prev;                                           \
    st_prev = _t;                                                        \
  } while (0)
#else
#define STAMP(k) ((void) 0)
#endif


static __device__ __forceinline__ void wave_fence()
{
  // LDS operations of one wavefront execute in program order; this only stops the
  // compiler from moving LDS accesses across a phase boundary.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// r = sqrt(r2) and inv = 1 / r from ONE v_rsq_f64 (r2 > 0, normal range: squared interatomic distances): two
// Newton steps on y ~ r2^-1/2 (relative error 2^-26 -> 2^-52 -> rounding), then r = r2 y with one correction step.
// Both results are within 1 ulp of the correctly rounded values the reference's std::sqrt and division give
// (pair_mtp.cpp:128-129) -- 1e-16 relative, seven orders inside the parity tolerance -- at 11 instead of 24 VALU
// instructions (sqrt expansion + division expansion).
static __device__ __forceinline__ void sqrt_and_inverse(double r2, double &r, double &inv)
{
  double y = __builtin_amdgcn_rsq(r2);
  const double h = 0.5 * r2;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  double s = r2 * y;
  s = fma(fma(-s, s, r2), 0.5 * y, s);
  r = s;
  inv = y;
}

static __device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

// a * b for per-lane operands below 2^23 (table rows, slots, byte offsets inside one LDS image): v_mul_i32_i24 /
// v_mad_i32_i24 issue at full rate, the 32- and 64-bit integer multiplies (v_mul_lo_u32, v_mad_u64_u32) at a quarter
static __device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// &base[3 j] for an [n][3] double array: 24 j as shift-adds in 32 bits (j < 2^27 atoms), added as an unsigned offset
static __device__ __forceinline__ const double *row3(const double *base, int j)
{
  unsigned j2 = (unsigned) j << 1;
  asm volatile("" : "+v"(j2));   // (keeps the optimiser from folding the shift-adds back into a quarter-rate v_mul_lo_u32)
  const unsigned j3 = j2 + (unsigned) j;
  return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (size_t) (j3 << 3));
}

// Requests every 64-byte line of a kernel's argument block at once.  The block is written by the host for each launch, so
// its lines miss every cache the first time a wavefront of a CU reads them; the kernel body reads its fields where it
// needs them, which made those misses SERIAL (a dozen lines, one memory round trip each: measured as +9 us per launch
// with the block in host memory against device memory).  After this the body's reads hit the scalar cache.
template <int BYTES, class PTR> static __device__ __forceinline__ void kernarg_touch(PTR ka)
{
  static_assert(BYTES <= 16 * 64, "kernarg_touch: up to sixteen lines");
  unsigned t;   // (ONE statement: no load may still be in flight when the compiler gets the register back)
  constexpr int LAST = BYTES - 4;   // lines past the end of the block re-read its last word
#define MTP_KA_OFF(k) "i"((k) * 64 < LAST ? (k) * 64 : LAST)
  asm volatile("s_load_dword %0, %1, %2\n\ts_load_dword %0, %1, %3\n\ts_load_dword %0, %1, %4\n\ts_load_dword %0, %1, %5\n\t"
               "s_load_dword %0, %1, %6\n\ts_load_dword %0, %1, %7\n\ts_load_dword %0, %1, %8\n\ts_load_dword %0, %1, %9\n\t"
               "s_load_dword %0, %1, %10\n\ts_load_dword %0, %1, %11\n\ts_load_dword %0, %1, %12\n\ts_load_dword %0, %1, %13\n\t"
               "s_load_dword %0, %1, %14\n\ts_load_dword %0, %1, %15\n\ts_load_dword %0, %1, %16\n\ts_load_dword %0, %1, %17\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(t)
               : "s"(ka), MTP_KA_OFF(0), MTP_KA_OFF(1), MTP_KA_OFF(2), MTP_KA_OFF(3), MTP_KA_OFF(4), MTP_KA_OFF(5), MTP_KA_OFF(6),
                 MTP_KA_OFF(7), MTP_KA_OFF(8), MTP_KA_OFF(9), MTP_KA_OFF(10), MTP_KA_OFF(11), MTP_KA_OFF(12), MTP_KA_OFF(13),
                 MTP_KA_OFF(14), MTP_KA_OFF(15)
               : "memory");
#undef MTP_KA_OFF
}

static __device__ __forceinline__ double uniform_f64(double v)   // v is wave-uniform: move it to SGPRs
{
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (b >> 32));
  return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

// Cross-lane moves inside a row of 16 lanes on the VALU (DPP), no LDS traffic: the partner of lane l is
// l^1, l^2 (quad permutes), l^7 (row_half_mirror) or l^15 (row_mirror).  Any of them flips exactly the
// lane bit a butterfly step splits on (bit 0, 1, 2, 3), which is all the reductions below need.
template <int CTRL> static __device__ __forceinline__ double dpp_f64(double v)
{
  const long long b = __double_as_longlong(v);
  int lo = (int) b, hi = (int) (b >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo));
}
// value of the lane that differs in bit log2(H) (and possibly lower bits).  H = 16 / 32 use gfx950's row / half
// swaps (lane maps measured with scripts/probes/permlane_probe.hip): with both operands equal to v,
// v_permlane16_swap leaves rows {0,0,2,2} of v in the first register and rows {1,1,3,3} in the second.
template <int H> static __device__ __forceinline__ double partner_f64(double v, int lane = 0)
{
  if constexpr (H == 1) return dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (H == 2) return dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (H == 4) return dpp_f64<0x141>(v);  // row_half_mirror
  else if constexpr (H == 8) return dpp_f64<0x140>(v);  // row_mirror
  else {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned) b, hi = (unsigned) (b >> 32);
    if constexpr (H == 16) {
      const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      const unsigned rl = (lane & 16) ? a[0] : a[1], rh = (lane & 16) ? c[0] : c[1];
      return __longlong_as_double((long long) (((unsigned long long) rh << 32) | rl));
    } else {
      static_assert(H == 32, "partner distance");
      const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      const unsigned rl = (lane & 32) ? a[0] : a[1], rh = (lane & 32) ? c[0] : c[1];
      return __longlong_as_double((long long) (((unsigned long long) rh << 32) | rl));
    }
  }
}

// v + (value of lane l^16) and v + (value of lane l^32) with gfx950's row / half swaps: with both operands equal
// to v, v_permlane16_swap leaves {rows 0,0,2,2} in one register and {rows 1,1,3,3} in the other (measured
// lane maps: scripts/probes/permlane_probe.hip), so their sum is the pair sum in every lane, in the same
// operand order everywhere; v_permlane32_swap does the same for the wavefront halves.  No LDS traffic.
static __device__ __forceinline__ double pair_sum16(double v)
{
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned) b, hi = (unsigned) (b >> 32);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double((long long) (((unsigned long long) c[0] << 32) | a[0])) +
      __longlong_as_double((long long) (((unsigned long long) c[1] << 32) | a[1]));
}
static __device__ __forceinline__ double pair_sum32(double v)
{
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned) b, hi = (unsigned) (b >> 32);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double((long long) (((unsigned long long) c[0] << 32) | a[0])) +
      __longlong_as_double((long long) (((unsigned long long) c[1] << 32) | a[1]));
}

static __device__ __forceinline__ double wave_sum(double v)   // same bits in every lane
{
  v += partner_f64<1>(v);
  v += partner_f64<2>(v);
  v += partner_f64<4>(v);
  v += partner_f64<8>(v);   // every lane: sum of its row of 16
  return pair_sum32(pair_sum16(v));
}

static __device__ __forceinline__ void lds_add(double *p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS read at (32-bit LDS byte address + compile-time byte offset): the offset lands in the
// ds_read immediate field, so inner loops spend no VALU on addressing.
typedef __attribute__((address_space(3))) const double lds_cdouble;
// (no generic -> LDS pointer casts: hipcc 7.2 miscompiles their null check on gfx950; LDS byte
// addresses are formed as the LDS address of the dynamic array + an offset instead)
// volatile: hipcc otherwise fuses neighbouring reads into ds_read2_b64, which moves 16 B per lane in
// 8 LDS cycles where two ds_read_b64 take 4 (MI355X_MICROARCH.md, LDS table)
static __device__ __forceinline__ double lds_ld(unsigned base, int off_doubles)   // off: constant after unrolling
{
  return *(volatile lds_cdouble *) (size_t) (base + 8u * (unsigned) off_doubles);
}

// Butterfly transpose-reduce over the lane bits below N: every lane enters with N partial
// sums v[0..N); on exit v[0] of lane l holds entry (l mod N) summed over the N lanes that
// differ from l only in those bits (steps within a row of 16 lanes run on DPP).  N-1 adds and N-1 exchanges instead of N*log2(N).
template <int N> struct Butterfly {
  static __device__ __forceinline__ void run(double *v, int lane)
  {
    constexpr int H = N / 2;
    const bool hi = (lane & H) != 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
      const double keep = hi ? v[i + H] : v[i];
      const double send = hi ? v[i] : v[i + H];
      v[i] = keep + partner_f64<H>(send, lane);
    }
    Butterfly<H>::run(v, lane);
  }
};
template <> struct Butterfly<1> {
  static __device__ __forceinline__ void run(double *, int) {}
};

// The same for NINE entries v[0..8] over the lane bits below 16 (the force totals: 3 force and 6 virial components):
// entry 8 pairs with entry 0 in the first step; entries 1..7 have no partner entry there, so that step is a plain
// pair sum for them (3 instead of 7 instructions each).  On exit v[0] of lane l holds entry l for l & 15 in 0..8.
static __device__ __forceinline__ void butterfly9(double *v, int lane)
{
  const bool hi = (lane & 8) != 0;
  {
    const double keep = hi ? v[8] : v[0];
    const double send = hi ? v[0] : v[8];
    v[0] = keep + partner_f64<8>(send, lane);
  }
#pragma unroll
  for (int i = 1; i < 8; i++) v[i] += partner_f64<8>(v[i], lane);
  Butterfly<8>::run(v, lane);
}

struct BlockTables {   // views into the workgroup-shared head of LDS
  const MtpRow8 *rows;
  const int *level, *slot, *seed_idx, *map, *pack, *coef, *smu, *fwd, *seg_fwd, *seg_bwd;
  const double *radial, *seed_val, *lin, *leaf_cf, *leaf_cb;
};

