// Device-resident pieces of a standalone MD step either side of the force call --
// what LAMMPS core does around Pair::compute and the reference therefore does not contain: the periodic ghost
// images of one GPU's own atoms (Comm::borders for one rank: rebuilt at every re-neighbouring), their per-step
// refresh and the fold of their forces onto the owners (Comm::forward_comm / reverse_comm; the pair style writes
// forces onto ghosts, /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:252-254, 315), and the two halves of a
// velocity-Verlet step (fix nve).  Positions, velocities, forces, ghost maps and the neighbour list stay in HBM;
// the host sees one integer (the ghost count) per re-neighbouring.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <new>
#include <string>

#include "../../include/mtp_mi355x.h"
#include "mtp_device.hpp"

namespace {

struct Box3 {
  double len[3], rg;
};

// number of periodic images (other than the atom itself) of wrapped position p that fall inside the shell
// [-rg, len + rg) around the box: per direction the atom has an image above the box when p < rg and one below when
// p >= len - rg (the bounds of lammps_mtp_kokkos_amd/driver.py make_ghosts)
__device__ __forceinline__ int image_flags(const Box3 &b, const double *p, int lo[3], int hi[3])
{
  int n = 1;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = p[a] >= b.len[a] - b.rg ? -1 : 0;   // shift -1 allowed
    hi[a] = p[a] < b.rg ? 1 : 0;                // shift +1 allowed
    n *= 1 + hi[a] - lo[a];
  }
  return n - 1;
}

// wraps the owned atoms into [0, len) and counts their ghost images
__global__ void __launch_bounds__(256) ghosts_count_kernel(Box3 b, double *__restrict__ x, int n, int *__restrict__ count)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double p[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double v = x[3 * (size_t) i + a];
    v -= floor(v / b.len[a]) * b.len[a];
    if (v >= b.len[a]) v -= b.len[a];   // floor() rounding at the upper edge
    p[a] = v;
    x[3 * (size_t) i + a] = v;
  }
  int lo[3], hi[3];
  count[i] = image_flags(b, p, lo, hi);
}

// ghost k of atom i (images in lexicographic shift order) -> owner[k], shift[k]
__global__ void __launch_bounds__(256) ghosts_fill_kernel(Box3 b, const double *__restrict__ x, int n,
                                                         const int *__restrict__ first, int *__restrict__ owner,
                                                         double *__restrict__ shift, int capacity)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {x[3 * (size_t) i], x[3 * (size_t) i + 1], x[3 * (size_t) i + 2]};
  int lo[3], hi[3];
  (void) image_flags(b, p, lo, hi);
  int k = first[i];
  for (int sx = lo[0]; sx <= hi[0]; sx++)
    for (int sy = lo[1]; sy <= hi[1]; sy++)
      for (int sz = lo[2]; sz <= hi[2]; sz++) {
        if (sx == 0 && sy == 0 && sz == 0) continue;
        if (k < capacity) {
          owner[k] = i;
          shift[3 * (size_t) k] = sx * b.len[0];
          shift[3 * (size_t) k + 1] = sy * b.len[1];
          shift[3 * (size_t) k + 2] = sz * b.len[2];
        }
        k++;
      }
}

// x[nlocal + k] = x[owner[k]] + shift[k]   (one lane per coordinate)
__global__ void __launch_bounds__(256) ghosts_forward_kernel(double *__restrict__ x, int nlocal,
                                                            const int *__restrict__ owner,
                                                            const double *__restrict__ shift, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const int k = e / 3, c = e - 3 * k;
  x[3 * (size_t) nlocal + e] = x[3 * (size_t) owner[k] + c] + shift[e];
}

// f[owner[k]] += f[nlocal + k]
__global__ void __launch_bounds__(256) ghosts_reverse_kernel(double *__restrict__ f, int nlocal,
                                                            const int *__restrict__ owner, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const int k = e / 3, c = e - 3 * k;
  unsafeAtomicAdd(&f[3 * (size_t) owner[k] + c], f[3 * (size_t) nlocal + e]);
}

__global__ void __launch_bounds__(256) ghosts_types_kernel(int *__restrict__ type, int nlocal, const int *__restrict__ owner,
                                                          int nghost)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < nghost) type[nlocal + k] = type[owner[k]];
}

// fix nve, first half: v += dtf f / m; x += dt v.  second half: v += dtf f / m.  (metal units: dtf = dt/2 * ftm2v)
__global__ void __launch_bounds__(256) nve_initial_kernel(double *__restrict__ x, double *__restrict__ v,
                                                         const double *__restrict__ f, const int *__restrict__ type,
                                                         const double *__restrict__ inv_mass, double dtf, double dt, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const double vv = v[e] + dtf * inv_mass[type[e / 3] - 1] * f[e];
  v[e] = vv;
  x[e] += dt * vv;
}
__global__ void __launch_bounds__(256) nve_final_kernel(double *__restrict__ v, const double *__restrict__ f,
                                                       const int *__restrict__ type, const double *__restrict__ inv_mass,
                                                       double dtf, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  v[e] += dtf * inv_mass[type[e / 3] - 1] * f[e];
}

// largest squared displacement since the last re-neighbouring and the kinetic energy sum m v^2, by 64-bit atomics
// on the non-negative fp64 bit pattern (max) / fp64 add
__global__ void __launch_bounds__(256) nve_monitor_kernel(const double *__restrict__ x, const double *__restrict__ x_ref,
                                                         const double *__restrict__ v, const int *__restrict__ type,
                                                         const double *__restrict__ mass, int n, double *__restrict__ out)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  double d2 = 0.0, ke = 0.0;
  if (i < n) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double d = x[3 * (size_t) i + a] - x_ref[3 * (size_t) i + a];
      d2 += d * d;
      ke += v[3 * (size_t) i + a] * v[3 * (size_t) i + a];
    }
    ke *= mass[type[i] - 1];
  }
  typedef hipcub::BlockReduce<double, 256> Red;
  __shared__ typename Red::TempStorage tmp;
  const double bmax = Red(tmp).Reduce(d2, hipcub::Max());
  __syncthreads();
  const double bsum = Red(tmp).Sum(ke);
  if (threadIdx.x == 0) {
    atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long) __double_as_longlong(bmax));
    unsafeAtomicAdd(out + 1, bsum);
  }
}

}   // namespace

struct mtp_ghosts {
  int device = 0, nlocal = 0, nghost = 0;
  int cap_local = 0, cap_ghost = 0;
  int *d_count = nullptr, *d_first = nullptr, *d_owner = nullptr;
  double *d_shift = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  std::string last_error;
  ~mtp_ghosts()
  {
    (void) hipSetDevice(device);
    for (void *p : {(void *) d_count, (void *) d_first, (void *) d_owner, (void *) d_shift, d_tmp})
      if (p) (void) hipFree(p);
  }
};

#define MD_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      g->last_error = std::string(#call) + ": " + hipGetErrorString(_e);                    \
      return MTP_ERR_DEVICE;                                                                \
    }                                                                                       \
  } while (0)

static int ghosts_null_stream(mtp_ghosts *g, const char *fn)
{
  g->last_error = std::string(fn) + ": a NULL stream is not accepted (no context to take a stream from)";
  return MTP_ERR_ARG;
}

extern "C" {

int mtp_ghosts_create(int device_id, mtp_ghosts **out)
{
  if (!out) return MTP_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return MTP_ERR_DEVICE;
  mtp_ghosts *g = new (std::nothrow) mtp_ghosts();
  if (!g) return MTP_ERR_ARG;
  g->device = device_id;
  *out = g;
  return MTP_OK;
}

void mtp_ghosts_destroy(mtp_ghosts *g) { delete g; }

const char *mtp_ghosts_last_error(const mtp_ghosts *g) { return g ? g->last_error.c_str() : "null ghosts"; }

int mtp_ghosts_build(mtp_ghosts *g, void *stream, double *d_x, int nlocal, int capacity, const double box[3],
                     double rghost, int *nall_out)
{
  if (!g || !d_x || nlocal < 0 || capacity < nlocal || !box || !(rghost > 0.0) || !nall_out) return MTP_ERR_ARG;
  if (!stream) return ghosts_null_stream(g, "mtp_ghosts_build");
  for (int a = 0; a < 3; a++)
    if (!(box[a] >= rghost)) {   // one image per direction and sign: the shell must not be thicker than the box
      g->last_error = "mtp_ghosts_build: box edge shorter than the ghost cutoff";
      return MTP_ERR_LIMIT;
    }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  MD_HIP(hipSetDevice(g->device));
  if (nlocal + 1 > g->cap_local) {
    if (g->d_count) (void) hipFree(g->d_count);
    if (g->d_first) (void) hipFree(g->d_first);
    g->d_count = g->d_first = nullptr;
    const size_t n = (size_t) nlocal + 1 + nlocal / 8;
    MD_HIP(hipMalloc((void **) &g->d_count, n * sizeof(int)));
    MD_HIP(hipMalloc((void **) &g->d_first, n * sizeof(int)));
    g->cap_local = (int) n;
    size_t need = 0;
    (void) hipcub::DeviceScan::ExclusiveSum(nullptr, need, g->d_count, g->d_first, (int) n, st);
    if (need > g->tmp_bytes) {
      if (g->d_tmp) (void) hipFree(g->d_tmp);
      g->d_tmp = nullptr;
      MD_HIP(hipMalloc(&g->d_tmp, need));
      g->tmp_bytes = need;
    }
  }
  Box3 b{{box[0], box[1], box[2]}, rghost};
  const int nb = (nlocal + 255) / 256;
  MD_HIP(hipMemsetAsync(g->d_count, 0, ((size_t) nlocal + 1) * sizeof(int), st));
  if (nlocal > 0) hipLaunchKernelGGL(ghosts_count_kernel, dim3(nb), dim3(256), 0, st, b, d_x, nlocal, g->d_count);
  size_t tb = g->tmp_bytes;
  MD_HIP(hipcub::DeviceScan::ExclusiveSum(g->d_tmp, tb, g->d_count, g->d_first, nlocal + 1, st));
  int total = 0;
  MD_HIP(hipMemcpyAsync(&total, g->d_first + nlocal, sizeof(int), hipMemcpyDeviceToHost, st));
  MD_HIP(hipStreamSynchronize(st));
  if (total > g->cap_ghost) {
    if (g->d_owner) (void) hipFree(g->d_owner);
    if (g->d_shift) (void) hipFree(g->d_shift);
    g->d_owner = nullptr;
    g->d_shift = nullptr;
    const size_t n = (size_t) total + total / 8 + 64;
    MD_HIP(hipMalloc((void **) &g->d_owner, n * sizeof(int)));
    MD_HIP(hipMalloc((void **) &g->d_shift, 3 * n * sizeof(double)));
    g->cap_ghost = (int) n;
  }
  g->nlocal = nlocal;
  g->nghost = total;
  *nall_out = nlocal + total;
  if (nlocal + total > capacity) {   // the caller's arrays are too short: sizes are reported, nothing is written
    g->last_error = "mtp_ghosts_build: capacity of the position array is smaller than owned + ghost atoms";
    return MTP_ERR_LIMIT;
  }
  if (nlocal > 0 && total > 0) {
    hipLaunchKernelGGL(ghosts_fill_kernel, dim3(nb), dim3(256), 0, st, b, d_x, nlocal, g->d_first, g->d_owner, g->d_shift,
                       g->cap_ghost);
    hipLaunchKernelGGL(ghosts_forward_kernel, dim3((3 * total + 255) / 256), dim3(256), 0, st, d_x, nlocal, g->d_owner,
                       g->d_shift, 3 * total);
  }
  MD_HIP(hipGetLastError());
  return MTP_OK;
}

int mtp_ghosts_forward(mtp_ghosts *g, void *stream, double *d_x)
{
  if (!g || !d_x) return MTP_ERR_ARG;
  if (!stream) return ghosts_null_stream(g, "mtp_ghosts_forward");
  if (g->nghost > 0)
    hipLaunchKernelGGL(ghosts_forward_kernel, dim3((3 * g->nghost + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_x, g->nlocal, g->d_owner, g->d_shift, 3 * g->nghost);
  MD_HIP(hipGetLastError());
  return MTP_OK;
}

int mtp_ghosts_reverse(mtp_ghosts *g, void *stream, double *d_f)
{
  if (!g || !d_f) return MTP_ERR_ARG;
  if (!stream) return ghosts_null_stream(g, "mtp_ghosts_reverse");
  if (g->nghost > 0)
    hipLaunchKernelGGL(ghosts_reverse_kernel, dim3((3 * g->nghost + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_f, g->nlocal, g->d_owner, 3 * g->nghost);
  MD_HIP(hipGetLastError());
  return MTP_OK;
}

int mtp_ghosts_reverse_finish(mtp_ghosts *g, mtp_context *ctx, void *stream, int eflag, int vflag, double *d_f, double *d_ev)
{
  if (!g || !ctx || !d_f) return MTP_ERR_ARG;
  // the tally fold of a force call made with finish_tallies = 0 rides in the launch that folds the ghost forces
  // (a context is at hand: NULL -> the context's stream, as in mtp_compute_device)
  return mtp_internal_finish_unpack(ctx, stream, eflag, vflag, d_ev, d_f, g->d_owner, d_f + 3 * (size_t) g->nlocal, 3 * g->nghost);
}

int mtp_ghosts_types(mtp_ghosts *g, void *stream, int *d_type)
{
  if (!g || !d_type) return MTP_ERR_ARG;
  if (!stream) return ghosts_null_stream(g, "mtp_ghosts_types");
  if (g->nghost > 0)
    hipLaunchKernelGGL(ghosts_types_kernel, dim3((g->nghost + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_type, g->nlocal, g->d_owner, g->nghost);
  MD_HIP(hipGetLastError());
  return MTP_OK;
}

int mtp_nve_initial(void *stream, int nlocal, double *d_x, double *d_v, const double *d_f, const int *d_type,
                    const double *d_inv_mass, double dtf, double dt)
{
  if (!stream || nlocal < 0 || (nlocal > 0 && (!d_x || !d_v || !d_f || !d_type || !d_inv_mass))) return MTP_ERR_ARG;
  if (nlocal > 0)
    hipLaunchKernelGGL(nve_initial_kernel, dim3((3 * nlocal + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_x, d_v, d_f, d_type, d_inv_mass, dtf, dt, 3 * nlocal);
  return hipGetLastError() == hipSuccess ? MTP_OK : MTP_ERR_DEVICE;
}

int mtp_nve_final(void *stream, int nlocal, double *d_v, const double *d_f, const int *d_type,
                  const double *d_inv_mass, double dtf)
{
  if (!stream || nlocal < 0 || (nlocal > 0 && (!d_v || !d_f || !d_type || !d_inv_mass))) return MTP_ERR_ARG;
  if (nlocal > 0)
    hipLaunchKernelGGL(nve_final_kernel, dim3((3 * nlocal + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_v, d_f, d_type, d_inv_mass, dtf, 3 * nlocal);
  return hipGetLastError() == hipSuccess ? MTP_OK : MTP_ERR_DEVICE;
}

int mtp_nve_monitor(void *stream, int nlocal, const double *d_x, const double *d_x_ref, const double *d_v,
                    const int *d_type, const double *d_mass, double *d_out2)
{
  if (!stream || nlocal < 0 || !d_out2 || (nlocal > 0 && (!d_x || !d_x_ref || !d_v || !d_type || !d_mass))) return MTP_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(d_out2, 0, 2 * sizeof(double), st) != hipSuccess) return MTP_ERR_DEVICE;
  if (nlocal > 0)
    hipLaunchKernelGGL(nve_monitor_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, st, d_x, d_x_ref, d_v, d_type, d_mass,
                       nlocal, d_out2);
  return hipGetLastError() == hipSuccess ? MTP_OK : MTP_ERR_DEVICE;
}

}   // extern "C"
