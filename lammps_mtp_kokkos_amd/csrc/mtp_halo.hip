// Ghost-atom halo exchange of the domain-decomposed force call, inside the library: device pack / unpack
// kernels and direct RCCL point-to-point groups over xGMI (include/mtp_mi355x.h, "multi-GPU halo").
//
// What LAMMPS' Comm::forward_comm / reverse_comm do around the pair style (the reference relies on them:
// /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:252-254 writes forces onto ghosts, :315 demands newton_pair on):
//   forward   ghost positions  <- owners' current positions + periodic shift
//   reverse   owners' forces   += forces the pair style left on the ghosts
// One process per GPU.  Every rank talks to every rank it shares a face, edge or corner with in ONE grouped
// exchange per direction (ncclGroupStart ... ncclSend / ncclRecv per peer ... ncclGroupEnd): on a fully connected
// xGMI node all seven links carry traffic in a single latency stage, instead of LAMMPS' x -> y -> z staging.
// The exchange runs on the halo's own stream, tied to the caller's stream by events, so force work that needs no
// ghosts (interior atoms) overlaps it (begin / end pairs).  Images of a rank's own atoms (periodic directions
// with one rank) take the same path: a send to and a receive from itself inside the group.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mtp_mi355x.h"
#include "mtp_device.hpp"

namespace {

struct HaloFail {
  std::string what;
};
#define HIP_OK(call)                                                                   \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) throw HaloFail{std::string(#call) + ": " + hipGetErrorString(_e)}; \
  } while (0)
#define NCCL_OK(call)                                                                   \
  do {                                                                                  \
    ncclResult_t _r = (call);                                                           \
    if (_r != ncclSuccess) throw HaloFail{std::string(#call) + ": " + ncclGetErrorString(_r)}; \
  } while (0)

// sendbuf[k] = x[send_idx[k]] + send_shift[k]: one lane per coordinate (consecutive lanes write consecutive
// doubles; the gather side reads three consecutive doubles per atom)
__global__ void __launch_bounds__(256) halo_pack_kernel(const double *__restrict__ x, const int *__restrict__ idx,
                                                       const double *__restrict__ shift, double *__restrict__ out,
                                                       int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const int k = e / 3, c = e - 3 * k;
  out[e] = x[3 * (size_t) idx[k] + c] + shift[e];
}

// The same with the step's force array zeroed beside it (mtp_halo_force_step: one launch instead of two; f is 16-byte
// aligned, nz doubles, nz - 2 * (nz / 2) in {0, 1})
__global__ void __launch_bounds__(256) halo_pack_zero_kernel(const double *__restrict__ x, const int *__restrict__ idx,
                                                            const double *__restrict__ shift, double *__restrict__ out,
                                                            int n3, double *__restrict__ f, size_t nz)
{
  const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (e < (size_t) n3) {
    const int k = (int) e / 3, c = (int) e - 3 * k;
    out[e] = x[3 * (size_t) idx[k] + c] + shift[e];
  }
  if (e < nz / 2) reinterpret_cast<double2 *>(f)[e] = make_double2(0.0, 0.0);
  if (e == 0 && (nz & 1)) f[nz - 1] = 0.0;
}

// f[send_idx[k]] += frecv[k]: an owned atom can be a ghost on several peers (and several images), so the adds
// are fp64 HBM atomics, as in the force kernel's own scatter
__global__ void __launch_bounds__(256) halo_unpack_add_kernel(double *__restrict__ f, const int *__restrict__ idx,
                                                             const double *__restrict__ in, int n3)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const int k = e / 3, c = e - 3 * k;
  unsafeAtomicAdd(&f[3 * (size_t) idx[k] + c], in[e]);
}

void copy_err(const std::string &s, char *err, int errlen)
{
  if (err && errlen > 0) std::snprintf(err, (size_t) errlen, "%s", s.c_str());
}

}   // namespace

struct mtp_halo {
  int device = 0, nranks = 1, rank = 0;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  bool overlap = false;   // mtp_halo_force_step: rows A | middle | C around the exchanges on the halo's stream (else one stream)
  hipEvent_t ev_fwd_ready = nullptr, ev_fwd_done = nullptr, ev_rev_ready = nullptr, ev_rev_done = nullptr,
             ev_red_ready = nullptr, ev_red_done = nullptr;
  int nlocal = 0, nghost = 0, nsend = 0;
  int *d_send_idx = nullptr;
  double *d_send_shift = nullptr, *d_sendbuf = nullptr, *d_frecv = nullptr;
  std::vector<int> send_counts, recv_counts, send_off, recv_off;
  std::string last_error;
  ~mtp_halo()
  {
    (void) hipSetDevice(device);
    if (comm_stream) (void) hipStreamSynchronize(comm_stream);
    if (comm) (void) ncclCommDestroy(comm);
    for (hipEvent_t e : {ev_fwd_ready, ev_fwd_done, ev_rev_ready, ev_rev_done, ev_red_ready, ev_red_done})
      if (e) (void) hipEventDestroy(e);
    if (comm_stream) (void) hipStreamDestroy(comm_stream);
    for (void *p : {(void *) d_send_idx, (void *) d_send_shift, (void *) d_sendbuf, (void *) d_frecv})
      if (p) (void) hipFree(p);
  }
};

extern "C" {

int mtp_halo_get_unique_id(void *id_out)
{
  if (!id_out) return MTP_ERR_ARG;
  static_assert(MTP_HALO_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return MTP_ERR_DEVICE;
  std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MTP_OK;
}

int mtp_halo_create(int device_id, int nranks, int rank, const void *unique_id, int nlocal, int nghost,
                    const int *send_idx, const double *send_shift, const int *send_counts, const int *recv_counts,
                    mtp_halo **out, char *err, int errlen)
{
  if (!out || !unique_id || nranks < 1 || rank < 0 || rank >= nranks || nlocal < 0 || nghost < 0 || !send_counts ||
      !recv_counts)
    return MTP_ERR_ARG;
  *out = nullptr;
  long long nsend = 0, nrecv = 0;
  for (int q = 0; q < nranks; q++) {
    if (send_counts[q] < 0 || recv_counts[q] < 0) return MTP_ERR_ARG;
    nsend += send_counts[q];
    nrecv += recv_counts[q];
  }
  if (nrecv != nghost) {
    copy_err("mtp_halo_create: recv_counts do not add up to nghost", err, errlen);
    return MTP_ERR_ARG;
  }
  if (nsend > 0 && (!send_idx || !send_shift)) return MTP_ERR_ARG;
  if (3 * nsend > 0x7fffffffll || 3ll * nghost > 0x7fffffffll) {
    copy_err("mtp_halo_create: more than 2^31-1 halo coordinates on one rank", err, errlen);
    return MTP_ERR_LIMIT;
  }
  for (long long k = 0; k < nsend; k++)
    if (send_idx[k] < 0 || send_idx[k] >= nlocal) {   // checked on the host: the kernels index x and f with it
      copy_err("mtp_halo_create: send_idx outside the owned atoms", err, errlen);
      return MTP_ERR_ARG;
    }
  mtp_halo *h = new (std::nothrow) mtp_halo();
  if (!h) return MTP_ERR_ARG;
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
      throw HaloFail{"no such HIP device (libmtp_mi355x has no CPU fallback)"};
    HIP_OK(hipSetDevice(device_id));
    h->device = device_id;
    h->nranks = nranks;
    h->rank = rank;
    h->nlocal = nlocal;
    h->nghost = nghost;
    h->nsend = (int) nsend;
    h->send_counts.assign(send_counts, send_counts + nranks);
    h->recv_counts.assign(recv_counts, recv_counts + nranks);
    h->send_off.assign((size_t) nranks + 1, 0);
    h->recv_off.assign((size_t) nranks + 1, 0);
    for (int q = 0; q < nranks; q++) {
      h->send_off[q + 1] = h->send_off[q] + send_counts[q];
      h->recv_off[q + 1] = h->recv_off[q] + recv_counts[q];
    }
    HIP_OK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (const char *e = std::getenv("MTP_HALO_OVERLAP")) h->overlap = std::atoi(e) != 0;   // tuning default
    for (hipEvent_t *e : {&h->ev_fwd_ready, &h->ev_fwd_done, &h->ev_rev_ready, &h->ev_rev_done, &h->ev_red_ready,
                          &h->ev_red_done})
      HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    const size_t n = (size_t) std::max<long long>(nsend, 1);
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->d_send_idx), n * sizeof(int)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->d_send_shift), 3 * n * sizeof(double)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->d_sendbuf), 3 * n * sizeof(double)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->d_frecv), 3 * n * sizeof(double)));
    if (nsend > 0) {
      HIP_OK(hipMemcpy(h->d_send_idx, send_idx, (size_t) nsend * sizeof(int), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(h->d_send_shift, send_shift, 3 * (size_t) nsend * sizeof(double), hipMemcpyHostToDevice));
    }
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    NCCL_OK(ncclCommInitRank(&h->comm, nranks, id, rank));
    int cnt = 0, me = -1;
    NCCL_OK(ncclCommCount(h->comm, &cnt));
    NCCL_OK(ncclCommUserRank(h->comm, &me));
    if (cnt != nranks || me != rank) throw HaloFail{"RCCL communicator disagrees with the requested rank layout"};
  } catch (const HaloFail &f) {
    copy_err(f.what, err, errlen);
    delete h;
    return MTP_ERR_DEVICE;
  }
  *out = h;
  return MTP_OK;
}

void mtp_halo_destroy(mtp_halo *h) { delete h; }

const char *mtp_halo_last_error(const mtp_halo *h) { return h ? h->last_error.c_str() : "null halo"; }

int mtp_halo_comm_count(const mtp_halo *h, int *nranks, int *rank, int *rccl_version)
{
  if (!h) return MTP_ERR_ARG;
  int cnt = 0, me = 0, ver = 0;
  if (ncclCommCount(h->comm, &cnt) != ncclSuccess || ncclCommUserRank(h->comm, &me) != ncclSuccess ||
      ncclGetVersion(&ver) != ncclSuccess)
    return MTP_ERR_DEVICE;
  if (nranks) *nranks = cnt;
  if (rank) *rank = me;
  if (rccl_version) *rccl_version = ver;
  return MTP_OK;
}

// One grouped exchange on the halo's stream: to every peer `sbuf + 3 soff[q]` (scount[q] atoms), from every peer
// into `rbuf + 3 roff[q]` (rcount[q] atoms).
static void exchange(mtp_halo *h, const double *sbuf, const std::vector<int> &soff, const std::vector<int> &scount,
                     double *rbuf, const std::vector<int> &roff, const std::vector<int> &rcount, hipStream_t on = nullptr)
{
  const hipStream_t st = on ? on : h->comm_stream;
  NCCL_OK(ncclGroupStart());
  for (int q = 0; q < h->nranks; q++) {
    if (scount[q] > 0) NCCL_OK(ncclSend(sbuf + 3 * (size_t) soff[q], 3 * (size_t) scount[q], ncclDouble, q, h->comm, st));
    if (rcount[q] > 0) NCCL_OK(ncclRecv(rbuf + 3 * (size_t) roff[q], 3 * (size_t) rcount[q], ncclDouble, q, h->comm, st));
  }
  NCCL_OK(ncclGroupEnd());
}

int mtp_halo_forward_begin(mtp_halo *h, void *stream, double *d_x)
{
  if (!h || !d_x) return MTP_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    if (h->nsend > 0) {
      const int n3 = 3 * h->nsend;
      hipLaunchKernelGGL(halo_pack_kernel, dim3((n3 + 255) / 256), dim3(256), 0, st, d_x, h->d_send_idx,
                         h->d_send_shift, h->d_sendbuf, n3);
      HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(h->ev_fwd_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_fwd_ready, 0));
    exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts);
    HIP_OK(hipEventRecord(h->ev_fwd_done, h->comm_stream));
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_forward_end(mtp_halo *h, void *stream)
{
  if (!h) return MTP_ERR_ARG;
  if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), h->ev_fwd_done, 0) != hipSuccess) {
    h->last_error = "hipStreamWaitEvent failed (forward halo)";
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_reverse_begin(mtp_halo *h, void *stream, const double *d_f)
{
  if (!h || !d_f) return MTP_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipEventRecord(h->ev_rev_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_rev_ready, 0));
    // the ghost rows of f go back the way the ghost positions came: sizes and peers swap roles
    exchange(h, d_f + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, h->d_frecv, h->send_off, h->send_counts);
    HIP_OK(hipEventRecord(h->ev_rev_done, h->comm_stream));
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_reverse_end(mtp_halo *h, void *stream, double *d_f)
{
  if (!h || !d_f) return MTP_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipStreamWaitEvent(st, h->ev_rev_done, 0));
    if (h->nsend > 0) {
      const int n3 = 3 * h->nsend;
      hipLaunchKernelGGL(halo_unpack_add_kernel, dim3((n3 + 255) / 256), dim3(256), 0, st, d_f, h->d_send_idx,
                         h->d_frecv, n3);
      HIP_OK(hipGetLastError());
    }
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_forward(mtp_halo *h, void *stream, double *d_x)
{
  const int rc = mtp_halo_forward_begin(h, stream, d_x);
  return rc != MTP_OK ? rc : mtp_halo_forward_end(h, stream);
}

int mtp_halo_reverse(mtp_halo *h, void *stream, double *d_f)
{
  const int rc = mtp_halo_reverse_begin(h, stream, d_f);
  return rc != MTP_OK ? rc : mtp_halo_reverse_end(h, stream, d_f);
}

// One domain-decomposed force call with both exchanges overlapped (rows of the installed list ordered interior A |
// boundary | interior C): zero f; forward halo || rows A; boundary rows; reverse halo || rows C; fold.  One entry
// point instead of eight, so a driver's per-call overhead is paid once per step.
int mtp_halo_force_step(mtp_halo *h, mtp_context *ctx, void *stream, int rows_a, int rows_b, int rows_c, double *d_x,
                        const int *d_type, int eflag, int vflag, int grade_flag, double *d_f, double *d_eatom,
                        double *d_vatom, double *d_ev, double *d_grades, double *d_max_grade, double *d_coeff_ders)
{
  if (!h || !ctx || !d_x || !d_f || rows_a < 0 || rows_b < 0 || rows_c < 0) return MTP_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(d_f) & 15u) return MTP_ERR_ARG;
  int rc = MTP_OK;
  if (!h->overlap) {
    // Default: everything on the caller's stream -- zero + pack, forward group, ONE launch over all rows, reverse group,
    // unpack.  Nothing overlaps, but there is no cross-stream hand-over (6-7 us each on this stack, four per step) and
    // the rows are not split into three launches that each cost a round of wavefronts: measured with the self-exchange
    // 0.097 vs 0.129 ms at 8 192 atoms and 0.507 vs 0.545 ms at 65 536 (mtp_halo_set_overlap(1) selects the other path).
    try {
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
      HIP_OK(hipSetDevice(h->device));
      const int n3 = 3 * h->nsend;
      const size_t nz = 3 * (size_t) (h->nlocal + h->nghost), work = std::max<size_t>((size_t) n3, (nz + 1) / 2);
      if (work > 0) {
        hipLaunchKernelGGL(halo_pack_zero_kernel, dim3((unsigned) ((work + 255) / 256)), dim3(256), 0, st, d_x, h->d_send_idx,
                           h->d_send_shift, h->d_sendbuf, n3, d_f, nz);
        HIP_OK(hipGetLastError());
      }
      exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, st);
      // (finish_tallies = 0: the tally fold rides in the unpack launch behind the reverse exchange)
      rc = mtp_compute_device_rows(ctx, stream, 0, rows_a + rows_b + rows_c, 0, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                   d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders);
      exchange(h, d_f + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, h->d_frecv, h->send_off, h->send_counts, st);
      const int rc2 = mtp_internal_finish_unpack(ctx, stream, eflag, vflag, d_ev, d_f, h->d_send_idx, h->d_frecv, n3);
      if (rc == MTP_OK) rc = rc2;
    } catch (const HaloFail &f) {
      h->last_error = f.what;
      return MTP_ERR_DEVICE;
    }
    return rc;
  }
  try {   // forward_begin with the zeroing of f folded into the pack launch
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_OK(hipSetDevice(h->device));
    const int n3 = 3 * h->nsend;
    const size_t nz = 3 * (size_t) (h->nlocal + h->nghost), work = std::max<size_t>((size_t) n3, (nz + 1) / 2);
    if (work > 0) {
      hipLaunchKernelGGL(halo_pack_zero_kernel, dim3((unsigned) ((work + 255) / 256)), dim3(256), 0, st, d_x, h->d_send_idx,
                         h->d_send_shift, h->d_sendbuf, n3, d_f, nz);
      HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(h->ev_fwd_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_fwd_ready, 0));
    exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts);
    HIP_OK(hipEventRecord(h->ev_fwd_done, h->comm_stream));
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  if (rc == MTP_OK && rows_a > 0)
    rc = mtp_compute_device_rows(ctx, stream, 0, rows_a, 0, d_x, d_type, eflag, vflag, grade_flag, d_f, d_eatom, d_vatom,
                                 d_ev, d_grades, d_max_grade, d_coeff_ders);
  if (rc == MTP_OK) rc = mtp_halo_forward_end(h, stream);
  if (rc == MTP_OK)
    rc = mtp_compute_device_rows(ctx, stream, rows_a, rows_b, rows_c == 0, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                 d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders);
  if (rc == MTP_OK) rc = mtp_halo_reverse_begin(h, stream, d_f);
  if (rc == MTP_OK && rows_c > 0)
    rc = mtp_compute_device_rows(ctx, stream, rows_a + rows_b, rows_c, 1, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                 d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders);
  if (rc == MTP_OK) rc = mtp_halo_reverse_end(h, stream, d_f);
  return rc;
}

int mtp_halo_set_overlap(mtp_halo *h, int enable)
{
  if (!h) return MTP_ERR_ARG;
  h->overlap = enable != 0;
  return MTP_OK;
}

int mtp_halo_get_overlap(const mtp_halo *h) { return h && h->overlap ? 1 : 0; }

int mtp_halo_allreduce(mtp_halo *h, void *stream, double *d_buf, int count, int op)
{
  if (!h || !d_buf || count < 0 || (op != MTP_REDUCE_SUM && op != MTP_REDUCE_MAX)) return MTP_ERR_ARG;
  if (count == 0) return MTP_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    // all collectives of this communicator are issued on one stream (the halo's), ordered with the caller's by events
    HIP_OK(hipEventRecord(h->ev_red_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_red_ready, 0));
    NCCL_OK(ncclAllReduce(d_buf, d_buf, (size_t) count, ncclDouble, op == MTP_REDUCE_SUM ? ncclSum : ncclMax, h->comm,
                          h->comm_stream));
    HIP_OK(hipEventRecord(h->ev_red_done, h->comm_stream));
    HIP_OK(hipStreamWaitEvent(st, h->ev_red_done, 0));
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

}   // extern "C"
