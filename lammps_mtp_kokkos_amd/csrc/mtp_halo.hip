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
//
// Streams: the halo calls have no context stream to fall back on, so a NULL `stream` is rejected (MTP_ERR_ARG);
// mtp_halo_force_step, which has a context, resolves NULL to the context's stream ONCE and runs every piece of the
// step -- pack, both groups, the force launch, the unpack -- on that one resolved stream.
//
// Without a communicator (mtp_halo_create with unique_id == NULL) a halo object still packs, unpacks and knows its
// per-peer segment tables; mtp_halo_local_exchange() then moves the segments between the halo objects of ALL ranks of a
// decomposition living in one process on one device (device-to-device copies driven by the same send / receive offset
// tables the RCCL group uses) -- how the multi-peer indexing of an 8-rank job is exercised on a one-GPU box.
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

// Host only: the per-peer segment tables of one rank -- send_off[q] / recv_off[q] = first atom of peer q's segment in
// the packed send buffer / among the ghosts -- with every check mtp_halo_create makes on the layout contract.
int mtp_halo_layout(int nranks, int nlocal, int nghost, const int *send_idx, const int *send_counts, const int *recv_counts,
                    int *send_off, int *recv_off, char *err, int errlen)
{
  if (nranks < 1 || nlocal < 0 || nghost < 0 || !send_counts || !recv_counts) return MTP_ERR_ARG;
  long long nsend = 0, nrecv = 0;
  for (int q = 0; q < nranks; q++) {
    if (send_counts[q] < 0 || recv_counts[q] < 0) return MTP_ERR_ARG;
    if (send_off) send_off[q] = (int) nsend;
    if (recv_off) recv_off[q] = (int) nrecv;
    nsend += send_counts[q];
    nrecv += recv_counts[q];
    if (3 * nsend > 0x7fffffffll || 3 * nrecv > 0x7fffffffll) {
      copy_err("mtp_halo_create: more than 2^31-1 halo coordinates on one rank", err, errlen);
      return MTP_ERR_LIMIT;
    }
  }
  if (send_off) send_off[nranks] = (int) nsend;
  if (recv_off) recv_off[nranks] = (int) nrecv;
  if (nrecv != nghost) {
    copy_err("mtp_halo_create: recv_counts do not add up to nghost", err, errlen);
    return MTP_ERR_ARG;
  }
  if (nsend > 0 && !send_idx) return MTP_ERR_ARG;
  for (long long k = 0; k < nsend; k++)
    if (send_idx[k] < 0 || send_idx[k] >= nlocal) {   // checked on the host: the kernels index x and f with it
      copy_err("mtp_halo_create: send_idx outside the owned atoms", err, errlen);
      return MTP_ERR_ARG;
    }
  return MTP_OK;
}

int mtp_halo_create(int device_id, int nranks, int rank, const void *unique_id, int nlocal, int nghost,
                    const int *send_idx, const double *send_shift, const int *send_counts, const int *recv_counts,
                    mtp_halo **out, char *err, int errlen)
{
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || nlocal < 0 || nghost < 0 || !send_counts || !recv_counts)
    return MTP_ERR_ARG;
  *out = nullptr;
  std::vector<int> soff((size_t) nranks + 1, 0), roff((size_t) nranks + 1, 0);
  const int lrc = mtp_halo_layout(nranks, nlocal, nghost, send_idx, send_counts, recv_counts, soff.data(), roff.data(), err, errlen);
  if (lrc != MTP_OK) return lrc;
  const long long nsend = soff[nranks];
  if (nsend > 0 && !send_shift) return MTP_ERR_ARG;
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
    h->send_off = soff;
    h->recv_off = roff;
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
    if (unique_id) {
      ncclUniqueId id;
      std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
      NCCL_OK(ncclCommInitRank(&h->comm, nranks, id, rank));
      int cnt = 0, me = -1;
      NCCL_OK(ncclCommCount(h->comm, &cnt));
      NCCL_OK(ncclCommUserRank(h->comm, &me));
      if (cnt != nranks || me != rank) throw HaloFail{"RCCL communicator disagrees with the requested rank layout"};
    }   // else: no communicator -- pack / unpack and mtp_halo_local_exchange only
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
  int cnt = h->nranks, me = h->rank, ver = 0;
  if (ncclGetVersion(&ver) != ncclSuccess) return MTP_ERR_DEVICE;
  if (h->comm && (ncclCommCount(h->comm, &cnt) != ncclSuccess || ncclCommUserRank(h->comm, &me) != ncclSuccess))
    return MTP_ERR_DEVICE;
  if (nranks) *nranks = cnt;
  if (rank) *rank = me;
  if (rccl_version) *rccl_version = ver;
  return MTP_OK;
}

// One grouped exchange on stream `st`: to every peer `sbuf + 3 soff[q]` (scount[q] atoms), from every peer
// into `rbuf + 3 roff[q]` (rcount[q] atoms).  A failing ncclSend / ncclRecv still closes the group (an open group
// would swallow every later RCCL call of the process).
static void exchange(mtp_halo *h, const double *sbuf, const std::vector<int> &soff, const std::vector<int> &scount,
                     double *rbuf, const std::vector<int> &roff, const std::vector<int> &rcount, hipStream_t st)
{
  if (!h->comm) throw HaloFail{"this halo was created without a communicator (unique_id == NULL): use mtp_halo_local_exchange"};
  NCCL_OK(ncclGroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int q = 0; q < h->nranks && bad == ncclSuccess; q++) {
    if (scount[q] > 0) bad = ncclSend(sbuf + 3 * (size_t) soff[q], 3 * (size_t) scount[q], ncclDouble, q, h->comm, st);
    if (bad == ncclSuccess && rcount[q] > 0)
      bad = ncclRecv(rbuf + 3 * (size_t) roff[q], 3 * (size_t) rcount[q], ncclDouble, q, h->comm, st);
  }
  const ncclResult_t end = ncclGroupEnd();
  if (bad != ncclSuccess) throw HaloFail{std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(bad)};
  if (end != ncclSuccess) throw HaloFail{std::string("ncclGroupEnd: ") + ncclGetErrorString(end)};
}

static void pack_forward(mtp_halo *h, hipStream_t st, const double *d_x)
{
  if (h->nsend > 0) {
    const int n3 = 3 * h->nsend;
    hipLaunchKernelGGL(halo_pack_kernel, dim3((n3 + 255) / 256), dim3(256), 0, st, d_x, h->d_send_idx, h->d_send_shift,
                       h->d_sendbuf, n3);
    HIP_OK(hipGetLastError());
  }
}

static void unpack_reverse(mtp_halo *h, hipStream_t st, double *d_f)
{
  if (h->nsend > 0) {
    const int n3 = 3 * h->nsend;
    hipLaunchKernelGGL(halo_unpack_add_kernel, dim3((n3 + 255) / 256), dim3(256), 0, st, d_f, h->d_send_idx, h->d_frecv, n3);
    HIP_OK(hipGetLastError());
  }
}

static int null_stream(mtp_halo *h, const char *fn)
{
  h->last_error = std::string(fn) + ": a NULL stream is not accepted (the halo has no stream of its own to order the "
                                    "caller's work on; pass the stream the surrounding kernels run on)";
  return MTP_ERR_ARG;
}

int mtp_halo_forward_begin(mtp_halo *h, void *stream, double *d_x)
{
  if (!h || !d_x) return MTP_ERR_ARG;
  if (!stream) return null_stream(h, "mtp_halo_forward_begin");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    pack_forward(h, st, d_x);
    HIP_OK(hipEventRecord(h->ev_fwd_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_fwd_ready, 0));
    exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts,
             h->comm_stream);
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
  if (!stream) return null_stream(h, "mtp_halo_forward_end");
  if (hipSetDevice(h->device) != hipSuccess ||
      hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), h->ev_fwd_done, 0) != hipSuccess) {
    h->last_error = "hipStreamWaitEvent failed (forward halo)";
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_reverse_begin(mtp_halo *h, void *stream, const double *d_f)
{
  if (!h || !d_f) return MTP_ERR_ARG;
  if (!stream) return null_stream(h, "mtp_halo_reverse_begin");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipEventRecord(h->ev_rev_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_rev_ready, 0));
    // the ghost rows of f go back the way the ghost positions came: sizes and peers swap roles
    exchange(h, d_f + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, h->d_frecv, h->send_off, h->send_counts,
             h->comm_stream);
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
  if (!stream) return null_stream(h, "mtp_halo_reverse_end");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipStreamWaitEvent(st, h->ev_rev_done, 0));
    unpack_reverse(h, st, d_f);
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

// One domain-decomposed force call (one entry point instead of eight, so a driver's per-call overhead is paid once
// per step).  Once the forward exchange has been issued the reverse exchange is ALWAYS issued too, whatever the
// force launches returned: a rank that skipped it would leave its peers blocked in their ncclRecv.  The first error
// is returned afterwards.
int mtp_halo_force_step(mtp_halo *h, mtp_context *ctx, void *stream, int rows_a, int rows_b, int rows_c, double *d_x,
                        const int *d_type, int eflag, int vflag, int grade_flag, double *d_f, double *d_eatom,
                        double *d_vatom, double *d_ev, double *d_grades, double *d_max_grade, double *d_coeff_ders)
{
  if (!h || !ctx || !d_x || !d_f || rows_a < 0 || rows_b < 0 || rows_c < 0) return MTP_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(d_f) & 15u) return MTP_ERR_ARG;
  // NULL -> the context's stream, resolved once: every launch and both groups below use `sv`
  void *sv = mtp_internal_resolve_stream(ctx, stream);
  hipStream_t st = reinterpret_cast<hipStream_t>(sv);
  int rc = MTP_OK;
  auto keep = [&](int r) {
    if (rc == MTP_OK) rc = r;
  };
  const int n3 = 3 * h->nsend;
  const size_t nz = 3 * (size_t) (h->nlocal + h->nghost), work = std::max<size_t>((size_t) n3, (nz + 1) / 2);
  bool forward_issued = false;
  try {
    HIP_OK(hipSetDevice(h->device));
    if (work > 0) {   // zero f + pack: one launch
      hipLaunchKernelGGL(halo_pack_zero_kernel, dim3((unsigned) ((work + 255) / 256)), dim3(256), 0, st, d_x, h->d_send_idx,
                         h->d_send_shift, h->d_sendbuf, n3, d_f, nz);
      HIP_OK(hipGetLastError());
    }
    if (!h->overlap) {
      // Default: everything on the one stream -- zero + pack, forward group, ONE launch over all rows, reverse group,
      // unpack.  Nothing overlaps, but there is no cross-stream hand-over (6-7 us each on this stack, four per step) and
      // the rows are not split into three launches that each cost a round of wavefronts: measured with the self-exchange
      // 0.097 vs 0.129 ms at 8 192 atoms and 0.507 vs 0.545 ms at 65 536 (mtp_halo_set_overlap(1) selects the other path).
      exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, st);
      forward_issued = true;
      // (finish_tallies = 0: the tally fold rides in the unpack launch behind the reverse exchange)
      keep(mtp_compute_device_rows(ctx, sv, 0, rows_a + rows_b + rows_c, 0, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                   d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders));
      exchange(h, d_f + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, h->d_frecv, h->send_off, h->send_counts, st);
      keep(mtp_internal_finish_unpack(ctx, sv, eflag, vflag, d_ev, d_f, h->d_send_idx, h->d_frecv, n3));
      return rc;
    }
    // Overlapped (rows of the installed list ordered interior A | boundary | interior C): forward halo || rows A;
    // boundary rows; reverse halo || rows C; unpack; the groups on the halo's own stream, tied to `st` by events.
    HIP_OK(hipEventRecord(h->ev_fwd_ready, st));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_fwd_ready, 0));
    exchange(h, h->d_sendbuf, h->send_off, h->send_counts, d_x + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts,
             h->comm_stream);
    forward_issued = true;
    HIP_OK(hipEventRecord(h->ev_fwd_done, h->comm_stream));
    if (rows_a > 0)
      keep(mtp_compute_device_rows(ctx, sv, 0, rows_a, 0, d_x, d_type, eflag, vflag, grade_flag, d_f, d_eatom, d_vatom,
                                   d_ev, d_grades, d_max_grade, d_coeff_ders));
    HIP_OK(hipStreamWaitEvent(st, h->ev_fwd_done, 0));
    if (rc == MTP_OK)
      keep(mtp_compute_device_rows(ctx, sv, rows_a, rows_b, rows_c == 0, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                   d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders));
    keep(mtp_halo_reverse_begin(h, sv, d_f));   // even after a failed launch: the peers wait for it
    if (rc == MTP_OK && rows_c > 0)
      keep(mtp_compute_device_rows(ctx, sv, rows_a + rows_b, rows_c, 1, d_x, d_type, eflag, vflag, grade_flag, d_f,
                                   d_eatom, d_vatom, d_ev, d_grades, d_max_grade, d_coeff_ders));
    keep(mtp_halo_reverse_end(h, sv, d_f));
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    // a HIP failure between the two exchanges: still try to meet the peers' receives before reporting it
    if (forward_issued && h->comm) {
      try {
        exchange(h, d_f + 3 * (size_t) h->nlocal, h->recv_off, h->recv_counts, h->d_frecv, h->send_off, h->send_counts,
                 h->overlap ? h->comm_stream : st);
      } catch (const HaloFail &) {
      }
    }
    return MTP_ERR_DEVICE;
  }
  return rc;
}

// The two kernels either side of an exchange, on their own (drivers that move the segments themselves, and the
// single-process rehearsal below): sendbuf[k] = x[send_idx[k]] + shift[k]   /   f[send_idx[k]] += frecv[k].
int mtp_halo_pack_forward(mtp_halo *h, void *stream, const double *d_x)
{
  if (!h || !d_x) return MTP_ERR_ARG;
  if (!stream) return null_stream(h, "mtp_halo_pack_forward");
  try {
    HIP_OK(hipSetDevice(h->device));
    pack_forward(h, reinterpret_cast<hipStream_t>(stream), d_x);
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_unpack_reverse(mtp_halo *h, void *stream, double *d_f)
{
  if (!h || !d_f) return MTP_ERR_ARG;
  if (!stream) return null_stream(h, "mtp_halo_unpack_reverse");
  try {
    HIP_OK(hipSetDevice(h->device));
    unpack_reverse(h, reinterpret_cast<hipStream_t>(stream), d_f);
  } catch (const HaloFail &f) {
    h->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
}

int mtp_halo_get_layout(const mtp_halo *h, int *nsend, int *send_off, int *send_counts, int *recv_off, int *recv_counts)
{
  if (!h) return MTP_ERR_ARG;
  if (nsend) *nsend = h->nsend;
  for (int q = 0; q <= h->nranks; q++) {
    if (send_off) send_off[q] = h->send_off[q];
    if (recv_off) recv_off[q] = h->recv_off[q];
    if (q < h->nranks && send_counts) send_counts[q] = h->send_counts[q];
    if (q < h->nranks && recv_counts) recv_counts[q] = h->recv_counts[q];
  }
  return MTP_OK;
}

// What the RCCL groups of all ranks do together, for the n = nranks halo objects of one decomposition living in one
// process on one device: direction 0 (forward) copies, for every pair (r, q), rank q's packed segment for r --
// sendbuf_q + 3 send_off_q[r], send_counts_q[r] atoms -- into r's ghost rows d_arrays[r] + 3 (nlocal_r + recv_off_r[q]);
// direction 1 (reverse) copies r's ghost rows of d_arrays[r] (forces) for q back into frecv_q + 3 send_off_q[r].  The
// per-peer offsets are the ones exchange() hands to ncclSend / ncclRecv; the counts of the two sides must agree.
int mtp_halo_local_exchange(mtp_halo *const *halos, int n, void *stream, int direction, double *const *d_arrays)
{
  if (!halos || n < 1 || !d_arrays || (direction != 0 && direction != 1)) return MTP_ERR_ARG;
  for (int r = 0; r < n; r++)
    if (!halos[r] || !d_arrays[r] || halos[r]->nranks != n || halos[r]->rank != r || halos[r]->device != halos[0]->device) {
      if (halos[0]) halos[0]->last_error = "mtp_halo_local_exchange: halos must be ranks 0..n-1 of one n-rank decomposition on one device";
      return MTP_ERR_ARG;
    }
  mtp_halo *h0 = halos[0];
  if (!stream) return null_stream(h0, "mtp_halo_local_exchange");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  try {
    HIP_OK(hipSetDevice(h0->device));
    for (int r = 0; r < n; r++)
      for (int q = 0; q < n; q++) {
        const mtp_halo *hr = halos[r], *hq = halos[q];
        const int cnt = hr->recv_counts[q];
        if (cnt != hq->send_counts[r]) throw HaloFail{"mtp_halo_local_exchange: rank " + std::to_string(q) + " sends " +
                                                      std::to_string(hq->send_counts[r]) + " atoms to rank " + std::to_string(r) +
                                                      ", which expects " + std::to_string(cnt)};
        if (cnt == 0) continue;
        double *ghost_rows = d_arrays[r] + 3 * ((size_t) hr->nlocal + (size_t) hr->recv_off[q]);
        const size_t bytes = 3 * (size_t) cnt * sizeof(double);
        if (direction == 0)
          HIP_OK(hipMemcpyAsync(ghost_rows, hq->d_sendbuf + 3 * (size_t) hq->send_off[r], bytes, hipMemcpyDeviceToDevice, st));
        else
          HIP_OK(hipMemcpyAsync(hq->d_frecv + 3 * (size_t) hq->send_off[r], ghost_rows, bytes, hipMemcpyDeviceToDevice, st));
      }
  } catch (const HaloFail &f) {
    h0->last_error = f.what;
    return MTP_ERR_DEVICE;
  }
  return MTP_OK;
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
  if (!stream) return null_stream(h, "mtp_halo_allreduce");
  if (!h->comm) {
    h->last_error = "mtp_halo_allreduce: this halo was created without a communicator";
    return MTP_ERR_STATE;
  }
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
