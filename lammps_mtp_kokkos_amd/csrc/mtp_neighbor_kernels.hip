// Device-resident full neighbour list: what LAMMPS' Neighbor/NPair classes hand
// the pair style through list->ilist / numneigh / firstneigh (pair_mtp.cpp:81-85, REQ_FULL at :317-318), built on
// the GPU for drivers that keep positions in HBM.  Cell list with cells of one list cutoff:
//
//   bin       cell id per atom (owned + ghosts), histogram with global atomics; ilist = 0 .. inum-1
//   scan      exclusive prefix over the cells: ONE workgroup (grids up to 65 536 cells; hipcub above that)
//   place     atoms into their cell's segment in the order the atomics land ...
//   order     ... then one wavefront per cell ranks the ids of its segment (ids ascending inside a cell: the list
//             never depends on the order in which atomics landed) and writes the positions in cell order
//             (grids above 65 536 cells: a stable hipcub radix sort of (cell id, atom id) pairs + a gather instead)
//   count     one WAVEFRONT per cell: 64 candidates of the 27 surrounding cells at a time against every owned atom
//             of the cell (ballot + popcount): number of atoms with r^2 <= cut^2
//   scan      row offsets first[inum + 1] (hipcub)
//   fill      the same walk writing neigh[] (prefix popcount = place in the row)
//
// Round 3: the round-2 build was 22 launches (rocprim radix sort = 7, three fills, iota, gather, two scans); this one
// is 10, and the placement costs three small kernels instead of a sort of all atoms.
//
// Integer work only after the distance test: the rows hold exactly the atoms j != i with |x_j - x_i|^2 <= cut^2
// (tests compare them as sets with a host KD-tree list).  HBM-bound and tiny next to a force call.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>

#include "mtp_device.hpp"

namespace {

struct CellGrid {
  double lo[3], inv_cell;
  int n[3];
};

__device__ __forceinline__ void cell_of(const CellGrid &g, const double *x, int i, int c[3])
{
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int v = (int) floor((x[3 * (size_t) i + a] - g.lo[a]) * g.inv_cell);
    c[a] = min(max(v, 0), g.n[a] - 1);   // atoms on or beyond the declared box go to the border cells
  }
}

__global__ void nb_bin(CellGrid g, const double *__restrict__ x, int nall, int inum, int *__restrict__ cell_id,
                       int *__restrict__ cell_count, int *__restrict__ iota, int *__restrict__ ilist)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  int c[3];
  cell_of(g, x, i, c);
  const int id = (c[0] * g.n[1] + c[1]) * g.n[2] + c[2];
  cell_id[i] = id;
  if (iota) iota[i] = i;
  if (i < inum) ilist[i] = i;
  atomicAdd(&cell_count[id], 1);
}

// Exclusive prefix over n <= 65 536 + 1 cell counts by ONE workgroup of 1 024 threads (a run of consecutive entries per
// thread, the run totals scanned through LDS); cursor[c] = start[c] is left for nb_place.
__global__ void __launch_bounds__(1024) nb_scan_cells(const int *count, int n, int *__restrict__ start, int *cursor)   // (cursor may be count)
{
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (n + 1023) / 1024, b = min(t * per, n), e = min(b + per, n);
  int sum = 0;
  for (int k = b; k < e; k++) sum += count[k];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele over the 1 024 run totals
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;   // exclusive
  for (int k = b; k < e; k++) {
    const int c = count[k];
    start[k] = run;
    cursor[k] = run;
    run += c;
  }
}

__global__ void nb_place(const int *__restrict__ cell_id, int nall, int *__restrict__ cursor, int *__restrict__ unordered)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nall) unordered[atomicAdd(&cursor[cell_id[i]], 1)] = i;
}

// One wavefront per cell: the ids of the cell's segment are ranked (rank = number of smaller ids: they are distinct), so
// the segment comes out ascending whatever order nb_place left it in; the positions follow in the same order.
__global__ void __launch_bounds__(256) nb_order_cell(int ncell, const int *__restrict__ cell_start,
                                                     const int *__restrict__ unordered, const double *__restrict__ x,
                                                     int *__restrict__ cell_atoms, double *__restrict__ xs)
{
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (c >= ncell) return;
  const int cs = __builtin_amdgcn_readfirstlane(cell_start[c]), ce = __builtin_amdgcn_readfirstlane(cell_start[c + 1]);
  for (int b0 = cs; b0 < ce; b0 += 64) {
    const bool have = b0 + lane < ce;
    const int id = have ? unordered[b0 + lane] : 0x7fffffff;
    int rank = 0;
    for (int q0 = cs; q0 < ce; q0 += 64) {   // (uniform) every chunk of the segment against this one
      const int other = q0 + lane < ce ? unordered[q0 + lane] : 0x7fffffff;
      const int nq = min(64, ce - q0);
      for (int e = 0; e < nq; e++) rank += __builtin_amdgcn_readlane(other, e) < id ? 1 : 0;
    }
    if (have) {
      const size_t k = (size_t) cs + rank;
      cell_atoms[k] = id;
      xs[3 * k] = x[3 * (size_t) id];
      xs[3 * k + 1] = x[3 * (size_t) id + 1];
      xs[3 * k + 2] = x[3 * (size_t) id + 2];
    }
  }
}

// positions in cell order: the candidate chunks of nb_walk_cell then read contiguous memory
__global__ void nb_gather(const double *__restrict__ x, const int *__restrict__ cell_atoms, int nall,
                          double *__restrict__ xs)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nall) return;
  const int j = cell_atoms[k];
  xs[3 * (size_t) k] = x[3 * (size_t) j];
  xs[3 * (size_t) k + 1] = x[3 * (size_t) j + 1];
  xs[3 * (size_t) k + 2] = x[3 * (size_t) j + 2];
}

// One wavefront per cell (round 2; before: one thread per atom walking ~580 candidates on its own, 0.40 ms for the two
// passes at 65 536 atoms).  The 27 cells around cell c are 9 runs of memory in cell order (the three cells along z are
// adjacent); the lanes hold the candidates of the concatenated runs in registers (12 chunks of 64 at a time: the usual
// neighbourhood in one batch), and every owned atom of the cell is tested against a chunk at once: its position comes
// through the scalar path, the hits are a ballot, their places in the row a prefix popcount -- rows come out in the
// same order as the serial walk (runs ascending, atom ids ascending inside a cell).  FILL = false counts (numneigh, max row length), FILL = true writes neigh[].
// SPLIT wavefronts share a cell: each loads every candidate but tests only the atoms a = sub (mod SPLIT) of the cell --
// no dependence between them (an atom's row belongs to one wavefront), 4x shorter serial chains for small systems.
template <bool FILL, int SPLIT>
__global__ __launch_bounds__(256) void nb_walk_cell(CellGrid g, int inum, double cutsq, int ncell,
                                                    const int *__restrict__ cell_start,
                                                    const int *__restrict__ cell_atoms, const double *__restrict__ xs,
                                                    int *__restrict__ numneigh, const int *__restrict__ first,
                                                    int *__restrict__ neigh, int *__restrict__ max_numneigh)
{
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int c = wid / SPLIT, sub = wid % SPLIT;
  if (c >= ncell) return;
  const int cs = __builtin_amdgcn_readfirstlane(cell_start[c]), ce = __builtin_amdgcn_readfirstlane(cell_start[c + 1]);
  if (cs == ce) return;
  const int cz = c % g.n[2], cy = (c / g.n[2]) % g.n[1], cx = c / (g.n[2] * g.n[1]);
  // the 9 runs (a, b) x [z - 1, z + 1]: first candidate and running total (wave-uniform)
  int run_beg[9], run_pre[10];
  run_pre[0] = 0;
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int a = cx + r / 3 - 1, b = cy + r % 3 - 1;
    int beg = 0, end = 0;
    if (a >= 0 && a < g.n[0] && b >= 0 && b < g.n[1]) {
      const int row = (a * g.n[1] + b) * g.n[2];
      beg = __builtin_amdgcn_readfirstlane(cell_start[row + max(cz - 1, 0)]);
      end = __builtin_amdgcn_readfirstlane(cell_start[row + min(cz + 1, g.n[2] - 1) + 1]);
    }
    run_beg[r] = beg;
    run_pre[r + 1] = run_pre[r] + (end - beg);
  }
  const int total = run_pre[9];
  constexpr int NB = 12;   // candidate chunks (of 64) held in registers at a time: 27 cells of ~22 atoms are 10
  for (int at = cs; at < ce; at += 64) {   // the cell's atoms, 64 at a time (lane l keeps the count of atom at + l)
    const int t = at + lane, nat = min(64, ce - at);
    const bool have = t < ce;
    const int il = have ? cell_atoms[t] : 0x7fffffff;
    const bool owned = il < inum;
    const unsigned long long owned_mask = __ballot(owned);
    if (owned_mask == 0ull) continue;   // ghosts have no row
    int cnt = 0;
    const int row0 = FILL && owned ? first[il] : 0;
    for (int q0 = 0; q0 < total; q0 += 64 * NB) {
      // the candidates of this batch, one per lane and chunk (kc = -1: none)
      double xc[NB], yc[NB], zc[NB];
      int kc[NB], jc[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int q = q0 + 64 * b + lane;
        const bool cand = q < total;
        int k = run_beg[0] + q;   // candidate index in cell order
#pragma unroll
        for (int r = 1; r < 9; r++)
          if (q >= run_pre[r]) k = run_beg[r] + (q - run_pre[r]);
        const size_t kk = (size_t) (cand ? k : cs);
        xc[b] = xs[3 * kk];
        yc[b] = xs[3 * kk + 1];
        zc[b] = xs[3 * kk + 2];
        kc[b] = cand ? k : -1;
        jc[b] = FILL ? cell_atoms[kk] : 0;
      }
      const int nb = min(NB, (total - q0 + 63) / 64);   // uniform
      for (int a = sub; a < nat; a += SPLIT) {   // uniform
        if (!((owned_mask >> a) & 1ull)) continue;
        // the atom's position through the scalar path (uniform address); its hits of this batch in a scalar
        const int ta = __builtin_amdgcn_readfirstlane(at + a);
        const double xi = xs[3 * (size_t) ta], yi = xs[3 * (size_t) ta + 1], zi = xs[3 * (size_t) ta + 2];
        const int base = FILL ? __builtin_amdgcn_readlane(row0, a) + __builtin_amdgcn_readlane(cnt, a) : 0;
        int hits = 0;
#pragma unroll
        for (int b = 0; b < NB; b++)
          if (b < nb) {   // uniform
            const double dx = xc[b] - xi, dy = yc[b] - yi, dz = zc[b] - zi;
            const bool hit = kc[b] >= 0 && kc[b] != ta && dx * dx + dy * dy + dz * dz <= cutsq;
            const unsigned long long m = __ballot(hit);
            if (FILL && hit)
              neigh[base + hits + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u))] = jc[b];
            hits += __popcll(m);
          }
        if (lane == a) cnt += hits;
      }
    }
    if (!FILL && owned && lane % SPLIT == sub) {
      numneigh[il] = cnt;
      atomicMax(max_numneigh, cnt);
    }
  }
}

// ---- LAMMPS-KOKKOS list view -> the internal CSR (mtp_set_neighbors_device_2d) -------------------------------------
// counts[ii] = d_numneigh[d_ilist[ii]] (+ their maximum)
__global__ void __launch_bounds__(256) nb2d_count(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                  int cap, int *__restrict__ counts, int *__restrict__ info)
{
  const int ii = blockIdx.x * 256 + threadIdx.x;
  if (ii > inum) return;
  int c = 0;
  if (ii < inum) {
    c = numneigh[ilist[ii]];
    if (c < 0 || c > cap) {   // a row longer than the view's second extent: refuse (the host reports it)
      atomicExch(info + 2, 1);
      c = 0;
    }
    atomicMax(info + 1, c);
  }
  counts[ii] = c;
}

// neigh[first[ii] + jj] = d_neighbors(i, jj), i = d_ilist[ii].  One wavefront per ROWS consecutive rows; the element
// (i, jj) sits at i * stride_i + jj * stride_jj (KOKKOS LayoutLeft on GPUs: stride_i = 1, lanes over jj read with
// stride extent(0) -- so for that layout lanes run over the rows instead and the wavefront walks jj; LayoutRight:
// stride_jj = 1, lanes over jj read coalesced).  The special-bond bits stay in the entries (the force kernel masks
// them with NEIGHMASK like pair_mtp.cpp:114).
template <bool LANES_OVER_ROWS>
__global__ void __launch_bounds__(256) nb2d_fill(int inum, const int *__restrict__ ilist, const int *__restrict__ first,
                                                 const int *__restrict__ nb, long long stride_i, long long stride_jj,
                                                 int *__restrict__ neigh)
{
  if (LANES_OVER_ROWS) {
    const int ii = blockIdx.x * 256 + threadIdx.x;
    if (ii >= inum) return;
    const int beg = first[ii], n = first[ii + 1] - beg;
    const int *row = nb + (long long) ilist[ii] * stride_i;
    for (int jj = 0; jj < n; jj++) neigh[beg + jj] = row[(long long) jj * stride_jj];
  } else {
    const int lane = threadIdx.x & 63;
    const int ii = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ii >= inum) return;
    const int beg = first[ii], n = first[ii + 1] - beg;
    const int *row = nb + (long long) ilist[ii] * stride_i;
    for (int jj = lane; jj < n; jj += 64) neigh[beg + jj] = row[(long long) jj * stride_jj];
  }
}

}   // namespace

// stage 1 (neigh == nullptr): counts[inum + 1] -> first[inum + 1], d_info = {entries, longest row, bad-row flag};
// stage 2: fills neigh[]
hipError_t mtp_launch_list_from_2d(int inum, const int *d_ilist, const int *d_numneigh, const int *d_neighbors,
                                   long long stride_i, long long stride_jj, int cap, int *counts, void *cub_tmp,
                                   size_t cub_bytes, int *first, int *neigh, int *d_info, hipStream_t st)
{
  hipError_t e;
  if (!neigh) {
    if ((e = hipMemsetAsync(d_info, 0, 3 * sizeof(int), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(nb2d_count, dim3((inum + 1 + 255) / 256), dim3(256), 0, st, inum, d_ilist, d_numneigh, cap, counts, d_info);
    if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, counts, first, inum + 1, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_info, first + inum, sizeof(int), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  } else if (inum > 0) {
    if (stride_jj == 1)
      hipLaunchKernelGGL((nb2d_fill<false>), dim3((inum + 3) / 4), dim3(256), 0, st, inum, d_ilist, first, d_neighbors, stride_i,
                         stride_jj, neigh);
    else
      hipLaunchKernelGGL((nb2d_fill<true>), dim3((inum + 255) / 256), dim3(256), 0, st, inum, d_ilist, first, d_neighbors,
                         stride_i, stride_jj, neigh);
  }
  return hipGetLastError();
}

// Builds the list into caller-provided device buffers in two calls around one host read of the entry count:
//   stage 1 (neigh == nullptr): bins, counts, row offsets; writes {total entries, max row length} to d_info[2]
//   stage 2 (neigh != nullptr): fills neigh[] (first[] must be the stage-1 result)
// scratch ints: cell_id[nall] | cell_atoms[nall] | cell_id_sorted[nall] | iota[nall] | cell_start[ncell + 1] |
//               cell_count[ncell + 1] | numneigh[inum + 1];   xs: 3 * nall doubles (positions in cell order)
hipError_t mtp_launch_neighbor_build(const double *x, int inum, int nall, double cutoff, const double lo[3],
                                     const int ncell3[3], int *scratch, double *xs, void *cub_tmp, size_t cub_bytes, int *ilist,
                                     int *first, int *neigh, int *d_info, hipStream_t st)
{
  CellGrid g;
  for (int a = 0; a < 3; a++) {
    g.lo[a] = lo[a];
    g.n[a] = ncell3[a];
  }
  g.inv_cell = 1.0 / cutoff;
  const int ncell = ncell3[0] * ncell3[1] * ncell3[2];
  int *cell_id = scratch, *cell_atoms = cell_id + nall, *cell_id_sorted = cell_atoms + nall, *iota = cell_id_sorted + nall;
  int *cell_start = iota + nall, *cell_count = cell_start + ncell + 1, *numneigh = cell_count + ncell + 1;
  const int T = 256;
  // four wavefronts per cell while one per cell would leave SIMDs empty (measured: 2 048 atoms, 343 cells: 0.197 ->
  // 0.134 ms per build; 65 536 atoms, 4 913 cells: 0.297 -> 0.369 ms); tuning override: MTP_NB_SPLIT=1|4
  bool split4 = ncell < 2048;
  if (const char *env = std::getenv("MTP_NB_SPLIT")) split4 = std::atoi(env) == 4;
  hipError_t e;
  if (!neigh) {
    // cell_count[ncell + 1] and numneigh[inum + 1] are neighbours in the scratch: one fill for both
    if ((e = hipMemsetAsync(cell_count, 0, sizeof(int) * ((size_t) ncell + 1 + (size_t) inum + 1), st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_info, 0, 2 * sizeof(int), st)) != hipSuccess) return e;
    const bool one_block_scan = ncell <= 65536 && std::getenv("MTP_NB_SORT") == nullptr;   // (MTP_NB_SORT: tests run the sort path)
    if (nall > 0)
      hipLaunchKernelGGL(nb_bin, dim3((nall + T - 1) / T), dim3(T), 0, st, g, x, nall, inum, cell_id, cell_count,
                         one_block_scan ? (int *) nullptr : iota, ilist);
    if (one_block_scan) {
      // cell_count doubles as the placement cursor once the prefix is taken
      hipLaunchKernelGGL(nb_scan_cells, dim3(1), dim3(1024), 0, st, cell_count, ncell + 1, cell_start, cell_count);
      if (nall > 0) {
        hipLaunchKernelGGL(nb_place, dim3((nall + T - 1) / T), dim3(T), 0, st, cell_id, nall, cell_count, cell_id_sorted);
        hipLaunchKernelGGL(nb_order_cell, dim3((ncell + 3) / 4), dim3(T), 0, st, ncell, cell_start, cell_id_sorted, x, cell_atoms, xs);
      }
    } else {
      if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, cell_count, cell_start, ncell + 1, st)) != hipSuccess)
        return e;
      if (nall > 0) {
        // atoms into cell order: a STABLE sort of (cell id, atom id) pairs keeps the ids ascending inside every cell
        int bits = 1;
        while ((1ll << bits) < (long long) ncell) bits++;
        if ((e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, cell_id, cell_id_sorted, iota, cell_atoms, nall, 0, bits,
                                                    st)) != hipSuccess)
          return e;
        hipLaunchKernelGGL(nb_gather, dim3((nall + T - 1) / T), dim3(T), 0, st, x, cell_atoms, nall, xs);
      }
    }
    if (inum > 0) {
      if (split4)
        hipLaunchKernelGGL((nb_walk_cell<false, 4>), dim3(ncell), dim3(T), 0, st, g, inum, cutoff * cutoff, ncell, cell_start,
                           cell_atoms, xs, numneigh, (const int *) nullptr, (int *) nullptr, d_info + 1);
      else
        hipLaunchKernelGGL((nb_walk_cell<false, 1>), dim3((ncell + 3) / 4), dim3(T), 0, st, g, inum, cutoff * cutoff, ncell,
                           cell_start, cell_atoms, xs, numneigh, (const int *) nullptr, (int *) nullptr, d_info + 1);
    }
    if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, numneigh, first, inum + 1, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_info, first + inum, sizeof(int), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  } else if (inum > 0) {
    if (split4)
      hipLaunchKernelGGL((nb_walk_cell<true, 4>), dim3(ncell), dim3(T), 0, st, g, inum, cutoff * cutoff, ncell, cell_start,
                         cell_atoms, xs, numneigh, first, neigh, (int *) nullptr);
    else
      hipLaunchKernelGGL((nb_walk_cell<true, 1>), dim3((ncell + 3) / 4), dim3(T), 0, st, g, inum, cutoff * cutoff, ncell,
                         cell_start, cell_atoms, xs, numneigh, first, neigh, (int *) nullptr);
  }
  return hipGetLastError();
}

// temporary storage for the scans over n entries and the pair sort of nall atoms
size_t mtp_neighbor_scan_bytes(int n, int nall)
{
  size_t scan = 0, sort = 0;
  (void) hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int *) nullptr, (int *) nullptr, n);
  (void) hipcub::DeviceRadixSort::SortPairs(nullptr, sort, (const int *) nullptr, (int *) nullptr, (const int *) nullptr,
                                            (int *) nullptr, std::max(nall, 1));
  return std::max(scan, sort);
}
