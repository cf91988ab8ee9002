// Device-resident full neighbour list: what LAMMPS' Neighbor/NPair classes hand
// the pair style through list->ilist / numneigh / firstneigh (pair_mtp.cpp:81-85, REQ_FULL at :317-318), built on
// the GPU for drivers that keep positions in HBM.  Cell list with cells of one list cutoff:
//
//   bin       cell id per atom (owned + ghosts), histogram with global atomics
//   scan      exclusive prefix over the cells (hipcub)
//   place     atom ids into their cell, then every cell sorted by id (deterministic lists)
//   count     one thread per owned atom walks its 27 cells: number of atoms with r^2 <= cut^2
//   scan      row offsets first[inum + 1]
//   fill      the same walk writing neigh[]
//
// Integer work only after the distance test: the rows hold exactly the atoms j != i with |x_j - x_i|^2 <= cut^2
// (tests compare them as sets with a host KD-tree list).  HBM-bound and tiny next to a force call.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

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

__global__ void nb_bin(CellGrid g, const double *__restrict__ x, int nall, int *__restrict__ cell_id,
                       int *__restrict__ cell_count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  int c[3];
  cell_of(g, x, i, c);
  const int id = (c[0] * g.n[1] + c[1]) * g.n[2] + c[2];
  cell_id[i] = id;
  atomicAdd(&cell_count[id], 1);
}

__global__ void nb_place(const int *__restrict__ cell_id, int nall, const int *__restrict__ cell_start,
                         int *__restrict__ cursor, int *__restrict__ cell_atoms)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  const int id = cell_id[i];
  cell_atoms[cell_start[id] + atomicAdd(&cursor[id], 1)] = i;
}

// one thread per cell: insertion sort by atom id (cells hold ~10 atoms), so list order never depends on the
// order the atomics of nb_place landed in
__global__ void nb_sort_cells(const int *__restrict__ cell_start, int ncell, int *__restrict__ cell_atoms)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int b = cell_start[c], e = cell_start[c + 1];
  for (int a = b + 1; a < e; a++) {
    const int v = cell_atoms[a];
    int k = a - 1;
    while (k >= b && cell_atoms[k] > v) {
      cell_atoms[k + 1] = cell_atoms[k];
      k--;
    }
    cell_atoms[k + 1] = v;
  }
}

// positions in cell order: the candidate loop of nb_walk then reads contiguous memory
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

template <bool FILL>
__global__ void nb_walk(CellGrid g, const double *__restrict__ x, int inum, double cutsq,
                        const int *__restrict__ cell_start, const int *__restrict__ cell_atoms,
                        const double *__restrict__ xs, int nall, int *__restrict__ numneigh,
                        const int *__restrict__ first, int *__restrict__ neigh, int *__restrict__ max_numneigh)
{
  // threads in cell order: the lanes of a wavefront sit in the same or adjacent cells and walk the same
  // candidates (their loads hit the same cache lines); ghosts have no row
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nall) return;
  const int i = cell_atoms[t];
  if (i >= inum) return;
  int c[3];
  cell_of(g, x, i, c);
  const double xi = xs[3 * (size_t) t], yi = xs[3 * (size_t) t + 1], zi = xs[3 * (size_t) t + 2];
  int cnt = 0;
  int *row = FILL ? neigh + first[i] : nullptr;
  for (int a = max(c[0] - 1, 0); a <= min(c[0] + 1, g.n[0] - 1); a++)
    for (int b = max(c[1] - 1, 0); b <= min(c[1] + 1, g.n[1] - 1); b++)
      for (int d = max(c[2] - 1, 0); d <= min(c[2] + 1, g.n[2] - 1); d++) {
        const int id = (a * g.n[1] + b) * g.n[2] + d;
        for (int k = cell_start[id]; k < cell_start[id + 1]; k++) {
          if (k == t) continue;
          const double dx = xs[3 * (size_t) k] - xi, dy = xs[3 * (size_t) k + 1] - yi, dz = xs[3 * (size_t) k + 2] - zi;
          if (dx * dx + dy * dy + dz * dz <= cutsq) {
            if (FILL) row[cnt] = cell_atoms[k];
            cnt++;
          }
        }
      }
  if (!FILL) {
    numneigh[i] = cnt;
    atomicMax(max_numneigh, cnt);
  }
}

__global__ void nb_iota(int *v, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

}   // namespace

// Builds the list into caller-provided device buffers in two calls around one host read of the entry count:
//   stage 1 (neigh == nullptr): bins, counts, row offsets; writes {total entries, max row length} to d_info[2]
//   stage 2 (neigh != nullptr): fills neigh[] (first[] must be the stage-1 result)
// scratch ints: cell_id[nall] | cell_atoms[nall] | cell_start[ncell + 1] | cell_count[ncell + 1] | cursor[ncell] | numneigh[inum + 1];
// xs: 3 * nall doubles (positions in cell order)
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
  int *cell_id = scratch, *cell_atoms = cell_id + nall, *cell_start = cell_atoms + nall;
  int *cell_count = cell_start + ncell + 1, *cursor = cell_count + ncell + 1, *numneigh = cursor + ncell;
  const int T = 256;
  hipError_t e;
  if (!neigh) {
    if ((e = hipMemsetAsync(cell_count, 0, sizeof(int) * (size_t) (2 * ncell + 1), st)) != hipSuccess) return e;   // + cursor
    if ((e = hipMemsetAsync(d_info, 0, 2 * sizeof(int), st)) != hipSuccess) return e;
    if (nall > 0) hipLaunchKernelGGL(nb_bin, dim3((nall + T - 1) / T), dim3(T), 0, st, g, x, nall, cell_id, cell_count);
    if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, cell_count, cell_start, ncell + 1, st)) != hipSuccess)
      return e;
    if (nall > 0) {
      hipLaunchKernelGGL(nb_place, dim3((nall + T - 1) / T), dim3(T), 0, st, cell_id, nall, cell_start, cursor, cell_atoms);
      hipLaunchKernelGGL(nb_sort_cells, dim3((ncell + T - 1) / T), dim3(T), 0, st, cell_start, ncell, cell_atoms);
      hipLaunchKernelGGL(nb_gather, dim3((nall + T - 1) / T), dim3(T), 0, st, x, cell_atoms, nall, xs);
    }
    if ((e = hipMemsetAsync(numneigh, 0, sizeof(int) * (size_t) (inum + 1), st)) != hipSuccess) return e;
    if (inum > 0) {
      hipLaunchKernelGGL(nb_walk<false>, dim3((nall + T - 1) / T), dim3(T), 0, st, g, x, inum, cutoff * cutoff, cell_start,
                         cell_atoms, xs, nall, numneigh, (const int *) nullptr, (int *) nullptr, d_info + 1);
      hipLaunchKernelGGL(nb_iota, dim3((inum + T - 1) / T), dim3(T), 0, st, ilist, inum);
    }
    if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, numneigh, first, inum + 1, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_info, first + inum, sizeof(int), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  } else if (inum > 0) {
    hipLaunchKernelGGL(nb_walk<true>, dim3((nall + T - 1) / T), dim3(T), 0, st, g, x, inum, cutoff * cutoff, cell_start,
                       cell_atoms, xs, nall, numneigh, first, neigh, (int *) nullptr);
  }
  return hipGetLastError();
}

size_t mtp_neighbor_scan_bytes(int n)
{
  size_t bytes = 0;
  (void) hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (int *) nullptr, (int *) nullptr, n);
  return bytes;
}
