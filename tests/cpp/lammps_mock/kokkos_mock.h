// MOCK of the slice of the LAMMPS KOKKOS package that the LMP_KOKKOS branch of lammps_plugin/pair_mtp_mi355x_plugin.cpp
// touches (AtomKokkos dual views, sync / modified, NeighListKokkos views with stride / extent / data, the execution
// space's stream) -- written from the public LAMMPS / Kokkos documentation for ONE purpose: to compile that branch and
// run it on device memory (tests/cpp/test_plugin_mock.cpp, mode runkk).  Test scaffolding of THIS repository; it is neither
// Kokkos nor LAMMPS and proves nothing about source compatibility with them.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>

#include "lammps_mock.h"

namespace Kokkos {
inline hipStream_t &mock_stream()
{
  static hipStream_t s = nullptr;
  return s;
}
class HIP {
 public:
  hipStream_t hip_stream() const { return mock_stream(); }
};
}   // namespace Kokkos

namespace LAMMPS_NS {

enum ExecutionSpace { Host, Device };
constexpr unsigned int X_MASK = 0x00000001, V_MASK = 0x00000002, F_MASK = 0x00000004, TYPE_MASK = 0x00000010;
struct LMPDeviceType {};

// a device array with the three calls the adapter makes on a Kokkos::View
template <class T> struct MockView {
  T *ptr = nullptr;
  size_t ext[2] = {0, 0}, str[2] = {0, 0};
  T *data() const { return ptr; }
  size_t extent(int k) const { return ext[k]; }
  size_t stride(int k) const { return str[k]; }
};
template <class T> struct MockDualView {
  MockView<T> d_view;
};

class AtomKokkos : public Atom {
 public:
  MockDualView<double> k_x, k_f;
  MockDualView<int> k_type;
  // what the adapter told the data manager (the driver checks the protocol: sync to the device before the kernels,
  // forces marked modified on the device after them)
  unsigned int synced_device = 0, modified_device = 0;
  int calls = 0, sync_call = -1, modified_call = -1;
  void sync(ExecutionSpace space, unsigned int mask)
  {
    if (space == Device) {
      synced_device |= mask;
      sync_call = calls++;
    }
  }
  void modified(ExecutionSpace space, unsigned int mask)
  {
    if (space == Device) {
      modified_device |= mask;
      modified_call = calls++;
    }
  }
};

template <class DeviceType> class NeighListKokkos : public NeighList {
 public:
  MockView<int> d_neighbors, d_numneigh, d_ilist;
};

}   // namespace LAMMPS_NS
