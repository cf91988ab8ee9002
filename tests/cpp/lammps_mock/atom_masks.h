// forwards to the MOCK of the LAMMPS KOKKOS package (kokkos_mock.h: test scaffolding, not LAMMPS / Kokkos)
#pragma once
#include "kokkos_mock.h"
