// MOCK (see lammps_mock.h)
#pragma once
#include "lammps_mock.h"
