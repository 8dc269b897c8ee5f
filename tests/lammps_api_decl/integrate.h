#include "lammps_api_decl.h"
