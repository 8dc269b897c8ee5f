/* -*- c++ -*- ----------------------------------------------------------
   USER-UCG/GPU: pair styles table_ucgld, table_ucg_bethe, table_ucg_bethe_density
   backed by libucg_hip.so (include/ucg_hip.h).  Same style names and the same
   pair_style / pair_coeff syntax as the reference (UCG/pair_table_ucgld.h:1-3,
   UCG/pair_table_ucg_bethe.h:31-33, UCG/pair_table_ucg_bethe_density.h:27-29).

   Compiles only inside a LAMMPS source tree (needs pair.h); see INTEGRATION.md.
------------------------------------------------------------------------- */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(table_ucgld,PairTableUCGLDGPU);
PairStyle(table_ucg_bethe,PairTableUCGBetheGPU);
PairStyle(table_ucg_bethe_density,PairTableUCGBetheDensityGPU);
// clang-format on
#else
#ifndef LMP_PAIR_TABLE_UCG_GPU_H
#define LMP_PAIR_TABLE_UCG_GPU_H

#include "pair.h"

#include <vector>

struct ucg_ctx;
struct ucg_pair;

namespace LAMMPS_NS {

class PairTableUCGGPU : public Pair {
 public:
  PairTableUCGGPU(class LAMMPS *, int style);
  ~PairTableUCGGPU() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;
  double single(int, int, int, int, double, double, double, double &) override;
  void *extract(const char *, int &) override;
  void write_restart(FILE *) override;
  void read_restart(FILE *) override;
  void write_restart_settings(FILE *) override;
  void read_restart_settings(FILE *) override;
  // table_ucg_bethe_density under MPI: the forward communication the reference declares and never performs
  // (UCG/pair_table_ucg_bethe_density.h:107-110, .cpp:280): two doubles per atom, between the passes of compute()
  int pack_forward_comm(int, int *, double *, int, int *) override;
  void unpack_forward_comm(int, int, double *) override;

 protected:
  int ucg_style;
  ucg_ctx *ctx = nullptr;
  ucg_pair *gpair = nullptr;
  bigint last_ncalls = -1, last_lastcall = -1;    // Neighbor's two build counters at the last upload (ncalls restarts per run)
  int last_nlocal = -1, last_nghost = -1;
  // resident mode (one rank, every per-step style of the deck from this package): the device arrays are authoritative
  // between the hooks, LAMMPS' arrays are bound as lazily synchronised mirrors (ucg_host_bind), ghosts are refreshed and
  // the re-neighbour decision is taken on the device.  The USER-UCG/GPU fixes ask for it through extract("ucg_resident").
  int resident = 0;
  int driver = 0;    // run_style verlet/ucg/gpu runs the whole step loop inside the library: compute() is not called
  int user_every = 1, user_delay = 0, user_check = 1;    // neigh_modify as the input gave it (the integrator takes decide() over)
  void bind_mirrors();
  std::vector<double> aux;    // [nall][2]: priors, then CV forces, of owned + ghost atoms (density style under MPI)
  double T = 0.0;
  int tabstyle = 0, tablength = 0;    // as given to pair_style: what the reference keeps in restart files
  void check(int rc, bool all);
  void upload_list();

 public:
  // for run_style verlet/ucg/gpu (verlet_ucg_gpu.cpp)
  ucg_ctx *device_context() const { return ctx; }
  ucg_pair *device_pair() const { return gpair; }
  int style_index() const { return ucg_style; }
};

class PairTableUCGLDGPU : public PairTableUCGGPU {
 public:
  PairTableUCGLDGPU(class LAMMPS *lmp) : PairTableUCGGPU(lmp, 0) {}
};
class PairTableUCGBetheGPU : public PairTableUCGGPU {
 public:
  PairTableUCGBetheGPU(class LAMMPS *lmp) : PairTableUCGGPU(lmp, 1) {}
};
class PairTableUCGBetheDensityGPU : public PairTableUCGGPU {
 public:
  PairTableUCGBetheDensityGPU(class LAMMPS *lmp) : PairTableUCGGPU(lmp, 2) {}
};

}    // namespace LAMMPS_NS
#endif
#endif
