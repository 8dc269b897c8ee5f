/* -*- c++ -*- ----------------------------------------------------------
   USER-UCG/GPU: atom style "ucg" for an UNMODIFIED upstream LAMMPS.

   Same data-file columns, same comm / border / exchange / restart field lists and the
   same data_atom_post clamps as the reference's AtomVecUCG (UCG/atom_vec_ucg.cpp:48-170),
   but the UCG per-atom arrays are OWNED BY THIS CLASS and registered through the public
   Atom::add_peratom() (atom.h:339) instead of being members of a patched Atom class
   (the reference's atom.h:180-192 / atom.cpp:590-609).  Pair styles and fixes reach them
   through AtomVecUCG::get(lmp).

   Compiles only inside a LAMMPS source tree (needs atom_vec.h); see INTEGRATION.md.
------------------------------------------------------------------------- */
#ifdef ATOM_CLASS
// clang-format off
AtomStyle(ucg,AtomVecUCG);
// clang-format on
#else
#ifndef LMP_ATOM_VEC_UCG_GPU_H
#define LMP_ATOM_VEC_UCG_GPU_H

#include "atom_vec.h"

namespace LAMMPS_NS {

// neigh_modify delay of a run whose re-neighbour decision the package takes on the device (fix_ucg_gpu.cpp); never a user's value
constexpr int UCG_GPU_DELAY_SENTINEL = 2000000000;

class AtomVecUCG : virtual public AtomVec {
 public:
  AtomVecUCG(class LAMMPS *);
  void grow_pointers() override;
  void force_clear(int, size_t) override;
  void data_atom_post(int) override;
  int property_atom(const std::string &) override;
  void pack_property_atom(int, double *, int, int) override;

  // the UCG per-atom fields (what the reference keeps in Atom)
  int *ucgstate = nullptr, *num_ucgstates = nullptr;
  double *ucgl = nullptr, *ucgvl = nullptr, *ucgml = nullptr, *ucgp = nullptr, *ucgforce = nullptr;
  double **ucgsoftmaxscores = nullptr;
  int max_ucgstates = 2;

  // the atom style of this LAMMPS instance, or an error if it is not "ucg"
  static AtomVecUCG *get(class LAMMPS *);

 protected:
  int *num_bond, *num_angle, *num_dihedral, *num_improper;
  int **nspecial;
};

}    // namespace LAMMPS_NS
#endif
#endif
