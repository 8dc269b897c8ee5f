// USER-UCG/GPU atom style "ucg" for unmodified upstream LAMMPS -- see atom_vec_ucg_gpu.h.
// Field lists follow UCG/atom_vec_ucg.cpp:48-90 of the reference word for word; only the
// ownership of the arrays differs (registered here, not in a patched Atom).
#include "atom_vec_ucg_gpu.h"

#include "atom.h"
#include "error.h"

#include <cstring>

using namespace LAMMPS_NS;

AtomVecUCG::AtomVecUCG(LAMMPS *lmp) : AtomVec(lmp)
{
  molecular = Atom::MOLECULAR;
  bonds_allow = angles_allow = dihedrals_allow = impropers_allow = 1;
  mass_type = PER_TYPE;
  forceclearflag = 1;
  atom->molecule_flag = atom->q_flag = 1;

  // own the UCG arrays: Atom::add_peratom(name, address, datatype, cols, threadflag)
  atom->add_peratom("ucgstate", &ucgstate, Atom::INT, 0);
  atom->add_peratom("num_ucgstates", &num_ucgstates, Atom::INT, 0);
  atom->add_peratom("ucgl", &ucgl, Atom::DOUBLE, 0);
  atom->add_peratom("ucgvl", &ucgvl, Atom::DOUBLE, 0);
  atom->add_peratom("ucgml", &ucgml, Atom::DOUBLE, 0);
  atom->add_peratom("ucgp", &ucgp, Atom::DOUBLE, 0);
  atom->add_peratom("ucgforce", &ucgforce, Atom::DOUBLE, 0, 1);
  atom->add_peratom("ucgsoftmaxscores", &ucgsoftmaxscores, Atom::DOUBLE, max_ucgstates, 1);

  const std::vector<std::string> full = {
      "q", "molecule", "num_bond", "bond_type", "bond_atom", "num_angle", "angle_type", "angle_atom1",
      "angle_atom2", "angle_atom3", "num_dihedral", "dihedral_type", "dihedral_atom1", "dihedral_atom2",
      "dihedral_atom3", "dihedral_atom4", "num_improper", "improper_type", "improper_atom1", "improper_atom2",
      "improper_atom3", "improper_atom4", "nspecial", "special"};
  const std::vector<std::string> ucg = {"ucgstate", "ucgl", "ucgvl", "ucgml", "ucgp", "ucgforce",
                                        "ucgsoftmaxscores", "num_ucgstates"};
  fields_grow = full;
  fields_grow.insert(fields_grow.end(), ucg.begin(), ucg.end());
  fields_copy = fields_grow;
  fields_border = {"q", "molecule", "ucgstate", "num_ucgstates", "ucgl", "ucgp"};
  fields_border_vel = {"q", "molecule", "ucgstate", "num_ucgstates", "ucgl", "ucgp", "ucgvl"};
  fields_comm = {"ucgstate", "ucgl", "ucgp"};
  fields_comm_vel = {"ucgstate", "ucgl", "ucgvl", "ucgp"};
  // the GPU pair styles gather over a full list and leave nothing on ghosts, but the CPU
  // styles of the reference still need the reverse sum, so the list is kept
  fields_reverse = {"ucgforce", "ucgsoftmaxscores"};
  fields_exchange = fields_grow;
  fields_restart = {"ucgstate", "ucgl", "ucgml", "ucgvl", "ucgp"};
  fields_data_atom = {"id", "molecule", "type", "q", "x", "ucgstate", "ucgl", "ucgml"};
  fields_data_vel = {"id", "v", "ucgvl"};

  setup_fields();
}

AtomVecUCG *AtomVecUCG::get(LAMMPS *lmp)
{
  auto avec = dynamic_cast<AtomVecUCG *>(lmp->atom->avec);
  if (!avec) lmp->error->all(FLERR, "This style requires atom style ucg.");
  return avec;
}

void AtomVecUCG::grow_pointers()
{
  num_bond = atom->num_bond;
  num_angle = atom->num_angle;
  num_dihedral = atom->num_dihedral;
  num_improper = atom->num_improper;
  nspecial = atom->nspecial;
}

void AtomVecUCG::force_clear(int n, size_t nbytes)
{
  // UCG/atom_vec_ucg.cpp:131-135
  memset(&ucgforce[n], 0, nbytes);
  memset(&ucgsoftmaxscores[n][0], 0, max_ucgstates * nbytes);
}

void AtomVecUCG::data_atom_post(int ilocal)
{
  // UCG/atom_vec_ucg.cpp:145-170
  num_bond[ilocal] = 0;
  num_angle[ilocal] = 0;
  num_dihedral[ilocal] = 0;
  num_improper[ilocal] = 0;
  nspecial[ilocal][0] = nspecial[ilocal][1] = nspecial[ilocal][2] = 0;
  if (ucgl[ilocal] < 0) ucgl[ilocal] = 0.;
  else if (ucgl[ilocal] > 1) ucgl[ilocal] = 1.;
  if (ucgstate[ilocal] < 0) ucgstate[ilocal] = 0;
  else if (ucgstate[ilocal] > 1) ucgstate[ilocal] = 1;
  ucgp[ilocal] = -1.0;
}

int AtomVecUCG::property_atom(const std::string &name)
{
  // UCG/atom_vec_ucg.cpp:172-181: lets an unpatched `compute property/atom` read the UCG fields
  if (name == "ucgstate") return 0;
  if (name == "ucgl") return 1;
  if (name == "ucgforce") return 2;
  if (name == "ucgvl") return 3;
  if (name == "ucgp") return 4;
  if (name == "ucgml") return 5;
  return -1;
}

void AtomVecUCG::pack_property_atom(int index, double *buf, int nvalues, int groupbit)
{
  int *mask = atom->mask;
  const int nlocal = atom->nlocal;
  int n = 0;
  for (int j = 0; j < nlocal; j++) {
    double v = 0.0;
    if (mask[j] & groupbit) {
      switch (index) {
        case 0: v = ucgstate[j]; break;
        case 1: v = ucgl[j]; break;
        case 2: v = ucgforce[j]; break;
        case 3: v = ucgvl[j]; break;
        case 4: v = ucgp[j]; break;
        case 5: v = ucgml[j]; break;
        default: error->all(FLERR, "Unknown property_atom index in AtomVecUCG::pack_property_atom");
      }
    }
    buf[n] = v;
    n += nvalues;
  }
}
