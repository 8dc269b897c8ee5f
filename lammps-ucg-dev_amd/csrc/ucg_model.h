// ucg_model.h -- host-side model of the UCG pair styles: state-settings map,
// tabulated potentials and the per-style options.  Plays the role of the
// setup half of PairTable_UCGLD / PairTable_UCG_Bethe / PairTable_UCG_Bethe_Density
// (settings, coeff, init_style, init_one, single); compute() is on the device.
#pragma once

#include <string>
#include <vector>

namespace ucg {

enum TabStyle { LOOKUP = 0, LINEAR = 1, SPLINE = 2, BITMAP = 3 };
enum RFlag { RNONE = 0, RLINEAR = 1, RSQ = 2, RBMP = 3 };
enum Style { STYLE_UCGLD = 0, STYLE_BETHE = 1, STYLE_BETHE_DENSITY = 2 };
enum Prior { PRIOR_CHEMPOT = 0, PRIOR_CHEMPOT_NOISE = 1, PRIOR_UCGL = 2, PRIOR_UCGP = 3 };
enum Method { METHOD_MF = 0, METHOD_BETHE = 1 };

struct InputError {
  std::string msg;
};

// One tabulated potential: what read_table() holds plus what compute_table() derives
// (UCG/pair_table_ucgld.h Table struct; UCG/pair_table_ucgld.cpp:897-1245).
class Table {
 public:
  int ninput = 0, rflag = RNONE, fpflag = 0, match = 0;
  double rlo = 0, rhi = 0, fplo = 0, fphi = 0, cut = 0;
  std::vector<double> rfile, efile, ffile, e2file, f2file;
  double innersq = 0, delta = 0, invdelta = 0, deltasq6 = 0;
  std::vector<double> rsq, e, f, de, df, e2, f2;
  // BITMAP tables (UCG/pair_table_ucgld.cpp:1247-1340): 2^tablength bins addressed by the bits of (float) rsq
  std::vector<double> drsq;
  int ntablebits = 0, nmask = 0, nshiftbits = 0;

  void read_file(const std::string &file, const std::string &keyword);
  void build(int tabstyle, int tablength, double cutoff);
  // (f/r, e) at rsq; returns 0 ok, 1 below inner, 2 beyond outer
  int eval(int tabstyle, int tablength, double rsq_, double &fval, double &eval_) const;

 private:
  void param_extract(const std::string &line);
  void spline_table();
};

// upstream Pair::init_bitmap (src/pair.cpp of LAMMPS; not part of the reference tree): which bits of a float in
// [inner^2, outer^2] index a table of 2^ntablebits bins
void init_bitmap(double inner, double outer, int ntablebits, int &masklo, int &maskhi, int &nmask, int &nshiftbits);

void cubic_spline(const double *x, const double *y, int n, double yp1, double ypn, double *y2);
double cubic_splint(const double *xa, const double *ya, const double *y2a, int n, double x);

class PairModel {
 public:
  explicit PairModel(int style_) : style(style_) {}

  int style;
  int tabstyle = SPLINE, tablength = 0;
  int n_actual = 0, n_formal = 0, max_states = 2;
  std::vector<int> n_states_per_type;   // [n_actual+1]
  std::vector<int> formal_from_actual;  // [(n_actual+1)*max_states]
  std::vector<int> actual_from_formal;  // [n_formal+1]
  std::vector<double> chem_pot;         // [n_formal+1]
  std::vector<Table> tables;
  bool allocated = false, initialized = false;
  std::vector<int> tabindex, setflag;   // [(n_formal+1)^2]
  std::vector<double> cutsq;            // [(n_formal+1)^2]
  double T = 0, kT = 0, cutforce = 0;
  // table_ucg_bethe keywords
  int pseudo_flag = 0, prior_flag = PRIOR_UCGL, method_flag = METHOD_BETHE, seed = 0;
  double noise_level = 0;
  std::vector<double> prior_prob_from_type;  // [(n_actual+1)*max_states]
  // table_ucg_bethe_density per-type options
  std::vector<int> use_density, use_state_entropy;
  std::vector<double> cv_thresholds, threshold_radii;

  void settings(int narg, const char *const *arg);
  void coeff(int ntypes, int narg, const char *const *arg);
  void init(int ntypes, double T_, double boltz);
  double single(int itype, int jtype, double rsq, double factor_lj, double &fforce) const;

 private:
  void read_state_settings(const std::string &file);
  void allocate();
};

}  // namespace ucg
