// ucg_model.cpp -- host-side setup of the UCG pair styles (see ucg_model.h).
//
// Follows, for parity of every stored bit that the kernels later interpolate:
//   read_state_settings  UCG/pair_table_ucgld.cpp:565-652, ...bethe_density.cpp:778-893
//   settings             UCG/pair_table_ucgld.cpp:654-716, UCG/pair_table_ucg_bethe.cpp:746-886
//   coeff                UCG/pair_table_ucgld.cpp:719-865
//   read_table / param_extract   :897-1017 / :1067-1102
//   spline_table / compute_table :1047-1065 / :1105-1245
//   spline / splint      :1375-1428
//   init_style/init_one  :867-895, UCG/pair_table_ucg_bethe.cpp:1038-1088
//   single               :1474-1520
// Built with -ffp-contract=off: the reference is plain x86-64 code without FMA fusion.
// BITMAP tables use upstream Pair::init_bitmap (absent from the reference tree), restated in init_bitmap().
#include "ucg_model.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "ucg_math.h"

namespace ucg {

namespace {

[[noreturn]] void input_error(const std::string &m) { throw InputError{m}; }

std::vector<std::string> split_ws(const std::string &line)
{
  std::vector<std::string> out;
  std::istringstream is(line);
  std::string w;
  while (is >> w) out.push_back(w);
  return out;
}

std::string strip_comment(const std::string &line)
{
  auto pos = line.find('#');
  return pos == std::string::npos ? line : line.substr(0, pos);
}

double to_double(const std::string &s, const char *what)
{
  char *end = nullptr;
  double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str()) input_error(std::string("Expected floating point parameter for ") + what + ": " + s);
  return v;
}

int to_int(const std::string &s, const char *what)
{
  char *end = nullptr;
  long v = std::strtol(s.c_str(), &end, 10);
  if (end == s.c_str()) input_error(std::string("Expected integer parameter for ") + what + ": " + s);
  return (int) v;
}

}  // namespace

void cubic_spline(const double *x, const double *y, int n, double yp1, double ypn, double *y2)
{
  std::vector<double> u((size_t) n);
  if (yp1 > 0.99e30) {
    y2[0] = u[0] = 0.0;
  } else {
    y2[0] = -0.5;
    u[0] = (3.0 / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - yp1);
  }
  for (int i = 1; i < n - 1; i++) {
    const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    const double p = sig * y2[i - 1] + 2.0;
    y2[i] = (sig - 1.0) / p;
    u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  double qn, un;
  if (ypn > 0.99e30) {
    qn = un = 0.0;
  } else {
    qn = 0.5;
    un = (3.0 / (x[n - 1] - x[n - 2])) * (ypn - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  }
  y2[n - 1] = (un - qn * u[n - 2]) / (qn * y2[n - 2] + 1.0);
  for (int k = n - 2; k >= 0; k--) y2[k] = y2[k] * y2[k + 1] + u[k];
}

double cubic_splint(const double *xa, const double *ya, const double *y2a, int n, double x)
{
  int klo = 0, khi = n - 1;
  while (khi - klo > 1) {
    const int k = (khi + klo) >> 1;
    if (xa[k] > x) khi = k;
    else klo = k;
  }
  const double h = xa[khi] - xa[klo];
  const double a = (xa[khi] - x) / h;
  const double b = (x - xa[klo]) / h;
  return a * ya[klo] + b * ya[khi] +
      ((a * a * a - a) * y2a[klo] + (b * b * b - b) * y2a[khi]) * (h * h) / 6.0;
}

namespace {
inline float int_as_float(int i)
{
  float f;
  std::memcpy(&f, &i, sizeof f);
  return f;
}
inline int float_as_int(float f)
{
  int i;
  std::memcpy(&i, &f, sizeof i);
  return i;
}
}  // namespace

// The table index is made of the low `nexpbits` exponent bits and the top `nmantbits` mantissa bits of the float;
// masklo / maskhi are the remaining high bits of inner^2 / outer^2.
void init_bitmap(double inner, double outer, int ntablebits, int &masklo, int &maskhi, int &nmask, int &nshiftbits)
{
  constexpr int MANT_DIG = 24, FLOAT_BITS = 32;
  if (ntablebits > FLOAT_BITS) input_error("Too many total bits for bitmapped lookup table");
  int nlowermin = 1;
  while (!((std::pow(2.0, (double) nlowermin) <= inner * inner) && (std::pow(2.0, (double) nlowermin + 1.0) > inner * inner))) {
    if (std::pow(2.0, (double) nlowermin) <= inner * inner) nlowermin++;
    else nlowermin--;
  }
  int nexpbits = 0;
  const double required_range = outer * outer / std::pow(2.0, (double) nlowermin);
  double available_range = 2.0;
  while (available_range < required_range) {
    nexpbits++;
    available_range = std::pow(2.0, std::pow(2.0, (double) nexpbits));
  }
  const int nmantbits = ntablebits - nexpbits;
  if (nexpbits > FLOAT_BITS - MANT_DIG) input_error("Too many exponent bits for lookup table");
  if (nmantbits + 1 > MANT_DIG) input_error("Too many mantissa bits for lookup table");
  if (nmantbits < 3) input_error("Too few bits for lookup table");
  nshiftbits = MANT_DIG - (nmantbits + 1);
  nmask = 1;
  for (int j = 0; j < ntablebits + nshiftbits; j++) nmask *= 2;
  nmask -= 1;
  maskhi = float_as_int((float) (outer * outer)) & ~nmask;
  masklo = float_as_int((float) (inner * inner)) & ~nmask;
}

void Table::param_extract(const std::string &line)
{
  ninput = 0;
  rflag = RNONE;
  fpflag = 0;
  const auto words = split_ws(line);
  size_t i = 0;
  while (i < words.size()) {
    const std::string &w = words[i++];
    if (w == "N") {
      if (i >= words.size()) input_error("Pair table parameters: missing value after N");
      ninput = to_int(words[i++], "N");
    } else if (w == "R" || w == "RSQ" || w == "BITMAP") {
      rflag = (w == "R") ? RLINEAR : (w == "RSQ") ? RSQ : RBMP;
      if (i + 1 >= words.size()) input_error("Pair table parameters: missing rlo/rhi");
      rlo = to_double(words[i++], "rlo");
      rhi = to_double(words[i++], "rhi");
    } else if (w == "FPRIME") {
      fpflag = 1;
      if (i + 1 >= words.size()) input_error("Pair table parameters: missing fplo/fphi");
      fplo = to_double(words[i++], "fplo");
      fphi = to_double(words[i++], "fphi");
    } else {
      input_error("Invalid keyword " + w + " in pair table parameters");
    }
  }
  if (ninput == 0) input_error("Pair table parameters did not set N");
}

void Table::read_file(const std::string &file, const std::string &keyword)
{
  std::ifstream in(file);
  if (!in) input_error("Cannot open pair table file " + file);
  std::string line;
  bool found = false;
  // find_section_start: first word of a non-comment line equals the keyword
  while (std::getline(in, line)) {
    const auto words = split_ws(strip_comment(line));
    if (words.empty()) continue;
    if (words[0] == keyword) { found = true; break; }
  }
  if (!found) input_error("Did not find keyword " + keyword + " in table file");
  // parameter line
  bool got = false;
  while (std::getline(in, line)) {
    line = strip_comment(line);
    if (split_ws(line).empty()) continue;
    got = true;
    break;
  }
  if (!got) input_error("Missing pair table parameter line for " + keyword);
  param_extract(line);
  // setup bitmap parameters for table to read in (:918-927)
  int masklo = 0, maskhi = 0, nm = 0, nsh = 0;
  ntablebits = 0;
  if (rflag == RBMP) {
    while (1 << ntablebits < ninput) ntablebits++;
    if (1 << ntablebits != ninput) input_error("Bitmapped table is incorrect length in table file");
    init_bitmap(rlo, rhi, ntablebits, masklo, maskhi, nm, nsh);
  }
  rfile.assign((size_t) ninput, 0.0);
  efile.assign((size_t) ninput, 0.0);
  ffile.assign((size_t) ninput, 0.0);
  // skip_line(): exactly one raw line (the blank line of the table format)
  if (!std::getline(in, line)) input_error("Premature end of pair table file " + file);
  for (int i = 0; i < ninput; i++) {
    std::vector<std::string> words;
    while (std::getline(in, line)) {
      words = split_ws(strip_comment(line));
      if (!words.empty()) break;
    }
    if (words.size() < 4) {
      char buf[256];
      std::snprintf(buf, sizeof buf, "Data missing when parsing pair table '%s' line %d of %d.",
                    keyword.c_str(), i + 1, ninput);
      input_error(buf);
    }
    const double rf = to_double(words[1], "r");
    efile[(size_t) i] = 1.0 * to_double(words[2], "e");
    ffile[(size_t) i] = 1.0 * to_double(words[3], "f");
    double rnew = rf;
    if (rflag == RLINEAR) rnew = rlo + (rhi - rlo) * i / (ninput - 1);
    else if (rflag == RSQ) {
      rnew = rlo * rlo + (rhi * rhi - rlo * rlo) * i / (ninput - 1);
      rnew = std::sqrt(rnew);
    } else if (rflag == RBMP) {
      float fl = int_as_float((i << nsh) | masklo);
      if (fl < rlo * rlo) fl = int_as_float((i << nsh) | maskhi);
      rnew = sqrtf(fl);
    }
    rfile[(size_t) i] = rnew;
  }
}

void Table::spline_table()
{
  e2file.assign((size_t) ninput, 0.0);
  f2file.assign((size_t) ninput, 0.0);
  const double ep0 = -ffile[0];
  const double epn = -ffile[(size_t) ninput - 1];
  cubic_spline(rfile.data(), efile.data(), ninput, ep0, epn, e2file.data());
  if (fpflag == 0) {
    fplo = (ffile[1] - ffile[0]) / (rfile[1] - rfile[0]);
    fphi = (ffile[(size_t) ninput - 1] - ffile[(size_t) ninput - 2]) /
        (rfile[(size_t) ninput - 1] - rfile[(size_t) ninput - 2]);
  }
  cubic_spline(rfile.data(), ffile.data(), ninput, fplo, fphi, f2file.data());
}

void Table::build(int tabstyle, int tablength, double cutoff)
{
  cut = cutoff;
  if (ninput <= 1) input_error("Invalid pair table length");
  double lo, hi;
  if (rflag == 0) { lo = rfile[0]; hi = rfile[(size_t) ninput - 1]; }
  else { lo = rlo; hi = rhi; }
  if (cut <= lo || cut > hi) input_error("Pair table cutoff outside of table");
  if (lo <= 0.0) input_error("Invalid pair table lower boundary");

  match = 0;
  if (tabstyle == LINEAR && ninput == tablength && rflag == RSQ && rhi == cut) match = 1;
  if (tabstyle == BITMAP && ninput == 1 << tablength && rflag == RBMP && rhi == cut) match = 1;
  if (rflag == RBMP && match == 0) input_error("Bitmapped table in file does not match requested table");
  if (match == 0) spline_table();

  const int tlm1 = tablength - 1;
  const double inner = rflag ? rlo : rfile[0];
  innersq = inner * inner;
  delta = (cut * cut - innersq) / tlm1;
  invdelta = 1.0 / delta;

  const double *rf = rfile.data(), *ef = efile.data(), *ff = ffile.data();
  const double *e2f = e2file.data(), *f2f = f2file.data();

  if (tabstyle == LOOKUP) {
    e.assign((size_t) tlm1, 0.0);
    f.assign((size_t) tlm1, 0.0);
    for (int i = 0; i < tlm1; i++) {
      const double r2 = innersq + (i + 0.5) * delta;
      const double r = std::sqrt(r2);
      e[(size_t) i] = cubic_splint(rf, ef, e2f, ninput, r);
      f[(size_t) i] = cubic_splint(rf, ff, f2f, ninput, r) / r;
    }
  } else if (tabstyle == LINEAR) {
    rsq.assign((size_t) tablength, 0.0);
    e.assign((size_t) tablength, 0.0);
    f.assign((size_t) tablength, 0.0);
    de.assign((size_t) tlm1, 0.0);
    df.assign((size_t) tlm1, 0.0);
    for (int i = 0; i < tablength; i++) {
      const double r2 = innersq + i * delta;
      const double r = std::sqrt(r2);
      rsq[(size_t) i] = r2;
      if (match) {
        e[(size_t) i] = efile[(size_t) i];
        f[(size_t) i] = ffile[(size_t) i] / r;
      } else {
        e[(size_t) i] = cubic_splint(rf, ef, e2f, ninput, r);
        f[(size_t) i] = cubic_splint(rf, ff, f2f, ninput, r) / r;
      }
    }
    for (int i = 0; i < tlm1; i++) {
      de[(size_t) i] = e[(size_t) i + 1] - e[(size_t) i];
      df[(size_t) i] = f[(size_t) i + 1] - f[(size_t) i];
    }
  } else if (tabstyle == BITMAP) {
    // bitmapped linear tables (:1247-1340): 2^N bins from inner to cut, spaced in bitmapped manner
    int masklo, maskhi;
    init_bitmap(inner, cut, tablength, masklo, maskhi, nmask, nshiftbits);
    ntablebits = tablength;
    const int ntable = 1 << tablength, ntablem1 = ntable - 1;
    rsq.assign((size_t) ntable, 0.0);
    e.assign((size_t) ntable, 0.0);
    f.assign((size_t) ntable, 0.0);
    de.assign((size_t) ntable, 0.0);
    df.assign((size_t) ntable, 0.0);
    drsq.assign((size_t) ntable, 0.0);
    float minrsq = int_as_float((0 << nshiftbits) | maskhi);
    for (int i = 0; i < ntable; i++) {
      float fl = int_as_float((i << nshiftbits) | masklo);
      if (fl < innersq) fl = int_as_float((i << nshiftbits) | maskhi);
      const double r = sqrtf(fl);
      rsq[(size_t) i] = fl;
      if (match) {
        e[(size_t) i] = efile[(size_t) i];
        f[(size_t) i] = ffile[(size_t) i] / r;
      } else {
        e[(size_t) i] = cubic_splint(rf, ef, e2f, ninput, r);
        f[(size_t) i] = cubic_splint(rf, ff, f2f, ninput, r) / r;
      }
      minrsq = (minrsq < fl) ? minrsq : fl;
    }
    innersq = minrsq;
    for (int i = 0; i < ntablem1; i++) {
      de[(size_t) i] = e[(size_t) i + 1] - e[(size_t) i];
      df[(size_t) i] = f[(size_t) i + 1] - f[(size_t) i];
      drsq[(size_t) i] = 1.0 / (rsq[(size_t) i + 1] - rsq[(size_t) i]);
    }
    // tables are connected periodically between 0 and ntablem1
    de[(size_t) ntablem1] = e[0] - e[(size_t) ntablem1];
    df[(size_t) ntablem1] = f[0] - f[(size_t) ntablem1];
    drsq[(size_t) ntablem1] = 1.0 / (rsq[0] - rsq[(size_t) ntablem1]);
    // the bin holding the largest r gets the deltas towards cut*cut
    const int itablemin = (float_as_int(minrsq) & nmask) >> nshiftbits;
    int itablemax = itablemin - 1;
    if (itablemin == 0) itablemax = ntablem1;
    int itablemaxm1 = itablemax - 1;
    if (itablemax == 0) itablemaxm1 = ntablem1;
    float fl = int_as_float((itablemax << nshiftbits) | maskhi);
    if (fl < cut * cut) {
      if (match) {
        de[(size_t) itablemax] = de[(size_t) itablemaxm1];
        df[(size_t) itablemax] = df[(size_t) itablemaxm1];
        drsq[(size_t) itablemax] = drsq[(size_t) itablemaxm1];
      } else {
        fl = (float) (cut * cut);
        const double r = sqrtf(fl);
        const double e_tmp = cubic_splint(rf, ef, e2f, ninput, r);
        const double f_tmp = cubic_splint(rf, ff, f2f, ninput, r) / r;
        de[(size_t) itablemax] = e_tmp - e[(size_t) itablemax];
        df[(size_t) itablemax] = f_tmp - f[(size_t) itablemax];
        drsq[(size_t) itablemax] = 1.0 / (fl - rsq[(size_t) itablemax]);
      }
    }
  } else {  // SPLINE
    rsq.assign((size_t) tablength, 0.0);
    e.assign((size_t) tablength, 0.0);
    f.assign((size_t) tablength, 0.0);
    e2.assign((size_t) tablength, 0.0);
    f2.assign((size_t) tablength, 0.0);
    deltasq6 = delta * delta / 6.0;
    for (int i = 0; i < tablength; i++) {
      const double r2 = innersq + i * delta;
      const double r = std::sqrt(r2);
      rsq[(size_t) i] = r2;
      if (match) {
        e[(size_t) i] = efile[(size_t) i];
        f[(size_t) i] = ffile[(size_t) i] / r;
      } else {
        e[(size_t) i] = cubic_splint(rf, ef, e2f, ninput, r);
        f[(size_t) i] = cubic_splint(rf, ff, f2f, ninput, r);
      }
    }
    const double ep0 = -f[0] / (2.0 * std::sqrt(innersq));
    const double epn = -f[(size_t) tlm1] / (2.0 * cut);
    cubic_spline(rsq.data(), e.data(), tablength, ep0, epn, e2.data());

    double fp0, fpn;
    const double secant_factor = 0.1;
    if (fpflag) {
      fp0 = (fplo / std::sqrt(innersq) - f[0] / innersq) / (2.0 * std::sqrt(innersq));
    } else {
      const double rsq1 = innersq;
      const double rsq2 = rsq1 + secant_factor * delta;
      fp0 = (cubic_splint(rf, ff, f2f, ninput, std::sqrt(rsq2)) / std::sqrt(rsq2) - f[0] / std::sqrt(rsq1)) /
          (secant_factor * delta);
    }
    if (fpflag && cut == rfile[(size_t) ninput - 1]) {
      fpn = (fphi / cut - f[(size_t) tlm1] / (cut * cut)) / (2.0 * cut);
    } else {
      const double rsq2 = cut * cut;
      const double rsq1 = rsq2 - secant_factor * delta;
      fpn = (f[(size_t) tlm1] / std::sqrt(rsq2) -
             cubic_splint(rf, ff, f2f, ninput, std::sqrt(rsq1)) / std::sqrt(rsq1)) /
          (secant_factor * delta);
    }
    for (int i = 0; i < tablength; i++) f[(size_t) i] /= std::sqrt(rsq[(size_t) i]);
    cubic_spline(rsq.data(), f.data(), tablength, fp0, fpn, f2.data());
  }
}

int Table::eval(int tabstyle, int tablength, double r2, double &fval, double &eval_) const
{
  const int tlm1 = tablength - 1;
  if (r2 < innersq) return 1;
  if (tabstyle == BITMAP) {  // :466-476: no outer-cutoff check in this branch
    const float fl = (float) r2;
    const int ib = (float_as_int(fl) & nmask) >> nshiftbits;
    const double fraction = ((double) fl - rsq[(size_t) ib]) * drsq[(size_t) ib];
    fval = f[(size_t) ib] + fraction * df[(size_t) ib];
    eval_ = e[(size_t) ib] + fraction * de[(size_t) ib];
    return 0;
  }
  const int it = static_cast<int>((r2 - innersq) * invdelta);
  if (it >= tlm1) return 2;
  if (tabstyle == LOOKUP) {
    fval = f[(size_t) it];
    eval_ = e[(size_t) it];
  } else if (tabstyle == LINEAR) {
    const double fraction = (r2 - rsq[(size_t) it]) * invdelta;
    fval = f[(size_t) it] + fraction * df[(size_t) it];
    eval_ = e[(size_t) it] + fraction * de[(size_t) it];
  } else {
    const double b = (r2 - rsq[(size_t) it]) * invdelta;
    const double a = 1.0 - b;
    fval = a * f[(size_t) it] + b * f[(size_t) it + 1] +
        ((a * a * a - a) * f2[(size_t) it] + (b * b * b - b) * f2[(size_t) it + 1]) * deltasq6;
    eval_ = a * e[(size_t) it] + b * e[(size_t) it + 1] +
        ((a * a * a - a) * e2[(size_t) it] + (b * b * b - b) * e2[(size_t) it + 1]) * deltasq6;
  }
  return 0;
}

// --------------------------------------------------------------------------------------

void PairModel::read_state_settings(const std::string &file)
{
  std::ifstream in(file);
  if (!in) input_error("Cannot open file " + file);
  std::string line;
  if (!std::getline(in, line)) input_error("Unexpected end of RLEUCG state settings file");
  {
    const auto w = split_ws(line);
    if (w.size() < 3) input_error("UCG state settings file: header must be 'n_actual n_formal max_states'");
    n_actual = to_int(w[0], "n_actual");
    n_formal = to_int(w[1], "n_formal");
    max_states = to_int(w[2], "max_states");
  }
  if (n_actual < 1 || n_formal < n_actual) input_error("UCG state settings file: inconsistent type counts");
  if (max_states < 2) max_states = 2;
  const size_t na = (size_t) n_actual + 1, nf = (size_t) n_formal + 1, ms = (size_t) max_states;
  n_states_per_type.assign(na, 0);
  actual_from_formal.assign(nf, 0);
  chem_pot.assign(nf, 0.0);
  formal_from_actual.assign(na * ms, 0);
  prior_prob_from_type.assign(na * ms, 0.0);
  use_density.assign(na, 0);
  use_state_entropy.assign(na, 0);
  cv_thresholds.assign(na, 0.0);
  threshold_radii.assign(na, 0.0);

  for (int i = 1; i <= n_actual; i++) {
    if (!std::getline(in, line)) input_error("Unexpected end of UCG state settings file");
    auto w = split_ws(line);
    if (w.size() < 2) input_error("UCG state settings file: expected '<type> <nstates>'");
    const int this_type = to_int(w[0], "type");
    n_states_per_type[(size_t) i] = to_int(w[1], "nstates");
    if (n_states_per_type[(size_t) i] < 1 || n_states_per_type[(size_t) i] > 2)
      input_error("Invalid number of states for atom type " + std::to_string(i) + ". Only 1 or 2 states are allowed.");
    if (this_type != i)
      input_error("Please write orderly. Invalid atom type " + std::to_string(this_type) +
                  " in UCG state settings file. Expected " + std::to_string(i) + ".");
    if (n_states_per_type[(size_t) i] != 2) continue;

    if (!std::getline(in, line)) input_error("Unexpected end of UCG state settings file");
    w = split_ws(line);
    if (w.size() < 2) input_error("Not enough formal types specified for atom type " + std::to_string(i) + ".");
    for (int j = 0; j < 2; j++) {
      const int ft = to_int(w[(size_t) j], "formal type");
      if (ft < 0 || ft > n_formal) input_error("Formal type out of range in UCG state settings file");
      formal_from_actual[(size_t) i * ms + (size_t) j] = ft;
      actual_from_formal[(size_t) ft] = i;
    }
    if (style == STYLE_BETHE_DENSITY) {
      if (w.size() < 3) input_error("Missing state type for atom type " + std::to_string(i) + ".");
      if (w.size() < 4) input_error("Missing entropy specification for atom type " + std::to_string(i) + ".");
      if (w[3] == "entropy") use_state_entropy[(size_t) i] = 1;
      else if (w[3] == "no_entropy") use_state_entropy[(size_t) i] = 0;
      else input_error("Unknown entropy specification: " + w[3] + ". Use 'entropy' or 'no_entropy'.");
      if (w[2] == "density") {
        use_density[(size_t) i] = 1;
        if (!std::getline(in, line)) input_error("Unexpected end of RLEUCG state settings file");
        const auto d = split_ws(line);
        if (d.size() < 2) input_error("Expected '<cv_threshold> <threshold_radius>' for a density type");
        cv_thresholds[(size_t) i] = to_double(d[0], "cv_threshold");
        threshold_radii[(size_t) i] = to_double(d[1], "threshold_radius");
      }
    }
    if (!std::getline(in, line)) input_error("Unexpected end of UCG state settings file");
    w = split_ws(line);
    if (w.size() < 2) input_error("Not enough chemical potentials specified for atom type " + std::to_string(i) + ".");
    for (int j = 0; j < 2; j++)
      chem_pot[(size_t) formal_from_actual[(size_t) i * ms + (size_t) j]] = to_double(w[(size_t) j], "chemical potential");
  }
}

void PairModel::settings(int narg, const char *const *arg)
{
  if (narg < 3) input_error("Illegal pair_style command: expected <lookup|linear|spline|bitmap> <N> <state settings file>");
  const std::string ts = arg[0];
  if (ts == "lookup") tabstyle = LOOKUP;
  else if (ts == "linear") tabstyle = LINEAR;
  else if (ts == "spline") tabstyle = SPLINE;
  else if (ts == "bitmap") tabstyle = BITMAP;
  else input_error("Unknown table style in pair_style command: " + ts);
  tablength = to_int(arg[1], "table length");
  if (tablength < 2) input_error("Illegal number of pair table entries: " + std::to_string(tablength));

  pseudo_flag = 0;
  prior_flag = PRIOR_UCGL;
  method_flag = METHOD_BETHE;
  noise_level = 0.0;

  read_state_settings(arg[2]);

  int iarg = 3;
  while (iarg < narg) {
    const std::string kw = arg[iarg];
    if (kw == "ewald" || kw == "pppm" || kw == "msm" || kw == "dispersion" || kw == "tip4p") {
      // KSpace compatibility assertions of pair_style table: accepted, no effect here
    } else if (style == STYLE_BETHE && kw == "method") {
      if (++iarg >= narg) input_error("Missing argument for pair_style table_ucg_bethe method");
      const std::string v = arg[iarg];
      if (v == "mf" || v == "meanfield") method_flag = METHOD_MF;
      else if (v == "bethe" || v == "Bethe") method_flag = METHOD_BETHE;
      else input_error("Unknown argument for pair_style table_ucg_bethe method: " + v + ", please write mf or bethe");
    } else if (style == STYLE_BETHE && kw == "pseudo") {
      if (++iarg >= narg) input_error("Missing argument for pair_style table_ucg_bethe pseudo");
      const std::string v = arg[iarg];
      if (v == "yes") pseudo_flag = 0;
      else if (v == "no") pseudo_flag = 1;
      else input_error("Unknown argument for pair_style table_ucg_bethe pseudo: " + v + ", please write yes or no");
    } else if (style == STYLE_BETHE && kw == "prior") {
      if (++iarg >= narg) input_error("Missing argument for pair_style table_ucg_bethe");
      const std::string v = arg[iarg];
      if (v == "chemical_potential") {
        iarg += 1;
        if (iarg >= narg) {
          prior_flag = PRIOR_CHEMPOT;
          iarg -= 1;
        } else if (std::string(arg[iarg]) == "noise") {
          prior_flag = PRIOR_CHEMPOT_NOISE;
          if (++iarg >= narg) input_error("Missing argument for prior chemical_potential noise: noise level must be set");
          noise_level = to_double(arg[iarg], "noise level");
          if (noise_level <= 0.0) noise_level = 0.0;
          if (++iarg >= narg) input_error("Missing argument for prior chemical_potential noise: random seed must be set");
          seed = to_int(arg[iarg], "seed");
          if (seed <= 0) seed = -seed + 1;
        }
      } else if (v == "ucgl") {
        prior_flag = PRIOR_UCGL;
      } else {
        input_error("Unknown argument for pair_style table_ucg_bethe prior: " + v + ", please write chemical_potential or ucgl");
      }
    } else if (style == STYLE_BETHE) {
      // the reference's table_ucg_bethe parser falls through silently on unknown words
    } else {
      input_error("Unknown pair_style table keyword: " + kw);
    }
    iarg++;
  }
  tables.clear();
  tabindex.clear();
  setflag.clear();
  cutsq.clear();
  allocated = false;
  initialized = false;
}

void PairModel::allocate()
{
  const size_t nt = (size_t) n_formal + 1;
  allocated = true;
  setflag.assign(nt * nt, 0);
  cutsq.assign(nt * nt, 0.0);
  tabindex.assign(nt * nt, 0);
}

namespace {
void parse_bounds(const std::string &s, int nmin, int nmax, int &lo, int &hi)
{
  const auto star = s.find('*');
  if (star == std::string::npos) lo = hi = to_int(s, "type");
  else if (s.size() == 1) { lo = nmin; hi = nmax; }
  else if (star == 0) { lo = nmin; hi = to_int(s.substr(1), "type"); }
  else if (star == s.size() - 1) { lo = to_int(s.substr(0, star), "type"); hi = nmax; }
  else { lo = to_int(s.substr(0, star), "type"); hi = to_int(s.substr(star + 1), "type"); }
  if (lo < nmin || hi > nmax || lo > hi) input_error("Invalid type range in pair_coeff: " + s);
}
}  // namespace

void PairModel::coeff(int ntypes, int narg, const char *const *arg)
{
  if (narg < 7) {
    if (narg == 6) input_error("This pair style requires explicit definition of cutoff for each table.");
    input_error("Too few arguments.");
  }
  if (n_states_per_type.empty()) input_error("pair_coeff before pair_style");
  if (!allocated) allocate();
  const size_t nt = (size_t) n_formal + 1, ms = (size_t) max_states;

  int ilo, ihi, jlo, jhi;
  parse_bounds(arg[0], 1, ntypes, ilo, ihi);
  parse_bounds(arg[1], 1, ntypes, jlo, jhi);
  const int Ns_i = to_int(arg[2], "Ns_i");
  const int Ns_j = to_int(arg[3], "Ns_j");
  for (int t = ilo; t < ihi; t++)
    if (t <= n_actual && Ns_i != n_states_per_type[(size_t) t])
      input_error("Number of states for atom type " + std::to_string(t) + " does not match the number of states in the settings file.");
  for (int t = jlo; t < jhi; t++)
    if (t <= n_actual && Ns_j != n_states_per_type[(size_t) t])
      input_error("Number of states for atom type " + std::to_string(t) + " does not match the number of states in the settings file.");
  if (narg != 4 + 3 * Ns_i * Ns_j)
    input_error("Incorrect number of arguments for pair_coeff command. Expected 4 + 3 * n_states_i * n_states_j arguments.");
  if (ihi > n_actual || jhi > n_actual)
    input_error("pair_coeff I J must name ACTUAL types (<= n_actual of the state settings file)");

  int this_i = 4;
  for (int s_i = 0; s_i < Ns_i; s_i++) {
    for (int s_j = 0; s_j < Ns_j; s_j++) {
      Table tb;
      tb.read_file(arg[this_i], arg[this_i + 1]);
      tb.build(tabstyle, tablength, to_double(arg[this_i + 2], "table cutoff"));
      const int id = (int) tables.size();
      tables.push_back(std::move(tb));
      int count = 0;
      for (int i = ilo; i <= ihi; i++) {
        for (int j = (jlo > i ? jlo : i); j <= jhi; j++) {
          const int fi = formal_from_actual[(size_t) i * ms + (size_t) s_i];
          const int fj = formal_from_actual[(size_t) j * ms + (size_t) s_j];
          if (fi == 0)
            input_error("Formal type not defined in pair_style command for actual type " + std::to_string(i) + ", state " + std::to_string(s_i));
          if (fj == 0)
            input_error("Formal type not defined in pair_style command for actual type " + std::to_string(j) + ", state " + std::to_string(s_j));
          tabindex[(size_t) fi * nt + (size_t) fj] = id;
          setflag[(size_t) fi * nt + (size_t) fj] = 1;
          count++;
        }
      }
      if (count == 0) input_error("Illegal pair_coeff command");
      this_i += 3;
    }
  }
  initialized = false;
}

void PairModel::init(int ntypes, double T_, double boltz)
{
  if (!allocated) input_error("All pair coeffs are not set");
  if (ntypes > n_formal) input_error("atom->ntypes exceeds n_formal of the state settings file");
  const size_t nt = (size_t) n_formal + 1, ms = (size_t) max_states;
  T = T_;
  kT = boltz * T;
  cutforce = 0.0;
  for (int i = 1; i <= ntypes; i++) {
    for (int j = i; j <= ntypes; j++) {
      if (setflag[(size_t) i * nt + (size_t) j] == 0) input_error("All pair coeffs are not set");
      tabindex[(size_t) j * nt + (size_t) i] = tabindex[(size_t) i * nt + (size_t) j];
      const double c = tables[(size_t) tabindex[(size_t) i * nt + (size_t) j]].cut;
      cutsq[(size_t) i * nt + (size_t) j] = cutsq[(size_t) j * nt + (size_t) i] = c * c;
      if (c > cutforce) cutforce = c;
    }
  }
  if (style == STYLE_BETHE) {
    double denomi = 0.0;
    for (int i = 1; i <= n_actual; i++) {
      const int ns = n_states_per_type[(size_t) i];
      if (ns == 0) continue;
      if (ns == 1) {
        prior_prob_from_type[(size_t) i * ms] = 1.0;
      } else {
        for (int j = 0; j < ns; j++) {
          prior_prob_from_type[(size_t) i * ms + (size_t) j] =
              ucg_exp(-chem_pot[(size_t) formal_from_actual[(size_t) i * ms + (size_t) j]] / kT);
          denomi += prior_prob_from_type[(size_t) i * ms + (size_t) j];
        }
        for (int j = 0; j < ns; j++) prior_prob_from_type[(size_t) i * ms + (size_t) j] /= denomi;
        denomi = 0.0;
      }
    }
  }
  initialized = true;
}

double PairModel::single(int itype, int jtype, double rsq, double factor_lj, double &fforce) const
{
  const size_t nt = (size_t) n_formal + 1;
  const Table &tb = tables[(size_t) tabindex[(size_t) itype * nt + (size_t) jtype]];
  double fv = 0.0, ev = 0.0;
  const int rc = tb.eval(tabstyle, tablength, rsq, fv, ev);
  if (rc == 1) input_error("Pair distance < table inner cutoff");
  if (rc == 2) input_error("Pair distance > table outer cutoff");
  fforce = factor_lj * fv;
  return factor_lj * ev;
}

}  // namespace ucg
