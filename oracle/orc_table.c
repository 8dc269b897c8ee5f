/* orc_table.c -- oracle table pipeline.  TEST INFRASTRUCTURE (see orc.h).
 *
 * Restates UCG/pair_table_ucgld.cpp:897-1428 (identical copies live in
 * pair_table_ucg_bethe.cpp and pair_table_ucg_bethe_density.cpp):
 *   read_table     :897-1017   param_extract :1067-1102
 *   spline_table   :1047-1065  compute_table :1105-1344
 *   spline         :1375-1404  splint        :1408-1428
 * and the per-pair interpolation block :436-482 (canonical copy single() :1474-1520).
 * BITMAP tables use upstream Pair::init_bitmap (absent from the reference tree), restated in orc_init_bitmap.
 * The file reader restates upstream TableFileReader (absent): comment ('#') and
 * blank lines are skipped while searching for the keyword line; the line after
 * it is the parameter line; ONE raw line is skipped; N data lines follow.
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPSILONR 1.0e-6

static void set_err(char *err, int errlen, const char *msg)
{
  if (err && errlen > 0) {
    strncpy(err, msg, (size_t) errlen - 1);
    err[errlen - 1] = '\0';
  }
}

/* UCG/pair_table_ucgld.cpp:1375-1404 */
void orc_spline(const double *x, const double *y, int n, double yp1, double ypn, double *y2)
{
  int i, k;
  double p, qn, sig, un;
  double *u = (double *) malloc(sizeof(double) * (size_t) n);

  if (yp1 > 0.99e30)
    y2[0] = u[0] = 0.0;
  else {
    y2[0] = -0.5;
    u[0] = (3.0 / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - yp1);
  }
  for (i = 1; i < n - 1; i++) {
    sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    p = sig * y2[i - 1] + 2.0;
    y2[i] = (sig - 1.0) / p;
    u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  if (ypn > 0.99e30)
    qn = un = 0.0;
  else {
    qn = 0.5;
    un = (3.0 / (x[n - 1] - x[n - 2])) * (ypn - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  }
  y2[n - 1] = (un - qn * u[n - 2]) / (qn * y2[n - 2] + 1.0);
  for (k = n - 2; k >= 0; k--) y2[k] = y2[k] * y2[k + 1] + u[k];
  free(u);
}

/* UCG/pair_table_ucgld.cpp:1408-1428 */
double orc_splint(const double *xa, const double *ya, const double *y2a, int n, double x)
{
  int klo = 0, khi = n - 1, k;
  double h, b, a, y;
  while (khi - klo > 1) {
    k = (khi + klo) >> 1;
    if (xa[k] > x)
      khi = k;
    else
      klo = k;
  }
  h = xa[khi] - xa[klo];
  a = (xa[khi] - x) / h;
  b = (x - xa[klo]) / h;
  y = a * ya[klo] + b * ya[khi] +
      ((a * a * a - a) * y2a[klo] + (b * b * b - b) * y2a[khi]) * (h * h) / 6.0;
  return y;
}

static void null_table(orc_table *tb)
{
  memset(tb, 0, sizeof(*tb));
}

void orc_table_free(orc_table *tb)
{
  free(tb->rfile); free(tb->efile); free(tb->ffile); free(tb->e2file); free(tb->f2file);
  free(tb->rsq); free(tb->e); free(tb->f); free(tb->de); free(tb->df); free(tb->e2); free(tb->f2); free(tb->drsq);
  null_table(tb);
}

/* whitespace tokenizer helpers */
static int is_blank_or_comment(const char *line)
{
  while (*line == ' ' || *line == '\t' || *line == '\r' || *line == '\n') line++;
  return (*line == '\0' || *line == '#');
}

static void strip_comment(char *line)
{
  char *h = strchr(line, '#');
  if (h) *h = '\0';
}

static float int_as_float(int i)
{
  float f;
  memcpy(&f, &i, sizeof f);
  return f;
}

static int float_as_int(float f)
{
  int i;
  memcpy(&i, &f, sizeof i);
  return i;
}

/* Pair::init_bitmap of upstream LAMMPS (src/pair.cpp): the table index is made of the low `nexpbits` exponent bits
 * and the top `nmantbits` mantissa bits of the float; masklo / maskhi are the remaining high bits of inner^2 /
 * outer^2 */
int orc_init_bitmap(double inner, double outer, int ntablebits, int *masklo, int *maskhi, int *nmask,
                    int *nshiftbits, char *err, int errlen)
{
  if (ntablebits > (int) sizeof(float) * 8) { set_err(err, errlen, "Too many total bits for bitmapped lookup table"); return 1; }
  int nlowermin = 1;
  while (!((pow(2.0, (double) nlowermin) <= inner * inner) && (pow(2.0, (double) nlowermin + 1.0) > inner * inner))) {
    if (pow(2.0, (double) nlowermin) <= inner * inner) nlowermin++;
    else nlowermin--;
  }
  int nexpbits = 0;
  double required_range = outer * outer / pow(2.0, (double) nlowermin);
  double available_range = 2.0;
  while (available_range < required_range) {
    nexpbits++;
    available_range = pow(2.0, pow(2.0, (double) nexpbits));
  }
  int nmantbits = ntablebits - nexpbits;
  if (nexpbits > (int) sizeof(float) * 8 - 24) { set_err(err, errlen, "Too many exponent bits for lookup table"); return 1; }
  if (nmantbits + 1 > 24) { set_err(err, errlen, "Too many mantissa bits for lookup table"); return 1; }
  if (nmantbits < 3) { set_err(err, errlen, "Too few bits for lookup table"); return 1; }
  *nshiftbits = 24 - (nmantbits + 1);
  int m = 1;
  for (int j = 0; j < ntablebits + *nshiftbits; j++) m *= 2;
  m -= 1;
  *nmask = m;
  *maskhi = float_as_int((float) (outer * outer)) & ~m;
  *masklo = float_as_int((float) (inner * inner)) & ~m;
  return 0;
}

/* param_extract :1067-1102 */
static int param_extract(orc_table *tb, char *line, char *err, int errlen)
{
  tb->ninput = 0;
  tb->rflag = ORC_RNONE;
  tb->fpflag = 0;
  char *save = NULL;
  char *word = strtok_r(line, " \t\r\n", &save);
  while (word) {
    if (strcmp(word, "N") == 0) {
      char *v = strtok_r(NULL, " \t\r\n", &save);
      if (!v) { set_err(err, errlen, "Pair table parameters: missing N value"); return 1; }
      tb->ninput = atoi(v);
    } else if (!strcmp(word, "R") || !strcmp(word, "RSQ") || !strcmp(word, "BITMAP")) {
      if (!strcmp(word, "R")) tb->rflag = ORC_RLINEAR;
      else if (!strcmp(word, "RSQ")) tb->rflag = ORC_RSQ;
      else tb->rflag = ORC_BMP;
      char *v1 = strtok_r(NULL, " \t\r\n", &save);
      char *v2 = strtok_r(NULL, " \t\r\n", &save);
      if (!v1 || !v2) { set_err(err, errlen, "Pair table parameters: missing rlo/rhi"); return 1; }
      tb->rlo = strtod(v1, NULL);
      tb->rhi = strtod(v2, NULL);
    } else if (!strcmp(word, "FPRIME")) {
      tb->fpflag = 1;
      char *v1 = strtok_r(NULL, " \t\r\n", &save);
      char *v2 = strtok_r(NULL, " \t\r\n", &save);
      if (!v1 || !v2) { set_err(err, errlen, "Pair table parameters: missing fplo/fphi"); return 1; }
      tb->fplo = strtod(v1, NULL);
      tb->fphi = strtod(v2, NULL);
    } else {
      char msg[256];
      snprintf(msg, sizeof msg, "Invalid keyword %s in pair table parameters", word);
      set_err(err, errlen, msg);
      return 1;
    }
    word = strtok_r(NULL, " \t\r\n", &save);
  }
  if (tb->ninput == 0) { set_err(err, errlen, "Pair table parameters did not set N"); return 1; }
  return 0;
}

/* read_table :897-1017 (unit conversion factor is 1: no unit_convert in LJ/real runs) */
int orc_table_read(orc_table *tb, const char *file, const char *keyword, char *err, int errlen)
{
  null_table(tb);
  FILE *fp = fopen(file, "r");
  if (!fp) {
    char msg[512];
    snprintf(msg, sizeof msg, "Cannot open pair table file %s", file);
    set_err(err, errlen, msg);
    return 1;
  }
  char line[1024];
  int found = 0;
  while (fgets(line, sizeof line, fp)) {
    if (is_blank_or_comment(line)) continue;
    strip_comment(line);
    char tmp[1024];
    strcpy(tmp, line);
    char *save = NULL;
    char *word = strtok_r(tmp, " \t\r\n", &save);
    if (word && strcmp(word, keyword) == 0) { found = 1; break; }
  }
  if (!found) {
    char msg[512];
    snprintf(msg, sizeof msg, "Did not find keyword %s in table file", keyword);
    set_err(err, errlen, msg);
    fclose(fp);
    return 1;
  }
  /* parameter line = next non-blank, non-comment line */
  int got = 0;
  while (fgets(line, sizeof line, fp)) {
    if (is_blank_or_comment(line)) continue;
    strip_comment(line);
    got = 1;
    break;
  }
  if (!got) { set_err(err, errlen, "Missing pair table parameter line"); fclose(fp); return 1; }
  if (param_extract(tb, line, err, errlen)) { fclose(fp); return 1; }
  /* setup bitmap parameters for table to read in :918-927 */
  int masklo = 0, maskhi = 0, nmask = 0, nshiftbits = 0;
  tb->ntablebits = 0;
  if (tb->rflag == ORC_BMP) {
    while (1 << tb->ntablebits < tb->ninput) tb->ntablebits++;
    if (1 << tb->ntablebits != tb->ninput) {
      set_err(err, errlen, "Bitmapped table is incorrect length in table file");
      fclose(fp);
      return 1;
    }
    if (orc_init_bitmap(tb->rlo, tb->rhi, tb->ntablebits, &masklo, &maskhi, &nmask, &nshiftbits, err, errlen)) {
      fclose(fp);
      return 1;
    }
  }
  tb->rfile = (double *) malloc(sizeof(double) * (size_t) tb->ninput);
  tb->efile = (double *) malloc(sizeof(double) * (size_t) tb->ninput);
  tb->ffile = (double *) malloc(sizeof(double) * (size_t) tb->ninput);

  /* reader.skip_line(): one raw line */
  if (!fgets(line, sizeof line, fp)) { set_err(err, errlen, "Premature end of table file"); fclose(fp); return 1; }

  for (int i = 0; i < tb->ninput; i++) {
    int have = 0;
    while (fgets(line, sizeof line, fp)) {
      if (is_blank_or_comment(line)) continue;
      strip_comment(line);
      have = 1;
      break;
    }
    if (!have) {
      char msg[256];
      snprintf(msg, sizeof msg, "Data missing when parsing pair table '%s' line %d of %d.", keyword, i + 1, tb->ninput);
      set_err(err, errlen, msg);
      fclose(fp);
      return 1;
    }
    char *save = NULL;
    char *t0 = strtok_r(line, " \t\r\n", &save);
    char *t1 = strtok_r(NULL, " \t\r\n", &save);
    char *t2 = strtok_r(NULL, " \t\r\n", &save);
    char *t3 = strtok_r(NULL, " \t\r\n", &save);
    if (!t0 || !t1 || !t2 || !t3) {
      char msg[256];
      snprintf(msg, sizeof msg, "Error parsing pair table '%s' line %d of %d.", keyword, i + 1, tb->ninput);
      set_err(err, errlen, msg);
      fclose(fp);
      return 1;
    }
    double rfile = strtod(t1, NULL);
    tb->efile[i] = 1.0 * strtod(t2, NULL);
    tb->ffile[i] = 1.0 * strtod(t3, NULL);
    double rnew = rfile;
    if (tb->rflag == ORC_RLINEAR)
      rnew = tb->rlo + (tb->rhi - tb->rlo) * i / (tb->ninput - 1);
    else if (tb->rflag == ORC_RSQ) {
      rnew = tb->rlo * tb->rlo + (tb->rhi * tb->rhi - tb->rlo * tb->rlo) * i / (tb->ninput - 1);
      rnew = sqrt(rnew);
    } else if (tb->rflag == ORC_BMP) {
      float fl = int_as_float((i << nshiftbits) | masklo);
      if (fl < tb->rlo * tb->rlo) fl = int_as_float((i << nshiftbits) | maskhi);
      rnew = sqrtf(fl);
    }
    tb->rfile[i] = rnew;
  }
  fclose(fp);
  return 0;
}

/* same as orc_table_read but from arrays already in memory (r recomputed from
 * rflag exactly as read_table does, :954-972) */
int orc_table_from_arrays(orc_table *tb, int ninput, const double *r, const double *e,
                          const double *f, int rflag, double rlo, double rhi, int fpflag,
                          double fplo, double fphi)
{
  null_table(tb);
  tb->ninput = ninput;
  tb->rflag = rflag;
  tb->rlo = rlo;
  tb->rhi = rhi;
  tb->fpflag = fpflag;
  tb->fplo = fplo;
  tb->fphi = fphi;
  tb->rfile = (double *) malloc(sizeof(double) * (size_t) ninput);
  tb->efile = (double *) malloc(sizeof(double) * (size_t) ninput);
  tb->ffile = (double *) malloc(sizeof(double) * (size_t) ninput);
  for (int i = 0; i < ninput; i++) {
    double rnew = r[i];
    if (rflag == ORC_RLINEAR)
      rnew = rlo + (rhi - rlo) * i / (ninput - 1);
    else if (rflag == ORC_RSQ) {
      rnew = rlo * rlo + (rhi * rhi - rlo * rlo) * i / (ninput - 1);
      rnew = sqrt(rnew);
    }
    tb->rfile[i] = rnew;
    tb->efile[i] = e[i];
    tb->ffile[i] = f[i];
  }
  return 0;
}

/* spline_table :1047-1065 */
static void spline_table(orc_table *tb)
{
  tb->e2file = (double *) malloc(sizeof(double) * (size_t) tb->ninput);
  tb->f2file = (double *) malloc(sizeof(double) * (size_t) tb->ninput);

  double ep0 = -tb->ffile[0];
  double epn = -tb->ffile[tb->ninput - 1];
  orc_spline(tb->rfile, tb->efile, tb->ninput, ep0, epn, tb->e2file);

  if (tb->fpflag == 0) {
    tb->fplo = (tb->ffile[1] - tb->ffile[0]) / (tb->rfile[1] - tb->rfile[0]);
    tb->fphi = (tb->ffile[tb->ninput - 1] - tb->ffile[tb->ninput - 2]) /
        (tb->rfile[tb->ninput - 1] - tb->rfile[tb->ninput - 2]);
  }
  double fp0 = tb->fplo;
  double fpn = tb->fphi;
  orc_spline(tb->rfile, tb->ffile, tb->ninput, fp0, fpn, tb->f2file);
}

/* coeff() per-table checks :795-829 + compute_table :1105-1245 */
int orc_table_build(orc_table *tb, int tabstyle, int tablength, double cut, char *err, int errlen)
{
  tb->cut = cut;
  if (tb->ninput <= 1) { set_err(err, errlen, "Invalid pair table length"); return 1; }
  double rlo, rhi;
  if (tb->rflag == 0) {
    rlo = tb->rfile[0];
    rhi = tb->rfile[tb->ninput - 1];
  } else {
    rlo = tb->rlo;
    rhi = tb->rhi;
  }
  if (tb->cut <= rlo || tb->cut > rhi) { set_err(err, errlen, "Pair table cutoff outside of table"); return 1; }
  if (rlo <= 0.0) { set_err(err, errlen, "Invalid pair table lower boundary"); return 1; }

  tb->match = 0;
  if (tabstyle == ORC_LINEAR && tb->ninput == tablength && tb->rflag == ORC_RSQ && tb->rhi == tb->cut)
    tb->match = 1;
  if (tabstyle == ORC_BITMAP && tb->ninput == 1 << tablength && tb->rflag == ORC_BMP && tb->rhi == tb->cut)
    tb->match = 1;
  if (tb->rflag == ORC_BMP && tb->match == 0) {
    set_err(err, errlen, "Bitmapped table in file does not match requested table");
    return 1;
  }

  if (tb->match == 0) spline_table(tb);

  const int tlm1 = tablength - 1;
  double inner;
  if (tb->rflag)
    inner = tb->rlo;
  else
    inner = tb->rfile[0];
  tb->innersq = inner * inner;
  tb->delta = (tb->cut * tb->cut - tb->innersq) / tlm1;
  tb->invdelta = 1.0 / tb->delta;

  if (tabstyle == ORC_LOOKUP) {
    tb->e = (double *) malloc(sizeof(double) * (size_t) tlm1);
    tb->f = (double *) malloc(sizeof(double) * (size_t) tlm1);
    double r, rsq;
    for (int i = 0; i < tlm1; i++) {
      rsq = tb->innersq + (i + 0.5) * tb->delta;
      r = sqrt(rsq);
      tb->e[i] = orc_splint(tb->rfile, tb->efile, tb->e2file, tb->ninput, r);
      tb->f[i] = orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, r) / r;
    }
  }

  if (tabstyle == ORC_LINEAR) {
    tb->rsq = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->e = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->f = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->de = (double *) malloc(sizeof(double) * (size_t) tlm1);
    tb->df = (double *) malloc(sizeof(double) * (size_t) tlm1);
    double r, rsq;
    for (int i = 0; i < tablength; i++) {
      rsq = tb->innersq + i * tb->delta;
      r = sqrt(rsq);
      tb->rsq[i] = rsq;
      if (tb->match) {
        tb->e[i] = tb->efile[i];
        tb->f[i] = tb->ffile[i] / r;
      } else {
        tb->e[i] = orc_splint(tb->rfile, tb->efile, tb->e2file, tb->ninput, r);
        tb->f[i] = orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, r) / r;
      }
    }
    for (int i = 0; i < tlm1; i++) {
      tb->de[i] = tb->e[i + 1] - tb->e[i];
      tb->df[i] = tb->f[i + 1] - tb->f[i];
    }
  }

  if (tabstyle == ORC_SPLINE) {
    tb->rsq = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->e = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->f = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->e2 = (double *) malloc(sizeof(double) * (size_t) tablength);
    tb->f2 = (double *) malloc(sizeof(double) * (size_t) tablength);

    tb->deltasq6 = tb->delta * tb->delta / 6.0;

    double r, rsq;
    for (int i = 0; i < tablength; i++) {
      rsq = tb->innersq + i * tb->delta;
      r = sqrt(rsq);
      tb->rsq[i] = rsq;
      if (tb->match) {
        tb->e[i] = tb->efile[i];
        tb->f[i] = tb->ffile[i] / r;
      } else {
        tb->e[i] = orc_splint(tb->rfile, tb->efile, tb->e2file, tb->ninput, r);
        tb->f[i] = orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, r);
      }
    }

    double ep0 = -tb->f[0] / (2.0 * sqrt(tb->innersq));
    double epn = -tb->f[tlm1] / (2.0 * tb->cut);
    orc_spline(tb->rsq, tb->e, tablength, ep0, epn, tb->e2);

    double fp0, fpn;
    double secant_factor = 0.1;
    if (tb->fpflag)
      fp0 = (tb->fplo / sqrt(tb->innersq) - tb->f[0] / tb->innersq) / (2.0 * sqrt(tb->innersq));
    else {
      double rsq1 = tb->innersq;
      double rsq2 = rsq1 + secant_factor * tb->delta;
      fp0 = (orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, sqrt(rsq2)) / sqrt(rsq2) -
             tb->f[0] / sqrt(rsq1)) /
          (secant_factor * tb->delta);
    }

    if (tb->fpflag && tb->cut == tb->rfile[tb->ninput - 1])
      fpn = (tb->fphi / tb->cut - tb->f[tlm1] / (tb->cut * tb->cut)) / (2.0 * tb->cut);
    else {
      double rsq2 = tb->cut * tb->cut;
      double rsq1 = rsq2 - secant_factor * tb->delta;
      fpn = (tb->f[tlm1] / sqrt(rsq2) -
             orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, sqrt(rsq1)) / sqrt(rsq1)) /
          (secant_factor * tb->delta);
    }

    for (int i = 0; i < tablength; i++) tb->f[i] /= sqrt(tb->rsq[i]);
    orc_spline(tb->rsq, tb->f, tablength, fp0, fpn, tb->f2);
  }

  /* bitmapped linear tables :1247-1340: 2^N bins from inner to cut, spaced in bitmapped manner */
  if (tabstyle == ORC_BITMAP) {
    int masklo, maskhi;
    if (orc_init_bitmap(inner, tb->cut, tablength, &masklo, &maskhi, &tb->nmask, &tb->nshiftbits, err, errlen)) return 1;
    const int ntable = 1 << tablength;
    const int ntablem1 = ntable - 1;
    tb->ntablebits = tablength;
    tb->rsq = (double *) malloc(sizeof(double) * (size_t) ntable);
    tb->e = (double *) malloc(sizeof(double) * (size_t) ntable);
    tb->f = (double *) malloc(sizeof(double) * (size_t) ntable);
    tb->de = (double *) malloc(sizeof(double) * (size_t) ntable);
    tb->df = (double *) malloc(sizeof(double) * (size_t) ntable);
    tb->drsq = (double *) malloc(sizeof(double) * (size_t) ntable);

    float minrsq = int_as_float((0 << tb->nshiftbits) | maskhi);
    float fl;
    double r;
    for (int i = 0; i < ntable; i++) {
      fl = int_as_float((i << tb->nshiftbits) | masklo);
      if (fl < tb->innersq) fl = int_as_float((i << tb->nshiftbits) | maskhi);
      r = sqrtf(fl);
      tb->rsq[i] = fl;
      if (tb->match) {
        tb->e[i] = tb->efile[i];
        tb->f[i] = tb->ffile[i] / r;
      } else {
        tb->e[i] = orc_splint(tb->rfile, tb->efile, tb->e2file, tb->ninput, r);
        tb->f[i] = orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, r) / r;
      }
      minrsq = (minrsq < fl) ? minrsq : fl; /* MIN(minrsq_lookup.f, rsq_lookup.f) */
    }
    tb->innersq = minrsq;

    for (int i = 0; i < ntablem1; i++) {
      tb->de[i] = tb->e[i + 1] - tb->e[i];
      tb->df[i] = tb->f[i + 1] - tb->f[i];
      tb->drsq[i] = 1.0 / (tb->rsq[i + 1] - tb->rsq[i]);
    }
    /* tables are connected periodically between 0 and ntablem1 */
    tb->de[ntablem1] = tb->e[0] - tb->e[ntablem1];
    tb->df[ntablem1] = tb->f[0] - tb->f[ntablem1];
    tb->drsq[ntablem1] = 1.0 / (tb->rsq[0] - tb->rsq[ntablem1]);

    /* the bin holding the largest r gets the deltas towards cut*cut */
    int itablemin = (float_as_int(minrsq) & tb->nmask) >> tb->nshiftbits;
    int itablemax = itablemin - 1;
    if (itablemin == 0) itablemax = ntablem1;
    int itablemaxm1 = itablemax - 1;
    if (itablemax == 0) itablemaxm1 = ntablem1;
    fl = int_as_float((itablemax << tb->nshiftbits) | maskhi);
    if (fl < tb->cut * tb->cut) {
      if (tb->match) {
        tb->de[itablemax] = tb->de[itablemaxm1];
        tb->df[itablemax] = tb->df[itablemaxm1];
        tb->drsq[itablemax] = tb->drsq[itablemaxm1];
      } else {
        fl = (float) (tb->cut * tb->cut);
        r = sqrtf(fl);
        double e_tmp = orc_splint(tb->rfile, tb->efile, tb->e2file, tb->ninput, r);
        double f_tmp = orc_splint(tb->rfile, tb->ffile, tb->f2file, tb->ninput, r) / r;
        tb->de[itablemax] = e_tmp - tb->e[itablemax];
        tb->df[itablemax] = f_tmp - tb->f[itablemax];
        tb->drsq[itablemax] = 1.0 / (fl - tb->rsq[itablemax]);
      }
    }
  }
  return 0;
}

/* the per-table block of the inner loop, UCG/pair_table_ucgld.cpp:436-482
 * (without factor_lj).  Returns 0 ok, 1 rsq < innersq, 2 itable >= tlm1. */
int orc_table_eval(const orc_table *tb, int tabstyle, int tablength, double rsq, double *fval,
                   double *eval)
{
  const int tlm1 = tablength - 1;
  int itable;
  double fraction, value, a, b, evdwl;
  if (rsq < tb->innersq) return 1;
  if (tabstyle == ORC_BITMAP) {
    /* :466-476: the index is cut out of the bits of (float) rsq; no outer-cutoff check in this branch */
    float fl = (float) rsq;
    itable = (float_as_int(fl) & tb->nmask) >> tb->nshiftbits;
    fraction = ((double) fl - tb->rsq[itable]) * tb->drsq[itable];
    *fval = tb->f[itable] + fraction * tb->df[itable];
    *eval = tb->e[itable] + fraction * tb->de[itable];
    return 0;
  }
  itable = (int) ((rsq - tb->innersq) * tb->invdelta);
  if (itable >= tlm1) return 2;
  if (tabstyle == ORC_LOOKUP) {
    value = tb->f[itable];
    evdwl = tb->e[itable];
  } else if (tabstyle == ORC_LINEAR) {
    fraction = (rsq - tb->rsq[itable]) * tb->invdelta;
    value = tb->f[itable] + fraction * tb->df[itable];
    evdwl = tb->e[itable] + fraction * tb->de[itable];
  } else {
    b = (rsq - tb->rsq[itable]) * tb->invdelta;
    a = 1.0 - b;
    value = a * tb->f[itable] + b * tb->f[itable + 1] +
        ((a * a * a - a) * tb->f2[itable] + (b * b * b - b) * tb->f2[itable + 1]) * tb->deltasq6;
    evdwl = a * tb->e[itable] + b * tb->e[itable + 1] +
        ((a * a * a - a) * tb->e2[itable] + (b * b * b - b) * tb->e2[itable + 1]) * tb->deltasq6;
  }
  *fval = value;
  *eval = evdwl;
  return 0;
}
