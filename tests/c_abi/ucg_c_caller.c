/* ucg_c_caller.c -- a plain-C caller of include/ucg_hip.h (compiled with gcc -std=c99 -Wall -Werror, linked against
 * libucg_hip.so): the first consumer that checks the PROTOTYPES of the C ABI, not only its symbol names.
 *
 * It plays the role of the LAMMPS-side glue for one deck (tests/test_c_abi.py writes the case file from the committed
 * golden vectors, tests/golden/ucg_golden.json):
 *   A. resident path: ctx -> units -> atoms_upload -> domain_set -> pair settings / coeff / init -> fix ucgld/langevin,
 *      fix ucgstate -> md_attach -> md_setup -> atoms_download: f, ucgforce, scores, ucgp must equal the golden bits;
 *      md_run(10) -> x, v, ucgl, ucgstate must equal the golden bits.
 *   B. drop-in path of a host-built list: atoms (+ ghosts) and the full list of a second context come from the caller's
 *      own arrays (here: downloaded from context A after the run) -> atoms_upload -> neigh_upload_full ->
 *      force_clear -> pair_compute -> download: f, scores equal context A's bit for bit, the energy to 1e-12.
 *   C. drop-in path hook by hook: a third context, the caller's (pinned) arrays bound as host mirrors (ucg_host_bind);
 *      Verlet::setup, then per step the hooks in upstream Verlet's order, one ABI call each -- initial_integrate ->
 *      re-neighbour decision on the device -> [ucg_host_sync of what exchange / borders read, re-neighbouring | forward
 *      halo] -> Pair::compute -> fix ucgld/langevin post_force -> fix ucgstate post_force -> final_integrate; nothing
 *      crosses PCIe on an ordinary step.  After the run ucg_host_sync(all): the MIRRORS must hold the golden bits.
 * Exit code 0 = all equal; every mismatch is printed.
 *
 * case file: text line "UCGCASE1 n ntypes style nsteps", then lines: table file, settings file, tabstyle, tablength,
 * "extra" words (one line, may be empty), then raw little-endian arrays in this order:
 *   boxlo[3] boxhi[3] x[n][3] v[n][3] ucgl[n] ucgvl[n] ucgml[n] ucgp[n] mass[ntypes+1]   (double)
 *   type[n] tag[n] mask[n] ucgstate[n]                                                      (int32)
 *   expected after setup: tag[n] (int32), f[n][3] scores[n][2] ucgforce[n] ucgp[n]          (double)
 *   expected after the run: tag[n] ucgstate[n] (int32), x[n][3] v[n][3] ucgl[n]             (double)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ucg_hip.h"

/* libamdhip64: pin the caller's arrays so that the mirror copies run at PCIe speed (flags 0 = hipHostRegisterDefault) */
int hipHostRegister(void *ptr, size_t bytes, unsigned int flags);
int hipHostUnregister(void *ptr);

static int failures = 0;

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != UCG_OK) {                                                                     \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? ucg_last_error(ctx) : "?"); \
      return 2;                                                                              \
    }                                                                                        \
  } while (0)

static double *rd(FILE *fh, size_t n)
{
  double *p = (double *) malloc((n ? n : 1) * sizeof(double));
  if (fread(p, sizeof(double), n, fh) != n) { fprintf(stderr, "case file too short\n"); exit(3); }
  return p;
}

static int *ri(FILE *fh, size_t n)
{
  int *p = (int *) malloc((n ? n : 1) * sizeof(int));
  if (fread(p, sizeof(int), n, fh) != n) { fprintf(stderr, "case file too short\n"); exit(3); }
  return p;
}

static void same_bits(const char *what, const double *a, const double *b, size_t n)
{
  if (memcmp(a, b, n * sizeof(double)) != 0) {
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += memcmp(a + i, b + i, sizeof(double)) != 0;
    fprintf(stderr, "MISMATCH %s: %zu of %zu values differ\n", what, bad, n);
    failures++;
  }
}

static void same_ints(const char *what, const int *a, const int *b, size_t n)
{
  if (memcmp(a, b, n * sizeof(int)) != 0) {
    fprintf(stderr, "MISMATCH %s\n", what);
    failures++;
  }
}

static void chomp(char *s)
{
  size_t n = strlen(s);
  while (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) s[--n] = 0;
}

int main(int argc, char **argv)
{
  ucg_ctx *ctx = NULL, *ctx2 = NULL;
  ucg_pair *pair = NULL, *pair2 = NULL;
  if (argc < 2) { fprintf(stderr, "usage: %s casefile\n", argv[0]); return 2; }
  FILE *fh = fopen(argv[1], "rb");
  if (!fh) { perror(argv[1]); return 2; }
  char line[4096], tabfile[4096], conffile[4096], tabstyle[64], tablen[64], extra[1024];
  int n = 0, ntypes = 0, style = 0, nsteps = 0;
  if (!fgets(line, sizeof line, fh) || sscanf(line, "UCGCASE1 %d %d %d %d", &n, &ntypes, &style, &nsteps) != 4) {
    fprintf(stderr, "bad case header\n");
    return 2;
  }
  if (!fgets(tabfile, sizeof tabfile, fh) || !fgets(conffile, sizeof conffile, fh) || !fgets(tabstyle, sizeof tabstyle, fh) ||
      !fgets(tablen, sizeof tablen, fh) || !fgets(extra, sizeof extra, fh)) return 2;
  chomp(tabfile); chomp(conffile); chomp(tabstyle); chomp(tablen); chomp(extra);
  const size_t N = (size_t) n;
  double *boxlo = rd(fh, 3), *boxhi = rd(fh, 3), *x = rd(fh, 3 * N), *v = rd(fh, 3 * N), *ucgl = rd(fh, N);
  double *ucgvl = rd(fh, N), *ucgml = rd(fh, N), *ucgp = rd(fh, N), *mass = rd(fh, (size_t) ntypes + 1);
  int *type = ri(fh, N), *tag = ri(fh, N), *mask = ri(fh, N), *ucgstate = ri(fh, N);
  int *e0_tag = ri(fh, N);
  double *e0_f = rd(fh, 3 * N), *e0_s = rd(fh, 2 * N), *e0_uf = rd(fh, N), *e0_p = rd(fh, N);
  int *e1_tag = ri(fh, N), *e1_st = ri(fh, N);
  double *e1_x = rd(fh, 3 * N), *e1_v = rd(fh, 3 * N), *e1_l = rd(fh, N);
  fclose(fh);

  if (ucg_abi_version() != UCG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  const double special[4] = {1.0, 1.0, 1.0, 1.0};
  CHECK(ucg_ctx_create(0, &ctx));
  CHECK(ucg_ctx_set_units(ctx, 1.0, 1.0, 1.0, 0.004, special));
  CHECK(ucg_atoms_upload(ctx, n, 0, ntypes, x, v, type, tag, mask, ucgstate, ucgl, ucgvl, ucgml, ucgp, mass));
  CHECK(ucg_domain_set(ctx, boxlo, boxhi, 2.5, 0.3, 1, 0, 1));
  CHECK(ucg_pair_create(ctx, style, &pair));
  /* pair_style <tabstyle> <N> <settings file> [extra words]; pair_coeff 1 1 2 2 (file keyword cutoff) x 4 */
  const char *sargv[16];
  int sargc = 0;
  sargv[sargc++] = tabstyle;
  sargv[sargc++] = tablen;
  sargv[sargc++] = conffile;
  for (char *w = strtok(extra, " "); w && sargc < 16; w = strtok(NULL, " ")) sargv[sargc++] = w;
  CHECK(ucg_pair_settings(pair, sargc, sargv));
  const char *cargv[16] = {"1", "1", "2", "2", tabfile, "UCG_00", "2.5", tabfile, "UCG_01", "2.5",
                           tabfile, "UCG_10", "2.5", tabfile, "UCG_11", "2.5"};
  CHECK(ucg_pair_coeff(pair, ntypes, 16, cargv));
  CHECK(ucg_pair_init(pair, ntypes, 1.0));
  if (ucg_pair_cutforce(pair) != 2.5) { fprintf(stderr, "cutforce %g\n", ucg_pair_cutforce(pair)); failures++; }
  if (style == UCG_STYLE_UCGLD) {
    CHECK(ucg_fix_langevin_create(ctx, 1.0, 1.0, 1.0, 48279, 0));
    CHECK(ucg_fix_ucgstate_create(ctx, 1, 0, 0, 0.01, 0));
  } else {
    CHECK(ucg_fix_ucgstate_create(ctx, 0, 1, 4242, 0.3, 0));
  }
  CHECK(ucg_md_attach(ctx, pair, 1, style == UCG_STYLE_UCGLD, 1));
  CHECK(ucg_md_setup(ctx, nsteps));

  int nl = 0, ng = 0;
  CHECK(ucg_atoms_counts(ctx, &nl, &ng));
  if (nl != n) { fprintf(stderr, "nlocal %d != %d\n", nl, n); return 2; }
  const size_t NA = (size_t) nl + (size_t) ng;
  double *gx = (double *) malloc(3 * NA * sizeof(double)), *gv = (double *) malloc(3 * N * sizeof(double));
  double *gf = (double *) malloc(3 * N * sizeof(double)), *gs = (double *) malloc(2 * N * sizeof(double));
  double *gl = (double *) malloc(NA * sizeof(double)), *gvl = (double *) malloc(N * sizeof(double));
  double *gml = (double *) malloc(N * sizeof(double)), *gp = (double *) malloc(NA * sizeof(double));
  double *guf = (double *) malloc(N * sizeof(double));
  int *gtype = (int *) malloc(NA * sizeof(int)), *gtag = (int *) malloc(NA * sizeof(int));
  int *gst = (int *) malloc(NA * sizeof(int)), *gns = (int *) malloc(N * sizeof(int));
  CHECK(ucg_atoms_download(ctx, 1, gx, gv, gf, gtype, gtag, gst, gns, gl, gvl, gml, gp, guf, gs));
  same_ints("setup tag", gtag, e0_tag, N);
  same_bits("setup f", gf, e0_f, 3 * N);
  same_bits("setup scores", gs, e0_s, 2 * N);
  same_bits("setup ucgforce", guf, e0_uf, N);
  same_bits("setup ucgp", gp, e0_p, N);

  CHECK(ucg_md_run(ctx, nsteps, 0));
  CHECK(ucg_pair_check_errors(pair));
  CHECK(ucg_atoms_download(ctx, 0, gx, gv, NULL, NULL, gtag, gst, NULL, gl, NULL, NULL, NULL, NULL, NULL));
  same_ints("run tag", gtag, e1_tag, N);
  same_ints("run ucgstate", gst, e1_st, N);
  same_bits("run x", gx, e1_x, 3 * N);
  same_bits("run v", gv, e1_v, 3 * N);
  same_bits("run ucgl", gl, e1_l, N);
  /* B: the same beads + ghosts and the list as a caller's own host arrays, on a second context */
  if (style != UCG_STYLE_BETHE_DENSITY) {
    CHECK(ucg_atoms_download(ctx, 1, gx, gv, gf, gtype, gtag, gst, gns, gl, gvl, gml, gp, guf, gs));
    int inum = 0;
    long long total = 0;
    CHECK(ucg_neigh_download(ctx, &inum, NULL, NULL, NULL, 0, &total));
    int *numneigh = (int *) malloc(N * sizeof(int)), *neigh = (int *) malloc((size_t) (total ? total : 1) * sizeof(int));
    long long *first = (long long *) malloc(N * sizeof(long long));
    CHECK(ucg_neigh_download(ctx, &inum, numneigh, first, neigh, total, &total));
    int *gmask = (int *) malloc(N * sizeof(int));
    for (int i = 0; i < n; i++) gmask[i] = 1;
    /* both contexts evaluate the state the run ended in (ghosts as the last halo left them) */
    ucg_ctx *keep = ctx;
    ctx = NULL;
    if (ucg_ctx_create(0, &ctx2) != UCG_OK) { fprintf(stderr, "second context\n"); return 2; }
    ctx = ctx2;
    CHECK(ucg_ctx_set_units(ctx2, 1.0, 1.0, 1.0, 0.004, special));
    CHECK(ucg_atoms_upload(ctx2, nl, ng, ntypes, gx, gv, gtype, gtag, gmask, gst, gl, gvl, gml, gp, mass));
    CHECK(ucg_neigh_upload_full(ctx2, inum, numneigh, first, neigh));
    CHECK(ucg_pair_create(ctx2, style, &pair2));
    CHECK(ucg_pair_settings(pair2, sargc, sargv));
    CHECK(ucg_pair_coeff(pair2, ntypes, 16, cargv));
    CHECK(ucg_pair_init(pair2, ntypes, 1.0));
    CHECK(ucg_force_clear(ctx2));
    double e2 = 0.0, vir2[6];
    CHECK(ucg_pair_compute(pair2, 1, 1, &e2, vir2));
    CHECK(ucg_pair_check_errors(pair2));
    double *f2 = (double *) malloc(3 * N * sizeof(double)), *s2 = (double *) malloc(2 * N * sizeof(double));
    CHECK(ucg_atoms_download(ctx2, 0, NULL, NULL, f2, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, s2));
    ctx = keep;
    double e1 = 0.0, vir1[6];
    CHECK(ucg_pair_compute(pair, 1, 1, &e1, vir1));
    double *f1 = (double *) malloc(3 * N * sizeof(double)), *s1 = (double *) malloc(2 * N * sizeof(double));
    CHECK(ucg_atoms_download(ctx, 0, NULL, NULL, f1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, s1));
    same_bits("uploaded list vs device list: f", f2, f1, 3 * N);
    same_bits("uploaded list vs device list: scores", s2, s1, 2 * N);
    /* (the energy is a block-wise double reduction: its last bits follow the order of the rows, which an uploaded list
       does not share with the device builder's; forces and scores do not depend on any order or are summed in row order) */
    if (!(fabs(e1 - e2) <= 1e-12 * fabs(e1))) { fprintf(stderr, "MISMATCH energy %.17g vs %.17g\n", e1, e2); failures++; }
    ucg_pair_destroy(pair2);
    ucg_ctx_destroy(ctx2);
  }

  /* C: hook by hook with bound host mirrors */
  {
    ucg_ctx *keep = ctx, *ctx3 = NULL;
    ucg_pair *pair3 = NULL;
    ctx = NULL;
    if (ucg_ctx_create(0, &ctx3) != UCG_OK) { fprintf(stderr, "third context\n"); return 2; }
    ctx = ctx3;
    CHECK(ucg_ctx_set_units(ctx3, 1.0, 1.0, 1.0, 0.004, special));
    CHECK(ucg_atoms_upload(ctx3, n, 0, ntypes, x, v, type, tag, mask, ucgstate, ucgl, ucgvl, ucgml, ucgp, mass));
    CHECK(ucg_domain_set(ctx3, boxlo, boxhi, 2.5, 0.3, 1, 0, 1));
    CHECK(ucg_pair_create(ctx3, style, &pair3));
    CHECK(ucg_pair_settings(pair3, sargc, sargv));
    CHECK(ucg_pair_coeff(pair3, ntypes, 16, cargv));
    CHECK(ucg_pair_init(pair3, ntypes, 1.0));
    const int lang = style == UCG_STYLE_UCGLD;
    if (lang) {
      CHECK(ucg_fix_langevin_create(ctx3, 1.0, 1.0, 1.0, 48279, 0));
      CHECK(ucg_fix_ucgstate_create(ctx3, 1, 0, 0, 0.01, 0));
    } else {
      CHECK(ucg_fix_ucgstate_create(ctx3, 0, 1, 4242, 0.3, 0));
    }
    /* the caller's arrays of the owned atoms (LAMMPS' atom->x ... ), pinned */
    double *mx = (double *) calloc(3 * N, sizeof(double)), *mv = (double *) calloc(3 * N, sizeof(double));
    double *mf = (double *) calloc(3 * N, sizeof(double)), *ms = (double *) calloc(2 * N, sizeof(double));
    double *ml = (double *) calloc(N, sizeof(double)), *mvl = (double *) calloc(N, sizeof(double));
    double *mp = (double *) calloc(N, sizeof(double)), *muf = (double *) calloc(N, sizeof(double));
    int *mst = (int *) calloc(N, sizeof(int)), *mns = (int *) calloc(N, sizeof(int));
    int pinned = 0;
    pinned += hipHostRegister(mx, 3 * N * sizeof(double), 0) == 0;
    pinned += hipHostRegister(mv, 3 * N * sizeof(double), 0) == 0;
    pinned += hipHostRegister(ml, N * sizeof(double), 0) == 0;
    CHECK(ucg_host_bind(ctx3, mx, mv, mf, mst, mns, ml, mvl, mp, muf, ms));
    CHECK(ucg_md_attach(ctx3, pair3, 1, lang, 1));
    CHECK(ucg_md_setup(ctx3, nsteps)); /* Verlet::setup(): lists, first forces, the fixes' setup */
    long long nre = 0, tr0[2], tr1[2];
    CHECK(ucg_host_status(ctx3, NULL, NULL, tr0));
    for (int s = 0; s < nsteps; s++) {
      CHECK(ucg_md_set_timestep(ctx3, (long long) s + 1));
      CHECK(ucg_fix_nve_initial(ctx3, 1));
      int due = 0, flag = 0;
      CHECK(ucg_decide_local(ctx3, &due, &flag));
      if (due && flag) {
        CHECK(ucg_host_sync(ctx3, UCG_F_X | UCG_F_V | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGP));
        CHECK(ucg_neigh_rebuild(ctx3));
        nre++;
      } else {
        CHECK(ucg_halo_forward(ctx3));
      }
      CHECK(ucg_pair_compute(pair3, 0, 0, NULL, NULL));
      if (lang) CHECK(ucg_fix_langevin_post_force(ctx3, 1, (long long) s + 1, 0, nsteps));
      CHECK(ucg_fix_ucgstate_post_force(ctx3));
      CHECK(ucg_fix_nve_final(ctx3, 1));
    }
    CHECK(ucg_pair_check_errors(pair3));
    CHECK(ucg_host_status(ctx3, NULL, NULL, tr1));
    if (tr1[0] != tr0[0] || tr1[1] - tr0[1] != nre) {
      fprintf(stderr, "MISMATCH transfers: %lld uploads, %lld downloads for %lld re-neighbourings\n", tr1[0] - tr0[0],
              tr1[1] - tr0[1], nre);
      failures++;
    }
    CHECK(ucg_host_sync(ctx3, UCG_F_ALL));
    CHECK(ucg_atoms_download(ctx3, 0, NULL, NULL, NULL, NULL, gtag, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    same_ints("hooks tag", gtag, e1_tag, N);
    same_ints("hooks mirror ucgstate", mst, e1_st, N);
    same_bits("hooks mirror x", mx, e1_x, 3 * N);
    same_bits("hooks mirror v", mv, e1_v, 3 * N);
    same_bits("hooks mirror ucgl", ml, e1_l, N);
    printf("ucg_c_caller: hook-by-hook run: %lld re-neighbourings, %lld downloads, %lld uploads, %d of 3 arrays pinned\n", nre,
           tr1[1] - tr0[1], tr1[0] - tr0[0], pinned);
    CHECK(ucg_host_bind(ctx3, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    hipHostUnregister(mx);
    hipHostUnregister(mv);
    hipHostUnregister(ml);
    ucg_pair_destroy(pair3);
    ucg_ctx_destroy(ctx3);
    ctx = keep;
  }

  long long info[16];
  CHECK(ucg_md_info(ctx, info));
  printf("ucg_c_caller: %d beads, %lld ghosts, %lld list entries, %lld rebuilds, %d mismatching groups\n", n, info[3],
         info[4], info[1], failures);
  ucg_pair_destroy(pair);
  ucg_ctx_destroy(ctx);
  return failures ? 1 : 0;
}
