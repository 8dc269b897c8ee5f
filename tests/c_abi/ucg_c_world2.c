/* ucg_c_world2.c -- a plain-C, multi-PROCESS caller of the decomposed step loop behind include/ucg_hip.h: what
 * `mpirun -np 2 lmp` with `run_style verlet/ucg/gpu comm mpi` does (lammps-ucg-dev_amd/lammps/verlet_ucg_gpu.cpp), without
 * LAMMPS and without MPI.  The process forks BEFORE any GPU call; every rank then
 *   ucg_ctx_create -> units -> atoms_upload (its slice of the beads) -> domain_set -> decomp_set -> pair settings / coeff /
 *   init -> fix ucgld/langevin (seed + rank), fix ucgstate mc (seed + rank), fix nve/ucgld/wall/hard -> ucg_comm_attach_host
 *   (four callbacks on HOST buffers: here over a socket pair per peer, in the glue over MPI) -> md_attach -> md_setup ->
 *   download: tag, f, ucgforce, scores of ITS bricks' beads in ITS local order must equal the bits the test computed with
 *   the oracle's decomposed run (oracle/orc_md.c: orc_world) -> md_run_until(nsteps) -> tag, ucgstate, x, v, ucgl likewise.
 * Exit code 0 = every rank found its bits.  compiled with gcc -std=c99 -Wall -Wextra -Werror (prototype check).
 *
 * case file: text line "UCGWORLD1 n ntypes nsteps px py pz", then lines: table file, settings file, tabstyle, tablength,
 * then raw little-endian arrays:
 *   boxlo[3] boxhi[3] x[n][3] v[n][3] ucgl[n] ucgvl[n] ucgml[n] ucgp[n] mass[ntypes+1]   (double)
 *   type[n] tag[n] mask[n] ucgstate[n]                                                      (int32)
 *   per rank r = 0 .. world-1:
 *     n0 (int32)  tag[n0] (int32)  f[n0][3] ucgforce[n0] scores[n0][2]  (double)          -- after ucg_md_setup
 *     n1 (int32)  tag[n1] ucgstate[n1] (int32)  x[n1][3] v[n1][3] ucgl[n1] (double)       -- after the run
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "ucg_hip.h"

#define MAXW 8

typedef struct {
  int rank, world;
  int fd[MAXW]; /* socket to every peer (fd[rank] unused) */
} Net;

static int failures = 0;

static int xwrite(int fd, const void *buf, size_t n)
{
  const char *p = (const char *) buf;
  while (n) {
    ssize_t k = write(fd, p, n);
    if (k < 0) { if (errno == EINTR) continue; return 1; }
    p += k;
    n -= (size_t) k;
  }
  return 0;
}

static int xread(int fd, void *buf, size_t n)
{
  char *p = (char *) buf;
  while (n) {
    ssize_t k = read(fd, p, n);
    if (k < 0) { if (errno == EINTR) continue; return 1; }
    if (k == 0) return 1;
    p += k;
    n -= (size_t) k;
  }
  return 0;
}

/* blocking pairwise exchange without deadlock: the lower rank of a pair sends first */
static int swap(const Net *N, int peer, const void *out, size_t nout, void *in, size_t nin)
{
  if (N->rank < peer) return xwrite(N->fd[peer], out, nout) || xread(N->fd[peer], in, nin);
  return xread(N->fd[peer], in, nin) || xwrite(N->fd[peer], out, nout);
}

static int cb_alltoallv(void *user, const void *send, const long long *sb, void *recv, const long long *rb, void *stream)
{
  const Net *N = (const Net *) user;
  long long so[MAXW + 1], ro[MAXW + 1];
  if (stream != NULL) return 1; /* ucg_comm_attach_host hands over host memory, completed */
  so[0] = ro[0] = 0;
  for (int r = 0; r < N->world; r++) { so[r + 1] = so[r] + sb[r]; ro[r + 1] = ro[r] + rb[r]; }
  if (sb[N->rank] != rb[N->rank]) return 1;
  memcpy((char *) recv + ro[N->rank], (const char *) send + so[N->rank], (size_t) sb[N->rank]);
  for (int r = 0; r < N->world; r++)
    if (r != N->rank && swap(N, r, (const char *) send + so[r], (size_t) sb[r], (char *) recv + ro[r], (size_t) rb[r])) return 1;
  return 0;
}

static int cb_alltoall_ll(void *user, const long long *send, long long *recv)
{
  const Net *N = (const Net *) user;
  recv[N->rank] = send[N->rank];
  for (int r = 0; r < N->world; r++)
    if (r != N->rank && swap(N, r, send + r, sizeof(long long), recv + r, sizeof(long long))) return 1;
  return 0;
}

static int cb_allreduce_ll(void *user, long long *buf, int n, int op)
{
  const Net *N = (const Net *) user;
  long long *mine = (long long *) malloc((size_t) (n ? n : 1) * sizeof(long long)), *in = (long long *) malloc((size_t) (n ? n : 1) * sizeof(long long));
  int rc = 0;
  memcpy(mine, buf, (size_t) n * sizeof(long long));
  for (int r = 0; r < N->world && !rc; r++) {
    if (r == N->rank) continue;
    rc = swap(N, r, mine, (size_t) n * sizeof(long long), in, (size_t) n * sizeof(long long));
    for (int i = 0; i < n && !rc; i++) buf[i] = op == 0 ? buf[i] + in[i] : op == 1 ? (in[i] > buf[i] ? in[i] : buf[i]) : (in[i] < buf[i] ? in[i] : buf[i]);
  }
  free(mine);
  free(in);
  return rc;
}

static int cb_allreduce_f64(void *user, double *buf, int n, int op)
{
  /* sum in RANK order on every rank, so that all ranks hold the same bits */
  const Net *N = (const Net *) user;
  double *all = (double *) malloc((size_t) (n ? n : 1) * (size_t) N->world * sizeof(double));
  int rc = 0;
  memcpy(all + (size_t) N->rank * (size_t) n, buf, (size_t) n * sizeof(double));
  for (int r = 0; r < N->world && !rc; r++)
    if (r != N->rank) rc = swap(N, r, buf, (size_t) n * sizeof(double), all + (size_t) r * (size_t) n, (size_t) n * sizeof(double));
  for (int i = 0; i < n && !rc; i++) {
    double acc = all[i];
    for (int r = 1; r < N->world; r++) {
      const double v = all[(size_t) r * (size_t) n + (size_t) i];
      acc = op == 0 ? acc + v : op == 1 ? (v > acc ? v : acc) : (v < acc ? v : acc);
    }
    buf[i] = acc;
  }
  free(all);
  return rc;
}

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    int rc_ = (call);                                                                                        \
    if (rc_ != UCG_OK) {                                                                                     \
      fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, rc_, ctx ? ucg_last_error(ctx) : "?"); \
      return 2;                                                                                              \
    }                                                                                                        \
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

static void same_bits(int rank, const char *what, const double *a, const double *b, size_t n)
{
  if (memcmp(a, b, n * sizeof(double)) != 0) {
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += memcmp(a + i, b + i, sizeof(double)) != 0;
    fprintf(stderr, "rank %d MISMATCH %s: %zu of %zu values differ\n", rank, what, bad, n);
    failures++;
  }
}

static void same_ints(int rank, const char *what, const int *a, const int *b, size_t n)
{
  if (memcmp(a, b, n * sizeof(int)) != 0) {
    fprintf(stderr, "rank %d MISMATCH %s\n", rank, what);
    failures++;
  }
}

static void chomp(char *s)
{
  size_t n = strlen(s);
  while (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) s[--n] = 0;
}

typedef struct {
  int n0, n1;
  int *tag0, *tag1, *st1;
  double *f0, *uf0, *s0, *x1, *v1, *l1;
} Expect;

static int run_rank(Net *N, const char *casefile)
{
  const int rank = N->rank;
  ucg_ctx *ctx = NULL;
  ucg_pair *pair = NULL;
  FILE *fh = fopen(casefile, "rb");
  if (!fh) { perror(casefile); return 2; }
  char line[4096], tabfile[4096], conffile[4096], tabstyle[64], tablen[64];
  int n = 0, ntypes = 0, nsteps = 0, grid[3] = {1, 1, 1};
  if (!fgets(line, sizeof line, fh) ||
      sscanf(line, "UCGWORLD1 %d %d %d %d %d %d", &n, &ntypes, &nsteps, &grid[0], &grid[1], &grid[2]) != 6) {
    fprintf(stderr, "bad case header\n");
    return 2;
  }
  if (grid[0] * grid[1] * grid[2] != N->world) { fprintf(stderr, "grid does not match the process count\n"); return 2; }
  if (!fgets(tabfile, sizeof tabfile, fh) || !fgets(conffile, sizeof conffile, fh) || !fgets(tabstyle, sizeof tabstyle, fh) ||
      !fgets(tablen, sizeof tablen, fh)) return 2;
  chomp(tabfile); chomp(conffile); chomp(tabstyle); chomp(tablen);
  const size_t NN = (size_t) n;
  double *boxlo = rd(fh, 3), *boxhi = rd(fh, 3), *x = rd(fh, 3 * NN), *v = rd(fh, 3 * NN), *ucgl = rd(fh, NN);
  double *ucgvl = rd(fh, NN), *ucgml = rd(fh, NN), *ucgp = rd(fh, NN), *mass = rd(fh, (size_t) ntypes + 1);
  int *type = ri(fh, NN), *tag = ri(fh, NN), *mask = ri(fh, NN), *ucgstate = ri(fh, NN);
  Expect E;
  memset(&E, 0, sizeof E);
  for (int r = 0; r < N->world; r++) { /* every rank reads past the others' blocks and keeps its own */
    Expect T;
    int *c = ri(fh, 1);
    T.n0 = *c; free(c);
    T.tag0 = ri(fh, (size_t) T.n0);
    T.f0 = rd(fh, 3 * (size_t) T.n0); T.uf0 = rd(fh, (size_t) T.n0); T.s0 = rd(fh, 2 * (size_t) T.n0);
    c = ri(fh, 1);
    T.n1 = *c; free(c);
    T.tag1 = ri(fh, (size_t) T.n1); T.st1 = ri(fh, (size_t) T.n1);
    T.x1 = rd(fh, 3 * (size_t) T.n1); T.v1 = rd(fh, 3 * (size_t) T.n1); T.l1 = rd(fh, (size_t) T.n1);
    if (r == rank) E = T;
  }
  fclose(fh);

  const double special[4] = {1.0, 1.0, 1.0, 1.0};
  /* the rank's share of the input: any split will do, the library sends every bead to the brick that owns it */
  const int lo = (int) ((long long) rank * n / N->world), hi = (int) ((long long) (rank + 1) * n / N->world), nl = hi - lo;
  CHECK(ucg_ctx_create(0, &ctx)); /* all ranks of this test share the one GPU of the box; a production run has one each */
  CHECK(ucg_ctx_set_units(ctx, 1.0, 1.0, 1.0, 0.004, special));
  CHECK(ucg_atoms_upload(ctx, nl, 0, ntypes, x + 3 * (size_t) lo, v + 3 * (size_t) lo, type + lo, tag + lo, mask + lo, ucgstate + lo,
                         ucgl + lo, ucgvl + lo, ucgml + lo, ucgp + lo, mass));
  CHECK(ucg_domain_set(ctx, boxlo, boxhi, 2.5, 0.3, 2, 0, 1));
  CHECK(ucg_decomp_set(ctx, grid, rank));
  CHECK(ucg_pair_create(ctx, UCG_STYLE_UCGLD, &pair));
  const char *sargv[3] = {tabstyle, tablen, conffile};
  CHECK(ucg_pair_settings(pair, 3, sargv));
  const char *cargv[16] = {"1", "1", "2", "2", tabfile, "UCG_00", "2.5", tabfile, "UCG_01", "2.5",
                           tabfile, "UCG_10", "2.5", tabfile, "UCG_11", "2.5"};
  CHECK(ucg_pair_coeff(pair, ntypes, 16, cargv));
  CHECK(ucg_pair_init(pair, ntypes, 1.0));
  CHECK(ucg_fix_langevin_create(ctx, 1.0, 1.0, 1.0, 48279, rank)); /* RanMars(seed + me), UCG/fix_ucgld_langevin.cpp:85 */
  CHECK(ucg_fix_ucgstate_create(ctx, 0, 1, 9127, 0.3, rank));       /* UCG/fix_ucgstate.cpp:62 */
  CHECK(ucg_fix_nve_wall_hard_set(ctx, 0, 0.1));
  ucg_comm_ops ops;
  ops.user = N;
  ops.rank = rank;
  ops.world = N->world;
  ops.alltoallv = cb_alltoallv;
  ops.alltoall_ll = cb_alltoall_ll;
  ops.allreduce_ll = cb_allreduce_ll;
  ops.allreduce_f64 = cb_allreduce_f64;
  CHECK(ucg_comm_attach_host(ctx, &ops));
  int tr[4] = {-1, -1, -1, -1};
  CHECK(ucg_comm_transport(ctx, tr));
  if (tr[0] != 0 || tr[3] != 1) { fprintf(stderr, "rank %d: transport report %d %d %d %d\n", rank, tr[0], tr[1], tr[2], tr[3]); failures++; }
  CHECK(ucg_md_attach(ctx, pair, 2, 1, 1));
  CHECK(ucg_md_setup(ctx, nsteps));

  int nloc = 0, ngh = 0;
  CHECK(ucg_atoms_counts(ctx, &nloc, &ngh));
  if (nloc != E.n0) { fprintf(stderr, "rank %d: %d beads after setup, expected %d\n", rank, nloc, E.n0); failures++; }
  else {
    const size_t M = (size_t) nloc;
    int *gtag = (int *) malloc((M ? M : 1) * sizeof(int));
    double *gf = (double *) malloc((3 * M + 1) * sizeof(double)), *guf = (double *) malloc((M ? M : 1) * sizeof(double));
    double *gs = (double *) malloc((2 * M + 1) * sizeof(double));
    CHECK(ucg_atoms_download(ctx, 0, NULL, NULL, gf, NULL, gtag, NULL, NULL, NULL, NULL, NULL, NULL, guf, gs));
    same_ints(rank, "tag after setup", gtag, E.tag0, M);
    same_bits(rank, "f after setup", gf, E.f0, 3 * M);
    same_bits(rank, "ucgforce after setup", guf, E.uf0, M);
    same_bits(rank, "scores after setup", gs, E.s0, 2 * M);
    free(gtag); free(gf); free(guf); free(gs);
  }

  /* the run in three calls, the last step of each with energy output (what Output::next makes of `thermo`) */
  const int a = nsteps / 3, b = nsteps / 3, c = nsteps - a - b;
  CHECK(ucg_md_run_until(ctx, a, 1));
  CHECK(ucg_md_run_until(ctx, b, 0));
  CHECK(ucg_md_run_until(ctx, c, 1));
  CHECK(ucg_pair_check_errors(pair));
  CHECK(ucg_atoms_counts(ctx, &nloc, &ngh));
  if (nloc != E.n1) { fprintf(stderr, "rank %d: %d beads after the run, expected %d\n", rank, nloc, E.n1); failures++; }
  else {
    const size_t M = (size_t) nloc;
    int *gtag = (int *) malloc((M ? M : 1) * sizeof(int)), *gst = (int *) malloc((M ? M : 1) * sizeof(int));
    double *gx = (double *) malloc((3 * M + 1) * sizeof(double)), *gv = (double *) malloc((3 * M + 1) * sizeof(double));
    double *gl = (double *) malloc((M ? M : 1) * sizeof(double));
    CHECK(ucg_atoms_download(ctx, 0, gx, gv, NULL, NULL, gtag, gst, NULL, gl, NULL, NULL, NULL, NULL, NULL));
    same_ints(rank, "tag after the run", gtag, E.tag1, M);
    same_ints(rank, "ucgstate after the run", gst, E.st1, M);
    same_bits(rank, "x after the run", gx, E.x1, 3 * M);
    same_bits(rank, "v after the run", gv, E.v1, 3 * M);
    same_bits(rank, "ucgl after the run", gl, E.l1, M);
    free(gtag); free(gst); free(gx); free(gv); free(gl);
  }
  long long info[16];
  CHECK(ucg_md_info(ctx, info));
  printf("rank %d: %d -> %d beads, %lld re-neighbourings, %d mismatching groups\n", rank, E.n0, E.n1, info[1], failures);
  CHECK(ucg_comm_detach(ctx));
  ucg_pair_destroy(pair);
  ucg_ctx_destroy(ctx);
  return failures ? 1 : 0;
}

int main(int argc, char **argv)
{
  if (argc < 3) { fprintf(stderr, "usage: %s casefile world\n", argv[0]); return 2; }
  const int world = atoi(argv[2]);
  if (world < 1 || world > MAXW) return 2;
  if (ucg_abi_version() != UCG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  /* one socket pair per pair of ranks, made before the fork so that every process inherits its ends */
  int sv[MAXW][MAXW][2];
  for (int i = 0; i < world; i++)
    for (int j = i + 1; j < world; j++)
      if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv[i][j]) != 0) { perror("socketpair"); return 2; }
  pid_t kids[MAXW];
  int myrank = 0;
  for (int r = 1; r < world; r++) { /* fork BEFORE any HIP call: a forked HIP runtime is not usable */
    kids[r] = fork();
    if (kids[r] < 0) { perror("fork"); return 2; }
    if (kids[r] == 0) { myrank = r; break; }
  }
  Net N;
  N.rank = myrank;
  N.world = world;
  for (int r = 0; r < world; r++) N.fd[r] = -1;
  for (int i = 0; i < world; i++)
    for (int j = i + 1; j < world; j++) {
      if (i == myrank) { N.fd[j] = sv[i][j][0]; close(sv[i][j][1]); }
      else if (j == myrank) { N.fd[i] = sv[i][j][1]; close(sv[i][j][0]); }
      else { close(sv[i][j][0]); close(sv[i][j][1]); }
    }
  int rc = run_rank(&N, argv[1]);
  fflush(stdout);
  if (myrank != 0) _exit(rc);
  for (int r = 1; r < world; r++) {
    int st = 0;
    if (waitpid(kids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      fprintf(stderr, "rank %d ended with status %d\n", r, st);
      rc = rc ? rc : 1;
    }
  }
  if (rc == 0) printf("all %d ranks found the decomposed oracle's bits\n", world);
  return rc;
}
