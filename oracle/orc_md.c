/* orc_md.c -- oracle MD driver: periodic box, ghosts, bins, neighbour lists and
 * the Verlet step order.  TEST INFRASTRUCTURE (see orc.h).
 *
 * None of this is in /root/reference: ghosts, binning, neighbour lists, atom
 * sorting and the integrator loop are upstream LAMMPS (Comm, Neighbor, AtomSort,
 * Verlet), absent here.  The STEP ORDER restates upstream Verlet::setup()/run()
 * as summarised in SURVEY.md section 3.1; the data-management steps follow the
 * "ucg-rebuild-v1" specification in DESIGN.md, which the HIP path implements
 * with the same IEEE operations so that whole trajectories can be compared bit
 * for bit:
 *
 *   rebuild: (1) wrap owned atoms into the box (Domain::pbc form);
 *            (2) bins of ~cutneigh/2 over the box extended by cutneigh;
 *            (3) sort owned atoms by (Morton code of the bin, tag);
 *            (4) ghosts = every periodic image x + s*prd (s in {-1,0,1}^3 \ 0) that
 *                falls in the extended box, sorted by (Morton code of the bin, tag, shift code);
 *            (5) full list rows: stencil bins in (dz,dy,dx) ascending order, owned
 *                atoms of the bin then ghost atoms of the bin, entry kept when
 *                rsq < cutneigh^2; bit 29 = (tag_k <= tag_m); the row is then stably
 *                partitioned into 4 distance classes (inside the force cutoff, then
 *                thirds of the skin shell).
 *   decide : every `every` steps, rebuild when any owned atom moved more than
 *            skin/2 since the last build (Neighbor::check_distance form).
 */
#include "orc_md.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *xrealloc(void *p, size_t n)
{
  void *q = realloc(p, n ? n : 1);
  if (!q) { fprintf(stderr, "orc_md: out of memory\n"); abort(); }
  return q;
}

orc_sim *orc_sim_create(int natoms, const double *boxlo, const double *boxhi, double cutforce,
                        double skin, int ntypes)
{
  orc_sim *s = (orc_sim *) calloc(1, sizeof(orc_sim));
  for (int d = 0; d < 3; d++) {
    s->boxlo[d] = boxlo[d];
    s->boxhi[d] = boxhi[d];
    s->prd[d] = boxhi[d] - boxlo[d];
    s->sublo[d] = boxlo[d];
    s->subhi[d] = boxhi[d];
  }
  s->cutforce = cutforce;
  s->skin = skin;
  s->cutneigh = cutforce + skin;
  s->every = 1;
  s->delay = 0;
  s->check = 1;
  s->ntypes = ntypes;
  s->boltz = s->ftm2v = s->mvv2e = 1.0;
  s->dt = 0.005;
  s->groupbit = 1;
  s->mode = 1;
  s->a.nlocal = natoms;
  s->a.nghost = 0;
  s->a.mass = (double *) calloc((size_t) ntypes + 1, sizeof(double));
  s->nmax = 0;
  orc_sim_grow(s, natoms + natoms / 2 + 1024);
  return s;
}

void orc_sim_grow(orc_sim *s, int nmax)
{
  if (nmax <= s->nmax) return;
  orc_atoms *a = &s->a;
  const size_t n = (size_t) nmax;
  a->x = (double *) xrealloc(a->x, 3 * n * sizeof(double));
  a->v = (double *) xrealloc(a->v, 3 * n * sizeof(double));
  a->f = (double *) xrealloc(a->f, 3 * n * sizeof(double));
  a->type = (int *) xrealloc(a->type, n * sizeof(int));
  a->tag = (int *) xrealloc(a->tag, n * sizeof(int));
  a->mask = (int *) xrealloc(a->mask, n * sizeof(int));
  a->ucgstate = (int *) xrealloc(a->ucgstate, n * sizeof(int));
  a->num_ucgstates = (int *) xrealloc(a->num_ucgstates, n * sizeof(int));
  a->ucgl = (double *) xrealloc(a->ucgl, n * sizeof(double));
  a->ucgvl = (double *) xrealloc(a->ucgvl, n * sizeof(double));
  a->ucgml = (double *) xrealloc(a->ucgml, n * sizeof(double));
  a->ucgp = (double *) xrealloc(a->ucgp, n * sizeof(double));
  a->ucgforce = (double *) xrealloc(a->ucgforce, n * sizeof(double));
  a->scores = (double *) xrealloc(a->scores, 2 * n * sizeof(double));
  s->xhold = (double *) xrealloc(s->xhold, 3 * n * sizeof(double));
  s->ghost_src = (int *) xrealloc(s->ghost_src, n * sizeof(int));
  s->ghost_rank = (int *) xrealloc(s->ghost_rank, n * sizeof(int));
  s->ghost_shift = (int *) xrealloc(s->ghost_shift, 3 * n * sizeof(int));
  s->bin_of = (int *) xrealloc(s->bin_of, n * sizeof(int));
  s->molecule = (int *) xrealloc(s->molecule, n * sizeof(int));
  for (int i = s->nmax; i < nmax; i++) {
    a->num_ucgstates[i] = 0;
    a->ucgforce[i] = 0.0;
    a->scores[2 * i] = a->scores[2 * i + 1] = 0.0;
    a->f[3 * i] = a->f[3 * i + 1] = a->f[3 * i + 2] = 0.0;
  }
  s->nmax = nmax;
}

void orc_sim_destroy(orc_sim *s)
{
  if (!s) return;
  orc_atoms *a = &s->a;
  free(a->x); free(a->v); free(a->f); free(a->type); free(a->tag); free(a->mask);
  free(a->ucgstate); free(a->num_ucgstates); free(a->ucgl); free(a->ucgvl); free(a->ucgml);
  free(a->ucgp); free(a->ucgforce); free(a->scores); free(a->mass);
  free(s->xhold); free(s->ghost_src); free(s->ghost_rank); free(s->ghost_shift); free(s->bin_of); free(s->molecule);
  orc_cs_destroy(s->cs);
  free(s->binstart_owned); free(s->binstart_ghost);
  free(s->full.ilist); free(s->full.numneigh); free(s->full.first); free(s->full.neigh);
  free(s->half.ilist); free(s->half.numneigh); free(s->half.first); free(s->half.neigh);
  free(s);
}

orc_atoms *orc_sim_atoms(orc_sim *s) { return &s->a; }
orc_list *orc_sim_full_list(orc_sim *s) { return &s->full; }
orc_list *orc_sim_half_list(orc_sim *s) { return &s->half; }

/* ---------------------------------------------------------------- rebuild */

static void pbc_wrap(orc_sim *s)
{
  /* upstream Domain::pbc(), orthogonal periodic box */
  double *x = s->a.x;
  for (int i = 0; i < s->a.nlocal; i++) {
    for (int d = 0; d < 3; d++) {
      if (x[3 * i + d] < s->boxlo[d]) x[3 * i + d] += s->prd[d];
      if (x[3 * i + d] >= s->boxhi[d]) {
        x[3 * i + d] -= s->prd[d];
        x[3 * i + d] = (x[3 * i + d] > s->boxlo[d]) ? x[3 * i + d] : s->boxlo[d];
      }
    }
  }
}

void orc_sim_setup_bins(orc_sim *s)
{
  const double target = 0.5 * s->cutneigh;
  s->nbins = 1;
  for (int d = 0; d < 3; d++) {
    s->bboxlo[d] = s->sublo[d] - s->cutneigh;
    const double ext = (s->subhi[d] + s->cutneigh) - s->bboxlo[d];
    int nb = (int) (ext / target);
    if (nb < 1) nb = 1;
    s->nbin[d] = nb;
    s->binsize[d] = ext / nb;
    s->bininv[d] = 1.0 / s->binsize[d];
    int sx = (int) (s->cutneigh * s->bininv[d]);
    if (sx * s->binsize[d] < s->cutneigh) sx++;
    s->sten[d] = sx;
    s->nbins *= nb;
  }
}

static int coord2bin(const orc_sim *s, const double *x)
{
  int ib[3];
  for (int d = 0; d < 3; d++) {
    int b = (int) ((x[d] - s->bboxlo[d]) * s->bininv[d]);
    if (b < 0) b = 0;
    if (b > s->nbin[d] - 1) b = s->nbin[d] - 1;
    ib[d] = b;
  }
  return (ib[2] * s->nbin[1] + ib[1]) * s->nbin[0] + ib[0];
}

/* Morton (Z-order) code of a bin: bits of (bx, by, bz) interleaved, 9 bits per dimension.
   Sorting beads along this curve makes every run of ~1000 consecutive beads a compact blob,
   so most neighbours of a workgroup's beads are the workgroup's own beads. */
static long long morton_of_bin(const orc_sim *s, int b)
{
  const unsigned bx = (unsigned) (b % s->nbin[0]), by = (unsigned) ((b / s->nbin[0]) % s->nbin[1]);
  const unsigned bz = (unsigned) (b / (s->nbin[0] * s->nbin[1]));
  long long m = 0;
  for (int i = 0; i < 9; i++) {
    m |= (long long) ((bx >> i) & 1u) << (3 * i);
    m |= (long long) ((by >> i) & 1u) << (3 * i + 1);
    m |= (long long) ((bz >> i) & 1u) << (3 * i + 2);
  }
  return m;
}

typedef struct { long long key; int idx; int code; int bin; int rank; } sortrec;

static int cmp_sortrec(const void *pa, const void *pb)
{
  const sortrec *a = (const sortrec *) pa, *b = (const sortrec *) pb;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  if (a->code != b->code) return a->code < b->code ? -1 : 1;
  return 0;
}

static void permute_d(double *arr, const sortrec *r, int n, int w, double *tmp)
{
  for (int i = 0; i < n; i++)
    for (int c = 0; c < w; c++) tmp[(size_t) w * i + c] = arr[(size_t) w * r[i].idx + c];
  memcpy(arr, tmp, sizeof(double) * (size_t) w * n);
}

static void permute_i(int *arr, const sortrec *r, int n, int *tmp)
{
  for (int i = 0; i < n; i++) tmp[i] = arr[r[i].idx];
  memcpy(arr, tmp, sizeof(int) * (size_t) n);
}

static void sort_owned(orc_sim *s)
{
  orc_atoms *a = &s->a;
  const int n = a->nlocal;
  sortrec *r = (sortrec *) malloc(sizeof(sortrec) * (size_t) n);
  for (int i = 0; i < n; i++) {
    r[i].bin = coord2bin(s, &a->x[3 * i]);
    r[i].key = (morton_of_bin(s, r[i].bin) << 32) | (unsigned int) a->tag[i];
    r[i].idx = i;
    r[i].code = 0;
  }
  qsort(r, (size_t) n, sizeof(sortrec), cmp_sortrec);
  double *td = (double *) malloc(sizeof(double) * 3 * (size_t) n);
  int *ti = (int *) malloc(sizeof(int) * (size_t) n);
  permute_d(a->x, r, n, 3, td);
  permute_d(a->v, r, n, 3, td);
  permute_d(a->ucgl, r, n, 1, td);
  permute_d(a->ucgvl, r, n, 1, td);
  permute_d(a->ucgml, r, n, 1, td);
  permute_d(a->ucgp, r, n, 1, td);
  permute_i(a->type, r, n, ti);
  permute_i(a->tag, r, n, ti);
  permute_i(a->mask, r, n, ti);
  permute_i(a->ucgstate, r, n, ti);
  permute_i(a->num_ucgstates, r, n, ti);
  permute_i(s->molecule, r, n, ti);
  for (int i = 0; i < n; i++) s->bin_of[i] = r[i].bin;
  free(td);
  free(ti);
  free(r);
}

static void build_ghosts(orc_sim *s)
{
  orc_atoms *a = &s->a;
  const int n = a->nlocal;
  double lo[3], hi[3];
  for (int d = 0; d < 3; d++) {
    lo[d] = s->sublo[d] - s->cutneigh;
    hi[d] = s->subhi[d] + s->cutneigh;
  }
  int cap = 1024, ng = 0;
  sortrec *r = (sortrec *) malloc(sizeof(sortrec) * (size_t) cap);
  for (int sz = -1; sz <= 1; sz++)
    for (int sy = -1; sy <= 1; sy++)
      for (int sx = -1; sx <= 1; sx++) {
        if (sx == 0 && sy == 0 && sz == 0) continue;
        const int code = (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1);
        for (int i = 0; i < n; i++) {
          double xs[3];
          xs[0] = a->x[3 * i + 0] + sx * s->prd[0];
          xs[1] = a->x[3 * i + 1] + sy * s->prd[1];
          xs[2] = a->x[3 * i + 2] + sz * s->prd[2];
          if (xs[0] < lo[0] || xs[0] >= hi[0] || xs[1] < lo[1] || xs[1] >= hi[1] ||
              xs[2] < lo[2] || xs[2] >= hi[2])
            continue;
          if (ng == cap) {
            cap *= 2;
            r = (sortrec *) xrealloc(r, sizeof(sortrec) * (size_t) cap);
          }
          r[ng].bin = coord2bin(s, xs);
          r[ng].key = (morton_of_bin(s, r[ng].bin) << 32) | (unsigned int) a->tag[i];
          r[ng].idx = i;
          r[ng].code = code;
          ng++;
        }
      }
  qsort(r, (size_t) ng, sizeof(sortrec), cmp_sortrec);
  orc_sim_grow(s, n + ng);
  a = &s->a;
  a->nghost = ng;
  for (int g = 0; g < ng; g++) {
    const int src = r[g].idx, code = r[g].code;
    const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
    s->ghost_src[g] = src;
    s->ghost_shift[3 * g + 0] = sx;
    s->ghost_shift[3 * g + 1] = sy;
    s->ghost_shift[3 * g + 2] = sz;
    s->bin_of[n + g] = r[g].bin;
    a->tag[n + g] = a->tag[src];
    a->type[n + g] = a->type[src];
    a->mask[n + g] = a->mask[src];
    s->molecule[n + g] = s->molecule[src];
  }
  free(r);
  orc_sim_forward_comm(s);
}

/* forward communication: owner -> its periodic images (fields_comm of
 * UCG/atom_vec_ucg.cpp:71: x + shift, ucgstate, ucgl, ucgp) */
void orc_sim_forward_comm(orc_sim *s)
{
  orc_atoms *a = &s->a;
  const int n = a->nlocal;
  for (int g = 0; g < a->nghost; g++) {
    const int src = s->ghost_src[g];
    for (int d = 0; d < 3; d++)
      a->x[3 * (n + g) + d] = a->x[3 * src + d] + s->ghost_shift[3 * g + d] * s->prd[d];
    a->ucgstate[n + g] = a->ucgstate[src];
    a->ucgl[n + g] = a->ucgl[src];
    a->ucgp[n + g] = a->ucgp[src];
    if (s->cs) a->type[n + g] = a->type[src]; /* fix cluster_switch forwards atom->type (UCG/fix_cluster_switch.cpp:498-527) */
  }
}

/* reverse communication (reference-order mode only): ghost sums -> owners
 * (fields_reverse of UCG/atom_vec_ucg.cpp:73: f, ucgforce, ucgsoftmaxscores) */
void orc_sim_reverse_comm(orc_sim *s)
{
  orc_atoms *a = &s->a;
  const int n = a->nlocal;
  for (int g = 0; g < a->nghost; g++) {
    const int src = s->ghost_src[g];
    for (int d = 0; d < 3; d++) a->f[3 * src + d] += a->f[3 * (n + g) + d];
    a->ucgforce[src] += a->ucgforce[n + g];
    a->scores[2 * src + 0] += a->scores[2 * (n + g) + 0];
    a->scores[2 * src + 1] += a->scores[2 * (n + g) + 1];
  }
}

static void build_bins(orc_sim *s)
{
  /* beads of one bin are contiguous in both classes (sorted by the bin's Morton code, then tag);
     record [start, end) per bin by boundary detection */
  const orc_atoms *a = &s->a;
  const int n = a->nlocal, ng = a->nghost;
  s->binstart_owned = (int *) xrealloc(s->binstart_owned, sizeof(int) * 2 * ((size_t) s->nbins + 1));
  s->binstart_ghost = (int *) xrealloc(s->binstart_ghost, sizeof(int) * 2 * ((size_t) s->nbins + 1));
  memset(s->binstart_owned, 0, sizeof(int) * 2 * ((size_t) s->nbins + 1));
  memset(s->binstart_ghost, 0, sizeof(int) * 2 * ((size_t) s->nbins + 1));
  for (int i = 0; i < n; i++) {
    const int b = s->bin_of[i];
    if (i == 0 || s->bin_of[i - 1] != b) s->binstart_owned[2 * b] = i;
    if (i == n - 1 || s->bin_of[i + 1] != b) s->binstart_owned[2 * b + 1] = i + 1;
  }
  for (int g = 0; g < ng; g++) {
    const int b = s->bin_of[n + g];
    if (g == 0 || s->bin_of[n + g - 1] != b) s->binstart_ghost[2 * b] = n + g;
    if (g == ng - 1 || s->bin_of[n + g + 1] != b) s->binstart_ghost[2 * b + 1] = n + g + 1;
  }
}

static void list_reserve(orc_list *l, int inum, long long nent)
{
  l->ilist = (int *) xrealloc(l->ilist, sizeof(int) * (size_t) (inum + 1));
  l->numneigh = (int *) xrealloc(l->numneigh, sizeof(int) * (size_t) (inum + 1));
  l->first = (long long *) xrealloc(l->first, sizeof(long long) * (size_t) (inum + 1));
  l->neigh = (int *) xrealloc(l->neigh, sizeof(int) * (size_t) (nent + 1));
}

static void build_lists(orc_sim *s)
{
  const orc_atoms *a = &s->a;
  const int n = a->nlocal;
  const double cutsq = s->cutneigh * s->cutneigh;
  const double *x = a->x;
  /* Rows are partitioned by the distance at BUILD time into four classes, each kept in
     stencil-traversal order: class 0 = inside the force cutoff, classes 1..3 = thirds of the
     skin shell.  Beads that start inside the cutoff therefore sit at the front of every row
     and the tail of a row is (at first) all skin: a wavefront whose 64 rows have all run
     into their skin part skips the pair arithmetic.  The order is part of the canonical
     summation order, so the CPU and the GPU builders must agree on it. */
  double cls_sq[3];
  for (int c = 0; c < 3; c++) {
    const double rc = s->cutforce + s->skin * c / 3.0;
    cls_sq[c] = rc * rc;
  }
  /* pass 0 counts per class, pass 1 fills */
  long long total = 0;
  int *ccount = (int *) malloc(sizeof(int) * 4 * (size_t) (n > 0 ? n : 1));
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) list_reserve(&s->full, n, total);
    long long pos = 0;
    for (int k = 0; k < n; k++) {
      const int b = s->bin_of[k];
      const int bx = b % s->nbin[0], by = (b / s->nbin[0]) % s->nbin[1], bz = b / (s->nbin[0] * s->nbin[1]);
      const long long start = pos;
      int cnt[4] = {0, 0, 0, 0};
      long long cpos[4] = {0, 0, 0, 0};
      if (pass == 1) {
        cpos[0] = start;
        for (int c = 1; c < 4; c++) cpos[c] = cpos[c - 1] + ccount[4 * k + c - 1];
      }
      for (int dz = -s->sten[2]; dz <= s->sten[2]; dz++) {
        const int cz = bz + dz;
        if (cz < 0 || cz >= s->nbin[2]) continue;
        for (int dy = -s->sten[1]; dy <= s->sten[1]; dy++) {
          const int cy = by + dy;
          if (cy < 0 || cy >= s->nbin[1]) continue;
          for (int dx = -s->sten[0]; dx <= s->sten[0]; dx++) {
            const int cx = bx + dx;
            if (cx < 0 || cx >= s->nbin[0]) continue;
            const int c = (cz * s->nbin[1] + cy) * s->nbin[0] + cx;
            for (int cls = 0; cls < 2; cls++) {
              const int *bs = cls ? s->binstart_ghost : s->binstart_owned;
              for (int m = bs[2 * c]; m < bs[2 * c + 1]; m++) {
                if (m == k) continue;
                const double delx = x[3 * k + 0] - x[3 * m + 0];
                const double dely = x[3 * k + 1] - x[3 * m + 1];
                const double delz = x[3 * k + 2] - x[3 * m + 2];
                const double rsq = delx * delx + dely * dely + delz * delz;
                if (rsq < cutsq) {
                  const int rc = (rsq < cls_sq[0]) ? 0 : (rsq < cls_sq[1]) ? 1 : (rsq < cls_sq[2]) ? 2 : 3;
                  if (pass == 1) {
                    const int orient = (a->tag[k] <= a->tag[m]) ? 1 : 0;
                    s->full.neigh[cpos[rc]++] = m | (orient << ORC_ORIENT_BIT);
                  }
                  cnt[rc]++;
                  pos++;
                }
              }
            }
          }
        }
      }
      if (pass == 0) {
        for (int c = 0; c < 4; c++) ccount[4 * k + c] = cnt[c];
      } else {
        s->full.ilist[k] = k;
        s->full.first[k] = start;
        s->full.numneigh[k] = (int) (pos - start);
      }
    }
    total = pos;
  }
  free(ccount);
  s->full.inum = n;

  /* half list for the reference-order mode: each pair once.  owned-owned: the
     lower tag is "i".  owned-ghost: lower tag owns the pair; equal tags (self
     image) by the z/y/x coordinate rule of LAMMPS' newton half lists. */
  list_reserve(&s->half, n, total);
  long long pos = 0;
  for (int k = 0; k < n; k++) {
    const long long start = pos;
    const int *row = s->full.neigh + s->full.first[k];
    for (int e = 0; e < s->full.numneigh[k]; e++) {
      const int m = row[e] & ORC_NEIGHMASK;
      int keep;
      if (a->tag[k] != a->tag[m]) keep = a->tag[k] < a->tag[m];
      else {
        if (x[3 * m + 2] != x[3 * k + 2]) keep = x[3 * m + 2] > x[3 * k + 2];
        else if (x[3 * m + 1] != x[3 * k + 1]) keep = x[3 * m + 1] > x[3 * k + 1];
        else keep = x[3 * m + 0] > x[3 * k + 0];
      }
      if (keep) s->half.neigh[pos++] = m;
    }
    s->half.ilist[k] = k;
    s->half.first[k] = start;
    s->half.numneigh[k] = (int) (pos - start);
  }
  s->half.inum = n;
}

void orc_sim_rebuild(orc_sim *s)
{
  pbc_wrap(s);
  orc_sim_setup_bins(s);
  sort_owned(s);
  build_ghosts(s);
  build_bins(s);
  build_lists(s);
  memcpy(s->xhold, s->a.x, sizeof(double) * 3 * (size_t) s->a.nlocal);
  s->ago = 0;
  s->nrebuild++;
}

static int check_distance(const orc_sim *s)
{
  const double deltasq = 0.25 * s->skin * s->skin;
  const double *x = s->a.x, *xh = s->xhold;
  for (int i = 0; i < s->a.nlocal; i++) {
    const double delx = x[3 * i + 0] - xh[3 * i + 0];
    const double dely = x[3 * i + 1] - xh[3 * i + 1];
    const double delz = x[3 * i + 2] - xh[3 * i + 2];
    const double rsq = delx * delx + dely * dely + delz * delz;
    if (rsq > deltasq) return 1;
  }
  return 0;
}

static int decide(orc_sim *s)
{
  /* upstream Neighbor::decide(): a fix with force_reneighbor whose next_reneighbor is this step
     forces the build before `ago` is even incremented */
  if (s->cs && s->cs->next_reneighbor == s->ntimestep) return 1;
  s->ago++;
  if (s->ago >= s->delay && s->ago % s->every == 0) {
    if (s->check == 0) return 1;
    return check_distance(s);
  }
  return 0;
}

/* ------------------------------------------------------------ force + fixes */

static int compute_forces(orc_sim *s, int eflag, int vflag)
{
  int rc;
  if (s->pair->style == ORC_STYLE_BETHE_DENSITY) {
    orc_force_clear(&s->a, 0);
    rc = orc_pair_density_compute(s->pair, &s->a, &s->full, s->mode, eflag, vflag, s->ghost_src, &s->ev);
  } else if (s->mode == 0) {
    orc_force_clear(&s->a, 1);
    rc = orc_pair_compute_half(s->pair, &s->a, &s->half, 1, eflag, vflag, &s->ev);
    orc_sim_reverse_comm(s);
  } else {
    orc_force_clear(&s->a, 0);
    rc = orc_pair_compute_gather(s->pair, &s->a, &s->full, eflag, vflag, &s->ev);
  }
  if (rc) s->pair_errors++;
  return rc;
}

static void post_force(orc_sim *s)
{
  /* fix order = definition order: thermostat first, then fix ucgstate
     (UCG/fix_ucgstate.cpp:143-154 requires that order) */
  if (s->lang) orc_fix_langevin_post_force(s->lang, &s->a, s->groupbit, s->ntimestep, s->beginstep, s->endstep);
  if (s->have_ucgstate) orc_fix_ucgstate_post_force(&s->ucgst, &s->a);
}

/* the integrator fix is defined first in a deck, so its post_force (the wall/hard bias) runs
   before the thermostat's; it has no setup() hook, so not at setup */
static void integrator_post_force(orc_sim *s)
{
  if (s->have_nve == 3) orc_fix_nve_wall_post_force(&s->a, s->wall_barrier, s->groupbit);
}

int orc_sim_setup(orc_sim *s, long long nsteps_planned)
{
  /* upstream Verlet::setup(): build lists, compute forces, then Modify::setup ->
     each fix's setup(); Fix_UCGLD_Langevin::setup and FixUCGState::setup both call
     post_force (UCG/fix_ucgld_langevin.cpp:187-197, UCG/fix_ucgstate.cpp:142-171) */
  s->beginstep = s->ntimestep;
  s->endstep = s->ntimestep + nsteps_planned;
  if (s->lang) orc_fix_langevin_init(s->lang, &s->a, s->dt, s->boltz, s->ftm2v, s->mvv2e);
  orc_sim_rebuild(s);
  int rc = compute_forces(s, 1, 1);
  post_force(s);
  return rc;
}

int orc_sim_run(orc_sim *s, long long nsteps, int thermo_every)
{
  int rc_any = 0;
  for (long long n = 0; n < nsteps; n++) {
    s->ntimestep++;
    const int ev = (thermo_every > 0 && (s->ntimestep % thermo_every == 0)) ? 1 : 0;
    if (s->have_nve == 1) orc_fix_nve_initial(&s->a, s->dt, s->ftm2v, s->groupbit);
    else if (s->have_nve >= 2) orc_fix_nve_wall_initial(&s->a, s->dt, s->ftm2v, s->groupbit);
    if (decide(s)) {
      /* FixClusterSwitch::pre_exchange (UCG/fix_cluster_switch.cpp:452-469) runs before Verlet's own
         exchange / borders / build; both rebuilds see the same positions, so one is done here */
      orc_sim_rebuild(s);
      if (s->cs && s->cs->next_reneighbor == s->ntimestep) {
        if (s->cs->switchFreq != 0) {
          orc_cs_check_cluster(s->cs, &s->a, s->molecule, &s->full);
          if (orc_cs_attempt_switch(s->cs, &s->a, s->molecule)) return 1;
          orc_sim_forward_comm(s); /* comm->forward_comm(this): types of the ghosts */
          s->cs->next_reneighbor = s->ntimestep + s->cs->switchFreq;
        }
      }
    } else
      orc_sim_forward_comm(s);
    int rc = compute_forces(s, ev, ev);
    if (rc) rc_any = rc;
    integrator_post_force(s);
    post_force(s);
    if (s->have_nve == 1) orc_fix_nve_final(&s->a, s->dt, s->ftm2v, s->groupbit);
    else if (s->have_nve >= 2) orc_fix_nve_wall_final(&s->a, s->dt, s->ftm2v, s->groupbit);
    if (s->lang) orc_fix_langevin_end_of_step(s->lang, &s->a, s->groupbit, s->boltz, s->mvv2e);
  }
  return rc_any;
}

/* ------------------------------------------------- accessors for the tests */

void orc_sim_set_run_params(orc_sim *s, double dt, int every, int delay, int check, int mode)
{
  s->dt = dt;
  s->every = every;
  s->delay = delay;
  s->check = check;
  s->mode = mode;
}

void orc_sim_set_units(orc_sim *s, double boltz, double ftm2v, double mvv2e)
{
  s->boltz = boltz;
  s->ftm2v = ftm2v;
  s->mvv2e = mvv2e;
}

void orc_sim_attach(orc_sim *s, orc_pair *pair, orc_fix_langevin *lang, int have_nve,
                    int have_ucgstate, int ld_flag, int mc_flag, int mc_seed, double mc_rate)
{
  s->pair = pair;
  s->lang = lang;
  s->have_nve = have_nve;
  s->have_ucgstate = have_ucgstate;
  if (have_ucgstate) orc_fix_ucgstate_init(&s->ucgst, ld_flag, mc_flag, mc_seed, mc_rate, 0);
}

void orc_sim_set_wall_barrier(orc_sim *s, double barrier) { s->wall_barrier = barrier; }

int *orc_sim_molecule(orc_sim *s) { return s->molecule; }
orc_cluster_switch *orc_sim_cs(orc_sim *s) { return s->cs; }

const char *orc_sim_cluster_switch(orc_sim *s, int mol_seed, int mol_offset, double cutoff, int seed, int switchFreq,
                                   const char *rateFile, const char *contactFile)
{
  orc_cs_destroy(s->cs);
  s->cs = orc_cs_create(&s->a, s->molecule, s->ntypes, s->groupbit, mol_seed, mol_offset, cutoff, seed, switchFreq,
                        rateFile, contactFile, s->ntimestep);
  return orc_cs_error(s->cs);
}

void orc_sim_get_info(const orc_sim *s, long long *out)
{
  out[0] = s->a.nlocal;
  out[1] = s->a.nghost;
  out[2] = s->nrebuild;
  out[3] = s->pair_errors;
  out[4] = s->ntimestep;
  out[5] = s->nbin[0];
  out[6] = s->nbin[1];
  out[7] = s->nbin[2];
  out[8] = s->sten[0];
  out[9] = s->sten[1];
  out[10] = s->sten[2];
  out[11] = s->full.inum ? (s->full.first[s->full.inum - 1] + s->full.numneigh[s->full.inum - 1]) : 0;
  out[12] = s->half.inum ? (s->half.first[s->half.inum - 1] + s->half.numneigh[s->half.inum - 1]) : 0;
}

void orc_sim_get_ev(const orc_sim *s, double *out)
{
  out[0] = s->ev.eng_vdwl;
  for (int c = 0; c < 6; c++) out[1 + c] = s->ev.virial[c];
  out[7] = s->lang ? s->lang->lambda_temp : 0.0;
}

const int *orc_sim_ghost_src(const orc_sim *s) { return s->ghost_src; }
const int *orc_sim_ghost_shift(const orc_sim *s) { return s->ghost_shift; }
const int *orc_sim_bin_of(const orc_sim *s) { return s->bin_of; }
double *orc_sim_mass(orc_sim *s) { return s->a.mass; }
int orc_sim_compute_forces(orc_sim *s, int eflag, int vflag) { return compute_forces(s, eflag, vflag); }


/* ------------------------------------------------------------------ decomposed runs (orc_md.h: orc_world) */

struct orc_world {
  int nranks, grid[3];
  orc_sim **r;
  orc_fix_langevin **lang; /* per rank, or NULL */
  orc_atoms input;         /* the beads handed in (rank 0's arrays until the first re-neighbouring) */
  double ev[7];
};

static double proc_bound(const orc_sim *s, const int *grid, int d, int i)
{
  /* LAMMPS' uniform bricks: boxlo + prd * i / procgrid, the last one ends at boxhi (same expression as the library) */
  return (i >= grid[d]) ? s->boxhi[d] : s->boxlo[d] + s->prd[d] * i / grid[d];
}

static int owner_of(const orc_sim *s, const int *grid, const double *x)
{
  int loc[3];
  for (int d = 0; d < 3; d++) {
    loc[d] = 0;
    for (int i = 1; i < grid[d]; i++)
      if (x[d] >= proc_bound(s, grid, d, i)) loc[d] = i;
  }
  return loc[0] + grid[0] * (loc[1] + grid[1] * loc[2]);
}

orc_world *orc_world_create(const int *grid3, int natoms, const double *boxlo, const double *boxhi, double cutforce,
                            double skin, int ntypes)
{
  orc_world *w = (orc_world *) calloc(1, sizeof(orc_world));
  w->nranks = grid3[0] * grid3[1] * grid3[2];
  for (int d = 0; d < 3; d++) w->grid[d] = grid3[d];
  w->r = (orc_sim **) calloc((size_t) w->nranks, sizeof(orc_sim *));
  w->lang = (orc_fix_langevin **) calloc((size_t) w->nranks, sizeof(orc_fix_langevin *));
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = orc_sim_create(me == 0 ? natoms : 0, boxlo, boxhi, cutforce, skin, ntypes);
    const int loc[3] = {me % grid3[0], (me / grid3[0]) % grid3[1], me / (grid3[0] * grid3[1])};
    for (int d = 0; d < 3; d++) {
      s->sublo[d] = proc_bound(s, grid3, d, loc[d]);
      s->subhi[d] = proc_bound(s, grid3, d, loc[d] + 1);
    }
    w->r[me] = s;
  }
  return w;
}

void orc_world_destroy(orc_world *w)
{
  if (!w) return;
  for (int me = 0; me < w->nranks; me++) {
    orc_sim_destroy(w->r[me]);
    if (w->lang[me]) orc_fix_langevin_destroy(w->lang[me]);
  }
  free(w->r);
  free(w->lang);
  free(w);
}

int orc_world_nranks(const orc_world *w) { return w->nranks; }
orc_sim *orc_world_rank(orc_world *w, int r) { return w->r[r]; }
orc_atoms *orc_world_input(orc_world *w) { return &w->r[0]->a; }
int *orc_world_input_molecule(orc_world *w) { return w->r[0]->molecule; }

void orc_world_set_run_params(orc_world *w, double dt, int every, int delay, int check)
{
  for (int me = 0; me < w->nranks; me++) orc_sim_set_run_params(w->r[me], dt, every, delay, check, 1);
}

void orc_world_attach(orc_world *w, orc_pair *pair, int have_langevin, double t_start, double t_stop, double t_period,
                      int lang_seed, int have_nve, double wall_barrier, int have_ucgstate, int ld_flag, int mc_flag,
                      int mc_seed, double mc_rate)
{
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    if (w->lang[me]) orc_fix_langevin_destroy(w->lang[me]);
    /* RanMars(seed + comm->me): UCG/fix_ucgld_langevin.cpp:85, UCG/fix_ucgstate.cpp:62 */
    w->lang[me] = have_langevin ? orc_fix_langevin_create(s->ntypes, t_start, t_stop, t_period, lang_seed, me) : NULL;
    s->pair = pair;
    s->lang = w->lang[me];
    s->have_nve = have_nve;
    s->wall_barrier = wall_barrier;
    s->have_ucgstate = have_ucgstate;
    if (have_ucgstate) orc_fix_ucgstate_init(&s->ucgst, ld_flag, mc_flag, mc_seed, mc_rate, me);
    for (int t = 0; t <= s->ntypes; t++) s->a.mass[t] = w->r[0]->a.mass[t];
  }
}

/* forward halo: every ghost takes x + shift * prd, state, lambda, ucgp from its owner (fields_comm, UCG/atom_vec_ucg.cpp:71) */
static void world_forward(orc_world *w)
{
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    orc_atoms *a = &s->a;
    const int n = a->nlocal;
    for (int g = 0; g < a->nghost; g++) {
      const orc_atoms *o = &w->r[s->ghost_rank[g]]->a;
      const int src = s->ghost_src[g];
      for (int d = 0; d < 3; d++) a->x[3 * (n + g) + d] = o->x[3 * src + d] + s->ghost_shift[3 * g + d] * s->prd[d];
      a->ucgstate[n + g] = o->ucgstate[src];
      a->ucgl[n + g] = o->ucgl[src];
      a->ucgp[n + g] = o->ucgp[src];
      a->type[n + g] = o->type[src]; /* (fix cluster_switch's forward_comm: the ghosts' new types) */
    }
  }
}

typedef struct {
  double x[3], v[3], ucgl, ucgvl, ucgml, ucgp;
  int type, tag, mask, state, nstates, mol;
} beadrec;

static void world_rebuild(orc_world *w)
{
  /* (1) wrap, (2) every bead to the rank whose brick holds it */
  int total = 0;
  for (int me = 0; me < w->nranks; me++) {
    pbc_wrap(w->r[me]);
    total += w->r[me]->a.nlocal;
  }
  beadrec *all = (beadrec *) malloc(sizeof(beadrec) * (size_t) (total ? total : 1));
  int *owner = (int *) malloc(sizeof(int) * (size_t) (total ? total : 1));
  int *count = (int *) calloc((size_t) (w->nranks > 0 ? w->nranks : 1), sizeof(int));
  int k = 0;
  for (int me = 0; me < w->nranks; me++) {
    const orc_sim *s = w->r[me];
    const orc_atoms *a = &s->a;
    for (int i = 0; i < a->nlocal; i++, k++) {
      beadrec *b = &all[k];
      for (int d = 0; d < 3; d++) {
        b->x[d] = a->x[3 * i + d];
        b->v[d] = a->v[3 * i + d];
      }
      b->ucgl = a->ucgl[i]; b->ucgvl = a->ucgvl[i]; b->ucgml = a->ucgml[i]; b->ucgp = a->ucgp[i];
      b->type = a->type[i]; b->tag = a->tag[i]; b->mask = a->mask[i]; b->state = a->ucgstate[i];
      b->nstates = a->num_ucgstates[i]; b->mol = s->molecule[i];
      owner[k] = owner_of(s, w->grid, b->x);
      count[owner[k]]++;
    }
  }
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    orc_sim_grow(s, count[me] + count[me] / 2 + 1024);
    s->a.nlocal = 0;
    s->a.nghost = 0;
  }
  for (k = 0; k < total; k++) {
    orc_sim *s = w->r[owner[k]];
    orc_atoms *a = &s->a;
    const int i = a->nlocal++;
    const beadrec *b = &all[k];
    for (int d = 0; d < 3; d++) {
      a->x[3 * i + d] = b->x[d];
      a->v[3 * i + d] = b->v[d];
    }
    a->ucgl[i] = b->ucgl; a->ucgvl[i] = b->ucgvl; a->ucgml[i] = b->ucgml; a->ucgp[i] = b->ucgp;
    a->type[i] = b->type; a->tag[i] = b->tag; a->mask[i] = b->mask; a->ucgstate[i] = b->state;
    a->num_ucgstates[i] = b->nstates; s->molecule[i] = b->mol;
  }
  free(all);
  free(owner);
  free(count);
  /* (3) each rank: bins of its brick, owned beads by (Morton bin, tag) */
  for (int me = 0; me < w->nranks; me++) {
    orc_sim_setup_bins(w->r[me]);
    sort_owned(w->r[me]);
  }
  /* (4) ghosts of rank me: every image (bead of rank q, shift) inside me's extended brick, its own unshifted beads
     excepted; sorted by (Morton bin, tag, shift code) */
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    double lo[3], hi[3];
    for (int d = 0; d < 3; d++) {
      lo[d] = s->sublo[d] - s->cutneigh;
      hi[d] = s->subhi[d] + s->cutneigh;
    }
    int cap = 4096, ng = 0;
    sortrec *rec = (sortrec *) malloc(sizeof(sortrec) * (size_t) cap);
    for (int q = 0; q < w->nranks; q++) {
      const orc_atoms *o = &w->r[q]->a;
      for (int code = 0; code < 27; code++) {
        if (q == me && code == 13) continue;
        const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
        for (int i = 0; i < o->nlocal; i++) {
          double xs[3];
          xs[0] = o->x[3 * i + 0] + sx * s->prd[0];
          xs[1] = o->x[3 * i + 1] + sy * s->prd[1];
          xs[2] = o->x[3 * i + 2] + sz * s->prd[2];
          if (xs[0] < lo[0] || xs[0] >= hi[0] || xs[1] < lo[1] || xs[1] >= hi[1] || xs[2] < lo[2] || xs[2] >= hi[2]) continue;
          if (ng == cap) {
            cap *= 2;
            rec = (sortrec *) xrealloc(rec, sizeof(sortrec) * (size_t) cap);
          }
          rec[ng].bin = coord2bin(s, xs);
          rec[ng].key = (morton_of_bin(s, rec[ng].bin) << 32) | (unsigned int) o->tag[i];
          rec[ng].idx = i;
          rec[ng].code = code;
          rec[ng].rank = q;
          ng++;
        }
      }
    }
    qsort(rec, (size_t) ng, sizeof(sortrec), cmp_sortrec);
    const int n = s->a.nlocal;
    orc_sim_grow(s, n + ng);
    orc_atoms *a = &s->a;
    a->nghost = ng;
    for (int g = 0; g < ng; g++) {
      const int code = rec[g].code;
      const orc_sim *os = w->r[rec[g].rank];
      s->ghost_rank[g] = rec[g].rank;
      s->ghost_src[g] = rec[g].idx;
      s->ghost_shift[3 * g + 0] = code % 3 - 1;
      s->ghost_shift[3 * g + 1] = (code / 3) % 3 - 1;
      s->ghost_shift[3 * g + 2] = code / 9 - 1;
      s->bin_of[n + g] = rec[g].bin;
      a->tag[n + g] = os->a.tag[rec[g].idx];
      a->type[n + g] = os->a.type[rec[g].idx];
      a->mask[n + g] = os->a.mask[rec[g].idx];
      s->molecule[n + g] = os->molecule[rec[g].idx];
    }
    free(rec);
  }
  world_forward(w);
  /* (5) rows */
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    build_bins(s);
    build_lists(s);
    memcpy(s->xhold, s->a.x, sizeof(double) * 3 * (size_t) s->a.nlocal);
    s->ago = 0;
    s->nrebuild++;
  }
}

/* ghosts take two doubles per bead from their owners (the density style's priors + their derivatives, then its CV forces) */
static void world_forward2(orc_world *w, double **field)
{
  for (int me = 0; me < w->nranks; me++) {
    const orc_sim *s = w->r[me];
    const int n = s->a.nlocal;
    for (int g = 0; g < s->a.nghost; g++) {
      const double *o = field[s->ghost_rank[g]];
      const int src = s->ghost_src[g];
      field[me][2 * (n + g)] = o[2 * src];
      field[me][2 * (n + g) + 1] = o[2 * src + 1];
    }
  }
}

static int world_forces(orc_world *w, int ev)
{
  int rc_any = 0;
  for (int c = 0; c < 7; c++) w->ev[c] = 0.0;
  if (w->r[0]->pair->style == ORC_STYLE_BETHE_DENSITY) {
    /* table_ucg_bethe_density: the three passes in lockstep; between them the ghosts' priors (and their derivatives) and
       CV forces come from the owner RANKS -- the forward_comm of UCG/pair_table_ucg_bethe_density.cpp:280 (App. B #7) */
    orc_density_work **wk = (orc_density_work **) calloc((size_t) w->nranks, sizeof(orc_density_work *));
    double **fld = (double **) calloc((size_t) w->nranks, sizeof(double *));
    for (int me = 0; me < w->nranks; me++) {
      orc_sim *s = w->r[me];
      if (orc_pair_density_check(s->pair, &s->a)) rc_any = 1;
      orc_force_clear(&s->a, 0);
      memset(&s->ev, 0, sizeof(s->ev));
      wk[me] = orc_density_work_create(s->a.nlocal + s->a.nghost);
    }
    if (!rc_any) {
      for (int me = 0; me < w->nranks; me++) orc_pair_density_pass1(w->r[me]->pair, &w->r[me]->a, &w->r[me]->full, wk[me]);
      for (int me = 0; me < w->nranks; me++) fld[me] = wk[me]->prior;
      world_forward2(w, fld);
      for (int me = 0; me < w->nranks; me++) fld[me] = wk[me]->partial;
      world_forward2(w, fld);
      for (int me = 0; me < w->nranks; me++)
        orc_pair_density_pass2(w->r[me]->pair, &w->r[me]->a, &w->r[me]->full, 1, ev, ev, wk[me], &w->r[me]->ev);
      for (int me = 0; me < w->nranks; me++) fld[me] = wk[me]->cv;
      world_forward2(w, fld);
      for (int me = 0; me < w->nranks; me++)
        orc_pair_density_pass3(w->r[me]->pair, &w->r[me]->a, &w->r[me]->full, 1, ev, NULL, wk[me], &w->r[me]->ev);
    }
    for (int me = 0; me < w->nranks; me++) {
      orc_sim *s = w->r[me];
      if (s->ev.err) {
        rc_any = 2;
        s->pair_errors++;
      }
      w->ev[0] += s->ev.eng_vdwl;
      for (int c = 0; c < 6; c++) w->ev[1 + c] += s->ev.virial[c];
      orc_density_work_destroy(wk[me]);
    }
    free(wk);
    free(fld);
    return rc_any;
  }
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    s->mode = 1;
    const int rc = compute_forces(s, ev, ev);
    if (rc) rc_any = rc;
    w->ev[0] += s->ev.eng_vdwl;
    for (int c = 0; c < 6; c++) w->ev[1 + c] += s->ev.virial[c];
  }
  return rc_any;
}

int orc_world_setup(orc_world *w, long long nsteps_planned)
{
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    s->beginstep = s->ntimestep;
    s->endstep = s->ntimestep + nsteps_planned;
  }
  world_rebuild(w);
  /* Fix_UCGLD_Langevin::init() reads atom->ucgml[type index] of the LOCAL bead order (App. B #5) */
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    if (s->lang) orc_fix_langevin_init(s->lang, &s->a, s->dt, s->boltz, s->ftm2v, s->mvv2e);
  }
  const int rc = world_forces(w, 1);
  for (int me = 0; me < w->nranks; me++) post_force(w->r[me]);
  return rc;
}

/* fix cluster_switch on a decomposed run.  The constructor's survey is reduced over the ranks (UCG/fix_cluster_switch.cpp:
   114-116 maxmol MAX and the two counts SUM, :158-159 mol_restrict / mol_state MAX), so it is made here from all beads at
   once -- the same numbers whatever the split, for molecules that are wholly ON or OFF -- and every rank gets a copy with
   its own RanPark streams, seeded alike (:56-57). */
const char *orc_world_cluster_switch(orc_world *w, int mol_seed, int mol_offset, double cutoff, int seed, int switchFreq,
                                     const char *rateFile, const char *contactFile)
{
  int total = 0;
  for (int me = 0; me < w->nranks; me++) total += w->r[me]->a.nlocal;
  orc_atoms u;
  memset(&u, 0, sizeof u);
  u.nlocal = total;
  u.tag = (int *) malloc(sizeof(int) * (size_t) (total + 1));
  u.type = (int *) malloc(sizeof(int) * (size_t) (total + 1));
  u.mask = (int *) malloc(sizeof(int) * (size_t) (total + 1));
  int *mol = (int *) malloc(sizeof(int) * (size_t) (total + 1));
  int k = 0;
  for (int me = 0; me < w->nranks; me++) {
    const orc_sim *s = w->r[me];
    for (int i = 0; i < s->a.nlocal; i++, k++) {
      u.tag[k] = s->a.tag[i];
      u.type[k] = s->a.type[i];
      u.mask[k] = s->a.mask[i];
      mol[k] = s->molecule[i];
    }
  }
  orc_sim *s0 = w->r[0];
  orc_cluster_switch *cs = orc_cs_create(&u, mol, s0->ntypes, s0->groupbit, mol_seed, mol_offset, cutoff, seed, switchFreq,
                                         rateFile, contactFile, s0->ntimestep);
  free(u.tag);
  free(u.type);
  free(u.mask);
  free(mol);
  for (int me = 0; me < w->nranks; me++) {
    orc_cs_destroy(w->r[me]->cs);
    w->r[me]->cs = NULL;
  }
  if (orc_cs_error(cs)) {
    w->r[0]->cs = cs; /* keeps the message alive */
    return orc_cs_error(cs);
  }
  w->r[0]->cs = cs;
  for (int me = 1; me < w->nranks; me++) w->r[me]->cs = orc_cs_clone(cs);
  return NULL;
}

/* check_cluster + attempt_switch on fresh lists (:452-469), the ranks in lockstep with the reference's reductions in between:
   labels MIN between the ranks' sweeps until no rank changes any (:664-683), mol_accept MAX (:793); then the ghosts take
   their owners' new types (comm->forward_comm(this)) */
static int world_cluster_step(orc_world *w)
{
  orc_cluster_switch *c0 = w->r[0]->cs;
  const int nm = c0->maxmol + 1;
  int *present = (int *) calloc((size_t) nm, sizeof(int));
  for (int me = 0; me < w->nranks; me++) orc_cs_presence(w->r[me]->cs, &w->r[me]->a, w->r[me]->molecule, present);
  int **lab = (int **) malloc(sizeof(int *) * (size_t) w->nranks);
  for (int me = 0; me < w->nranks; me++) {
    lab[me] = (int *) malloc(sizeof(int) * (size_t) nm);
    orc_cs_labels_init(w->r[me]->cs, present, lab[me]);
  }
  for (;;) {
    int any = 0;
    for (int me = 0; me < w->nranks; me++) {
      orc_sim *s = w->r[me];
      if (orc_cs_sweep_local(s->cs, &s->a, s->molecule, &s->full, lab[me])) any = 1;
    }
    for (int i = 0; i < nm; i++) {
      int m = lab[0][i];
      for (int me = 1; me < w->nranks; me++)
        if (lab[me][i] < m) m = lab[me][i];
      for (int me = 0; me < w->nranks; me++) lab[me][i] = m;
    }
    if (!any) break;
  }
  int rc = 0;
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    orc_cs_finalize(s->cs, lab[me]);
    orc_cs_attempt_local(s->cs, &s->a, s->molecule);
    free(lab[me]);
  }
  free(lab);
  free(present);
  for (int i = 0; i < nm; i++) {
    int m = -1;
    for (int me = 0; me < w->nranks; me++)
      if (w->r[me]->cs->mol_accept[i] > m) m = w->r[me]->cs->mol_accept[i];
    for (int me = 0; me < w->nranks; me++) w->r[me]->cs->mol_accept[i] = m;
  }
  for (int me = 0; me < w->nranks; me++) {
    orc_sim *s = w->r[me];
    if (orc_cs_attempt_apply(s->cs, &s->a)) rc = 1;
    s->cs->next_reneighbor = s->ntimestep + s->cs->switchFreq;
  }
  world_forward(w);
  return rc;
}

int orc_world_run(orc_world *w, long long nsteps, int thermo_every)
{
  int rc_any = 0;
  for (long long n = 0; n < nsteps; n++) {
    int flag = 0;
    int ev = 0;
    for (int me = 0; me < w->nranks; me++) {
      orc_sim *s = w->r[me];
      s->ntimestep++;
      ev = (thermo_every > 0 && (s->ntimestep % thermo_every == 0)) ? 1 : 0;
      if (s->have_nve == 1) orc_fix_nve_initial(&s->a, s->dt, s->ftm2v, s->groupbit);
      else if (s->have_nve >= 2) orc_fix_nve_wall_initial(&s->a, s->dt, s->ftm2v, s->groupbit);
      if (decide(s)) flag = 1; /* Neighbor::decide(): MPI_Allreduce of the ranks' flags */
    }
    if (flag) {
      world_rebuild(w);
      orc_sim *s0 = w->r[0];
      if (s0->cs && s0->cs->next_reneighbor == s0->ntimestep && s0->cs->switchFreq != 0 && world_cluster_step(w)) return 1;
    } else
      world_forward(w);
    const int rc = world_forces(w, ev);
    if (rc) rc_any = rc;
    for (int me = 0; me < w->nranks; me++) {
      orc_sim *s = w->r[me];
      integrator_post_force(s);
      post_force(s);
      if (s->have_nve == 1) orc_fix_nve_final(&s->a, s->dt, s->ftm2v, s->groupbit);
      else if (s->have_nve >= 2) orc_fix_nve_wall_final(&s->a, s->dt, s->ftm2v, s->groupbit);
      if (s->lang) orc_fix_langevin_end_of_step(s->lang, &s->a, s->groupbit, s->boltz, s->mvv2e);
    }
  }
  return rc_any;
}

void orc_world_get_ev(const orc_world *w, double *out7)
{
  for (int c = 0; c < 7; c++) out7[c] = w->ev[c];
}
