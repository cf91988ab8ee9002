/* TEST INFRASTRUCTURE ONLY (see mtp_oracle.h).  PARITY UNPINNED against a running
 * reference; pinned by tests/test_oracle.py's independent checks.
 *
 * CPU restatement of /root/reference/LAMMPS/ML-MTP/{pair_mtp,pair_mtp_extrapolation,
 * mtp_radial_basis,mtp_rb_chevbyshev_basis}.cpp written fresh in C.  The arithmetic keeps
 * the reference's operation order (sums run in the same index order, the same three
 * divides per basic moment) so that a future run of the real reference can be compared
 * to the last bits.
 */
#define _POSIX_C_SOURCE 200809L
#include "mtp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* Text reader: the subset of LAMMPS TextFileReader / ValueTokenizer behaviour the
 * reference relies on (pair_mtp.cpp:346-351): fgets into a resizable buffer, optional
 * stripping from '#', wordless lines skipped, tokens split on a separator set. */

typedef struct {
  FILE *fp;
  char *line;
  int bufsize;
  int ignore_comments;
} reader_t;

static const char *WS = " \t\r\n\f";

static int count_words(const char *s)
{
  int n = 0;
  while (*s) {
    s += strspn(s, WS);
    if (!*s) break;
    n++;
    s += strcspn(s, WS);
  }
  return n;
}

static char *reader_next_line(reader_t *r)
{
  for (;;) {
    char *p = fgets(r->line, r->bufsize, r->fp);
    if (!p) return NULL;
    if (r->ignore_comments && (p = strchr(r->line, '#'))) *p = '\0';
    if (count_words(r->line) > 0) return r->line;
  }
}

static void reader_set_bufsize(reader_t *r, int n)
{
  free(r->line);
  r->bufsize = n;
  r->line = (char *) malloc((size_t) n);
}

typedef struct {
  char *s;       /* owned copy */
  char *cur;
  const char *seps;
} tok_t;

static void tok_init(tok_t *t, const char *line, const char *seps)
{
  t->s = strdup(line);
  t->cur = t->s;
  t->seps = seps;
}
static void tok_free(tok_t *t)
{
  free(t->s);
  t->s = NULL;
}
/* returns NULL when no more tokens (the reference's TokenizerException) */
static char *tok_next(tok_t *t)
{
  t->cur += strspn(t->cur, t->seps);
  if (!*t->cur) return NULL;
  char *w = t->cur;
  t->cur += strcspn(t->cur, t->seps);
  if (*t->cur) *t->cur++ = '\0';
  return w;
}
static int tok_int(tok_t *t, int *v)
{
  char *w = tok_next(t), *e;
  if (!w) return -1;
  long l = strtol(w, &e, 10);
  if (*e) return -1;
  *v = (int) l;
  return 0;
}
static int tok_double(tok_t *t, double *v)
{
  char *w = tok_next(t), *e;
  if (!w) return -1;
  *v = strtod(w, &e);
  if (*e) return -1;
  return 0;
}

#define SEPS " \t\r\n\f=, "
#define SEPS_DASH " \t\r\n\f=, -"
#define SEPS_BRACE " \t\r\n\f=, {},"

#define FAIL(code, ...)                      \
  do {                                       \
    if (err) snprintf(err, errlen, __VA_ARGS__); \
    rc = (code);                             \
    goto done;                               \
  } while (0)

/* pair_mtp.cpp:335-570, mtp_radial_basis.cpp:59-102, pair_mtp_extrapolation.cpp:528-612 */
int mtp_oracle_read_file(const char *path, int want_selection, mtp_oracle_model *m, char *err,
                         int errlen)
{
  int rc = 0;
  memset(m, 0, sizeof(*m));
  FILE *fp = fopen(path, "rb");
  if (!fp) {
    if (err) snprintf(err, errlen, "cannot open potential file %s", path);
    return -2;
  }
  reader_t rd = {fp, NULL, 0, 1};
  reader_set_bufsize(&rd, 1024);
  tok_t tk = {0};
  char *line, *kw;

#define NEXT(seps)                                            \
  do {                                                        \
    tok_free(&tk);                                            \
    line = reader_next_line(&rd);                             \
    if (!line) FAIL(-3, "unexpected end of MTP file");        \
    tok_init(&tk, line, seps);                                \
    kw = tok_next(&tk);                                       \
    if (!kw) kw = (char *) "";                                \
  } while (0)

  NEXT(SEPS); /* :351-355 */
  if (strcmp(kw, "MTP")) FAIL(-4, "Only MTP potential files are accepted.");
  line = reader_next_line(&rd); /* :356-358 exact compare including the newline */
  if (!line || strcmp(line, "version = 1.1.0\n")) FAIL(-4, "MTP file must have version \"1.1.0\"");

  NEXT(SEPS); /* :361-372 optional potential_name */
  if (!strcmp(kw, "potential_name")) NEXT(SEPS);
  m->scaling = 1; /* :375-381 */
  if (!strcmp(kw, "scaling")) {
    if (tok_double(&tk, &m->scaling)) FAIL(-5, "bad scaling");
    NEXT(SEPS);
  }
  if (strcmp(kw, "species_count")) FAIL(-5, "Error reading MTP file. Species count not found.");
  if (tok_int(&tk, &m->species_count)) FAIL(-5, "bad species_count");

  NEXT(SEPS); /* :396-406 optional potential_tag */
  if (!strcmp(kw, "potential_tag")) NEXT(SEPS);
  if (strcmp(kw, "radial_basis_type")) /* :409-422 */
    FAIL(-5, "Error reading MTP file. No radial basis set type is specified.");
  {
    char *ty = tok_next(&tk);
    if (!ty || strcmp(ty, "RBChebyshev"))
      FAIL(-5, "Error reading MTP file. The specified radial basis set type, %s, was not found..",
           ty ? ty : "");
  }
  /* RadialMTPBasis::ReadBasisProperties, mtp_radial_basis.cpp:59-102 */
  NEXT(SEPS);
  if (!strcmp(kw, "scaling")) { /* read, then overwritten by pair_mtp.cpp:416 */
    double dummy;
    if (tok_double(&tk, &dummy)) FAIL(-5, "bad radial scaling");
    NEXT(SEPS);
  }
  if (strcmp(kw, "min_val") && strcmp(kw, "min_dist"))
    FAIL(-5, "Error in reading MTP file. Cannot read lower cutoff.");
  if (tok_double(&tk, &m->min_cutoff)) FAIL(-5, "bad min_dist");
  NEXT(SEPS);
  if (strcmp(kw, "max_val") && strcmp(kw, "max_dist"))
    FAIL(-5, "Error in reading MTP file. Cannot read upper cutoff.");
  if (tok_double(&tk, &m->max_cutoff)) FAIL(-5, "bad max_dist");
  NEXT(SEPS);
  if (strcmp(kw, "radial_basis_size"))
    FAIL(-5, "Error in reading MTP file. Cannot read radial basis set size.");
  if (tok_int(&tk, &m->radial_basis_size)) FAIL(-5, "bad radial_basis_size");

  NEXT(SEPS); /* :425-429 */
  if (strcmp(kw, "radial_funcs_count"))
    FAIL(-5, "Error in reading MTP file. Cannot read radial function count.");
  if (tok_int(&tk, &m->radial_func_count)) FAIL(-5, "bad radial_funcs_count");
  NEXT(SEPS); /* :432-439 */
  if (strcmp(kw, "radial_coeffs")) {
    if (!strcmp(kw, "magnetic_basis_type")) FAIL(-6, "Magnetic basis is currently not supported.");
    FAIL(-5, "Error in reading MTP file. Cannot read radial coeffs count.");
  }
  {
    int Sp = m->species_count, R = m->radial_basis_size, Mu = m->radial_func_count;
    int pairs = Sp * Sp, per_pair = R * Mu;
    m->radial_basis_coeffs = (double *) calloc((size_t) pairs * per_pair, sizeof(double));
    for (int i = 0; i < pairs; i++) { /* :450-469 */
      int t1, t2;
      tok_free(&tk);
      line = reader_next_line(&rd);
      if (!line) FAIL(-3, "unexpected end of MTP file in radial_coeffs");
      tok_init(&tk, line, SEPS_DASH);
      if (tok_int(&tk, &t1) || tok_int(&tk, &t2)) FAIL(-5, "bad species pair header");
      if (t1 < 0 || t2 < 0 || t1 >= Sp || t2 >= Sp) FAIL(-5, "species pair out of range");
      int off = (t1 * Sp + t2) * per_pair;
      for (int j = 0; j < Mu; j++) {
        tok_free(&tk);
        line = reader_next_line(&rd);
        if (!line) FAIL(-3, "unexpected end of MTP file in radial_coeffs");
        tok_init(&tk, line, SEPS_BRACE);
        for (int k = 0; k < R; k++)
          if (tok_double(&tk, &m->radial_basis_coeffs[off + j * R + k]))
            FAIL(-5, "bad radial coefficient");
      }
    }
  }
  NEXT(SEPS); /* :472-476 */
  if (strcmp(kw, "alpha_moments_count"))
    FAIL(-5, "Error reading MTP file. Alpha moment count not found.");
  if (tok_int(&tk, &m->alpha_moment_count)) FAIL(-5, "bad alpha_moments_count");
  NEXT(SEPS); /* :481-485 */
  if (strcmp(kw, "alpha_index_basic_count"))
    FAIL(-5, "Error reading MTP file. Alpha moment count not found.");
  if (tok_int(&tk, &m->alpha_index_basic_count)) FAIL(-5, "bad alpha_index_basic_count");

  reader_set_bufsize(&rd, m->alpha_index_basic_count * 20 + 20); /* :489-492 */
  NEXT(SEPS_BRACE);
  if (strcmp(kw, "alpha_index_basic")) FAIL(-5, "Error reading MTP file. Alpha index basic not found.");
  {
    int B = m->alpha_index_basic_count, mumax = 0, P = 0;
    m->alpha_index_basic = (int *) calloc((size_t) B * 4, sizeof(int));
    for (int i = 0; i < B; i++) {
      for (int j = 0; j < 4; j++)
        if (tok_int(&tk, &m->alpha_index_basic[4 * i + j])) FAIL(-5, "bad alpha_index_basic entry");
      if (m->alpha_index_basic[4 * i] > mumax) mumax = m->alpha_index_basic[4 * i];
      int s = m->alpha_index_basic[4 * i + 1] + m->alpha_index_basic[4 * i + 2] +
          m->alpha_index_basic[4 * i + 3];
      if (s > P) P = s;
    }
    if (mumax != m->radial_func_count - 1) /* :506-507 */
      FAIL(-7, "Wrong number of radial functions specified!");
    m->max_alpha_index_basic = P + 1; /* :510-515 */
  }
  NEXT(SEPS); /* :518-522 */
  if (strcmp(kw, "alpha_index_times_count"))
    FAIL(-5, "Error reading MTP file. Alpha index times count not found.");
  if (tok_int(&tk, &m->alpha_index_times_count)) FAIL(-5, "bad alpha_index_times_count");
  reader_set_bufsize(&rd, m->alpha_index_times_count * 32 + 20); /* :525-528 */
  NEXT(SEPS_BRACE);
  if (strcmp(kw, "alpha_index_times")) FAIL(-5, "Error reading MTP file. Alpha index times not found.");
  {
    int T = m->alpha_index_times_count;
    m->alpha_index_times = (int *) calloc((size_t) (T > 0 ? T : 1) * 4, sizeof(int));
    for (int i = 0; i < T * 4; i++)
      if (tok_int(&tk, &m->alpha_index_times[i])) FAIL(-5, "bad alpha_index_times entry");
  }
  NEXT(SEPS); /* :539-543 */
  if (strcmp(kw, "alpha_scalar_moments"))
    FAIL(-5, "Error reading MTP file. Alpha scalar moment count not found.");
  if (tok_int(&tk, &m->alpha_scalar_count)) FAIL(-5, "bad alpha_scalar_moments");
  NEXT(SEPS_BRACE); /* :546-553 */
  if (strcmp(kw, "alpha_moment_mapping"))
    FAIL(-5, "Error reading MTP file. Alpha moment mappings not found.");
  m->alpha_moment_mapping = (int *) calloc((size_t) m->alpha_scalar_count, sizeof(int));
  for (int i = 0; i < m->alpha_scalar_count; i++)
    if (tok_int(&tk, &m->alpha_moment_mapping[i])) FAIL(-5, "bad alpha_moment_mapping entry");
  NEXT(SEPS_BRACE); /* :556-561 */
  if (strcmp(kw, "species_coeffs")) FAIL(-5, "Error reading MTP file. Species coefficients not found.");
  m->species_coeffs = (double *) calloc((size_t) m->species_count, sizeof(double));
  for (int i = 0; i < m->species_count; i++)
    if (tok_double(&tk, &m->species_coeffs[i])) FAIL(-5, "bad species_coeffs entry");
  NEXT(SEPS_BRACE); /* :564-569 */
  if (strcmp(kw, "moment_coeffs")) FAIL(-5, "Error reading MTP file. Moment coefficients not found.");
  m->linear_coeffs = (double *) calloc((size_t) m->alpha_scalar_count, sizeof(double));
  for (int i = 0; i < m->alpha_scalar_count; i++)
    if (tok_double(&tk, &m->linear_coeffs[i])) FAIL(-5, "bad moment_coeffs entry");

  m->coeff_count = m->species_count * m->species_count * m->radial_func_count * m->radial_basis_size +
      m->species_count + m->alpha_scalar_count; /* pair_mtp_extrapolation.cpp:533 */

  if (want_selection) { /* pair_mtp_extrapolation.cpp:545-612 */
    rd.ignore_comments = 0;
    tok_free(&tk);
    line = reader_next_line(&rd);
    if (!line)
      FAIL(-8, "No selection state found! Consider training/retraining or disabling extrapolation!");
    tok_init(&tk, line, SEPS);
    kw = tok_next(&tk);
    if (!kw || strcmp(kw, "#MVS_v1.1"))
      FAIL(-8, "Error in reading MTP file selection state. Please verify MVS version is #MVS_v1.1!");
    rd.ignore_comments = 1;
    double energy_weight_d = 0, site_en_weight_d = 0, dummy;
    static const char *names[5] = {"energy_weight", "force_weight", "stress_weight", "site_en_weight",
                                   "weight_scaling"};
    for (int w = 0; w < 5; w++) {
      NEXT(SEPS);
      if (strcmp(kw, names[w])) FAIL(-8, "Error in reading MTP file, %s", names[w]);
      if (w == 0 && tok_double(&tk, &energy_weight_d)) FAIL(-8, "bad energy_weight");
      if (w == 3 && tok_double(&tk, &site_en_weight_d)) FAIL(-8, "bad site_en_weight");
      (void) dummy;
    }
    int energy_weight = (int) energy_weight_d, site_en_weight = (int) site_en_weight_d; /* :576,592 */
    if (energy_weight + site_en_weight > 1)
      FAIL(-9, "Error, the MTP currently only supports configuration mode (energy_weight=1) "
               "or neighbourhood mode (site_en_weight=1). Please retrain the MTP with the correct modes!");
    m->configuration_mode = (energy_weight == 1);
    size_t n = (size_t) m->coeff_count * m->coeff_count;
    m->active_set = (double *) malloc(n * sizeof(double));
    m->inverse_active_set = (double *) malloc(n * sizeof(double));
    fgetc(fp); /* :607 skip the '#' in front of the binary block */
    if (fread(m->active_set, sizeof(double), n, fp) != n ||
        fread(m->inverse_active_set, sizeof(double), n, fp) != n)
      FAIL(-10, "Unexpected end of file while reading the active set");
    m->has_selection = 1;
  }

done:
  tok_free(&tk);
  free(rd.line);
  fclose(fp);
  if (rc) mtp_oracle_free(m);
  return rc;
#undef NEXT
}

void mtp_oracle_free(mtp_oracle_model *m)
{
  free(m->alpha_index_basic);
  free(m->alpha_index_times);
  free(m->alpha_moment_mapping);
  free(m->radial_basis_coeffs);
  free(m->linear_coeffs);
  free(m->species_coeffs);
  free(m->active_set);
  free(m->inverse_active_set);
  memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------------------------------ */
/* mtp_rb_chevbyshev_basis.cpp:29-54 */
void mtp_oracle_radial_basis(const mtp_oracle_model *m, double dist, double *vals, double *ders)
{
  const double min_cutoff = m->min_cutoff, max_cutoff = m->max_cutoff, scaling = m->scaling;
  const int size = m->radial_basis_size;
  double ksi = (2 * dist - (min_cutoff + max_cutoff)) / (max_cutoff - min_cutoff);
  vals[0] = scaling * (1 * (dist - max_cutoff) * (dist - max_cutoff));
  if (size > 1) vals[1] = scaling * (ksi * (dist - max_cutoff) * (dist - max_cutoff));
  for (int i = 2; i < size; i++) vals[i] = 2 * ksi * vals[i - 1] - vals[i - 2];
  if (!ders) return;
  double mult = 2.0 / (max_cutoff - min_cutoff);
  ders[0] = scaling * 2 * (dist - max_cutoff);
  if (size > 1)
    ders[1] = scaling * (mult * (dist - max_cutoff) * (dist - max_cutoff) + 2 * ksi * (dist - max_cutoff));
  for (int i = 2; i < size; i++) ders[i] = 2 * (mult * vals[i - 1] + ksi * ders[i - 1]) - ders[i - 2];
}

/* pair_mtp_extrapolation.cpp:347-358 */
double mtp_oracle_grade(const mtp_oracle_model *m, const double *c)
{
  const int C = m->coeff_count;
  double max_grade = 0;
  for (int i = 0; i < C; i++) {
    double g = 0;
    const double *row = m->inverse_active_set + (size_t) i * C;
    for (int j = 0; j < C; j++) g += c[j] * row[j];
    if (fabs(g) > max_grade) max_grade = fabs(g);
  }
  return max_grade;
}

/* ------------------------------------------------------------------------------------ */
/* One implementation serves both compute paths; `ext` switches on the extras of
 * pair_mtp_extrapolation.cpp (radial Jacobian :193-198, candidate vector :235-252 and
 * :323-329, grade :332-336).  With ext == 0 it is pair_mtp.cpp:72-280 line for line. */
/* f += v as the reference writes it (pair_mtp.cpp:248-254); when several threads of the cpu_baseline driver
 * (mtp_oracle_mt.c) share one force array the same add is made atomic (compare-and-swap on the bit pattern) */
static _Thread_local int shared_force_array = 0;
void mtp_oracle_share_force_array(int on) { shared_force_array = on; }
static inline void force_add(double *p, double v)
{
  if (!shared_force_array) {
    *p += v;
    return;
  }
  unsigned long long *q = (unsigned long long *) p, old = __atomic_load_n(q, __ATOMIC_RELAXED), neu;
  do {
    double d;
    memcpy(&d, &old, sizeof d);
    d += v;
    memcpy(&neu, &d, sizeof d);
  } while (!__atomic_compare_exchange_n(q, &old, neu, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}

static int compute_impl(const mtp_oracle_model *m, int ext, int inum, const int *ilist,
                        const int *first, const int *neigh, const double *x, const int *type,
                        int eflag, int vflag, double *f, double *eng_vdwl, double *eatom,
                        double *virial, double *vatom, double *grades, double *max_grade_out,
                        double *coeff_ders_out, long natoms)
{
  const int Sp = m->species_count, R = m->radial_basis_size, Mu = m->radial_func_count;
  const int A = m->alpha_moment_count, B = m->alpha_index_basic_count;
  const int T = m->alpha_index_times_count, S = m->alpha_scalar_count;
  const int P = m->max_alpha_index_basic;
  const int per_pair = R * Mu, radial_coeff_count = Sp * Sp * per_pair;
  const int C = m->coeff_count;
  const double cutsq = m->max_cutoff * m->max_cutoff; /* pair_mtp.cpp:449,456 */
  const int eflag_global = eflag & 1, eflag_atom = eflag & 2, vflag_atom = vflag & 4;
  const int(*aib)[4] = (const int(*)[4]) m->alpha_index_basic;
  const int(*ait)[4] = (const int(*)[4]) m->alpha_index_times;
  int rc = 0;

  int jac_size = 0;
  double *moment_jacobian = NULL; /* [jnum][B][3] */
  char *within_cutoff = NULL;
  double *mom = (double *) malloc(sizeof(double) * A);
  double *ders = (double *) malloc(sizeof(double) * A);
  double *rb_vals = (double *) malloc(sizeof(double) * R);
  double *rb_ders = (double *) malloc(sizeof(double) * R);
  double *dist_powers = (double *) malloc(sizeof(double) * P);
  double(*coord_powers)[3] = (double(*)[3]) malloc(sizeof(double) * 3 * P);
  double *radial_vals = (double *) malloc(sizeof(double) * Mu);
  double *radial_ders = (double *) malloc(sizeof(double) * Mu);
  double *radial_jacobian = NULL, *coeff_ders = NULL; /* [B][Sp][per_pair], [C] */
  double max_grade = 0;
  if (ext) {
    radial_jacobian = (double *) malloc(sizeof(double) * B * Sp * per_pair);
    coeff_ders = (double *) calloc((size_t) C, sizeof(double)); /* :97-98 */
  }
  dist_powers[0] = coord_powers[0][0] = coord_powers[0][1] = coord_powers[0][2] = 1; /* :647 */

  for (int ii = 0; ii < inum; ii++) { /* :88 */
    const int i = ilist[ii];
    const int itype = type[i] - 1;
    if (itype >= Sp) { rc = -1; goto done; }
    const int jnum = first[ii + 1] - first[ii];
    const int *jlist = neigh + first[ii];
    double nbh_energy = 0;
    const double xi[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};

    if (jac_size < jnum) { /* :99-104 */
      moment_jacobian = (double *) realloc(moment_jacobian, sizeof(double) * (size_t) jnum * B * 3);
      within_cutoff = (char *) realloc(within_cutoff, (size_t) jnum);
      jac_size = jnum;
    }
    for (int k = 0; k < A; k++) mom[k] = 0.0;
    for (int k = 0; k < A; k++) ders[k] = 0.0;
    if (ext) {
      for (int k = 0; k < B * Sp * per_pair; k++) radial_jacobian[k] = 0.0; /* ext :124-127 */
      if (!m->configuration_mode)
        for (int k = 0; k < C; k++) coeff_ders[k] = 0.0; /* ext :129-130 */
    }

    for (int jj = 0; jj < jnum; jj++) { /* :112 */
      const int j = jlist[jj] & MTP_ORACLE_NEIGHMASK;
      const int jtype = type[j] - 1;
      if (jtype >= Sp) { rc = -1; goto done; }
      const double r[3] = {x[3 * j] - xi[0], x[3 * j + 1] - xi[1], x[3 * j + 2] - xi[2]};
      const double rsq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      if (rsq > cutsq) { /* :123 */
        within_cutoff[jj] = 0;
        continue;
      }
      within_cutoff[jj] = 1;
      const double dist = sqrt(rsq);
      mtp_oracle_radial_basis(m, dist, rb_vals, rb_ders); /* :130 */

      for (int k = 1; k < P; k++) { /* :133-136 */
        dist_powers[k] = dist_powers[k - 1] * dist;
        for (int a = 0; a < 3; a++) coord_powers[k][a] = coord_powers[k - 1][a] * r[a];
      }
      for (int mu = 0; mu < Mu; mu++) { /* :139-151 */
        double val = 0, der = 0;
        int pair_offset = itype * Sp + jtype;
        int offset = pair_offset * per_pair + mu * R;
        for (int ri = 0; ri < R; ri++) {
          val += m->radial_basis_coeffs[offset + ri] * rb_vals[ri];
          der += m->radial_basis_coeffs[offset + ri] * rb_ders[ri];
        }
        radial_vals[mu] = val;
        radial_ders[mu] = der;
      }
      double *jac = moment_jacobian + (size_t) jj * B * 3;
      for (int k = 0; k < B; k++) { /* :154-192 */
        int mu = aib[k][0];
        double val = radial_vals[mu];
        double der = radial_ders[mu];
        int norm_rank = aib[k][1] + aib[k][2] + aib[k][3];
        double norm_fac = 1.0 / dist_powers[norm_rank];
        double pow0 = coord_powers[aib[k][1]][0];
        double pow1 = coord_powers[aib[k][2]][1];
        double pow2 = coord_powers[aib[k][3]][2];
        double pw = pow0 * pow1 * pow2;
        if (ext) { /* ext :193-198 */
          double *rj = radial_jacobian + ((size_t) k * Sp + jtype) * per_pair + mu * R;
          for (int ri = 0; ri < R; ri++) rj[ri] += rb_vals[ri] * norm_fac * pw;
        }
        val *= norm_fac;
        der = der * norm_fac - norm_rank * val / dist;
        mom[k] += val * pw;
        pw *= der / dist;
        jac[3 * k + 0] = pw * r[0];
        jac[3 * k + 1] = pw * r[1];
        jac[3 * k + 2] = pw * r[2];
        if (aib[k][1] != 0) jac[3 * k + 0] += val * aib[k][1] * coord_powers[aib[k][1] - 1][0] * pow1 * pow2;
        if (aib[k][2] != 0) jac[3 * k + 1] += val * aib[k][2] * pow0 * coord_powers[aib[k][2] - 1][1] * pow2;
        if (aib[k][3] != 0) jac[3 * k + 2] += val * aib[k][3] * pow0 * pow1 * coord_powers[aib[k][3] - 1][2];
      }
    }

    for (int k = 0; k < T; k++) { /* :196-201 */
      double val0 = mom[ait[k][0]];
      double val1 = mom[ait[k][1]];
      int val2 = ait[k][2];
      mom[ait[k][3]] += val2 * val0 * val1;
    }

    if (!ext) {
      if (eflag_atom || eflag_global) { /* :204-212 */
        nbh_energy = m->species_coeffs[itype];
        for (int k = 0; k < S; k++) nbh_energy += m->linear_coeffs[k] * mom[m->alpha_moment_mapping[k]];
        if (eflag_atom) eatom[i] = nbh_energy;
        if (eflag_global) *eng_vdwl += nbh_energy;
      }
    } else { /* ext :235-252 */
      int linear_basis_offset = radial_coeff_count + Sp;
      if (eflag_atom || eflag_global) {
        nbh_energy = m->species_coeffs[itype];
        for (int k = 0; k < S; k++) {
          double basis_member = mom[m->alpha_moment_mapping[k]];
          coeff_ders[linear_basis_offset + k] += basis_member;
          nbh_energy += m->linear_coeffs[k] * basis_member;
        }
        if (eflag_atom) eatom[i] = nbh_energy;
        if (eflag_global) *eng_vdwl += nbh_energy;
      } else
        for (int k = 0; k < S; k++) coeff_ders[linear_basis_offset + k] += mom[m->alpha_moment_mapping[k]];
      coeff_ders[radial_coeff_count + itype] += 1;
    }

    for (int k = 0; k < S; k++) ders[m->alpha_moment_mapping[k]] = m->linear_coeffs[k]; /* :217-218 */
    for (int k = T - 1; k >= 0; k--) { /* :221-233 */
      int a0 = ait[k][0], a1 = ait[k][1], multiplier = ait[k][2], a3 = ait[k][3];
      double val0 = mom[a0], val1 = mom[a1], val3 = ders[a3];
      ders[a1] += val3 * multiplier * val0;
      ders[a0] += val3 * multiplier * val1;
    }

    for (int jj = 0; jj < jnum; jj++) { /* :236-278 */
      const int j = jlist[jj] & MTP_ORACLE_NEIGHMASK;
      if (!within_cutoff[jj]) continue;
      const double *jac = moment_jacobian + (size_t) jj * B * 3;
      double temp_force[3] = {0, 0, 0};
      for (int k = 0; k < B; k++)
        for (int a = 0; a < 3; a++) temp_force[a] += ders[k] * jac[3 * k + a];
      force_add(&f[3 * i + 0], temp_force[0]);
      force_add(&f[3 * i + 1], temp_force[1]);
      force_add(&f[3 * i + 2], temp_force[2]);
      force_add(&f[3 * j + 0], -temp_force[0]);
      force_add(&f[3 * j + 1], -temp_force[1]);
      force_add(&f[3 * j + 2], -temp_force[2]);
      if (vflag) {
        const double r[3] = {x[3 * j] - xi[0], x[3 * j + 1] - xi[1], x[3 * j + 2] - xi[2]};
        virial[0] -= temp_force[0] * r[0];
        virial[1] -= temp_force[1] * r[1];
        virial[2] -= temp_force[2] * r[2];
        virial[3] -= (temp_force[0] * r[1] + temp_force[1] * r[0]) / 2;
        virial[4] -= (temp_force[0] * r[2] + temp_force[2] * r[0]) / 2;
        virial[5] -= (temp_force[1] * r[2] + temp_force[2] * r[1]) / 2;
        if (vflag_atom) {
          double *va = vatom + 6 * (size_t) i;
          va[0] -= temp_force[0] * r[0];
          va[1] -= temp_force[1] * r[1];
          va[2] -= temp_force[2] * r[2];
          va[3] -= (temp_force[0] * r[1] + temp_force[1] * r[0]) / 2;
          va[4] -= (temp_force[0] * r[2] + temp_force[2] * r[0]) / 2;
          va[5] -= (temp_force[1] * r[2] + temp_force[2] * r[1]) / 2;
        }
      }
    }

    if (ext) {
      for (int k = 0; k < B; k++) /* ext :323-329 */
        for (int jjtype = 0; jjtype < Sp; jjtype++) {
          int offset = (itype * Sp + jjtype) * per_pair;
          const double *rj = radial_jacobian + ((size_t) k * Sp + jjtype) * per_pair;
          for (int ri = 0; ri < per_pair; ri++) coeff_ders[offset + ri] += ders[k] * rj[ri];
        }
      if (!m->configuration_mode) { /* ext :332-336 */
        double grade = mtp_oracle_grade(m, coeff_ders);
        if (grade > max_grade) max_grade = grade;
        if (grades) grades[i] = grade;
      }
    }
  }

  if (ext) { /* compile_grades, ext :363-382 (single rank: the all-reduces are identities) */
    if (coeff_ders_out) memcpy(coeff_ders_out, coeff_ders, sizeof(double) * C);
    if (m->configuration_mode) {
      max_grade = mtp_oracle_grade(m, coeff_ders);
      if (natoms > 0) max_grade /= (double) natoms;
      else max_grade = 0.0;
    }
    if (max_grade_out) *max_grade_out = max_grade;
  }

done:
  free(moment_jacobian);
  free(within_cutoff);
  free(mom);
  free(ders);
  free(rb_vals);
  free(rb_ders);
  free(dist_powers);
  free(coord_powers);
  free(radial_vals);
  free(radial_ders);
  free(radial_jacobian);
  free(coeff_ders);
  return rc;
}

int mtp_oracle_compute(const mtp_oracle_model *m, int inum, const int *ilist, const int *first,
                       const int *neigh, const double *x, const int *type, int eflag, int vflag,
                       double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
  return compute_impl(m, 0, inum, ilist, first, neigh, x, type, eflag, vflag, f, eng_vdwl, eatom,
                      virial, vatom, NULL, NULL, NULL, 0);
}

int mtp_oracle_compute_extrapolation(const mtp_oracle_model *m, int inum, const int *ilist,
                                     const int *first, const int *neigh, const double *x,
                                     const int *type, int eflag, int vflag, double *f,
                                     double *eng_vdwl, double *eatom, double *virial,
                                     double *vatom, double *grades, double *max_grade,
                                     double *coeff_ders, long natoms)
{
  if (!m->has_selection) return -2;
  return compute_impl(m, 1, inum, ilist, first, neigh, x, type, eflag, vflag, f, eng_vdwl, eatom,
                      virial, vatom, grades, max_grade, coeff_ders, natoms);
}
