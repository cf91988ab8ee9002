/* TEST INFRASTRUCTURE ONLY -- threads-over-atoms driver around the serial CPU restatement
 * (mtp_oracle_compute, which follows /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:72-280), used for bench.py's
 * `cpu_baseline` leg (ii): every thread walks a contiguous slice of ilist with a PRIVATE force array (the
 * reference scatters forces onto neighbours, pair_mtp.cpp:248-254), then all threads sum the private arrays slice
 * by slice.  The reference itself is serial per MPI rank; threads over atoms with no halo cost are an optimistic
 * stand-in for "one rank per core" (SURVEY.md section 8d), i.e. conservative for any speed-up claim.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mtp_oracle.h"

typedef struct {
  const mtp_oracle_model *m;
  int tid, nthreads, nall, inum;
  const int *ilist, *first, *neigh, *type;
  const double *x;
  int eflag, vflag;
  double **fpriv;        /* [nthreads] private force arrays, [nall][3] */
  double *f;             /* shared result (accumulated) */
  double eng, virial[6];
  double *eatom, *vatom;
  pthread_barrier_t *bar;
  int rc;
} job_t;

static void *worker(void *arg)
{
  job_t *j = (job_t *) arg;
  const long r0 = (long) j->inum * j->tid / j->nthreads, r1 = (long) j->inum * (j->tid + 1) / j->nthreads;
  double *fp = j->fpriv[j->tid];
  memset(fp, 0, sizeof(double) * 3 * (size_t) j->nall);
  /* tallies on this thread's stack: the serial code updates them per pair, and the job records of neighbouring
   * threads share cache lines */
  double eng = 0.0, virial[6] = {0, 0, 0, 0, 0, 0};
  /* `first` holds absolute offsets into neigh, so a slice of rows is the same list seen from row r0 */
  j->rc = mtp_oracle_compute(j->m, (int) (r1 - r0), j->ilist + r0, j->first + r0, j->neigh, j->x, j->type, j->eflag,
                             j->vflag, fp, &eng, j->eatom, virial, j->vatom);
  j->eng = eng;
  memcpy(j->virial, virial, sizeof(virial));
  pthread_barrier_wait(j->bar);
  /* reduction: thread t owns the coordinates [c0, c1) of every private array */
  const long n3 = 3L * j->nall, c0 = n3 * j->tid / j->nthreads, c1 = n3 * (j->tid + 1) / j->nthreads;
  for (int t = 0; t < j->nthreads; t++) {
    const double *src = j->fpriv[t];
    for (long c = c0; c < c1; c++) j->f[c] += src[c];
  }
  return NULL;
}

/* Same contract as mtp_oracle_compute (f, virial, vatom, *eng_vdwl accumulate; eatom assigned), nthreads >= 1. */
int mtp_oracle_compute_mt(const mtp_oracle_model *m, int nthreads, int nall, int inum, const int *ilist,
                          const int *first, const int *neigh, const double *x, const int *type, int eflag,
                          int vflag, double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
  if (nthreads < 1) nthreads = 1;
  if (nthreads > inum && inum > 0) nthreads = inum;
  job_t *jobs = (job_t *) calloc((size_t) nthreads, sizeof(job_t));
  pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
  double **fpriv = (double **) calloc((size_t) nthreads, sizeof(double *));
  pthread_barrier_t bar;
  int rc = 0, started = 0;
  if (!jobs || !th || !fpriv) rc = -2;
  for (int t = 0; rc == 0 && t < nthreads; t++)
    if (!(fpriv[t] = (double *) malloc(sizeof(double) * 3 * (size_t) (nall > 0 ? nall : 1)))) rc = -2;
  if (rc == 0 && pthread_barrier_init(&bar, NULL, (unsigned) nthreads) != 0) rc = -2;
  if (rc == 0) {
    for (int t = 0; t < nthreads; t++) {
      job_t *j = &jobs[t];
      j->m = m;
      j->tid = t;
      j->nthreads = nthreads;
      j->nall = nall;
      j->inum = inum;
      j->ilist = ilist;
      j->first = first;
      j->neigh = neigh;
      j->type = type;
      j->x = x;
      j->eflag = eflag;
      j->vflag = vflag;
      j->fpriv = fpriv;
      j->f = f;
      j->eatom = eatom;
      j->vatom = vatom;
      j->bar = &bar;
      if (pthread_create(&th[t], NULL, worker, j) != 0) {
        rc = -2;   /* cannot happen without leaving the started threads stuck at the barrier: abort hard */
        abort();
      }
      started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    for (int t = 0; t < nthreads; t++) {
      if (jobs[t].rc) rc = jobs[t].rc;
      if (eng_vdwl) *eng_vdwl += jobs[t].eng;
      if (virial)
        for (int q = 0; q < 6; q++) virial[q] += jobs[t].virial[q];
    }
  }
  if (fpriv)
    for (int t = 0; t < nthreads; t++) free(fpriv[t]);
  free(fpriv);
  free(th);
  free(jobs);
  return rc;
}
