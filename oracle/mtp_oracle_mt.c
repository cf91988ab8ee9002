/* TEST INFRASTRUCTURE ONLY -- threads-over-atoms driver around the serial CPU restatement
 * (mtp_oracle_compute, which follows /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:72-280), used for bench.py's
 * `cpu_baseline` leg (ii): every thread walks a contiguous slice of ilist and adds its forces into the ONE force
 * array with atomic adds (the reference scatters forces onto neighbours, pair_mtp.cpp:248-254; slices are spatial
 * slabs, so contention is confined to slab borders).  The reference itself is serial per MPI rank; threads over
 * atoms with no halo cost are an optimistic stand-in for "one rank per core" (SURVEY.md section 8d), i.e.
 * conservative for any speed-up claim.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mtp_oracle.h"

typedef struct {
  const mtp_oracle_model *m;
  int tid, nthreads, inum;
  const int *ilist, *first, *neigh, *type;
  const double *x;
  int eflag, vflag;
  double *f, *eatom, *vatom;
  double eng, virial[6];
  int rc;
  char pad[64];   /* job records of neighbouring threads on different cache lines */
} job_t;

static void *worker(void *arg)
{
  job_t *j = (job_t *) arg;
  const long r0 = (long) j->inum * j->tid / j->nthreads, r1 = (long) j->inum * (j->tid + 1) / j->nthreads;
  /* tallies on this thread's stack: the serial code updates them per pair */
  double eng = 0.0, virial[6] = {0, 0, 0, 0, 0, 0};
  mtp_oracle_share_force_array(1);
  /* `first` holds absolute offsets into neigh, so a slice of rows is the same list seen from row r0 */
  j->rc = mtp_oracle_compute(j->m, (int) (r1 - r0), j->ilist + r0, j->first + r0, j->neigh, j->x, j->type, j->eflag,
                             j->vflag, j->f, &eng, j->eatom, virial, j->vatom);
  mtp_oracle_share_force_array(0);
  j->eng = eng;
  memcpy(j->virial, virial, sizeof(virial));
  return NULL;
}

/* Same contract as mtp_oracle_compute (f, virial, vatom, *eng_vdwl accumulate; eatom assigned), nthreads >= 1. */
int mtp_oracle_compute_mt(const mtp_oracle_model *m, int nthreads, int nall, int inum, const int *ilist,
                          const int *first, const int *neigh, const double *x, const int *type, int eflag,
                          int vflag, double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
  (void) nall;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > inum && inum > 0) nthreads = inum;
  job_t *jobs = (job_t *) calloc((size_t) nthreads, sizeof(job_t));
  pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
  int rc = 0, started = 0;
  if (!jobs || !th) rc = -2;
  for (int t = 0; rc == 0 && t < nthreads; t++) {
    job_t *j = &jobs[t];
    j->m = m;
    j->tid = t;
    j->nthreads = nthreads;
    j->inum = inum;
    j->ilist = ilist;
    j->first = first;
    j->neigh = neigh;
    j->type = type;
    j->x = x;
    j->eflag = eflag;
    j->vflag = vflag;
    j->f = f;
    j->eatom = eatom;
    j->vatom = vatom;
    if (pthread_create(&th[t], NULL, worker, j) != 0) {
      rc = -2;
      break;
    }
    started++;
  }
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  for (int t = 0; t < started; t++) {
    if (jobs[t].rc) rc = jobs[t].rc;
    if (eng_vdwl) *eng_vdwl += jobs[t].eng;
    if (virial)
      for (int q = 0; q < 6; q++) virial[q] += jobs[t].virial[q];
  }
  free(th);
  free(jobs);
  return rc;
}
