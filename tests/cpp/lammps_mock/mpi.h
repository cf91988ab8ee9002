// MOCK: one-rank MPI for the adapter's test driver (see lammps_mock.h)
#pragma once
#include <cstring>
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef int MPI_Info;
struct MPI_Status {
  int count;
};
#define MPI_IN_PLACE ((void *) 1)
#define MPI_DOUBLE 8
#define MPI_INT 4
#define MPI_CHAR 1
#define MPI_SUM 1
#define MPI_MAX 2
#define MPI_COMM_TYPE_SHARED 1
#define MPI_INFO_NULL 0
inline int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm) { return 0; }
inline int MPI_Scan(const void *s, void *r, int n, MPI_Datatype t, MPI_Op, MPI_Comm)
{
  std::memcpy(r, s, (size_t) n * (size_t) t);
  return 0;
}
inline int MPI_Comm_split_type(MPI_Comm, int, int, MPI_Info, MPI_Comm *out)
{
  *out = 0;
  return 0;
}
inline int MPI_Comm_rank(MPI_Comm, int *r)
{
  *r = 0;
  return 0;
}
inline int MPI_Comm_free(MPI_Comm *) { return 0; }
inline int MPI_Send(const void *, int, MPI_Datatype, int, int, MPI_Comm) { return 0; }
inline int MPI_Recv(void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Status *) { return 0; }
inline int MPI_Probe(int, int, MPI_Comm, MPI_Status *st)
{
  st->count = 0;
  return 0;
}
inline int MPI_Get_count(const MPI_Status *st, MPI_Datatype, int *n)
{
  *n = st->count;
  return 0;
}
