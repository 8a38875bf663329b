/* TEST INFRASTRUCTURE ONLY -- a host-memory stand-in for the five RCCL entry points libl2q.so resolves
 * at run time (csrc/comm.hip; selected with L2Q_RCCL_LIB).  Ranks are processes on one machine and meet
 * through files in the directory FAKE_RCCL_DIR: ncclCommInitRank is a rendezvous of exactly `nranks`
 * distinct ranks under ONE unique id (a rank that arrives with another id, a duplicate rank or a missing
 * rank makes it fail after FAKE_RCCL_TIMEOUT_S seconds instead of hanging), ncclAllReduce sums the ranks'
 * HOST buffers in rank order.  It exists so that the bootstrap of utils.dist.NativeComm (id drawn on rank
 * 0, broadcast over the torch.distributed group, communicator created on every rank, collective, destroy)
 * can run with world_size 2 on the CPU-only build container. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct fake_comm { int nranks, rank, seq; char tag[40]; } *ncclComm_t;

static const char* dir(void) { const char* d = getenv("FAKE_RCCL_DIR"); return d ? d : "/tmp"; }
static double timeout_s(void) { const char* t = getenv("FAKE_RCCL_TIMEOUT_S"); return t ? atof(t) : 20.0; }
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static int wait_for(const char* path) {
  const double t0 = now();
  while (access(path, F_OK) != 0) {
    if (now() - t0 > timeout_s()) return 0;
    usleep(2000);
  }
  return 1;
}

static void publish(const char* path, const void* data, size_t n) {   /* atomic: write + rename */
  char tmp[600];
  snprintf(tmp, sizeof tmp, "%s.tmp%d", path, (int)getpid());
  FILE* f = fopen(tmp, "wb");
  if (n) fwrite(data, 1, n, f);
  fclose(f);
  rename(tmp, path);
}

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "fake-%d-%ld", (int)getpid(), (long)(now() * 1e6));
  return 0;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  ncclComm_t c = (ncclComm_t)calloc(1, sizeof **out);
  c->nranks = nranks; c->rank = rank;
  unsigned h = 2166136261u;
  for (int i = 0; i < 128; ++i) h = (h ^ (unsigned char)id.internal[i]) * 16777619u;
  snprintf(c->tag, sizeof c->tag, "%08x", h);
  char path[512];
  snprintf(path, sizeof path, "%s/%s.init.%d", dir(), c->tag, rank);
  if (access(path, F_OK) == 0) { free(c); return 5; }          /* duplicate rank: invalid usage */
  publish(path, "", 0);
  for (int r = 0; r < nranks; ++r) {
    snprintf(path, sizeof path, "%s/%s.init.%d", dir(), c->tag, r);
    if (!wait_for(path)) { free(c); return 6; }                 /* a rank never arrived under this id */
  }
  *out = c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t c, void* stream) {
  (void)stream;
  if (op != 0 || (dtype != 7 && dtype != 8)) return 4;
  const size_t eb = dtype == 8 ? 8 : 4, nbytes = count * eb;
  char path[512];
  const int seq = c->seq++;
  snprintf(path, sizeof path, "%s/%s.ar%d.%d", dir(), c->tag, seq, c->rank);
  publish(path, send, nbytes);
  char* tmp = (char*)malloc(nbytes);
  double* accd = (double*)calloc(count, sizeof(double));
  for (int r = 0; r < c->nranks; ++r) {
    snprintf(path, sizeof path, "%s/%s.ar%d.%d", dir(), c->tag, seq, r);
    if (!wait_for(path)) { free(tmp); free(accd); return 6; }
    FILE* f = fopen(path, "rb");
    if (fread(tmp, 1, nbytes, f) != nbytes) { fclose(f); free(tmp); free(accd); return 2; }
    fclose(f);
    for (size_t i = 0; i < count; ++i) accd[i] += dtype == 8 ? ((double*)tmp)[i] : ((float*)tmp)[i];
  }
  for (size_t i = 0; i < count; ++i) {
    if (dtype == 8) ((double*)recv)[i] = accd[i]; else ((float*)recv)[i] = (float)accd[i];
  }
  free(tmp); free(accd);
  return 0;
}

int ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
int ncclCommAbort(ncclComm_t c) { free(c); return 0; }
/* version code major * 10000 + minor * 100 + patch; FAKE_RCCL_VERSION overrides it (an ABI the wrapper refuses) */
int ncclGetVersion(int* v) {
  const char* e = getenv("FAKE_RCCL_VERSION");
  *v = e ? atoi(e) : 22105;
  return 0;
}

const char* ncclGetErrorString(int e) {
  switch (e) {
    case 0: return "no error";
    case 5: return "invalid usage (duplicate rank)";
    case 6: return "remote rank did not arrive (timeout)";
    default: return "fake rccl error";
  }
}
