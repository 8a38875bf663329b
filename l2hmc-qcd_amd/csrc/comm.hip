// comm.hip -- per-device initialisation and the ONE collective of the path: the sum of the flat
// training-gradient buffer over ranks (RCCL over xGMI).  The reference's counterpart is the DDP
// wrap of the dynamics (trainers/pytorch/trainer.py:246-257, bucketed all-reduce in
// loss.backward(), :1296-1304) on the process group of utils/dist.py:126-144.  Sampling needs no
// collective: chains are independent and sharded over ranks.
//
// librccl is resolved at run time (dlopen of the soname): libl2q.so itself has no link-time
// dependency on it, and inside a PyTorch process the copy PyTorch already loaded is the one used.
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>

#include "l2q_common.hpp"

// The handful of RCCL declarations this file needs, spelled out (values as in <rccl/rccl.h> of
// ROCm 7.x = NCCL's ABI): building libl2q.so needs neither the RCCL headers nor the library.
extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace l2q {
namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional: teardown without a handshake
  int version = 0;
  bool ok = false;
  char why[256] = "librccl not loaded";
};

// Resolved ONCE (std::call_once: concurrent first calls are safe); a failed resolution is
// remembered with its reason, and EVERY entry that finds it failed reports that reason again
// (l2q_last_error must not be left holding an unrelated, older message).
// L2Q_RCCL_LIB names another library with the same five entry points (tests: a host-memory fake).
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("L2Q_RCCL_LIB");
    const char* names[3] = {env && *env ? env : "librccl.so.1", "librccl.so", nullptr};
    for (int i = 0; names[i] && !r.h; ++i) {
      r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
      if (env && *env) break;                       // an explicit choice is not second-guessed
    }
    if (!r.h) {
      const char* e = dlerror();
      snprintf(r.why, sizeof r.why, "%s not found (%s)", names[0], e ? e : "dlopen failed");
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.h, "ncclCommAbort");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString;
    if (!r.ok) { snprintf(r.why, sizeof r.why, "librccl: missing symbols"); return; }
    // The declarations above are spelled out by hand (ncclUniqueId = 128 bytes, ncclFloat32 = 7,
    // ncclFloat64 = 8, ncclSum = 0: the ABI of NCCL / RCCL 2.x).  Whatever library was loaded must say it
    // speaks that ABI: ncclGetVersion's code is major * 10000 + minor * 100 + patch (major * 1000 + ... before
    // 2.9); anything but major 2 is refused instead of being called with a guessed layout.
    if (!r.GetVersion || r.GetVersion(&r.version) != ncclSuccess) {
      r.ok = false;
      snprintf(r.why, sizeof r.why, "librccl: ncclGetVersion missing or failing (cannot verify the ABI)");
      return;
    }
    const int major = r.version >= 10000 ? r.version / 10000 : r.version / 1000;
    if (major != 2) {
      r.ok = false;
      snprintf(r.why, sizeof r.why, "librccl reports version code %d (major %d); this wrapper declares the "
               "2.x ABI by hand and refuses anything else", r.version, major);
    }
  });
  if (!r.ok) set_error("%s", r.why);
  return r;
}

int rccl_fail(const char* what, ncclResult_t e) {
  set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "rccl error");
  return L2Q_EHIP;
}

}  // namespace
}  // namespace l2q

using namespace l2q;

extern "C" {

static_assert(sizeof(ncclUniqueId) == L2Q_COMM_ID_BYTES, "ncclUniqueId size");

int l2q_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("l2q_init: no HIP device visible");
    return L2Q_EHIP;
  }
  L2Q_REQUIRE(device >= 0 && device < n && device < 16, L2Q_EINVAL, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) {
    set_error("l2q_init: hipSetDevice(%d) failed", device);
    return L2Q_EHIP;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return L2Q_EHIP;
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    set_error("l2q_init: device %d is %s; libl2q.so carries gfx950 code only", device, p.gcnArchName);
    return L2Q_EHIP;
  }
  (void)tuning();          // this device's knob table (defaults on first touch)
  return device;
}

int l2q_comm_unique_id(void* id_out) {
  L2Q_REQUIRE(id_out, L2Q_EINVAL, "null pointer");
  Rccl& r = rccl();
  if (!r.ok) return L2Q_EHIP;
  ncclUniqueId id;
  ncclResult_t e = r.GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, &id, sizeof id);
  return L2Q_OK;
}

int l2q_comm_init(const void* id, int nranks, int rank, void** comm_out) {
  L2Q_REQUIRE(id && comm_out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, L2Q_EINVAL, "bad rank / nranks");
  Rccl& r = rccl();
  if (!r.ok) return L2Q_EHIP;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t c = nullptr;
  ncclResult_t e = r.CommInitRank(&c, nranks, uid, rank);
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  *comm_out = (void*)c;
  return L2Q_OK;
}

int l2q_allreduce_grads(void* comm, void* grad, long n, int elem_bytes, void* stream) {
  L2Q_REQUIRE(comm && grad, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(elem_bytes == 4 || elem_bytes == 8, L2Q_EINVAL, "elem_bytes must be 4 (fp32) or 8 (fp64)");
  Rccl& r = rccl();
  if (!r.ok) return L2Q_EHIP;
  ncclResult_t e = r.AllReduce(grad, grad, (size_t)n, elem_bytes == 8 ? ncclFloat64 : ncclFloat32, ncclSum,
                               (ncclComm_t)comm, (hipStream_t)stream);
  if (e != ncclSuccess) return rccl_fail("ncclAllReduce", e);
  return L2Q_OK;
}

int l2q_comm_destroy(void* comm) {
  L2Q_REQUIRE(comm, L2Q_EINVAL, "null pointer");
  Rccl& r = rccl();
  if (!r.ok) return L2Q_EHIP;
  ncclResult_t e = r.CommDestroy((ncclComm_t)comm);
  if (e != ncclSuccess) return rccl_fail("ncclCommDestroy", e);
  return L2Q_OK;
}

int l2q_comm_abort(void* comm) {
  L2Q_REQUIRE(comm, L2Q_EINVAL, "null pointer");
  Rccl& r = rccl();
  if (!r.ok) return L2Q_EHIP;
  // ncclCommAbort frees the communicator without waiting for the peers (garbage collection / interpreter
  // shutdown, when they may be gone already); a library without it falls back to the orderly destroy
  ncclResult_t e = r.CommAbort ? r.CommAbort((ncclComm_t)comm) : r.CommDestroy((ncclComm_t)comm);
  if (e != ncclSuccess) return rccl_fail("ncclCommAbort", e);
  return L2Q_OK;
}

int l2q_comm_version(void) {
  Rccl& r = rccl();
  return r.ok ? r.version : L2Q_EHIP;
}

}  // extern "C"
