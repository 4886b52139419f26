// dm_comm.hip — libdriftcomm: gather / all-reduce of doubles over RCCL (see include/driftcomm.h).
// A separate shared library on purpose: libdriftmi.so carries no RCCL dependency (the Python layer talks through
// torch.distributed, whose "nccl" backend is torch's own RCCL build); a host without torch links this one.
#include "../../include/driftcomm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

struct dm_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int rank = 0, size = 1, device = 0;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;  // ordering against the caller's stream (see order_in / order_out)
};

static thread_local std::string g_err;

static int fail(const char* what, const char* detail) {
  g_err = std::string(what) + ": " + detail;
  return -1;
}

#define COMM_NCCL(call)                                                 \
  do {                                                                  \
    ncclResult_t r__ = (call);                                          \
    if (r__ != ncclSuccess) return fail(#call, ncclGetErrorString(r__)); \
  } while (0)
#define COMM_HIP(call)                                                 \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) return fail(#call, hipGetErrorString(e__)); \
  } while (0)

// The collective runs on the communicator's stream; the buffer was produced on, and will be consumed from, the caller's
// stream `user` (NULL = the legacy default stream).  Before: the communicator's stream waits for everything enqueued on
// `user` so far.  After: `user` waits for the collective.  Nothing blocks the host.
static int order_in(dm_comm* c, hipStream_t user) {
  if (user == c->stream) return 0;
  COMM_HIP(hipEventRecord(c->ev_in, user));
  COMM_HIP(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  return 0;
}
static int order_out(dm_comm* c, hipStream_t user) {
  if (user == c->stream) return 0;
  COMM_HIP(hipEventRecord(c->ev_out, c->stream));
  COMM_HIP(hipStreamWaitEvent(user, c->ev_out, 0));
  return 0;
}

extern "C" {

const char* dm_comm_last_error(void) { return g_err.c_str(); }

int dm_comm_unique_id(void* id_out) {
  if (!id_out) return fail("dm_comm_unique_id", "null argument");
  static_assert(sizeof(ncclUniqueId) == DM_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  COMM_NCCL(ncclGetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int dm_comm_init_rank(int nranks, int rank, const void* id, int device, void* stream, dm_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail("dm_comm_init_rank", "bad argument");
  COMM_HIP(hipSetDevice(device));
  dm_comm* c = new dm_comm();
  c->rank = rank;
  c->size = nranks;
  c->device = device;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail("hipStreamCreate", hipGetErrorString(e)); }
    c->own_stream = true;
  }
  if (hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return fail("hipEventCreate", "cannot create the ordering events");
  }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return fail("ncclCommInitRank", ncclGetErrorString(r));
  }
  *out = c;
  return 0;
}

int dm_comm_destroy(dm_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int dm_comm_rank(const dm_comm* c) { return c ? c->rank : -1; }
int dm_comm_size(const dm_comm* c) { return c ? c->size : -1; }

int dm_allreduce_f64(dm_comm* c, double* data_dev, size_t n, void* user_stream) {
  if (!c || (!data_dev && n)) return fail("dm_allreduce_f64", "bad argument");
  if (n == 0) return 0;
  hipStream_t user = reinterpret_cast<hipStream_t>(user_stream);
  COMM_HIP(hipSetDevice(c->device));
  if (order_in(c, user) != 0) return -1;
  COMM_NCCL(ncclAllReduce(data_dev, data_dev, n, ncclDouble, ncclSum, c->comm, c->stream));
  return order_out(c, user);
}

int dm_gather_f64(dm_comm* c, const double* send_dev, double* recv_dev, size_t n, int root, void* user_stream) {
  if (!c || root < 0 || root >= c->size || (!send_dev && n) || (c->rank == root && !recv_dev && n))
    return fail("dm_gather_f64", "bad argument");
  if (n == 0) return 0;
  hipStream_t user = reinterpret_cast<hipStream_t>(user_stream);
  COMM_HIP(hipSetDevice(c->device));
  if (order_in(c, user) != 0) return -1;
  COMM_NCCL(ncclGather(send_dev, recv_dev, n, ncclDouble, root, c->comm, c->stream));
  return order_out(c, user);
}

int dm_comm_sync(dm_comm* c) {
  if (!c) return fail("dm_comm_sync", "null communicator");
  COMM_HIP(hipSetDevice(c->device));
  COMM_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
