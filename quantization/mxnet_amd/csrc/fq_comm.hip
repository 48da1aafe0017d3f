// libfakequant — the (tiny) collectives of multi-GPU calibration over RCCL, for integrators without torch.distributed
// (see fq_common.h for the list of translation units and the design rules)
//
// one process per GPU; the only exchanges of the path are the all-reduce of L + 1 doubles per naive-EMA
// calibration step, the all-reduce(max / sum) of the KL ranges / histograms and the all-reduce of the evaluation counters -
// all of them <= 434 KB, latency-bound, on the compute stream.  quantization/mxnet_amd/dist.py issues them through
// torch.distributed (whose "nccl" backend IS RCCL); an MXNet integrator has no torch, so the same collectives are exported
// here.  librccl.so is bound at run time (dlopen: first the copy already mapped into the process - torch bundles one - then
// the system's), so the library has no link-time dependency on it and loads on machines without RCCL.
#include "fq_common.h"

#include <dlfcn.h>

namespace {

using namespace fqi;

constexpr int kUniqueIdBytes = 128;                       // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;                                       // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclInt64 = 4, kNcclFloat32 = 7, kNcclFloat64 = 8;   // rccl.h

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;
Comm g_comm = nullptr;
int g_world = 0;
std::mutex g_comm_mu;

int bind_rccl() {
  if (g_rccl.handle != nullptr) return FQ_OK;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names)
    if (h == nullptr) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already uses (torch's)
  for (const char* nm : names)
    if (h == nullptr) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return fail(FQ_ERR_INVALID, "fq_comm: librccl.so cannot be loaded (%s)", dlerror());
  Rccl r;
  r.handle = h;
  r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce)
    return fail(FQ_ERR_INVALID, "fq_comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
  g_rccl = r;
  return FQ_OK;
}

int rccl_fail(const char* what, int rc) {
  return fail(FQ_ERR_HIP, "%s: %s (RCCL result %d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
}

int all_reduce(void* buf, int64_t count, int op, int dtype, fqStream_t stream, const char* who) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  FQ_REQUIRE(g_comm != nullptr, "%s: call fq_comm_init first", who);
  FQ_REQUIRE(buf != nullptr && count >= 0, "%s: null buffer or negative count", who);
  FQ_REQUIRE(op == FQ_COMM_SUM || op == FQ_COMM_MAX, "%s: op must be FQ_COMM_SUM or FQ_COMM_MAX", who);
  if (count == 0) return FQ_OK;
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, dtype, op == FQ_COMM_SUM ? kNcclSum : kNcclMax, g_comm,
                                   (hipStream_t)stream);
  if (rc != 0) return rccl_fail(who, rc);
  return FQ_OK;
}

}  // namespace

extern "C" {

int fq_comm_unique_id(void* id128) {
  FQ_REQUIRE(id128 != nullptr, "fq_comm_unique_id: null pointer");
  if (int rc = bind_rccl()) return rc;
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) return rccl_fail("fq_comm_unique_id", rc);
  memcpy(id128, id.internal, kUniqueIdBytes);
  return FQ_OK;
}

int fq_comm_init(int rank, int world, const void* id128) {
  FQ_REQUIRE(id128 != nullptr, "fq_comm_init: null unique id");
  FQ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "fq_comm_init: rank %d of %d", rank, world);
  if (int rc = bind_rccl()) return rc;
  std::lock_guard<std::mutex> lk(g_comm_mu);
  FQ_REQUIRE(g_comm == nullptr, "fq_comm_init: a communicator already exists (fq_comm_destroy first)");
  UniqueId id;
  memcpy(id.internal, id128, kUniqueIdBytes);
  Comm c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, world, id, rank);       // on the device that is current for this thread
  if (rc != 0) return rccl_fail("fq_comm_init", rc);
  g_comm = c;
  g_world = world;
  return FQ_OK;
}

int fq_comm_world(void) { return g_world; }

int fq_allreduce_f32(float* buf, int64_t count, int op, fqStream_t stream) {
  return all_reduce(buf, count, op, kNcclFloat32, stream, "fq_allreduce_f32");
}

int fq_allreduce_f64(double* buf, int64_t count, int op, fqStream_t stream) {
  return all_reduce(buf, count, op, kNcclFloat64, stream, "fq_allreduce_f64");
}

int fq_allreduce_i64(int64_t* buf, int64_t count, int op, fqStream_t stream) {
  return all_reduce(buf, count, op, kNcclInt64, stream, "fq_allreduce_i64");
}

int fq_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (g_comm == nullptr) return FQ_OK;
  const int rc = g_rccl.comm_destroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  if (rc != 0) return rccl_fail("fq_comm_destroy", rc);
  return FQ_OK;
}

}  // extern "C"
