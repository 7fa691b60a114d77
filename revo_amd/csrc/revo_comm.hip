// revo_comm.hip -- the batched mode's ONE collective behind the C ABI (include/revo_hip.h: revo_comm_*).
//
// SURVEY 8(e) / north_star: independent frame-pairs shard across the GPUs of a node, one process per GPU, and the only
// exchange is a final gather of the 96-byte pair records over RCCL / xGMI.  Up to round 5 that all-gather lived in
// Python (torch.distributed); a C++ host -- the reference is C++ (main.cpp:22-47) -- could not reach the multi-GPU path
// through this header.  Here: a communicator handle (ncclGetUniqueId on rank 0 -> the caller ships 128 bytes to the
// other ranks by whatever it has: MPI, a file, a socket, torch.distributed's store -> ncclCommInitRank on the context's
// device) and one call that enqueues the all-gather of n records per rank on a stream; revo_pipeline_set_comm
// (revo_pipeline.hip) makes the pipeline handle enqueue it in its after-grid slot itself.
//
// RCCL is loaded at RUN TIME (dlopen), not linked: the single-GPU drop-in must load on a box without RCCL, and inside a
// process that already carries an RCCL (PyTorch bundles its own librccl.so) the library joins THAT copy instead of
// bringing a second one into the address space.  Order: $REVO_RCCL_LIB, an already loaded librccl.so / librccl.so.1,
// then librccl.so.1 / librccl.so from the loader's path, then /opt/rocm/lib.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/revo_hip.h"

extern "C" void revo_ctx_retain_(revo_ctx*);
extern "C" void revo_ctx_release_(revo_ctx*);
extern "C" int revo_ctx_device_(const revo_ctx*);
extern "C" void revo_set_error_(const char* msg);

namespace {

int fail(int code, const std::string& msg) {
  revo_set_error_(msg.c_str());
  return code;
}

// the slice of rccl.h this file needs (rccl.h:40-43,187,220,260,339,678): opaque communicator, by-value 128-byte id
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[REVO_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;  // ncclSuccess = 0
enum { ncclInt8 = 0 };     // ncclDataType_t: ncclInt8 = ncclChar = 0

struct Rccl {
  void* handle = nullptr;
  std::string path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string why;  // why loading failed
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    struct Try { const char* name; int flags; };
    const char* env = getenv("REVO_RCCL_LIB");
    const Try tries[] = {
        {env && *env ? env : nullptr, RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_NOLOAD},     // the copy the process already carries (PyTorch's)
        {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
    };
    for (const Try& t : tries) {
      if (!t.name) continue;
      void* h = dlopen(t.name, t.flags);
      if (!h) continue;
      r.handle = h;
      r.path = t.name;
      break;
    }
    if (!r.handle) { r.why = "no RCCL library could be loaded (librccl.so.1; set REVO_RCCL_LIB)"; return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
      r.why = "the RCCL library " + r.path + " lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      r.handle = nullptr;
    }
  });
  return &r;
}

int nccl_fail(const char* what, ncclResult_t e) {
  Rccl* r = rccl();
  return fail(REVO_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error") + " (" + std::to_string(e) + ")");
}

}  // namespace

struct revo_comm {
  revo_ctx* ctx = nullptr;
  int device = 0, world = 1, rank = 0;
  ncclComm_t comm = nullptr;
};

extern "C" int revo_comm_available(char* path_out, size_t cap, int* version_out) {
  Rccl* r = rccl();
  if (!r->handle) return fail(REVO_ERR_HIP, r->why);
  if (path_out && cap) { strncpy(path_out, r->path.c_str(), cap - 1); path_out[cap - 1] = 0; }
  if (version_out) { int v = 0; if (r->GetVersion) (void)r->GetVersion(&v); *version_out = v; }
  return REVO_OK;
}

extern "C" int revo_comm_unique_id(uint8_t id[REVO_COMM_ID_BYTES]) {
  if (!id) return fail(REVO_ERR_INVALID_ARG, "null argument");
  Rccl* r = rccl();
  if (!r->handle) return fail(REVO_ERR_HIP, r->why);
  ncclUniqueId u;
  const ncclResult_t e = r->GetUniqueId(&u);
  if (e) return nccl_fail("ncclGetUniqueId", e);
  memcpy(id, u.internal, REVO_COMM_ID_BYTES);
  return REVO_OK;
}

extern "C" int revo_comm_create(revo_ctx* ctx, const uint8_t id[REVO_COMM_ID_BYTES], int world_size, int rank, revo_comm** out) {
  if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  Rccl* r = rccl();
  if (!r->handle) return fail(REVO_ERR_HIP, r->why);
  const int device = revo_ctx_device_(ctx);
  if (hipSetDevice(device) != hipSuccess) return fail(REVO_ERR_HIP, "hipSetDevice");
  ncclUniqueId u;
  memcpy(u.internal, id, REVO_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, world_size, u, rank);  // collective: every rank of the id calls it
  if (e) return nccl_fail("ncclCommInitRank", e);
  revo_comm* h = new revo_comm();
  h->ctx = ctx; h->device = device; h->world = world_size; h->rank = rank; h->comm = c;
  revo_ctx_retain_(ctx);
  *out = h;
  return REVO_OK;
}

extern "C" void revo_comm_destroy(revo_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  Rccl* r = rccl();
  if (c->comm && r->handle) (void)r->CommDestroy(c->comm);
  revo_ctx* ctx = c->ctx;
  delete c;
  revo_ctx_release_(ctx);
}

extern "C" int revo_comm_world(const revo_comm* c, int* world_size, int* rank) {
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null communicator");
  if (world_size) *world_size = c->world;
  if (rank) *rank = c->rank;
  return REVO_OK;
}

extern "C" int revo_comm_allgather_records(revo_comm* c, const revo_pair_result* d_send, revo_pair_result* d_recv, int n_records,
                                           void* stream) {
  if (!c || !d_send || !d_recv || n_records <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  Rccl* r = rccl();
  if (!r->handle) return fail(REVO_ERR_HIP, r->why);
  if (hipSetDevice(c->device) != hipSuccess) return fail(REVO_ERR_HIP, "hipSetDevice");
  // records are plain bytes to the collective: nothing is reduced (96 B x n per rank: latency-bound, one ring step per peer)
  const ncclResult_t e = r->AllGather(d_send, d_recv, sizeof(revo_pair_result) * (size_t)n_records, ncclInt8, c->comm, (hipStream_t)stream);
  if (e) return nccl_fail("ncclAllGather", e);
  return REVO_OK;
}
