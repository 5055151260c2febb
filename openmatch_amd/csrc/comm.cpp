// RCCL entry points of the multi-GPU path: one communicator per process / GPU over xGMI,
// created from a unique id that the host layer distributes through its own rendezvous (openmatch_amd uses the
// torch.distributed store the launcher already set up).  Replaces, behind the C ABI,
//   * DRModel.dist_gather_tensor           (modeling/dense_retrieval_model.py:247-258)   -> om_allgather_rows
//   * DistributedDataParallel's gradient all-reduce under HF Trainer (trainer/dense_trainer.py) -> om_allreduce_grads
//   * faiss-GPU's shard merge traffic      (retriever/dense_retriever.py:43-58)          -> om_exchange_topk
// RCCL is bound at run time (dlopen, preferring a copy that is already loaded -- torch ships one): the library has
// no link-time dependency on it and single-GPU use never touches it.  Every call is asynchronous on the caller's stream.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {
struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int*);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  const char* (*GetErrorString)(ncclResult_t);
  bool ok = false;
  std::string why;
};
Api g_api;
std::once_flag g_once;

void bind() {
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;            // a copy that is already in the process (torch's)
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { g_api.why = std::string("RCCL not found: ") + dlerror(); return; }
#define OM_SYM(F)                                                                    \
  g_api.F = (decltype(g_api.F))dlsym(h, "nccl" #F);                                  \
  if (!g_api.F) { g_api.why = "RCCL symbol nccl" #F " missing"; return; }
  OM_SYM(GetUniqueId) OM_SYM(CommInitRank) OM_SYM(CommDestroy) OM_SYM(CommCount) OM_SYM(AllGather) OM_SYM(AllReduce) OM_SYM(Send)
  OM_SYM(Recv) OM_SYM(GroupStart) OM_SYM(GroupEnd) OM_SYM(GetErrorString)
#undef OM_SYM
  g_api.ok = true;
}
int api() {
  std::call_once(g_once, bind);
  if (!g_api.ok) { om_set_error("om_comm: " + g_api.why); return 1; }
  return 0;
}
}  // namespace
#define OM_RCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) { om_set_error(std::string(__func__) + ": " #expr " -> " + g_api.GetErrorString(r_)); return 1; } \
  } while (0)

extern "C" int om_comm_unique_id(void* id128) {
  if (!id128) OM_FAIL("null argument");
  if (api()) return 1;
  static_assert(sizeof(ncclUniqueId) == OM_COMM_ID_BYTES, "unique id size");
  OM_RCCL(g_api.GetUniqueId((ncclUniqueId*)id128));
  return 0;
}

extern "C" int om_comm_init(const void* id128, int world, int rank, void** comm) {
  if (!id128 || !comm) OM_FAIL("null argument");
  if (world < 1 || rank < 0 || rank >= world) OM_FAIL("bad world / rank");
  if (api()) return 1;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t c = nullptr;
  OM_RCCL(g_api.CommInitRank(&c, world, id, rank));      // binds the CURRENT hip device of the calling thread
  *comm = (void*)c;
  return 0;
}

extern "C" int om_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (api()) return 1;
  OM_RCCL(g_api.CommDestroy((ncclComm_t)comm));
  return 0;
}

// number of ranks of the communicator, as RCCL itself reports it (ncclCommCount)
extern "C" int om_comm_count(void* comm, int* count) {
  if (!comm || !count) OM_FAIL("null argument");
  if (api()) return 1;
  OM_RCCL(g_api.CommCount((ncclComm_t)comm, count));
  return 0;
}

// recv[w * rows : (w+1) * rows] = rank w's `send` (rank-major row order, as the reference's cat of all_gather)
extern "C" int om_allgather_rows(void* comm, const void* send, void* recv, int64_t rows, int64_t row_bytes, void* stream) {
  if (!comm || !send || !recv) OM_FAIL("null argument");
  if (rows < 0 || row_bytes <= 0) OM_FAIL("bad shape");
  if (rows == 0) return 0;
  if (api()) return 1;
  OM_RCCL(g_api.AllGather(send, recv, (size_t)(rows * row_bytes), ncclUint8, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

// in place over n f32 values: sum, or mean when average != 0 (what DDP leaves in param.grad)
extern "C" int om_allreduce_grads(void* comm, float* buf, int64_t n, int average, void* stream) {
  if (!comm || !buf) OM_FAIL("null argument");
  if (n <= 0) return 0;
  if (api()) return 1;
  OM_RCCL(g_api.AllReduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

// Candidates by query range: D / I hold this shard's [world][q_block][k] results for ALL queries (padded to whole blocks);
// block w travels to rank w, which receives recvD / recvI [world][q_block][k] = every shard's candidates for ITS block.
extern "C" int om_exchange_topk(void* comm, int world, const float* D, const int64_t* I, int64_t q_block, int k,
                                float* recvD, int64_t* recvI, void* stream) {
  if (!comm || !D || !I || !recvD || !recvI) OM_FAIL("null argument");
  if (world < 1 || q_block <= 0 || k <= 0) OM_FAIL("bad shape");
  if (api()) return 1;
  const size_t n = (size_t)q_block * k;
  hipStream_t s = (hipStream_t)stream;
  OM_RCCL(g_api.GroupStart());
  for (int w = 0; w < world; ++w) {
    OM_RCCL(g_api.Send(D + w * n, n, ncclFloat32, w, (ncclComm_t)comm, s));
    OM_RCCL(g_api.Recv(recvD + w * n, n, ncclFloat32, w, (ncclComm_t)comm, s));
    OM_RCCL(g_api.Send(I + w * n, n, ncclInt64, w, (ncclComm_t)comm, s));
    OM_RCCL(g_api.Recv(recvI + w * n, n, ncclInt64, w, (ncclComm_t)comm, s));
  }
  OM_RCCL(g_api.GroupEnd());
  return 0;
}
