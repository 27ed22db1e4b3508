// comm.hip -- gradient all-reduce over RCCL (include/osdiff.h: osd_comm_*, osd_allreduce_grads_begin/end).
//
// The slot it fills in the reference is between loss.backward() and clip_grad_norm_ (utils/train.py:239-244): the
// reference is single-process; data parallel is this build's addition.  One communicator per
// process (= per GPU); bucket b of the flat gradient is reduced on the communicator's own stream behind the event
// osd_train_loss_fwd_bwd recorded when that bucket became final, so the collective overlaps the rest of backward.
//
// RCCL is bound at run time (dlopen of the librccl the process already has -- PyTorch ships its own -- else the
// ROCm one): libosdiff.so has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>
#include <new>
#include <rccl/rccl.h>
#include "handle.h"

namespace osd {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }     // already in the process?
  if (!lib) for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
  if (!lib) { set_error("RCCL not found (librccl.so): %s", dlerror()); return nullptr; }
  auto sym = [&](const char* s) { return dlsym(lib, s); };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
    set_error("librccl.so lacks an expected symbol");
    return nullptr;
  }
  api.lib = lib;
  return &api;
}

#define OSD_NCCL(api, call)                                                                   \
  do {                                                                                        \
    ncclResult_t r__ = (call);                                                                \
    if (r__ != ncclSuccess) {                                                                 \
      ::osd::set_error("%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r__), __FILE__, __LINE__); \
      return OSD_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

}  // namespace osd

struct osd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;      // the collectives' own stream
  hipEvent_t done = nullptr;
};

using namespace osd;

extern "C" {

int osd_comm_unique_id(void* id_out128) {
  if (!id_out128) { set_error("null argument"); return OSD_EINVAL; }
  RcclApi* api = rccl();
  if (!api) return OSD_EUNSUPPORTED;
  ncclUniqueId id;
  OSD_NCCL(api, api->GetUniqueId(&id));
  static_assert(sizeof(id) == OSD_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out128, &id, sizeof(id));
  return OSD_OK;
}

int osd_comm_destroy(osd_comm* c) {
  if (!c) return OSD_OK;
  hipError_t e = hipSetDevice(c->device);
  if (c->stream) e = hipStreamSynchronize(c->stream);
  RcclApi* api = rccl();
  if (c->comm && api) api->CommDestroy(c->comm);
  if (c->done) e = hipEventDestroy(c->done);
  if (c->stream) e = hipStreamDestroy(c->stream);
  (void)e;
  delete c;
  return OSD_OK;
}

int osd_comm_create(const void* id128, int rank, int world, int device, osd_comm** out) {
  if (!id128 || !out) { set_error("null argument"); return OSD_EINVAL; }
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) { set_error("rank %d outside [0,%d)", rank, world); return OSD_EINVAL; }
  RcclApi* api = rccl();
  if (!api) return OSD_EUNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  osd_comm* c = new (std::nothrow) osd_comm();
  if (!c) { set_error("out of host memory"); return OSD_ENOMEM; }
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  int rc = OSD_OK;
  // default priority: a non-default stream priority upsets the hardware-queue assignment of this stack (measured with the torch
  // binding on the 2-rank rehearsal: 6.3 -> 94-311 ms per step with a highest-priority comm stream; the low-priority side stream
  // of train.hip slowed unrelated, later-created streams 3.3x) -- DESIGN.md section 7
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) { set_error("stream/event creation failed"); rc = OSD_EHIP; }
  if (rc == OSD_OK) {
    const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", api->GetErrorString(r)); c->comm = nullptr; rc = OSD_EHIP; }
  }
  if (rc != OSD_OK) { osd_comm_destroy(c); return rc; }
  *out = c;
  return OSD_OK;
}

int osd_allreduce_grads_begin(osd_handle* h, osd_comm* c, float* flat_grad, const int64_t* start, const int64_t* end, void* const* events,
                              int n_buckets) {
  if (!h || !c || !flat_grad || !start || !end || n_buckets < 1) { set_error("bad argument"); return OSD_EINVAL; }
  if (c->device != h->cfg.device) { set_error("communicator is on device %d, handle on %d", c->device, h->cfg.device); return OSD_EINVAL; }
  for (int b = 0; b < n_buckets; ++b)
    if (start[b] < 0 || end[b] < start[b]) { set_error("bucket %d: bad range", b); return OSD_EINVAL; }
  RcclApi* api = rccl();
  if (!api) return OSD_EUNSUPPORTED;
  OSD_HIP(hipSetDevice(c->device));
  if (!events) {                     // no per-bucket events: order the whole message behind the handle's stream
    OSD_HIP(hipEventRecord(c->done, h->stream));
    OSD_HIP(hipStreamWaitEvent(c->stream, c->done, 0));
  }
  for (int b = 0; b < n_buckets; ++b) {
    if (events) OSD_HIP(hipStreamWaitEvent(c->stream, (hipEvent_t)events[b], 0));
    if (end[b] == start[b]) continue;
    // SUM of gradients pre-scaled by 1/world in the loss (loss_scale) = the data-parallel mean
    OSD_NCCL(api, api->AllReduce(flat_grad + start[b], flat_grad + start[b], (size_t)(end[b] - start[b]), ncclFloat32, ncclSum, c->comm, c->stream));
  }
  return OSD_OK;
}

int osd_allreduce_grads_end(osd_handle* h, osd_comm* c) {
  if (!h || !c) { set_error("null argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(c->device));
  OSD_HIP(hipEventRecord(c->done, c->stream));
  OSD_HIP(hipStreamWaitEvent(h->stream, c->done, 0));      // clip + AdamW on the handle's stream see reduced gradients
  return OSD_OK;
}

}  // extern "C"
