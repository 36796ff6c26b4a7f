// Data-parallel gradient exchange over RCCL / xGMI, issued by the library itself (frozen_ppo.py:586-603, 624-637).
//
// One process per GPU.  A communicator object owns the RCCL communicator of this rank, a communication stream and the
// events that fence it against the compute stream: the two bucket all-reduces of an optimizer step are enqueued from
// inside igi_teacher_update_dp_rccl -- no Python callback, no host synchronisation, the host only enqueues.
//   compute stream : phase 0 | record e0 | phase 1 ..................... | wait eD | all-reduce(late bucket) | clip + Adam
//   comm stream    :           wait e0 | all-reduce(early bucket: 2 ranges) | record eD
// The 1/world of the reference's "grads / rank_size" is folded into the Adam kernel (grad_scale).  At 1.4 MB + 0.3 MB the
// collectives are latency-bound on xGMI; what the schedule buys is that the large one runs under the ~80 us of
// latent / env_mlp backward.  `overlap == 0` is the reference's serial schedule: one all-reduce of the whole flat
// gradient on the compute stream after backward.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include "../../include/igi_ppo.h"
#include "teacher.h"

struct igi_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;       // communication stream (non-blocking: does not synchronise with stream 0)
  hipEvent_t ev[2][3] = {};           // per step parity: [0] phase 0 done, [2] early bucket reduced ([1] spare)
  hipEvent_t aev[2] = {};             // igi_comm_all_reduce_async_f32: [0] payload ready (compute), [1] reduced (comm)
  int rank = 0, world = 1, device = 0;
  char err[192] = "";
};

namespace igi {

#define IGI_NCCL_TRY(c, expr)                                                          \
  do {                                                                                 \
    ncclResult_t _r = (expr);                                                          \
    if (_r != ncclSuccess) {                                                           \
      snprintf((c)->err, sizeof((c)->err), "%s: %s", #expr, ncclGetErrorString(_r));  \
      return IGI_E_COMM;                                                               \
    }                                                                                  \
  } while (0)

static int comm_unique_id(void* id128) {
  if (!id128) return IGI_E_BADARG;
  static_assert(sizeof(ncclUniqueId) == IGI_COMM_ID_BYTES, "igi_comm_unique_id hands out an ncclUniqueId");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return IGI_E_COMM;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

static int comm_destroy(igi_comm* c);

// why the last igi_comm_create of this thread failed (the handle does not exist yet when it does)
static thread_local char g_comm_create_err[192] = "";

static int comm_create(const void* id128, int rank, int world, igi_comm** out) {
  g_comm_create_err[0] = 0;
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return IGI_E_BADARG;
  *out = nullptr;
  igi_comm* c = new igi_comm();
  c->rank = rank; c->world = world;
  // any failure below releases what was built so far (comm_destroy takes partially built objects) and keeps the reason
  auto fail = [&](int code, const char* what, const char* detail) {
    snprintf(g_comm_create_err, sizeof(g_comm_create_err), "%s: %s", what, detail);
    comm_destroy(c);
    return code;
  };
  hipError_t he = hipGetDevice(&c->device);
  if (he != hipSuccess) return fail((int)he, "hipGetDevice", hipGetErrorString(he));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);   // binds to the current device
  if (r != ncclSuccess) { c->comm = nullptr; return fail(IGI_E_COMM, "ncclCommInitRank", ncclGetErrorString(r)); }
  he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (he != hipSuccess) { c->stream = nullptr; return fail((int)he, "hipStreamCreateWithFlags", hipGetErrorString(he)); }
  // The fences order two streams of ONE device (compute -> communication -> compute).  On a one-rank communicator a
  // device-scope release is enough and saves 0.15 ms per update (measured, round 3).  With peers the early bucket's RCCL
  // kernels hand gradient bytes to other GPUs over xGMI right behind that event, and no N > 1 run has shown device
  // scope to be safe there: system-scope events are the default for world > 1.  IGI_EVENT_SYSFENCE=0 / 1 overrides.
  bool sysfence = world > 1;
  { const char* e = getenv("IGI_EVENT_SYSFENCE"); if (e) sysfence = atoi(e) != 0; }
  const unsigned evflags = hipEventDisableTiming | (sysfence ? 0u : (unsigned)hipEventDisableSystemFence);
  for (int q = 0; q < 2; ++q)
    for (int e = 0; e < 3; ++e) {
      he = hipEventCreateWithFlags(&c->ev[q][e], evflags);
      if (he != hipSuccess) { c->ev[q][e] = nullptr; return fail((int)he, "hipEventCreateWithFlags", hipGetErrorString(he)); }
    }
  for (int e = 0; e < 2; ++e) {
    he = hipEventCreateWithFlags(&c->aev[e], evflags);
    if (he != hipSuccess) { c->aev[e] = nullptr; return fail((int)he, "hipEventCreateWithFlags", hipGetErrorString(he)); }
  }
  *out = c;
  return 0;
}

static int comm_destroy(igi_comm* c) {
  if (!c) return 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int e = 0; e < 2; ++e)
    if (c->aev[e]) (void)hipEventDestroy(c->aev[e]);
  for (int q = 0; q < 2; ++q)
    for (int e = 0; e < 3; ++e)
      if (c->ev[q][e]) (void)hipEventDestroy(c->ev[q][e]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return 0;
}

// in place, in the order of `stream` (RCCL enqueues its kernels there)
static int comm_all_reduce_sum(igi_comm* c, float* buf, long long n, hipStream_t s) {
  if (!c || !buf || n < 0) return IGI_E_BADARG;
  if (n == 0) return 0;
  IGI_NCCL_TRY(c, ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
  return 0;
}

// The same reduction on the communicator's OWN stream, fenced against `compute`: everything enqueued on `compute` so far
// is visible to it (event), the host does not block, and work enqueued on `compute` afterwards runs concurrently with
// the collective -- the student's early gradient bucket under the rest of its backward (ext_adapt.py:833-851 reduces
// everything after backward).  comm_join makes `compute` wait for the last such reduction.  One in flight at a time.
static int comm_all_reduce_async(igi_comm* c, float* buf, long long n, hipStream_t compute) {
  if (!c || !buf || n < 0) return IGI_E_BADARG;
  IGI_HIP_TRY(hipEventRecord(c->aev[0], compute));
  IGI_HIP_TRY(hipStreamWaitEvent(c->stream, c->aev[0], 0));
  if (n > 0) IGI_NCCL_TRY(c, ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, c->stream));
  IGI_HIP_TRY(hipEventRecord(c->aev[1], c->stream));
  return 0;
}
static int comm_join(igi_comm* c, hipStream_t compute) {
  if (!c) return IGI_E_BADARG;
  IGI_HIP_TRY(hipStreamWaitEvent(compute, c->aev[1], 0));
  return 0;
}

static int comm_broadcast(igi_comm* c, void* buf, long long bytes, int root, hipStream_t s) {
  if (!c || !buf || bytes < 0 || root < 0 || root >= c->world) return IGI_E_BADARG;
  if (bytes == 0) return 0;
  IGI_NCCL_TRY(c, ncclBroadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, s));
  return 0;
}

// the (up to) two ranges of a bucket as ONE RCCL group: one launch on the communication stream
static int comm_reduce_ranges(igi_comm* c, float* grads, const long long* off, const long long* len, int n, hipStream_t s) {
  IGI_NCCL_TRY(c, ncclGroupStart());
  for (int i = 0; i < n; ++i)
    if (len[i] > 0)
      IGI_NCCL_TRY(c, ncclAllReduce(grads + off[i], grads + off[i], (size_t)len[i], ncclFloat32, ncclSum, c->comm, s));
  IGI_NCCL_TRY(c, ncclGroupEnd());
  return 0;
}

// The whole data-parallel update as one host call with the gradient exchange issued natively (see the header of this
// file).  stats_sum (optional, E * n_mb * IGI_STATS_PER_STEP floats): the per-step statistics summed over the ranks
// (the KL all-reduce of frozen_ppo.py:624-627 and the loss aggregation of :387-396 ride here, once per update, on the
// communication stream behind the last step's collectives); the per-rank values stay in st->stats.
static int teacher_update_dp_rccl(const igi_teacher_cfg* c, const igi_rollout* ro, const igi_teacher_state* st,
                                  int64_t adam_t0, igi_comm* cm, int overlap, float* stats_sum, hipStream_t s) {
  if (!cm || !cm->comm) return IGI_E_BADARG;
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!st->grads) return IGI_E_BADARG;
  const float scale = 1.0f / (float)cm->world;
  const GradBuckets gb = grad_buckets(p);
  const int total = p.E * p.nmb;
  int slot = 0;
  for (int e = 0; e < p.E; ++e) {
    for (int i = 0; i < p.nmb; ++i, ++slot) {
      const bool skip_gather = slot > 0;   // the previous step's fused tail gathered this minibatch
      if (overlap) {
        hipEvent_t* ev = cm->ev[slot & 1];
        if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, 0, skip_gather))) return rc;
        IGI_HIP_TRY(hipEventRecord(ev[0], s));
        IGI_HIP_TRY(hipStreamWaitEvent(cm->stream, ev[0], 0));
        if ((rc = comm_reduce_ranges(cm, st->grads, gb.off, gb.len, 2, cm->stream))) return rc;
        IGI_HIP_TRY(hipEventRecord(ev[2], cm->stream));
        if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, 1))) return rc;
        // the late bucket is on the critical path whatever stream carries it: it goes out on the compute stream (no
        // cross-stream hop behind phase 1; measured on a one-rank communicator: both buckets on the communication
        // stream cost 34 us per step of event hand-offs).  The compute stream joins the early bucket -- which finished
        // under phase 1 -- FIRST: two collectives of one communicator are then never in flight on two streams at once
        // (RCCL serialises a communicator's launches internally; the explicit order does not rely on it)
        IGI_HIP_TRY(hipStreamWaitEvent(s, ev[2], 0));
        if ((rc = comm_reduce_ranges(cm, st->grads, gb.off + 2, gb.len + 2, 2, s))) return rc;
      } else {
        if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, -1, skip_gather))) return rc;
        if ((rc = comm_all_reduce_sum(cm, st->grads, p.P, s))) return rc;
      }
      const bool more = slot + 1 < total;
      if ((rc = teacher_apply(c, st, slot, adam_t0 + slot + 1, scale, s, more ? ro : nullptr, (slot + 1) % p.nmb,
                              slot + 1)))
        return rc;
    }
  }
  if (stats_sum && st->stats) {
    const size_t n = (size_t)total * IGI_STATS_PER_STEP;
    IGI_HIP_TRY(hipMemcpyAsync(stats_sum, st->stats, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if ((rc = comm_all_reduce_sum(cm, stats_sum, (long long)n, s))) return rc;
  }
  return 0;
}

}  // namespace igi
