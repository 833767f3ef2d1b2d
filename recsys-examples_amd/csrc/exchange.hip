// The exchanges of the row-wise sharded lookup INSIDE the library: one C call per stage of a sharded step, RCCL driven directly
// on the caller's HIP streams.
//
// What a stage was before (dynamicemb/input_dist.py, sharded.py): several c10d calls -- each one an argument check, a Work
// object, an event pair between the caller's stream and the process group's internal stream -- with torch allocations and
// Python in between: 0.30 ms per forced-W=1 step against 0.12 ms unsharded, all of the difference on the host
// (profiles/r04_sharded_w1_bench.json).  Here a stage is a fixed launch sequence on ONE stream the caller names:
//
//   input dist, first half   bucketize -> all-to-all of the bag lengths -> offsets of the received stream -> per-peer key
//                            counts into pinned memory (the one host read of the exact exchange, KJTAllToAll's)
//   input dist, second half  the host reads the counts (by then long written), the caller allocates the receive buffer,
//                            all-to-all-v of the keys -> recat (src, f, b) -> (f, src, b)
//   output dist (pooled)     all-to-all of the W [B, total_D] blocks of partial sums -> their sum in the output type
//   backward (pooled)        all-gather of the output gradients
//   sequence                 all-to-all-v of rows / row gradients by the key counts of the input dist
//
// Restates the collectives of (reference) corelib/dynamicemb/dynamicemb/input_dist.py:199-285 (RwSparseFeaturesDist ->
// TorchRec KJTAllToAll) and planner/rw_sharding.py:85-158, 191-261 (the output dists); the all-to-alls are grouped
// ncclSend / ncclRecv pairs, one per peer: xGMI is a full mesh, every pair of GPUs has its own link.
//
// RCCL is NOT a link-time dependency: the process already holds torch's librccl.so (torch.distributed's "nccl" backend IS
// RCCL); mi355_rw_load_rccl() takes its path and binds the eight entry points used here.  Two communicators per handle --
// one for the input dist (it runs on the caller's exchange stream, under the previous batch's compute), one for the output /
// gradient exchanges on the compute stream: operations of ONE communicator are serialised in issue order whatever stream they
// are on, which would tie the two streams together.
#include "common.h"
#include "../../include/recsys_amd.h"
#include <dlfcn.h>
#include <cstring>
#include <string>

namespace {

struct NcclId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
constexpr int kNcclInt8 = 0;                      // ncclDataType_t: every exchange here moves bytes

struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(Comm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommAbort)(Comm) = nullptr;          // optional: the hang guard of the one-time self-check (mi355_rw_abort)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} g_rccl;

constexpr int kRing = 8;      // input dists in flight (a later one may start before an earlier one has been read)

struct RwExchange {
  Comm comm_in = nullptr, comm_out = nullptr;
  int world = 0, rank = 0;
  int64_t* splits[kRing] = {};      // pinned host [2 W]: keys sent to / received from every peer
  hipEvent_t ev_counts[kRing] = {}, ev_keys[kRing] = {};
  hipEvent_t ev_fork = nullptr;
  bool live[kRing] = {};            // ticket handed out by input_begin and not yet consumed by input_keys
  int next = 0;
};

void free_exchange(RwExchange* x, bool abort) {
  if (!x) return;
  if (x->comm_in) { if (abort && g_rccl.CommAbort) g_rccl.CommAbort(x->comm_in); else g_rccl.CommDestroy(x->comm_in); }
  if (x->comm_out) { if (abort && g_rccl.CommAbort) g_rccl.CommAbort(x->comm_out); else g_rccl.CommDestroy(x->comm_out); }
  x->comm_in = x->comm_out = nullptr;
  for (int i = 0; i < kRing; ++i) {
    if (x->splits[i]) (void)hipHostFree(x->splits[i]);
    if (x->ev_counts[i]) (void)hipEventDestroy(x->ev_counts[i]);
    if (x->ev_keys[i]) (void)hipEventDestroy(x->ev_keys[i]);
    x->splits[i] = nullptr; x->ev_counts[i] = x->ev_keys[i] = nullptr;
  }
  if (x->ev_fork) (void)hipEventDestroy(x->ev_fork);
  x->ev_fork = nullptr;
}

int nccl_fail(int rc, const char* what) {
  std::string m = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  mi355_set_error(m.c_str());
  return MI355_ELAUNCH;
}

#define RW_NCCL(call, what)                          \
  do {                                               \
    const int rc__ = (call);                         \
    if (rc__ != 0) return nccl_fail(rc__, what);     \
  } while (0)
#define RW_HIP(call)                                                                           \
  do {                                                                                         \
    const hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) { mi355_set_error(hipGetErrorString(e__)); return MI355_ELAUNCH; }  \
  } while (0)
#define RW_RC(call)                \
  do {                             \
    const int rc2__ = (call);      \
    if (rc2__ != 0) return rc2__;  \
  } while (0)

// all-to-all of byte ranges: peer p gets send[soff[p] .. + scnt[p]), its bytes land at recv[roff[p] .. + rcnt[p]).
// The rank's OWN range never goes through RCCL (round 6): a send / receive pair to oneself is a copy kernel plus ~15 us of host
// time per call -- a stream-ordered device copy instead; with one rank no RCCL call is made at all.
int all_to_all_bytes(Comm comm, int world, int rank, const char* send, const int64_t* soff, const int64_t* scnt, char* recv,
                     const int64_t* roff, const int64_t* rcnt, hipStream_t stream) {
  if (scnt[rank] > 0) RW_HIP(hipMemcpyAsync(recv + roff[rank], send + soff[rank], (size_t)scnt[rank], hipMemcpyDeviceToDevice, stream));
  if (world == 1) return 0;
  RW_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    RW_NCCL(g_rccl.Send(send + soff[p], (size_t)scnt[p], kNcclInt8, p, comm, stream), "ncclSend");
    RW_NCCL(g_rccl.Recv(recv + roff[p], (size_t)rcnt[p], kNcclInt8, p, comm, stream), "ncclRecv");
  }
  RW_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  return 0;
}

// skip_self: the rank's own block does not travel at all -- the caller reads it in place (else it is copied on the stream)
int all_to_all_equal(Comm comm, int world, int rank, const void* send, void* recv, int64_t bytes_per_peer, hipStream_t stream,
                     bool skip_self = false) {
  if (!skip_self && bytes_per_peer > 0)
    RW_HIP(hipMemcpyAsync((char*)recv + rank * bytes_per_peer, (const char*)send + rank * bytes_per_peer, (size_t)bytes_per_peer,
                          hipMemcpyDeviceToDevice, stream));
  if (world == 1) return 0;
  RW_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    RW_NCCL(g_rccl.Send((const char*)send + p * bytes_per_peer, (size_t)bytes_per_peer, kNcclInt8, p, comm, stream), "ncclSend");
    RW_NCCL(g_rccl.Recv((char*)recv + p * bytes_per_peer, (size_t)bytes_per_peer, kNcclInt8, p, comm, stream), "ncclRecv");
  }
  RW_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  return 0;
}

}  // namespace

extern "C" {

int mi355_rw_load_rccl(const char* path) {
  if (g_rccl.so) return 0;
  MI355_CHECK_ARG(path != nullptr, "path of librccl.so required");
  void* so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!so) { mi355_set_error(dlerror()); return MI355_EINVAL; }
  // a partial bind leaves nothing behind: the handle is closed and every pointer reset before the error goes out
#define RW_SYM(field, name)                                                                      \
  do {                                                                                           \
    *(void**)(&g_rccl.field) = dlsym(so, name);                                                  \
    if (!g_rccl.field) {                                                                         \
      g_rccl = Rccl();                                                                           \
      dlclose(so);                                                                               \
      mi355_set_error("librccl.so: symbol " name " missing");                                    \
      return MI355_EINVAL;                                                                       \
    }                                                                                            \
  } while (0)
  RW_SYM(GetUniqueId, "ncclGetUniqueId");
  RW_SYM(CommInitRank, "ncclCommInitRank");
  RW_SYM(CommDestroy, "ncclCommDestroy");
  RW_SYM(GroupStart, "ncclGroupStart");
  RW_SYM(GroupEnd, "ncclGroupEnd");
  RW_SYM(Send, "ncclSend");
  RW_SYM(Recv, "ncclRecv");
  RW_SYM(AllGather, "ncclAllGather");
  RW_SYM(GetErrorString, "ncclGetErrorString");
#undef RW_SYM
  *(void**)(&g_rccl.CommAbort) = dlsym(so, "ncclCommAbort");
  g_rccl.so = so;
  return 0;
}

int mi355_rw_unique_id(void* out, int64_t bytes) {
  MI355_CHECK_ARG(g_rccl.so, "mi355_rw_load_rccl first");
  MI355_CHECK_ARG(out && bytes >= (int64_t)sizeof(NcclId), "128 bytes required");
  RW_NCCL(g_rccl.GetUniqueId((NcclId*)out), "ncclGetUniqueId");
  return 0;
}

int mi355_rw_create(const void* id_in, const void* id_out, int world, int rank, void** handle) {
  MI355_CHECK_ARG(g_rccl.so, "mi355_rw_load_rccl first");
  MI355_CHECK_ARG(id_in && id_out && handle && world >= 1 && world <= 512 && rank >= 0 && rank < world, "bad arguments");
  RwExchange* x = new RwExchange();
  x->world = world; x->rank = rank;
  NcclId a, b;
  memcpy(&a, id_in, sizeof(a)); memcpy(&b, id_out, sizeof(b));
  int rc = g_rccl.CommInitRank(&x->comm_in, world, a, rank);
  if (rc == 0) rc = g_rccl.CommInitRank(&x->comm_out, world, b, rank);
  if (rc != 0) { free_exchange(x, false); delete x; return nccl_fail(rc, "ncclCommInitRank"); }
  hipError_t e = hipSuccess;
  for (int i = 0; i < kRing && e == hipSuccess; ++i) {
    e = hipHostMalloc((void**)&x->splits[i], 2 * sizeof(int64_t) * world, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_counts[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_keys[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_fork, hipEventDisableTiming);
  if (e != hipSuccess) {   // nothing of a half-built exchange survives (communicators included)
    free_exchange(x, false);
    delete x;
    mi355_set_error(hipGetErrorString(e));
    return MI355_ELAUNCH;
  }
  *handle = x;
  return 0;
}

int mi355_rw_destroy(void* handle) {
  RwExchange* x = (RwExchange*)handle;
  if (!x) return 0;
  (void)hipDeviceSynchronize();
  free_exchange(x, false);
  delete x;
  return 0;
}

// the hang guard: both communicators are aborted (a collective kernel stuck on a peer that never came returns), the handle is
// freed without a device synchronisation.  After this the caller uses the c10d sequence.
int mi355_rw_abort(void* handle) {
  RwExchange* x = (RwExchange*)handle;
  if (!x) return 0;
  free_exchange(x, true);
  delete x;
  return 0;
}

int mi355_rw_input_begin(void* handle, int64_t num_features, int64_t batch_size, const int64_t* offsets, const void* keys,
                         const int64_t* block_sizes, const int32_t* dist_types, int64_t* new_lengths, int64_t* new_offsets,
                         void* new_keys, int64_t* unbucketize_permute, int64_t* recv_lengths, int64_t* recv_offsets,
                         hipStream_t producer_stream, hipStream_t stream, int* ticket) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && ticket && num_features >= 1 && batch_size >= 0, "bad arguments");
  const int W = x->world;
  const int64_t FB = num_features * batch_size;
  const int slot = x->next % kRing;
  if (x->live[slot]) {     // kRing input dists in flight: the oldest ticket's pinned counts and events are still owed to its finish
    mi355_set_error("mi355_rw_input_begin: every ticket of the ring is in flight (finish the oldest input dist first)");
    return MI355_EINVAL;
  }
  if (producer_stream != stream) {      // the batch was produced on the caller's compute stream
    RW_HIP(hipEventRecord(x->ev_fork, producer_stream));
    RW_HIP(hipStreamWaitEvent(stream, x->ev_fork, 0));
  }
  RW_RC(mi355_block_bucketize(W, FB, batch_size, offsets, keys, block_sizes, dist_types, nullptr, new_lengths, new_offsets,
                              new_keys, nullptr, unbucketize_permute, stream));
  RW_RC(all_to_all_equal(x->comm_in, W, x->rank, new_lengths, recv_lengths, FB * (int64_t)sizeof(int64_t), stream));
  RW_RC(mi355_exclusive_offsets(recv_lengths, W * FB, recv_offsets, stream));
  RW_RC(mi355_peer_splits(new_offsets, recv_offsets, FB, W, x->splits[slot], stream));
  RW_HIP(hipEventRecord(x->ev_counts[slot], stream));
  x->live[slot] = true;
  x->next++;
  *ticket = slot;
  return 0;
}

// 1: the key counts of `ticket` have been written (mi355_rw_input_counts will not wait), 0: not yet
int mi355_rw_input_counts_ready(void* handle, int ticket) {
  RwExchange* x = (RwExchange*)handle;
  if (!x || ticket < 0 || ticket >= kRing) return 0;
  return hipEventQuery(x->ev_counts[ticket]) == hipSuccess ? 1 : 0;
}

int mi355_rw_input_counts(void* handle, int ticket, int64_t* send_splits, int64_t* recv_splits, int64_t* totals) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && ticket >= 0 && ticket < kRing && send_splits && recv_splits && totals, "bad arguments");
  RW_HIP(hipEventSynchronize(x->ev_counts[ticket]));
  int64_t ns = 0, nr = 0;
  for (int p = 0; p < x->world; ++p) {
    send_splits[p] = x->splits[ticket][p];
    recv_splits[p] = x->splits[ticket][x->world + p];
    ns += send_splits[p]; nr += recv_splits[p];
  }
  totals[0] = ns; totals[1] = nr;
  return 0;
}

int mi355_rw_input_keys(void* handle, int ticket, int64_t num_features, int64_t batch_size, const void* new_keys,
                        void* recv_keys, const int64_t* recv_lengths, const int64_t* recv_offsets, int64_t* fm_lengths,
                        int64_t* fm_offsets, void* fm_keys, hipStream_t stream, hipStream_t consumer_stream) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && ticket >= 0 && ticket < kRing, "bad arguments");
  const int W = x->world;
  int64_t soff[512], scnt[512], roff[512], rcnt[512];
  MI355_CHECK_ARG(W <= 512, "world size");
  int64_t s = 0, r = 0;
  for (int p = 0; p < W; ++p) {
    scnt[p] = x->splits[ticket][p] * 8; rcnt[p] = x->splits[ticket][W + p] * 8;
    soff[p] = s; roff[p] = r;
    s += scnt[p]; r += rcnt[p];
  }
  RW_RC(all_to_all_bytes(x->comm_in, W, x->rank, (const char*)new_keys, soff, scnt, (char*)recv_keys, roff, rcnt, stream));
  if (W > 1 && num_features > 1) {   // recat (src, f, b) -> (f, src, b)
    MI355_CHECK_ARG(fm_lengths && fm_offsets && fm_keys, "recat buffers required");
    RW_RC(mi355_permute_lengths(W, num_features, batch_size, recv_lengths, fm_lengths, stream));
    RW_RC(mi355_exclusive_offsets(fm_lengths, W * num_features * batch_size, fm_offsets, stream));
    RW_RC(mi355_permute_bags(W, num_features, batch_size, 8, r / 8, recv_offsets, fm_offsets, recv_keys, fm_keys, stream));
  }
  RW_HIP(hipEventRecord(x->ev_keys[ticket], stream));
  x->live[ticket] = false;
  if (consumer_stream != stream) RW_HIP(hipStreamWaitEvent(consumer_stream, x->ev_keys[ticket], 0));
  return 0;
}

// 1: the key exchange of `ticket` (mi355_rw_input_keys) has completed on the device, 0: not yet
int mi355_rw_keys_ready(void* handle, int ticket) {
  RwExchange* x = (RwExchange*)handle;
  if (!x || ticket < 0 || ticket >= kRing) return 0;
  return hipEventQuery(x->ev_keys[ticket]) == hipSuccess ? 1 : 0;
}

// drops a ticket whose second half will never run (the caller fell back to the c10d sequence)
int mi355_rw_input_cancel(void* handle, int ticket) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && ticket >= 0 && ticket < kRing, "bad arguments");
  x->live[ticket] = false;
  return 0;
}

int mi355_rw_wait_keys(void* handle, int ticket, hipStream_t consumer_stream) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && ticket >= 0 && ticket < kRing, "bad arguments");
  RW_HIP(hipStreamWaitEvent(consumer_stream, x->ev_keys[ticket], 0));
  return 0;
}

int mi355_rw_output_pooled(void* handle, const void* send, void* recv, int64_t numel_per_block, int wire_dtype, void* out,
                           int out_dtype, hipStream_t stream) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && send && recv && out, "bad arguments");
  const int64_t eb = wire_dtype == 0 /*float32*/ ? 4 : 2;
  // the rank's own block stays where the lookup wrote it: W - 1 blocks travel, the sum reads block `rank` from `send`
  RW_RC(all_to_all_equal(x->comm_out, x->world, x->rank, send, recv, numel_per_block * eb, stream, true));
  return mi355_sum_chunks_self(recv, wire_dtype, x->world, numel_per_block, (const char*)send + x->rank * numel_per_block * eb,
                               x->rank, out, out_dtype, stream);
}

int mi355_rw_allgather(void* handle, const void* send, void* recv, int64_t bytes, hipStream_t stream) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && send && recv && bytes >= 0, "bad arguments");
  RW_NCCL(g_rccl.AllGather(send, recv, (size_t)bytes, kNcclInt8, x->comm_out, stream), "ncclAllGather");
  return 0;
}

int mi355_rw_alltoallv(void* handle, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                       int64_t elem_bytes, hipStream_t stream) {
  RwExchange* x = (RwExchange*)handle;
  MI355_CHECK_ARG(x && send_counts && recv_counts && elem_bytes > 0 && x->world <= 512, "bad arguments");
  int64_t soff[512], scnt[512], roff[512], rcnt[512];
  int64_t s = 0, r = 0;
  for (int p = 0; p < x->world; ++p) {
    scnt[p] = send_counts[p] * elem_bytes; rcnt[p] = recv_counts[p] * elem_bytes;
    soff[p] = s; roff[p] = r;
    s += scnt[p]; r += rcnt[p];
  }
  return all_to_all_bytes(x->comm_out, x->world, x->rank, (const char*)send, soff, scnt, (char*)recv, roff, rcnt, stream);
}

}  // extern "C"
