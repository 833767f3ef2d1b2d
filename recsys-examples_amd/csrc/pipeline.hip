// Sync-free forward / backward pipelines of the DynamicEmb lookup: the whole per-step sequence
// of BatchedDynamicEmbeddingTablesV2 (HBM-only storage) launched from ONE C call, with every
// intermediate count kept on the device.
//
// Restates the orchestration of (reference, corelib/dynamicemb/dynamicemb/):
//   dynamicemb_prefetch + _prefetch_hbm_direct_path   batched_dynamicemb_function.py:559-833
//   DynamicEmbeddingFunction.forward / backward        batched_dynamicemb_function.py:1042-1300
//   dynamicemb_eval_forward                            batched_dynamicemb_function.py:836-932
// The reference needs 2-4 host syncs per step (segmented_unique sizes, flagged_compact counts);
// here the step is a fixed launch sequence (hipGraph capturable).
#include "common.h"
#include "../../include/recsys_amd.h"
#include "internal.h"
#include "roctx.h"

extern "C" {

// Scratch layout helper: byte offsets of the per-step arrays inside one workspace.
static inline int64_t al(int64_t x) { return (x + 255) / 256 * 256; }

int64_t mi355_demb_forward_workspace_bytes(int64_t num_keys, int64_t num_tables) {
  return al(8 * (num_tables + 1)) /*table_range*/ + al(8 * num_keys) /*unique_keys*/ + al(num_keys) /*founds*/ +
         al(num_keys) /*results*/ + mi355_segmented_unique_workspace_bytes(num_keys) + 256;
}

// Forward of one batch.
//  keys [num_keys] grouped by feature-major bags, offsets [num_bags+1], feature_offsets [T+1] (device).
//  Persisted outputs (caller-allocated, consumed by the backward): reverse_indices [num_keys] i64,
//  unique_offsets [T+1] i64 (unique_offsets[T] = number of uniques), table_ids [num_keys] i64,
//  slots [num_keys] i64, row_addr [num_keys] i64.
//  train != 0: missing keys are inserted (insert_policy / insert_scores) and their rows initialised in
//  place; train == 0: missing keys contribute zeros (eval, key_value_table.py:2915-2949).
//  pooling: combiner 0 sum / 1 mean -> out [batch_size, total_D]; combiner -1 sequence -> out [num_keys, dim];
//  combiner -2: no output (prefetch of BatchedDynamicEmbeddingTablesV2.prefetch, batched_dynamicemb_tables.py:1090-1137).
int mi355_demb_forward(
    /* table */ void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity, int64_t num_scores,
    int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel,
    /* values */ const int64_t* table_ptrs, const int64_t* table_value_dims, const int64_t* table_emb_dims,
    int value_dtype, int64_t emb_dim,
    int64_t value_dim,
    /* batch */ const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
    const int64_t* feature_offsets, int64_t num_tables,
    /* policies */ int train, int find_policy, const void* find_scores, int insert_policy, const void* insert_scores,
    uint64_t timer_override, int pin,
    /* initializer */ int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
    /* output */ int combiner, const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype, int aligned16,
    /* persisted */ int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids, int64_t* slots,
    int64_t* row_addr, int64_t* freq /* nullable: per-unique occurrence counts (LFU scores) */,
    int32_t* csr_cnt, int32_t* csr_rank /* nullable: CSR ingredients for mi355_demb_backward */,
    /* early CSR (nullable): the workspace the caller will hand to mi355_demb_backward(prepared = 1) for THIS batch; the
       key-grouping half of the backward then runs on a side stream, forked right after the dedup, under the lookup /
       gather kernels of this forward.  The buffer must stay alive until that backward has been issued. */
    void* backward_workspace, int64_t backward_workspace_bytes,
    /* out (nullable): token of the side-stream join point, for mi355_demb_backward(prepared = 2 + token); -1 when no
       early CSR was started.  NULL: the early CSR is joined on `stream` before this call returns. */
    int* join_token,
    /* scratch */ void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(counter || !train, "a training forward needs the ref-counter array (found slots are pinned across the insert)");
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_demb_forward_workspace_bytes(num_keys, num_tables),
                  "workspace too small");
  if (join_token) *join_token = -1;
  if (num_keys == 0 && combiner < 0) return MI355_OK;  // (pooled output of an empty batch is still zero-filled below)
  uint8_t* w = (uint8_t*)workspace;
  int64_t* table_range = (int64_t*)w; w += al(8 * (num_tables + 1));
  void* unique_keys = w; w += al(8 * num_keys);
  uint8_t* founds = w; w += al(num_keys);
  uint8_t* results = w; w += al(num_keys);
  void* uws = w;
  const int64_t uws_bytes = mi355_segmented_unique_workspace_bytes(num_keys);
  const int64_t* nu_dev = unique_offsets + num_tables;
  int rc;
#define STEP(call) do { rc = (call); if (rc != MI355_OK) return rc; } while (0)
  // table ranges, dedup, table ids of the unique keys: one call, no separate range / memset / expand launches
  { mi355::RoctxRange rr("op:segmented_unique");
  STEP(mi355i_segmented_unique(keys, num_keys, nullptr, num_tables, nullptr, freq ? 1 : 0, unique_keys, reverse_indices,
                               unique_offsets, freq, csr_cnt, csr_rank, offsets, feature_offsets, num_bags, table_range,
                               table_ids, uws, uws_bytes, stream)); }
  if (backward_workspace && train && csr_cnt && csr_rank && num_keys > 0 && combiner != -2) {
    MI355_CHECK_ARG(backward_workspace_bytes >= mi355_demb_backward_workspace_bytes(num_keys, emb_dim), "backward workspace too small");
    hipStream_t side = mi355i_side_fork(stream);
    if (!side) { mi355_set_error("early CSR fork failed"); return MI355_ELAUNCH; }
    uint8_t* bw = (uint8_t*)backward_workspace;
    int32_t* bptr = (int32_t*)bw; bw += al(4 * (num_keys + 1));
    int32_t* bcsr = (int32_t*)bw; bw += al(4 * num_keys);
    void* gws = bw;
    const int64_t gws_bytes = mi355_group_by_unique_workspace_bytes(num_keys, num_keys);
    bw += gws_bytes;
    STEP(mi355_group_by_unique_csr(csr_cnt, csr_rank, reverse_indices, num_keys, combiner >= 0 ? offsets : nullptr,
                                   num_bags, num_keys, nu_dev, bptr, bcsr, gws, gws_bytes, bw,
                                   mi355_backward_workspace_bytes(num_keys, emb_dim), emb_dim, side));
    const int tok = mi355i_side_mark();
    if (tok < 0) { mi355_set_error("early CSR join record failed"); return MI355_ELAUNCH; }
    if (join_token) *join_token = tok;
    else STEP(mi355i_side_join(tok, stream));
  }
  if (!find_scores) find_scores = freq;      // LFU: scores are the occurrence counts of this batch
  if (!insert_scores) insert_scores = freq;
  if (num_keys > 0) {
    { mi355::RoctxRange rr("op:storage_find");
    STEP(mi355i_table_lookup(storage, table_bucket_offsets, bucket_capacity, num_scores, num_keys, nu_dev, unique_keys,
                            table_ids, find_scores, find_policy, timer_override, nullptr, founds, slots, stream)); }
    if (train) {
      mi355::RoctxRange rr("op:storage_insert+initializer");
      // The slots this batch's lookup found must not be evicted by this batch's insert (the reference pins them with
      // increment_counter before the insert, _prefetch_hbm_direct_path batched_dynamicemb_function.py:559-696): +1 on the
      // found slots now, released after the unlock pass unless the caller keeps the pin (prefetch).
      STEP(mi355_table_update_counter(counter, counter_numel, slots, num_keys, nu_dev, 1, table_ids, table_bucket_offsets,
                                      bucket_capacity, stream));
      // insert (slots stay locked); then ONE launch publishes the keys (unlock), writes the row address of every unique
      // key and initialises the rows that are new
      STEP(mi355i_table_insert(storage, table_bucket_offsets, bucket_capacity, num_scores, bucket_sizes, counter,
                               num_keys, nu_dev, unique_keys, table_ids, insert_scores, insert_policy, timer_override,
                               founds, slots, results, table_ptrs, table_value_dims, value_dtype == 0 ? 4 : 2, nullptr,
                               stream));
      STEP(mi355i_unlock_init_rows(storage, table_bucket_offsets, bucket_capacity, num_scores, slots, table_ptrs,
                                   value_dtype == 0 ? 4 : 2, row_addr, init_mode, p0, p1, p2, p3, seed, state_init, num_keys,
                                   nu_dev, unique_keys, value_dtype, emb_dim, value_dim, results, founds, table_ids,
                                   table_emb_dims, table_value_dims, stream));
      // the pre-insert pin covered the FOUND slots only (-1 elsewhere then); prefetch keeps a pin on every slot of the
      // batch, so: release the found ones when nothing is kept, or add the new ones when it is
      if (pin)
        STEP(mi355i_table_update_counter_where(counter, counter_numel, slots, num_keys, nu_dev, 1, table_ids,
                                               table_bucket_offsets, bucket_capacity, founds, 0, stream));
      else
        STEP(mi355i_table_update_counter_where(counter, counter_numel, slots, num_keys, nu_dev, -1, table_ids,
                                               table_bucket_offsets, bucket_capacity, founds, 1, stream));
    } else {
      STEP(mi355_row_addresses(num_keys, nu_dev, slots, table_ids, table_ptrs, table_value_dims,
                               value_dtype == 0 ? 4 : 2, row_addr, stream));
    }
  }
  mi355::RoctxRange rr_g("op:gather_embedding");
  if (combiner >= 0) {
    STEP(mi355_gather_pooled(nullptr, 0, row_addr, value_dtype, reverse_indices, num_keys, offsets, num_bags, batch_size,
                             combiner, emb_dim, D_offsets, total_D, out, out_dtype, aligned16, stream));
  } else if (combiner == -2) {
    // prefetch: the index stage only (dedup, find / insert, pin, row addresses); the rows are gathered by the forward
  } else {
    STEP(mi355_gather_rows(nullptr, 0, row_addr, value_dtype, reverse_indices, num_keys, nullptr, emb_dim, out,
                           emb_dim, out_dtype, aligned16, stream));
  }
#undef STEP
  return MI355_OK;
}

int64_t mi355_demb_backward_workspace_bytes(int64_t num_keys, int64_t dim) {
  return al(4 * (num_keys + 1)) /*ptr*/ + al(4 * (num_keys > 0 ? num_keys : 1)) /*csr*/ +
         mi355_group_by_unique_workspace_bytes(num_keys, num_keys) + mi355_backward_workspace_bytes(num_keys, dim) + 256;
}

// Backward of one batch: group keys by unique row, reduce the gradients, apply the optimizer in place,
// release the pins taken by the forward.
int mi355_demb_backward(
    const int64_t* reverse_indices, int64_t num_keys, const int64_t* unique_offsets, int64_t num_tables,
    const int64_t* offsets, int64_t num_bags, int64_t batch_size, const void* grads, int64_t grad_stride, int grad_dtype,
    const int32_t* D_offsets, int64_t dim, int combiner, const int64_t* row_addr, int value_dtype, int opt_kind,
    float lr, float beta1, float beta2, float eps, float weight_decay, int64_t iter_num, int64_t state_offset,
    int round_grad, int aligned16,
    /* unpin */ int32_t* counter, int64_t counter_numel, const int64_t* slots, const int64_t* table_ids,
    const int64_t* table_bucket_offsets, int64_t bucket_capacity, int unpin,
    /* CSR ingredients persisted by the forward (both or neither) */ const int32_t* csr_cnt, const int32_t* csr_rank,
    /* prepared != 0: `workspace` is the buffer the forward of this batch received as backward_workspace -- the grouping is
       already done: 1 = on this stream, 2 + token = on the library's side stream (joined here; token from the forward) */
    int prepared,
    void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_demb_backward_workspace_bytes(num_keys, dim), "workspace too small");
  if (num_keys == 0) return MI355_OK;
  uint8_t* w = (uint8_t*)workspace;
  int32_t* ptr = (int32_t*)w; w += al(4 * (num_keys + 1));
  int32_t* csr = (int32_t*)w; w += al(4 * num_keys);
  void* gws = w;
  const int64_t gws_bytes = mi355_group_by_unique_workspace_bytes(num_keys, num_keys);
  w += gws_bytes;
  void* bws = w;
  const int64_t bws_bytes = mi355_backward_workspace_bytes(num_keys, dim);
  const int64_t* nu_dev = unique_offsets + num_tables;
  int rc;
  if (prepared >= 2) {
    rc = mi355i_side_join(prepared - 2, stream);
    if (rc != MI355_OK) { mi355_set_error("early CSR join failed"); return rc; }
  } else if (prepared == 1) {
    rc = MI355_OK;
  } else if (csr_cnt && csr_rank)
    rc = mi355_group_by_unique_csr(csr_cnt, csr_rank, reverse_indices, num_keys, combiner >= 0 ? offsets : nullptr, num_bags,
                                   num_keys, nu_dev, ptr, csr, gws, gws_bytes, bws, bws_bytes, dim, stream);
  else
    rc = mi355_group_by_unique(reverse_indices, num_keys, combiner >= 0 ? offsets : nullptr, num_bags, num_keys, nu_dev, ptr,
                               csr, gws, gws_bytes, bws, bws_bytes, dim, stream);
  if (rc != MI355_OK) return rc;
  mi355::RoctxRange rr_b("op:reduce_grads+optimizer_update");
  // (prepared by the fused forward: CSR reference entries point into the tile lists it left at the head of the grouping
  //  workspace; every other producer writes plain source ids, for which the pointer is never dereferenced)
  rc = mi355i_backward_fused(ptr, csr, num_keys, num_keys, nu_dev, grads, grad_stride, grad_dtype, offsets, D_offsets,
                             batch_size, dim, combiner, row_addr, value_dtype, opt_kind, lr, beta1, beta2, eps, weight_decay,
                             iter_num, state_offset, round_grad, nullptr, 0, aligned16, bws, bws_bytes,
                             prepared ? (const int32_t*)gws : nullptr, num_bags == batch_size, stream);
  if (rc != MI355_OK) return rc;
  if (unpin)
    rc = mi355_table_update_counter(counter, counter_numel, slots, num_keys, nu_dev, -1, table_ids, table_bucket_offsets,
                                    bucket_capacity, stream);
  return rc;
}


// ---- round 5: the pre-bound training step -------------------------------------------------------------------------------------
// A module's fused forward takes ~60 arguments of which a handful change from step to step; marshalling them through ctypes on
// every call (plus the Python that gathers them) was a third of the host time of a step driven through module.forward() +
// autograd (bench.py step_via_autograd_ms 0.179 vs 0.128 on the driver's box).  A plan holds everything that is constant for a
// module (table, value buffers, policies, initializer, optimizer); the per-step calls take the batch, the output and ONE step
// buffer whose layout (the persisted arrays of the step + both workspaces) is computed here -- mi355_demb_step_layout is the
// single source of it.  Reference counterpart: the constructor state of BatchedDynamicEmbeddingTablesV2
// (batched_dynamicemb_tables.py:462-787) that DynamicEmbeddingFunction.forward / backward read on every step (:999-1088).
struct DembPlan {
  void* storage; const int64_t* tbo; int64_t C, ns; int32_t* bucket_sizes; int32_t* counter; int64_t counter_numel;
  int32_t* aux; int64_t aux_numel, num_buckets;
  const int64_t* table_ptrs; const int64_t* table_value_dims; const int64_t* table_emb_dims; int value_dtype; int64_t emb_dim, value_dim;
  const int64_t* feature_offsets; int64_t T;
  int find_policy, insert_policy, use_count, pin;
  int init_mode; float p0, p1, p2, p3; uint64_t seed; float state_init;
  int combiner; const int32_t* D_offsets; int64_t total_D; int out_dtype, aligned16;
  int opt_kind; float beta1, beta2, eps, weight_decay;
  // staged forwards (mi355_demb_plan_stage): per step-ring slot, the fork point of a prefetch and the end of its index stage
  hipEvent_t ev_fork[4] = {}, ev_done[4] = {};
};

void* mi355_demb_plan_create(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity, int64_t num_scores,
                             int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel, int32_t* aux, int64_t aux_numel,
                             int64_t num_buckets, const int64_t* table_ptrs, const int64_t* table_value_dims,
                             const int64_t* table_emb_dims, int value_dtype, int64_t emb_dim, int64_t value_dim,
                             const int64_t* feature_offsets, int64_t num_tables, int find_policy, int insert_policy, int use_count,
                             int pin, int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
                             int combiner, const int32_t* D_offsets, int64_t total_D, int out_dtype, int aligned16, int opt_kind,
                             float beta1, float beta2, float eps, float weight_decay) {
  DembPlan* p = new DembPlan{storage, table_bucket_offsets, bucket_capacity, num_scores, bucket_sizes, counter, counter_numel,
                             aux, aux_numel, num_buckets, table_ptrs, table_value_dims, table_emb_dims, value_dtype, emb_dim,
                             value_dim, feature_offsets, num_tables, find_policy, insert_policy, use_count, pin, init_mode, p0,
                             p1, p2, p3, seed, state_init, combiner, D_offsets, total_D, out_dtype, aligned16, opt_kind, beta1,
                             beta2, eps, weight_decay};
  return p;
}
void mi355_demb_plan_destroy(void* plan) {
  DembPlan* p = (DembPlan*)plan;
  if (!p) return;
  for (int i = 0; i < 4; ++i) {
    if (p->ev_fork[i]) (void)hipEventDestroy(p->ev_fork[i]);
    if (p->ev_done[i]) (void)hipEventDestroy(p->ev_done[i]);
  }
  delete p;
}

// Byte offsets of the arrays of one step buffer: out[0..9] = rev, tids, slots, row_addr, freq (8 n each), csr_cnt, csr_rank
// (4 n each), unique_offsets (8 (T + 1)), forward workspace, backward workspace; out[10] = total bytes, out[11] / out[12] = the two
// workspace sizes.  (dynamicemb/batched_dynamicemb_tables.py: _FusedStep reads its layout from here.)
void mi355_demb_step_layout(int64_t num_keys, int64_t num_tables, int64_t dim, int train, int64_t* out) {
  const int64_t n1 = num_keys > 0 ? num_keys : 1;
  int64_t off = 0;
  for (int k = 0; k < 5; ++k) { out[k] = off; off += al(8 * n1); }
  for (int k = 5; k < 7; ++k) { out[k] = off; off += al(4 * n1); }
  out[7] = off; off += al(8 * (num_tables + 1));
  const int64_t fwd_b = mi355_demb_forward_fused_workspace_bytes(num_keys, num_tables);
  const int64_t bwd_b = train ? mi355_demb_backward_workspace_bytes(num_keys, dim) : 0;
  out[8] = off; off += al(fwd_b);
  out[9] = off; off += al(bwd_b);
  out[10] = off; out[11] = fwd_b; out[12] = bwd_b;
}
int64_t mi355_demb_plan_step_bytes(void* plan, int64_t num_keys) {
  const DembPlan* p = (const DembPlan*)plan;
  int64_t lay[13];
  mi355_demb_step_layout(num_keys, p->T, p->emb_dim, 1, lay);
  return lay[10];
}

// Training forward of one batch through the plan.  Returns MI355_OK, an error, or MI355_ENOSPC (1) when `step_buf` is smaller
// than mi355_demb_plan_step_bytes(num_keys) (nothing was launched).  *state (out): the join_token of mi355_demb_forward_fused
// (<= -2: path (c), lazy reverse indices -- -2 - *state = the step's epoch for mi355_demb_plan_backward, 0: nothing to ask;
// -1: everything on `stream`).
int mi355_demb_plan_forward(void* plan, const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags,
                            int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* out, void* step_buf,
                            int64_t step_bytes, int* state, hipStream_t stream) {
  const DembPlan* p = (const DembPlan*)plan;
  MI355_CHECK_ARG(p && step_buf, "plan forward: null plan / step buffer");
  int64_t lay[13];
  mi355_demb_step_layout(num_keys, p->T, p->emb_dim, 1, lay);
  if (step_bytes < lay[10]) return 1;
  uint8_t* b = (uint8_t*)step_buf;
  return mi355_demb_forward_fused(p->storage, p->tbo, p->C, p->ns, p->bucket_sizes, p->counter, p->counter_numel, p->aux,
                                  p->aux_numel, p->num_buckets, p->table_ptrs, p->table_value_dims, p->table_emb_dims,
                                  p->value_dtype, p->emb_dim, p->value_dim, keys, num_keys, offsets, num_bags, batch_size,
                                  p->feature_offsets, p->T, 1, p->find_policy, p->insert_policy, score_value, p->use_count,
                                  timer_override, p->pin, p->init_mode, p->p0, p->p1, p->p2, p->p3, p->seed, p->state_init,
                                  p->combiner, p->D_offsets, p->total_D, out, p->out_dtype, p->aligned16, (int64_t*)(b + lay[0]),
                                  (int64_t*)(b + lay[7]), (int64_t*)(b + lay[1]), (int64_t*)(b + lay[2]), (int64_t*)(b + lay[3]),
                                  p->use_count ? (int64_t*)(b + lay[4]) : nullptr, (int32_t*)(b + lay[5]), (int32_t*)(b + lay[6]),
                                  b + lay[9], lay[12], 0, state, b + lay[8], lay[11], stream);
}

// The same forward in two halves (round 6: the reference's prefetch pipeline -- BatchedDynamicEmbeddingTablesV2.prefetch,
// batched_dynamicemb_tables.py:1090-1137, PrefetchTrainPipelineSparseDist, train_pipeline.py:533-692 -- on path (c)).  stage 1: the
// index stage only (probe + partition kernel: every table mutation of the step, the per-occurrence row addresses, the backward's
// CSR) -- `out` unused; issued for batch k + 1 on a side stream, it runs under the backward of batch k.  stage 2: the gather of a
// step whose stage 1 ran earlier (same batch arguments, same step buffer), on the stream that needs the output.  stage 0 =
// mi355_demb_plan_forward.  protect_score: an eviction of this call takes no slot whose score is >= it (recency scores: the score
// of the oldest step still in flight keeps every row of every in-flight step where it is; ~0: no limit).  Returns 3, nothing
// launched, when the batch is not eligible for path (c) (the caller uses its one-call forward / the pinning prefetch).
// Stream order of a staged step, kept inside the library (an event pair per slot, slot in 0..3, -1: the caller orders the streams
// itself): stage 1 with fork_from != stream -- `stream` first waits for what `fork_from` holds at this point (the batch was
// produced there, and the rows the previous backward writes must be final before an eviction may re-initialise one), and the end
// of the index stage is marked; stage 2 -- `stream` waits for that mark before it gathers.
int mi355_demb_plan_stage(void* plan, int stage, uint64_t protect_score, const void* keys, int64_t num_keys, const int64_t* offsets,
                          int64_t num_bags, int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* out,
                          void* step_buf, int64_t step_bytes, int* state, hipStream_t fork_from, int slot, hipStream_t stream) {
  DembPlan* p = (DembPlan*)plan;
  MI355_CHECK_ARG(p && stage >= 0 && stage <= 2 && slot >= -1 && slot < 4, "plan stage: stage 0..2, slot -1..3");
  if (slot >= 0 && stage != 0) {
    if (!p->ev_fork[slot]) {
      if (hipEventCreateWithFlags(&p->ev_fork[slot], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->ev_done[slot], hipEventDisableTiming) != hipSuccess) {
        mi355_set_error("plan stage: event creation failed");
        return MI355_ELAUNCH;
      }
    }
    if (stage == 1 && fork_from != stream) {
      if (hipEventRecord(p->ev_fork[slot], fork_from) != hipSuccess || hipStreamWaitEvent(stream, p->ev_fork[slot], 0) != hipSuccess) {
        mi355_set_error("plan stage: fork failed");
        return MI355_ELAUNCH;
      }
    }
    if (stage == 2 && hipStreamWaitEvent(stream, p->ev_done[slot], 0) != hipSuccess) {
      mi355_set_error("plan stage: join failed");
      return MI355_ELAUNCH;
    }
  }
  mi355i_fused_stage(stage, protect_score);
  const int rc = mi355_demb_plan_forward(plan, keys, num_keys, offsets, num_bags, batch_size, score_value, timer_override, out, step_buf,
                                         step_bytes, state, stream);
  mi355i_fused_stage(0, ~0ull);
  if (rc == 0 && stage == 1 && slot >= 0 && hipEventRecord(p->ev_done[slot], stream) != hipSuccess) {
    mi355_set_error("plan stage: mark failed");
    return MI355_ELAUNCH;
  }
  return rc;
}

// Backward of the step whose forward went through mi355_demb_plan_forward on the same buffer (prepared: 1 when that forward
// left the CSR in the buffer -- any non-empty batch --, 0 to regroup here).  The optimizer's hyper-parameters travel with the
// call (a learning-rate schedule must not rebuild the plan).
int mi355_demb_plan_backward(void* plan, void* step_buf, int64_t step_bytes, int64_t num_keys, const int64_t* offsets,
                             int64_t num_bags, int64_t batch_size, const void* grads, int64_t grad_stride, int grad_dtype,
                             int grad_aligned16, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t iter_num, int prepared, int epoch, hipStream_t stream) {
  const DembPlan* p = (const DembPlan*)plan;
  MI355_CHECK_ARG(p && step_buf, "plan backward: null plan / step buffer");
  if (epoch > 0) {      // a path-(c) step: did its partition lists hold every record?  (one read of pinned memory)
    const int f = mi355_demb_fused_step_flooded(epoch, 20000);
    if (f < 0) { mi355_set_error("plan backward: the forward of this step never reported (GPU stuck?)"); return MI355_ELAUNCH; }
    if (f == 3) { mi355_set_error("plan backward: a gather block of this step's forward gave up waiting for its partition block: the step's output lacks rows"); return MI355_ELAUNCH; }
    if (f > 0) return 2;   // nothing launched: mi355_demb_plan_rerun first, then this call again with epoch = 0
  }
  int64_t lay[13];
  mi355_demb_step_layout(num_keys, p->T, p->emb_dim, 1, lay);
  MI355_CHECK_ARG(step_bytes >= lay[10], "plan backward: step buffer too small");
  uint8_t* b = (uint8_t*)step_buf;
  return mi355_demb_backward((const int64_t*)(b + lay[0]), num_keys, (const int64_t*)(b + lay[7]), p->T, offsets, num_bags,
                             batch_size, grads, grad_stride, grad_dtype, p->D_offsets, p->emb_dim, p->combiner,
                             (const int64_t*)(b + lay[3]), p->value_dtype, p->opt_kind, lr, beta1, beta2, eps,
                             weight_decay, iter_num, -1, 1, p->aligned16 && grad_aligned16, p->counter, p->counter_numel,
                             (const int64_t*)(b + lay[2]), (const int64_t*)(b + lay[1]), p->tbo, p->C, p->pin,
                             (const int32_t*)(b + lay[5]), (const int32_t*)(b + lay[6]), prepared, b + lay[9], lay[12], stream);
}


// Regroups a flooded step (mi355_demb_plan_backward returned 2) on the per-slot-counter path: the batch arguments of the step's
// mi355_demb_plan_forward call again, plus its epoch (-2 - *state of that call).
int mi355_demb_plan_rerun(void* plan, const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags,
                          int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* step_buf,
                          int64_t step_bytes, int epoch, hipStream_t stream) {
  const DembPlan* p = (const DembPlan*)plan;
  MI355_CHECK_ARG(p && step_buf, "plan rerun: null plan / step buffer");
  int64_t lay[13];
  mi355_demb_step_layout(num_keys, p->T, p->emb_dim, 1, lay);
  MI355_CHECK_ARG(step_bytes >= lay[10], "plan rerun: step buffer too small");
  uint8_t* b = (uint8_t*)step_buf;
  return mi355_demb_forward_fused_rerun(p->storage, p->tbo, p->C, p->ns, p->bucket_sizes, p->counter, p->counter_numel, p->aux,
                                        p->aux_numel, p->num_buckets, p->table_ptrs, p->table_value_dims, p->table_emb_dims,
                                        p->value_dtype, p->emb_dim, p->value_dim, keys, num_keys, offsets, num_bags, batch_size,
                                        p->feature_offsets, p->T, 1, p->find_policy, p->insert_policy, score_value, p->use_count,
                                        timer_override, p->pin, p->init_mode, p->p0, p->p1, p->p2, p->p3, p->seed, p->state_init,
                                        p->combiner, p->D_offsets, p->total_D, nullptr, p->out_dtype, p->aligned16,
                                        (int64_t*)(b + lay[0]), (int64_t*)(b + lay[7]), (int64_t*)(b + lay[1]), (int64_t*)(b + lay[2]),
                                        (int64_t*)(b + lay[3]), p->use_count ? (int64_t*)(b + lay[4]) : nullptr,
                                        (int32_t*)(b + lay[5]), (int32_t*)(b + lay[6]), b + lay[9], lay[12], 0, epoch, b + lay[8],
                                        lay[11], stream);
}

}  // extern "C"
