// Entry points shared between the translation units of librecsys_amd.so but not part of the public C ABI: the
// fused variants used by the one-call pipelines (pipeline.hip).  A launch on this GPU costs ~5 us even when the
// kernel does next to nothing, and the DynamicEmb step is a chain of ~25 dependent launches, so the pipelines fold
// the plumbing kernels into their neighbours.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {

// mi355_segmented_unique_csr with two optional fusions:
//  * segmented_range == NULL: the table ranges are derived from (offsets, feature_offsets, feature_x_batch) by the
//    same kernel that clears the scratch set (replaces mi355_get_table_range + a memset); they are left in
//    table_range_out [num_tables+1];
//  * table_ids_out != NULL: table id of every unique key (replaces mi355_expand_table_ids).
int mi355i_segmented_unique(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                            const int64_t* input_frequencies, int count_freq, void* unique_keys,
                            int64_t* output_indices, int64_t* table_offsets, int64_t* freq, int32_t* csr_cnt,
                            int32_t* csr_rank, const int64_t* offsets, const int64_t* feature_offsets,
                            int64_t feature_x_batch, int64_t* table_range_out, int64_t* table_ids_out, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream);

// mi355_table_lookup without the slot lock for ASSIGN / GLOBAL_TIMER score updates (single-stream callers only)
int mi355i_table_lookup(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores, int64_t n,
                        const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                        int policy, uint64_t timer_override, int64_t* score_out, uint8_t* founds, int64_t* indices,
                        hipStream_t stream);

// mi355_table_insert whose unlock pass also produces the row address of every key (replaces mi355_row_addresses)
int mi355i_table_insert(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                        int32_t* bucket_sizes, int32_t* counter, int64_t n, const int64_t* n_dev, const void* keys,
                        const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                        const uint8_t* skip, int64_t* indices, uint8_t* results, const int64_t* table_ptrs,
                        const int64_t* table_value_dims, int elem_bytes, int64_t* row_addr_out, hipStream_t stream);

// the unlock pass of mi355i_table_insert (row_addr_out == NULL there: insert only) fused with mi355_init_rows
int mi355i_unlock_init_rows(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                            const int64_t* indices, const int64_t* table_ptrs, int elem_bytes, int64_t* row_addr_out, int mode,
                            float p0, float p1, float p2, float p3, uint64_t seed, float state_init, int64_t n,
                            const int64_t* n_dev, const void* keys, int dtype, int64_t emb_dim, int64_t value_dim,
                            const uint8_t* results, const uint8_t* skip, const int64_t* table_ids,
                            const int64_t* table_emb_dims, const int64_t* table_value_dims, hipStream_t stream);

int mi355i_table_update_counter_where(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                                      const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                                      const int64_t* table_bucket_offsets, int64_t C, const uint8_t* flags, int want,
                                      hipStream_t stream);

// second half of the fused forward (fused_fwd.hip): scan of the per-unique counts, hot-row registration, CSR scatter and
// reverse indices from the slot-indexed unique ids
// records of the partitioned fused forward (fused_fwd.hip): per (tile, key) record the unique id and the rank base of the
// tile inside the row's list; plus the per-unique row addresses and the per-occurrence address array that keys resolved
// late (deferred eviction) are patched into
struct PartRefs {
  const int2* rec_out = nullptr;     // {unique id, rank base; ~base (< 0) when the row address was resolved late}
  const int64_t* row_addr = nullptr;
  int64_t* occ_addr = nullptr;
};
int mi355i_csr_from_slots(const int32_t* csr_rank, const int32_t* occ_slot, const int32_t* uidmap, int64_t* reverse_indices,
                          int64_t n, const int64_t* offsets, int64_t num_bags, int32_t* ptr, int32_t* csr_src,
                          void* hot_workspace, int64_t hot_workspace_bytes, int64_t dim, int32_t* hdr_reset,
                          const PartRefs* part, hipStream_t stream, const int* gate = nullptr, int gate_val = 0, int* mark = nullptr);
// (gate / gate_val / mark: the opt-in overflow re-run of the fused forward -- the kernel returns at once unless *gate == gate_val
//  and sets *mark = 1 when it runs)

// mi355_backward_fused with the tile lists that CSR reference entries (< 0: source id = tile_bags[~entry]) point into
int mi355i_backward_fused(const int32_t* ptr, const int32_t* csr_src, int64_t num_keys, int64_t max_unique,
                          const int64_t* nu_dev, const void* grads, int64_t grad_stride, int grad_dtype,
                          const int64_t* offsets, const int32_t* D_offsets, int64_t batch_size, int64_t dim, int combiner,
                          const int64_t* row_addr, int weight_dtype, int opt_kind, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int64_t iter_num, int64_t state_offset, int round_grad,
                          void* out, int64_t out_stride, int aligned16, void* workspace, int64_t workspace_bytes,
                          const int32_t* tile_bags, int one_feature, hipStream_t stream);

// stage (0 whole forward, 1 index stage only, 2 gather only) and eviction limit of the NEXT mi355_demb_forward_fused call of this
// thread (fused_fwd.hip; mi355_demb_plan_stage brackets its call with it)
void mi355i_fused_stage(int stage, uint64_t protect);

// bench.py's live kernel timing (err.hip): event of slot (0 gather, 1 backward kernel), end 0 / 1
void mi355i_prof_mark(int slot, int end, hipStream_t stream);

// the library's side stream (fused_fwd.hip): fork from `stream`, mark the join point of the side work (-> token), make a
// stream wait for a token, make a stream wait for whatever side work has not been joined yet
hipStream_t mi355i_side_fork(hipStream_t stream);
int mi355i_side_mark(void);
int mi355i_side_join(int token, hipStream_t stream);
int mi355i_side_join_pending(hipStream_t stream);
}
