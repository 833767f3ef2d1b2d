/*
 * recsys_amd.h -- C ABI of librecsys_amd.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the two hot paths of NVIDIA/recsys-examples:
 *   A. DynamicEmb lookup  -- replaces the pybind11 module `dynamicemb_extensions`
 *      (corelib/dynamicemb/src/module_bind.cu:33-44); every entry point below cites the
 *      reference function it stands in for.
 *   B. HSTU jagged attention -- replaces `hstu_attn_2_cuda.varlen_fwd / varlen_bwd`
 *      (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:335-725) / `torch.ops.fbgemm.hstu_varlen_{fwd,bwd}_*`.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (unless a parameter says "host"), sizes, a hipStream_t.
 *     No torch types.  The caller owns every buffer; nothing is allocated or freed here.
 *   - every function returns 0 on success, MI355_EINVAL (-1) for a rejected argument,
 *     MI355_ELAUNCH (-2) for a HIP launch error; mi355_last_error() holds the message
 *     (thread local).  Per-key failure is DATA (index -1, InsertResult), never an error code,
 *     as in the reference (src/check.h:40-60; kernels.cuh).
 *   - everything is asynchronous on `stream`; no entry point synchronises the host.
 *   - `n_dev` / `nu_dev` (nullable): the element count may live on the device; kernels are
 *     launched for the upper bound `n` and use min(n, *n_dev).  This removes the host syncs
 *     the reference API bakes in (h_num_missing, num_evicted.item(), ...).
 *   - keys are 64-bit (int64 or uint64 bit patterns), indices/offsets int64, as in the reference.
 *   - dtype codes: 0 = float32, 1 = bfloat16, 2 = float16.
 *   - scratch comes from the caller: mi355_*_workspace_bytes() gives the size.
 *
 * Reference paths are relative to /root/reference/corelib/dynamicemb/ unless noted.
 */
#ifndef RECSYS_AMD_H_
#define RECSYS_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define MI355_OK 0
#define MI355_EINVAL (-1)
#define MI355_ELAUNCH (-2)

/* ScorePolicy: src/table_operation/score.cuh:30-43 (pybind enum table.cu:186-192) */
enum { MI355_POLICY_CONST = 0, MI355_POLICY_ASSIGN = 1, MI355_POLICY_ACCUMULATE = 2,
       MI355_POLICY_GLOBAL_TIMER = 3, MI355_POLICY_LRU_LFU = 4 };
/* InsertResult: src/table_operation/types.cuh:52-61 */
enum { MI355_RES_INSERT = 0, MI355_RES_RECLAIM = 1, MI355_RES_ASSIGN = 2, MI355_RES_EVICT = 3,
       MI355_RES_DUPLICATED = 4, MI355_RES_BUSY = 5, MI355_RES_ILLEGAL = 6, MI355_RES_INIT = 7 };

const char* mi355_last_error(void);
int mi355_abi_version(void);

/* ------------------------------------------------------------------ scored hash table ---- */

/* LinearBucketTable._init_table, scored_hashtable.py:476-495 (keys = Empty, scores = 0,
 * digests = empty digest).  storage: num_buckets * C * (9 + 8*num_scores) bytes. */
int mi355_table_init(void* storage, int64_t num_buckets, int64_t bucket_capacity, int64_t num_scores,
                     hipStream_t stream);

/* table_lookup, src/table_operation/lookup.cu:151-191 + table_lookup_kernel kernels.cuh:81-187.
 * founds: uint8 (bool), indices: table-relative slot or -1, score_out as the reference.
 * timer_override != 0 replaces the device clock for GLOBAL_TIMER / LRU_LFU (test hook). */
int mi355_table_lookup(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                       int64_t num_scores, int64_t n, const int64_t* n_dev, const void* keys,
                       const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                       int64_t* score_out, uint8_t* founds, int64_t* indices, hipStream_t stream);

/* table_insert / table_insert_and_evict (+ table_unlock_kernel), src/table_operation/insert.cu:36-45,
 * insert_and_evict.cu:81-93, kernels.cuh:189-585.  Keys must be unique (scored_hashtable.py:148).
 * num_evicted == NULL -> plain insert; otherwise the evicted (key, index, score, table_id) streams are
 * compacted (arbitrary order) and *num_evicted (device int64) counts them.
 * skip (nullable uint8[n]): keys with skip[i] != 0 are ignored (indices[i] untouched). */
int mi355_table_insert(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                       int64_t num_scores, int32_t* bucket_sizes, int32_t* counter, int64_t n,
                       const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                       int policy, uint64_t timer_override, const uint8_t* skip, int64_t* indices,
                       uint8_t* results, int64_t* score_out, int64_t* num_evicted, void* evicted_keys,
                       int64_t* evicted_indices, int64_t* evicted_scores, int64_t* evicted_table_ids,
                       hipStream_t stream);

/* table_erase, src/table_operation/erase.cu + table_erase_kernel kernels.cuh:587-652 */
int mi355_table_erase(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                      int64_t num_scores, int32_t* bucket_sizes, int64_t n, const void* keys,
                      const int64_t* table_ids, int64_t* indices, hipStream_t stream);

/* table_export_batch, src/table_operation/export_batch.cu:22-124 (kernel kernels.cuh:655-705): the valid slots of
 * [offset, offset+batch) whose score[score_index] >= threshold (has_threshold) -> compacted (keys, scores,
 * indices = slot - table_begin), *counter = how many.  Output is in slot order (the reference: atomic order). */
int64_t mi355_table_export_batch_workspace_bytes(int64_t batch);
int mi355_table_export_batch(const void* storage, int64_t num_buckets, int64_t bucket_capacity, int64_t num_scores,
                             int64_t batch, int64_t offset, int has_threshold, uint64_t threshold,
                             int64_t table_begin, int64_t score_index, int64_t* counter, void* keys,
                             int64_t* scores, int64_t* indices, void* workspace, int64_t workspace_bytes,
                             hipStream_t stream);

/* table_count_matched, src/table_operation/count_matched.cu:23-93: number of valid slots in [begin, end)
 * (negative = whole table) with score[score_index] >= threshold (unsigned compare) -> *num_matched. */
int mi355_table_count_matched(const void* storage, int64_t num_buckets, int64_t bucket_capacity, int64_t num_scores,
                              uint64_t threshold, int64_t begin, int64_t end, int64_t score_index,
                              int64_t* num_matched, hipStream_t stream);

/* table_{copy,gather,scatter}_score_blocks, src/table_operation/insert.cu:155-254 (kernels.cuh:839-910):
 * mode 0 copy src[src_slots]->dst[dst_slots], 1 gather src[src_slots]->dense [n, num_scores], 2 scatter
 * dense->dst[dst_slots]; slots are table-relative, *_bkt_begin the table's first bucket; slot < 0 skipped
 * (gather writes zeros). */
int mi355_table_score_blocks(int mode, const void* src_storage, int64_t src_bucket_capacity, int64_t src_bkt_begin,
                             void* dst_storage, int64_t dst_bucket_capacity, int64_t dst_bkt_begin,
                             int64_t num_scores, int64_t n, const int64_t* src_slots, const int64_t* dst_slots,
                             int64_t* dense, hipStream_t stream);

/* table_update_counter_with_layout, insert_and_evict.cu:27-58,397-430 (main-table region) */
int mi355_table_update_counter(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                               const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                               const int64_t* table_bucket_offsets, int64_t bucket_capacity, hipStream_t stream);

/* ---- overflow region of cache tables (scored_hashtable.py:426-474, DynamicEmbCache key_value_table.py:1522-1590).
 * ovf_storage: a second table arena with ONE bucket of ovf_capacity (= 3 x bucket_capacity) slots per logical table
 * (same SoA layout, mi355_table_init(ovf_storage, num_tables, ovf_capacity, num_scores)); ovf_bucket_sizes int32[T];
 * ovf_counter int32[T * ovf_capacity] (the tail of the ref-counter array); ovf_output_offsets int64[T] = main capacity
 * of each table: an overflow entry's table-relative index is ovf_output_offsets[t] + position.
 *
 * table_lookup with ovf_storage (lookup.cu:82-150, kernels.cuh:153-183,711-736): keys the main table does not hold are
 * searched in their table's overflow bucket (linear probing from hash % ovf_capacity, stops at Empty). */
int mi355_table_lookup_overflow(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                                int64_t num_scores, int64_t n, const int64_t* n_dev, const void* keys,
                                const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                                void* ovf_storage, int64_t ovf_capacity, const int64_t* ovf_output_offsets,
                                int64_t* score_out, uint8_t* founds, int64_t* indices, hipStream_t stream);

/* table_insert_and_evict with counter and overflow (insert_and_evict.cu:201-395, kernels.cuh:389-566,738-800): keys whose
 * main bucket is entirely pinned (result Busy) find-or-insert in the overflow bucket; its victims are entries with
 * ref-counter 0.  `results` is required (Insert / Evict / Assign name overflow successes too; compare the index with
 * ovf_output_offsets to tell the regions apart).  Evicted stream as mi355_table_insert: a main-table eviction reports
 * (victim key, its slot), an overflow eviction (victim key, overflow index), a key nobody took (key, -(i+1)).
 * workspace: mi355_table_insert_overflow_workspace_bytes(n). */
int64_t mi355_table_insert_overflow_workspace_bytes(int64_t n);
int mi355_table_insert_overflow(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                                int64_t num_scores, int32_t* bucket_sizes, int32_t* counter, int64_t n,
                                const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                                int policy, uint64_t timer_override, const uint8_t* skip, void* ovf_storage,
                                int64_t ovf_capacity, int32_t* ovf_bucket_sizes, int32_t* ovf_counter,
                                const int64_t* ovf_output_offsets, int64_t* indices, uint8_t* results, int64_t* score_out,
                                int64_t* num_evicted, void* evicted_keys, int64_t* evicted_indices,
                                int64_t* evicted_scores, int64_t* evicted_table_ids, void* workspace,
                                int64_t workspace_bytes, hipStream_t stream);

/* table_update_counter_with_layout with the overflow layout (insert_and_evict.cu:27-58): slot < ovf_output_offsets[t]
 * -> counter[table_bucket_offsets[t]*C + slot], else counter[main_capacity + t*ovf_capacity + slot - offsets[t]]. */
int mi355_table_update_counter_overflow(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                                        const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                                        const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                                        int64_t main_capacity, const int64_t* ovf_output_offsets, int64_t ovf_capacity,
                                        hipStream_t stream);

/* device_timestamp, src/torch_utils.cu:22-40,150 */
int mi355_device_timestamp(int64_t* out, hipStream_t stream);

/* ------------------------------------------------------------------------ index ops ---- */

/* segmented_unique_cuda, src/unique_op.cu:484-714.  Outputs sized n; table_offsets[T] is the number
 * of uniques (device).  Unique order = first occurrence inside each table (deterministic). */
int64_t mi355_segmented_unique_workspace_bytes(int64_t n);
int mi355_segmented_unique(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                           const int64_t* input_frequencies, int count_freq, void* unique_keys,
                           int64_t* output_indices, int64_t* table_offsets, int64_t* freq, void* workspace,
                           int64_t workspace_bytes, hipStream_t stream);
/* Same, and (both nullable) csr_cnt int32[n]: occurrences of unique key u in the batch; csr_rank int32[n]: position of
 * occurrence i inside the list of its unique key (a permutation of 0..cnt-1 per key, order unspecified).  They let
 * mi355_group_by_unique_csr build the backward's CSR with a scan and a scatter -- no histogram pass, no atomics. */
int mi355_segmented_unique_csr(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                               const int64_t* input_frequencies, int count_freq, void* unique_keys,
                               int64_t* output_indices, int64_t* table_offsets, int64_t* freq, int32_t* csr_cnt,
                               int32_t* csr_rank, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* expand_table_ids_cuda, src/unique_op.cu:719-750 */
int mi355_expand_table_ids(const int64_t* offsets, int64_t num_tables, int64_t n, const int64_t* n_dev,
                           int64_t* table_ids, hipStream_t stream);

/* get_table_range, src/index_calculation.cu:77-127: range[t] = offsets[feature_offsets[t] * B] with
 * B = feature_x_batch / feature_offsets[T] computed on the device (feature_x_batch = len(offsets) - 1) */
int mi355_get_table_range(const int64_t* offsets, const int64_t* feature_offsets, int64_t num_tables,
                          int64_t feature_x_batch, int64_t* table_range, hipStream_t stream);

/* flagged_compact, src/index_calculation.cu:129-232 -- order preserving; the count stays on the
 * device in *count_out (the reference syncs the host here).  Arrays are 8-byte words. */
int64_t mi355_flagged_compact_workspace_bytes(int64_t n);
int mi355_flagged_compact(const uint8_t* flags, int64_t n, const int64_t* n_dev, int64_t* count_out,
                          int64_t* out_index, int num_arrays, const void* const* inputs, void* const* outputs,
                          void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* Groups the keys of a batch by unique row (counting sort keyed by the reverse index): stands in for
 * generate_gather_ids_pooled_kernel + cub::DeviceRadixSort of reduce_grads, dynamic_emb_op.cu:140-263.
 * ptr: int32[max_unique+1], csr_src: int32[n] = bag id f*B+b (offsets != NULL) or key position. */
int64_t mi355_group_by_unique_workspace_bytes(int64_t n, int64_t max_unique);
int64_t mi355_hot_rows_workspace_bytes(int64_t num_keys, int64_t dim);
/* hot_workspace (nullable, mi355_hot_rows_workspace_bytes(n, dim) bytes): receives the task list of the
 * rows with more than 16 occurrences; hand the SAME buffer to mi355_backward_fused as `workspace`. */
int mi355_group_by_unique(const int64_t* reverse_indices, int64_t n, const int64_t* offsets, int64_t num_bags,
                          int64_t max_unique, const int64_t* nu_dev, int32_t* ptr, int32_t* csr_src,
                          void* workspace, int64_t workspace_bytes, void* hot_workspace,
                          int64_t hot_workspace_bytes, int64_t dim, hipStream_t stream);
int64_t mi355_group_by_unique_csr_workspace_bytes(int64_t max_unique);
int mi355_group_by_unique_csr(const int32_t* csr_cnt, const int32_t* csr_rank, const int64_t* reverse_indices, int64_t n,
                              const int64_t* offsets, int64_t num_bags, int64_t max_unique, const int64_t* nu_dev,
                              int32_t* ptr, int32_t* csr_src, void* workspace, int64_t workspace_bytes,
                              void* hot_workspace, int64_t hot_workspace_bytes, int64_t dim, hipStream_t stream);

/* block_bucketize_sparse_features, src/sparse_block_bucketize_features.cu:220-350,366-830:
 * dist_type per feature 0 continuous / 1 roundrobin / 2 hash_roundrobin; offsets has num_bags+1 entries
 * (feature-major F*B bags); new_lengths [W*num_bags], new_offsets [W*num_bags+1]. */
int mi355_block_bucketize(int64_t world_size, int64_t num_bags, int64_t batch_size, const int64_t* offsets,
                          const void* indices, const int64_t* block_sizes, const int32_t* dist_type_per_feature,
                          const float* weights, int64_t* new_lengths, int64_t* new_offsets, void* new_indices,
                          float* new_weights, int64_t* unbucketize_permute, hipStream_t stream);
/* The same op with the two remaining arguments of the reference's signature (sparse_block_bucketize_features.cu:366-380):
 * `batch_size_per_feature` -- passed as feature_bag_starts [num_features + 1], the first bag of every feature (its exclusive
 * prefix sum; nullptr: every feature has batch_size bags) -- and `block_bucketize_pos` -- uneven shard boundaries, the sorted
 * boundaries of all features concatenated plus their offsets [num_features + 1] (:262-292, 341-347: rank = last boundary <= idx,
 * new index = idx - that boundary; indices outside the boundaries: idx % W, idx / W; the dist types do not apply). */
int mi355_block_bucketize_ex(int64_t world_size, int64_t num_bags, int64_t batch_size, const int64_t* offsets,
                             const void* indices, const int64_t* block_sizes, const int32_t* dist_type_per_feature,
                             const float* weights, int64_t* new_lengths, int64_t* new_offsets, void* new_indices,
                             float* new_weights, int64_t* unbucketize_permute, const int64_t* feature_bag_starts,
                             int64_t num_features, const int64_t* block_bucketize_pos_concat,
                             const int64_t* block_bucketize_pos_offsets, hipStream_t stream);

/* Glue of the row-wise input dist (dynamicemb/input_dist.py :199-285 + TorchRec KJTAllToAll, third party): exclusive
 * offsets of a lengths vector (offsets has n+1 entries); the keys sent to / received from every peer as differences of
 * the send / receive offsets at the peer boundaries (`splits` [2*W]: send then receive; may be pinned host memory --
 * it is the one host read of the exchange); pseudo-bags of at most `chunk` keys over per-table unique-key lists for
 * the pre-communication dedup (shard/embedding.py:183-275), lengths [T*num_chunks], offsets [T*num_chunks+1]. */
int mi355_exclusive_offsets(const int64_t* lengths, int64_t n, int64_t* offsets, hipStream_t stream);
int mi355_peer_splits(const int64_t* send_offsets, const int64_t* recv_offsets, int64_t bags_per_peer, int64_t world_size,
                      int64_t* splits, hipStream_t stream);
int mi355_chunk_bags(const int64_t* unique_offsets, int64_t num_tables, int64_t chunk, int64_t num_chunks, int64_t* lengths,
                     int64_t* offsets, hipStream_t stream);

/* compute_dedup_lengths_cuda, src/unique_op.cu:753-789 (kernel lookup_kernel.cuh:1049-1090): lengths/offsets that
 * spread each table's unique keys evenly over its (feature, batch) bags.  new_offsets has new_lengths_size+1. */
int mi355_compute_dedup_lengths(const int64_t* unique_offsets, const int64_t* table_offsets_in_feature,
                                int64_t num_tables, int64_t local_batch_size, int64_t new_lengths_size,
                                int64_t* new_lengths, int64_t* new_offsets, hipStream_t stream);

/* segmented_sum_cuda, src/index_calculation.cu:38-75: int32 data, int64 offsets [S+1] -> int64 sums [S]. */
int mi355_segmented_sum(const int32_t* data, const int64_t* offsets, int64_t num_segments, int64_t* out,
                        hipStream_t stream);

/* Bag re-ordering of a received key stream, (source rank, feature, batch) -> (feature, source rank, batch):
 * the recat / permute_2D_sparse_data step of TorchRec's KJTAllToAll (third party), call site
 * dynamicemb/input_dist.py:239-285.  Offsets are exclusive scans of the respective lengths. */
int mi355_permute_lengths(int64_t num_sources, int64_t num_features, int64_t batch_size, const int64_t* in_lengths,
                          int64_t* out_lengths, hipStream_t stream);
/* elem_bytes: bytes per element of the permuted stream (8 for keys, D*sizeof(T) for embedding rows);
 * num_elements: length of the stream (host value, picks the short-bag or long-bag kernel). */
int mi355_permute_bags(int64_t num_sources, int64_t num_features, int64_t batch_size, int64_t elem_bytes,
                       int64_t num_elements, const int64_t* in_offsets, const int64_t* out_offsets, const void* in_keys,
                       void* out_keys, hipStream_t stream);

/* --------------------------------------------------- exchanges of the row-wise sharded lookup ---- */

/* The collectives of a row-wise (model-parallel) sharded step, driven from INSIDE the library -- one call per stage, RCCL on
 * the HIP streams the caller names (csrc/exchange.hip).  Replaces, for GPU tensors, the c10d call sequence of
 * dynamicemb/input_dist.py:199-285 (RwSparseFeaturesDist -> TorchRec KJTAllToAll: lengths all-to-all, keys all-to-all-v, recat)
 * and of the output dists of planner/rw_sharding.py:85-158 (sequence: rows back), :191-261 (pooled: reduce-scatter of partial
 * sums, here all-to-all + local sum; backward all-gather).  RCCL is bound at run time from the librccl.so the process already
 * holds (torch's): mi355_rw_load_rccl(path).  Rank 0 draws two unique ids (128 bytes each) and hands them to every rank by
 * any means (the module broadcasts them over the existing process group); mi355_rw_create is collective. */
int mi355_rw_load_rccl(const char* librccl_path);
int mi355_rw_unique_id(void* out, int64_t bytes);
int mi355_rw_create(const void* id_input_dist, const void* id_output_dist, int world_size, int rank, void** handle);
int mi355_rw_destroy(void* handle);
/* hang guard of the caller's one-time self-check: aborts both communicators (ncclCommAbort: a collective kernel waiting for a
 * peer that never came returns) and frees the handle without a device synchronisation; the caller continues on the c10d
 * sequence.  No reference counterpart (TorchRec relies on ProcessGroupNCCL's watchdog for the same purpose). */
int mi355_rw_abort(void* handle);
/* input dist, first half, on `stream` (behind what `producer_stream` holds): block_bucketize -> all-to-all of the lengths
 * [W][F*B] -> exclusive offsets of the received lengths -> per-peer key counts into pinned memory.  Arrays as
 * mi355_block_bucketize; recv_lengths [W*F*B], recv_offsets [W*F*B+1].  *ticket names the counts for the second half. */
int mi355_rw_input_begin(void* handle, int64_t num_features, int64_t batch_size, const int64_t* offsets, const void* keys,
                         const int64_t* block_sizes, const int32_t* dist_types, int64_t* new_lengths, int64_t* new_offsets,
                         void* new_keys, int64_t* unbucketize_permute, int64_t* recv_lengths, int64_t* recv_offsets,
                         hipStream_t producer_stream, hipStream_t stream, int* ticket);
/* the one host read of the exact exchange: waits for the launch that wrote the counts; send_splits / recv_splits [W] (host),
 * totals[0] = keys sent, totals[1] = keys received (the size of the buffer mi355_rw_input_keys fills) */
int mi355_rw_input_counts(void* handle, int ticket, int64_t* send_splits, int64_t* recv_splits, int64_t* totals);
/* 1 when mi355_rw_input_counts(ticket) would not wait (the launch that writes the counts has completed), else 0 */
int mi355_rw_input_counts_ready(void* handle, int ticket);
/* second half on `stream`: all-to-all-v of the 8-byte keys, recat to feature-major when W > 1 and F > 1 (fm_* buffers, else
 * unused: the received order IS feature-major); `consumer_stream` (if another stream) is ordered behind it */
int mi355_rw_input_keys(void* handle, int ticket, int64_t num_features, int64_t batch_size, const void* new_keys,
                        void* recv_keys, const int64_t* recv_lengths, const int64_t* recv_offsets, int64_t* fm_lengths,
                        int64_t* fm_offsets, void* fm_keys, hipStream_t stream, hipStream_t consumer_stream);
int mi355_rw_wait_keys(void* handle, int ticket, hipStream_t consumer_stream);
/* 1 when the key exchange of `ticket` has completed on the device, else 0 (an event query, never waits) */
int mi355_rw_keys_ready(void* handle, int ticket);
/* gives a ticket back whose second half will not run (at most 8 input dists may be in flight: a 9th mi355_rw_input_begin
 * before the oldest mi355_rw_input_keys is refused) */
int mi355_rw_input_cancel(void* handle, int ticket);
/* pooled output dist: all-to-all of the W blocks of numel_per_block partial sums (wire_dtype) + their fp32 sum -> out */
int mi355_rw_output_pooled(void* handle, const void* send, void* recv, int64_t numel_per_block, int wire_dtype, void* out,
                           int out_dtype, hipStream_t stream);
/* pooled backward: all-gather of `bytes` bytes per rank */
int mi355_rw_allgather(void* handle, const void* send, void* recv, int64_t bytes, hipStream_t stream);
/* sequence output dist / its backward: all-to-all-v of rows; counts in elements of elem_bytes (host arrays [W]) */
int mi355_rw_alltoallv(void* handle, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                       int64_t elem_bytes, hipStream_t stream);

/* ------------------------------------------------------------------------ value ops ---- */

/* out[r] = sum_c in[c][r] (fp32 partial pooled sums of the W shards -> output dtype): the local half of the
 * pooled output dist (TorchRec RwPooledEmbeddingSharding reduce-scatter, planner/rw_sharding.py:191-261). */
int mi355_sum_chunks(const float* in, int64_t chunks, int64_t numel_per_chunk, void* out, int out_dtype,
                     hipStream_t stream);
/* the same for chunks in a 16-bit wire type (TorchRec's qcomm codec on the reduce-scatter: the shards' partial sums cross the
 * fabric in bf16 / fp16, fbgemm_gpu quantize_comm; here the sum reads them directly, fp32 accumulation) */
int mi355_sum_chunks_typed(const void* in, int in_dtype, int64_t chunks, int64_t numel_per_chunk, void* out, int out_dtype,
                           hipStream_t stream);
/* the same with chunk `self_index` read from `self_chunk` instead of `in` (the rank's own block stays where the lookup wrote it:
 * it does not travel through the all-to-all); sum order unchanged */
int mi355_sum_chunks_self(const void* in, int in_dtype, int64_t chunks, int64_t numel_per_chunk, const void* self_chunk,
                          int64_t self_index, void* out, int out_dtype, hipStream_t stream);


/* gather_embedding_pooled, src/dynamic_emb_op.cu:106-133 (kernels lookup_kernel.cuh:859-998).
 * Source: dense [*, src_stride] (reference form) or table rows through row_addr[u] (fused form;
 * address 0 = missing row = zeros).  offsets feature-major [num_bags+1]; combiner 0 sum / 1 mean;
 * D_offsets (int32[F+1]) NULL for uniform dim.  aligned16: rows, dims and column offsets allow
 * 4-element vector access. */
int mi355_gather_pooled(const void* src, int64_t src_stride, const int64_t* row_addr, int src_dtype,
                        const int64_t* reverse_indices, int64_t num_keys, const int64_t* offsets, int64_t num_bags,
                        int64_t batch_size, int combiner, int64_t dim, const int32_t* D_offsets, int64_t total_D,
                        void* dst, int dst_dtype, int aligned16, hipStream_t stream);

/* gather_embedding, src/dynamic_emb_op.cu:79-104 (index NULL = identity) */
int mi355_gather_rows(const void* src, int64_t src_stride, const int64_t* row_addr, int src_dtype,
                      const int64_t* index, int64_t n, const int64_t* n_dev, int64_t dim, void* dst,
                      int64_t dst_stride, int dst_dtype, int aligned16, hipStream_t stream);

/* load_from_flat_table_{contiguous,emb,value} (is_load=1) / store_to_flat_table_{contiguous,value}
 * (is_load=0), src/dynamic_emb_op.cu:294-684.  region 0/1/2 as NumRegions there. */
int mi355_flat_table_copy(int is_load, int region, int64_t n, const int64_t* n_dev, void* dense,
                          int64_t dense_dim, int64_t dense_stride, int dtype, const int64_t* indices,
                          const int64_t* table_ids, int64_t scalar_table_id, const int64_t* table_ptrs,
                          const int64_t* table_value_dims, const int64_t* table_emb_dims, int64_t max_emb_dim,
                          hipStream_t stream);

/* row_addr[u] = table_ptrs[tid] + slot * value_dim * elem_bytes (0 for slot < 0): the address form of
 * the (table_ptrs, table_ids, indices) triple every flat-table kernel of the reference takes. */
int mi355_row_addresses(int64_t n, const int64_t* n_dev, const int64_t* slots, const int64_t* table_ids,
                        const int64_t* table_ptrs, const int64_t* table_value_dims, int elem_bytes,
                        int64_t* row_addr, hipStream_t stream);

/* {uniform,normal,truncated_normal,const,debug}_init, src/initializer.cu:64-212.  mode 0..4. */
int mi355_init_rows(int mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
                    int64_t n, const int64_t* n_dev, const void* keys, const int64_t* sel,
                    const int64_t* row_addr, void* dense, int64_t dense_stride, int dtype, int64_t emb_dim,
                    int64_t value_dim, const uint8_t* results, const uint8_t* skip, const int64_t* table_ids,
                    const int64_t* table_emb_dims, const int64_t* table_value_dims, hipStream_t stream);

/* ------------------------------------------------------------------------- backward ---- */

/* reduce_grads (opt_kind 0, src/dynamic_emb_op.cu:159-285) and reduce_grads fused with
 * {sgd,adam,adagrad,rowwise_adagrad}_update_for_flat_table (opt_kind 1..4, src/optimizer.cu:77-240)
 * over the CSR of mi355_group_by_unique.  combiner -1 sequence / 0 sum / 1 mean.
 * opt_kind 0 writes the reduced gradients to `out` [max_unique, out_stride] and `weight_dtype` is then the dtype
 * of `out` (the reference returns the grad dtype; fp32 keeps the sums exact for the sharded backward). */
int64_t mi355_backward_workspace_bytes(int64_t num_keys, int64_t dim);
int mi355_backward_fused(const int32_t* ptr, const int32_t* csr_src, int64_t num_keys, int64_t max_unique,
                         const int64_t* nu_dev, const void* grads, int64_t grad_stride, int grad_dtype,
                         const int64_t* offsets, const int32_t* D_offsets, int64_t batch_size, int64_t dim,
                         int combiner, const int64_t* row_addr, int weight_dtype, int opt_kind, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t iter_num,
                         int64_t state_offset, int round_grad, void* out, int64_t out_stride, int aligned16,
                         void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* {sgd,adam,adagrad,rowwise_adagrad}_update_for_flat_table (row_addr) / ..._for_padded_buffer
 * (dense_rows), src/optimizer.cu:77-476, on dense unique gradients. */
int mi355_optimizer_update(int opt_kind, const void* grads, int64_t grad_stride, int grad_dtype, int64_t n,
                           const int64_t* n_dev, const int64_t* row_addr, void* dense_rows, int64_t dense_stride,
                           int weight_dtype, int64_t dim, int64_t state_offset, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int64_t iter_num, int aligned16,
                           hipStream_t stream);
/* the padded-buffer form over rows of tables with DIFFERENT embedding widths (update4_padded_buffer_kernel,
 * src/optimizer_kernel.cuh:473-493): row u is table_emb_dims[table_ids[u]] wide, its optimizer state starts at state_offset
 * (= the widest embedding) whatever its own width; table_ids == nullptr: every row is `dim` wide (the entry above). */
int mi355_optimizer_update_tables(int opt_kind, const void* grads, int64_t grad_stride, int grad_dtype, int64_t n,
                                  const int64_t* n_dev, const int64_t* row_addr, void* dense_rows, int64_t dense_stride,
                                  int weight_dtype, int64_t dim, int64_t state_offset, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int64_t iter_num, int aligned16, const int64_t* table_ids,
                                  const int64_t* table_emb_dims, hipStream_t stream);

/* ---------------------------------------------------------------- growable buffers ---- */

/* VMMTensor / HostVMMTensor (src/vmm_tensor.cu:30-585, pybind :555-585): a buffer that grows IN PLACE.  Address space for
 * `reserve_bytes` is reserved once; `initial_bytes` (then every mi355_vmm_extend) maps zero-filled physical memory at the
 * tail -- HBM through hipMemCreate / hipMemMap for host == 0, pinned host memory (mmap + hipHostRegister, addressable by
 * the kernels through the same pointer) for host != 0.  The data pointer never changes. */
int mi355_vmm_create(int64_t reserve_bytes, int64_t initial_bytes, int device, int host, void** handle_out);
int mi355_vmm_extend(void* handle, int64_t new_total_bytes);
void* mi355_vmm_data(void* handle);
int64_t mi355_vmm_mapped_bytes(void* handle);
int64_t mi355_vmm_reserved_bytes(void* handle);
int mi355_vmm_destroy(void* handle);

/* ------------------------------------------------------------------ fused pipelines ---- */

/* One-call forward of BatchedDynamicEmbeddingTablesV2 with HBM-only storage:
 * dynamicemb_prefetch (_prefetch_hbm_direct_path) + DynamicEmbeddingFunction.forward, or
 * dynamicemb_eval_forward when train == 0 (dynamicemb/batched_dynamicemb_function.py:559-932,1042-1191).
 * No host sync; persisted arrays feed mi355_demb_backward. */
int64_t mi355_demb_forward_workspace_bytes(int64_t num_keys, int64_t num_tables);
int mi355_demb_forward(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                       int64_t num_scores, int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel,
                       const int64_t* table_ptrs, const int64_t* table_value_dims, const int64_t* table_emb_dims,
                       int value_dtype, int64_t emb_dim, int64_t value_dim, const void* keys, int64_t num_keys,
                       const int64_t* offsets, int64_t num_bags, int64_t batch_size, const int64_t* feature_offsets,
                       int64_t num_tables, int train, int find_policy, const void* find_scores, int insert_policy,
                       const void* insert_scores, uint64_t timer_override, int pin, int init_mode, float p0,
                       float p1, float p2, float p3, uint64_t seed, float state_init, int combiner,
                       const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype, int aligned16,
                       int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids, int64_t* slots,
                       int64_t* row_addr, int64_t* freq, int32_t* csr_cnt /* nullable */,
                       int32_t* csr_rank /* nullable */,
                       void* backward_workspace /* nullable: early CSR, see mi355_demb_backward(prepared) */,
                       int64_t backward_workspace_bytes,
                       int* join_token /* out, nullable: side-stream join point of the early CSR (-1: none) */,
                       void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* The same step with the index stage FUSED into one kernel (csrc/fused_fwd.hip): tile dedup in LDS, hash-table probe of
 * the tile's distinct keys, per-slot occurrence counting (dedup BY SLOT), in-place insert + first-touch initialisation of
 * unseen keys, deferred min-score eviction for full buckets -- segmented_unique_cuda (src/unique_op.cu:484-714) +
 * table_lookup / table_insert / table_unlock (src/table_operation/kernels.cuh:81-585) + initializer + the pin bracket of
 * _prefetch_hbm_direct_path (dynamicemb/batched_dynamicemb_function.py:559-696) in ONE launch; the pooled / sequence
 * gather reads per-occurrence row addresses and the unique numbering + the backward's CSR run on the library's side
 * stream (use_side_stream) under it.  `aux`: persistent int32 scratch of mi355_demb_aux_numel() elements owned by the
 * table, zero-initialised once by the caller and left all-zero by every call.  Scores are scalar: `score_value`, or the
 * per-key occurrence count when use_count != 0 (LFU).  Unique keys come out in representative order (not first
 * occurrence); in eval mode (train == 0) only `out` is produced.  join_token as in mi355_demb_forward. */
int64_t mi355_demb_aux_numel(int64_t total_slots, int64_t num_buckets);
/* measurement hook of bench.py: while enabled, the library brackets the launches of its two bandwidth kernels with HIP
 * events on the launch stream; mi355_profile_ms(slot) returns the duration of the last one (-1: none) */
int mi355_profile_kernels(int enable);
float mi355_profile_ms(int slot /* 0: gather of the fused forward, 1: backward kernel */);
/* `stream` waits for the side-stream work a forward announced through join_token */
int mi355_side_join(int token, hipStream_t stream);
/* Slot-range partitions the fused forward uses for a training batch of n keys (0: the per-slot-counter path).  One table,
 * 64 K .. 1 M keys and at least 8 buckets per partition take the partitioned index stage: (tile, key) records grouped by
 * slot range, one block per range merges them in LDS -- no global atomic per key, no per-slot scratch. */
int mi355_demb_forward_fused_partitions(int64_t n, int64_t num_tables, int64_t num_buckets);
/* Round 3: a pooled training forward of the partitioned stage (path (c), MI355_FUSED_PART=2) returns *join_token == -2.  Its
 * partition kernel wrote the backward's CSR itself (no scatter kernel); the per-occurrence outputs nothing on the training path reads --
 * reverse_indices (the `inverse` of segmented_unique_cuda, src/unique_op.cu:484-714) and csr_rank -- are produced by this call
 * on demand, from the step's forward workspace (untouched since the forward). */
int mi355_demb_fused_materialize(void* workspace, int64_t workspace_bytes, int64_t num_keys, int64_t num_tables,
                                 const int64_t* row_addr, int64_t* reverse_indices, int32_t* csr_rank, hipStream_t stream);
int64_t mi355_demb_forward_fused_workspace_bytes(int64_t num_keys, int64_t num_tables);
int mi355_demb_forward_fused(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                             int64_t num_scores, int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel,
                             int32_t* aux, int64_t aux_numel, int64_t num_buckets, const int64_t* table_ptrs,
                             const int64_t* table_value_dims, const int64_t* table_emb_dims, int value_dtype,
                             int64_t emb_dim, int64_t value_dim, const void* keys, int64_t num_keys,
                             const int64_t* offsets, int64_t num_bags, int64_t batch_size,
                             const int64_t* feature_offsets, int64_t num_tables, int train, int find_policy,
                             int insert_policy, uint64_t score_value, int use_count, uint64_t timer_override, int pin,
                             int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
                             int combiner, const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype,
                             int aligned16, int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids,
                             int64_t* slots, int64_t* row_addr, int64_t* freq, int32_t* csr_cnt, int32_t* csr_rank,
                             void* backward_workspace, int64_t backward_workspace_bytes, int use_side_stream,
                             int* join_token, void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* Round 6: no step ever loses its update (the reference's unique op serves any key stream in one pass, src/unique_op.cu:484-714;
 * DynamicEmbeddingFunction.backward updates every unique row of every step, batched_dynamicemb_function.py:1193-1300).  Path (c)
 * gives every slot-range partition a fixed record list; a key stream that defeats the hash can flood one.  Such a step's FORWARD
 * output is complete, its CSR is not.  A path-(c) training forward therefore returns *join_token = -(2 + epoch), epoch > 0: its
 * partition kernel stores {epoch, flooded} into pinned host memory as its first action, and the caller asks
 * mi355_demb_fused_step_flooded(epoch, wait_ms) before it uses the step's CSR or unique numbering: 0 = complete; 1 (or 2: the
 * notice was overwritten 64 steps later, treated alike) = flooded: run mi355_demb_forward_fused_rerun (the arguments of the step's
 * forward call + its epoch; `out` untouched), which regroups the step on the per-slot-counter path over the same buffers, then
 * the backward as usual; -1 = the forward has not reached its partition kernel within wait_ms; 3 = a gather block of a
 * sequence lookup abandoned its bounded wait for a partition block of the same launch (the output lacks rows: an error).  The steady state pays one host
 * read of pinned memory per step and no launch.  *join_token == -2 (a forward captured into a hipGraph, pin != 0, or
 * MI355_FUSED_OVERFLOW_RERUN=1): the re-run chain rode behind the gather in the same call, gated on the device; nothing to ask. */
int mi355_demb_fused_step_flooded(int epoch, int wait_ms);
int mi355_demb_forward_fused_rerun(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity,
                                   int64_t num_scores, int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel,
                                   int32_t* aux, int64_t aux_numel, int64_t num_buckets, const int64_t* table_ptrs,
                                   const int64_t* table_value_dims, const int64_t* table_emb_dims, int value_dtype,
                                   int64_t emb_dim, int64_t value_dim, const void* keys, int64_t num_keys,
                                   const int64_t* offsets, int64_t num_bags, int64_t batch_size,
                                   const int64_t* feature_offsets, int64_t num_tables, int train, int find_policy,
                                   int insert_policy, uint64_t score_value, int use_count, uint64_t timer_override, int pin,
                                   int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
                                   int combiner, const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype,
                                   int aligned16, int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids,
                                   int64_t* slots, int64_t* row_addr, int64_t* freq, int32_t* csr_cnt, int32_t* csr_rank,
                                   void* backward_workspace, int64_t backward_workspace_bytes, int use_side_stream,
                                   int epoch, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* Round 5: the pre-bound training step.  Replaces, for the steady-state training step, the per-call argument marshalling of
 * DynamicEmbeddingFunction.forward / backward (batched_dynamicemb_function.py:1042-1300), which re-reads the constructor state of
 * BatchedDynamicEmbeddingTablesV2 (batched_dynamicemb_tables.py:462-787, 999-1088) on every step: a plan holds the table, value
 * buffers, policies, initializer and optimizer of ONE module; a step is then plan_forward(batch, out, step buffer) +
 * plan_backward(step buffer, grads).  The step buffer holds the persisted arrays of mi355_demb_forward_fused and both
 * workspaces; mi355_demb_step_layout is the only definition of its layout (13 int64: ten offsets -- rev, tids, slots, row_addr,
 * freq, csr_cnt, csr_rank, unique_offsets, forward workspace, backward workspace --, total bytes, the two workspace sizes).
 * plan_forward returns 1 (nothing launched) when the buffer is smaller than mi355_demb_plan_step_bytes(num_keys). */
void* mi355_demb_plan_create(void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity, int64_t num_scores,
                             int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel, int32_t* aux, int64_t aux_numel,
                             int64_t num_buckets, const int64_t* table_ptrs, const int64_t* table_value_dims,
                             const int64_t* table_emb_dims, int value_dtype, int64_t emb_dim, int64_t value_dim,
                             const int64_t* feature_offsets, int64_t num_tables, int find_policy, int insert_policy,
                             int use_count, int pin, int init_mode, float p0, float p1, float p2, float p3, uint64_t seed,
                             float state_init, int combiner, const int32_t* D_offsets, int64_t total_D, int out_dtype,
                             int aligned16, int opt_kind, float beta1, float beta2, float eps, float weight_decay);
void mi355_demb_plan_destroy(void* plan);
void mi355_demb_step_layout(int64_t num_keys, int64_t num_tables, int64_t dim, int train, int64_t* out13);
int64_t mi355_demb_plan_step_bytes(void* plan, int64_t num_keys);
int mi355_demb_plan_forward(void* plan, const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags,
                            int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* out, void* step_buf,
                            int64_t step_bytes, int* state, hipStream_t stream);
/* Round 6: the forward in two halves -- the reference's prefetch pipeline (BatchedDynamicEmbeddingTablesV2.prefetch,
 * batched_dynamicemb_tables.py:1090-1137; PrefetchTrainPipelineSparseDist, train_pipeline.py:533-692) on the fast index path.
 * stage 1 = index stage only (issued for batch k + 1 on a side stream under the backward of batch k), stage 2 = the gather of a
 * step whose stage 1 ran earlier, stage 0 = mi355_demb_plan_forward.  protect_score: evictions of this call spare every slot whose
 * score is >= it (what the reference's ref-counter pins achieve; exact for the recency policies STEP / TIMESTAMP).  Returns 3 with
 * nothing launched when the batch is not eligible for the partitioned index path. */
int mi355_demb_plan_stage(void* plan, int stage, uint64_t protect_score, const void* keys, int64_t num_keys, const int64_t* offsets,
                          int64_t num_bags, int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* out,
                          void* step_buf, int64_t step_bytes, int* state, hipStream_t fork_from, int slot, hipStream_t stream);
/* (slot 0..3: the stream order of the staged step is kept by the library -- stage 1 on `stream` starts behind what `fork_from`
 *  holds at the call and marks its end, stage 2 waits for the mark; slot -1: the caller orders its streams itself) */
int mi355_demb_plan_backward(void* plan, void* step_buf, int64_t step_bytes, int64_t num_keys, const int64_t* offsets,
                             int64_t num_bags, int64_t batch_size, const void* grads, int64_t grad_stride, int grad_dtype,
                             int grad_aligned16, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t iter_num, int prepared, int epoch, hipStream_t stream);
/* (epoch > 0: the step's overflow notice is read first; a flooded step returns 2 with nothing launched -- call
 *  mi355_demb_plan_rerun(the batch of the step's plan_forward call, its epoch = -2 - *state) and then plan_backward with epoch 0) */
int mi355_demb_plan_rerun(void* plan, const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags,
                          int64_t batch_size, uint64_t score_value, uint64_t timer_override, void* step_buf,
                          int64_t step_bytes, int epoch, hipStream_t stream);

/* hipStream_t of the library's side stream (early CSR build of mi355_demb_forward); NULL if it cannot be created */
void* mi355_early_csr_stream(void);

/* One-call backward: DynamicEmbeddingFunction.backward (batched_dynamicemb_function.py:1193-1300):
 * reduce_grads + optimizer.fused_update_for_flat_table + decrement_counter.  Early CSR: when the forward was given
 * `backward_workspace`, the key-grouping half of this call already ran on the library's side stream under the forward's
 * own lookup / gather kernels; pass the same buffer with prepared = 2 + join_token (1: the grouping was issued on this
 * very stream, nothing to join) and only the reduce + optimizer kernel remains. */
int64_t mi355_demb_backward_workspace_bytes(int64_t num_keys, int64_t dim);
int mi355_demb_backward(const int64_t* reverse_indices, int64_t num_keys, const int64_t* unique_offsets,
                        int64_t num_tables, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
                        const void* grads, int64_t grad_stride, int grad_dtype, const int32_t* D_offsets,
                        int64_t dim, int combiner, const int64_t* row_addr, int value_dtype, int opt_kind, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int64_t iter_num,
                        int64_t state_offset, int round_grad, int aligned16, int32_t* counter,
                        int64_t counter_numel, const int64_t* slots, const int64_t* table_ids,
                        const int64_t* table_bucket_offsets, int64_t bucket_capacity, int unpin,
                        const int32_t* csr_cnt, const int32_t* csr_rank /* from the forward, or both NULL */,
                        int prepared /* 0 group here; 1 / 2 + token: workspace == the forward's backward_workspace */,
                        void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* ---------------------------------------------------------------- HSTU jagged attention ---- */

/* hstu_varlen_fwd (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:335-523; torch.ops.fbgemm.hstu_varlen_fwd_80,
 * examples/hstu/ops/fused_hstu_op.py:318-337): out = M * SiLU(alpha q k^T) v / scaling_seqlen per jagged
 * sequence.  q, k, v, out: bf16 [total, H, d], element strides between tokens / heads given explicitly
 * (last dim contiguous); cu_seqlens int32 [batch+1] (shared by q and k); num_contexts / num_targets
 * int32 [batch] or NULL; causal = window (-1, 0), else full (-1, -1); head_dim in {32, 64, 128, 256}. */
int mi355_hstu_attn_fwd(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                        int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                        int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                        const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim,
                        int64_t max_seqlen, const int32_t* num_contexts, const int32_t* num_targets,
                        int64_t target_group_size, int causal, float alpha, float scaling_seqlen,
                        hipStream_t stream);

/* Inference forward of the same op: the queries may be the LAST tokens of a longer key sequence (cu_seqlens_k;
 * the reference's delta-q) and the history keys / values may live in a paged cache [num_pages, 2, page_size, H, d]
 * addressed through (page_offsets [batch+1], page_ids, last_page_lens [batch]) -- hstu_attn_varlen_func(...,
 * kv_cache=, page_offsets=, page_ids=, last_page_lens=) of examples/hstu/modules/paged_hstu_infer_layer.py:492-514,
 * kernel hstu_fwd.h Paged_KV paths :104-131,516-545,785; reference statement _hstu_paged_kv_attention,
 * examples/hstu/test/test_paged_hstu_attn_kernel.py:179-256.  With a cache, k / v hold [new history | candidates]
 * per sequence and only their candidate rows are read (the history, new tokens included, is in the cache);
 * cu_seqlens_k[b+1]-cu_seqlens_k[b] = cached length + num_targets[b].  NULL cu_seqlens_k / kv_cache = training call. */
int mi355_hstu_attn_fwd_kv(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                           int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                           int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads,
                           int64_t head_dim, int64_t max_seqlen_q, const int32_t* num_contexts,
                           const int32_t* num_targets, int64_t target_group_size, int causal, float alpha,
                           float scaling_seqlen, const void* kv_cache, const int32_t* page_offsets,
                           const int32_t* page_ids, const int32_t* last_page_lens, int64_t page_size,
                           hipStream_t stream);

/* mi355_hstu_attn_fwd_kv under a local attention window (hstu_attn_varlen_func(window_size=(left, right)) with cu_seqlens_k longer
 * than cu_seqlens_q and / or kv_cache; the reference composes Is_local with the delta-q offset and Paged_KV, hstu_fwd.h:104-131,
 * 463-470,516-545): the window runs over absolute positions, query r of a sequence sitting at Lk - Lq + r.  No contextual /
 * target rows (hstu_attn_interface.py:238-245). */
int mi355_hstu_attn_fwd_kv_window(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                                  int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                                  int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                                  const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads,
                                  int64_t head_dim, int64_t max_seqlen_q, int64_t window_left, int64_t window_right,
                                  float alpha, float scaling_seqlen, const void* kv_cache, const int32_t* page_offsets,
                                  const int32_t* page_ids, const int32_t* last_page_lens, int64_t page_size,
                                  hipStream_t stream);
int mi355_hstu_attn_fwd_kv_window_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                                      int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                                      int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                                      const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads,
                                      int64_t head_dim, int64_t max_seqlen_q, int64_t window_left, int64_t window_right,
                                      float alpha, float scaling_seqlen, const void* kv_cache, const int32_t* page_offsets,
                                      const int32_t* page_ids, const int32_t* last_page_lens, int64_t page_size,
                                      hipStream_t stream);

/* mi355_hstu_attn_fwd_kv with a relative attention bias (hstu_attn_varlen_func(rab=...) with cu_seqlens_k longer than
 * cu_seqlens_q and / or kv_cache; hstu_fwd.h:104-131,516-545 with Has_rab): rab [batch][heads or 1][max_seqlen_k][max_seqlen_k]
 * is indexed by ABSOLUTE positions -- query r of a sequence is row Lk - Lq + r.  Mask as window_size: (-1, 0) causal
 * (num_contexts / num_targets allowed), (-1, -1) full, otherwise a local window. */
int mi355_hstu_attn_fwd_kv_rab(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                               int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                               int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q,
                               const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads, int64_t head_dim,
                               int64_t max_seqlen_q, int64_t max_seqlen_k, const int32_t* num_contexts,
                               const int32_t* num_targets, int64_t target_group_size, int64_t window_left,
                               int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                               int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, const void* kv_cache,
                               const int32_t* page_offsets, const int32_t* page_ids, const int32_t* last_page_lens,
                               int64_t page_size, hipStream_t stream);
int mi355_hstu_attn_fwd_kv_rab_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                               int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                               int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q,
                               const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads, int64_t head_dim,
                               int64_t max_seqlen_q, int64_t max_seqlen_k, const int32_t* num_contexts,
                               const int32_t* num_targets, int64_t target_group_size, int64_t window_left,
                               int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                               int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, const void* kv_cache,
                               const int32_t* page_offsets, const int32_t* page_ids, const int32_t* last_page_lens,
                               int64_t page_size, hipStream_t stream);

/* Arbitrary mask functions (`func` of hstu_attn_varlen_func, hstu_api.cpp:170-180; applied per element inside the reference's
 * kernels, hstu_fwd.h:139-145, 493-556 / hstu_bwd.h) read INSIDE the kernels: int32 func[heads or 1][n_func][>= total_q] (n_func
 * odd; head stride 0 = one set for all heads; func_bound_stride = elements between the bounds of a token; last dimension
 * contiguous).  Query token t sees key position j of its sequence iff j < func[0][t] or func[2p-1][t] <= j < func[2p][t], p >= 1;
 * a masked pair gets func_neg (< 0, finite in the operand type: -1e9 bf16, -6e4 fp16) added to q.k, so SiLU and SiLU' are exactly
 * 0 there.  Mask arguments as the *_rab entry points (the other masks apply on top).  Forward: training keys, delta-q keys and
 * the paged cache; backward: self attention over contiguous keys.  No [batch, heads, N, N] tensor exists anywhere. */
int mi355_hstu_attn_fwd_kv_func(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                                int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                                int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                                int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                                int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                                const void* kv_cache, const int32_t* page_offsets, const int32_t* page_ids,
                                const int32_t* last_page_lens, int64_t page_size, hipStream_t stream);
int mi355_hstu_attn_fwd_kv_func_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                                int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                                int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                                int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                                int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                                const void* kv_cache, const int32_t* page_offsets, const int32_t* page_ids,
                                const int32_t* last_page_lens, int64_t page_size, hipStream_t stream);
/* (round 6) func_workspace (nullable): room for the tables of the tile skipping -- 72 bytes x (total_tokens / 128 + batch + 1)
 * per function set (heads, or 1 when the head stride is 0); the reference derives the valid key blocks from the func extents the
 * same way (hstu_fwd.h:80-83, 139-151, 228-291, 411-412).  NULL / too small: the key-major passes visit every query tile.
 * workspace / workspace_bytes (nullable): the P / dS exchange scratch of mi355_hstu_attn_bwd (same sizing calls); used for functions
 * of up to two bands (n_func <= 5) at head_dim 256 when func_workspace is given, otherwise the recomputing passes run. */
int mi355_hstu_attn_bwd_func(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                             int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                             int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                             const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                             const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                             int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                             int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                             void* func_workspace, int64_t func_workspace_bytes, void* workspace, int64_t workspace_bytes,
                             hipStream_t stream);
int mi355_hstu_attn_bwd_func_f16(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                             int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                             int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                             const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                             const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                             int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                             int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                             void* func_workspace, int64_t func_workspace_bytes, void* workspace, int64_t workspace_bytes,
                             hipStream_t stream);

/* append_kvcache (torch.ops.paged_kvcache_ops.append_kvcache, examples/commons/ops/cuda_ops/csrc/
 * paged_kvcache_ops_kernel.cu:106-140, call site paged_hstu_infer_layer.py:350-364): new-history token i (i < *nnz_dev,
 * or < max_nnz when max_nnz > 0) of sequence batch_indices[i] is written at position positions[i] of that user's
 * cache (page kv_indices[kv_indptr[batch] + pos / page_size], slot pos % page_size; NHD layout); its source row in
 * append_key / append_value is i + seqlen_offsets[batch].  nnz_upper bounds the launch when the count is on the device. */
int mi355_append_kvcache(void* kv_cache, const int32_t* kv_indices, const int32_t* kv_indptr, int64_t num_heads,
                         int64_t head_dim, int64_t page_size, const void* append_key, const void* append_value,
                         int64_t k_row_stride, int64_t v_row_stride, int64_t k_head_stride, int64_t v_head_stride,
                         const int32_t* batch_indices, const int32_t* positions, const int32_t* seqlen_offsets,
                         const int32_t* nnz_dev, int64_t max_nnz, int64_t nnz_upper, hipStream_t stream);

/* hstu_varlen_bwd (hstu_api.cpp:525-719; hstu_varlen_bwd_80, fused_hstu_op.py:682-706): dq, dk, dv
 * contiguous bf16 [total, H, d].  Deterministic (two passes, no atomics): `deterministic` of the
 * reference is always on. */
int64_t mi355_hstu_attn_bwd_workspace_bytes(int64_t total_tokens, int64_t num_heads, int64_t head_dim);
/* Local (sliding window) attention: window_size = (left, right) of hstu_attn_varlen_func (hstu_attn_interface.py:196-222,
 * hstu_api.cpp:154-165): query i sees keys i - left .. i + right, a negative side is unbounded ((-1, 0) = causal,
 * (-1, -1) = full).  No contextual / target rows with a window; self attention only. */
int mi355_hstu_attn_fwd_window(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                               int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                               int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens,
                               int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen, int64_t window_left,
                               int64_t window_right, float alpha, float scaling_seqlen, hipStream_t stream);
int mi355_hstu_attn_bwd_window(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                               int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                               int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                               const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim,
                               int64_t max_seqlen, int64_t window_left, int64_t window_right, float alpha,
                               float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* Relative attention bias: `rab` / `has_drab` of hstu_attn_varlen_func (hstu_attn_interface.py:185-279; varlen_fwd /
 * varlen_bwd hstu_api.cpp:100-111,253-263,417-430,659-667).  rab: bf16 [batch][heads or 1][max_seqlen][max_seqlen] by its
 * batch / head / row strides in elements (head stride 0: one matrix for all heads), added to q_i . k_j before alpha and
 * SiLU.  Mask as window_size of hstu_attn_varlen_func: (-1, 0) causal (num_contexts / num_targets allowed), (-1, -1) full,
 * otherwise a local window.  Self attention over contiguous keys.  drab (nullable): bf16, one matrix per head with its own
 * strides, zero-filled by the caller; receives d loss / d rab at every position inside the sequences. */
int mi355_hstu_attn_fwd_rab(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                            int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                            int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens, int64_t batch,
                            int64_t num_heads, int64_t head_dim, int64_t max_seqlen, const int32_t* num_contexts,
                            const int32_t* num_targets, int64_t target_group_size, int64_t window_left, int64_t window_right,
                            float alpha, float scaling_seqlen, const void* rab, int64_t rab_batch_stride,
                            int64_t rab_head_stride, int64_t rab_row_stride, hipStream_t stream);
int mi355_hstu_attn_bwd_rab(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                            int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                            int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                            const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                            const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                            int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                            int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, void* drab,
                            int64_t drab_batch_stride, int64_t drab_head_stride, int64_t drab_row_stride, hipStream_t stream);
/* Optional scratch of mi355_hstu_attn_bwd: given a 16-byte aligned workspace of at least this many bytes, the dK pass
 * hands dS to the dQ pass through it (bf16, B * H * ceil(max_seqlen/32)^2 sub-tiles of 2 KB) and the dQ pass skips the
 * S / dP recomputation (head_dim >= 128: P travels too and the dV pass becomes one GEMM); with a smaller (or no)
 * workspace the passes recompute.  0: exchange switched off.
 * Results are bit-identical either way (the dQ GEMM consumes the same bf16 dS). */
int64_t mi355_hstu_attn_bwd_ds_bytes(int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen);
/* The same exchange under a byte cap: the workspace holds one chunk of (sequence, head) units at a time, each with the
 * ceil(L_b / 32)^2 tiles of its own length (scratch by the jagged sum, not B x max_seqlen^2), and mi355_hstu_attn_bwd walks
 * the chunks -- the reference's backward needs O(T) memory (hstu_bwd.h:687-729: dQ by atomics, no score-sized buffer); this
 * keeps the one-GEMM dV / dQ passes of the exchange with a bounded buffer.  Returns the workspace bytes to allocate
 * (<= cap_bytes, 256-byte aligned base required; 0 = no exchange fits, the recomputing passes run).  total_tokens: rows of q;
 * plain_causal: causal mask without contextual rows, window or bias -- the sub-tiles above the diagonal are then not stored. */
int64_t mi355_hstu_attn_bwd_ds_bytes_capped(int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                                            int64_t total_tokens, int64_t cap_bytes, int plain_causal);
/* total tokens of the NEXT mi355_hstu_attn_bwd call on this thread (its signature is the reference's hstu_varlen_bwd and
 * does not carry them): bounds the number of chunk passes; optional. */
void mi355_hstu_attn_bwd_hint_tokens(int64_t total_tokens);
/* rows of q of the NEXT mi355_hstu_attn_fwd / _fwd_kv / _fwd_window call on this thread (optional): batch x max_seqlen rows
 * = a dense batch, which the head-dim-256 forward runs with two row blocks (the z-th heaviest and z-th lightest of a column)
 * per workgroup.  The fp16 entry points have their own hint (suffix _f16). */
void mi355_hstu_attn_fwd_hint_tokens(int64_t total_tokens);
void mi355_hstu_attn_fwd_hint_tokens_f16(int64_t total_tokens);
int mi355_hstu_attn_bwd(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                        int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                        int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride,
                        int64_t do_head_stride, const int32_t* cu_seqlens, int64_t batch, int64_t num_heads,
                        int64_t head_dim, int64_t max_seqlen, const int32_t* num_contexts,
                        const int32_t* num_targets, int64_t target_group_size, int causal, float alpha,
                        float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* fp16 operands (hstu_api.cpp:359-366: "HSTU only supports fp16 and bf16"): the same seven entry points for q / k / v / out /
 * gradients (and rab) in IEEE half instead of bf16 -- same kernels compiled with v_mfma_f32_32x32x16_f16, fp32
 * accumulation, round-to-nearest-even packing.  The size queries and append_kvcache are type-agnostic. */
int mi355_hstu_attn_fwd_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                        int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                        int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                        const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim,
                        int64_t max_seqlen, const int32_t* num_contexts, const int32_t* num_targets,
                        int64_t target_group_size, int causal, float alpha, float scaling_seqlen,
                        hipStream_t stream);
int mi355_hstu_attn_fwd_kv_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                           int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                           int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads,
                           int64_t head_dim, int64_t max_seqlen_q, const int32_t* num_contexts,
                           const int32_t* num_targets, int64_t target_group_size, int causal, float alpha,
                           float scaling_seqlen, const void* kv_cache, const int32_t* page_offsets,
                           const int32_t* page_ids, const int32_t* last_page_lens, int64_t page_size,
                           hipStream_t stream);
int mi355_hstu_attn_bwd_f16(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                        int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                        int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride,
                        int64_t do_head_stride, const int32_t* cu_seqlens, int64_t batch, int64_t num_heads,
                        int64_t head_dim, int64_t max_seqlen, const int32_t* num_contexts,
                        const int32_t* num_targets, int64_t target_group_size, int causal, float alpha,
                        float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int mi355_hstu_attn_fwd_window_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                               int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                               int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens,
                               int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen, int64_t window_left,
                               int64_t window_right, float alpha, float scaling_seqlen, hipStream_t stream);
int mi355_hstu_attn_bwd_window_f16(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                               int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                               int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                               const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim,
                               int64_t max_seqlen, int64_t window_left, int64_t window_right, float alpha,
                               float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int mi355_hstu_attn_fwd_rab_f16(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                            int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                            int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens, int64_t batch,
                            int64_t num_heads, int64_t head_dim, int64_t max_seqlen, const int32_t* num_contexts,
                            const int32_t* num_targets, int64_t target_group_size, int64_t window_left, int64_t window_right,
                            float alpha, float scaling_seqlen, const void* rab, int64_t rab_batch_stride,
                            int64_t rab_head_stride, int64_t rab_row_stride, hipStream_t stream);
int mi355_hstu_attn_bwd_rab_f16(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                            int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                            int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                            const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                            const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                            int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                            int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, void* drab,
                            int64_t drab_batch_stride, int64_t drab_head_stride, int64_t drab_row_stride, hipStream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* RECSYS_AMD_H_ */
