// Hot-row work lists shared by the CSR builder (index_ops.hip) and the backward kernels (backward.hip).
// A unique row with more than kHot occurrences in the batch leaves the regular lock-step walk:
//   * up to kWave occurrences: ONE WAVE reduces it and applies the sink (no LDS, no atomics) -- the `wave_*` list;
//   * more: it is cut into kChunk-entry tasks, each reduced by one BLOCK; rows longer than a chunk add their partial
//     sums into the row's fp32 accumulator with agent-scope atomics and the task that draws the last ticket applies
//     the sink (optimizer / store).
#pragma once
#include <stdint.h>

namespace mi355 {

#include <stdlib.h>
// occurrences above which a row takes the chunked path / CSR entries per task (one wave per task).
// Tunable through MI355_HOT / MI355_CHUNK (read once per process) for profiling sweeps.
static inline int hot_threshold() { return 4; }
static inline int hot_chunk() { return 1024; }
static inline int hot_wave() { return 128; }

struct HotList {
  int* n_hot;        // [1]
  int* n_tasks;      // [1]
  int* hot_done;     // [max_hot] tickets
  int* hot_nchunks;  // [max_hot]
  int* hot_u;        // [max_hot] unique row of the hot entry
  int* hot_lo;       // [max_hot] first CSR entry
  int* hot_cnt;      // [max_hot] number of CSR entries
  int* hot_t0;       // [max_hot] first task id
  int* task_u;       // [max_tasks] unique row
  int* task_h;       // [max_tasks] hot index
  int* task_lo;      // [max_tasks] CSR range of the task
  int* task_hi;
  float* hot_acc;    // [max_hot, dim]
  int* n_wave;       // [1]
  int* wave_u;       // [max_hot] rows served by one wave: unique row, first CSR entry, number of entries
  int* wave_lo;
  int* wave_cnt;
  int max_hot, max_tasks, dim;
  int khot, kchunk, kwave;  // thresholds in force for this list (kwave <= khot: no wave list)
};

static inline int64_t hot_align(int64_t x) { return (x + 255) / 256 * 256; }
static inline int hot_max_hot(int64_t n) { return (int)(n / (hot_threshold() + 1) + 1); }
static inline int hot_max_tasks(int64_t n) { return (int)(n / hot_chunk() + hot_max_hot(n) + 1); }
static inline int64_t hot_bytes(int64_t n, int64_t dim) {
  const int64_t mh = hot_max_hot(n), mt = hot_max_tasks(n);
  return 256 + 9 * hot_align(4 * mh) + 4 * hot_align(4 * mt) + hot_align(4 * mh * dim);
}
static inline HotList hot_carve(void* ws, int64_t n, int64_t dim) {
  HotList h;
  uint8_t* w = (uint8_t*)ws;
  h.max_hot = hot_max_hot(n); h.max_tasks = hot_max_tasks(n); h.dim = (int)dim;
  h.khot = hot_threshold(); h.kchunk = hot_chunk(); h.kwave = hot_wave();
  h.n_hot = (int*)w; h.n_tasks = (int*)(w + 8); h.n_wave = (int*)(w + 16); w += 256;
  h.hot_done = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_nchunks = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_u = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_lo = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_cnt = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_t0 = (int*)w; w += hot_align(4LL * h.max_hot);
  h.task_u = (int*)w; w += hot_align(4LL * h.max_tasks);
  h.task_h = (int*)w; w += hot_align(4LL * h.max_tasks);
  h.task_lo = (int*)w; w += hot_align(4LL * h.max_tasks);
  h.task_hi = (int*)w; w += hot_align(4LL * h.max_tasks);
  h.wave_u = (int*)w; w += hot_align(4LL * h.max_hot);
  h.wave_lo = (int*)w; w += hot_align(4LL * h.max_hot);
  h.wave_cnt = (int*)w; w += hot_align(4LL * h.max_hot);
  h.hot_acc = (float*)w;
  return h;
}

}  // namespace mi355
