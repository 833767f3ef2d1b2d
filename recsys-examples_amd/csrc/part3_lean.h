// Round 5: the partition kernel of path (c) as a LEAN 256-thread block that rides in the gather's launch.  Included by fused_fwd.hip.
//
// The training forward's chain was probe -> partition -> gather, each waiting for the one before; but the gather reads only what the
// PROBE wrote (per-occurrence row addresses) -- the partition kernel's output (unique numbering, CSR) is the backward's input.  The
// partition kernel is a latency chain (record loads, LDS merge, scan, look-back, outputs: ~12 us of block life, 20 us as a kernel with
// its boundary) that leaves the chip's memory system idle; the gather is the opposite.  Round 3 tried them as one launch with the
// 1 024-thread partition blocks and lost (54-60 us against 21 + 30 apart): such a block takes a CU's whole register file, so the
// gather's waves had to wait for it, and the kernel's register allocation (the maximum over both roles) halved the gather's
// occupancy.  Here the partition block is rebuilt to fit NEXT TO the gather: 256 threads, <= 64 registers (the gather's own
// allocation: every per-record and per-entry quantity lives in LDS or in the record's output entry, nothing in register arrays),
// ~45 KB of LDS.  The first P blocks of the grid are partition blocks -- one per CU at P = 256, taking 4 of its 32 wave slots --, the
// rest the gather's.  The partition work is slower than on 1 024 threads (four rounds of 256 records instead of one) and nobody
// waits for it: it finishes under the gather.
// MEASURED (tools/ab_probe_c.py --var MI355_PART_FUSED, profiles/r05_part_fused.txt): sequence lookups gain (8 x 16 K tokens
// 0.0761 -> 0.0679 ms: the row copy is indifferent to what shares its CU), the POOLED C2 step loses (0.1227 -> 0.1288 ms): next to
// 24 streaming waves per CU every round trip of the partition block's chain takes 2-3x as long -- its life goes from 12 to 38 us
// (50 for the slowest block) and the launch ends with it, 54 us against 21 + 30 apart; raising the role's wave priority changes
// nothing, and with the gather blocks FIRST in the grid a batch whose keys all go through the eviction would deadlock on the
// ready flags.  Hence: on by default for sequence lookups only (MI355_PART_FUSED: 0 off, 1 sequence, 2 pooled as well).
// A gather lane that meets an occurrence whose key went through the partition block's eviction (address word 1: bucket full) waits for
// that partition's ready flag (LateRefs::ready); partition blocks are dispatched ahead of every gather block, so the wait cannot deadlock.
//
// Restates (reference, corelib/dynamicemb/): segmented_unique_cuda's numbering (src/unique_op.cu:484-714) + the key grouping of
// reduce_grads (src/dynamic_emb_op.cu:159-285) + table_insert_and_evict for deferred keys (src/table_operation/kernels.cuh:226-287).
#pragma once

namespace mi355 {

constexpr int kP3lThreads = 512;

template <int CAP>
__device__ __forceinline__ void part3_lean(FusedArgs& a, const EmitOut& o, int* __restrict__ ptr, int* __restrict__ csr_src, const HotList& hot,
                                           int p) {
  constexpr int HASH = CAP, T = kP3lThreads, kEnt = HASH / T, kSubCapL = CAP / kPartSub, kDefMax = CAP / 4;
  __shared__ int h_slot[HASH], h_cnt[HASH], h_pl[HASH];
  __shared__ unsigned short h_lid[HASH];
  __shared__ int d_rec[kDefMax], d_ent[kDefMax], d_base[kDefMax];
  __shared__ int s_lock[256];
  __shared__ unsigned s_late[HASH / 32];
  __shared__ int s_nd, s_nbig;
  __shared__ unsigned s_fresh[kDefMax / 32];          // deferred records that took a fresh slot (row initialised by the block, part_evict)
  constexpr int kBigMax = 256;
  __shared__ int b_pos[kBigMax], b_ref[kBigMax], b_cnt[kBigMax];
  const int tid = (int)threadIdx.x;
  const int64_t rec_base = (int64_t)p * CAP;
  // a latency chain next to 24 streaming waves of the gather: its few instructions go first (MI355_PART_PRIO=0 in FusedArgs::dbg bit 2 turns it off)
  if (!(a.dbg & 4)) __builtin_amdgcn_s_setprio(3);
  QST(0);
  if (a.notice && p == 0 && tid == 0) publish_notice(a);
  const int mv = a.pcount[p * kPartSub + (tid & (kPartSub - 1))];
  for (int i = tid; i < HASH; i += T) { h_slot[i] = -1; h_cnt[i] = 0; }
  if (tid < 256) s_lock[tid] = 0;
  if (tid < HASH / 32) s_late[tid] = 0;
  if (tid == 0) { s_nd = 0; s_nbig = 0; }
  if (tid < kDefMax / 32) s_fresh[tid] = 0;
  __syncthreads();
  QST(1);
  int c1, c2, c3, total;
  {
    const int lim = a.mt ? CAP : kSubCapL;       // (several tables: one list per partition)
    int m0 = __builtin_amdgcn_readlane(mv, 0), m1 = __builtin_amdgcn_readlane(mv, 1), m2 = __builtin_amdgcn_readlane(mv, 2),
        m3 = __builtin_amdgcn_readlane(mv, 3);
    m0 = m0 < lim ? m0 : lim;
    if (a.mt) { m1 = 0; m2 = 0; m3 = 0; }
    m1 = m1 < lim ? m1 : lim; m2 = m2 < lim ? m2 : lim; m3 = m3 < lim ? m3 : lim;
    c1 = m0; c2 = c1 + m1; c3 = c2 + m2; total = c3 + m3;
  }
  if (tid < kPartSub) a.pcount[p * kPartSub + tid] = 0;       // clean for the next step
  auto rec_index = [&](int f) -> int {            // flat index -> index inside the partition's list
    return f < c1 ? f : (f < c2 ? kSubCapL + f - c1 : (f < c3 ? 2 * kSubCapL + f - c2 : 3 * kSubCapL + f - c3));
  };
  int tbl = 0;
  bool first_of_table = p == 0;
  if (a.mt) { tbl = a.ptab[p]; first_of_table = p == 0 || a.ptab[p - 1] != tbl; }
  const int64_t tp0 = a.table_ptrs[tbl], rowb = a.table_value_dims[tbl] * a.elem_bytes, s0 = a.tbo[tbl] * a.t.C;
  // ---- merge pass: the records of a slot meet in its hash entry; every record learns its entry and the rank base of its tile;
  //      the record that claims an entry leaves the position of the unique row's key there
  {
    uint4 nxt = a.rec[rec_base + rec_index(tid < total ? tid : 0)];
    for (int f0 = 0; f0 < total; f0 += T) {
      const int f = f0 + tid;
      const uint4 rc = nxt;
      const int fn = f + T;
      nxt = a.rec[rec_base + rec_index(fn < total ? fn : 0)];
      if (f >= total) continue;
      const int idx = rec_index(f);
      const int sl = (int)rc.z, cn = (int)rc.w;
      int en = 0, bs = 0;
      if (sl >= 0) {
        bool cl;
        en = p2_insert<HASH>(h_slot, sl, &cl);
        if (cl) h_pl[en] = (int)rc.x;
        bs = atomicAdd(&h_cnt[en], cn);
      } else {
        const int dj = atomicAdd(&s_nd, 1);
        if (dj < kDefMax) {
          d_rec[dj] = idx;                         // (entry and rank base come from the eviction below)
        } else {                                   // beyond what one step evicts for: no slot this step (like an insert that returns Busy)
          bool cl;
          en = p2_insert<HASH>(h_slot, (int)a.S, &cl);
          if (cl) h_pl[en] = (int)rc.x;
          bs = atomicAdd(&h_cnt[en], cn);
          a.rec[rec_base + idx].z = (uint32_t)a.S;
          a.rec[rec_base + idx].w = (uint32_t)(cn | kRecLate);
        }
      }
      a.rec_out4[rec_base + idx] = make_int4(en, bs, 0, 0);
    }
  }
  QST(2);
  __syncthreads();
  QST(3);
  const int nd = s_nd < kDefMax ? s_nd : kDefMax;
  if (nd > 0) {      // (block uniform)
    part_evict<HASH, int, T>(a, nd, rec_base, d_rec, d_ent, d_base, s_lock, s_late, h_slot, h_cnt, tbl, tp0, rowb, s0, nullptr, nullptr, s_fresh);
    __syncthreads();
    for (int e = tid; e < nd; e += T) {
      // the first record (rank base 0) of an entry created by the eviction owns the unique row's key
      const int ent = d_ent[e], bs = d_base[e];
      if (((s_late[ent >> 5] >> (ent & 31)) & 1u) && bs == 0) h_pl[ent] = (int)a.rec[rec_base + d_rec[e]].x;
      a.rec_out4[rec_base + d_rec[e]] = make_int4(ent, bs, 0, 0);
    }
    __syncthreads();
  }
  if (a.part_ready) {   // the gather of this launch may read the evicted-into slots out of the records now
    if (tid == 0) {
      if (nd > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_store(&a.part_ready[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- one scan over the hash ENTRIES (entry order = unique order), two passes over a thread's kEnt entries so that nothing
  //      per entry stays in registers: sums first, then -- with the block prefix -- local unique id and occurrence prefix
  const bool hots = hot.n_tasks != nullptr;
  int v5[5] = {0, 0, 0, 0, 0}, tot5[5];
  for (int k = 0; k < kEnt; ++k) {
    const int e = tid * kEnt + k;
    if (h_slot[e] == -1) continue;
    const int c = h_cnt[e];
    ++v5[0];
    v5[1] += c;
    if (hots && c > hot.khot && c <= hot.kwave) ++v5[4];
    else if (hots && c > hot.khot) { ++v5[2]; v5[3] += (c + hot.kchunk - 1) / hot.kchunk; }
  }
  unsigned long long* tb = a.tstat + a.P;
  block_scan5<T>(v5, tot5, [&](const int (&t5)[5]) {
    stat_store(a.tstat + p, kStatAgg | ((unsigned long long)t5[0] << 31) | (unsigned)t5[1]);
    stat_store(tb + p, kStatAgg | ((unsigned long long)t5[2] << 40) | ((unsigned long long)t5[3] << 20) | (unsigned long long)t5[4]);
  });
  const int nu = tot5[0], tot2 = tot5[1], th = tot5[2], tt = tot5[3], tw = tot5[4];
  QST(4);
  unsigned long long pre_a = 0, pre_b = 0;
  lookback_sum2_1024<T>(a.tstat, tb, p, pre_a, pre_b);
  QST(5);
  const int upre = (int)(pre_a >> 31), spre = (int)(pre_a & 0x7fffffffull);
  // ---- outputs per unique row, and the entry's (local id, occurrence prefix) for the output pass
  {
    int h_ex = v5[2] + (int)(pre_b >> 40), t_ex = v5[3] + (int)((pre_b >> 20) & 0xfffff), w_ex = v5[4] + (int)(pre_b & 0xfffff);
    int lid = v5[0], pre = v5[1];
    for (int k = 0; k < kEnt; ++k) {
      const int e = tid * kEnt + k;
      const int gs = h_slot[e];
      if (gs == -1) continue;
      const int c = h_cnt[e];
      int64_t kp = (int64_t)h_pl[e];
      kp = kp < a.n ? kp : a.n - 1;
      const uint64_t ukey = a.keys[kp < 0 ? 0 : kp];
      const int uid = upre + lid, pv = spre + pre;
      h_pl[e] = pre;
      h_lid[e] = (unsigned short)lid;
      o.csr_cnt[uid] = c;
      if (o.freq) o.freq[uid] = c;
      o.row_addr[uid] = gs < a.S ? tp0 + ((int64_t)gs - s0) * rowb : 0;
      if (o.table_ids) o.table_ids[uid] = tbl;
      o.slots[uid] = gs < a.S ? (int64_t)gs - s0 : -1;
      o.unique_keys[uid] = ukey;
      ptr[uid] = pv;
      if (hots && c > hot.khot && c <= hot.kwave) {
        const int w = w_ex++;
        if (w < hot.max_hot) { hot.wave_u[w] = uid; hot.wave_lo[w] = pv; hot.wave_cnt[w] = c; }
      } else if (hots && c > hot.khot) {
        const int nch = (c + hot.kchunk - 1) / hot.kchunk;
        const int h = h_ex++, t0 = t_ex;
        t_ex += nch;
        if (h < hot.max_hot && t0 + nch <= hot.max_tasks) {
          hot.hot_done[h] = 0;
          hot.hot_nchunks[h] = nch;
          hot.hot_u[h] = uid;
          hot.hot_lo[h] = pv;
          hot.hot_cnt[h] = c;
          hot.hot_t0[h] = t0;
          for (int cc = 0; cc < nch; ++cc) {
            hot.task_u[t0 + cc] = uid;
            hot.task_h[t0 + cc] = h;
            hot.task_lo[t0 + cc] = pv + cc * hot.kchunk;
            const int hi2 = pv + (cc + 1) * hot.kchunk;
            hot.task_hi[t0 + cc] = hi2 < pv + c ? hi2 : pv + c;
          }
          if (nch > 1)
            for (int e2 = 0; e2 < hot.dim; ++e2) hot.hot_acc[(int64_t)h * hot.dim + e2] = 0.f;
        }
      }
      ++lid; pre += c;
    }
  }
  QST(6);
  __syncthreads();     // h_pl / h_lid of every entry
  // ---- output pass: unique id / rank base / CSR position of every record (lazy reverse indices) and the CSR entries
  {
    int i0 = rec_index(tid < total ? tid : 0);
    uint4 nrc = a.rec[rec_base + i0];
    int4 nro = a.rec_out4[rec_base + i0];
    for (int f0 = 0; f0 < total; f0 += T) {
      const int f = f0 + tid;
      const uint4 rc = nrc;
      const int4 ro = nro;
      const int idx = i0;
      const int fn = f + T;
      i0 = rec_index(fn < total ? fn : 0);
      nrc = a.rec[rec_base + i0];
      nro = a.rec_out4[rec_base + i0];
      if (f >= total) continue;
      const int en = ro.x, bs = ro.y;
      const int uid = upre + (int)h_lid[en];
      const int pos = spre + h_pl[en] + bs;
      const int cn = (int)rc.w & ~kRecLate, br = (int)rc.y;
      a.rec_out4[rec_base + idx] = make_int4(((int)rc.w & kRecLate) ? ~uid : uid, bs, pos, 0);
      if (cn == 1) csr_src[pos] = br;
      else if (cn <= 8) {
        for (int j = 0; j < cn; ++j) csr_src[pos + j] = ~(br + j);
      } else {
        const int q = atomicAdd(&s_nbig, 1);
        if (q < kBigMax) { b_pos[q] = pos; b_ref[q] = br; b_cnt[q] = cn; }
        else for (int j = 0; j < cn; ++j) csr_src[pos + j] = ~(br + j);
      }
    }
    __syncthreads();
    const int nbig = s_nbig < kBigMax ? s_nbig : kBigMax;
    for (int q = tid >> 6; q < nbig; q += T >> 6) {
      const int pos = b_pos[q], br = b_ref[q], cn = b_cnt[q];
      for (int j = lane_id(); j < cn; j += 64) csr_src[pos + j] = ~(br + j);
    }
  }
  QST(7);
  // unique rows in front of the partition's table (the first partition of every table; partitions are table-major)
  if (first_of_table && tid == 0)
    o.table_offsets[tbl] = __hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : upre;
  if (p == (int)a.P - 1 && tid == 0) {
    int U = upre + nu;
    const int O = spre + tot2;
    int nh = (int)(pre_b >> 40) + th, ntk = (int)((pre_b >> 20) & 0xfffff) + tt, nwv = (int)(pre_b & 0xfffff) + tw;
    // a flagged step (a record list overflowed in the probe kernel): no row may be updated from an incomplete CSR -- the step
    // reports zero unique rows, its backward does nothing, and the module raises at its next check
    if (__hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { U = 0; nh = 0; ntk = 0; nwv = 0; }
    if (hots) { *hot.n_hot = nh; *hot.n_tasks = ntk; *hot.n_wave = nwv; }
    o.table_offsets[a.T] = U;
    *o.total = O;
    if (U) ptr[U] = O;
  }
  QST(8);
  QST(9);
}

// (the POOLED gather with the partition blocks in front was measured a loss at C2 -- 0.1227 -> 0.1288 ms, profiles/r05_part_fused.txt --
//  and removed in round 6: pooled batches keep the partition kernel of their own)
// the sequence gather of path (c) with the partition blocks in front
template <int SDT, int DDT>
__global__ void __launch_bounds__(kP3lThreads, 8)
gather_rows_part_kernel(FusedArgs a, EmitOut o, int* __restrict__ ptr, int* __restrict__ csr_src, HotList hot,
                        const int64_t* __restrict__ occ_addr, LateRefs late, int64_t n, int D, void* dst, int64_t dst_stride, int lpr_log2) {
  if ((int)blockIdx.x < a.P) { part3_lean<kPartCap>(a, o, ptr, csr_src, hot, (int)blockIdx.x); return; }
  const int64_t bid = (int64_t)blockIdx.x - a.P;
  const int64_t i0 = (bid * (kP3lThreads >> 6) + (threadIdx.x >> 6)) * 64;
  if (i0 >= n) return;
  const int64_t j = i0 + lane_id();
  uintptr_t rp = j < n ? (uintptr_t)occ_addr[j] : 0;
  if (__ballot(rp == 1)) { if (rp == 1) rp = late_row<true>(late, j); }
  wave_copy_rows<SDT, DDT>(rp, i0, n, D, dst, dst_stride, lpr_log2);
}

}  // namespace mi355
