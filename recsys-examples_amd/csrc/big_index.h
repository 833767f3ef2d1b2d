// Round 5: the index stage of batches beyond 1 M keys (the 16x batch of SURVEY 8(d): 5.8 M keys) on the partitioned path.
// Included by fused_fwd.hip.  Restates (reference, corelib/dynamicemb/): segmented_unique_cuda (src/unique_op.cu:484-714) -- whose
// one code path serves any batch size.
//
// Why the lists of path (c) do not scale as they are: a tile reserves its records in the partitions' lists with one returning device
// atomic per (tile, partition) it touches -- tiles x min(P, pairs per tile) of them.  At 5.8 M keys that is 2 816 tiles x ~1 100
// partitions = 3.1 M atomics, as many as the per-slot counters of path (b) cost (they resolve at the memory side, ~10 G/s: 350 us).
// Here (DESIGN 6b of round 3, built):
//   1. probe_c_kernel<kStage> writes a tile's records TILE-MAJOR -- dense behind the tile's first position -- and the tile's record
//      count: no histogram, no reservation, no atomic;
//   2. split_records_kernel: a block takes a run of consecutive tiles (~16 K records), counts them per partition in LDS, reserves
//      with ONE atomic per (block, partition) -- blocks x P, a tenth of the per-tile count -- and moves the records into the
//      partitions' lists (L2-hot second pass); every staged record leaves a forwarding entry, which is what the per-occurrence
//      references of the probe kernel (late rows of the gather, lazily materialised reverse indices) go through;
//   3. fused_part3s_kernel: the partition kernel of path (c) in a STREAMING form.  With thousands of tiles a key of the Zipf head
//      leaves one record per tile -- 2 816 records of ONE slot at the 16x batch, all in one partition -- so a partition's list is
//      not bounded by its distinct slots any more (first build, 4 096-record lists: the 16x step overflowed on every batch).  The
//      lists hold kPartCapBig = 32 768 records (P = keys / 11 264 partitions: few, large partitions amortise the kernel's fixed round
//      trips -- at P = keys / 2 816 the 2 048 blocks of the 16x batch took 300 us, 37 us each in eight generations); the kernel walks them twice, 1 024 records at a time (merge pass: LDS hash of the
//      partition's distinct slots, rank base of every record; output pass: CSR entries), with the per-record state kept in the
//      record's output entry between the passes -- nothing per record lives in registers.
#pragma once

namespace mi355 {

constexpr int kSplitThreads = 1024;

__device__ __forceinline__ int part_of_code(const FusedArgs& a, int z, int cshift) {
  // slot code of a record: a global slot, S (no slot: the last partition), or -(bucket + 2) (bucket full: deferred to the partition kernel)
  if (z >= 0) return z < a.S ? (int)((uint32_t)(z >> cshift) / (uint32_t)(a.spp >> cshift)) : a.P - 1;
  return (int)((uint32_t)(-z - 2) / (uint32_t)(a.spp >> cshift));
}

__global__ void __launch_bounds__(kSplitThreads) split_records_kernel(FusedArgs a, int tiles_per_block, int ntiles) {
  __shared__ int s_h[kPartMaxBig];      // records of this block per partition, then the running position inside the block's share
  __shared__ int s_base[kPartMaxBig];   // start of the block's share in the partition's sub-list
  const int tid = (int)threadIdx.x;
  const int t0 = (int)blockIdx.x * tiles_per_block;
  const int t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  const int cshift = __builtin_ctzll((unsigned long long)a.t.C);
  const int sub = (int)blockIdx.x % kPartSub;
  const int subcap = a.cap / kPartSub;
  for (int p = tid; p < a.P; p += kSplitThreads) s_h[p] = 0;
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int c = a.tile_cnt[t];
    const int64_t base = (int64_t)t * a.tl;
    for (int r = tid; r < c; r += kSplitThreads) atomicAdd(&s_h[part_of_code(a, (int)a.stage_rec[base + r].z, cshift)], 1);
  }
  __syncthreads();
  for (int p = tid; p < a.P; p += kSplitThreads) {
    const int c = s_h[p];
    s_base[p] = c ? atomicAdd(&a.pcount[p * kPartSub + sub], c) : 0;
    s_h[p] = 0;
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int c = a.tile_cnt[t];
    const int64_t base = (int64_t)t * a.tl;
    for (int r = tid; r < c; r += kSplitThreads) {
      const uint4 rec = a.stage_rec[base + r];
      const int pk = part_of_code(a, (int)rec.z, cshift);
      const int idx = s_base[pk] + atomicAdd(&s_h[pk], 1);
      int ref = -1;
      if (idx < subcap) {
        ref = pk * a.cap + sub * subcap + idx;
        a.rec[ref] = rec;
      } else {
        a.hdr[a.ovf_word] = a.ovf_val;   // a partition received more records than it can hold: the step is flagged (see the module)
      }
      a.fwd[base + r] = ref;
    }
  }
}


// The streaming partition kernel (see the header).  Same outputs as fused_part3_kernel; one block per partition.
constexpr int kP3sHash = 8192;          // hash entries for the distinct slots of a partition (avg ~2 700 at P = keys / 11 264); past ~7 000 claimed
                                        // entries (1 024 threads may claim at once) the step is flagged
constexpr int kP3sDef = 1024;           // deferred records (bucket full) evicted for per step and partition

__global__ void __launch_bounds__(kP3Threads)
fused_part3s_kernel(FusedArgs a, EmitOut o, int* __restrict__ ptr, int* __restrict__ csr_src, HotList hot) {
  constexpr int HASH = kP3sHash, kEnt = HASH / kP3Threads;
  __shared__ int h_slot[HASH], h_cnt[HASH], h_pl[HASH];
  __shared__ unsigned short h_lid[HASH];
  __shared__ int d_rec[kP3sDef], d_ent[kP3sDef], d_base[kP3sDef];
  __shared__ int s_lock[256];
  __shared__ unsigned s_late[HASH / 32];
  __shared__ int s_nd, s_nbig, s_nclaim;
  constexpr int kBigMax = 512;
  __shared__ int b_pos[kBigMax], b_ref[kBigMax], b_cnt[kBigMax];
  const int p = blockIdx.x, tid = (int)threadIdx.x;
  const int cap = a.cap, subcap = a.cap / kPartSub;
  QST(0);
  if (a.notice && p == 0 && tid == 0) publish_notice(a);
  const int64_t rec_base = (int64_t)p * cap;
  const int mv = a.pcount[p * kPartSub + (tid & (kPartSub - 1))];
  for (int i = tid; i < HASH; i += kP3Threads) { h_slot[i] = -1; h_cnt[i] = 0; }
  if (tid < 256) s_lock[tid] = 0;
  if (tid < HASH / 32) s_late[tid] = 0;
  if (tid == 0) { s_nd = 0; s_nbig = 0; s_nclaim = 0; }
  __syncthreads();
  QST(1);
  int msub[kPartSub];
#pragma unroll
  for (int r = 0; r < kPartSub; ++r) { const int m = __builtin_amdgcn_readlane(mv, r); msub[r] = m < subcap ? m : subcap; }
  if (tid < kPartSub) a.pcount[p * kPartSub + tid] = 0;       // clean for the next step
  const int tbl = 0;
  const int64_t tp0 = a.table_ptrs[tbl], rowb = a.table_value_dims[tbl] * a.elem_bytes, s0 = a.tbo[tbl] * a.t.C;
  // ---- merge pass: the records of a slot meet in its hash entry; every record learns its entry and the rank base of its tile.
  //      The four sub-lists are walked as ONE index space (flat index -> sub-list by three compares), 1 024 records a round, the
  //      next round's records in flight while this round's are merged.
  const int c1 = msub[0], c2 = c1 + msub[1], c3 = c2 + msub[2], total = c3 + msub[3];
  auto rec_index = [&](int f) -> int {            // flat index -> index inside the partition's list
    return f < c1 ? f : (f < c2 ? subcap + f - c1 : (f < c3 ? 2 * subcap + f - c2 : 3 * subcap + f - c3));
  };
  //      A record that claims an entry owns the unique row's KEY: it leaves the key's position in the entry (h_pl, which the scan
  //      below overwrites with the occurrence prefix once the owners of the entries have read it).  Fetching the key per RECORD --
  //      4.2 M random 8-byte loads at the 16x batch, three times the unique rows -- cost 40 K cycles per block wherever it sat
  //      (profiles/r05_stamps_16x_a.txt, _b.txt); the entries' owners fetch 1.4 M, under the scan and the look-back.
  {
    uint4 nxt = a.rec[rec_base + rec_index(tid < total ? tid : 0)];
    for (int f0 = 0; f0 < total; f0 += kP3Threads) {
      const int f = f0 + tid;
      const uint4 rc = nxt;
      const int fn = f + kP3Threads;
      nxt = a.rec[rec_base + rec_index(fn < total ? fn : 0)];
      if (f >= total) continue;
      const int idx = rec_index(f);
      const int sl = (int)rc.z, cn = (int)rc.w;
      int en = 0, bs = 0;
      if (sl >= 0) {
        // (a partition the hash cannot hold -- more distinct slots than entries: the step is flagged, the record joins the row-less entry)
        int want = sl;
        if (__hip_atomic_load(&s_nclaim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > HASH - 1088 && p2_find<HASH>(h_slot, sl) < 0) {
          want = (int)a.S;
          a.hdr[a.ovf_word] = a.ovf_val;
        }
        bool cl;
        en = p2_insert<HASH>(h_slot, want, &cl);
        if (cl) { atomicAdd(&s_nclaim, 1); h_pl[en] = (int)rc.x; }
        bs = atomicAdd(&h_cnt[en], cn);
      } else {
        const int dj = atomicAdd(&s_nd, 1);
        if (dj < kP3sDef) {
          d_rec[dj] = idx;                         // (entry and rank base come from the eviction below)
        } else {                                   // beyond what one step evicts for: no slot this step (like an insert that returns Busy)
          bool cl;
          en = p2_insert<HASH>(h_slot, (int)a.S, &cl);
          if (cl) { atomicAdd(&s_nclaim, 1); h_pl[en] = (int)rc.x; }
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
  const int nd = s_nd < kP3sDef ? s_nd : kP3sDef;
  if (nd > 0) {      // (block uniform)
    part_evict<HASH>(a, nd, rec_base, d_rec, d_ent, d_base, s_lock, s_late, h_slot, h_cnt, tbl, tp0, rowb, s0);
    __syncthreads();
    for (int e = tid; e < nd; e += kP3Threads) {
      // the first record (rank base 0) of an entry created by the eviction owns the unique row's key
      const int ent = d_ent[e], bs = d_base[e];
      if (((s_late[ent >> 5] >> (ent & 31)) & 1u) && bs == 0) h_pl[ent] = (int)a.rec[rec_base + d_rec[e]].x;
      a.rec_out4[rec_base + d_rec[e]] = make_int4(ent, bs, 0, 0);
    }
    __syncthreads();
  }
  // ---- one scan over the hash ENTRIES: local unique id, occurrence prefix, hot-list positions (entry order = unique order)
  const bool hots = hot.n_tasks != nullptr;
  int es[kEnt], ec[kEnt];
  uint64_t uk[kEnt];                      // key of every unique row of mine: in flight under the scan and the look-back
  int v5[5] = {0, 0, 0, 0, 0}, tot5[5];
#pragma unroll
  for (int k = 0; k < kEnt; ++k) {
    const int e = tid * kEnt + k;
    es[k] = h_slot[e];
    ec[k] = es[k] != -1 ? h_cnt[e] : 0;
    int64_t kp = es[k] != -1 ? (int64_t)h_pl[e] : 0;
    kp = kp < a.n ? kp : a.n - 1;
    uk[k] = a.keys[kp < 0 ? 0 : kp];
    if (es[k] == -1) continue;
    ++v5[0];
    v5[1] += ec[k];
    if (hots && ec[k] > hot.khot && ec[k] <= hot.kwave) ++v5[4];
    else if (hots && ec[k] > hot.khot) { ++v5[2]; v5[3] += (ec[k] + hot.kchunk - 1) / hot.kchunk; }
  }
  unsigned long long* tb = a.tstat + a.P;
  block_scan5<kP3Threads>(v5, tot5, [&](const int (&t5)[5]) {
    stat_store(a.tstat + p, kStatAgg | ((unsigned long long)t5[0] << 31) | (unsigned)t5[1]);
    stat_store(tb + p, kStatAgg | ((unsigned long long)t5[2] << 40) | ((unsigned long long)t5[3] << 20) | (unsigned long long)t5[4]);
  });
  const int nu = tot5[0], tot2 = tot5[1], th = tot5[2], tt = tot5[3], tw = tot5[4];
  {
    int lid = v5[0], pre = v5[1];
#pragma unroll
    for (int k = 0; k < kEnt; ++k) {
      if (es[k] == -1) continue;
      h_pl[tid * kEnt + k] = pre;
      h_lid[tid * kEnt + k] = (unsigned short)lid;
      ++lid; pre += ec[k];
    }
  }
  QST(4);
  unsigned long long pre_a = 0, pre_b = 0;
  lookback_sum2_1024(a.tstat, tb, p, pre_a, pre_b);     // (its barriers also publish h_pl / h_lid to the block)
  QST(5);
  const int upre = (int)(pre_a >> 31), spre = (int)(pre_a & 0x7fffffffull);
  // ---- outputs per unique row: by the thread that owns the entry (consecutive entries -> consecutive unique ids)
  {
    int h_ex = v5[2] + (int)(pre_b >> 40), t_ex = v5[3] + (int)((pre_b >> 20) & 0xfffff), w_ex = v5[4] + (int)(pre_b & 0xfffff);
    int uid = upre + v5[0], pv = spre + v5[1];
#pragma unroll
    for (int k = 0; k < kEnt; ++k) {
      const int gs = es[k];
      if (gs == -1) continue;
      const int c = ec[k];
      o.csr_cnt[uid] = c;
      if (o.freq) o.freq[uid] = c;
      o.row_addr[uid] = gs < a.S ? tp0 + ((int64_t)gs - s0) * rowb : 0;
      if (o.table_ids) o.table_ids[uid] = tbl;
      o.slots[uid] = gs < a.S ? (int64_t)gs - s0 : -1;
      o.unique_keys[uid] = uk[k];
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
      ++uid; pv += c;
    }
  }
  QST(6);
  // ---- output pass: unique id / rank base / CSR position of every record (lazy reverse indices), the unique row's key, the CSR
  //      entries; the next round's record and state in flight.  Long lists (a hot key's occurrences in one tile) are collected and
  //      expanded by whole waves behind the loop (beyond 512 of them: by their own thread).
  {
    int i0 = rec_index(tid < total ? tid : 0);
    uint4 nrc = a.rec[rec_base + i0];
    int4 nro = a.rec_out4[rec_base + i0];
    for (int f0 = 0; f0 < total; f0 += kP3Threads) {
      const int f = f0 + tid;
      const uint4 rc = nrc;
      const int4 ro = nro;
      const int idx = i0;
      const int fn = f + kP3Threads;
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
    for (int q = tid >> 6; q < nbig; q += kP3Threads >> 6) {
      const int pos = b_pos[q], br = b_ref[q], cn = b_cnt[q];
      for (int j = lane_id(); j < cn; j += 64) csr_src[pos + j] = ~(br + j);
    }
  }
  QST(7);
  if (p == 0 && tid == 0) o.table_offsets[0] = __hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : upre;
  if (p == (int)a.P - 1 && tid == 0) {
    int U = upre + nu;
    const int O = spre + tot2;
    int nh = (int)(pre_b >> 40) + th, ntk = (int)((pre_b >> 20) & 0xfffff) + tt, nwv = (int)(pre_b & 0xfffff) + tw;
    // a flagged step (a list or the hash overflowed): no row may be updated from an incomplete CSR -- zero unique rows (see the module)
    if (__hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { U = 0; nh = 0; ntk = 0; nwv = 0; }
    if (hots) { *hot.n_hot = nh; *hot.n_tasks = ntk; *hot.n_wave = nwv; }
    o.table_offsets[a.T] = U;
    *o.total = O;
    if (U) ptr[U] = O;
  }
  QST(8);
  QST(9);
}

}  // namespace mi355
