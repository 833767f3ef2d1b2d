// Round 5: the probe kernel of path (c) (fused_fwd.hip), rebuilt around its LATENCY.  Included by fused_fwd.hip behind
// fused_probe_kernel, whose <kTrain, kPart, kFast, kBags> instantiation it replaces (same inputs, same records / tile lists /
// per-occurrence arrays; MI355_PROBE_C=0 brings the old kernel back).
//
// Restates (reference, corelib/dynamicemb/): segmented_unique_cuda (src/unique_op.cu:484-714), table_lookup_kernel /
// table_insert_kernel (src/table_operation/kernels.cuh:81-585).
//
// What the phase stamps show for the old kernel (profiles/r03_index_phase_stamps.txt, r05_index_phase_stamps_before.txt): 176
// blocks of 1024 threads on 256 CUs (one per CU, 31 % of the CUs idle), a block life of 21 us made of eleven barrier-separated
// phases, of which
//   * the first is the 64-ary search of the tile's bag range by waves 0 / 1 -- three dependent loads -- with the other fourteen
//     waves waiting at the barrier behind it (14-17 %);
//   * the probe phase runs a dependent re-probe (digest vector again, then key words one by one) for every key whose first
//     candidate slot was a digest false positive -- 6 % of the keys, i.e. nearly every wave (17 %);
//   * the LDS dedup serialises two keys per thread through four dependent LDS operations each (10 %).
// And what the first form of THIS kernel showed (profiles/r05_index_phase_stamps_1024x2.txt): 352 blocks of 1024 keys at two per
// CU are no faster -- a block that shares its CU takes 22 us, one that has it alone 12.7 us (the kernel is bound by what a CU can
// issue, not by latency a neighbour could cover), and a search that is pipelined through the phases still gates them, because
// each round is a full memory round trip for the two waves that run it.
// Here:
//   * ONE block per CU, all in one generation: the tile length is a run-time value, ceil(n / #CUs) rounded to 64 (1 408 keys at
//     C2), up to the kernel's capacity (1 024 keys, one per thread; 2 048, two per thread); beyond 2 048 x #CUs keys the tiles are
//     full and the grid runs in generations;
//   * no search: a block GUESSES its bag range from its position (bag ~ position x bags / keys), loads the offsets of a
//     2 048-bag window around the guess with the key loads (coalesced, nothing depends on anything), and checks that the window
//     covers the tile; only a block whose window does not (bag lengths far from uniform) searches;
//   * (round 6) no digest vector at all on the fast path: the key's HOME GROUP -- the 16 key words its probe starts in, one aligned
//     128-byte line at the usual 128-slot buckets -- is fetched by 8 lanes and compared directly: one random line per key
//     instead of two dependent ones (round 5: digest vector, then the key words of the first two digest matches); keys that
//     are not in their home group (and misses) take the full probe (thread_probe), as before;
//   * five barriers instead of eleven; the dedup hash comes out of the table hash (no second fmix64); the representative of a key
//     is whichever occurrence claimed the LDS entry (no atomicMin); rows inserted by the block are initialised by the wave that
//     inserted them, straight from registers.
#pragma once

namespace mi355 {

template <int TILE, int THREADS, int WPS, bool kMT, bool kSeq>
__global__ void __launch_bounds__(THREADS, WPS) probe_c_kernel(FusedArgs a) {
  constexpr int PER = TILE / THREADS;
  constexpr int HASH = 2 * TILE;
  constexpr int NW = THREADS / 64;
  constexpr int NPB = (kPartMax + THREADS - 1) / THREADS;   // partitions a thread reserves for
  constexpr int WIN = 2 * THREADS;                          // bags of the guessed window (two per thread)
  static_assert(TILE % THREADS == 0 && (TILE & (TILE - 1)) == 0 && TILE <= 4096 && THREADS >= 128, "tile shape");
  __shared__ uint64_t s_key[TILE];
  __shared__ int s_tab[HASH];          // dedup: tile position of the key's representative; after the probe: (partition << 12) | position
  __shared__ int s_cnt[HASH];          // dedup: occurrences inside the tile; after the probe: slot code of the key's record
  __shared__ uint16_t s_sm[HASH];      // per dedup entry: start of the key's list in the tile | multi flag << 15
  __shared__ int s_bag[kSeq ? 1 : TILE];
  __shared__ int s_hist[kPartMax];     // records of this tile per partition, then their base in the partition's list
  __shared__ int s_brange[2], s_wmax[NW], s_wsum[NW], s_cover;
  // round 6: the HOME GROUP of a key -- the 16 key words its probe starts in, one aligned 128-byte line at the usual 128-slot
  // buckets -- is fetched by 8 lanes (16 bytes each) and compared directly: one random line per key instead of two dependent ones
  // (16-byte digest vector, then the key words of the digest matches).  s_grp: the line of every key (bucket * groups per
  // bucket + group; -1: none), s_pos: position of the key inside its group (-1: not there -> the full probe)
  __shared__ int s_grp[TILE];
  __shared__ signed char s_pos[TILE];
  __shared__ uint16_t s_t[kMT ? TILE : 1];
  __shared__ int64_t s_seg[kMT ? kFusedMaxT + 1 : 1], s_tbo[kMT ? kFusedMaxT + 1 : 1], s_tptr[kMT ? kFusedMaxT : 1];
  __shared__ int s_rowb[kMT ? kFusedMaxT : 1];
  __shared__ uint64_t s_magic[kMT ? kFusedMaxT : 1];
  __shared__ int s_pt[kMT ? kFusedMaxT : 1], s_pb[kMT ? kFusedMaxT + 1 : 1];
  __shared__ uint64_t s_psc[kMT ? kFusedMaxT : 1];
  PST(0);
  const int T = a.T;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tl = a.tl;                 // keys of a tile (<= TILE, multiple of 64)
  const int64_t tile0 = (int64_t)blockIdx.x * tl;
  const int64_t tile_end = tile0 + tl < a.n ? tile0 + tl : a.n;
  if (!a.timer) a.timer = device_clock();
  // ---- phase 0: everything that depends on nothing goes out first
  uint64_t kreg[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int64_t i = tile0 + q * THREADS + tid;
    kreg[q] = a.keys[i < a.n ? i : a.n - 1];
  }
  // the guessed bag window [wlo, wlo + WIN): bag of a position ~ position * bags / keys; thread t takes the bags wlo + t and
  // wlo + THREADS + t
  int wlo = 0;
  int64_t wo[2][2];
  if constexpr (!kSeq) {
    const int64_t nb = a.num_bags;
    const int64_t g0 = (int64_t)((double)tile0 * (double)nb / (double)a.n), g1 = (int64_t)((double)(tile_end - 1) * (double)nb / (double)a.n);
    int64_t lo = (g0 + g1) / 2 - WIN / 2;
    lo = lo + WIN > nb ? nb - WIN : lo;
    lo = lo < 0 ? 0 : lo;
    wlo = (int)lo;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t b = lo + h * THREADS + tid;
      const int64_t bc = b < nb ? b : nb - 1;
      wo[h][0] = a.offsets[bc];
      wo[h][1] = a.offsets[bc + 1];
    }
  }
  // (one table: its scalars come in through the VECTOR memory path -- an index the compiler cannot see is zero --, because as
  //  uniform scalar loads they share the wave's LDS counter and the first barrier, which waits for the LDS initialisation, waited
  //  for their round trip as well: 2.6 us in front of barrier 0, profiles/r05_index_phase_stamps_1cu_a.txt)
  int64_t m_tbo0 = 0, m_tbo1 = 0, m_tptr0 = 0;
  int m_rowb0 = 0;
  if constexpr (!kMT) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    m_tbo0 = a.tbo[z]; m_tbo1 = a.tbo[z + 1]; m_tptr0 = a.table_ptrs[z]; m_rowb0 = (int)a.table_value_dims[z] * a.elem_bytes;
  }
  auto tbo_of = [&](int t) -> int64_t { if constexpr (!kMT) return t == 0 ? m_tbo0 : m_tbo1; else return s_tbo[t]; };
  auto tptr_of = [&](int t) -> int64_t { if constexpr (!kMT) return m_tptr0; else return s_tptr[t]; };
  auto rowb_of = [&](int t) -> int { if constexpr (!kMT) return m_rowb0; else return s_rowb[t]; };
  if constexpr (kMT) {
    for (int t = tid; t <= T; t += THREADS) {
      s_seg[t] = a.offsets[a.feature_offsets[t] * a.batch];
      s_tbo[t] = a.tbo[t];
      if (t < T) {
        s_tptr[t] = a.table_ptrs[t];
        s_rowb[t] = (int)a.table_value_dims[t] * a.elem_bytes;
        const uint64_t nb = (uint64_t)(a.tbo[t + 1] - a.tbo[t]);
        s_magic[t] = nb ? ~0ull / nb : 0ull;
      }
    }
  }
  if (blockIdx.x == 0) {
    if (a.hot_counters && tid < 3) a.hot_counters[2 * tid] = 0;   // n_hot, n_tasks, n_wave
    if (a.rerun_mark && tid == 0) *a.rerun_mark = 0;
  }
  if (a.tstat) {   // the 2 P look-back words of the partition kernel (+ its P ready flags, two per word, behind them)
    const int nw = 2 * a.P + (a.part_ready ? (a.P + 1) / 2 : 0);
    const int per = (nw + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int k = tid; k < per; k += THREADS) {
      const int wd = (int)blockIdx.x * per + k;
      if (wd < nw) a.tstat[wd] = 0ull;
    }
  }
  for (int s = tid; s < HASH; s += THREADS) { s_tab[s] = -1; s_cnt[s] = 0; }
  for (int p = tid; p < a.P; p += THREADS) s_hist[p] = 0;
  if constexpr (!kSeq) for (int k = tid; k < TILE; k += THREADS) s_bag[k] = -1;
  if (tid == 0) s_cover = 1;
  __syncthreads();   // 0: the cleared LDS (and the tables' metadata); nothing has been waited for yet
  PST(1);
  if constexpr (kMT) {   // partitions per table: one each, the rest in proportion to the tables' keys in this batch
    if (tid < T) {
      const uint64_t nt_ = (uint64_t)(s_seg[tid + 1] - s_seg[tid]);
      s_pt[tid] = 1 + (a.n > 0 ? (int)((uint64_t)(a.P - T) * nt_ / (uint64_t)a.n) : 0);
    }
  }
  // ---- phase 1: hash, bucket (division-free: the bucket capacity is a power of two here), first digest vector of EVERY key;
  //      bag marks from the window
  const int cshift = __builtin_ctzll((unsigned long long)a.t.C);
  const int Cm = (int)a.t.C - 1;
  int bq[PER], tq[PER], pkq[PER];  // bucket (-1: key without a home), table, partition of the key's bucket
  int64_t hq[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int li = q * THREADS + tid;
    const int64_t i = tile0 + li;
    int tt = 0;
    if constexpr (kMT) {
      int lo = 0, hi = T + 1;      // first t with seg[t] > i
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_seg[mid] <= i) lo = mid + 1; else hi = mid; }
      tt = lo - 1 < 0 ? 0 : (lo - 1 >= T ? T - 1 : lo - 1);
      s_t[li] = (uint16_t)tt;
    }
    tq[q] = tt;
    const uint64_t key = kreg[q];
    s_key[li] = key;
    const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
    const int64_t bb = tbo_of(tt);
    const uint64_t nb = (uint64_t)(tbo_of(tt + 1) - bb);
    const uint64_t x = (uint64_t)hash >> cshift;
    uint64_t magic;
    if constexpr (kMT) magic = s_magic[tt]; else magic = a.magic0;
    uint64_t r = x - __umul64hi(x, magic) * nb;
    if (r >= nb) r -= nb;
    if (r >= nb) r -= nb;
    const bool ok = li < tl && i < a.n && is_valid(key) && nb > 0;
    hq[q] = hash;
    bq[q] = ok ? (int)(bb + (int64_t)r) : -1;
    if constexpr (!kMT) pkq[q] = bq[q] < 0 ? a.P - 1 : (int)((uint32_t)bq[q] / (uint32_t)(a.spp >> cshift));
    const int start = ((int)hash & Cm) & ~15;
    s_grp[li] = ok ? (int)(((int64_t)bq[q] << (cshift - 4)) + (start >> 4)) : -1;
  }
  if constexpr (!kSeq) {
    // a bag marks the tile position of its first key (a bag that began in an earlier tile: position 0); the window covers the
    // tile iff its first bag starts at or before the tile and its end lies at or behind the tile's end
    const int64_t nb = a.num_bags;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t b = (int64_t)wlo + h * THREADS + tid;
      const int64_t o0 = wo[h][0], o1 = wo[h][1];
      if (b < nb && o1 > o0 && o1 > tile0 && o0 < tile_end) s_bag[o0 > tile0 ? o0 - tile0 : 0] = (int)b;
      if (h == 0 && tid == 0 && o0 > tile0) s_cover = 0;                                   // (offsets[0] = 0: bag 0 always covers)
      if (b == (int64_t)wlo + WIN - 1 && b < nb - 1 && o1 < tile_end) s_cover = 0;          // the window ends inside the tile
    }
  }
  PST(2);
  __syncthreads();   // A: s_key / s_t, the bag marks
  PST(3);
  if constexpr (!kSeq) {
    if (!s_cover) {    // (block uniform, rare) the guess missed: waves 0 / 1 search the tile's bag range, everybody marks from it
      if (tid < 128) {
        const int64_t skey = tid < 64 ? tile0 : tile_end - 1;
        int lo = 0, hi = (int)a.num_bags;
        while (hi > lo) {
          const int step = (hi - lo + 63) >> 6;
          const int64_t pi = (int64_t)lo + (int64_t)(lane + 1) * step - 1;
          const bool gt = pi >= hi ? true : a.offsets[pi] > skey;
          const uint64_t gm = __ballot(gt);
          if (!gm) { lo = hi; break; }
          const int first = __ffsll((unsigned long long)gm) - 1;
          const int64_t nhi = (int64_t)lo + (int64_t)(first + 1) * step - 1;
          lo = lo + first * step;
          hi = nhi < hi ? (int)nhi : hi;
        }
        if (lane == 0) s_brange[wv] = lo - 1;
      }
      for (int k = tid; k < TILE; k += THREADS) s_bag[k] = -1;
      __syncthreads();
      const int blo = s_brange[0] < 0 ? 0 : s_brange[0], bhi = s_brange[1];
      for (int b = blo + tid; b <= bhi; b += THREADS) {
        const int64_t o0 = a.offsets[b], o1 = a.offsets[b + 1];
        if (o1 > o0 && o1 > tile0 && o0 < tile_end) s_bag[o0 > tile0 ? o0 - tile0 : 0] = b;
      }
      __syncthreads();
    }
  }
  if constexpr (kMT) {
    if (tid < 64) {      // (T <= 128: two tables per lane of wave 0)
      const int l = tid;
      const int v0 = l < T ? s_pt[l] : 0, v1 = l + 64 < T ? s_pt[l + 64] : 0;
      const int i0 = wave_incl_scan(v0);
      const int i1 = wave_incl_scan(v1) + __shfl(i0, 63, 64);
      if (l < T) {
        s_pb[l] = i0 - v0;
        const uint64_t nb = (uint64_t)(s_tbo[l + 1] - s_tbo[l]);
        s_psc[l] = nb ? ((uint64_t)v0 << 32) / nb : 0ull;
      }
      if (l + 64 < T) {
        s_pb[l + 64] = i1 - v1;
        const uint64_t nb = (uint64_t)(s_tbo[l + 65] - s_tbo[l + 64]);
        s_psc[l + 64] = nb ? ((uint64_t)v1 << 32) / nb : 0ull;
      }
      if (l == 63) s_pb[T] = i1;
    }
  }
  // the home groups of the wave's 64 PER keys: 8 keys per load instruction, lane = (key, 16-byte chunk); issued here, behind
  // barrier A and in front of the dedup, for EVERY occurrence (for the representatives only, behind the dedup: slower, the batch
  // no longer runs under the dedup -- profiles/r06_probe_keyline.txt)
  constexpr int NLN = 8 * PER;
  uint4 lnq[NLN];
  auto issue_lines = [&]() {
    const int gsh = cshift - 4, gmask = (1 << gsh) - 1;
#pragma unroll
    for (int j = 0; j < NLN; ++j) {
      const int k = (j >> 3) * THREADS + wv * 64 + (j & 7) * 8 + (lane >> 3);
      const int grp = s_grp[k];
      const int gb = grp < 0 ? 0 : grp >> gsh, gg = grp < 0 ? 0 : grp & gmask;
      lnq[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(a.t.keys(gb)) + gg * 128 + (lane & 7) * 16);
    }
  };
  auto compare_lines = [&]() {   // a lane holds two key words of key k; the octet's match (keys are unique: at most one) -> s_pos[k]
#pragma unroll
    for (int j = 0; j < NLN; ++j) {
      const int k = (j >> 3) * THREADS + wv * 64 + (j & 7) * 8 + (lane >> 3);
      const uint64_t key = s_key[k];
      const uint64_t w0 = ((uint64_t)lnq[j].y << 32) | lnq[j].x, w1 = ((uint64_t)lnq[j].w << 32) | lnq[j].z;
      const bool live = s_grp[k] >= 0;
      const uint64_t b0 = __ballot(live && w0 == key), b1 = __ballot(live && w1 == key);
      const int sh = lane & ~7;
      const uint32_t x0 = (uint32_t)(b0 >> sh) & 0xffu, x1 = (uint32_t)(b1 >> sh) & 0xffu;
      int pos = -1;
      if (x0) pos = 2 * (__ffs(x0) - 1);
      else if (x1) pos = 2 * (__ffs(x1) - 1) + 1;
      if ((lane & 7) == 0) s_pos[k] = (signed char)pos;
    }
  };
  issue_lines();
  // ---- phase 2: LDS dedup.  The entry's hash comes out of the table hash (bits above the digest); whoever claims the entry
  //      represents the key.  Running maximum of the bag marks, first half (thread t owns the positions t PER ..).
  //      The claimer also counts the pair in the tile's histogram over the partitions (the partition follows from the BUCKET,
  //      known since phase 1): the histogram is complete at barrier B, so the global reservation goes out right behind it and
  //      its round trip -- 65 K returning device atomics, ~6 us when they all queue at once -- runs under phases 3 and 4.
  int hh[PER], rk[PER], lpq[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int li = q * THREADS + tid;
    hh[q] = -1;
    rk[q] = 0;
    lpq[q] = -1;
    if (li < tl && tile0 + li < a.n) {
      const uint64_t key = kreg[q];
      int h = (int)((uint32_t)((uint64_t)hq[q] >> 40) + (kMT ? (uint32_t)tq[q] * 0x9E3779B1u : 0u)) & (HASH - 1);
      bool claimed = false;
      while (true) {
        const int cur = atomicCAS(&s_tab[h], -1, li);
        if (cur == -1) { claimed = true; break; }
        bool same = s_key[cur] == key;
        if constexpr (kMT) same = same && s_t[cur] == (uint16_t)tq[q];
        if (same) break;
        h = (h + 1) & (HASH - 1);
      }
      hh[q] = h;
      rk[q] = atomicAdd(&s_cnt[h], 1);
      if constexpr (!kMT) { if (claimed) lpq[q] = pkq[q] * 4096 + atomicAdd(&s_hist[pkq[q]], 1); }
    }
  }
  int bagv[PER], bag_incl = -1;
  if constexpr (!kSeq) {
    int m = -1;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int v = s_bag[tid * PER + k]; m = v > m ? v : m; bagv[k] = m; }
    bag_incl = wave_incl_max(m);
    if (lane == 63) s_wmax[wv] = bag_incl;
  }
  PST(4);
  __syncthreads();   // B: the dedup, the waves' bag maxima
  PST(5);
  // ---- phase 3: representatives; list starts of the multi-occurrence keys (first half of a block scan); the tile's histogram
  //      over the partitions; key words of the first two digest matches; bags of all occurrences (second half of the scan)
  bool isrep[PER];
  int mcnt[PER], m_mine = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    isrep[q] = hh[q] >= 0 && s_tab[hh[q]] == q * THREADS + tid;
    const int c = isrep[q] ? s_cnt[hh[q]] : 0;
    mcnt[q] = c > 1 ? c : 0;
    m_mine += mcnt[q];
  }
  const int m_incl = wave_incl_scan(m_mine);
  if (lane == 63) s_wsum[wv] = m_incl;
  const int sub = kMT ? 0 : (int)blockIdx.x % kPartSub;
  int my_base[NPB];
  auto reserve = [&]() {   // one returning atomic per (tile, partition) reserves the tile's records in the partition's list
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int p = tid + j * THREADS;
      my_base[j] = 0;
      if (p < a.P) {
        const int c = s_hist[p];
        // (FusedArgs::dbg bit 3 skips the reservation -- results wrong, a timing probe: 24.3 -> 19.6 us at C2, the 65 K returning
        //  atomics of a launch drain for ~5 us.  A layout without them -- every tile owning q slots of every partition's list --
        //  was built and dropped: profiles/r05_cells_ab.txt)
        if (c) my_base[j] = (a.dbg & 8) ? 0 : atomicAdd(&a.pcount[p * kPartSub + sub], c);
        if constexpr (kMT) {
          if (blockIdx.x == 0) {   // the partition kernel learns its table from here
            int t = 0;
            while (t + 1 < T && s_pb[t + 1] <= p) ++t;
            a.ptab[p] = t;
          }
        }
      }
    }
  };
  if constexpr (!kMT) reserve();
  if constexpr (kMT) {   // several tables: the partition of a pair needs the tables' partition ranges (wave 0, behind barrier A)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (isrep[q]) {   // the table's partitions split its bucket range evenly; keys without a home: its last partition
        const int t = tq[q];
        const int p0 = s_pb[t], p1 = s_pb[t + 1];
        const int x = bq[q] < 0 ? p1 - p0 - 1 : (int)(((uint64_t)((int64_t)bq[q] - s_tbo[t]) * s_psc[t]) >> 32);
        const int pk = p0 + (x < p1 - p0 ? x : p1 - p0 - 1);
        lpq[q] = pk * 4096 + atomicAdd(&s_hist[pk], 1);
      }
    }
  }
  compare_lines();
  if constexpr (!kSeq) {
    int bbase = -1;
    for (int k = 0; k < wv; ++k) bbase = s_wmax[k] > bbase ? s_wmax[k] : bbase;
    int prev = __shfl_up(bag_incl, 1, 64);
    if (lane == 0) prev = -1;
    prev = prev > bbase ? prev : bbase;
#pragma unroll
    for (int k = 0; k < PER; ++k) s_bag[tid * PER + k] = bagv[k] > prev ? bagv[k] : prev;
  }
  PST(6);
  __syncthreads();   // C: the histogram, the wave sums, the bags; everybody has read the dedup's s_tab / s_cnt
  PST(7);
  // ---- phase 4: list starts; the probe itself (several tables: the reservation goes out here)
  if constexpr (kMT) reserve();
  {
    int st = m_incl - m_mine;
    for (int k = 0; k < wv; ++k) st += s_wsum[k];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (isrep[q]) s_sm[hh[q]] = (uint16_t)(mcnt[q] ? (st | 0x8000) : 0);
      st += mcnt[q];
    }
  }
  int cnt_tile[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    cnt_tile[q] = 0;
    bool inserted = false;
    int gslot = (int)a.S;              // default: no slot
    if (isrep[q]) {
      const uint64_t key = kreg[q];
      const int cnt = s_cnt[hh[q]];
      cnt_tile[q] = cnt;
      bool defer = false;
      if (bq[q] >= 0) {
        const int64_t b = bq[q];
        int slot;
        const int gpos = s_pos[q * THREADS + tid];
        if (gpos >= 0) slot = (((int)hq[q] & Cm) & ~15) + gpos;
        else slot = thread_probe<true>(a, b, key, hq[q], cnt, inserted);
        if (slot >= 0) {
          gslot = (int)(b * a.t.C + slot);
          // (Assign / timer scores are the same value from every tile: plain stores)
          if (!inserted) score_found(a, a.t.scores(b) + (int64_t)slot * a.t.ns, cnt);
        } else if (slot == -2) {
          defer = true;
        }
      }
      s_tab[hh[q]] = lpq[q];                                // (partition, position among the tile's records of the partition)
      s_cnt[hh[q]] = defer ? -bq[q] - 2 : gslot;            // slot code of the record
    }
    // first-touch initialisation of the rows this wave inserted: the whole wave per row, coalesced stores (steady state: none)
    uint64_t todo = __ballot(inserted);
    while (todo) {
      const int src = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int g = __shfl(gslot, src, 64), t = __shfl(tq[q], src, 64);
      const uint32_t klo = __shfl((int)(uint32_t)kreg[q], src, 64), khi = __shfl((int)(uint32_t)(kreg[q] >> 32), src, 64);
      void* rp = reinterpret_cast<void*>((uintptr_t)(tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t)));
      wave_init_row(a, rp, ((uint64_t)khi << 32) | klo, (int)a.table_emb_dims[t], (int)a.table_value_dims[t]);
    }
  }
#pragma unroll
  for (int j = 0; j < NPB; ++j) {
    const int p = tid + j * THREADS;
    if (p < a.P) s_hist[p] = my_base[j];
  }
  PST(8);
  __syncthreads();   // D: the records' slot codes, the reserved bases
  PST(9);
  PST(10);
  // ---- phase 5: records (by the representatives) and the per-occurrence arrays
  constexpr int kListCap = kMT ? kPartCap : kSubCap;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (hh[q] < 0) continue;
    const int li = q * THREADS + tid;
    const int64_t i = tile0 + li;
    const int v = s_tab[hh[q]];
    const int pk = v >> 12, idx = s_hist[pk] + (v & 4095);
    const int ref = idx < kListCap ? pk * kPartCap + sub * kSubCap + idx : -1;
    const int g = s_cnt[hh[q]];
    const int sm = s_sm[hh[q]];
    int bag;
    if constexpr (kSeq) bag = (int)i; else bag = s_bag[li];
    if (isrep[q]) {
      if (ref >= 0)
        a.rec[ref] = make_uint4((uint32_t)i, (uint32_t)((sm & 0x8000) ? (int)tile0 + (sm & 0x7fff) : bag), (uint32_t)g, (uint32_t)cnt_tile[q]);
      else
        a.hdr[a.ovf_word] = a.ovf_val;   // a partition received more records than it can hold: the step is flagged (see the module)
    }
    a.occ_slot[i] = ref;
    a.occ_trank[i] = rk[q];
    if (sm & 0x8000) a.tile_bags[tile0 + (sm & 0x7fff) + rk[q]] = bag;
    const int t = tq[q];
    // address word 1: the row comes out of the partition kernel's eviction (gather_dev.h: LateRefs)
    a.occ_addr[i] = (g >= 0 && g < a.S) ? tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t) : (g <= -2 ? 1 : 0);
  }
  if (blockIdx.x == 0)
    for (int t = tid; t <= T; t += THREADS) {
      if constexpr (kMT) a.seg_out[t] = s_seg[t]; else a.seg_out[t] = t == 0 ? 0 : a.n;
    }
  PST(11);
}

}  // namespace mi355
