// HSTU jagged attention forward for gfx950 (MFMA bf16).
//
//   O[i,h,:] = (1/scaling_seqlen) * sum_j M(i,j) * SiLU(alpha * <q_i,h , k_j,h>) * v_j,h      per jagged sequence
//
// Replaces (reference): hstu_varlen_fwd -> hstu_compute_attn_1rowblock
// (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:335-523, src/hstu_fwd.h:47-700; mask apply_mask :473-556;
// SiLU utils.h:86-94), PyTorch statement examples/hstu/ops/pt_ops/pt_hstu_attention.py:45-196.
// NOT softmax attention: no running max / sum, the key loop is a plain accumulation.
//
// MI355X design (nothing of the CuTe m16n8k16 tiling survives)
//  * one workgroup = 4 wave64 = 128 query rows of one (sequence, head); each wave owns 32 query rows and
//    keeps Q (B operand) and the whole O accumulator (d x 32, fp32) in registers/AGPRs.
//  * v_mfma_f32_32x32x16_bf16 everywhere.  GEMM 1 computes S^T = K Q^T (A = K tile from LDS, rows = keys):
//    the accumulator layout then has the QUERY in the lane (col = lane & 31) and keys in the registers,
//    which is exactly the B-operand layout GEMM 2 (O^T = V^T P^T) needs -- P goes from accumulator to
//    operand with a convert/pack only: no LDS round trip, no permlane.  The register->key order of the
//    accumulator ({0-3, 8-11 | 4-7, 12-15} per 16 keys and lane half) is absorbed by storing V^T in LDS
//    with the same key permutation, so the A fragment of GEMM 2 is one ds_read_b128.
//  * K tile [64 x d] row-major and V^T tile [d x 64] in LDS, rows padded by 16 B: every ds_read_b128 of
//    a fragment is bank-conflict free.  V is transposed on the way in (4 keys x 8 d per lane, v_perm
//    packs, ds_write_b64).
//  * per-sequence offsets (cu_seqlens) index the jagged batch; key tiles beyond what the mask of the
//    row block can reach are skipped (causal / contextual / target rules), the mask itself is applied
//    per element on the fp32 accumulator together with alpha, SiLU and 1/scaling_seqlen.
#include "common.h"
#include "../../include/recsys_amd.h"
#include <stdlib.h>
#include <type_traits>


// Operand type.  This file is compiled twice: as it stands (bf16 operands) and through hstu_attn_f16.hip, which defines
// HSTU_F16 (fp16 operands: hstu_api.cpp:359-366 accepts both).  Everything between memory and the MFMA is 16-bit data moved
// as bits, so what differs is the MFMA instruction, the fp32 -> 16-bit packing and the 16-bit -> fp32 read of the bias; the
// fp16 copy lives in an inline namespace of its own and its entry points carry the suffix _f16.
#ifndef HSTU_F16
#define HSTU_F16 0
#endif
#if HSTU_F16
#define HSTU_FN(name) name##_f16
#define HSTU_NS_BEGIN namespace mi355 { inline namespace hstu_f16 {
#define HSTU_NS_END } }
#else
#define HSTU_FN(name) name
#define HSTU_NS_BEGIN namespace mi355 {
#define HSTU_NS_END }
#endif

HSTU_NS_BEGIN

#if HSTU_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;   // (the name stays: "the 8-element MFMA operand")
#define HSTU_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
#define HSTU_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;   // first-class 16-B value (HIP's uint4 struct arrays end up in scratch)

#ifndef HSTU_XSTEP
#define HSTU_XSTEP 64   // rows per step of the one-GEMM backward passes (32 / 64)
#endif
#ifndef HSTU_XDB
#define HSTU_XDB 8      // fragments per MFMA batch there
#endif
#ifndef HSTU_XOCC
#define HSTU_XOCC 1     // blocks per CU the compiler must leave room for
#endif
constexpr int kBM = 128;  // query rows per workgroup (32 per wave)
constexpr int kBN = 64;   // keys per tile

#ifndef HSTU_QLDS_MIN
#define HSTU_QLDS_MIN 512   // head dims from which the forward keeps its Q fragments in LDS instead of registers (none)
#endif
#ifndef HSTU_VTR
#define HSTU_VTR 0           // forward: V tile row-major in LDS, V^T fragments through ds_read_b64_tr_b16 (no transposing commit).
                            // OFF: correct on every head dim (tests), but no faster -- d = 256: 591 vs 603 TFLOP/s at L = 4096,
                            // d = 128: +3 %; the transposing commit was not the bottleneck.  Kept for the backward rework.
#endif
#ifndef HSTU_BWD_TR
#define HSTU_BWD_TR 1        // backward: GEMM 3/4/5 A operands by transpose reads from row-major tiles (no transposed copies)
#endif
#ifndef HSTU_DB_MIN
#define HSTU_DB_MIN 1024    // head dims from which the forward double-buffers its K / V tiles in LDS (one barrier per tile).
                            // OFF: measured slower at d = 256 (L = 4096: 512 vs 586 TFLOP/s single-buffered, same at L = 512)
#endif

struct AttnArgs {
  const uint16_t* q; const uint16_t* k; const uint16_t* v;
  uint16_t* out;
  int64_t q_row, k_row, v_row, o_row;   // element stride between tokens
  int64_t q_head, k_head, v_head, o_head;  // element stride between heads
  const int* cu_seqlens;
  const int* num_contexts;  // [B] or nullptr
  const int* num_targets;   // [B] or nullptr
  int H, causal, group;
  int wl, wr;                  // local window: keys i - wl .. i + wr of query i (-1: unbounded on that side)
  int wskip;                   // 1: tile loops are clipped to the window's band (0: the mask alone applies it; for A/B tests)
  int rot, max_len, colmajor;  // block -> (sequence, head, rank) maps: see seq_head_of_block
  // relative attention bias (hstu_api.cpp:100-106,417-430): rab[b][h][i][j] (bf16, padded to max_seqlen_k in i and j) is
  // added to q_i . k_j before alpha and SiLU; head stride 0 = one bias matrix shared by all heads.  NULL: none.
  const uint16_t* rab; int64_t rab_b, rab_h, rab_r;
  // arbitrary mask functions (`func`, hstu_api.cpp:170-180; applied per element in hstu_fwd.h:139-145, 493-556): int32
  // func[h][p][t], p < n_func (odd), t = query token (row of q) -- token t sees key column j (position inside its sequence) iff
  // j < func[h][0][t] or func[h][2p-1][t] <= j < func[h][2p][t] for a p >= 1.  Read where the bias would be added: a masked pair
  // gets func_neg added to q.k (SiLU and SiLU' underflow to 0 exactly -- the dense-bias statement of the same mask, bit for
  // bit, without the [batch, heads, N, N] tensor).  Head stride 0: one function set for all heads.  NULL: none.
  const int32_t* func; int64_t func_h, func_p; int n_func; float func_neg;
  float alpha, inv_scale;
  // ---- inference extensions (forward only; NULL / 0 for training) ----
  const int* cu_seqlens_k;     // [B+1] key offsets when the keys are longer than the queries (delta-q); NULL = same as q
  const uint16_t* kv_cache;    // paged KV [num_pages, 2, page_size, H, d] (hstu_fwd.h Paged_KV paths :104-131,516-545)
  const int* page_offsets;     // [B+1] into page_ids
  const int* page_ids;         // page of every (sequence, page slot)
  const int* last_page_lens;   // [B] valid tokens of the last page
  int page_size;
};

struct SeqInfo { int start, L, c, hlen; bool has_ctx, has_tgt; int wl = -1, wr = -1; };

// M(i, j) of the reference (_get_valid_attn_mask / apply_mask); i, j are positions inside the sequence
__device__ __forceinline__ bool attn_allowed(int i, int j, const SeqInfo& s, int causal, int group) {
  const int idi = s.has_ctx ? (i - s.c + 1 > 0 ? i - s.c + 1 : 0) : i;
  const int idj = s.has_ctx ? (j - s.c + 1 > 0 ? j - s.c + 1 : 0) : j;
  bool ok = (i == j) || (causal ? idi > idj : idi != idj);
  if (s.has_tgt) {
    const int gi = i >= s.hlen ? (i - s.hlen) / group : -1;
    const int gj = j >= s.hlen ? (j - s.hlen) / group : -1;
    ok = ok && (gi == gj || gi < 0 || gj < 0);
  }
  if (s.has_ctx) ok = ok || (idi == 0 && j < s.hlen);
  return ok && i < s.L && j < s.L;
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  uint32_t r;
#if HSTU_F16
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#else
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
  return r;
}

__device__ __forceinline__ float silu_f(float x) {
  return x * __frcp_rn(1.0f + __expf(-x));
}

// MFMA wrappers.  The builtin (not inline asm) is deliberate: with asm MFMAs the hazard recogniser cannot see the
// instruction, and every variant tried (AGPR "+a" accumulators, VGPR "+v" accumulators, padded with s_nop, with a
// memory clobber) produced timing-dependent wrong sums on gfx950 although the same structure with builtins is
// bit-stable.  What the asm forms were meant to achieve -- keeping the long-lived output accumulators in AGPRs -- is
// done with pin_agpr() below, which costs no instructions.
__device__ __forceinline__ void mfma_a(f32x16_t& c, const bf16x8_t& a, const bf16x8_t& b) {
  c = HSTU_MFMA(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_v(f32x16_t& c, const bf16x8_t& a, const bf16x8_t& b) {
  c = HSTU_MFMA(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_v0(f32x16_t& c, const bf16x8_t& a, const bf16x8_t& b) {  // c = a b
  f32x16_t z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  c = HSTU_MFMA(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void mfma_fence() {}
// keeps a long-lived accumulator resident in AGPRs at this program point (the allocator otherwise splits its live
// range and parks it in VGPRs across the high-pressure staging code, paying a full copy in and out every iteration)
template <int N>
__device__ __forceinline__ void pin_agpr(f32x16_t (&c)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+a"(c[i]));
}
template <int N> __device__ __forceinline__ void fence_v(f32x16_t (&)[N]) {}   // builtins: the compiler inserts the waits
template <int N> __device__ __forceinline__ void fence_a(f32x16_t (&c)[N]) { pin_agpr(c); }
// The kernels that run TWO waves per SIMD (256 registers per wave) must not mention AGPRs at all: one "+a" constraint makes hipcc
// split the file 128 VGPRs + 128 AGPRs, and every value beyond the 128 -- the K / V / Q fragment sets of the S waves -- is then
// parked in AGPRs and copied back with four v_accvgpr_read_b32 (+ s_nop) in front of EVERY MFMA that uses it: 50 clocks per MFMA
// instead of 32 in the dK pass's S waves.  Without the pins the same kernels get up to 242 architectural VGPRs, no AGPRs, no copies.
#ifndef HSTU_2W_AGPR
#define HSTU_2W_AGPR 0
#endif
template <int N>
__device__ __forceinline__ void pin_agpr_2w(f32x16_t (&c)[N]) {
#if HSTU_2W_AGPR
  pin_agpr(c);
#endif
}
template <int N> __device__ __forceinline__ void fence_a_2w(f32x16_t (&c)[N]) { pin_agpr_2w(c); }

// Row-side mask state of one query (causal case).  M(i, j) of the reference collapses to
//   j <= jmax  and  (j < hlen  or  j >= jlo)
// with jmax = i (history / target rows) or hlen-1 (contextual rows: they see the whole history), and
// jlo = first key of the row's target group (0 for non-target rows).  Non causal: j < L.
// Local window (hstu_api.cpp:154-165; no contextual / target rows with it): additionally i - wl <= j <= i + wr.  The
// left bound reuses the target-group fields, which a window leaves free (hlen = 0: no key is "history", jlo = i - wl), so
// that key_ok costs the plain masks nothing extra.
struct RowMask { int jmax, jlo, hlen; };
__device__ __forceinline__ RowMask row_mask(int i, const SeqInfo& s, int causal, int group) {
  RowMask m;
  m.hlen = s.hlen;
  if (!causal) {
    m.jmax = s.L - 1; m.jlo = 0; m.hlen = s.L;
    if (s.wr >= 0 && i + s.wr < m.jmax) m.jmax = i + s.wr;
  } else {
    m.jmax = (s.has_ctx && i < s.c) ? s.hlen - 1 : i;
    if (m.jmax > s.L - 1) m.jmax = s.L - 1;
    m.jlo = (s.has_tgt && i >= s.hlen) ? s.hlen + ((i - s.hlen) / group) * group : 0;
  }
  if (s.wl >= 0) { m.hlen = 0; m.jlo = i - s.wl; }
  return m;
}
__device__ __forceinline__ bool key_ok(int j, const RowMask& m) {
  return (j <= m.jmax) & ((j < m.hlen) | (j >= m.jlo));
}
// Key tiles (steps of `step` keys) a block of query rows first .. last can reach through the window: [beg, end)
__device__ __forceinline__ int band_key_begin(const AttnArgs& a, int first_row, int step) {
  const int lo = first_row - a.wl;
  return (a.wskip && a.wl >= 0 && lo > 0) ? (lo / step) * step : 0;
}
__device__ __forceinline__ int band_key_end(const AttnArgs& a, int last_row, int end) {
  const int hi = last_row + a.wr + 1;
  return (a.wskip && a.wr >= 0 && hi < end) ? hi : end;
}

// Block -> (sequence, head).  The grid is (H, B, blocks) and the hardware deals consecutive workgroup ids round-robin over
// the 8 XCDs, so with b = blockIdx.y, h = blockIdx.x an XCD only ever sees the sequences of ONE residue class of
// (h + H b) mod 8 -- at H = 4 the sequences of one parity.  On a jagged batch whose few long sequences share a parity
// half the chip idles (C4 shape, 7 sequences of 4096 among 32: 5 odd, 2 even -> forward 830 us, 300 TFLOP/s).  Rotating
// the slot by H per dispatch rank -- sequence b + z takes the z-th heaviest block -- deals every sequence's row blocks over
// all XCDs: 442 us, 567 TFLOP/s on the same batch (rotation by 1 or H + 1: 632 / 503 us).  A batch whose sequences all
// have max_len rows gains nothing and keeps the plain grid, where a (sequence, head) column stays on one XCD's L2
// (rot < 0: rotate by -rot unless the batch is dense; rot > 0: always; 0: never).  All loads below are independent.
struct BlockSeq { int b, h, start, end, z; };   // z: the block's rank inside its (sequence, head) column
// A DENSE batch (every sequence has max_len rows) takes the column-major map instead (a.colmajor): the q blocks of one
// (sequence, head) column stream the SAME K / V tiles, and at L = 4096 the forward moves 64 KB of them per 8.4 MFLOP tile --
// 128 FLOP per byte, i.e. 640 TFLOP/s at the ~5 TB/s the fabric delivers, which is what the rank-major grid measures
// (645): there the 256 resident blocks are two ranks of ALL 128 columns, each column's tiles fetched again for every rank.
// Column-major: workgroup id i goes to XCD i mod 8 (hardware), and ids with the same residue are made the blocks of ONE
// column after another, heaviest first -- an XCD's 32 CUs then walk a column's key tiles in step and all but the first
// reader hit that XCD's L2.  Needs B * H to be a multiple of 8 (else the plain grid).  Measured: dense 32 x 4096 forward 1.71 ->
// 1.66 ms, backward 5.11 -> 4.94 ms; 8 x 4096 unchanged; C3 (4 blocks per column) 48 -> 58 us, hence columns of >= 16 blocks only.
// (bz, nz): the block's rank and the ranks of its launch -- blockIdx.z / gridDim.z, or a role's share of them where two passes
// ride in ONE launch (hstu_bwd_vq8_kernel)
__device__ __forceinline__ BlockSeq seq_head_of_block(const AttnArgs& a, unsigned bz, unsigned nz) {
  const int b0 = blockIdx.y, h0 = blockIdx.x, z0 = (int)bz;
  if (a.rot == 0 && a.colmajor == 0) return {b0, h0, a.cu_seqlens[b0], a.cu_seqlens[b0 + 1], z0};
  const unsigned bh = gridDim.x * gridDim.y;
  const unsigned step = (unsigned)(a.rot < 0 ? -a.rot : a.rot);
  const unsigned lin0 = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned slot = (lin0 + step * bz) % bh;
  const int b1 = (int)(slot / gridDim.x), h1 = (int)(slot - (unsigned)b1 * gridDim.x);
  // column-major candidate
  const unsigned lin = lin0 + bh * bz, xcd = lin & 7u, k = lin >> 3;
  const unsigned col = xcd + 8u * (k / nz);
  const int z2 = (int)(k % nz);
  const bool cm_ok = a.colmajor != 0 && (bh & 7u) == 0 && nz >= 16;   // (long columns only: at 4 blocks per column it costs 20 %)
  const int b2 = cm_ok ? (int)(col / gridDim.x) : b0, h2 = cm_ok ? (int)(col % gridDim.x) : h0;
  const int t0 = a.cu_seqlens[0], t1 = a.cu_seqlens[gridDim.y];
  const int s0 = a.cu_seqlens[b0], e0 = a.cu_seqlens[b0 + 1], s1 = a.cu_seqlens[b1], e1 = a.cu_seqlens[b1 + 1];
  const int s2 = a.cu_seqlens[b2], e2 = a.cu_seqlens[b2 + 1];
  const bool dense = t1 - t0 == (int)gridDim.y * a.max_len;
  if (dense) return cm_ok ? BlockSeq{b2, h2, s2, e2, z2} : BlockSeq{b0, h0, s0, e0, z0};
  if (a.rot > 0 || a.rot < 0) return BlockSeq{b1, h1, s1, e1, z0};
  return BlockSeq{b0, h0, s0, e0, z0};
}
__device__ __forceinline__ BlockSeq seq_head_of_block(const AttnArgs& a) { return seq_head_of_block(a, blockIdx.z, gridDim.z); }
// Row block a query-block owner of dispatch rank `rank` takes: heaviest first = the latest rows first (causal).  With contextual
// rows the FIRST block is the heaviest of all -- its contextual rows reach every history key -- and goes first: left at the
// end of the order it was a 8-tile tail behind a CU's other blocks (C3 shape with 4 contextual rows: forward 66 -> 5x us).
__device__ __forceinline__ int row_block_of_rank(int rank, int nblk, const AttnArgs& a, int b) {
  const bool ctx_first = a.causal && a.num_contexts != nullptr && a.num_contexts[b] > 0;
  if (!ctx_first) return nblk - 1 - rank;
  return rank == 0 ? 0 : nblk - rank;
}
#if HSTU_F16
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
#else
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
#endif
// S^T accumulators (lane = query row, registers = keys (rr & 3) + 8 (rr >> 2) + 4 hi of each 32-key sub-tile) += rab[i][.]
// `row` points at rab[b][h][i][0] (NULL for a row past the sequence); 2-byte loads: the bias path is not a tuned one.
template <int NT>
__device__ __forceinline__ void add_rab_row(f32x16_t (&acc)[NT], const uint16_t* row, int n0, int hi, int L) {
  if (row == nullptr) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
      if (key < L) acc[t][rr] += bf16_bits_to_f32(row[key]);
    }
}
// `func` masks: does query token `tok` see key column j?  (ft = &func[h][0][tok], fp = stride between the bounds)
__device__ __forceinline__ bool func_sees(const int32_t* ft, int64_t fp, int nf, int j) {
  bool ok = j < ft[0];
  for (int p = 1; p + 1 < nf; p += 2) ok |= (ft[p * fp] <= j) & (j < ft[(p + 1) * fp]);
  return ok;
}
// lane = query row (token `tok`), registers = keys as in add_rab_row: masked pairs get `neg` added
// `free_below`: keys below it are exempt for this row (a contextual row sees the whole history whatever the functions say,
// hstu_fwd.h:519-524: the context test `continue`s in front of every other mask)
template <int NT>
__device__ __forceinline__ void add_func_row(f32x16_t (&acc)[NT], const AttnArgs& a, int h, int64_t tok, bool row_live, int n0, int hi, int L,
                                             int free_below) {
  if (!row_live) return;
  const int32_t* ft = a.func + (int64_t)h * a.func_h + tok;
  unsigned ok[NT];
  int f0 = ft[0];
  f0 = f0 > free_below ? f0 : free_below;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ok[t] = 0;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) ok[t] |= (unsigned)((n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi) < f0) << rr;
  }
  for (int p = 1; p + 1 < a.n_func; p += 2) {
    const int lo = ft[p * a.func_p], up = ft[(p + 1) * a.func_p];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
        ok[t] |= (unsigned)((lo <= key) & (key < up)) << rr;
      }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
      if (key < L) acc[t][rr] += ((ok[t] >> rr) & 1u) ? 0.f : a.func_neg;
    }
}
// ---- `func` masks: tile skipping (round 6).  The reference derives the key blocks a query block can reach from the extents of
// its rows' functions before the tile loop and visits only those (hstu_fwd.h:80-83, 139-151, 228-291, 411-412: sValidBlockIds /
// sn_valid_block_max).  Here: what a group of query rows reaches = the prefix [0, f0) (largest first bound of the group) and the
// hull [lo, hi) of every band of every row; a key tile that meets neither is skipped.  Conservative (a hull may cover a gap
// between two bands) and exact in its effect: a skipped tile's P is all zero.
struct FuncExt { int f0, lo, hi, f0min; };   // lo >= hi: no band; f0min: the SMALLEST prefix of the group -- keys below it are seen by every row
__device__ __forceinline__ FuncExt func_ext_row(const AttnArgs& a, int h, int64_t tok, bool live, int free_below) {
  FuncExt e{0, 0x7fffffff, 0, 0x7fffffff};
  if (!live) return e;
  const int32_t* ft = a.func + (int64_t)h * a.func_h + tok;
  const int f0 = ft[0];
  e.f0 = f0 > free_below ? f0 : free_below;
  e.f0min = e.f0;
  for (int p = 1; p + 1 < a.n_func; p += 2) {
    const int lo = ft[p * a.func_p], up = ft[(p + 1) * a.func_p];
    if (lo < up) { e.lo = lo < e.lo ? lo : e.lo; e.hi = up > e.hi ? up : e.hi; }
  }
  return e;
}
__device__ __forceinline__ FuncExt func_ext_merge(const FuncExt& x, const FuncExt& y) {
  return FuncExt{x.f0 > y.f0 ? x.f0 : y.f0, x.lo < y.lo ? x.lo : y.lo, x.hi > y.hi ? x.hi : y.hi, x.f0min < y.f0min ? x.f0min : y.f0min};
}
// over the 32 rows of a wave (lane & 31 = row, both halves hold the same rows)
__device__ __forceinline__ FuncExt func_ext_wave(FuncExt e) {
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    FuncExt o{__shfl_xor(e.f0, off, 64), __shfl_xor(e.lo, off, 64), __shfl_xor(e.hi, off, 64), __shfl_xor(e.f0min, off, 64)};
    e = func_ext_merge(e, o);
  }
  return e;
}
// over the 64 rows of a wave whose lanes hold one row each
__device__ __forceinline__ FuncExt func_ext_wave64(FuncExt e) {
  e = func_ext_wave(e);
  const FuncExt o{__shfl_xor(e.f0, 32, 64), __shfl_xor(e.lo, 32, 64), __shfl_xor(e.hi, 32, 64), __shfl_xor(e.f0min, 32, 64)};
  return func_ext_merge(e, o);
}
__device__ __forceinline__ bool func_ext_hits(const FuncExt& e, int n0, int n1) { return n0 < e.f0 || (e.lo < n1 && e.hi > n0); }
__device__ __forceinline__ int func_ext_end(const FuncExt& e) { return e.f0 > e.hi ? e.f0 : e.hi; }                    // keys from here on: unseen
__device__ __forceinline__ int func_ext_begin(const FuncExt& e) { return e.f0 > 0 ? 0 : (e.lo < e.hi ? e.lo : 0x7fffffff); }   // keys below: unseen
// block-wide merge of the waves' extents (NW waves; one barrier; `slot` = 3 * NW ints of LDS)
template <int NW>
__device__ __forceinline__ FuncExt func_ext_block(const FuncExt& wx, int wv, int lane, int* slot) {
  if (lane == 0) { slot[3 * wv] = wx.f0; slot[3 * wv + 1] = wx.lo; slot[3 * wv + 2] = wx.hi; }
  __syncthreads();
  FuncExt bx{slot[0], slot[1], slot[2], 0};
#pragma unroll
  for (int w = 1; w < NW; ++w) bx = func_ext_merge(bx, FuncExt{slot[3 * w], slot[3 * w + 1], slot[3 * w + 2], 0});
  return bx;
}
// index of key block n0 / 128 of sequence b in the per-key-block table of the backward (hstu_func_kvis_kernel): sequences start at
// arbitrary tokens, floor(start / 128) + b leaves every sequence ceil(L / 128) slots of its own
__device__ __forceinline__ int64_t func_kvis_index(int start, int b, int n0) { return (int64_t)(start >> 7) + b + (n0 >> 7); }

// For the key-major passes of the backward: per (function set, sequence, 128-key block) the range of query rows [first, last) that
// reach the block (multiples of 32; first >= last: nobody does).  One block per (sequence, function set): the extents of the
// sequence's 32-row groups in LDS, then one thread per key block walks them.
// gext (second table, 4 x as many entries): the extents {smallest prefix, largest prefix, band hull lo, hi} of every 32-row group,
// entry floor(start / 32) + b + group -- a key-major step asks it whether its tile needs the per-element test at all.
__device__ __forceinline__ int64_t func_gext_index(int start, int b, int row) { return (int64_t)(start >> 5) + b + (row >> 5); }
__global__ void __launch_bounds__(256)
hstu_func_kvis_kernel(AttnArgs a, int nfh, int2* __restrict__ kvis, int64_t kvis_h, int4* __restrict__ gext) {
  extern __shared__ int s_ext[];      // [4][ng]
  const int b = blockIdx.x, fh = blockIdx.y;
  const int start = a.cu_seqlens[b], L = a.cu_seqlens[b + 1] - start;
  const int ng = (L + 31) >> 5, nkb = (L + 127) >> 7;
  int* e_f0 = s_ext; int* e_lo = s_ext + ng; int* e_hi = s_ext + 2 * ng; int* e_fm = s_ext + 3 * ng;
  for (int g = threadIdx.x; g < ng; g += blockDim.x) { e_f0[g] = 0; e_lo[g] = 0x7fffffff; e_hi[g] = 0; e_fm[g] = 0x7fffffff; }
  __syncthreads();
  const int c = a.num_contexts ? a.num_contexts[b] : 0;
  const int hlen = L - (a.num_targets ? a.num_targets[b] : 0);
  for (int r = threadIdx.x; r < L; r += blockDim.x) {
    const FuncExt e = func_ext_row(a, fh, (int64_t)start + r, true, (a.num_contexts && r < c) ? hlen : 0);
    atomicMax(&e_f0[r >> 5], e.f0);
    atomicMin(&e_lo[r >> 5], e.lo);
    atomicMax(&e_hi[r >> 5], e.hi);
    atomicMin(&e_fm[r >> 5], e.f0min);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < ng; g += blockDim.x) {
    const int64_t idx = func_gext_index(start, b, g << 5);
    if (idx < 4 * kvis_h) gext[(int64_t)fh * 4 * kvis_h + idx] = make_int4(e_fm[g], e_f0[g], e_lo[g], e_hi[g]);
  }
  for (int kb = threadIdx.x; kb < nkb; kb += blockDim.x) {
    const int k0 = kb << 7, k1 = k0 + 128;
    int first = 0x7fffffff, last = 0;
    for (int g = 0; g < ng; ++g) {
      const FuncExt e{e_f0[g], e_lo[g], e_hi[g], 0};
      if (func_ext_hits(e, k0, k1)) { first = first < (g << 5) ? first : (g << 5); last = (g + 1) << 5; }
    }
    const int64_t idx = func_kvis_index(start, b, k0);
    if (idx < kvis_h) kvis[(int64_t)fh * kvis_h + idx] = make_int2(first, last);
  }
  (void)nfh;
}

// dS = dP * (alpha / N) * SiLU'(x), x = alpha * acc, sg = sigmoid(x).  One statement of the roundings for every kernel that
// forms dS (the dK pass and the recomputing dQ pass must agree bit for bit with each other and with the exchanged dS):
// contraction is switched off and the one fused step is spelled out, so that the code around a call site cannot change it.
__device__ __forceinline__ float ds_value(float acc, float dp, float sg, float alpha, float c_ds) {
#pragma clang fp contract(off)
  const float x = acc * alpha;
  const float inner = __builtin_fmaf(x, 1.0f - sg, 1.0f);
  return dp * c_ds * sg * inner;
}
// SiLU(alpha * acc) * inv_scale from the raw accumulator: 4 plain VALU + 2 transcendental ops
__device__ __forceinline__ float silu_scaled(float acc, float neg_alpha_log2e, float alpha_inv_scale) {
  const float t = __builtin_amdgcn_exp2f(acc * neg_alpha_log2e);
  return acc * alpha_inv_scale * __builtin_amdgcn_rcpf(1.0f + t);
}

// rows [row0, row0+NR) of a [*, H, D] tensor -> LDS [NR][D+8] row-major (zeros beyond nvalid)
template <int D, int NR>
__device__ __forceinline__ void stage_rows(uint16_t* dst, const uint16_t* src, int64_t row_stride, int row0, int nvalid) {
  constexpr int NCH = NR * D / 8;
#pragma unroll
  for (int ch = threadIdx.x; ch < NCH; ch += 256) {
    const int r = ch / (D / 8), dc = ch % (D / 8);
    uint4 t = make_uint4(0, 0, 0, 0);
    if (row0 + r < nvalid) t = *reinterpret_cast<const uint4*>(src + (int64_t)(row0 + r) * row_stride + 8 * dc);
    *reinterpret_cast<uint4*>(dst + r * (D + 8) + 8 * dc) = t;
  }
}
#ifndef HSTU_TIMING
#define HSTU_TIMING 0
#endif
#ifndef HSTU_XCH_NT
#define HSTU_XCH_NT 0   // 1 = P / dS exchange tiles with non-temporal stores (dK pass) and loads (one-GEMM passes): written once, read once,
                        // 0.5 GB each at 8 x 4096, they evict the Q / dO / K rows the same kernels stream through L2 again and again (TCC
                        // counters: those rows miss L2 3-7 x, profiles/r04_pmc_hstu_traffic.txt).  Measured: -1..-2 % (8 x 4096 backward 648 against
                        // 659-666 TFLOP/s, C3 140-149 against 137-144 us): off.
#endif
__device__ __forceinline__ void xch_store(u32x4_t* p, const u32x4_t& v) {
#if HSTU_XCH_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ u32x4_t xch_load(const u32x4_t* p) {
#if HSTU_XCH_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
#ifndef HSTU_X8_PROBE
#define HSTU_X8_PROBE 0   // timing probes of the one-GEMM passes (results wrong): 1 = no fragment reads after the first batch, 2 = no exchange loads, 4 = no DMA after the first step
#endif
#if HSTU_TIMING
__device__ unsigned long long g_hstu_dbg[8 * 65536];
__device__ __forceinline__ unsigned tick() {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define TICK(v) const unsigned v = tick()
#define TACC(i, a, b) tsum[i] += (b) - (a)
#else
#define TICK(v)
#define TACC(i, a, b)
#endif

template <int D, bool kWin = false, bool kRab = false>   // kWin: local window, kRab: attention bias; variants of their own so that the plain path pays nothing
__global__ void __launch_bounds__(256) hstu_fwd_kernel(AttnArgs a) {
  constexpr int KS = D + 8;    // padded K row (elements)
  constexpr bool kVTR = HSTU_VTR != 0;
  // V tile in LDS: kVTR -- row-major [kBN][VR] and the V^T fragments of GEMM 2 come out of the hardware transpose read
  // (row stride D + 32 elements = 16 dwords mod 64: the 32 lanes of a half-wave hit 64 distinct banks); otherwise
  // transposed [D][VS] by the committing threads (perm + 8-byte stores)
  constexpr int VS = kBN + 8;  // padded V^T row (elements)
  constexpr int VR = D == 32 ? 32 : D + 32;   // V row (elements): row stride = 16 dwords mod 64 (48 at d = 64: also conflict free)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int TILE = kBN * KS + (kVTR ? kBN * VR : D * VS);   // elements of one staged (K, V) tile pair
  // Optional (HSTU_DB_MIN, off by default): the tile pair DOUBLE-BUFFERED in LDS -- tile n+1 is committed into the other
  // buffer inside the barrier interval in which tile n is consumed, one barrier per key tile instead of two.
  constexpr bool kDB = D >= HSTU_DB_MIN;
  uint16_t* Ks = smem;                 // [kBN][KS]           (of the tile being consumed)
  uint16_t* Vt = smem + kBN * KS;      // [D][VS], key positions permuted inside every 16-group
  constexpr bool QLDS = D >= HSTU_QLDS_MIN;      // Q fragments in LDS instead of 64 VGPRs (not used: 452 registers fit at d = 256)
  uint16_t* Qs = smem + (kDB ? 2 : 1) * TILE;    // [kBM][KS] (QLDS only)

  const BlockSeq bs = seq_head_of_block(a);   // grid (H, B, blocks): see launch_fwd
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  const int Lq = bs.end - s.start;
  // keys: the same tokens (training), a longer key sequence (delta-q: the queries are its LAST Lq positions), or the
  // user's paged cache followed by the candidate tokens of k / v
  const int kstart = a.cu_seqlens_k ? a.cu_seqlens_k[b] : s.start;
  s.L = a.cu_seqlens_k ? a.cu_seqlens_k[b + 1] - kstart : Lq;
  const int dq = s.L - Lq;   // absolute position of query row r is dq + r
  const int nblk = (Lq + kBM - 1) / kBM;
  if (bs.z >= nblk || dq < 0) return;
  const int m0 = row_block_of_rank(bs.z, nblk, a, b) * kBM;  // heaviest row blocks first
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  const int ntgt = s.has_tgt ? a.num_targets[b] : 0;
  s.hlen = s.L - ntgt;
  s.wl = kWin ? a.wl : -1; s.wr = kWin ? a.wr : -1;
  const bool paged = a.kv_cache != nullptr;
  int cachelen = 0, pg0 = 0;
  if (paged) {
    pg0 = a.page_offsets[b];
    cachelen = (a.page_offsets[b + 1] - pg0 - 1) * a.page_size + a.last_page_lens[b];
    if (cachelen < 0) cachelen = 0;
  }

  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qrow0 = m0 + 32 * wv;          // local (query tensor) row
  const int qloc = qrow0 + l31;
  const int qi = dq + qloc;                // absolute position
  const bool wave_live = qrow0 < Lq;

  // keys this row block can reach (absolute positions)
  int last_row = dq + (m0 + kBM - 1 < Lq - 1 ? m0 + kBM - 1 : Lq - 1);
  int n_end = s.L;
  if (a.causal) {
    n_end = last_row + 1;
    if (s.has_ctx && dq + m0 < s.c && s.hlen > n_end) n_end = s.hlen;
  }
  if (kWin) n_end = band_key_end(a, last_row, n_end);
  int n_beg = kWin ? band_key_begin(a, dq + m0, kBN) : 0;
  // the wave's own reach (skips MFMA work on tiles past it)
  int w_last = dq + (qrow0 + 31 < Lq - 1 ? qrow0 + 31 : Lq - 1);
  int w_end = s.L;
  if (a.causal) {
    w_end = w_last + 1;
    if (s.has_ctx && dq + qrow0 < s.c && s.hlen > w_end) w_end = s.hlen;
  }
  if (kWin) w_end = band_key_end(a, w_last, w_end);
  int w_beg = kWin ? band_key_begin(a, dq + qrow0, kBN) : 0;
  // `func` masks: the key tiles the block / the wave can reach at all, from the extents of their rows' functions
  FuncExt wx{0x7fffffff, 0, 0x7fffffff, 0};       // (no functions: every tile "hits"; with functions and no skipping: no tile is "full")
  if constexpr (kRab) {
    if (a.func && a.wskip) {
      __shared__ int s_fx[3 * 4];
      wx = func_ext_wave(func_ext_row(a, h, (int64_t)s.start + qloc, qloc < Lq, (s.has_ctx && qi < s.c) ? s.hlen : 0));
      const FuncExt bx = func_ext_block<4>(wx, wv, lane, s_fx);
      const int be = func_ext_end(bx), bb = func_ext_begin(bx), we = func_ext_end(wx), wb = func_ext_begin(wx);
      if (be < n_end) n_end = be;
      if (we < w_end) w_end = we;
      if (bb > n_beg) n_beg = bb >= n_end ? n_end : (bb / kBN) * kBN;
      if (wb > w_beg) w_beg = wb >= 0x7fffff00 ? 0x7fffff00 : (wb / kBN) * kBN;
    }
  }

  // ---- Q fragments (B operand of GEMM 1): lane = (query l31, k half hi), 8 consecutive d per 16-slice
  bf16x8_t qf[QLDS ? 1 : D / 16];
  if constexpr (QLDS) {
    stage_rows<D, kBM>(Qs, a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head, a.q_row, m0, Lq);
  } else {
    const uint16_t* qp = a.q + (int64_t)(s.start + (qloc < Lq ? qloc : 0)) * a.q_row + (int64_t)h * a.q_head + 8 * hi;
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (qloc < Lq) t = *reinterpret_cast<const uint4*>(qp + 16 * sl);
      qf[sl] = *reinterpret_cast<bf16x8_t*>(&t);
    }
  }

  f32x16_t acc_o[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;

  const float nal2e = -a.alpha * 1.44269504088896f, ais = a.alpha * a.inv_scale;
  const RowMask rm = row_mask(qi < s.L ? qi : s.L - 1, s, a.causal, a.group);
  // ---- software-pipelined staging (issue-early / write-late): the K / V tile of step n+1 is fetched
  // into registers while step n computes; it is written to LDS (V transposed) after the barrier.
  constexpr int KCH = kBN * D / 8;            // 16-B chunks of the K tile
  constexpr int KPT = (KCH + 255) / 256;      // per thread
  constexpr int VCH = (kBN / 4) * (D / 8);    // (4 keys x 8 d) blocks of the V tile
  constexpr int VPT = (VCH + 255) / 256;
  u32x4_t kreg[KPT], vreg[kVTR ? 1 : VPT][4], vrow[kVTR ? KPT : 1];
  // key j of the sequence: cached token (page table walk) or a token of k / v.  With a cache, k / v hold
  // [new history | candidates] per sequence and only the candidates are read from them (the history is in the cache).
  // (the keys that are not in the cache are the LAST Lk - cachelen rows of the sequence's k / v: key j >= cachelen is row Lq - Lk + j.
  //  With the candidates as targets that is the former Lq - ntgt - cachelen + j; written this way it also holds without targets,
  //  e.g. under a local window)
  const int64_t tok0 = paged ? (int64_t)s.start + Lq - s.L : (int64_t)kstart;
  const uint16_t* kbase = a.k + (int64_t)h * a.k_head;
  const uint16_t* vbase = a.v + (int64_t)h * a.v_head;
  const int64_t pg_slot = (int64_t)a.H * D, pg_kv = (int64_t)a.page_size * pg_slot;
  auto kv_row = [&](int j, int which) -> const uint16_t* {
    const uint16_t* direct = (which ? vbase : kbase) + (tok0 + j) * (which ? a.v_row : a.k_row);
    if (!paged) return direct;
    const int jc = j < cachelen ? j : 0;
    const int page = a.page_ids[pg0 + jc / a.page_size];
    const uint16_t* cached = a.kv_cache + ((int64_t)page * 2 + which) * pg_kv + (int64_t)(jc % a.page_size) * pg_slot + (int64_t)h * D;
    return j < cachelen ? cached : direct;
  };
  // Fast path of the fetch (contiguous keys, tile entirely inside the sequence): per-thread base pointers are formed
  // once, a tile costs one scalar multiply and its loads one 64-bit add each (K) or an immediate offset (V).  The general
  // path below recomputes (token, page) -> address per 16-byte load: ~20 VALU with quarter-rate 32-bit multiplies, 16 to
  // 40 loads per thread and tile -- as many VALU cycles as the tile's MFMAs take.
  constexpr int KROWS = 256 / (D / 8);        // K rows covered by the 256 threads per load round
  const uint16_t* k_thr = kbase + (tok0 + (int)threadIdx.x / (D / 8)) * a.k_row + 8 * ((int)threadIdx.x % (D / 8));
  const uint16_t* vr_thr = vbase + (tok0 + (int)threadIdx.x / (D / 8)) * a.v_row + 8 * ((int)threadIdx.x % (D / 8));
  const int64_t kstep = (int64_t)KROWS * a.k_row, vrstep = (int64_t)KROWS * a.v_row;
  const uint16_t* v_thr;
  {
    const int kgpos = (int)threadIdx.x % (kBN / 4), g16 = kgpos >> 2, pg = kgpos & 3;
    const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);
    v_thr = vbase + (tok0 + 16 * g16 + 4 * ak) * a.v_row + 8 * ((int)threadIdx.x / (kBN / 4));
  }
  auto fetch = [&](int n0) {
    // (d = 256 only: the kernel runs one wave per SIMD there anyway; at d = 128 the extra pointers cost the second wave)
    if (D >= 256 && !paged && n0 + kBN <= s.L) {
      const uint16_t* kp = k_thr + (int64_t)n0 * a.k_row;
#pragma unroll
      for (int i = 0; i < KPT; ++i) kreg[i] = *reinterpret_cast<const u32x4_t*>(kp + i * kstep);
      if constexpr (kVTR) {
        const uint16_t* vp = vr_thr + (int64_t)n0 * a.v_row;
#pragma unroll
        for (int i = 0; i < KPT; ++i) vrow[i] = *reinterpret_cast<const u32x4_t*>(vp + i * vrstep);
      } else {
        const uint16_t* vp = v_thr + (int64_t)n0 * a.v_row;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int i = 0; i < VPT; ++i)
            if (VCH % 256 == 0 || (int)threadIdx.x + 256 * i < VCH)
              vreg[i][kk] = *reinterpret_cast<const u32x4_t*>(vp + 8 * (256 / (kBN / 4)) * i);
          vp += a.v_row;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int key = ch / (D / 8), dc = ch % (D / 8);
      // rows past the sequence end are clamped to its last row: their P is masked to 0, so any finite data do
      const int row = n0 + key < s.L ? n0 + key : s.L - 1;
      if (KCH % 256 == 0 || ch < KCH) kreg[i] = *reinterpret_cast<const u32x4_t*>(kv_row(row, 0) + 8 * dc);
    }
    if constexpr (kVTR) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {   // V rows exactly like K rows
        const int ch = threadIdx.x + 256 * i;
        const int key = ch / (D / 8), dc = ch % (D / 8);
        const int row = n0 + key < s.L ? n0 + key : s.L - 1;
        if (KCH % 256 == 0 || ch < KCH) vrow[i] = *reinterpret_cast<const u32x4_t*>(kv_row(row, 1) + 8 * dc);
      }
    } else {
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int kgpos = ch % (kBN / 4), dc = ch / (kBN / 4);
      const int g16 = kgpos >> 2, pg = kgpos & 3;
      const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);  // position group -> actual key group
      const int key0 = 16 * g16 + 4 * ak;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int row = n0 + key0 + kk < s.L ? n0 + key0 + kk : s.L - 1;
        if (VCH % 256 == 0 || ch < VCH) vreg[i][kk] = *reinterpret_cast<const u32x4_t*>(kv_row(row, 1) + 8 * dc);
      }
    }
    }
  };
  auto commit = [&](uint16_t* Kd, uint16_t* Vd) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int key = ch / (D / 8), dc = ch % (D / 8);
      if (KCH % 256 == 0 || ch < KCH) *reinterpret_cast<u32x4_t*>(Kd + key * KS + 8 * dc) = kreg[i];
    }
    if constexpr (kVTR) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int ch = threadIdx.x + 256 * i;
        const int key = ch / (D / 8), dc = ch % (D / 8);
        if (KCH % 256 == 0 || ch < KCH) *reinterpret_cast<u32x4_t*>(Vd + key * VR + 8 * dc) = vrow[i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      if (VCH % 256 != 0 && ch >= VCH) continue;
      const int kgpos = ch % (kBN / 4), dc = ch / (kBN / 4);
      const int g16 = kgpos >> 2, pg = kgpos & 3;
      const u32x4_t w0 = vreg[i][0], w1 = vreg[i][1], w2 = vreg[i][2], w3 = vreg[i][3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
        uint2 o;
        o.x = __builtin_amdgcn_perm(w1[e >> 1], w0[e >> 1], sel);
        o.y = __builtin_amdgcn_perm(w3[e >> 1], w2[e >> 1], sel);
        *reinterpret_cast<uint2*>(Vd + (8 * dc + e) * VS + 16 * g16 + 4 * pg) = o;
      }
    }
  };

  // A operand of GEMM 2: V^T[32 d of tile dt][16 keys of slice ks], lane (d = l31, k half = hi) holds the 8 keys
  // (j&3) + 8*(j>>2) + 4*hi of the slice (the register order of the S accumulator, see ew()).
  // kVTR: two hardware transpose reads per fragment.  Within a 16-lane group the read returns to lane l column l of the
  // 4 x 16 block whose row (i>>2), columns 4*(i&3)..+3 lane i points at: out_l[j] = in_{4j + (l>>2)}[l & 3] (probed on
  // gfx950).  Groups: lanes 16g..16g+15 = d half (g & 1), k half (g >> 1) of the fragment.
  const int tr_il = lane & 15;
  const int tr_off = ((4 * hi + (tr_il >> 2)) * VR + 16 * ((lane >> 4) & 1) + 4 * (tr_il & 3));   // elements, inside (slice, d tile)
  auto v_frag = [&](const uint16_t* Vb, int dt, int ks) -> bf16x8_t {
    if constexpr (kVTR) {
      typedef short v4s_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
      const uint16_t* p0 = Vb + (16 * ks) * VR + 32 * dt + tr_off;
      const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
      const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * VR));
      typedef short v8s_t __attribute__((ext_vector_type(8)));
      const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
      return __builtin_bit_cast(bf16x8_t, r);
    } else {
      return *reinterpret_cast<const bf16x8_t*>(Vb + (32 * dt + l31) * VS + 16 * ks + 8 * hi);
    }
  };

  if (n_end > n_beg) {
    fetch(n_beg);
    if constexpr (kDB) {
      commit(smem, smem + kBN * KS);
      if (n_beg + kBN < n_end) fetch(n_beg + kBN);
    }
  }
  int it = 0;
#if HSTU_TIMING
  unsigned tsum[7] = {0, 0, 0, 0, 0, 0, 0};
  const unsigned t_start = tick();
#endif
  for (int n0 = n_beg; n0 < n_end; n0 += kBN, ++it) {
    pin_agpr(acc_o);
    TICK(t0);
    __syncthreads();   // kDB: everyone is done with the other buffer, and this tile's commit (previous interval) is visible
    if constexpr (kDB) {
      uint16_t* cur = smem + (it & 1) * TILE;
      uint16_t* oth = smem + ((it & 1) ^ 1) * TILE;
      Ks = cur;
      Vt = cur + kBN * KS;
      if (n0 + kBN < n_end) {
        commit(oth, oth + kBN * KS);                     // tile n0 + kBN (in registers since the previous interval)
        if (n0 + 2 * kBN < n_end) fetch(n0 + 2 * kBN);
      }
    } else {
      TICK(t1);
      commit(Ks, Vt);
      pin_agpr(acc_o);
      TICK(t2);
      __syncthreads();
      TICK(t3);
      if (n0 + kBN < n_end) fetch(n0 + kBN);
      TICK(t4);
      TACC(0, t0, t1); TACC(1, t1, t2); TACC(2, t2, t3); TACC(3, t3, t4);
    }
    pin_agpr(acc_o);
    if (!wave_live || n0 >= w_end || n0 < w_beg) continue;
    if constexpr (kRab) { if (!func_ext_hits(wx, n0, n0 + kBN)) continue; }     // (a gap between the prefix and the bands)
    TICK(t5);

    // ---- GEMM 1: S^T[64 keys x 32 q] = K Q^T, two 32-key tiles.  Operand fragments are fetched from LDS
    // in batches of 8 ahead of the 8 MFMAs that consume them (hipcc otherwise emits read-wait-mfma triples).
    f32x16_t acc_s[2];
    {
      // fragment batches are double-buffered: the LDS reads of batch n+1 are in flight while the 8 MFMAs of batch n
      // issue (one wave per SIMD at d = 256: nothing else hides the LDS latency)
      constexpr int SLB = D / 16 < 4 ? D / 16 : 4;   // 16-wide d slices per batch
      constexpr int NBAT = (D / 16) / SLB;
      bf16x8_t kfr[2][SLB][2], qfr[2][SLB];
      auto load_b = [&](int bi, int buf) {
#pragma unroll
        for (int u = 0; u < SLB; ++u) {
          const int sl = SLB * bi + u;
          if constexpr (QLDS) qfr[buf][u] = *reinterpret_cast<const bf16x8_t*>(Qs + (32 * wv + l31) * KS + 16 * sl + 8 * hi);
          else qfr[buf][u] = qf[sl];
#pragma unroll
          for (int t = 0; t < 2; ++t)
            kfr[buf][u][t] = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + l31) * KS + 16 * sl + 8 * hi);
        }
      };
      load_b(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT; ++bi) {
        if (bi + 1 < NBAT) load_b(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SLB; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (bi == 0 && u == 0) mfma_v0(acc_s[t], kfr[bi & 1][u][t], qfr[bi & 1][u]);
            else mfma_v(acc_s[t], kfr[bi & 1][u][t], qfr[bi & 1][u]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    fence_v(acc_s);
    if constexpr (kRab) {
      if (a.func) {     // (a tile below the smallest prefix of the wave's rows is seen by all of them: nothing to test)
        if (n0 + kBN > wx.f0min) add_func_row<2>(acc_s, a, h, (int64_t)s.start + qloc, qloc < Lq, n0, hi, s.L, (s.has_ctx && qi < s.c) ? s.hlen : 0);
      } else {
      const uint16_t* row = qi < s.L ? a.rab + (int64_t)b * a.rab_b + (int64_t)h * a.rab_h + (int64_t)qi * a.rab_r : nullptr;
      add_rab_row<2>(acc_s, row, n0, hi, s.L);
      }
    }
    pin_agpr(acc_o);
    TICK(t6);
    TACC(4, t5, t6);
    // ---- P^T = mask * SiLU(alpha S^T) / scaling, packed straight into the B operand of GEMM 2.
    // Tiles strictly below the diagonal of every row of the wave need no per-element mask.
    const bool full = a.causal && s.wl < 0 && (n0 + kBN - 1 <= dq + qrow0) && (!s.has_ctx || dq + qrow0 >= s.c) && (!s.has_tgt || n0 + kBN - 1 < s.hlen);
    if constexpr (D >= 256) {
    // ---- GEMM 2: O^T[D x 32 q] += V^T[D x 64 keys] P^T, software-pipelined with the SiLU epilogue of GEMM 1: the
    // 16-key slice ks+1 of P is computed (VALU + transcendental pipes) while the MFMAs of slice ks run -- at one wave
    // per SIMD nothing else overlaps the ~1.8 K cycles of SiLU per tile with the ~2 K cycles of MFMA.  `full` is
    // hoisted out as a compile-time variant so that the pipelined region is one basic block.
    constexpr int NDT = D / 32;
    constexpr int DB = NDT < 8 ? NDT : 8;
    pin_agpr(acc_o);
    auto gemm2 = [&](auto fullc) {
      constexpr bool kFull = decltype(fullc)::value;
      auto ew = [&](int ks) -> bf16x8_t {   // P^T slice ks: keys 16*ks .. 16*ks+15 of the tile (register order of acc_s)
        const int t = ks >> 1, r0 = (ks & 1) * 8;
        uint32_t pk[4];
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          float p2[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rr = r0 + r + u;
            const float pv = silu_scaled(acc_s[t][rr], nal2e, ais);
            if (kFull) p2[u] = pv;
            else {
              const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
              p2[u] = key_ok(key, rm) ? pv : 0.f;
            }
          }
          pk[r >> 1] = pack_bf16(p2[0], p2[1]);
        }
        const u32x4_t x = {pk[0], pk[1], pk[2], pk[3]};
        return __builtin_bit_cast(bf16x8_t, x);
      };
      constexpr int NBAT2 = 4 * (NDT / DB);
      bf16x8_t vfr[2][DB];
      auto load_v = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u)
          vfr[buf][u] = v_frag(Vt, dt0 + u, ks);
      };
      load_v(0, 0);
      constexpr bool kPipe = true;
      bf16x8_t pf[4];
      pf[0] = ew(0);
      if constexpr (!kPipe) { pf[1] = ew(1); pf[2] = ew(2); pf[3] = ew(3); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
        const bool last_of_ks = (bi % (NDT / DB)) == (NDT / DB) - 1;
        if (bi + 1 < NBAT2) load_v(bi + 1, (bi + 1) & 1);
        if constexpr (kPipe) {
          if (last_of_ks && ks + 1 < 4) pf[ks + 1] = ew(ks + 1);   // independent of this batch's MFMAs: fills their shadow
        } else {
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < DB; ++u) mfma_a(acc_o[dt0 + u], vfr[bi & 1][u], pf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (full) gemm2(std::true_type{}); else gemm2(std::false_type{});
    TICK(t7);
    TACC(5, t6, t7);
#if HSTU_TIMING
    tsum[6] += 1;
#endif
    } else {
    // smaller head dims: 2-3 waves per SIMD already overlap SiLU with another wave's MFMAs, and the pipelined form costs
    // ~30 VGPRs (one occupancy step at d = 128)
    bf16x8_t pf[4];
    if (full) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          pk[r >> 1] = pack_bf16(silu_scaled(acc_s[t][r], nal2e, ais), silu_scaled(acc_s[t][r + 1], nal2e, ais));
        uint4 lo4 = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi4 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        pf[2 * t] = *reinterpret_cast<bf16x8_t*>(&lo4);
        pf[2 * t + 1] = *reinterpret_cast<bf16x8_t*>(&hi4);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p2[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rr = r + u;
            const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            const float pv = silu_scaled(acc_s[t][rr], nal2e, ais);
            p2[u] = key_ok(key, rm) ? pv : 0.f;
          }
          pk[r >> 1] = pack_bf16(p2[0], p2[1]);
        }
        uint4 lo4 = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi4 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        pf[2 * t] = *reinterpret_cast<bf16x8_t*>(&lo4);
        pf[2 * t + 1] = *reinterpret_cast<bf16x8_t*>(&hi4);
      }
    }
    // ---- GEMM 2: O^T[D x 32 q] += V^T[D x 64 keys] P^T: key slice outer, so consecutive MFMAs hit
    // independent accumulators; fragments again fetched in batches
    constexpr int NDT = D / 32;
    constexpr int DB = NDT < 8 ? NDT : 8;
    pin_agpr(acc_o);
    {
      constexpr int NBAT2 = 4 * (NDT / DB);
      bf16x8_t vfr[2][DB];
      auto load_v = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u)
          vfr[buf][u] = v_frag(Vt, dt0 + u, ks);
      };
      load_v(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        if (bi + 1 < NBAT2) load_v(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) mfma_a(acc_o[dt0 + u], vfr[bi & 1][u], pf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
  }
  fence_a(acc_o);
#if HSTU_TIMING
  {
    const unsigned t_end = tick();
    if (lane == 0) {
      const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long* d = g_hstu_dbg + ((size_t)(blk * 4 + wv) % 65536) * 8;
      for (int i = 0; i < 7; ++i) d[i] = tsum[i];
      d[7] = t_end - t_start;
    }
  }
#endif

  // ---- epilogue: O^T accumulator -> out[token][head][d] (bf16), 4 consecutive d per store
  if (qloc < Lq) {
    uint16_t* op = a.out + (int64_t)(s.start + qloc) * a.o_row + (int64_t)h * a.o_head;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        o.x = pack_bf16(acc_o[dt][4 * g4 + 0], acc_o[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_o[dt][4 * g4 + 2], acc_o[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g4 + 4 * hi) = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// Forward with LDS-DMA staging (the default at head dim 256 with contiguous keys; MI355_HSTU_DMA=0 turns it off).  Same GEMMs, masks and register
// layouts as hstu_fwd_kernel; what differs is how a (K, V) tile reaches LDS.  The stamps of the register-staged kernel
//  put 1.2 K cycles per tile into issuing 16 global loads per wave and 0.9 K into writing the
// prefetched registers to LDS (K rows + the transposing V commit), against 2 K cycles of MFMA; `global_load_lds_dwordx4`
// moves 1 KB per wave-instruction from global memory straight into LDS: no staging VGPRs, no commit, and with the tile
// pair double-buffered one barrier per tile.  A DMA instruction writes 64 x 16 B contiguously (two 512-B rows), so rows
// cannot be padded against bank conflicts; instead each lane FETCHES the 16-byte chunk that belongs at its LDS slot under
// an XOR swizzle (free on the load side):
//   K rows (read back with ds_read_b128, 16-lane groups over 16 different rows): chunk' = chunk ^ (row & 15)
//   V rows (read back transposed with ds_read_b64_tr_b16, a half-wave covers 4 rows x 64 B): chunk' = chunk ^ ((row & 3) << 2)
// -- the same bank sets per lane group as the padded layouts (strides of 4 resp. 16 dwords mod 64).
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef __attribute__((address_space(1))) const void* glb_void_t;

// (hstu_fwd_dma_kernel, the one-kind forward with these images -- round 3's default at head dim 256 -- was removed in round 5: the
//  two-waves-per-SIMD kernels below win on every shape, profiles/r04_hstu_fwd_variants_a.txt; the images and swizzles live on in them)

// ---------------------------------------------------------------------------------------------------
// Forward with TWO waves per SIMD (round 4; head dim 256, contiguous keys, no bias): an 8-wave workgroup whose waves are
// specialised by GEMM instead of sharing one instruction stream.
//   waves 0-3 ("S waves"):  hold the Q fragments (64 registers); per 64-key tile S^T = K Q^T (32 MFMAs), alpha / SiLU /
//                           mask / 1/N on the accumulator, pack to bf16 -- the result is already the B operand of GEMM 2
//                           -- and hand it on through LDS: 4 x ds_write_b128 per lane and tile (4 KB per wave).
//   waves 4-7 ("O waves"):  hold the O^T accumulator (128 AGPRs); per tile read that P^T back (4 x ds_read_b128, the same
//                           lane-linear image: conflict free) and run O^T += V^T P^T (32 MFMAs, V^T fragments by
//                           transpose reads).  They also issue the LDS-DMA of the K / V tiles: they have no VALU work.
// Wave w and wave w + 4 own the same 32 query rows and sit on the same SIMD (a workgroup's waves go round the 4 SIMDs), so
// every SIMD alternates a matrix + VALU stream with a matrix + memory stream: the SiLU of one no longer stalls the MFMAs of
// the other, and neither wave needs more than ~200 registers (the one-stream kernel: 222 + 160).  The O wave runs ONE tile
// behind its S wave: iteration i = { S wave: tile i -> P[i & 1] | O wave: P[(i - 1) & 1], V tile i - 1 }, one barrier per
// iteration.  LDS = K ring 2 x 32 KB + V ring 2 x 32 KB + P ring 2 x 4 x 4 KB = 160 KB, all of it.
// The K / V images and swizzles are the ones described above.
// ---------------------------------------------------------------------------------------------------
#ifndef HSTU_PC_SDMA
#define HSTU_PC_SDMA 0   // LDS-DMA instructions per tile and tensor issued by each S wave (of 32; the O waves issue the rest)
#endif
#ifndef HSTU_PC_DSPREAD
#define HSTU_PC_DSPREAD 0   // O waves: 0 = their LDS-DMA share in one burst behind the barrier, n = dealt over the first n MFMA batches of GEMM 2
#endif
#ifndef HSTU_PC_KBUF
#define HSTU_PC_KBUF 3   // S waves: K fragment batches (4 slices) in registers, KBUF - 1 of them in flight ahead of the MFMAs
#endif
#ifndef HSTU_PC_VBUF
#define HSTU_PC_VBUF 3   // O waves: V^T fragment batches likewise
#endif
#ifndef HSTU_PC_PROBE
#define HSTU_PC_PROBE 0  // timing probes, results WRONG on purpose: 1 = SiLU without transcendentals, 2 = no SiLU (pack the raw S), 4 = DMA of
                         // the first two tiles only, 8 = no GEMM 2 MFMAs, 16 = no GEMM 1 MFMAs
#endif
#ifndef HSTU_PC_PRIO
#define HSTU_PC_PRIO 0   // static wave priority: 1 = the O waves (the younger half) at s_setprio 1, 2 = the S waves
#endif
template <int D, bool kWin>
__global__ void __launch_bounds__(512) hstu_fwd_pc_kernel(AttnArgs a) {
  static_assert(D == 256, "rows of 32 chunks");
  constexpr int CPR = D / 8;            // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;         // rows per DMA wave-instruction (1 KB)
  constexpr int ROWB = D;               // row stride in LDS (elements): unpadded
  constexpr int TENS = kBN * ROWB;      // elements of one K (or V) tile
  constexpr int NINS = kBN / RPI;       // DMA instructions per tile and tensor (32)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [K 0 | K 1 | V 0 | V 1 | P 0 | P 1]
  uint16_t* const Kring = smem;
  uint16_t* const Vring = smem + 2 * TENS;
  uint16_t* const Pring = smem + 4 * TENS;   // [2][4 pairs][4 key slices][64 lanes] x 16 B

  const BlockSeq bs = seq_head_of_block(a);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  const int Lq = bs.end - s.start;
  const int kstart = a.cu_seqlens_k ? a.cu_seqlens_k[b] : s.start;
  s.L = a.cu_seqlens_k ? a.cu_seqlens_k[b + 1] - kstart : Lq;
  const int dq = s.L - Lq;
  const int nblk = (Lq + kBM - 1) / kBM;
  if (bs.z >= nblk || dq < 0) return;
  const int m0 = row_block_of_rank(bs.z, nblk, a, b) * kBM;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = kWin ? a.wl : -1; s.wr = kWin ? a.wr : -1;

  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wv >> 2, pw = wv & 3;    // role 0: S wave, 1: O wave; pw: the pair's 32-row group
  const int qrow0 = m0 + 32 * pw, qloc = qrow0 + l31, qi = dq + qloc;
  const bool wave_live = qrow0 < Lq;
  int last_row = dq + (m0 + kBM - 1 < Lq - 1 ? m0 + kBM - 1 : Lq - 1);
  int n_end = s.L;
  if (a.causal) {
    n_end = last_row + 1;
    if (s.has_ctx && dq + m0 < s.c && s.hlen > n_end) n_end = s.hlen;
  }
  if (kWin) n_end = band_key_end(a, last_row, n_end);
  const int n_beg = kWin ? band_key_begin(a, dq + m0, kBN) : 0;
  int w_last = dq + (qrow0 + 31 < Lq - 1 ? qrow0 + 31 : Lq - 1);
  int w_end = s.L;
  if (a.causal) {
    w_end = w_last + 1;
    if (s.has_ctx && dq + qrow0 < s.c && s.hlen > w_end) w_end = s.hlen;
  }
  if (kWin) w_end = band_key_end(a, w_last, w_end);
  const int w_beg = kWin ? band_key_begin(a, dq + qrow0, kBN) : 0;
  const int T = n_end > n_beg ? (n_end - n_beg + kBN - 1) / kBN : 0;   // key tiles of the block

  // ---- the DMA of one tile of one tensor: instruction j moves rows 2 j, 2 j + 1.  S wave pw issues instructions
  // [HSTU_PC_SDMA pw, + HSTU_PC_SDMA), O wave pw the rest dealt evenly.  Lane -> (row inside the instruction, LDS chunk slot p);
  // the lane fetches global chunk p ^ swizzle(row).  Addressing: a wave-uniform 64-bit row base (SGPR arithmetic) plus a
  // per-lane 32-bit offset that depends on the instruction only through j mod 8 -- NMY registers per tensor computed once
  // (64-bit per-lane pointers per instruction cost 2 multiplies and a spilled pointer each, and the reload's vmcnt(0)
  // serialised the DMAs).  Rows past the sequence end (the sequence's last tile only) are read clamped: their P is zero.
  const uint16_t* kg = a.k + (int64_t)kstart * a.k_row + (int64_t)h * a.k_head;
  const uint16_t* vg = a.v + (int64_t)kstart * a.v_row + (int64_t)h * a.v_head;
  const int dma_r = lane / CPR, dma_p = lane % CPR;
  constexpr int NS = HSTU_PC_SDMA, NO = (NINS - 4 * NS) / 4;
  static_assert(4 * NS + 4 * NO == NINS, "the instruction split must cover the tile");
  constexpr int NMY = NS > NO ? NS : NO;
  const int j_first = role == 0 ? NS * pw : 4 * NS + NO * pw;
  const int n_my = role == 0 ? NS : NO;
  uint32_t kvoff[NMY], vvoff[NMY];
#pragma unroll
  for (int u = 0; u < NMY; ++u) {
    const int r = RPI * (j_first + u) + dma_r;
    kvoff[u] = (uint32_t)dma_r * (uint32_t)a.k_row * 2u + 16u * (uint32_t)(dma_p ^ (r & 15));
    vvoff[u] = (uint32_t)dma_r * (uint32_t)a.v_row * 2u + 16u * (uint32_t)(dma_p ^ ((r & 3) << 2));
  }
  // The DMA instruction is issued from inline asm: hipcc models the builtin as an LDS store in flight and puts a
  // vmcnt(0) in front of the next transpose read of ANY LDS address (seen in the removed one-kind DMA forward's GEMM 2: the prefetch
  // it was meant to overlap is drained first).  Here the completion is counted by hand: vmcnt(0) + barrier at the loop head.
  bool dma_on = true;
  auto dma16 = [&](const char* sbase, uint32_t voff, uint32_t lds_byte) {
    if ((HSTU_PC_PROBE & 4) && !dma_on) return;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(sbase) : "memory");
  };
  auto issue_dma = [&](const uint16_t* g, int64_t g_row, const uint32_t (&voff)[NMY], uint16_t* ring, int tile, int u0, int u1) {
    const int n0 = n_beg + kBN * tile;                                // (instructions [u0, u1) of this wave's share)
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(ring + (tile & 1) * TENS + RPI * j_first * ROWB));
    if (n0 + kBN <= s.L) {
      const char* sb = reinterpret_cast<const char*>(g + (int64_t)(n0 + RPI * j_first) * g_row);
      const int64_t step = (int64_t)RPI * g_row * 2;
#pragma unroll
      for (int u = 0; u < NMY; ++u)
        if (u >= u0 && u < u1 && u < n_my) dma16(sb + u * step, voff[u], dst + u * (RPI * ROWB * 2));
    } else {
      const uint32_t rowterm = (uint32_t)dma_r * (uint32_t)g_row * 2u;
#pragma unroll
      for (int u = 0; u < NMY; ++u)
        if (u >= u0 && u < u1 && u < n_my) {
          const int row0 = n0 + RPI * (j_first + u);                   // wave-uniform: the instruction's first row
          const int rowc = row0 < s.L ? row0 : s.L - 1;                // clamped to the sequence
          const uint32_t drop = row0 + 1 < s.L ? 0u : 0xffffffffu;     // its second row is past the end: read the first again
          dma16(reinterpret_cast<const char*>(g + (int64_t)rowc * g_row), voff[u] - (rowterm & drop), dst + u * (RPI * ROWB * 2));
        }
    }
  };

  // ---- S wave state: Q fragments (B operand of GEMM 1) and the row mask
  bf16x8_t qf[D / 16];
  const float nal2e = -a.alpha * 1.44269504088896f, ais = a.alpha * a.inv_scale;
  const RowMask rm = row_mask(qi < s.L ? qi : s.L - 1, s, a.causal, a.group);
  // ---- O wave state
  f32x16_t acc_o[D / 32];

  if (T > 0) issue_dma(kg, a.k_row, kvoff, Kring, 0, 0, NMY);
  if (HSTU_PC_PRIO != 0 && role == (HSTU_PC_PRIO == 1 ? 1 : 0)) __builtin_amdgcn_s_setprio(1);

  // fragment addresses under the swizzles described above
  const int kx = l31 & 15;
  const int il = lane & 15, g1 = (lane >> 4) & 1, vq = il >> 2;
  const int v_row_off = (4 * hi + vq) * ROWB + 4 * (il & 1);
  const int v_chunk_lo = 2 * g1 + ((il & 3) >> 1);
  auto v_frag = [&](const uint16_t* Vb, int dt, int ks) -> bf16x8_t {
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef short v8s_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
    const uint16_t* p0 = Vb + (16 * ks) * ROWB + v_row_off + 8 * ((4 * (dt ^ vq)) + v_chunk_lo);
    const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
    const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * ROWB));
    const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };

  // Two loops, one per role, with the same trip count and ONE barrier per iteration each (a single loop with the role
  // test inside keeps Q and the O accumulator live together: 192 registers before anything else, 363 spilled).
#if HSTU_TIMING
  unsigned tsum[7] = {0, 0, 0, 0, 0, 0, 0};   // wait for own DMA, barrier, DMA issue, role, GEMM 1, SiLU + hand-off | GEMM 2, tiles
  const unsigned t_start = tick();
  auto t_dump = [&]() {
    const unsigned t_end = tick();
    if (lane == 0) {
      const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long* d = g_hstu_dbg + ((size_t)(blk * 8 + wv) % 65536) * 8;
      for (int i = 0; i < 7; ++i) d[i] = tsum[i];
      d[6] |= (unsigned long long)role << 32;
      d[7] = t_end - t_start;
    }
  };
#endif
  auto head = [&](int it) {
    if (HSTU_PC_PROBE & 4) dma_on = it < 1;
    TICK(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (K tile it, V tile it - 1) have landed ...
    TICK(t1);
    __syncthreads();                                    // ... everyone's have, P[it - 1] is written, the other buffers are free
    TICK(t2);
    if (role == 0 || HSTU_PC_DSPREAD == 0) {
      if (it + 1 < T) issue_dma(kg, a.k_row, kvoff, Kring, it + 1, 0, NMY);
      if (it < T) issue_dma(vg, a.v_row, vvoff, Vring, it, 0, NMY);
    }
    TICK(t3);
    TACC(0, t0, t1); TACC(1, t1, t2); TACC(2, t2, t3);
  };
  if (role == 0) {
    // =========================== S wave: tile `it` -> P ring slot it & 1 ===========================
    {
      const uint16_t* qp = a.q + (int64_t)(s.start + (qloc < Lq ? qloc : 0)) * a.q_row + (int64_t)h * a.q_head + 8 * hi;
#pragma unroll
      for (int sl = 0; sl < D / 16; ++sl) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (qloc < Lq) t = *reinterpret_cast<const uint4*>(qp + 16 * sl);
        qf[sl] = *reinterpret_cast<bf16x8_t*>(&t);
      }
    }
    for (int it = 0; it <= T; ++it) {
      head(it);
      const int n0 = n_beg + kBN * it;
      if (it >= T || !wave_live || n0 >= w_end || n0 < w_beg) continue;
      const uint16_t* Ks = Kring + (it & 1) * TENS;
      TICK(t5);
      // mask mode of the tile (wave-uniform): 0 = every key visible, 1 = key <= jmax only (plain causal / sequence end:
      // one compare per element against a per-lane threshold), 2 = the general rule (contextual / target rows, windows)
      const bool full = a.causal && s.wl < 0 && (n0 + kBN - 1 <= dq + qrow0) && (!s.has_ctx || dq + qrow0 >= s.c) && (!s.has_tgt || n0 + kBN - 1 < s.hlen);
      const int mode = full ? 0 : ((!s.has_ctx && !s.has_tgt && s.wl < 0) ? 1 : 2);
      u32x4_t* pdst = reinterpret_cast<u32x4_t*>(Pring) + (((it & 1) * 4 + pw) * 4) * 64 + lane;
      // One tile, software-pipelined inside the wave: GEMM 1 of the second 32-key sub-tile carries the SiLU of the first
      // (one element pair per two MFMAs), so that only the second sub-tile's SiLU runs without MFMAs of this wave.
      auto tile = [&](auto modec) {
        constexpr int kMode = decltype(modec)::value;
        constexpr int SLB = 4, NBAT = (D / 16) / SLB, NKB = HSTU_PC_KBUF;   // fragment batches of 4 slices, NKB - 1 batches in flight
        f32x16_t acc_s[2];
        // (measured and rejected: two accumulator chains per sub-tile -- even / odd slices of d, added in the SiLU -- so that
        // consecutive MFMAs never share an accumulator: 805 vs 841 TFLOP/s at 32 x 4096; the chain is not what paces GEMM 1)
        const int th = rm.jmax - n0 - 4 * hi;       // kMode 1: element (t, rr) is visible iff 32 t + (rr & 3) + 8 (rr >> 2) <= th
        // NE elements r0 .. r0 + NE - 1 of sub-tile t -> NE / 2 packed P words, stage by stage over the group
        auto group = [&](auto nec, int t, int r0, uint32_t* out) {
          constexpr int NE = decltype(nec)::value;
          float x[NE], e[NE], y[NE];
#pragma unroll
          for (int i = 0; i < NE; ++i) x[i] = acc_s[t][r0 + i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = x[i] * nal2e;
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = (HSTU_PC_PROBE & 1) ? e[i] * 0.5f : __builtin_amdgcn_exp2f(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = 1.0f + e[i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = (HSTU_PC_PROBE & 1) ? e[i] * 0.25f : __builtin_amdgcn_rcpf(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) y[i] = (HSTU_PC_PROBE & 2) ? x[i] : x[i] * ais * e[i];
          if (kMode != 0) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
              const int rr = r0 + i, off = 32 * t + (rr & 3) + 8 * (rr >> 2);
              const bool ok = kMode == 1 ? off <= th : key_ok(n0 + off + 4 * hi, rm);
              y[i] = ok ? y[i] : 0.f;
            }
          }
#pragma unroll
          for (int i = 0; i < NE; i += 2) out[i >> 1] = pack_bf16(y[i], y[i + 1]);
        };
        bf16x8_t kfr[NKB][SLB];
        auto load_b = [&](int gb) {               // global batch gb = NBAT t + bi
          const int t = gb / NBAT, bi = gb % NBAT;
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            const int sl = SLB * bi + u;
            const int ch = ((2 * sl) ^ (kx & 14)) + (hi ^ (kx & 1));
            kfr[gb % NKB][u] = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + l31) * ROWB + 8 * ch);
          }
        };
        auto mfma_b = [&](int gb) {
          const int t = gb / NBAT, bi = gb % NBAT;
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            f32x16_t& c = acc_s[t];
            if (HSTU_PC_PROBE & 16) { if (bi == 0) for (int z = 0; z < 16; ++z) c[z] = __builtin_bit_cast(float, __builtin_bit_cast(u32x4_t, kfr[0][0])[z & 3]); }
            else if (bi == 0 && u == 0) mfma_v0(c, kfr[gb % NKB][u], qf[SLB * bi + u]);
            else mfma_v(c, kfr[gb % NKB][u], qf[SLB * bi + u]);
          }
        };
#pragma unroll
        for (int gb = 0; gb < NKB - 1; ++gb) load_b(gb);
        // ---- sub-tile 0: MFMAs only
#pragma unroll
        for (int gb = 0; gb < NBAT; ++gb) {
          if (gb + NKB - 1 < 2 * NBAT) load_b(gb + NKB - 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_b(gb);
          __builtin_amdgcn_sched_barrier(0);
        }
        TICK(t6);
        TACC(4, t5, t6);
        // ---- sub-tile 1: MFMAs + the SiLU of sub-tile 0 (4 elements per batch of 4 MFMAs)
        uint32_t pk[8];
#pragma unroll
        for (int gb = NBAT; gb < 2 * NBAT; ++gb) {
          const int bi = gb - NBAT;
          if (gb + NKB - 1 < 2 * NBAT) load_b(gb + NKB - 1);
          mfma_b(gb);
          group(std::integral_constant<int, 4>{}, 0, 4 * bi, pk + 2 * bi);
          if (bi == 1) pdst[0 * 64] = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
          if (bi == 3) pdst[1 * 64] = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
          __builtin_amdgcn_sched_barrier(0);
        }
        TICK(t6b);
        TACC(3, t6, t6b);
        // ---- the SiLU of sub-tile 1
        group(std::integral_constant<int, 8>{}, 1, 0, pk);
        pdst[2 * 64] = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
        group(std::integral_constant<int, 8>{}, 1, 8, pk + 4);
        pdst[3 * 64] = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
        TICK(t7);
        TACC(5, t6b, t7);
      };
      if (mode == 0) tile(std::integral_constant<int, 0>{});
      else if (mode == 1) tile(std::integral_constant<int, 1>{});
      else tile(std::integral_constant<int, 2>{});
#if HSTU_TIMING
      tsum[6] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA may be in flight into LDS when the block retires)
#if HSTU_TIMING
    t_dump();
#endif
    return;
  }
  {
    // =========================== O wave: tile `it - 1` from P ring slot (it - 1) & 1 ===========================
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
    for (int it = 0; it <= T; ++it) {
      pin_agpr_2w(acc_o);
      head(it);
      pin_agpr_2w(acc_o);
      const int tl = it - 1, n0 = n_beg + kBN * tl;
      const bool live = it != 0 && wave_live && n0 < w_end && n0 >= w_beg;
      // HSTU_PC_DSPREAD: this wave's DMA share (K tile it + 1, V tile it) is dealt over the first MFMA batches of GEMM 2
      // when there is a GEMM 2 and both tiles lie inside the sequence (no clamped rows); otherwise one burst, as at the head.
      const int nk0 = n_beg + kBN * (it + 1), nv0 = n_beg + kBN * it;
      const bool spread = HSTU_PC_DSPREAD != 0 && live && it + 1 < T && nk0 + kBN <= s.L;   // (then V tile `it` is inside as well)
      if (HSTU_PC_DSPREAD != 0 && !spread) {
        if (it + 1 < T) issue_dma(kg, a.k_row, kvoff, Kring, it + 1, 0, NMY);
        if (it < T) issue_dma(vg, a.v_row, vvoff, Vring, it, 0, NMY);
      }
      if (!live) continue;
      const uint16_t* Vt = Vring + (tl & 1) * TENS;
      TICK(t6);
      const u32x4_t* psrc = reinterpret_cast<const u32x4_t*>(Pring) + (((tl & 1) * 4 + pw) * 4) * 64 + lane;
      {
        // (ONE copy of GEMM 2 with the DMA statements behind a wave-uniform test: two specialised copies made hipcc carry
        // the accumulator through a phi and spill it)
        const char* kb = reinterpret_cast<const char*>(kg + (int64_t)(nk0 + RPI * j_first) * a.k_row);
        const char* vb = reinterpret_cast<const char*>(vg + (int64_t)(nv0 + RPI * j_first) * a.v_row);
        const int64_t kstep = (int64_t)RPI * a.k_row * 2, vstep = (int64_t)RPI * a.v_row * 2;
        const uint32_t kdst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(Kring + ((it + 1) & 1) * TENS + RPI * j_first * ROWB));
        const uint32_t vdst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(Vring + (it & 1) * TENS + RPI * j_first * ROWB));
        bf16x8_t pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pf[ks] = __builtin_bit_cast(bf16x8_t, psrc[ks * 64]);
        constexpr int NDT = D / 32, DB = 4, NVB = HSTU_PC_VBUF;
        constexpr int NBAT2 = 4 * (NDT / DB);
        bf16x8_t vfr[NVB][DB];
        auto load_v = [&](int bi) {
          const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
          for (int u = 0; u < DB; ++u) vfr[bi % NVB][u] = v_frag(Vt, dt0 + u, ks);
        };
#pragma unroll
        for (int bi = 0; bi < NVB - 1; ++bi) load_v(bi);
        pin_agpr_2w(acc_o);
#pragma unroll
        for (int bi = 0; bi < NBAT2; ++bi) {
          const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
          if (bi + NVB - 1 < NBAT2) load_v(bi + NVB - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            if (HSTU_PC_PROBE & 8) { if (u == 0 && bi == 0) acc_o[0][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4_t, vfr[bi % NVB][0])[0] ^ __builtin_bit_cast(u32x4_t, pf[ks])[0]); }
            else mfma_a(acc_o[dt0 + u], vfr[bi % NVB][u], pf[ks]);
            if (HSTU_PC_DSPREAD != 0 && bi < HSTU_PC_DSPREAD && (u & 1)) {   // behind every second MFMA of the first batches: PER K + PER V instructions
              constexpr int NSLOT = 2 * (HSTU_PC_DSPREAD ? HSTU_PC_DSPREAD : 1), PER = (NO + NSLOT - 1) / NSLOT;
              const int slot = 2 * bi + (u >> 1);
              if (spread) {
#pragma unroll
                for (int x = PER * slot; x < PER * slot + PER; ++x)
                  if (x < NO) {
                    dma16(kb + x * kstep, kvoff[x], kdst + x * (RPI * ROWB * 2));
                    dma16(vb + x * vstep, vvoff[x], vdst + x * (RPI * ROWB * 2));
                  }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      TICK(t7);
      TACC(5, t6, t7);
#if HSTU_TIMING
      tsum[6] += 1;
#endif
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA may be in flight into LDS when the block retires)
#if HSTU_TIMING
  t_dump();
#endif
  {
    fence_a_2w(acc_o);
    if (qloc < Lq) {
      uint16_t* op = a.out + (int64_t)(s.start + qloc) * a.o_row + (int64_t)h * a.o_head;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 o;
          o.x = pack_bf16(acc_o[dt][4 * g4 + 0], acc_o[dt][4 * g4 + 1]);
          o.y = pack_bf16(acc_o[dt][4 * g4 + 2], acc_o[dt][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g4 + 4 * hi) = o;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The S-wave / O-wave forward over PAIRS of row blocks (round 4; the default at head dim 256): one workgroup takes the z-th
// heaviest and the z-th lightest row block of a (sequence, head) column and runs them as ONE tile stream -- block A's key
// tiles, then block B's -- through the same rings.  Under a causal mask every pair carries the same number of key tiles
// (C3: 8 + 2 and 6 + 4 of a column's four blocks; 32 x 4096: 33 each), so the grid is balanced by construction and, at C3,
// exactly one workgroup per CU; the second block's first K tile is prefetched under the first block's last tile, the O waves
// store block A's rows while the S waves already work on block B's first tile, and the pipeline fills and drains once per
// pair instead of once per block.  Same tiles, same MFMA order, same roundings as hstu_fwd_pc_kernel: bit-identical output.
// ---------------------------------------------------------------------------------------------------
template <int D, bool kWin>
__global__ void __launch_bounds__(512) hstu_fwd_pair_kernel(AttnArgs a) {
  static_assert(D == 256, "rows of 32 chunks");
  constexpr int CPR = D / 8, RPI = 64 / CPR, ROWB = D, TENS = kBN * ROWB, NINS = kBN / RPI;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [K 0 | K 1 | V 0 | V 1 | P 0 | P 1]
  uint16_t* const Kring = smem;
  uint16_t* const Vring = smem + 2 * TENS;
  uint16_t* const Pring = smem + 4 * TENS;

  const BlockSeq bs = seq_head_of_block(a);   // grid (H, B, ceil(blocks / 2)): z = the pair's rank inside its column
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  const int Lq = bs.end - s.start;
  const int kstart = a.cu_seqlens_k ? a.cu_seqlens_k[b] : s.start;
  s.L = a.cu_seqlens_k ? a.cu_seqlens_k[b + 1] - kstart : Lq;
  const int dq = s.L - Lq;
  const int nblk = (Lq + kBM - 1) / kBM;
  if (2 * bs.z >= nblk || dq < 0) return;
  const int rank0 = bs.z, rank1 = nblk - 1 - bs.z;
  const bool two = rank1 > rank0;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = kWin ? a.wl : -1; s.wr = kWin ? a.wr : -1;

  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wv >> 2, pw = wv & 3;
  // ---- one row block of the pair
  struct Blk { int m0, qrow0, qloc, n_beg, w_beg, w_end, T; bool wave_live; RowMask rm; };
  auto setup = [&](int rank) -> Blk {
    Blk k;
    k.m0 = row_block_of_rank(rank, nblk, a, b) * kBM;
    k.qrow0 = k.m0 + 32 * pw;
    k.qloc = k.qrow0 + l31;
    k.wave_live = k.qrow0 < Lq;
    const int last_row = dq + (k.m0 + kBM - 1 < Lq - 1 ? k.m0 + kBM - 1 : Lq - 1);
    int n_end = s.L;
    if (a.causal) {
      n_end = last_row + 1;
      if (s.has_ctx && dq + k.m0 < s.c && s.hlen > n_end) n_end = s.hlen;
    }
    if (kWin) n_end = band_key_end(a, last_row, n_end);
    k.n_beg = kWin ? band_key_begin(a, dq + k.m0, kBN) : 0;
    const int w_last = dq + (k.qrow0 + 31 < Lq - 1 ? k.qrow0 + 31 : Lq - 1);
    k.w_end = s.L;
    if (a.causal) {
      k.w_end = w_last + 1;
      if (s.has_ctx && dq + k.qrow0 < s.c && s.hlen > k.w_end) k.w_end = s.hlen;
    }
    if (kWin) k.w_end = band_key_end(a, w_last, k.w_end);
    k.w_beg = kWin ? band_key_begin(a, dq + k.qrow0, kBN) : 0;
    k.T = n_end > k.n_beg ? (n_end - k.n_beg + kBN - 1) / kBN : 0;
    const int qi = dq + k.qloc;
    k.rm = row_mask(qi < s.L ? qi : s.L - 1, s, a.causal, a.group);
    return k;
  };
  const Blk B0 = setup(rank0);
  Blk B1 = B0;
  if (two) B1 = setup(rank1);
  const int T0 = B0.T, T1 = two ? B1.T : 0, N = T0 + T1;     // items of the tile stream: block A's tiles, then block B's
  auto item_n0 = [&](int i) { return i < T0 ? B0.n_beg + kBN * i : B1.n_beg + kBN * (i - T0); };

  // ---- LDS-DMA (as hstu_fwd_pc_kernel; the O waves issue everything)
  const uint16_t* kg = a.k + (int64_t)kstart * a.k_row + (int64_t)h * a.k_head;
  const uint16_t* vg = a.v + (int64_t)kstart * a.v_row + (int64_t)h * a.v_head;
  const int dma_r = lane / CPR, dma_p = lane % CPR;
  constexpr int NO = NINS / 4;
  const int j_first = NO * pw;
  uint32_t kvoff[NO], vvoff[NO];
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int r = RPI * (j_first + u) + dma_r;
    kvoff[u] = (uint32_t)dma_r * (uint32_t)a.k_row * 2u + 16u * (uint32_t)(dma_p ^ (r & 15));
    vvoff[u] = (uint32_t)dma_r * (uint32_t)a.v_row * 2u + 16u * (uint32_t)(dma_p ^ ((r & 3) << 2));
  }
  auto dma16 = [&](const char* sbase, uint32_t voff, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(sbase) : "memory");
  };
  auto issue_dma = [&](const uint16_t* g, int64_t g_row, const uint32_t (&voff)[NO], uint16_t* ring, int item) {
    const int n0 = item_n0(item);
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(ring + (item & 1) * TENS + RPI * j_first * ROWB));
    if (n0 + kBN <= s.L) {
      const char* sb = reinterpret_cast<const char*>(g + (int64_t)(n0 + RPI * j_first) * g_row);
      const int64_t step = (int64_t)RPI * g_row * 2;
#pragma unroll
      for (int u = 0; u < NO; ++u) dma16(sb + u * step, voff[u], dst + u * (RPI * ROWB * 2));
    } else {
      const uint32_t rowterm = (uint32_t)dma_r * (uint32_t)g_row * 2u;
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        const int row0 = n0 + RPI * (j_first + u);
        const int rowc = row0 < s.L ? row0 : s.L - 1;
        const uint32_t drop = row0 + 1 < s.L ? 0u : 0xffffffffu;
        dma16(reinterpret_cast<const char*>(g + (int64_t)rowc * g_row), voff[u] - (rowterm & drop), dst + u * (RPI * ROWB * 2));
      }
    }
  };
  const float nal2e = -a.alpha * 1.44269504088896f, ais = a.alpha * a.inv_scale;
  const int kx = l31 & 15;
  const int il = lane & 15, g1 = (lane >> 4) & 1, vq = il >> 2;
  const int v_row_off = (4 * hi + vq) * ROWB + 4 * (il & 1);
  const int v_chunk_lo = 2 * g1 + ((il & 3) >> 1);
  auto v_frag = [&](const uint16_t* Vb, int dt, int ks) -> bf16x8_t {
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef short v8s_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
    const uint16_t* p0 = Vb + (16 * ks) * ROWB + v_row_off + 8 * ((4 * (dt ^ vq)) + v_chunk_lo);
    const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
    const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * ROWB));
    const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };

  if (role == 0) {
    // =========================== S waves: item `it` -> P ring slot it & 1 ===========================
    bf16x8_t qf[D / 16];
    auto load_q = [&](const Blk& k) {
      const uint16_t* qp = a.q + (int64_t)(s.start + (k.qloc < Lq ? k.qloc : 0)) * a.q_row + (int64_t)h * a.q_head + 8 * hi;
#pragma unroll
      for (int sl = 0; sl < D / 16; ++sl) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (k.qloc < Lq) t = *reinterpret_cast<const uint4*>(qp + 16 * sl);
        qf[sl] = *reinterpret_cast<bf16x8_t*>(&t);
      }
    };
    // Two loops over ONE stream of iterations (block A's items, then block B's + the draining iteration): with a single loop
    // and the block switch inside it hipcc carried Q / the O accumulator through phis and spilled them.
    auto s_iter = [&](int it) {
      __syncthreads();                     // (no vmcnt wait here: the S waves issue no DMA)
      if (it >= N) return;
      const bool second = it >= T0;
      const int t_in = second ? it - T0 : it;
      const int n_beg = second ? B1.n_beg : B0.n_beg, w_end = second ? B1.w_end : B0.w_end, w_beg = second ? B1.w_beg : B0.w_beg;
      const int qrow0 = second ? B1.qrow0 : B0.qrow0;
      const bool wave_live = second ? B1.wave_live : B0.wave_live;
      const int n0 = n_beg + kBN * t_in;
      if (!wave_live || n0 >= w_end || n0 < w_beg) return;
      RowMask rm;
      rm.jmax = second ? B1.rm.jmax : B0.rm.jmax; rm.jlo = second ? B1.rm.jlo : B0.rm.jlo; rm.hlen = second ? B1.rm.hlen : B0.rm.hlen;
      const uint16_t* Ks = Kring + (it & 1) * TENS;
      const bool full = a.causal && s.wl < 0 && (n0 + kBN - 1 <= dq + qrow0) && (!s.has_ctx || dq + qrow0 >= s.c) && (!s.has_tgt || n0 + kBN - 1 < s.hlen);
      const int mode = full ? 0 : ((!s.has_ctx && !s.has_tgt && s.wl < 0) ? 1 : 2);
      u32x4_t* pdst = reinterpret_cast<u32x4_t*>(Pring) + (((it & 1) * 4 + pw) * 4) * 64 + lane;
      auto tile = [&](auto modec) {
        constexpr int kMode = decltype(modec)::value;
        constexpr int SLB = 4, NBAT = (D / 16) / SLB, NKB = HSTU_PC_KBUF;
        f32x16_t acc_s[2];
        const int th = rm.jmax - n0 - 4 * hi;
        auto group = [&](auto nec, int t, int r0, uint32_t* out) {
          constexpr int NE = decltype(nec)::value;
          float x[NE], e[NE], y[NE];
#pragma unroll
          for (int i = 0; i < NE; ++i) x[i] = acc_s[t][r0 + i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = x[i] * nal2e;
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = 1.0f + e[i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) y[i] = x[i] * ais * e[i];
          if (kMode != 0) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
              const int rr = r0 + i, off = 32 * t + (rr & 3) + 8 * (rr >> 2);
              const bool ok = kMode == 1 ? off <= th : key_ok(n0 + off + 4 * hi, rm);
              y[i] = ok ? y[i] : 0.f;
            }
          }
#pragma unroll
          for (int i = 0; i < NE; i += 2) out[i >> 1] = pack_bf16(y[i], y[i + 1]);
        };
        bf16x8_t kfr[NKB][SLB];
        auto load_b = [&](int gb) {
          const int t = gb / NBAT, bi = gb % NBAT;
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            const int sl = SLB * bi + u;
            const int ch = ((2 * sl) ^ (kx & 14)) + (hi ^ (kx & 1));
            kfr[gb % NKB][u] = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + l31) * ROWB + 8 * ch);
          }
        };
        auto mfma_b = [&](int gb) {
          const int t = gb / NBAT, bi = gb % NBAT;
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            if (bi == 0 && u == 0) mfma_v0(acc_s[t], kfr[gb % NKB][u], qf[SLB * bi + u]);
            else mfma_v(acc_s[t], kfr[gb % NKB][u], qf[SLB * bi + u]);
          }
        };
#pragma unroll
        for (int gb = 0; gb < NKB - 1; ++gb) load_b(gb);
#pragma unroll
        for (int gb = 0; gb < NBAT; ++gb) {
          if (gb + NKB - 1 < 2 * NBAT) load_b(gb + NKB - 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_b(gb);
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t pk[8];
#pragma unroll
        for (int gb = NBAT; gb < 2 * NBAT; ++gb) {
          const int bi = gb - NBAT;
          if (gb + NKB - 1 < 2 * NBAT) load_b(gb + NKB - 1);
          mfma_b(gb);
          group(std::integral_constant<int, 4>{}, 0, 4 * bi, pk + 2 * bi);
          if (bi == 1) pdst[0 * 64] = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
          if (bi == 3) pdst[1 * 64] = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
          __builtin_amdgcn_sched_barrier(0);
        }
        group(std::integral_constant<int, 8>{}, 1, 0, pk);
        pdst[2 * 64] = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
        group(std::integral_constant<int, 8>{}, 1, 8, pk + 4);
        pdst[3 * 64] = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
      };
      if (mode == 0) tile(std::integral_constant<int, 0>{});
      else if (mode == 1) tile(std::integral_constant<int, 1>{});
      else tile(std::integral_constant<int, 2>{});
    };
    // (block B's queries are loaded at the switch, while the O waves finish block A's last tile and store its rows.  Fetched
    // at kernel entry next to block A's -- hipcc parks them in scratch -- the prologue burst of all CUs doubles: C3 39.3 ->
    // 44.2 us, 32 x 1024 101.7 -> 115.3 us.)
    load_q(B0);
    for (int it = 0; it < T0; ++it) s_iter(it);
    if (two) load_q(B1);
    for (int it = T0; it <= N; ++it) s_iter(it);
    return;
  }
  // =========================== O waves: item `it - 1`; DMA of K item it + 1 and V item it ===========================
  f32x16_t acc_o[D / 32];
  auto zero_acc = [&]() {
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
  };
  auto store_rows = [&](int qloc) {
    fence_a_2w(acc_o);
    if (qloc < Lq) {
      uint16_t* op = a.out + (int64_t)(s.start + qloc) * a.o_row + (int64_t)h * a.o_head;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 o;
          o.x = pack_bf16(acc_o[dt][4 * g4 + 0], acc_o[dt][4 * g4 + 1]);
          o.y = pack_bf16(acc_o[dt][4 * g4 + 2], acc_o[dt][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g4 + 4 * hi) = o;
        }
    }
  };
  zero_acc();
  if (N > 0) issue_dma(kg, a.k_row, kvoff, Kring, 0);
  auto o_iter = [&](int it) {
    pin_agpr_2w(acc_o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (K item it, V item it - 1) have landed ...
    __syncthreads();                                    // ... everyone's have, P[it - 1] is written, the other buffers are free
    if (it + 1 < N) issue_dma(kg, a.k_row, kvoff, Kring, it + 1);
    if (it < N) issue_dma(vg, a.v_row, vvoff, Vring, it);
    pin_agpr_2w(acc_o);
    const int tl = it - 1;
    if (tl >= 0) {
      const bool second = tl >= T0;
      const int t_in = second ? tl - T0 : tl;
      const int n0 = (second ? B1.n_beg : B0.n_beg) + kBN * t_in;
      const bool live = (second ? B1.wave_live : B0.wave_live) && n0 < (second ? B1.w_end : B0.w_end) && n0 >= (second ? B1.w_beg : B0.w_beg);
      if (live) {
        const uint16_t* Vt = Vring + (tl & 1) * TENS;
        const u32x4_t* psrc = reinterpret_cast<const u32x4_t*>(Pring) + (((tl & 1) * 4 + pw) * 4) * 64 + lane;
        bf16x8_t pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pf[ks] = __builtin_bit_cast(bf16x8_t, psrc[ks * 64]);
        constexpr int NDT = D / 32, DB = 4, NVB = HSTU_PC_VBUF;
        constexpr int NBAT2 = 4 * (NDT / DB);
        bf16x8_t vfr[NVB][DB];
        auto load_v = [&](int bi) {
          const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
          for (int u = 0; u < DB; ++u) vfr[bi % NVB][u] = v_frag(Vt, dt0 + u, ks);
        };
#pragma unroll
        for (int bi = 0; bi < NVB - 1; ++bi) load_v(bi);
        pin_agpr_2w(acc_o);
#pragma unroll
        for (int bi = 0; bi < NBAT2; ++bi) {
          const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
          if (bi + NVB - 1 < NBAT2) load_v(bi + NVB - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < DB; ++u) mfma_a(acc_o[dt0 + u], vfr[bi % NVB][u], pf[ks]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  // iterations 0 .. T0 finish block A (its last tile is item T0 - 1, computed in iteration T0); T0 + 1 .. N are block B's
  for (int it = 0; it <= (two ? T0 : N); ++it) o_iter(it);
  if (two) {
    store_rows(B0.qloc);     // block A's rows leave while the S waves already work on block B's first tile
    zero_acc();
    for (int it = T0 + 1; it <= N; ++it) o_iter(it);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA may be in flight into LDS when the block retires)
  store_rows(two ? B1.qloc : B0.qloc);
}

// ---------------------------------------------------------------------------------------------------
// The S-wave / O-wave forward with TWO MFMAs per LDS fragment (round 4; the default at head dim 256).
// tools/ubench_mfma.hip: the matrix pipe issues a 32x32x16 MFMA every 32 clocks whatever the partner wave does -- but in
// hstu_fwd_pc_kernel / hstu_fwd_pair_kernel every MFMA consumes a fresh 1 KB fragment from LDS (each wave owns 32 query rows),
// 4 SIMDs x 1 KB / 32 clk = the CU's whole 128 B/clk of LDS bandwidth before the DMA writes and the P hand-off: 352 KB per
// 64-key tile, ~4 000 clocks per tile measured = 88 B/clk.  They are LDS-bound; that is the "45-52 clocks per MFMA" of their stamps.
// Here a wave owns 64 query rows, so that every K and every V^T fragment read feeds two MFMAs:
//   S wave (half, sub):  query rows 64 half .. + 63 (Q fragments: 128 registers), keys 32 sub .. + 31 of the tile:
//                        16 K fragment reads (16 KB), 32 MFMAs, SiLU of 32 keys x 64 rows, P^T to the P ring.
//   O wave (half, sub):  the same 64 rows, output columns 128 sub .. + 127 (accumulator: 128 registers): reads the half's P^T
//                        (8 KB) and 16 V^T fragments (16 KB), 32 MFMAs; issues the tile's LDS-DMA as before.
// LDS traffic per tile: K 64 + V 64 + P 16 written / 32 read + DMA 64 = 240 KB (352 before).  Rings, swizzles, the one barrier
// per tile, the lag of one tile between the roles and the optional (heavy, light) pairing of row blocks (kPair) are those of
// the two kernels above; same MFMA order per output element, same roundings: bit-identical output.
// Measured and rejected: the S waves software-pipelined across tiles (GEMM 1 of tile t between the SiLU groups of tile t - 1, the
// O waves two tiles behind): Q (128) + two accumulator sets (64) + fragments do not fit 256 registers -- hipcc spills 45 of them and
// reloads Q fragments from scratch inside the loop: 708-745 TFLOP/s against 866-948 at L = 4096 (same bits).
// ---------------------------------------------------------------------------------------------------
#ifndef HSTU_Q2_KBUF
#define HSTU_Q2_KBUF 3   // S waves: K fragment batches (2 slices = 4 MFMAs) in registers, KBUF - 1 in flight
#endif
#ifndef HSTU_Q2_PK
#define HSTU_Q2_PK 0   // S waves: SiLU's plain VALU work as packed fp32 pairs
#endif
#ifndef HSTU_Q2_VBUF
#define HSTU_Q2_VBUF 3   // O waves: V^T fragment batches (2 fragments = 4 MFMAs) likewise
#endif
// kFunc (round 6): `func` masks of up to two bands (n_func <= 5) -- the rows' bounds live in registers of the S waves, the tile stream is
// clipped to what the block's functions reach, a half skips the tiles in the gap between its prefix and its bands, and the tiles
// below every row's prefix take the mask-free path
template <int D, bool kWin, bool kPair, bool kFunc = false>
__global__ void __launch_bounds__(512) hstu_fwd_q2_kernel(AttnArgs a) {
  static_assert(D == 256, "rows of 32 chunks");
  constexpr int CPR = D / 8, RPI = 64 / CPR, ROWB = D, TENS = kBN * ROWB, NINS = kBN / RPI;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [K 0 | K 1 | V 0 | V 1 | P 0 | P 1]
  uint16_t* const Kring = smem;
  uint16_t* const Vring = smem + 2 * TENS;
  u32x4_t* const Pring = reinterpret_cast<u32x4_t*>(smem + 4 * TENS);   // [2 slots][2 halves][2 q tiles][4 key slices][64 lanes] x 16 B

  const BlockSeq bs = seq_head_of_block(a);   // grid (H, B, blocks) or, kPair, (H, B, ceil(blocks / 2))
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  const int Lq = bs.end - s.start;
  const int kstart = a.cu_seqlens_k ? a.cu_seqlens_k[b] : s.start;
  s.L = a.cu_seqlens_k ? a.cu_seqlens_k[b + 1] - kstart : Lq;
  const int dq = s.L - Lq;
  const int nblk = (Lq + kBM - 1) / kBM;
  if ((kPair ? 2 * bs.z : bs.z) >= nblk || dq < 0) return;
  const int rank0 = bs.z, rank1 = kPair ? nblk - 1 - bs.z : bs.z;
  const bool two = kPair && rank1 > rank0;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = kWin ? a.wl : -1; s.wr = kWin ? a.wr : -1;

  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wv >> 2, pw = wv & 3, half = pw >> 1, sub = pw & 1;
  // ---- one row block of the stream, seen from this wave's 64-row half
  // t_lo .. t_hi: the tiles that reach the half's rows; S waves: tiles below tf need no mask (a prefix of the stream under every
  // mask rule), tile t_dead has this wave's 32 keys past every row's reach (its P is zero, the O waves read the whole tile)
  // g_lo .. g_hi (kFunc): tiles in the half's gap; jt / js (kFunc): the BLOCK's gap -- from tile jt on the stream continues js keys
  // further right (the tiles wholly between the block's prefix and its bands are not part of the stream: no DMA, no barrier)
  struct Blk { int m0, hrow0, n_beg, T, t_lo, t_hi, tf, t_dead, g_lo, g_hi, jt, js; };
  auto tile_n0 = [&](const Blk& k, int t) { int n = k.n_beg + kBN * t; if constexpr (kFunc) n += t >= k.jt ? k.js : 0; return n; };
  auto setup = [&](int rank) -> Blk {
    Blk k;
    k.m0 = row_block_of_rank(rank, nblk, a, b) * kBM;
    k.hrow0 = k.m0 + 64 * half;
    bool half_live = k.hrow0 < Lq;
    k.g_lo = k.g_hi = 0; k.jt = 0x7fffffff; k.js = 0;
    const int last_row = dq + (k.m0 + kBM - 1 < Lq - 1 ? k.m0 + kBM - 1 : Lq - 1);
    int n_end = s.L;
    if (a.causal) {
      n_end = last_row + 1;
      if (s.has_ctx && dq + k.m0 < s.c && s.hlen > n_end) n_end = s.hlen;
    }
    if (kWin) n_end = band_key_end(a, last_row, n_end);
    k.n_beg = kWin ? band_key_begin(a, dq + k.m0, kBN) : 0;
    const int h_last = dq + (k.hrow0 + 63 < Lq - 1 ? k.hrow0 + 63 : Lq - 1);
    int h_end = s.L;
    if (a.causal) {
      h_end = h_last + 1;
      if (s.has_ctx && dq + k.hrow0 < s.c && s.hlen > h_end) h_end = s.hlen;
    }
    if (kWin) h_end = band_key_end(a, h_last, h_end);
    int h_beg = kWin ? band_key_begin(a, dq + k.hrow0, kBN) : 0;     // (a multiple of the tile, >= n_beg)
    FuncExt hx{0x7fffffff, 0, 0x7fffffff, 0};
    bool gap_half = false;
    if constexpr (kFunc) {
      if (a.wskip) {     // extents of the half's 64 rows and of the block's 128: one row per lane, no barrier
        auto ext64 = [&](int row0) {
          const int r = row0 + lane;
          return func_ext_wave64(func_ext_row(a, h, (int64_t)s.start + r, r < Lq, (s.has_ctx && dq + r < s.c) ? s.hlen : 0));
        };
        hx = ext64(k.hrow0);
        const FuncExt bx = func_ext_merge(hx, ext64(k.m0 + 64 * (1 - half)));
        const int be = func_ext_end(bx), bb = func_ext_begin(bx), he = func_ext_end(hx), hb = func_ext_begin(hx);
        if (be < n_end) n_end = be;
        if (he < h_end) h_end = he;
        if (bb > k.n_beg) k.n_beg = bb >= n_end ? n_end : (bb / kBN) * kBN;
        if (hb >= h_end) half_live = false;
        else if (hb > h_beg) h_beg = (hb / kBN) * kBN;
        if (h_beg < k.n_beg) h_beg = k.n_beg;
        if (bx.lo < bx.hi && bx.lo > bx.f0) {     // the block's gap: tiles wholly inside [bx.f0, min(bx.lo, n_end)) leave the stream
          const int lo_c = bx.lo < n_end ? bx.lo : n_end;
          const int gs = bx.f0 > k.n_beg ? (bx.f0 - k.n_beg + kBN - 1) / kBN : 0, ge = lo_c > k.n_beg ? (lo_c - k.n_beg) / kBN : 0;
          if (gs < ge) { k.jt = gs; k.js = (ge - gs) * kBN; }
        }
        gap_half = half_live && hx.lo < hx.hi && hx.lo > hx.f0;
      }
    }
    if constexpr (!kFunc) {
      k.T = n_end > k.n_beg ? (n_end - k.n_beg + kBN - 1) / kBN : 0;
      k.t_lo = (h_beg - k.n_beg) / kBN;
      k.t_hi = h_end > k.n_beg ? (h_end - k.n_beg + kBN - 1) / kBN : 0;
    } else {
      // key position -> tile index of the stream (a key inside the block's gap: the first tile behind it)
      const int GA = k.n_beg + kBN * (k.jt < 0x7fffffff ? k.jt : 0), GB = GA + k.js;
      auto t_floor = [&](int key) { if (k.js > 0 && key > GA) { if (key < GB) return k.jt; key -= k.js; } return (key - k.n_beg) / kBN; };
      auto t_ceil = [&](int key) { if (k.js > 0 && key > GA) { if (key < GB) return k.jt; key -= k.js; } return (key - k.n_beg + kBN - 1) / kBN; };
      k.T = n_end > k.n_beg ? t_ceil(n_end) : 0;
      k.t_lo = t_floor(h_beg);
      k.t_hi = h_end > k.n_beg ? t_ceil(h_end) : 0;
      if (gap_half) {     // the tiles wholly between the HALF's prefix and its bands' hull that are still in the stream
        const int g0 = hx.f0 > k.n_beg ? t_ceil(hx.f0) : 0, g1 = hx.lo > k.n_beg ? t_floor(hx.lo) : 0;
        if (g0 < g1) { k.g_lo = g0; k.g_hi = g1; }
      }
    }
    if (k.t_hi > k.T) k.t_hi = k.T;
    if (!half_live || k.t_hi < k.t_lo) k.t_hi = k.t_lo;
    // mode 0 (no mask): causal -- no window, the half holds no contextual row, the wave's last key <= the half's first row and
    // inside the history; otherwise -- no mask rule at all and the wave's last key inside the sequence
    const bool fc = a.causal ? (s.wl < 0 && (!s.has_ctx || dq + k.hrow0 >= s.c)) : (!s.has_ctx && !s.has_tgt && s.wl < 0 && s.wr < 0);
    int lim = a.causal ? dq + k.hrow0 : s.L - 1;
    if (a.causal && s.has_tgt && s.hlen - 1 < lim) lim = s.hlen - 1;
    const int num = lim - 31 - 32 * sub - k.n_beg;
    k.tf = (fc && num >= 0) ? num / kBN + 1 : 0;
    if constexpr (kFunc) {     // ... and below every row's prefix
      const int num2 = hx.f0min - 32 - 32 * sub - k.n_beg;
      const int tf2 = (a.wskip && half_live && num2 >= 0) ? num2 / kBN + 1 : 0;
      if (tf2 < k.tf) k.tf = tf2;
    }
    if constexpr (kFunc) { if (k.tf > k.jt) k.tf = k.jt; }
    if constexpr (kFunc) k.t_dead = (sub == 1 && k.t_hi > k.t_lo && tile_n0(k, k.t_hi - 1) + 32 >= h_end) ? k.t_hi - 1 : -1;
    else k.t_dead = (sub == 1 && k.t_hi > k.t_lo && k.n_beg + kBN * (k.t_hi - 1) + 32 >= h_end) ? k.t_hi - 1 : -1;
    return k;
  };
  const Blk B0 = setup(rank0);
  Blk B1 = B0;
  if (two) B1 = setup(rank1);
  const int T0 = B0.T, T1 = two ? B1.T : 0, N = T0 + T1;     // items of the tile stream: block A's tiles, then block B's
  auto item_n0 = [&](int i) {
    if constexpr (kFunc) return i < T0 ? tile_n0(B0, i) : tile_n0(B1, i - T0);
    else return i < T0 ? B0.n_beg + kBN * i : B1.n_beg + kBN * (i - T0);
  };
#if HSTU_TIMING
  unsigned tsum[7] = {0, 0, 0, 0, 0, 0, 0};   // O: wait for own DMA | barrier | O: DMA issue | - | S: GEMM 1 | S: SiLU + hand-off, O: GEMM 2 | tiles
  const unsigned t_start = tick();
  auto t_dump = [&]() {
    const unsigned t_end = tick();
    if (lane == 0) {
      const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long* d = g_hstu_dbg + ((size_t)(blk * 8 + wv) % 65536) * 8;
      for (int i = 0; i < 7; ++i) d[i] = tsum[i];
      d[6] |= (unsigned long long)role << 32;
      d[7] = t_end - t_start;
    }
  };
#endif

  if (role == 0) {
    // =========================== S waves: item `it` -> P ring slot it & 1 ===========================
    bf16x8_t qf[2][D / 16];
    RowMask rm[2];
    int f_pre[2], f_lo[2][2], f_len[2][2];     // kFunc: the row sees key j iff j < f_pre or (unsigned)(j - f_lo[b]) < f_len[b]
    auto load_q = [&](const Blk& k) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qloc = k.hrow0 + 32 * qt + l31;
        if constexpr (kFunc) {
          const int32_t* ft = a.func + (int64_t)h * a.func_h + s.start + (qloc < Lq ? qloc : 0);
          const int free_below = (s.has_ctx && dq + qloc < s.c) ? s.hlen : 0;
          const int f0 = qloc < Lq ? ft[0] : 0;
          f_pre[qt] = f0 > free_below ? f0 : free_below;
#pragma unroll
          for (int bnd = 0; bnd < 2; ++bnd) {
            int lo = 0, up = 0;
            if (qloc < Lq && 2 * bnd + 2 < a.n_func) { lo = ft[(int64_t)(2 * bnd + 1) * a.func_p]; up = ft[(int64_t)(2 * bnd + 2) * a.func_p]; }
            lo = lo > 0 ? lo : 0;
            f_lo[qt][bnd] = lo;
            f_len[qt][bnd] = up > lo ? up - lo : 0;
          }
        }
        const uint16_t* qp = a.q + (int64_t)(s.start + (qloc < Lq ? qloc : 0)) * a.q_row + (int64_t)h * a.q_head + 8 * hi;
#pragma unroll
        for (int sl = 0; sl < D / 16; ++sl) {
          uint4 t = make_uint4(0, 0, 0, 0);
          if (qloc < Lq) t = *reinterpret_cast<const uint4*>(qp + 16 * sl);
          qf[qt][sl] = *reinterpret_cast<bf16x8_t*>(&t);
        }
        const int qi = dq + qloc;
        rm[qt] = row_mask(qi < s.L ? qi : s.L - 1, s, a.causal, a.group);
      }
    };
    const float nal2e = -a.alpha * 1.44269504088896f, ais = a.alpha * a.inv_scale;
    const int kx = l31 & 15;
#if HSTU_TIMING
    unsigned t_prev = 0;
#endif
    const int m12 = (!s.has_ctx && !s.has_tgt && s.wl < 0) ? 1 : 2;   // the masked tiles' rule: key <= jmax only, or the general one
    auto s_iter = [&](int it, const Blk& k, int t_in) {    // t_in < 0: the draining iteration (barrier only)
      TICK(t0);
#if HSTU_TIMING
      if (t_prev) TACC(3, t_prev, t0);
      t_prev = 0;
#endif
      __syncthreads();                     // (no vmcnt wait here: the S waves issue no DMA)
      TICK(t1);
      TACC(1, t0, t1);
      if (t_in < k.t_lo || t_in >= k.t_hi) return;
      if constexpr (kFunc) { if (t_in >= k.g_lo && t_in < k.g_hi) return; }
      u32x4_t* pdst = Pring + ((((it & 1) * 2 + half) * 2) * 4 + 2 * sub) * 64 + lane;   // q tile qt: + 256 qt, second key slice: + 64
      if (t_in == k.t_dead) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        pdst[0] = z; pdst[64] = z; pdst[256] = z; pdst[320] = z;
        return;
      }
      int k0 = k.n_beg + kBN * t_in + 32 * sub;        // this wave's 32 keys
      if constexpr (kFunc) k0 = tile_n0(k, t_in) + 32 * sub;
      const uint16_t* Ks = Kring + (it & 1) * TENS + (32 * sub + l31) * ROWB;
      auto tile = [&](auto modec) {
        constexpr int kMode = decltype(modec)::value;
        constexpr int SLB = 2, NBAT = (D / 16) / SLB, NKB = HSTU_Q2_KBUF;
        TICK(t5);
        TACC(0, t1, t5);
        f32x16_t acc_s[2];
        bf16x8_t kfr[NKB][SLB];
        auto load_b = [&](int bi) {
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            const int sl = SLB * bi + u;
            const int ch = ((2 * sl) ^ (kx & 14)) + (hi ^ (kx & 1));
            kfr[bi % NKB][u] = *reinterpret_cast<const bf16x8_t*>(Ks + 8 * ch);
          }
        };
#pragma unroll
        for (int bi = 0; bi < NKB - 1; ++bi) load_b(bi);
#pragma unroll
        for (int bi = 0; bi < NBAT; ++bi) {
          if (bi + NKB - 1 < NBAT) load_b(bi + NKB - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < SLB; ++u)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              if (bi == 0 && u == 0) mfma_v0(acc_s[qt], kfr[bi % NKB][u], qf[qt][SLB * bi + u]);
              else mfma_v(acc_s[qt], kfr[bi % NKB][u], qf[qt][SLB * bi + u]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        TICK(t6);
        TACC(4, t5, t6);
        // SiLU, mask, 1 / N, pack: elements r0 .. r0 + 7 of q tile qt -> one key slice (4 packed words), stage by stage
        auto group = [&](int qt, int r0, uint32_t* out) {
          constexpr int NE = 8;
          float y[NE];
#if HSTU_Q2_PK
          {   // the plain multiplies and the add as packed fp32 pairs (v_pk_mul_f32 / v_pk_add_f32): same operations, same roundings
            f32x2_t x2[NE / 2], e2[NE / 2];
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) x2[i] = f32x2_t{acc_s[qt][r0 + 2 * i], acc_s[qt][r0 + 2 * i + 1]};
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) e2[i] = x2[i] * nal2e;
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) e2[i] = f32x2_t{__builtin_amdgcn_exp2f(e2[i].x), __builtin_amdgcn_exp2f(e2[i].y)};
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) e2[i] = e2[i] + 1.0f;
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) e2[i] = f32x2_t{__builtin_amdgcn_rcpf(e2[i].x), __builtin_amdgcn_rcpf(e2[i].y)};
#pragma unroll
            for (int i = 0; i < NE / 2; ++i) { const f32x2_t t = x2[i] * ais * e2[i]; y[2 * i] = t.x; y[2 * i + 1] = t.y; }
          }
#else
          float x[NE], e[NE];
#pragma unroll
          for (int i = 0; i < NE; ++i) x[i] = acc_s[qt][r0 + i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = x[i] * nal2e;
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = 1.0f + e[i];
#pragma unroll
          for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
#pragma unroll
          for (int i = 0; i < NE; ++i) y[i] = x[i] * ais * e[i];
#endif
          if (kMode != 0) {
            const int th = rm[qt].jmax - k0 - 4 * hi;     // kMode 1: element rr is visible iff (rr & 3) + 8 (rr >> 2) <= th
            const int kb = k0 + 4 * hi;                   // kMode 3: the general rule and the row's functions
            const int fa = kFunc ? f_pre[qt] - kb : 0, fb0 = kFunc ? kb - f_lo[qt][0] : 0, fb1 = kFunc ? kb - f_lo[qt][1] : 0;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
              const int rr = r0 + i, off = (rr & 3) + 8 * (rr >> 2);
              bool ok = kMode == 1 ? off <= th : key_ok(k0 + off + 4 * hi, rm[qt]);
              if constexpr (kFunc && kMode == 3)
                ok = ok & ((off < fa) | ((unsigned)(off + fb0) < (unsigned)f_len[qt][0]) | ((unsigned)(off + fb1) < (unsigned)f_len[qt][1]));
              y[i] = ok ? y[i] : 0.f;
            }
          }
#pragma unroll
          for (int i = 0; i < NE; i += 2) out[i >> 1] = pack_bf16(y[i], y[i + 1]);
        };
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            uint32_t pk[4];
            group(qt, 8 * hf, pk);
            pdst[256 * qt + 64 * hf] = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
          }
        TICK(t7);
        TACC(5, t6, t7);
#if HSTU_TIMING
        tsum[6] += 1;
        t_prev = t7;
#endif
      };
      if constexpr (kFunc) {
        if (t_in < k.tf) tile(std::integral_constant<int, 0>{});
        else tile(std::integral_constant<int, 3>{});
      } else {
        if (t_in < k.tf) tile(std::integral_constant<int, 0>{});
        else if (m12 == 1) tile(std::integral_constant<int, 1>{});
        else tile(std::integral_constant<int, 2>{});
      }
    };
    // Two loops over ONE stream of iterations (as hstu_fwd_pair_kernel): block B's queries are loaded at the switch
    load_q(B0);
    for (int it = 0; it < T0; ++it) s_iter(it, B0, it);
    if (two) {
      load_q(B1);
      for (int it = T0; it < N; ++it) s_iter(it, B1, it - T0);
    }
    s_iter(N, B0, -1);
#if HSTU_TIMING
    t_dump();
#endif
    return;
  }
  // =========================== O waves: item `it - 1`; DMA of K item it + 1 and V item it ===========================
  // (the DMA's base address is an "s" operand of the inline asm: spell out that these offsets are wave-uniform)
  auto uni64 = [](int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi32 << 32) | lo);
  };
  const uint16_t* kg = a.k + uni64((int64_t)kstart * a.k_row + (int64_t)h * a.k_head);
  const uint16_t* vg = a.v + uni64((int64_t)kstart * a.v_row + (int64_t)h * a.v_head);
  const int dma_r = lane / CPR, dma_p = lane % CPR;
  constexpr int NO = NINS / 4;
  const int j_first = NO * pw;
  uint32_t kvoff[NO], vvoff[NO];
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int r = RPI * (j_first + u) + dma_r;
    kvoff[u] = (uint32_t)dma_r * (uint32_t)a.k_row * 2u + 16u * (uint32_t)(dma_p ^ (r & 15));
    vvoff[u] = (uint32_t)dma_r * (uint32_t)a.v_row * 2u + 16u * (uint32_t)(dma_p ^ ((r & 3) << 2));
  }
  auto dma16 = [&](const char* sbase_any, uint32_t voff, uint32_t lds_byte) {
    const char* sbase = reinterpret_cast<const char*>(uni64((int64_t)reinterpret_cast<uintptr_t>(sbase_any)));   // (folds away where hipcc sees the uniformity itself)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(sbase) : "memory");
  };
  auto issue_dma = [&](const uint16_t* g, int64_t g_row, const uint32_t (&voff)[NO], uint16_t* ring, int item) {
    const int n0 = __builtin_amdgcn_readfirstlane(item_n0(item));
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(ring + (item & 1) * TENS + RPI * j_first * ROWB));
    if (n0 + kBN <= s.L) {
      const char* sb = reinterpret_cast<const char*>(g + (int64_t)(n0 + RPI * j_first) * g_row);
      const int64_t step = (int64_t)RPI * g_row * 2;
#pragma unroll
      for (int u = 0; u < NO; ++u) dma16(sb + u * step, voff[u], dst + u * (RPI * ROWB * 2));
    } else {
      const uint32_t rowterm = (uint32_t)dma_r * (uint32_t)g_row * 2u;
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        const int row0 = n0 + RPI * (j_first + u);
        const int rowc = row0 < s.L ? row0 : s.L - 1;
        const uint32_t drop = row0 + 1 < s.L ? 0u : 0xffffffffu;
        dma16(reinterpret_cast<const char*>(g + (int64_t)rowc * g_row), voff[u] - (rowterm & drop), dst + u * (RPI * ROWB * 2));
      }
    }
  };
  const int il = lane & 15, g1 = (lane >> 4) & 1, vq = il >> 2;
  const int v_row_off = (4 * hi + vq) * ROWB + 4 * (il & 1);
  const int v_chunk_lo = 2 * g1 + ((il & 3) >> 1);
  auto v_frag = [&](const uint16_t* Vb, int dt, int ks) -> bf16x8_t {
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef short v8s_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
    const uint16_t* p0 = Vb + (16 * ks) * ROWB + v_row_off + 8 * ((4 * (dt ^ vq)) + v_chunk_lo);
    const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
    const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * ROWB));
    const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };
  f32x16_t acc_o[8];     // [4 column tiles of this wave's 128][2 q tiles]
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[t][r] = 0.f;
  };
  auto store_rows = [&](const Blk& k) {
    fence_a_2w(acc_o);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qloc = k.hrow0 + 32 * qt + l31;
      if (qloc < Lq) {
        uint16_t* op = a.out + (int64_t)(s.start + qloc) * a.o_row + (int64_t)h * a.o_head + 128 * sub;
#pragma unroll
        for (int dtl = 0; dtl < 4; ++dtl)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            uint2 o;
            o.x = pack_bf16(acc_o[2 * dtl + qt][4 * g4 + 0], acc_o[2 * dtl + qt][4 * g4 + 1]);
            o.y = pack_bf16(acc_o[2 * dtl + qt][4 * g4 + 2], acc_o[2 * dtl + qt][4 * g4 + 3]);
            *reinterpret_cast<uint2*>(op + 32 * dtl + 8 * g4 + 4 * hi) = o;
          }
      }
    }
  };
  zero_acc();
  if (N > 0) issue_dma(kg, a.k_row, kvoff, Kring, 0);
  auto o_iter = [&](int it, const Blk& k, int t_in) {     // k, t_in: the block and tile of item it - 1 (t_in < 0: none)
    pin_agpr_2w(acc_o);
    TICK(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (K item it, V item it - 1) have landed ...
    TICK(t1);
    __syncthreads();                                    // ... everyone's have, P[it - 1] is written, the other buffers are free
    TICK(t2);
    if (it + 1 < N) issue_dma(kg, a.k_row, kvoff, Kring, it + 1);
    if (it < N) issue_dma(vg, a.v_row, vvoff, Vring, it);
    pin_agpr_2w(acc_o);
    TICK(t3);
    TACC(0, t0, t1); TACC(1, t1, t2); TACC(2, t2, t3);
    const int tl = it - 1;
    {
      if (t_in >= k.t_lo && t_in < k.t_hi && !(kFunc && t_in >= k.g_lo && t_in < k.g_hi)) {
        const uint16_t* Vt = Vring + (tl & 1) * TENS;
        const u32x4_t* psrc = Pring + (((tl & 1) * 2 + half) * 2 * 4) * 64 + lane;
        bf16x8_t pf[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) pf[qt][ks] = __builtin_bit_cast(bf16x8_t, psrc[(4 * qt + ks) * 64]);
        constexpr int DB = 2, NVB = HSTU_Q2_VBUF, NBAT2 = 16 / DB;   // 16 V^T fragments per tile: (key slice ks, column tile dtl)
        bf16x8_t vfr[NVB][DB];
        auto load_v = [&](int bi) {
          const int ks = bi / (4 / DB), dtl0 = (bi % (4 / DB)) * DB;
#pragma unroll
          for (int u = 0; u < DB; ++u) vfr[bi % NVB][u] = v_frag(Vt, 4 * sub + dtl0 + u, ks);
        };
#pragma unroll
        for (int bi = 0; bi < NVB - 1; ++bi) load_v(bi);
        pin_agpr_2w(acc_o);
#pragma unroll
        for (int bi = 0; bi < NBAT2; ++bi) {
          const int ks = bi / (4 / DB), dtl0 = (bi % (4 / DB)) * DB;
          if (bi + NVB - 1 < NBAT2) load_v(bi + NVB - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < DB; ++u)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) mfma_a(acc_o[2 * (dtl0 + u) + qt], vfr[bi % NVB][u], pf[qt][ks]);
          __builtin_amdgcn_sched_barrier(0);
        }
        TICK(t4);
        TACC(5, t3, t4);
#if HSTU_TIMING
        tsum[6] += 1;
#endif
      }
    }
  };
  // iterations 0 .. T0 finish block A (its last tile is item T0 - 1, computed in iteration T0); T0 + 1 .. N are block B's
  for (int it = 0; it <= T0; ++it) o_iter(it, B0, it - 1);
  if (two) {
    store_rows(B0);          // block A's rows leave while the S waves already work on block B's first tile
    zero_acc();
    for (int it = T0 + 1; it <= N; ++it) o_iter(it, B1, it - 1 - T0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA may be in flight into LDS when the block retires)
#if HSTU_TIMING
  t_dump();
#endif
  store_rows(two ? B1 : B0);
}

// ---------------------------------------------------------------------------------------------------
// Backward (reference: hstu_varlen_bwd -> hstu_bwd.h, core maths :687-729).  With s = alpha <q,k>:
//   P  = M SiLU(s) / N            dV = P^T dO
//   dP = dO V^T                   dS = M dP SiLU'(s) alpha / N       dQ = dS K      dK = dS^T Q
// Two deterministic passes instead of the reference's fp32 atomics on dQ:
//   pass A (key-block owner):   S, dP -> dV, dK      pass B (query-block owner): S^T, dP^T -> dQ
// Both reuse the operand-layout trick of the forward: the accumulator of the first GEMM is already the B
// operand of the next one; the transposed operand is staged through LDS with the matching permutation.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dsilu_f(float x) {
  const float sg = __frcp_rn(1.0f + __expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}

// ---- register-prefetched tile staging (issue the global loads of tile n+1 before the MFMAs of tile n, write them
// to LDS after the barrier).  Rows past the sequence end are read clamped and zeroed with selects: no branches.
// RowTile: rows [row0, row0+NR) -> LDS [NR][D+8] row-major.
// Row stride (elements) of a tile that is read with ds_read_b64_tr_b16: 16 dwords mod 64 (48 at d = 64) keeps the 32
// lanes of a half-wave on 64 distinct banks.
template <int D> struct TrStride { static constexpr int value = D == 32 ? 32 : D + 32; };
// A operand [32 rows of tile dt][16 k of slice ks] of a GEMM whose A is the TRANSPOSE of a row-major LDS tile
// base[k][...]: lane (row = l31, k half = hi) receives the 8 k positions (j&3) + 8*(j>>2) + 4*hi of the slice -- the
// register order of the 32x32 accumulator that produced the B operand.  Two hardware transpose reads; within a 16-lane
// group out_l[j] = in_{4j + (l>>2)}[l & 3] (probed on gfx950).
template <int STRIDE>
__device__ __forceinline__ bf16x8_t tr_frag(const uint16_t* base, int dt, int ks, int lane, int hi) {
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef short v8s_t __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
  const int il = lane & 15;
  const uint16_t* p0 = base + (16 * ks + 4 * hi + (il >> 2)) * STRIDE + 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (il & 3);
  const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
  const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * STRIDE));
  const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int D, int NR>
struct RowTile {
  static constexpr int NCH = NR * D / 8, PT = (NCH + 255) / 256;
  u32x4_t r[PT];
  __device__ __forceinline__ void fetch(const uint16_t* src, int64_t row_stride, int row0, int L) {
    if constexpr (NCH % 256 == 0) {
      if (row0 + NR <= L) {   // tile inside the sequence (uniform): one address per thread, one 64-bit add per load
        const uint16_t* p = src + (int64_t)(row0 + (int)threadIdx.x / (D / 8)) * row_stride + 8 * ((int)threadIdx.x % (D / 8));
        const int64_t step = (int64_t)(256 / (D / 8)) * row_stride;
#pragma unroll
        for (int i = 0; i < PT; ++i) r[i] = *reinterpret_cast<const u32x4_t*>(p + i * step);
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int row = row0 + ch / (D / 8), dc = ch % (D / 8);
      r[i] = *reinterpret_cast<const u32x4_t*>(src + (int64_t)(row < L ? row : L - 1) * row_stride + 8 * (NCH % 256 == 0 || ch < NCH ? dc : 0));
    }
  }
  // row0 / L of the tile being written: rows past the sequence end become zeros here, not at fetch time, so that
  // nothing waits on the loads until the tile is needed
  __device__ __forceinline__ void commit(uint16_t* dst, int row0, int L) const {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      const u32x4_t v = (row0 + ch / (D / 8) < L) ? r[i] : z;
      if (NCH % 256 == 0 || ch < NCH) *reinterpret_cast<u32x4_t*>(dst + (ch / (D / 8)) * (D + 8) + 8 * (ch % (D / 8))) = v;
    }
  }
  // the same rows once more with the transpose-read stride (tr_frag): the image a TransTile would have transposed
  __device__ __forceinline__ void commit_tr(uint16_t* dst, int row0, int L) const {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      const u32x4_t v = (row0 + ch / (D / 8) < L) ? r[i] : z;
      if (NCH % 256 == 0 || ch < NCH) *reinterpret_cast<u32x4_t*>(dst + (ch / (D / 8)) * TrStride<D>::value + 8 * (ch % (D / 8))) = v;
    }
  }
};
// TransTile: the same rows transposed -> LDS [D][NR+8], row positions permuted inside every 16-group
// ({0-3,8-11,4-7,12-15}) so that an MFMA accumulator's register order is directly the k order of the next GEMM.
template <int D, int NR>
struct TransTile {
  static constexpr int NCH = (NR / 4) * (D / 8), PT = (NCH + 255) / 256;
  u32x4_t r[PT][4];
  __device__ __forceinline__ void fetch(const uint16_t* src, int64_t row_stride, int row0, int L) {
    if constexpr (NCH % 256 == 0) {
      if (row0 + NR <= L) {   // tile inside the sequence (uniform): four row pointers per thread, immediate column offsets
        const int gpos = (int)threadIdx.x % (NR / 4), g16 = gpos >> 2, pg = gpos & 3;
        const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);
        const uint16_t* p = src + (int64_t)(row0 + 16 * g16 + 4 * ak) * row_stride + 8 * ((int)threadIdx.x / (NR / 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int i = 0; i < PT; ++i) r[i][kk] = *reinterpret_cast<const u32x4_t*>(p + 8 * (256 / (NR / 4)) * i);
          p += row_stride;
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ch = (threadIdx.x + 256 * i) % NCH;   // threads past NCH repeat a block (their writes are skipped)
      const int gpos = ch % (NR / 4), dc = ch / (NR / 4);
      const int g16 = gpos >> 2, pg = gpos & 3;
      const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);
      const int r0 = row0 + 16 * g16 + 4 * ak;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int row = r0 + kk;
        r[i][kk] = *reinterpret_cast<const u32x4_t*>(src + (int64_t)(row < L ? row : L - 1) * row_stride + 8 * dc);
      }
    }
  }
  __device__ __forceinline__ void commit(uint16_t* dst, int row0, int L) const {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      if (NCH % 256 != 0 && ch >= NCH) continue;
      const int gpos = ch % (NR / 4), dc = ch / (NR / 4);
      const int g16 = gpos >> 2, pg = gpos & 3;
      const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);
      const int r0 = row0 + 16 * g16 + 4 * ak;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      const u32x4_t w0 = r0 + 0 < L ? r[i][0] : z, w1 = r0 + 1 < L ? r[i][1] : z, w2 = r0 + 2 < L ? r[i][2] : z,
                    w3 = r0 + 3 < L ? r[i][3] : z;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
        uint2 o;
        o.x = __builtin_amdgcn_perm(w1[e >> 1], w0[e >> 1], sel);
        o.y = __builtin_amdgcn_perm(w3[e >> 1], w2[e >> 1], sel);
        *reinterpret_cast<uint2*>(dst + (8 * dc + e) * (NR + 8) + 16 * g16 + 4 * pg) = o;
      }
    }
  }
};
struct NoTile {
  __device__ __forceinline__ void fetch(const uint16_t*, int64_t, int, int) {}
  __device__ __forceinline__ void commit(uint16_t*, int, int) const {}
  __device__ __forceinline__ void commit_tr(uint16_t*, int, int) const {}
};
template <bool C, typename A, typename B> struct SelT { typedef A type; };
template <typename A, typename B> struct SelT<false, A, B> { typedef B type; };

// B-operand fragments of the wave's own 32 rows (lane = row l31, 8 consecutive d per 16-slice); clamped rows
template <int D>
__device__ __forceinline__ void load_own_frags(bf16x8_t (&f)[D / 16], const uint16_t* base, int64_t row_stride, int row, int L, int hi) {
  const uint16_t* p = base + (int64_t)(row < L ? row : L - 1) * row_stride + 8 * hi;
#pragma unroll
  for (int sl = 0; sl < D / 16; ++sl) {
    const u32x4_t t = *reinterpret_cast<const u32x4_t*>(p + 16 * sl);
    f[sl] = __builtin_bit_cast(bf16x8_t, t);
  }
}

struct BwdAttnArgs {
  AttnArgs f;                  // q, k, v (+ strides); f.out unused
  const uint16_t* dout; int64_t do_row, do_head;
  uint16_t* dq; uint16_t* dk; uint16_t* dv;   // contiguous [T, H, D]
  // dS exchange between the dK pass and the dQ pass (nullable): 32 x 32 sub-tiles of dS in the dK pass's register layout
  // (lane = key, 16 query rows per lane: 32 bytes per lane, 2 KB per sub-tile), indexed [b][h][key group][query group]
  uint16_t* ds_ws;
  uint16_t* p_ws;              // same layout, P = SiLU(alpha S) / scale: the dV pass then needs no S recomputation either
  int ng;                      // 32-row groups per sequence the buffer is laid out for: ceil(max_seqlen / 32)
  int bq_kv;                   // query rows per step of the dK pass (the dQ pass must know which sub-tiles it wrote)
  // Jagged, chunked layout (round 4; NULL = the dense layout above): the buffer holds ONE chunk of (sequence, head) units at
  // a time, unit u = b H + h at tile offset plan_base[u] with its OWN ceil(L_b / 32)^2 tiles -- the scratch is sized by the
  // jagged sum, not by B x max_seqlen^2, and capped; the three passes run once per chunk and blocks of other chunks leave
  // at once (hstu_bwd_plan_kernel fills the plan on the device: no host read of the lengths).
  const int64_t* plan_base;
  const int32_t* plan_chunk;
  int chunk;
  int tri;                     // plain causal mask (no contextual rows, no window, no bias): only the sub-tiles with key group <=
                               // query group exist -- a unit has ng (ng + 1) / 2 tiles, index qg (qg + 1) / 2 + kg; the others
                               // are all zero, never written and never read (xch_absent)
  // d loss / d rab (hstu_api.cpp:659-667): [b][h][i][j] bf16, zero-filled by the caller; the dK pass writes dS there
  uint16_t* drab; int64_t drab_b, drab_h, drab_r;
  // `func` masks (round 6): per (function set, 128-key block) the query rows [first, last) that reach the block
  // (hstu_func_kvis_kernel, launched in front of the passes); NULL: every query tile is visited
  const int2* func_kvis; int64_t func_kvis_h;
  const int4* func_gext;       // per 32-row group: {smallest prefix, largest prefix, band hull}; 4 func_kvis_h entries per function set
};
// where a (sequence, head) unit's sub-tiles live: read ONCE per block (inside the step loops a load of the plan could not be
// hoisted over the stores and cost a scalar-load latency per step: +27 % on a jagged batch)
struct XchUnit { int64_t base; int ngb; int tri; };
__device__ __forceinline__ XchUnit xch_unit(const BwdAttnArgs& g, int b, int h, int L) {
  XchUnit u;
  if (g.plan_base) { u.base = g.plan_base[b * g.f.H + h]; u.ngb = (L + 31) >> 5; u.tri = g.tri; }
  else { u.base = ((int64_t)b * g.f.H + h) * g.ng * g.ng; u.ngb = g.ng; u.tri = 0; }
  return u;
}
__device__ __forceinline__ int64_t xch_tile(const XchUnit& u, int kg, int qg) {   // element offset of sub-tile (kg, qg): 1024 bf16 = 2 KB
  return (u.base + (u.tri ? (int64_t)qg * (qg + 1) / 2 + kg : (int64_t)kg * u.ngb + qg)) * 1024;
}
__device__ __forceinline__ bool xch_absent(const XchUnit& u, int kg, int qg) { return u.tri && kg > qg; }
__device__ __forceinline__ bool xch_other_chunk(const BwdAttnArgs& g, int b, int h) {
  return g.plan_chunk != nullptr && g.plan_chunk[b * g.f.H + h] != g.chunk;
}
// Plan of the chunked exchange (one wave): the (sequence, head) units, in index order, are cut into the FEWEST contiguous
// chunks of at most cap_tiles tiles (greedy), and among the cuts with that many chunks the one with the smallest largest
// chunk is taken -- every lane tries one capacity between the largest unit and the cap -- because a chunk's passes are as
// long as its longest column of blocks however few units it holds: chunks of (15, 15, 2) units cost three full passes,
// (11, 11, 10) not much more than one and a half.  A unit = ceil(L_b / 32)^2 tiles of 2 KB (tri: ng (ng + 1) / 2) in each of
// the dS and P regions.  nchunks_out[0] = chunks used (diagnostics: the host launches its upper bound).
// Round 5: the sequence lengths are staged in LDS by 256 threads first.  Read from global memory inside the 64 lanes' counting loops
// (three dependent passes over B x H units, one scalar-cache miss per step) the kernel took 34-43 us of every jagged backward
// (profiles/r04_step_timeline.txt); from LDS it is a few microseconds.
constexpr int kPlanMaxU = 4096;        // (sequence, head) units whose tile prefix is staged in LDS (more: the serial form)
constexpr int kPlanMaxChunks = 2048;   // chunk starts kept in LDS for the parallel assignment (more: the serial form)
// The greedy cut of the (sequence, head) units into chunks of at most c tiles -- `if (cur + need > c && cur > 0) new chunk` over the
// units in order -- without walking the units (round 5: the walk, once per lane for its trial capacity and once more by lane 0 with
// two global stores per unit, was 34-43 us of ONE wave in front of every jagged backward, profiles/r04_step_timeline.txt).
// F[u] = tiles in front of unit u (a prefix array in LDS, built by all threads); a chunk that starts at unit s ends in front of
// e = L(F[s] + c), L(x) = the largest e with F[e] <= x (binary search) -- or, when not even the first unit with tiles fits
// (z = L(F[s]) is that unit: the units s .. z - 1 have none), in front of z + 1: exactly the walk's `cur > 0` rule
// (tests/test_hstu_plan_cpu.py holds the equivalence).  A trial capacity costs chunks x log U LDS reads instead of U dependent
// steps; the assignment (base, chunk of every unit) is done by all 256 threads from the list of chunk starts.
struct PlanView {
  const int64_t* F; int U;
  __device__ __forceinline__ int L(int64_t x) const {      // largest e in [0, U] with F[e] <= x   (F[0] = 0 <= x)
    int lo = 0, hi = U;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (F[mid] <= x) lo = mid; else hi = mid - 1; }
    return lo;
  }
  __device__ __forceinline__ int next(int s, int64_t c) const {   // first unit of the chunk behind the one that starts at s
    const int64_t fs = F[s];
    const int z = L(fs), e = L(fs + c);
    return e > z ? e : (z + 1 < U ? z + 1 : U);
  }
  __device__ __forceinline__ int count(int64_t c) const {
    int n = 0;
    for (int s = 0; s < U; s = next(s, c)) ++n;
    return n > 0 ? n : 1;
  }
};
__global__ void __launch_bounds__(256) hstu_bwd_plan_kernel(const int* cu, int B, int H, int64_t cap_tiles, int tri, int64_t* base,
                                                            int32_t* chunk, int32_t* nchunks_out, int host_chunks, int* host_err) {
  __shared__ int64_t s_F[kPlanMaxU + 1];
  __shared__ int64_t s_part[256];
  __shared__ int s_start[kPlanMaxChunks + 1];
  __shared__ int64_t s_umax;
  __shared__ int s_n;
  __shared__ int64_t s_c;
  if (blockIdx.x != 0) return;
  const int tid = threadIdx.x;
  const int64_t U64 = (int64_t)B * H;
  const bool staged = U64 <= kPlanMaxU;
  if (staged) {
    const int U = (int)U64;
    // ---- tiles of every unit, their prefix (256 threads: a slice of units each, the slices' sums scanned by wave 0)
    const int per = (U + 255) / 256, u0 = tid * per < U ? tid * per : U, u1 = u0 + per < U ? u0 + per : U;
    int64_t sum = 0, mx = 1;
    {
      int b = u0 / H, h = u0 - b * H;
      int64_t nd = 0;
      for (int u = u0; u < u1; ++u) {
        if (u == u0 || h == 0) { const int64_t n = (cu[b + 1] - cu[b] + 31) >> 5; nd = tri ? n * (n + 1) / 2 : n * n; }
        s_F[u] = nd;                   // (the unit's own tiles for now)
        sum += nd;
        mx = nd > mx ? nd : mx;
        if (++h == H) { h = 0; ++b; }
      }
    }
    s_part[tid] = sum;
    if (tid == 0) s_umax = 1;
    __syncthreads();
    atomicMax((unsigned long long*)&s_umax, (unsigned long long)mx);
    if (tid < 64) {                    // exclusive scan of the 256 slice sums: four per lane, then across the wave
      int64_t v[4], run = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = s_part[4 * tid + k]; run += v[k]; }
      int64_t inc = run;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = ((int64_t)__shfl_up((int)(inc >> 32), off, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)inc, off, 64);
        if (tid >= off) inc += o;
      }
      int64_t ex = inc - run;
#pragma unroll
      for (int k = 0; k < 4; ++k) { s_part[4 * tid + k] = ex; ex += v[k]; }
    }
    __syncthreads();
    {
      int64_t run = s_part[tid];
      for (int u = u0; u < u1; ++u) { const int64_t nd = s_F[u]; s_F[u] = run; run += nd; }
      if (u1 == U) s_F[U] = run;       // (the thread whose slice ends the units; empty slices behind it hold the same total)
    }
    __syncthreads();
    if (s_F[U] <= cap_tiles) {          // everything fits one chunk (the usual case under the default cap): no capacity search
      for (int u = tid; u < U; u += 256) { base[u] = s_F[u]; chunk[u] = 0; }
      if (tid == 0) nchunks_out[0] = 1;
      return;
    }
    const PlanView v{s_F, U};
    if (tid < 64) {
      const int64_t umax = s_umax;
      const int best = v.count(cap_tiles);
      const int64_t lo = umax < cap_tiles ? umax : cap_tiles;
      const int64_t mine = lo + (cap_tiles - lo) * (tid + 1) / 64;                     // lane 63 tries the cap itself
      const bool ok = v.count(mine) <= best;
      const unsigned long long okm = __ballot(ok);
      const int pick = __ffsll(okm) - 1;                                               // the smallest capacity that still needs `best` chunks
      if (tid == 0) {
        const int64_t c = lo + (cap_tiles - lo) * (pick + 1) / 64;
        int n = 0;
        for (int s0 = 0; s0 < U; s0 = v.next(s0, c)) { if (n < kPlanMaxChunks) s_start[n] = s0; ++n; }
        if (n == 0) { s_start[0] = 0; n = 1; }
        s_n = n; s_c = c;
        nchunks_out[0] = n;
        // the host launches `host_chunks` chunk passes from a bound over (T, B, max L); a plan that needs more (a token hint that
        // does not belong to this batch) would leave units unprocessed: say so where the next call of the library sees it
        if (n > host_chunks && host_err) *host_err = n;
      }
    }
    __syncthreads();
    const int n = s_n;
    if (n <= kPlanMaxChunks) {
      for (int u = tid; u < U; u += 256) {
        int lo = 0, hi = n - 1;                    // the chunk of unit u: the last start <= u
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[mid] <= u) lo = mid; else hi = mid - 1; }
        base[u] = s_F[u] - s_F[s_start[lo]];
        chunk[u] = lo;
      }
      return;
    }
    // (more chunks than the list holds: lane 0 walks them below with the capacity chosen above)
  }
  // ---- the serial form: one lane walks the units (more than kPlanMaxU units / more than kPlanMaxChunks chunks)
  if (tid >= 64) return;
  const int lane = tid;
  auto unit = [&](int b) -> int64_t {
    const int64_t ngb = (int64_t)((cu[b + 1] - cu[b] + 31) >> 5);
    return tri ? ngb * (ngb + 1) / 2 : ngb * ngb;
  };
  auto count = [&](int64_t c) -> int {
    int64_t cur = 0;
    int n = 1;
    for (int b = 0; b < B; ++b) {
      const int64_t need = unit(b);
      for (int h = 0; h < H; ++h) {
        if (cur + need > c && cur > 0) { ++n; cur = 0; }
        cur += need;
      }
    }
    return n;
  };
  int64_t c;
  if (staged) c = s_c;
  else {
    int64_t umax = 1;
    for (int b = 0; b < B; ++b) { const int64_t u = unit(b); umax = u > umax ? u : umax; }
    const int best = count(cap_tiles);
    const int64_t lo = umax < cap_tiles ? umax : cap_tiles;
    const int64_t mine = lo + (cap_tiles - lo) * (lane + 1) / 64;
    const bool ok = count(mine) <= best;
    const unsigned long long okm = __ballot(ok);
    const int pick = __ffsll(okm) - 1;
    c = lo + (cap_tiles - lo) * (pick + 1) / 64;
  }
  if (lane != 0) return;
  int64_t cur = 0;
  int n = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t need = unit(b);
    for (int h = 0; h < H; ++h) {
      if (cur + need > c && cur > 0) { ++n; cur = 0; }
      base[b * H + h] = cur;
      chunk[b * H + h] = n;
      cur += need;
    }
  }
  nchunks_out[0] = n + 1;
  if (n + 1 > host_chunks && host_err) *host_err = n + 1;
}

// Query steps (multiples of `bq` rows) the dK pass runs for the key block n0 .. n0 + kBM - 1: [0, c_end) and [jump, lim).
// The contextual rows (they see all history keys), then from the step holding row n0 on (causal) or from the first row
// whose right window reaches key n0 (non causal); up to the last row whose left window still reaches the block.
// hstu_bwd_v_p_kernel and hstu_bwd_q_ds_kernel replay it to tell which exchanged sub-tiles exist.
struct KvSpan { int jump, c_end, lim; };
__device__ __forceinline__ KvSpan kv_span(const AttnArgs& a, const SeqInfo& s, int n0, int bq) {
  KvSpan v{0, 0, s.L};
  if (a.causal) {
    v.jump = (n0 / bq) * bq;
    if (s.has_ctx && s.c > 0 && n0 < s.hlen) v.c_end = ((s.c + bq - 1) / bq) * bq;
  } else if (a.wskip && a.wr >= 0 && n0 > a.wr) {
    v.jump = ((n0 - a.wr) / bq) * bq;
  }
  if (a.wskip && a.wl >= 0 && n0 + kBM + a.wl < v.lim) v.lim = n0 + kBM + a.wl;
  return v;
}
__device__ __forceinline__ bool kv_visited(const KvSpan& v, int step) { return (step < v.c_end || step >= v.jump) && step < v.lim; }
// `func` masks: the span narrowed to the query rows [vis.x, vis.y) that reach the key block at all (hstu_func_kvis_kernel's table).
// ONE statement of the rule for the pass that writes the exchanged sub-tiles and the passes that read them.
__device__ __forceinline__ KvSpan kv_span_clip(KvSpan v, int2 vis, int bq) {
  const int first = vis.x < vis.y ? (vis.x / bq) * bq : 0x7fffff00;
  if (first > v.jump) v.jump = first;
  if (vis.y < v.lim) v.lim = vis.y;
  if (v.c_end > v.lim) v.c_end = v.lim;
  return v;
}
__device__ __forceinline__ int2 func_kvis_of(const BwdAttnArgs& g, int start, int b, int h, int n0) {
  const int64_t kidx = func_kvis_index(start, b, n0);
  if (g.f.func && g.func_kvis && kidx < g.func_kvis_h) return g.func_kvis[(int64_t)(g.f.func_h ? h : 0) * g.func_kvis_h + kidx];
  return make_int2(0, 0x7fffffff);
}

// pass A: one workgroup = 128 keys (32 per wave) of one (sequence, head); loops over query tiles of BQ rows.
// MODE 0: dV and dK together (d <= 64).  Larger d: the two output accumulators plus the S / dP accumulators and the
// K / V fragments exceed the register file of one wave, so the pass is split: MODE 1 = dV only (S -> P -> dV),
// MODE 2 = dK only (S, dP -> dS -> dK).  Loop structure as in the forward: register-prefetched tiles, explicit
// AGPR output accumulators, double-buffered LDS fragment batches, branch-free mask.
template <int D, int BQ, int MODE, bool kPre, bool kXP = false, bool kRab = false>
__global__ void __launch_bounds__(256) hstu_bwd_kv_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr bool kDV = MODE != 2, kDK = MODE != 1;   // kXP (MODE 2): P is computed as well and left for the dV pass
  constexpr int RS = D + 8, TS = BQ + 8, NT = BQ / 32;
  constexpr bool kTR = HSTU_BWD_TR != 0;            // Q^T / dO^T operands by transpose reads from row-major images
  constexpr int TRS = TrStride<D>::value;
  constexpr int TIMG = kTR ? BQ * TRS : D * TS;     // elements of one "transposed" image
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Qs = smem;                              // [BQ][RS]   (S)
  uint16_t* dOs = Qs + BQ * RS;                     // [BQ][RS]   (dP; kDK only)
  uint16_t* Qt = dOs + (kDK ? BQ * RS : 0);         // [D][TS] or [BQ][TRS]   (dK; kDK only)
  uint16_t* dOt = Qt + (kDK ? TIMG : 0);            // [D][TS] or [BQ][TRS]   (dV; kDV only)

  const BlockSeq bs = seq_head_of_block(a);   // grid (H, B, blocks): see launch_fwd
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int n0 = bs.z * kBM;   // earliest key blocks (seen by most queries) first
  if (n0 >= s.L || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int key0 = n0 + 32 * wv;
  const int kj = key0 + l31;
  const bool wave_live = key0 < s.L;
  const bool plain = !s.has_ctx && !s.has_tgt && s.wl < 0 && s.wr < 0;   // block-uniform: mask is kj <= qi (causal) or kj < L
  // The general masks from the KEY's side (the lane owns key kj, the query rows run over its registers): per-lane constants
  // once, a few compares per element -- row_mask() per element would cost a division by the group size each.
  //   contextual row i < c: sees the history keys;  other rows: kj <= i;  a target key is seen only up to the end of its
  //   own group (row_mask / key_ok: jlo(i) <= kj  <=>  i < hlen + (floor((kj - hlen) / g) + 1) g for kj <= i)
  const bool key_in = kj < s.L, key_hist = kj < s.hlen;
  const int key_iend = (s.has_tgt && kj >= s.hlen) ? s.hlen + ((kj - s.hlen) / a.group + 1) * a.group : 0x7fffffff;
  const int ctx_end = s.has_ctx ? s.c : 0;

  const uint16_t* qbase = a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head;
  const uint16_t* dobase = g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head;
  // K / V fragments of the wave's 32 keys.  Keys beyond the sequence read a clamped row: kj < L is part of the mask.
  bf16x8_t kf[D / 16], vf[kDK ? D / 16 : 1];
  load_own_frags<D>(kf, a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head, a.k_row, kj, s.L, hi);
  if constexpr (kDK) load_own_frags<D>(vf, a.v + (int64_t)s.start * a.v_row + (int64_t)h * a.v_head, a.v_row, kj, s.L, hi);

  f32x16_t acc_dv[kDV ? D / 32 : 1], acc_dk[kDK ? D / 32 : 1];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (kDV) acc_dv[dt][r] = 0.f;
      if (kDK) acc_dk[dt][r] = 0.f;
    }

  const float neg_alpha_log2e = -a.alpha * 1.4426950408889634f;
  const float c_p = a.alpha * a.inv_scale;   // P  = acc * c_p * sigmoid
  const float c_ds = a.alpha * a.inv_scale;  // dS = dP * c_ds * sigmoid * (1 + x (1 - sigmoid))
  // query tiles that can see this key block: the tiles holding contextual rows (they see all history keys), then
  // from the tile containing row n0 on (causal); everything (non causal)
  const KvSpan span = kv_span(a, s, n0, BQ);
  int jump = span.jump, c_end = span.c_end, i_lim = span.lim;
  if constexpr (kRab) {
    const int64_t kidx = func_kvis_index(s.start, b, n0);
    if (a.func && g.func_kvis && kidx < g.func_kvis_h) {   // `func` masks: only the query tiles between the first and the last row that reach this key block
      const int2 vis = g.func_kvis[(int64_t)(a.func_h ? h : 0) * g.func_kvis_h + kidx];
      const int first = vis.x < vis.y ? (vis.x / BQ) * BQ : 0x7fffff00;
      if (first > jump) jump = first;
      if (vis.y < i_lim) i_lim = vis.y;
      if (c_end > i_lim) c_end = i_lim;
    }
  }
  auto advance = [&](int i) { i += BQ; return (i >= c_end && i < jump) ? jump : i; };
  int i0 = c_end > 0 ? 0 : jump;

  // kTR: the "transposed" images are second row-major copies of the SAME fetched registers (no second global read,
  // no perm network); only MODE 1, which has no other use for dO rows, fetches dO for that image alone
  typename SelT<true, RowTile<D, BQ>, NoTile>::type q_rows;
  typename SelT<kDK || (kTR && kDV), RowTile<D, BQ>, NoTile>::type do_rows;
  typename SelT<kDK && !kTR, TransTile<D, BQ>, NoTile>::type q_tr;
  typename SelT<kDV && !kTR, TransTile<D, BQ>, NoTile>::type do_tr;
  auto fetch_all = [&](int i) {
    q_rows.fetch(qbase, a.q_row, i, s.L);
    do_rows.fetch(dobase, g.do_row, i, s.L);
    q_tr.fetch(qbase, a.q_row, i, s.L);
    do_tr.fetch(dobase, g.do_row, i, s.L);
  };
  // kPre: small head dims run several waves per SIMD, which hides the staging latency better than holding a tile in
  // registers does (the prefetch registers would halve the occupancy); d = 256 runs one wave per SIMD and prefetches
  if (kPre && i0 < i_lim) fetch_all(i0);
#if HSTU_TIMING
  unsigned tsum[7] = {0, 0, 0, 0, 0, 0, 0};
  const unsigned t_start = tick();
#endif
  for (; i0 < i_lim; i0 = advance(i0)) {
    if (kDV) pin_agpr(acc_dv);
    if (kDK) pin_agpr(acc_dk);
    TICK(t0);
    __syncthreads();
    TICK(t1);
    if (!kPre) fetch_all(i0);
    TICK(t1b);
    q_rows.commit(Qs, i0, s.L);
    if (kDK) do_rows.commit(dOs, i0, s.L);
    if constexpr (kTR) {
      if (kDK) q_rows.commit_tr(Qt, i0, s.L);
      if (kDV) do_rows.commit_tr(dOt, i0, s.L);
    } else {
      q_tr.commit(Qt, i0, s.L);
      do_tr.commit(dOt, i0, s.L);
    }
    if (kDV) pin_agpr(acc_dv);
    if (kDK) pin_agpr(acc_dk);
    TICK(t2);
    __syncthreads();
    TICK(t3);
    if (kPre) {
      const int nx = advance(i0);
      if (nx < i_lim) fetch_all(nx);
    }
    if (kDV) pin_agpr(acc_dv);
    if (kDK) pin_agpr(acc_dk);
    TACC(0, t0, t1); TACC(1, t1, t1b); TACC(2, t1b, t2); TACC(3, t2, t3);
    if (!wave_live) continue;
    TICK(t4);
    // GEMM 1 / 2: S[q x keys] = Q K^T, dP[q x keys] = dO V^T (A from LDS rows, B = register fragments)
    f32x16_t acc_s[NT], acc_p[kDK ? NT : 1];
    {
      constexpr int SLB = (kDK ? 4 : 8) / NT < D / 16 ? (kDK ? 4 : 8) / NT : D / 16;   // slices per batch (~8 MFMAs)
      constexpr int NBAT = (D / 16) / SLB;
      bf16x8_t qa[2][SLB][NT], da[2][kDK ? SLB : 1][NT];
      auto load_b = [&](int bi, int buf) {
#pragma unroll
        for (int u = 0; u < SLB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int off = (32 * t + l31) * RS + 16 * (SLB * bi + u) + 8 * hi;
            qa[buf][u][t] = *reinterpret_cast<const bf16x8_t*>(Qs + off);
            if (kDK) da[buf][u][t] = *reinterpret_cast<const bf16x8_t*>(dOs + off);
          }
      };
      load_b(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT; ++bi) {
        if (bi + 1 < NBAT) load_b(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SLB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int sl = SLB * bi + u;
            if (sl == 0) mfma_v0(acc_s[t], qa[bi & 1][u][t], kf[sl]);
            else mfma_v(acc_s[t], qa[bi & 1][u][t], kf[sl]);
            if (kDK) {
              if (sl == 0) mfma_v0(acc_p[t], da[bi & 1][u][t], vf[sl]);
              else mfma_v(acc_p[t], da[bi & 1][u][t], vf[sl]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    fence_v(acc_s);
    if (kDK) fence_v(acc_p);
    TICK(t5);
    TACC(4, t4, t5);
    if constexpr (kRab) {   // lane = key kj, registers = query rows: rab[qi][kj]
      if (a.func) {         // (the bounds of a query row are the same words for the 32 lanes of a half-wave: one transaction)
        // a step whose query rows all see every key of the block (the block lies below their smallest prefix) needs no test
        bool func_full = false;
        if (g.func_gext) {
          const int64_t gi = func_gext_index(s.start, b, i0);
          if (gi + NT <= 4 * g.func_kvis_h) {
            int fmin = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int f = g.func_gext[(int64_t)(a.func_h ? h : 0) * 4 * g.func_kvis_h + gi + t].x;
              fmin = f < fmin ? f : fmin;
            }
            func_full = n0 + kBM <= fmin;
          }
        }
        if (kj < s.L && !func_full) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const int qi = i0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
              if (qi < s.L && !(s.has_ctx && qi < s.c && kj < s.hlen) &&
                  !func_sees(a.func + (int64_t)h * a.func_h + s.start + qi, a.func_p, a.n_func, kj)) acc_s[t][rr] += a.func_neg;
            }
        }
      } else
      if (kj < s.L) {
        const uint16_t* col = a.rab + (int64_t)b * a.rab_b + (int64_t)h * a.rab_h + kj;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int qi = i0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            if (qi < s.L) acc_s[t][rr] += bf16_bits_to_f32(col[(int64_t)qi * a.rab_r]);
          }
      }
    }
    if (kDV) pin_agpr(acc_dv);
    if (kDK) pin_agpr(acc_dk);
    // P and dS packed as B operands (k = query rows held in the registers, lane = key).  The mask variant is chosen once
    // per step and wave and the 32 elements of a lane are then ONE basic block: with the (block-uniform) choice inside the
    // element loop hipcc emitted a branch per element, every SiLU' chain (exp -> add -> rcp -> mul ...) ran alone with its
    // full latency exposed, and this phase took 7.0 K of the 15.6 K cycles of a step at d = 256 (HSTU_TIMING stamps).
    bf16x8_t pf[kDV ? BQ / 16 : 1], sf[kDK ? BQ / 16 : 1];
    auto elementwise = [&](auto modec) {
      // 0: every (query, key) pair of the wave's step is visible; 1: key <= query (plain causal); 2: key < L;
      // 3: contextual / target rows (causal); 4: local window
      constexpr int kMask = decltype(modec)::value;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint32_t pk[8], sk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p2[2], s2[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rr = r + u;
            const int qi = i0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            bool ok = true;
            if constexpr (kMask == 1) ok = kj <= qi;
            else if constexpr (kMask == 2) ok = kj < s.L;
            else if constexpr (kMask == 3) ok = (qi < ctx_end ? key_hist : ((kj <= qi) & key_in)) & (key_hist | (qi < key_iend));
            else if constexpr (kMask == 4)
              ok = key_in & ((s.wl < 0) | (kj >= qi - s.wl)) & (a.causal ? (kj <= qi) : ((s.wr < 0) | (kj <= qi + s.wr)));
            const float acc = acc_s[t][rr];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc * neg_alpha_log2e));
            if (kDV || kXP) p2[u] = ok ? acc * c_p * sg : 0.f;
            if (kDK) {
              const float dsv = ds_value(acc, acc_p[t][rr], sg, a.alpha, c_ds);
              s2[u] = ok ? dsv : 0.f;
            }
          }
          if (kDV || kXP) pk[r >> 1] = pack_bf16(p2[0], p2[1]);
          if (kDK) sk[r >> 1] = pack_bf16(s2[0], s2[1]);
          if constexpr (kRab && kDK) {   // d rab = dS (x = alpha (q.k + rab): the same factor alpha as d (q.k))
            if (g.drab && kj < s.L) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int qi = i0 + 32 * t + ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * hi;
                if (qi < s.L)
                  g.drab[(int64_t)b * g.drab_b + (int64_t)h * g.drab_h + (int64_t)qi * g.drab_r + kj] = (uint16_t)(sk[r >> 1] >> (16 * u));
              }
            }
          }
        }
        if constexpr (kXP) {
          if (i0 + 32 * t < s.L && !xch_absent(xu, key0 >> 5, (i0 >> 5) + t)) {   // P for the dV pass, same layout as dS below
            u32x4_t* tp = reinterpret_cast<u32x4_t*>(g.p_ws + xch_tile(xu, key0 >> 5, (i0 >> 5) + t)) + 2 * lane;
            tp[0] = u32x4_t{pk[0], pk[1], pk[2], pk[3]}; tp[1] = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
          }
        }
        if (kDV) {
          const u32x4_t x0 = {pk[0], pk[1], pk[2], pk[3]}, x1 = {pk[4], pk[5], pk[6], pk[7]};
          pf[2 * t] = __builtin_bit_cast(bf16x8_t, x0); pf[2 * t + 1] = __builtin_bit_cast(bf16x8_t, x1);
        }
        if (kDK) {
          const u32x4_t y0 = {sk[0], sk[1], sk[2], sk[3]}, y1 = {sk[4], sk[5], sk[6], sk[7]};
          sf[2 * t] = __builtin_bit_cast(bf16x8_t, y0); sf[2 * t + 1] = __builtin_bit_cast(bf16x8_t, y1);
          if (g.ds_ws && i0 + 32 * t < s.L && !xch_absent(xu, key0 >> 5, (i0 >> 5) + t)) {   // hand dS to the dQ pass: the B-operand registers as they are, 32 bytes per lane
            u32x4_t* tp = reinterpret_cast<u32x4_t*>(g.ds_ws + xch_tile(xu, key0 >> 5, (i0 >> 5) + t)) + 2 * lane;
            tp[0] = y0; tp[1] = y1;
          }
        }
      }
    };
    if (!plain) {
      if (s.wl >= 0 || s.wr >= 0) elementwise(std::integral_constant<int, 4>{}); else elementwise(std::integral_constant<int, 3>{});
    } else if (a.causal) {
      if (key0 + 31 <= i0) elementwise(std::integral_constant<int, 0>{}); else elementwise(std::integral_constant<int, 1>{});
    } else {
      if (key0 + 31 < s.L) elementwise(std::integral_constant<int, 0>{}); else elementwise(std::integral_constant<int, 2>{});
    }
    if (kDV) pin_agpr(acc_dv);
    if (kDK) pin_agpr(acc_dk);
    TICK(t6);
    TACC(5, t5, t6);
#if HSTU_TIMING
    tsum[6] += 1;
#endif
    // GEMM 3 / 4: dV^T[D x keys] += dO^T P, dK^T[D x keys] += Q^T dS
    {
      constexpr int NDT = D / 32;
      constexpr int DB = (MODE == 0 ? 4 : 8) < NDT ? (MODE == 0 ? 4 : 8) : NDT;   // d tiles per batch (~8 MFMAs)
      constexpr int NBAT2 = (BQ / 16) * (NDT / DB);
      bf16x8_t fa[2][kDV ? DB : 1], fb[2][kDK ? DB : 1];
      auto load_t = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) {
          if constexpr (kTR) {
            if (kDV) fa[buf][u] = tr_frag<TRS>(dOt, dt0 + u, ks, lane, hi);
            if (kDK) fb[buf][u] = tr_frag<TRS>(Qt, dt0 + u, ks, lane, hi);
          } else {
            const int off = (32 * (dt0 + u) + l31) * TS + 16 * ks + 8 * hi;
            if (kDV) fa[buf][u] = *reinterpret_cast<const bf16x8_t*>(dOt + off);
            if (kDK) fb[buf][u] = *reinterpret_cast<const bf16x8_t*>(Qt + off);
          }
        }
      };
      load_t(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        if (bi + 1 < NBAT2) load_t(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) {
          if (kDV) mfma_a(acc_dv[dt0 + u], fa[bi & 1][u], pf[ks]);
          if (kDK) mfma_a(acc_dk[dt0 + u], fb[bi & 1][u], sf[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (kDV) fence_a(acc_dv);
  if (kDK) fence_a(acc_dk);
#if HSTU_TIMING
  if (kXP) {
    const unsigned t_end = tick();
    if (lane == 0) {
      const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long* d = g_hstu_dbg + ((size_t)(blk * 4 + wv) % 65536) * 8;
      for (int i = 0; i < 6; ++i) d[i] = tsum[i];
      d[6] = tsum[6]; d[7] = t_end - t_start;
    }
  }
#endif
  if (kj < s.L) {
    uint16_t* dvp = g.dv + ((int64_t)(s.start + kj) * a.H + h) * D;
    uint16_t* dkp = g.dk + ((int64_t)(s.start + kj) * a.H + h) * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        if (kDV) {
          o.x = pack_bf16(acc_dv[dt][4 * g4 + 0], acc_dv[dt][4 * g4 + 1]);
          o.y = pack_bf16(acc_dv[dt][4 * g4 + 2], acc_dv[dt][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g4 + 4 * hi) = o;
        }
        if (kDK) {
          o.x = pack_bf16(acc_dk[dt][4 * g4 + 0], acc_dk[dt][4 * g4 + 1]);
          o.y = pack_bf16(acc_dk[dt][4 * g4 + 2], acc_dk[dt][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g4 + 4 * hi) = o;
        }
      }
  }
}

// pass B: one workgroup = 128 queries (32 per wave); loops over key tiles of BK keys -> dQ
template <int D, int BK, bool kPre, bool kRab = false>
__global__ void __launch_bounds__(256) hstu_bwd_q_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr int RS = D + 8, TS = BK + 8, NT = BK / 32;
  constexpr bool kTR = HSTU_BWD_TR != 0;   // K^T operand of the dQ GEMM by transpose reads from a row-major image
  constexpr int TRS = TrStride<D>::value;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Ks = smem;               // [BK][RS]
  uint16_t* Vs = Ks + BK * RS;       // [BK][RS]
  uint16_t* Kt = Vs + BK * RS;       // [D][TS] or [BK][TRS]

  const BlockSeq bs = seq_head_of_block(a);   // grid (H, B, blocks): see launch_fwd
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int nblk = (s.L + kBM - 1) / kBM;
  if (bs.z >= nblk) return;
  const int m0 = row_block_of_rank(bs.z, nblk, a, b) * kBM;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qrow0 = m0 + 32 * wv;
  const int qi = qrow0 + l31;
  const bool wave_live = qrow0 < s.L;
  int last_row = m0 + kBM - 1 < s.L - 1 ? m0 + kBM - 1 : s.L - 1;
  int n_end = s.L;
  if (a.causal) { n_end = last_row + 1; if (s.has_ctx && m0 < s.c && s.hlen > n_end) n_end = s.hlen; }
  int w_last = qrow0 + 31 < s.L - 1 ? qrow0 + 31 : s.L - 1;
  int w_end = s.L;
  if (a.causal) { w_end = w_last + 1; if (s.has_ctx && qrow0 < s.c && s.hlen > w_end) w_end = s.hlen; }
  n_end = band_key_end(a, last_row, n_end);
  w_end = band_key_end(a, w_last, w_end);
  int n_beg = band_key_begin(a, m0, BK), w_beg = band_key_begin(a, qrow0, BK);
  FuncExt wx{0x7fffffff, 0, 0x7fffffff, 0};       // (no functions: every tile "hits"; with functions and no skipping: no tile is "full")
  if constexpr (kRab) {
    if (a.func && a.wskip) {        // key tiles the block / the wave can reach, from the extents of their rows' functions (as in the forward)
      __shared__ int s_fx[3 * 4];
      wx = func_ext_wave(func_ext_row(a, h, (int64_t)s.start + qi, qi < s.L, (s.has_ctx && qi < s.c) ? s.hlen : 0));
      const FuncExt bx = func_ext_block<4>(wx, wv, lane, s_fx);
      const int be = func_ext_end(bx), bb = func_ext_begin(bx), we = func_ext_end(wx), wb = func_ext_begin(wx);
      if (be < n_end) n_end = be;
      if (we < w_end) w_end = we;
      if (bb > n_beg) n_beg = bb >= n_end ? n_end : (bb / BK) * BK;
      if (wb > w_beg) w_beg = wb >= 0x7fffff00 ? 0x7fffff00 : (wb / BK) * BK;
    }
  }

  // Q / dO fragments of the wave's 32 queries; rows beyond the sequence read a clamped row and are never stored
  bf16x8_t qf[D / 16], dof[D / 16];
  load_own_frags<D>(qf, a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head, a.q_row, qi, s.L, hi);
  load_own_frags<D>(dof, g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head, g.do_row, qi, s.L, hi);
  const RowMask rm = row_mask(qi < s.L ? qi : s.L - 1, s, a.causal, a.group);
  const float neg_alpha_log2e = -a.alpha * 1.4426950408889634f;
  const float c_ds = a.alpha * a.inv_scale;
  f32x16_t acc_dq[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dq[dt][r] = 0.f;

  const uint16_t* kbase = a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head;
  const uint16_t* vbase = a.v + (int64_t)s.start * a.v_row + (int64_t)h * a.v_head;
  RowTile<D, BK> k_rows, v_rows;
  typename SelT<!kTR, TransTile<D, BK>, NoTile>::type k_tr;
  auto fetch_all = [&](int n) {
    k_rows.fetch(kbase, a.k_row, n, s.L);
    v_rows.fetch(vbase, a.v_row, n, s.L);
    k_tr.fetch(kbase, a.k_row, n, s.L);
  };
  if (kPre && n_end > n_beg) fetch_all(n_beg);
  for (int n0 = n_beg; n0 < n_end; n0 += BK) {
    pin_agpr(acc_dq);
    __syncthreads();
    if (!kPre) fetch_all(n0);
    k_rows.commit(Ks, n0, s.L);
    v_rows.commit(Vs, n0, s.L);
    if constexpr (kTR) k_rows.commit_tr(Kt, n0, s.L); else k_tr.commit(Kt, n0, s.L);
    pin_agpr(acc_dq);
    __syncthreads();
    if (kPre && n0 + BK < n_end) fetch_all(n0 + BK);
    pin_agpr(acc_dq);
    if (!wave_live || n0 >= w_end || n0 < w_beg) continue;
    if constexpr (kRab) { if (!func_ext_hits(wx, n0, n0 + BK)) continue; }
    f32x16_t acc_s[NT], acc_p[NT];   // S^T, dP^T [keys x q]
    {
      constexpr int SLB = 4 / NT < D / 16 ? 4 / NT : D / 16;
      constexpr int NBAT = (D / 16) / SLB;
      bf16x8_t ka[2][SLB][NT], va[2][SLB][NT];
      auto load_b = [&](int bi, int buf) {
#pragma unroll
        for (int u = 0; u < SLB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int off = (32 * t + l31) * RS + 16 * (SLB * bi + u) + 8 * hi;
            ka[buf][u][t] = *reinterpret_cast<const bf16x8_t*>(Ks + off);
            va[buf][u][t] = *reinterpret_cast<const bf16x8_t*>(Vs + off);
          }
      };
      load_b(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT; ++bi) {
        if (bi + 1 < NBAT) load_b(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SLB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int sl = SLB * bi + u;
            if (sl == 0) { mfma_v0(acc_s[t], ka[bi & 1][u][t], qf[sl]); mfma_v0(acc_p[t], va[bi & 1][u][t], dof[sl]); }
            else { mfma_v(acc_s[t], ka[bi & 1][u][t], qf[sl]); mfma_v(acc_p[t], va[bi & 1][u][t], dof[sl]); }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    fence_v(acc_s);
    fence_v(acc_p);
    if constexpr (kRab) {
      if (a.func) { if (n0 + BK > wx.f0min) add_func_row<NT>(acc_s, a, h, (int64_t)s.start + qi, qi < s.L, n0, hi, s.L, (s.has_ctx && qi < s.c) ? s.hlen : 0); }
      else {
      const uint16_t* row = qi < s.L ? a.rab + (int64_t)b * a.rab_b + (int64_t)h * a.rab_h + (int64_t)qi * a.rab_r : nullptr;
      add_rab_row<NT>(acc_s, row, n0, hi, s.L);
      }
    }
    pin_agpr(acc_dq);
    bf16x8_t sf[BK / 16];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      uint32_t sk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float s2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rr = r + u;
          const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          const float acc = acc_s[t][rr];
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc * neg_alpha_log2e));
          s2[u] = key_ok(key, rm) ? ds_value(acc, acc_p[t][rr], sg, a.alpha, c_ds) : 0.f;
        }
        sk[r >> 1] = pack_bf16(s2[0], s2[1]);
      }
      const u32x4_t y0 = {sk[0], sk[1], sk[2], sk[3]}, y1 = {sk[4], sk[5], sk[6], sk[7]};
      sf[2 * t] = __builtin_bit_cast(bf16x8_t, y0); sf[2 * t + 1] = __builtin_bit_cast(bf16x8_t, y1);
    }
    pin_agpr(acc_dq);
    {
      constexpr int NDT = D / 32;
      constexpr int DB = 8 < NDT ? 8 : NDT;
      constexpr int NBAT2 = (BK / 16) * (NDT / DB);
      bf16x8_t fk[2][DB];
      auto load_t = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u)
          if constexpr (kTR) fk[buf][u] = tr_frag<TRS>(Kt, dt0 + u, ks, lane, hi);
          else fk[buf][u] = *reinterpret_cast<const bf16x8_t*>(Kt + (32 * (dt0 + u) + l31) * TS + 16 * ks + 8 * hi);
      };
      load_t(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        if (bi + 1 < NBAT2) load_t(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) mfma_a(acc_dq[dt0 + u], fk[bi & 1][u], sf[ks]);   // dQ^T[D x q]
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  fence_a(acc_dq);
  if (qi < s.L) {
    uint16_t* dqp = g.dq + ((int64_t)(s.start + qi) * a.H + h) * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        o.x = pack_bf16(acc_dq[dt][4 * g4 + 0], acc_dq[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_dq[dt][4 * g4 + 2], acc_dq[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(dqp + 32 * dt + 8 * g4 + 4 * hi) = o;
      }
  }
}

// pass B': dQ from the dS the dK pass left in HBM -- no S / dP recomputation (2 of the 3 GEMMs of the dQ pass) and no
// SiLU: per 32-key sub-tile a wave loads its 2 KB of dS (lane-major, as the dK pass held it: lane = key), turns it into
// its own operand layout (lane = query) with four hardware transpose reads from a wave-private LDS patch, and runs
// dQ^T[D x q] += K^T[D x keys] dS^T[keys x q].  A sub-tile the dK pass did not visit (entirely masked) counts as zero.
template <int D>
__global__ void __launch_bounds__(256, HSTU_XOCC) hstu_bwd_q_ds_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr int BK = HSTU_XSTEP, NT = BK / 32;   // 32-key sub-tiles per step
  constexpr int TRS = TrStride<D>::value;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Kt = smem;                          // [BK][TRS] row-major K tile, read transposed
  uint16_t* dSw = Kt + BK * TRS + NT * 1024 * (threadIdx.x >> 6);   // wave-private dS patch (2 KB per sub-tile)

  const BlockSeq bs = seq_head_of_block(a);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int nblk = (s.L + kBM - 1) / kBM;
  if (bs.z >= nblk || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  const int m0 = row_block_of_rank(bs.z, nblk, a, b) * kBM;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qrow0 = m0 + 32 * wv;
  const int qi = qrow0 + l31;
  const bool wave_live = qrow0 < s.L;
  int last_row = m0 + kBM - 1 < s.L - 1 ? m0 + kBM - 1 : s.L - 1;
  int n_end = s.L;
  if (a.causal) { n_end = last_row + 1; if (s.has_ctx && m0 < s.c && s.hlen > n_end) n_end = s.hlen; }
  int w_last = qrow0 + 31 < s.L - 1 ? qrow0 + 31 : s.L - 1;
  int w_end = s.L;
  if (a.causal) { w_end = w_last + 1; if (s.has_ctx && qrow0 < s.c && s.hlen > w_end) w_end = s.hlen; }
  n_end = band_key_end(a, last_row, n_end);
  w_end = band_key_end(a, w_last, w_end);
  const int n_beg = band_key_begin(a, m0, BK), w_beg = band_key_begin(a, qrow0, BK);
  // query tile of the dK pass that holds this wave's rows
  const int it_kv = (qrow0 / g.bq_kv) * g.bq_kv;

  f32x16_t acc_dq[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dq[dt][r] = 0.f;

  const uint16_t* kbase = a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head;
  RowTile<D, BK> k_rows;
  // transpose-read addresses of the dS patch (see DESIGN.md / tr_frag): lane i of a 16-lane group points at the 8-byte
  // chunk {4 consecutive queries} of key (i >> 2) of the 4-key group; the group's lane l receives keys 0..3 for ITS query
  const int il = lane & 15, g16 = (lane >> 4) & 1;
  const int tr_lane_off = ((32 * (il & 1) + 4 * hi + (il >> 2)) * 32 + (2 * g16 + ((il & 3) >> 1)) * 8) / 2;   // elements
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef short v8s_t __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;

  u32x4_t ds0[NT], ds1[NT];
  auto tile_written = [&](int n0) -> bool {   // did the dK pass visit sub-tile (keys n0.., this wave's queries)?
    if (n0 >= s.L) return false;
    return kv_visited(kv_span(a, s, (n0 / kBM) * kBM, g.bq_kv), it_kv);
  };
  auto fetch_all = [&](int n) {
    k_rows.fetch(kbase, a.k_row, n, s.L);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt = n + 32 * t;
      if (wave_live && nt < w_end && tile_written(nt) && !xch_absent(xu, nt >> 5, qrow0 >> 5)) {
        const u32x4_t* tp = reinterpret_cast<const u32x4_t*>(g.ds_ws + xch_tile(xu, nt >> 5, qrow0 >> 5)) + 2 * lane;
        ds0[t] = tp[0]; ds1[t] = tp[1];
      } else {
        ds0[t] = u32x4_t{0u, 0u, 0u, 0u}; ds1[t] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };
  if (n_end > n_beg) fetch_all(n_beg);
  for (int n0 = n_beg; n0 < n_end; n0 += BK) {
    pin_agpr(acc_dq);
    __syncthreads();
    k_rows.commit_tr(Kt, n0, s.L);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      *reinterpret_cast<u32x4_t*>(dSw + 1024 * t + 16 * lane) = ds0[t];
      *reinterpret_cast<u32x4_t*>(dSw + 1024 * t + 16 * lane + 8) = ds1[t];
    }
    pin_agpr(acc_dq);
    __syncthreads();
    if (n0 + BK < n_end) fetch_all(n0 + BK);
    pin_agpr(acc_dq);
    if (!wave_live || n0 >= w_end || n0 < w_beg) continue;
    bf16x8_t sf[2 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint16_t* pp = dSw + 1024 * t + tr_lane_off;
        const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(pp + (8 * (2 * half)) * 16));
        const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(pp + (8 * (2 * half + 1)) * 16));
        const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        sf[2 * t + half] = __builtin_bit_cast(bf16x8_t, r);
      }
    {
      constexpr int NDT = D / 32;
      constexpr int DB = HSTU_XDB < NDT ? HSTU_XDB : NDT;
      constexpr int NBAT2 = (BK / 16) * (NDT / DB);
      bf16x8_t fk[2][DB];
      auto load_t = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) fk[buf][u] = tr_frag<TRS>(Kt, dt0 + u, ks, lane, hi);
      };
      load_t(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        if (bi + 1 < NBAT2) load_t(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) mfma_a(acc_dq[dt0 + u], fk[bi & 1][u], sf[ks]);   // dQ^T[D x q]
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  fence_a(acc_dq);
  if (qi < s.L) {
    uint16_t* dqp = g.dq + ((int64_t)(s.start + qi) * a.H + h) * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        o.x = pack_bf16(acc_dq[dt][4 * g4 + 0], acc_dq[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_dq[dt][4 * g4 + 2], acc_dq[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(dqp + 32 * dt + 8 * g4 + 4 * hi) = o;
      }
  }
}

// pass A': dV from the P the dK pass left in HBM -- the key-block owner loads, per 32-query sub-tile, its 2 KB of P straight
// into the B-operand registers (the dK pass's lane = key layout IS the operand layout here) and runs
// dV^T[D x keys] += dO^T[D x q] P[q x keys] against a transposed read of the staged dO rows: one GEMM, no SiLU.
template <int D>
__global__ void __launch_bounds__(256, HSTU_XOCC) hstu_bwd_v_p_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr int TRS = TrStride<D>::value;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int BQ = HSTU_XSTEP, NT = BQ / 32;          // 32-query sub-tiles per step
  uint16_t* dOt = smem;                         // [BQ][TRS] row-major dO tile, read transposed
  const BlockSeq bs = seq_head_of_block(a);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int n0 = bs.z * kBM;
  if (n0 >= s.L || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int key0 = n0 + 32 * wv;
  const int kj = key0 + l31;
  const bool wave_live = key0 < s.L;
  const uint16_t* dobase = g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head;
  f32x16_t acc_dv[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dv[dt][r] = 0.f;
  // the query rows the dK pass visited for this key block, in steps of BQ rows (its own steps are bq_kv rows: a sub-tile
  // of a step may lie outside the visited set -- or past the sequence -- and then counts as zero)
  const KvSpan span = kv_span(a, s, n0, g.bq_kv);
  const int jump = span.jump, c_end = span.c_end;
  const int jump_s = (jump / BQ) * BQ, cend_s = ((c_end + BQ - 1) / BQ) * BQ;
  int i_lim = ((span.lim + g.bq_kv - 1) / g.bq_kv) * g.bq_kv;   // end of the dK pass' last step
  if (i_lim > s.L) i_lim = s.L;
  auto advance = [&](int i) { i += BQ; return (i >= cend_s && i < jump_s) ? jump_s : i; };
  auto visited = [&](int i) { return i < s.L && kv_visited(span, (i / g.bq_kv) * g.bq_kv); };
  int i0 = c_end > 0 ? 0 : jump_s;
  RowTile<D, BQ> do_rows;
  u32x4_t p0[NT], p1[NT];
  auto fetch_all = [&](int i) {
    do_rows.fetch(dobase, g.do_row, i, s.L);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int it = i + 32 * t;
      if (wave_live && visited(it) && !xch_absent(xu, key0 >> 5, it >> 5)) {
        const u32x4_t* tp = reinterpret_cast<const u32x4_t*>(g.p_ws + xch_tile(xu, key0 >> 5, it >> 5)) + 2 * lane;
        p0[t] = tp[0]; p1[t] = tp[1];
      } else {
        p0[t] = u32x4_t{0u, 0u, 0u, 0u}; p1[t] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };
  if (i0 < i_lim) fetch_all(i0);
  for (; i0 < i_lim; i0 = advance(i0)) {
    pin_agpr(acc_dv);
    __syncthreads();
    do_rows.commit_tr(dOt, i0, s.L);
    bf16x8_t pf[2 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { pf[2 * t] = __builtin_bit_cast(bf16x8_t, p0[t]); pf[2 * t + 1] = __builtin_bit_cast(bf16x8_t, p1[t]); }
    pin_agpr(acc_dv);
    __syncthreads();
    {
      const int nx = advance(i0);
      if (nx < i_lim) fetch_all(nx);
    }
    pin_agpr(acc_dv);
    if (!wave_live) continue;
    {
      constexpr int NDT = D / 32;
      constexpr int DB = HSTU_XDB < NDT ? HSTU_XDB : NDT;
      constexpr int NBAT2 = (BQ / 16) * (NDT / DB);
      bf16x8_t fa[2][DB];
      auto load_t = [&](int bi, int buf) {
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) fa[buf][u] = tr_frag<TRS>(dOt, dt0 + u, ks, lane, hi);
      };
      load_t(0, 0);
#pragma unroll
      for (int bi = 0; bi < NBAT2; ++bi) {
        if (bi + 1 < NBAT2) load_t(bi + 1, (bi + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = bi / (NDT / DB), dt0 = (bi % (NDT / DB)) * DB;
#pragma unroll
        for (int u = 0; u < DB; ++u) mfma_a(acc_dv[dt0 + u], fa[bi & 1][u], pf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  fence_a(acc_dv);
  if (kj < s.L) {
    uint16_t* dvp = g.dv + ((int64_t)(s.start + kj) * a.H + h) * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        o.x = pack_bf16(acc_dv[dt][4 * g4 + 0], acc_dv[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_dv[dt][4 * g4 + 2], acc_dv[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g4 + 4 * hi) = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// The one-GEMM passes of the exchange with EIGHT waves (round 4, head dim 256): one workgroup owns 256 keys (dV from P) or
// 256 query rows (dQ from dS), 32 per wave, two waves per SIMD (128 accumulator registers + < 128 others each), and streams
// the other side's rows -- dO resp. K -- in 64-row tiles that go global -> LDS by LDS-DMA (inline asm: see
// hstu_fwd_pc_kernel), double-buffered, one barrier per step, read back transposed under the forward's V swizzle.  Against
// the 4-wave kernels: the staged tile feeds twice the MFMAs (half the L2 -> LDS bytes per FLOP: those passes moved 13 B per
// clock and CU, what L2 + HBM deliver), no staging registers, no commit phase, and a second wave per SIMD to cover the P / dS
// loads and the fragment reads.
// ---------------------------------------------------------------------------------------------------
#ifndef HSTU_KVPC_MIDBAR
#define HSTU_KVPC_MIDBAR 0   // 1 = a second barrier per step: the S waves arrive after their GEMMs, the K waves after their DMA issue, so the dK GEMM runs under the elementwise phase
#endif
#ifndef HSTU_KVPC_PROBE
#define HSTU_KVPC_PROBE 0   // timing probe (wrong results): 1 = the S waves' GEMMs reuse their first fragment batches (no LDS reads)
#endif
#ifndef HSTU_KVPC_KSLEEP
#define HSTU_KVPC_KSLEEP 0   // K waves of the dK pass: s_sleep units (64 cycles each) between the DMA issue and the dK GEMM
#endif
#ifndef HSTU_KVPC_FBUF
#define HSTU_KVPC_FBUF 3   // S waves of the dK pass: Q / dO fragment batches in registers (FBUF - 1 in flight ahead of the MFMAs)
#endif
#ifndef HSTU_X8_VBUF
#define HSTU_X8_VBUF 2   // fragment batches (4 slices) in registers in the 8-wave passes (3 spills at 128 + 128 registers)
#endif
template <int NW>
struct Dma64T {   // LDS-DMA of one 64-row x 256-column bf16 tile by NW waves: 32 instructions of 2 rows, 32 / NW per wave
  static constexpr int PER = 32 / NW;
  uint32_t voff[PER];
  int j0;
  __device__ __forceinline__ void init(int wv, int lane, int64_t row_stride) {
    j0 = PER * wv;
    const int dr = lane >> 5, dp = lane & 31;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int r = 2 * (j0 + u) + dr;
      voff[u] = (uint32_t)dr * (uint32_t)row_stride * 2u + 16u * (uint32_t)(dp ^ ((r & 3) << 2));
    }
  }
  static __device__ __forceinline__ void dma16(const char* sbase_any, uint32_t vo, uint32_t lds_byte) {
    // (the base is an "s" operand: say that it is wave-uniform; the two readfirstlanes fold away where hipcc sees it itself)
    const uint64_t pa = reinterpret_cast<uintptr_t>(sbase_any);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)pa), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32));
    const char* sbase = reinterpret_cast<const char*>(((uint64_t)hi32 << 32) | lo);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(lds_byte), "s"(sbase) : "memory");
  }
  // rows row0 .. row0 + 63 of `g` (element row stride row_stride) -> the tile at `dst`; rows past L are read clamped
  __device__ __forceinline__ void issue(const uint16_t* g, int64_t row_stride, int row0, int L, uint16_t* dst, int lane) const {
    const uint32_t d0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(dst + 2 * j0 * 256));   // (an "s" operand of the DMA statement: must be provably uniform)
    if (row0 + 64 <= L) {
      const char* sb = reinterpret_cast<const char*>(g + (int64_t)(row0 + 2 * j0) * row_stride);
      const int64_t step = 2 * row_stride * 2;
#pragma unroll
      for (int u = 0; u < PER; ++u) dma16(sb + u * step, voff[u], d0 + u * 1024);
    } else {
      const uint32_t rowterm = (uint32_t)(lane >> 5) * (uint32_t)row_stride * 2u;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int r0 = row0 + 2 * (j0 + u);
        const int rc = r0 < L ? r0 : L - 1;
        const uint32_t drop = r0 + 1 < L ? 0u : 0xffffffffu;
        dma16(reinterpret_cast<const char*>(g + (int64_t)rc * row_stride), voff[u] - (rowterm & drop), d0 + u * 1024);
      }
    }
  }
};
typedef Dma64T<8> Dma64;
// A fragment [32 d x 16 rows] of a DMA-staged tile, read transposed (the forward's v_frag)
__device__ __forceinline__ bf16x8_t tr_frag_sw(const uint16_t* tile, int dt, int ks, int lane, int hi) {
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef short v8s_t __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
  const int il = lane & 15, g1 = (lane >> 4) & 1, vq = il >> 2;
  const uint16_t* p0 = tile + (16 * ks + 4 * hi + vq) * 256 + 4 * (il & 1) + 8 * (4 * (dt ^ vq) + 2 * g1 + ((il & 3) >> 1));
  const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0));
  const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p0 + 8 * 256));
  const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  return __builtin_bit_cast(bf16x8_t, r);
}
// acc^T[256 x 32] += X^T[256 x 64 rows of tile BUF of the ring] B[64 x 32]: 32 MFMAs, fragment batches of 4.  BUF is a
// compile-time constant (the step loops are unrolled by two): every fragment address is then one of eight per-lane registers
// plus an immediate -- with a run-time buffer the addresses of the other buffer were carried in registers and spilled.
template <int BUF>
__device__ __forceinline__ void gemm_x8(f32x16_t (&acc)[8], const uint16_t* ring, const bf16x8_t (&bf)[4], int lane, int hi) {
  constexpr int NB = 8, NVB = HSTU_X8_VBUF;
  const uint16_t* tile = ring + BUF * 64 * 256;
  bf16x8_t fr[NVB][4];
  auto load = [&](int bi) {
#pragma unroll
    for (int u = 0; u < 4; ++u) fr[bi % NVB][u] = tr_frag_sw(tile, 4 * (bi & 1) + u, bi >> 1, lane, hi);
  };
#pragma unroll
  for (int bi = 0; bi < NVB - 1; ++bi) load(bi);
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    if (bi + NVB - 1 < NB && !(HSTU_X8_PROBE & 1)) load(bi + NVB - 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) mfma_a(acc[4 * (bi & 1) + u], fr[(HSTU_X8_PROBE & 1) ? 0 : bi % NVB][u], bf[bi >> 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int D>
__device__ __forceinline__ void store_acc_rows(const f32x16_t (&acc)[D / 32], uint16_t* rowp, int hi) {
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 o;
      o.x = pack_bf16(acc[dt][4 * g4 + 0], acc[dt][4 * g4 + 1]);
      o.y = pack_bf16(acc[dt][4 * g4 + 2], acc[dt][4 * g4 + 3]);
      *reinterpret_cast<uint2*>(rowp + 32 * dt + 8 * g4 + 4 * hi) = o;
    }
}

// dV from the stored P, 256 keys per workgroup.  The query steps are the union of what the dK pass ran for the block's two
// 128-key halves (its blocks are kBM keys); a wave consults the span of ITS half to tell which sub-tiles exist.
template <int D, int NW, bool kFunc = false>
__device__ __forceinline__ void hstu_bwd_v_p8_body(const BwdAttnArgs& g, unsigned bz, unsigned nz) {
  constexpr int kBM8 = 32 * NW;   // keys per workgroup
  static_assert(D == 256, "DMA rows of 32 chunks");
  const AttnArgs& a = g.f;
  constexpr int BQ = 64, NT = 2, TILE = BQ * D;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [2][64][256] dO tiles
  const BlockSeq bs = seq_head_of_block(a, bz, nz);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int n0 = bs.z * kBM8;
  if (n0 >= s.L || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int key0 = n0 + 32 * wv, kj = key0 + l31;
  const bool wave_live = key0 < s.L;
  const uint16_t* dobase = g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head;
  f32x16_t acc[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  KvSpan sp0 = kv_span(a, s, n0, g.bq_kv);
  if constexpr (kFunc) { static_assert(NW == 4, "one key block of the dK pass per workgroup"); sp0 = kv_span_clip(sp0, func_kvis_of(g, s.start, b, h, n0), g.bq_kv); }
  const KvSpan sp1 = (NW == 8 && n0 + kBM < s.L) ? kv_span(a, s, n0 + kBM, g.bq_kv) : sp0;
  const KvSpan mine = wv < 4 ? sp0 : sp1;
  const int jump = sp0.jump < sp1.jump ? sp0.jump : sp1.jump, c_end = sp0.c_end > sp1.c_end ? sp0.c_end : sp1.c_end;
  const int lim = sp0.lim > sp1.lim ? sp0.lim : sp1.lim;
  const int jump_s = (jump / BQ) * BQ, cend_s = ((c_end + BQ - 1) / BQ) * BQ;
  int i_lim = ((lim + g.bq_kv - 1) / g.bq_kv) * g.bq_kv;
  if (i_lim > s.L) i_lim = s.L;
  auto advance = [&](int i) { i += BQ; return (i >= cend_s && i < jump_s) ? jump_s : i; };
  auto visited = [&](int i) { return i < s.L && kv_visited(mine, (i / g.bq_kv) * g.bq_kv); };
  Dma64T<NW> dma;
  dma.init(wv, lane, g.do_row);
  u32x4_t pn0[NT], pn1[NT];
  auto fetch_p = [&](int i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int it = i + 32 * t;
      if (!(HSTU_X8_PROBE & 2) && wave_live && visited(it) && !xch_absent(xu, key0 >> 5, it >> 5)) {
        const u32x4_t* tp = reinterpret_cast<const u32x4_t*>(g.p_ws + xch_tile(xu, key0 >> 5, it >> 5)) + 2 * lane;
        pn0[t] = xch_load(tp); pn1[t] = xch_load(tp + 1);
      } else {
        pn0[t] = u32x4_t{0u, 0u, 0u, 0u}; pn1[t] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };
  int i0 = c_end > 0 ? 0 : jump_s;
  if (i0 < i_lim) { dma.issue(dobase, g.do_row, i0, s.L, smem, lane); fetch_p(i0); }
  auto step = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    pin_agpr_2w(acc);
    bf16x8_t pf[2 * NT];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the step (and its P words) have arrived
#pragma unroll
    for (int t = 0; t < NT; ++t) { pf[2 * t] = __builtin_bit_cast(bf16x8_t, pn0[t]); pf[2 * t + 1] = __builtin_bit_cast(bf16x8_t, pn1[t]); }
    __syncthreads();                                    // everyone's have; everyone is done with the other buffer
    {
      const int nx = advance(i0);
      if (nx < i_lim) { if (!(HSTU_X8_PROBE & 4)) dma.issue(dobase, g.do_row, nx, s.L, smem + (BUF ^ 1) * TILE, lane); fetch_p(nx); }
    }
    pin_agpr_2w(acc);
    if (wave_live) gemm_x8<BUF>(acc, smem, pf, lane, hi);
    i0 = advance(i0);
  };
  while (i0 < i_lim) {
    step(std::integral_constant<int, 0>{});
    if (i0 >= i_lim) break;
    step(std::integral_constant<int, 1>{});
  }
  fence_a_2w(acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (kj < s.L) store_acc_rows<D>(acc, g.dv + ((int64_t)(s.start + kj) * a.H + h) * D, hi);
}

// dQ from the stored dS, 256 query rows per workgroup (the layout juggling of hstu_bwd_q_ds_kernel: the 2 KB sub-tile goes
// through a wave-private LDS patch and comes back through transpose reads)
template <int D, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) hstu_bwd_v_p8_kernel(BwdAttnArgs g) {
  hstu_bwd_v_p8_body<D, NW>(g, blockIdx.z, gridDim.z);
}

template <int D, int NW, bool kFunc = false>
__device__ __forceinline__ void hstu_bwd_q_ds8_body(const BwdAttnArgs& g, unsigned bz, unsigned nz) {
  constexpr int kBM8 = 32 * NW;   // query rows per workgroup
  static_assert(D == 256, "DMA rows of 32 chunks");
  const AttnArgs& a = g.f;
  constexpr int BK = 64, NT = 2, TILE = BK * D;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [2][64][256] K tiles | 8 waves x 2 x 2 KB dS patches
  const BlockSeq bs = seq_head_of_block(a, bz, nz);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int nblk = (s.L + kBM8 - 1) / kBM8;
  if (bs.z >= nblk || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  const int m0 = row_block_of_rank(bs.z, nblk, a, b) * kBM8;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint16_t* dSw = smem + 2 * TILE + NT * 1024 * wv;   // (NW patches behind the two K tiles)
  const int qrow0 = m0 + 32 * wv, qi = qrow0 + l31;
  const bool wave_live = qrow0 < s.L;
  int last_row = m0 + kBM8 - 1 < s.L - 1 ? m0 + kBM8 - 1 : s.L - 1;
  int n_end = s.L;
  if (a.causal) { n_end = last_row + 1; if (s.has_ctx && m0 < s.c && s.hlen > n_end) n_end = s.hlen; }
  int w_last = qrow0 + 31 < s.L - 1 ? qrow0 + 31 : s.L - 1;
  int w_end = s.L;
  if (a.causal) { w_end = w_last + 1; if (s.has_ctx && qrow0 < s.c && s.hlen > w_end) w_end = s.hlen; }
  n_end = band_key_end(a, last_row, n_end);
  w_end = band_key_end(a, w_last, w_end);
  int n_beg = band_key_begin(a, m0, BK), w_beg = band_key_begin(a, qrow0, BK);
  const int it_kv = (qrow0 / g.bq_kv) * g.bq_kv;   // query tile of the dK pass that holds this wave's rows
  // kFunc: lane l holds the table entry of key block l (the query rows that reach it: which sub-tiles the dK pass wrote); the key
  // loop is clipped to what the block's / the wave's row groups reach (their extents: hstu_func_kvis_kernel's second table)
  int2 vis_all = make_int2(0, 0x7fffffff);
  int gap_a = 0x7fffffff, gap_b = 0x7fffffff;     // kFunc: the key steps in [gap_a, gap_b) lie wholly between the block's prefix and its bands: not visited
  int wgap_a = 0x7fffffff, wgap_b = 0x7fffffff;   // ... and the wave's own (its MFMAs skip them)
  if constexpr (kFunc) {
    if (kBM * lane < s.L) vis_all = func_kvis_of(g, s.start, b, h, kBM * lane);
    if (g.func_gext && a.wskip) {
      FuncExt bx{0, 0x7fffffff, 0, 0x7fffffff}, wx = bx;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int row = m0 + 32 * w;
        const int64_t gi = func_gext_index(s.start, b, row);
        FuncExt e{0x7fffffff, 0, 0x7fffffff, 0};     // (no entry: everything reachable)
        if (row >= s.L) e = FuncExt{0, 0x7fffffff, 0, 0x7fffffff};
        else if (gi < 4 * g.func_kvis_h) { const int4 t = g.func_gext[(int64_t)(a.func_h ? h : 0) * 4 * g.func_kvis_h + gi]; e = FuncExt{t.y, t.z, t.w, t.x}; }
        bx = func_ext_merge(bx, e);
        if (w == wv) wx = e;
      }
      const int be = func_ext_end(bx), bb = func_ext_begin(bx), we = func_ext_end(wx), wb = func_ext_begin(wx);
      if (be < n_end) n_end = be;
      if (we < w_end) w_end = we;
      if (bb > n_beg) n_beg = bb >= n_end ? n_end : (bb / BK) * BK;
      if (wb > w_beg) w_beg = wb >= 0x7fffff00 ? 0x7fffff00 : (wb / BK) * BK;
      if (bx.lo < bx.hi && bx.lo > bx.f0) {
        const int ga = ((bx.f0 + BK - 1) / BK) * BK, gb = (bx.lo / BK) * BK;
        if (ga < gb) { gap_a = ga; gap_b = gb; }
      }
      if (wx.lo < wx.hi && wx.lo > wx.f0) {
        const int ga = ((wx.f0 + BK - 1) / BK) * BK, gb = (wx.lo / BK) * BK;
        if (ga < gb) { wgap_a = ga; wgap_b = gb; }
      }
      if (n_beg >= gap_a && n_beg < gap_b) n_beg = gap_b;
    }
  }
  auto next_step = [&](int n) { n += BK; if constexpr (kFunc) { if (n >= gap_a && n < gap_b) n = gap_b; } return n; };
  f32x16_t acc[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  const uint16_t* kbase = a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head;
  const int il = lane & 15, g16 = (lane >> 4) & 1;
  const int tr_lane_off = ((32 * (il & 1) + 4 * hi + (il >> 2)) * 32 + (2 * g16 + ((il & 3) >> 1)) * 8) / 2;   // elements
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef short v8s_t __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
  Dma64T<NW> dma;
  dma.init(wv, lane, a.k_row);
  u32x4_t ds0[NT], ds1[NT];
  auto tile_written = [&](int n0) -> bool {
    if (n0 >= s.L) return false;
    KvSpan sp = kv_span(a, s, (n0 / kBM) * kBM, g.bq_kv);
    if constexpr (kFunc) {
      const int kb = __builtin_amdgcn_readfirstlane(n0 / kBM);
      int2 vis;
      if (kb < 64) { vis.x = __builtin_amdgcn_readlane(vis_all.x, kb); vis.y = __builtin_amdgcn_readlane(vis_all.y, kb); }
      else vis = func_kvis_of(g, s.start, b, h, kb * kBM);
      sp = kv_span_clip(sp, vis, g.bq_kv);
    }
    return kv_visited(sp, it_kv);
  };
  auto fetch_ds = [&](int n) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt = n + 32 * t;
      if (!(HSTU_X8_PROBE & 2) && wave_live && nt < w_end && tile_written(nt) && !xch_absent(xu, nt >> 5, qrow0 >> 5)) {
        const u32x4_t* tp = reinterpret_cast<const u32x4_t*>(g.ds_ws + xch_tile(xu, nt >> 5, qrow0 >> 5)) + 2 * lane;
        ds0[t] = xch_load(tp); ds1[t] = xch_load(tp + 1);
      } else {
        ds0[t] = u32x4_t{0u, 0u, 0u, 0u}; ds1[t] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };
  if (n_end > n_beg) { dma.issue(kbase, a.k_row, n_beg, s.L, smem, lane); fetch_ds(n_beg); }
  int n0 = n_beg;
  auto step = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    pin_agpr_2w(acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the wave's dS sub-tiles of this step -> its private patch (read back transposed below; the previous step's reads of the
    // patch are complete: their MFMAs have been issued)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      *reinterpret_cast<u32x4_t*>(dSw + 1024 * t + 16 * lane) = ds0[t];
      *reinterpret_cast<u32x4_t*>(dSw + 1024 * t + 16 * lane + 8) = ds1[t];
    }
    __syncthreads();
    { const int nx = next_step(n0); if (nx < n_end) { if (!(HSTU_X8_PROBE & 4)) dma.issue(kbase, a.k_row, nx, s.L, smem + (BUF ^ 1) * TILE, lane); fetch_ds(nx); } }
    pin_agpr_2w(acc);
    if (wave_live && n0 < w_end && n0 >= w_beg && !(kFunc && n0 >= wgap_a && n0 < wgap_b)) {
      bf16x8_t sf[2 * NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const uint16_t* pp = dSw + 1024 * t + tr_lane_off;
          const v4s_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(pp + (8 * (2 * half)) * 16));
          const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(pp + (8 * (2 * half + 1)) * 16));
          const v8s_t r = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
          sf[2 * t + half] = __builtin_bit_cast(bf16x8_t, r);
        }
      gemm_x8<BUF>(acc, smem, sf, lane, hi);
    }
    n0 = next_step(n0);
  };
  while (n0 < n_end) {
    step(std::integral_constant<int, 0>{});
    if (n0 >= n_end) break;
    step(std::integral_constant<int, 1>{});
  }
  fence_a_2w(acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (qi < s.L) store_acc_rows<D>(acc, g.dq + ((int64_t)(s.start + qi) * a.H + h) * D, hi);
}

template <int D, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) hstu_bwd_q_ds8_kernel(BwdAttnArgs g) {
  hstu_bwd_q_ds8_body<D, NW>(g, blockIdx.z, gridDim.z);
}

// Round 6: the two one-GEMM passes in ONE launch -- they are independent (dV reads P, dQ reads dS, both written by the dK pass) and
// at C3's 512 rows each is a single generation of blocks whose launch ramp and tail the other can fill: the first half of the
// block ranks takes the dV role, the second the dQ role (LDS = the larger of the two: 80 KB, two workgroups per CU as before).
template <int D, int NW, bool kFunc = false>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) hstu_bwd_vq8_kernel(BwdAttnArgs g) {
  const unsigned nz = gridDim.z >> 1;
  if (blockIdx.z < nz) hstu_bwd_v_p8_body<D, NW, kFunc>(g, blockIdx.z, nz);
  else hstu_bwd_q_ds8_body<D, NW, kFunc>(g, blockIdx.z - nz, nz);
}

// ---------------------------------------------------------------------------------------------------
// What bounds the one-GEMM dV / dQ passes at long sequences (probes of profiles/r04_hstu_bwd_x8_probes.txt, 8 x 4096, one chunk):
// hstu_bwd_v_p8_kernel 260 us -- 158 without its P loads, 129 without loads and row DMA, 111 without the LDS fragment reads as well
// (HSTU_X8_PROBE).  The 0.54 GB of P (and of dS) that a pass reads were written by the dK pass and no cache holds them: at the
// ~5.4 TB/s HBM delivers they are 100 us per pass, and 200 us of writes inside the dK pass -- 0.4 of the backward's 1.02 ms.
// It is bandwidth, not latency: the same passes with the exchange tiles brought in by LDS-DMA TWO steps ahead (one 8-wave
// workgroup per CU, a three-slot ring of 96 KB next to the row ring, counted vmcnt, bit-identical results) ran 635-642 TFLOP/s
// against 662-667 (profiles/r04_hstu_bwd_xe_ab.txt) and were removed again.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// The dK pass of the exchange with two waves per SIMD (round 4, head dim 256, no bias): the forward's S-wave / O-wave split
// applied to the backward.  One workgroup = 128 keys; waves 0-3 ("S waves") hold the K and V fragments of their 32 keys
// (128 registers) and compute, per 32-query step, S = Q K^T and dP = dO V^T (32 MFMAs), the SiLU / SiLU' elementwise phase,
// and write P and dS to the exchange buffer (for the one-GEMM dV / dQ passes) and dS to an LDS hand-off; waves 4-7 ("K waves")
// hold the dK accumulator (128 AGPRs), read that dS one step later and run dK^T += Q^T dS (16 MFMAs, Q^T fragments by
// transpose reads), and issue the LDS-DMA of the step's three images: Q rows and dO rows (K-style swizzle, b128 row reads of
// the S waves) and Q rows once more under the V-style swizzle (transpose reads of the K waves).  One barrier per step,
// every ring two deep; the step loop is unrolled by two so that every LDS address is a register plus an immediate.
// Same MFMA order and roundings as hstu_bwd_kv_kernel<256, 32, 2>: bit-identical dK, P and dS.
// Measured and rejected: the S waves in two types -- wave 2 t holds the K fragments of BOTH key tiles of pair t and computes S for
// 64 keys, wave 2 t + 1 the V fragments and dP, each hands the other tile's accumulator over through LDS behind a second barrier
// (every Q / dO fragment read then feeds two MFMAs: 64 KB of LDS reads per step instead of 128).  Same time (8 x 4096 backward
// 641-644 against 644-646 TFLOP/s, profiles/r04_hstu_bwd_split_ab.txt): the stamps put the S waves' GEMM phase at 1 690 clocks
// either way -- it is not the fragment reads that pace those 32 MFMAs.
// ---------------------------------------------------------------------------------------------------
// kFunc (round 6): `func` masks of up to two bands (n_func <= 5) on the exchange backward.  The K waves stage, one step ahead and
// next to the step's images, the bounds of its 32 query rows and the extents of that row group ([2][6][32] ints of LDS); an S wave
// whose 32 keys lie below every row's prefix takes the plain paths, otherwise the general rule and the rows' functions per element.
// The steps follow kv_span_clip -- the one-GEMM passes replay the same span and find exactly the sub-tiles written here.
template <int D, bool kFunc = false>
__global__ void __launch_bounds__(512) hstu_bwd_kv_pc_kernel(BwdAttnArgs g) {
  static_assert(D == 256, "DMA rows of 32 chunks");
  const AttnArgs& a = g.f;
  constexpr int BQ = 32, IMG = BQ * D;               // elements of one staged image (16 KB)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* const Qr = smem;                         // [2][32][256] Q rows, K-style swizzle
  uint16_t* const dOr = smem + 2 * IMG;              // [2][32][256] dO rows, K-style swizzle
  uint16_t* const Qt = smem + 4 * IMG;               // [2][32][256] Q rows, V-style swizzle (read transposed)
  uint16_t* const Hs = smem + 6 * IMG;               // [2][4 pairs][2 slices][64 lanes] x 16 B: dS hand-off
  int* const Fs = reinterpret_cast<int*>(smem + 6 * IMG + 2 * 4 * 2 * 64 * 8);   // kFunc: [2][6][32] row bounds (5) + group extents

  const BlockSeq bs = seq_head_of_block(a);
  const int b = bs.b, h = bs.h;
  SeqInfo s;
  s.start = bs.start;
  s.L = bs.end - s.start;
  const int n0 = bs.z * kBM;
  if (n0 >= s.L || xch_other_chunk(g, b, h)) return;
  const XchUnit xu = xch_unit(g, b, h, s.L);
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  s.wl = a.wl; s.wr = a.wr;
  const int lane = lane_id(), hi = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wv >> 2, pw = wv & 3;
  const int key0 = n0 + 32 * pw, kj = key0 + l31;
  const bool wave_live = key0 < s.L;
  const bool plain = !s.has_ctx && !s.has_tgt && s.wl < 0 && s.wr < 0;
  const bool key_in = kj < s.L, key_hist = kj < s.hlen;
  const int key_iend = (s.has_tgt && kj >= s.hlen) ? s.hlen + ((kj - s.hlen) / a.group + 1) * a.group : 0x7fffffff;
  const int ctx_end = s.has_ctx ? s.c : 0;
  const uint16_t* qbase = a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head;
  const uint16_t* dobase = g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head;
  const float neg_alpha_log2e = -a.alpha * 1.4426950408889634f;
  const float c_p = a.alpha * a.inv_scale, c_ds = a.alpha * a.inv_scale;
  KvSpan span = kv_span(a, s, n0, BQ);
  if constexpr (kFunc) span = kv_span_clip(span, func_kvis_of(g, s.start, b, h, n0), BQ);
  const int jump = span.jump, c_end = span.c_end, i_lim = span.lim;
  auto advance = [&](int i) { i += BQ; return (i >= c_end && i < jump) ? jump : i; };
  const int first = c_end > 0 ? 0 : jump;
#if HSTU_TIMING
  unsigned tsum[7] = {0, 0, 0, 0, 0, 0, 0};   // S: barrier, -, -, -, GEMM S + dP, elementwise + stores, steps | K: vmcnt, barrier, DMA issue, -, -, GEMM dK, steps
  const unsigned t_start = tick();
  auto t_dump = [&]() {
    const unsigned t_end = tick();
    if (lane == 0) {
      const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long* d = g_hstu_dbg + ((size_t)(blk * 8 + wv) % 65536) * 8;
      for (int i = 0; i < 7; ++i) d[i] = tsum[i];
      d[6] |= (unsigned long long)role << 32;
      d[7] = t_end - t_start;
    }
  };
#endif

  // ---- LDS-DMA of a 32-row image: 16 instructions of 2 rows, 4 per K wave (rows 8 pw .. 8 pw + 7)
  const int dr = lane >> 5, dp = lane & 31;
  uint32_t vq_k[4], vdo_k[4], vq_v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 2 * (4 * pw + u) + dr;
    vq_k[u] = (uint32_t)dr * (uint32_t)a.q_row * 2u + 16u * (uint32_t)(dp ^ (r & 15));
    vdo_k[u] = (uint32_t)dr * (uint32_t)g.do_row * 2u + 16u * (uint32_t)(dp ^ (r & 15));
    vq_v[u] = (uint32_t)dr * (uint32_t)a.q_row * 2u + 16u * (uint32_t)(dp ^ ((r & 3) << 2));
  }
  auto issue_img = [&](const uint16_t* gsrc, int64_t row_stride, const uint32_t (&voff)[4], int row0, uint16_t* dst) {
    const uint32_t d0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)(dst + 8 * pw * D));
    if (row0 + BQ <= s.L) {
      const char* sb = reinterpret_cast<const char*>(gsrc + (int64_t)(row0 + 8 * pw) * row_stride);
      const int64_t step = 2 * row_stride * 2;
#pragma unroll
      for (int u = 0; u < 4; ++u) Dma64::dma16(sb + u * step, voff[u], d0 + u * 1024);
    } else {
      const uint32_t rowterm = (uint32_t)dr * (uint32_t)row_stride * 2u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r0 = row0 + 2 * (4 * pw + u);
        const int rc = r0 < s.L ? r0 : s.L - 1;
        const uint32_t drop = r0 + 1 < s.L ? 0u : 0xffffffffu;
        Dma64::dma16(reinterpret_cast<const char*>(gsrc + (int64_t)rc * row_stride), voff[u] - (rowterm & drop), d0 + u * 1024);
      }
    }
  };

  if (role == 0) {
    // =========================== S waves ===========================
    bf16x8_t kf[D / 16], vf[D / 16];
    load_own_frags<D>(kf, a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head, a.k_row, kj, s.L, hi);
    load_own_frags<D>(vf, a.v + (int64_t)s.start * a.v_row + (int64_t)h * a.v_head, a.v_row, kj, s.L, hi);
    const int kx = l31 & 15;
    int cur = first, prev_valid = 0;
    auto step = [&](auto parc) {
      constexpr int PAR = decltype(parc)::value;
      TICK(t0);
      __syncthreads();     // (no vmcnt wait here: the S waves issue no DMA, and their exchange stores must not be drained per step)
      TICK(t1);
      TACC(0, t0, t1);
      const int i0 = cur;
      const bool have = i0 < i_lim;
      prev_valid = have;
      cur = have ? advance(i0) : i0;
      if (!have || !wave_live) { if (HSTU_KVPC_MIDBAR) __builtin_amdgcn_s_barrier(); return; }
      const uint16_t* Qs = Qr + PAR * IMG;
      const uint16_t* Ds = dOr + PAR * IMG;
      f32x16_t acc_s, acc_p;
      {
        constexpr int SLB = 2, NBAT = (D / 16) / SLB, NFB = HSTU_KVPC_FBUF;   // batches of 2 slices (4 MFMAs), NFB - 1 in flight
        bf16x8_t qa[NFB][SLB], da[NFB][SLB];
        auto load_b = [&](int bi) {
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            const int sl = SLB * bi + u;
            const int off = l31 * D + 8 * ((2 * sl + hi) ^ kx);
            qa[bi % NFB][u] = *reinterpret_cast<const bf16x8_t*>(Qs + off);
            da[bi % NFB][u] = *reinterpret_cast<const bf16x8_t*>(Ds + off);
          }
        };
#pragma unroll
        for (int bi = 0; bi < NFB - 1; ++bi) load_b(bi);
#pragma unroll
        for (int bi = 0; bi < NBAT; ++bi) {
          if (bi + NFB - 1 < NBAT && !(HSTU_KVPC_PROBE & 1)) load_b(bi + NFB - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < SLB; ++u) {
            const int sl = SLB * bi + u;
            if (sl == 0) { mfma_v0(acc_s, qa[bi % NFB][u], kf[sl]); mfma_v0(acc_p, da[bi % NFB][u], vf[sl]); }
            else { mfma_v(acc_s, qa[bi % NFB][u], kf[sl]); mfma_v(acc_p, da[bi % NFB][u], vf[sl]); }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (HSTU_KVPC_MIDBAR) __builtin_amdgcn_s_barrier();   // (no LDS hazard to order: a pure phase alignment)
      TICK(t2);
      TACC(4, t1, t2);
      const bool tail = i0 + BQ > s.L;     // rows past the sequence are clamped copies here (the 4-wave kernel stages zeros): masked
      auto elementwise = [&](auto modec, auto tailc) {
        constexpr int kMask = decltype(modec)::value;
        constexpr bool kTail = decltype(tailc)::value;
        uint32_t pk[8], sk[8];
        int fb[5][4];      // kMask 5: bounds of the four rows 8 (r >> 2) + 4 hi + 0 .. 3, reloaded per group of four registers
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p2[2], s2[2];
          if constexpr (kMask == 5) {
            if ((r & 3) == 0) {
              const int* fs = Fs + PAR * 192 + 4 * hi + 8 * (r >> 2);
#pragma unroll
              for (int pb = 0; pb < 5; ++pb) {
                const int4 t4 = *reinterpret_cast<const int4*>(fs + 32 * pb);
                fb[pb][0] = t4.x; fb[pb][1] = t4.y; fb[pb][2] = t4.z; fb[pb][3] = t4.w;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rr = r + u;
            const int qi = i0 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            bool ok = true;
            if constexpr (kMask == 1) ok = kj <= qi;
            else if constexpr (kMask == 2) ok = kj < s.L;
            else if constexpr (kMask == 3) ok = (qi < ctx_end ? key_hist : ((kj <= qi) & key_in)) & (key_hist | (qi < key_iend));
            else if constexpr (kMask == 4)
              ok = key_in & ((s.wl < 0) | (kj >= qi - s.wl)) & (a.causal ? (kj <= qi) : ((s.wr < 0) | (kj <= qi + s.wr)));
            else if constexpr (kMask == 5) {     // every rule at once, and the row's functions
              const bool c1 = a.causal ? (kj <= qi) : ((s.wr < 0) | (kj <= qi + s.wr));
              const bool c2 = (s.wl < 0) | (kj >= qi - s.wl);
              ok = key_in & (qi < ctx_end ? key_hist : (c1 & c2)) & (key_hist | (qi < key_iend));
              const int e = rr & 3;
              ok = ok & ((kj < fb[0][e]) | ((fb[1][e] <= kj) & (kj < fb[2][e])) | ((fb[3][e] <= kj) & (kj < fb[4][e])));
            }
            if constexpr (kTail) ok = ok & (qi < s.L);
            const float acc = acc_s[rr];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc * neg_alpha_log2e));
            p2[u] = ok ? acc * c_p * sg : 0.f;
            const float dsv = ds_value(acc, acc_p[rr], sg, a.alpha, c_ds);
            s2[u] = ok ? dsv : 0.f;
          }
          pk[r >> 1] = pack_bf16(p2[0], p2[1]);
          sk[r >> 1] = pack_bf16(s2[0], s2[1]);
        }
        const u32x4_t y0 = {sk[0], sk[1], sk[2], sk[3]}, y1 = {sk[4], sk[5], sk[6], sk[7]};
        u32x4_t* hp = reinterpret_cast<u32x4_t*>(Hs) + ((PAR * 4 + pw) * 2) * 64 + lane;
        hp[0] = y0; hp[64] = y1;                                  // dS -> the K wave of the pair (next step)
        if (i0 < s.L && !xch_absent(xu, key0 >> 5, i0 >> 5)) {    // P and dS -> the one-GEMM dV / dQ passes
          u32x4_t* tp = reinterpret_cast<u32x4_t*>(g.p_ws + xch_tile(xu, key0 >> 5, i0 >> 5)) + 2 * lane;
          xch_store(tp, u32x4_t{pk[0], pk[1], pk[2], pk[3]}); xch_store(tp + 1, u32x4_t{pk[4], pk[5], pk[6], pk[7]});
        }   // (dS goes to the exchange buffer from the K wave, which has it in registers one step later: two stores fewer here)
      };
      auto ew = [&](auto modec) { if (tail) elementwise(modec, std::true_type{}); else elementwise(modec, std::false_type{}); };
      bool func_test = false;
      if constexpr (kFunc) func_test = !(key0 + 32 <= Fs[PAR * 192 + 160]);    // (the group's smallest prefix: below it every row sees every key)
      if (func_test) ew(std::integral_constant<int, 5>{});
      else if (!plain) {
        if (s.wl >= 0 || s.wr >= 0) ew(std::integral_constant<int, 4>{}); else ew(std::integral_constant<int, 3>{});
      } else if (a.causal) {
        if (key0 + 31 <= i0) ew(std::integral_constant<int, 0>{}); else ew(std::integral_constant<int, 1>{});
      } else {
        if (key0 + 31 < s.L) ew(std::integral_constant<int, 0>{}); else ew(std::integral_constant<int, 2>{});
      }
      TICK(t3);
      TACC(5, t2, t3);
#if HSTU_TIMING
      tsum[6] += 1;
#endif
    };
    while (cur < i_lim || prev_valid) {
      step(std::integral_constant<int, 0>{});
      if (!(cur < i_lim || prev_valid)) break;
      step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if HSTU_TIMING
    t_dump();
#endif
    return;
  }
  // =========================== K waves ===========================
  f32x16_t acc_dk[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dk[dt][r] = 0.f;
  // kFunc: thread (pw, lane) of the K waves fetches ONE word of a step's bounds: p = (64 pw + lane) / 32 (0 .. 4: the row's bounds,
  // 5: the extents of the row group, lanes 0 .. 3), r = lane & 31 -- written to LDS behind the next step's vmcnt wait
  int fword = 0;
  auto fetch_func = [&](int i) {
    if constexpr (kFunc) {
      const int pb = 2 * pw + (lane >> 5), r = lane & 31, qi = i + r;
      int v = 0;
      if (pb < 5) {
        if (qi < s.L && pb < a.n_func) v = a.func[(int64_t)h * a.func_h + (int64_t)pb * a.func_p + s.start + qi];
        if (pb == 0 && s.has_ctx && qi < s.c && s.hlen > v) v = s.hlen;      // a contextual row sees the whole history
      } else if (pb == 5) {
        v = r == 0 ? 0 : (r == 1 ? 0x7fffffff : (r == 2 ? 0 : 0x7fffffff));   // (no table: never "all visible")
        const int64_t gi = func_gext_index(s.start, b, i);
        if (g.func_gext && gi < 4 * g.func_kvis_h && r < 4)
          v = reinterpret_cast<const int*>(g.func_gext + (int64_t)(a.func_h ? h : 0) * 4 * g.func_kvis_h + gi)[r];
      }
      fword = v;
    }
  };
  auto commit_func = [&](int par) {
    if constexpr (kFunc) { const int pb = 2 * pw + (lane >> 5); if (pb < 6) Fs[par * 192 + 32 * pb + (lane & 31)] = fword; }
  };
  if (first < i_lim) {
    issue_img(qbase, a.q_row, vq_k, first, Qr);
    issue_img(dobase, g.do_row, vdo_k, first, dOr);
    fetch_func(first);
  }
  int cur = first, prev_valid = 0, prev_i = 0;
  auto step = [&](auto parc) {
    constexpr int PAR = decltype(parc)::value;
    pin_agpr_2w(acc_dk);
    TICK(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    commit_func(PAR);                // (the bounds of THIS step's rows, fetched one step ago)
    TICK(t1);
    __syncthreads();
    TICK(t2);
    TACC(0, t0, t1); TACC(1, t1, t2);
    const int i0 = cur;
    const bool have = i0 < i_lim, had = prev_valid != 0;
    const int ip = prev_i;           // the step the S wave finished last
    prev_i = i0;
    const int nxt = have ? advance(i0) : i0;
    if (have) {
      if (nxt < i_lim) {
        issue_img(qbase, a.q_row, vq_k, nxt, Qr + (PAR ^ 1) * IMG);
        issue_img(dobase, g.do_row, vdo_k, nxt, dOr + (PAR ^ 1) * IMG);
        fetch_func(nxt);
      }
      issue_img(qbase, a.q_row, vq_v, i0, Qt + PAR * IMG);      // read by the K waves in the NEXT step
    }
    prev_valid = have;
    cur = nxt;
    pin_agpr_2w(acc_dk);
    if (HSTU_KVPC_MIDBAR) __builtin_amdgcn_s_barrier();
    TICK(t3);
    TACC(2, t2, t3);
    if (!had || !wave_live) return;
    // The S waves open a step with 32 MFMAs and close it with ~250 VALU instructions; the K waves' 16 MFMAs belong under the
    // second half.  Issued straight behind the DMA they fall into the S waves' GEMM (both streams then take turns on the
    // SIMD's matrix pipe at ~64 cycles per MFMA and the pipe idles through the elementwise phase): park first.
    if (HSTU_KVPC_KSLEEP) __builtin_amdgcn_s_sleep(HSTU_KVPC_KSLEEP);
    // dK^T[256 x 32 keys] += Q^T[256 x 32 q] dS[32 q x 32 keys] of the PREVIOUS step (rings slot PAR ^ 1)
    const u32x4_t* hp = reinterpret_cast<const u32x4_t*>(Hs) + (((PAR ^ 1) * 4 + pw) * 2) * 64 + lane;
    bf16x8_t sf[2];
    const u32x4_t y0 = hp[0], y1 = hp[64];
    sf[0] = __builtin_bit_cast(bf16x8_t, y0);
    sf[1] = __builtin_bit_cast(bf16x8_t, y1);
    if (ip < s.L && !xch_absent(xu, key0 >> 5, ip >> 5)) {   // dS of that step -> the one-GEMM dQ pass
      u32x4_t* tq = reinterpret_cast<u32x4_t*>(g.ds_ws + xch_tile(xu, key0 >> 5, ip >> 5)) + 2 * lane;
      xch_store(tq, y0); xch_store(tq + 1, y1);
    }
    const uint16_t* tile = Qt + (PAR ^ 1) * IMG;
    constexpr int NB = 4;
    bf16x8_t fr[2][4];
    auto load = [&](int bi) {
#pragma unroll
      for (int u = 0; u < 4; ++u) fr[bi & 1][u] = tr_frag_sw(tile, 4 * (bi & 1) + u, bi >> 1, lane, hi);
    };
    load(0);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi + 1 < NB) load(bi + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) mfma_a(acc_dk[4 * (bi & 1) + u], fr[bi & 1][u], sf[bi >> 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    TICK(t4);
    TACC(5, t3, t4);
#if HSTU_TIMING
    tsum[6] += 1;
#endif
  };
  while (cur < i_lim || prev_valid) {
    step(std::integral_constant<int, 0>{});
    if (!(cur < i_lim || prev_valid)) break;
    step(std::integral_constant<int, 1>{});
  }
  fence_a_2w(acc_dk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if HSTU_TIMING
  t_dump();
#endif
  if (kj < s.L) store_acc_rows<D>(acc_dk, g.dk + ((int64_t)(s.start + kj) * a.H + h) * D, hi);
}

template <int D, int BQ, int MODE, bool kPre, bool kXP = false, bool kRab = false>
static void launch_bwd_kv(const BwdAttnArgs& g, dim3 grid, hipStream_t stream) {
  constexpr bool kDV = MODE != 2, kDK = MODE != 1;
  const size_t timg = HSTU_BWD_TR ? (size_t)BQ * (D == 32 ? 32 : D + 32) : (size_t)D * (BQ + 8);
  const size_t smem = (size_t)(BQ * (D + 8) + (kDK ? BQ * (D + 8) + timg : 0) + (kDV ? timg : 0)) * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_kv_kernel<D, BQ, MODE, kPre, kXP, kRab>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((hstu_bwd_kv_kernel<D, BQ, MODE, kPre, kXP, kRab>), grid, dim3(256), smem, stream, g);
}
template <int D, int BK, bool kPre, bool kRab = false>
static void launch_bwd_q(const BwdAttnArgs& g, dim3 grid, hipStream_t stream) {
  const size_t smem_q = (size_t)(2 * BK * (D + 8) + (HSTU_BWD_TR ? (size_t)BK * (D == 32 ? 32 : D + 32) : (size_t)D * (BK + 8))) * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_q_kernel<D, BK, kPre, kRab>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q);
    attr_set = true;
  }
  hipLaunchKernelGGL((hstu_bwd_q_kernel<D, BK, kPre, kRab>), grid, dim3(256), smem_q, stream, g);
}

template <int D>
static void launch_bwd_v_p(const BwdAttnArgs& g, dim3 grid, hipStream_t stream) {
  const size_t smem = (size_t)(HSTU_XSTEP * TrStride<D>::value) * sizeof(uint16_t);
  hipLaunchKernelGGL((hstu_bwd_v_p_kernel<D>), grid, dim3(256), smem, stream, g);
}

template <int D>
static void launch_bwd_q_ds(const BwdAttnArgs& g, dim3 grid, hipStream_t stream) {
  const size_t smem = (size_t)(HSTU_XSTEP * TrStride<D>::value + 4 * (HSTU_XSTEP / 32) * 1024) * sizeof(uint16_t);
  hipLaunchKernelGGL((hstu_bwd_q_ds_kernel<D>), grid, dim3(256), smem, stream, g);
}

// the DMA-staged one-GEMM passes (head dim 256): MI355_HSTU_X8 = 8: one 8-wave workgroup of 256 rows per CU; 4 (default):
// 4-wave workgroups of 128 rows, TWO per CU -- the same two waves per SIMD, twice the blocks (causal work balances over the
// CUs: a chunk of the capped exchange may hold only one 256-row block per CU); 0: the register-staged 4-wave kernels
template <int NW, bool kFunc = false>
static void launch_bwd_x8(const BwdAttnArgs& g, int B, int max_seqlen, hipStream_t stream) {
  const size_t smem_v = (size_t)2 * 64 * 256 * sizeof(uint16_t), smem_q = smem_v + (size_t)NW * 2 * 1024 * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_v_p8_kernel<256, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_v);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_q_ds8_kernel<256, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q);
    attr_set = true;
  }
  dim3 grid(g.f.H, B, (max_seqlen + 32 * NW - 1) / (32 * NW));
#ifndef HSTU_VQ_MERGED
#define HSTU_VQ_MERGED 1
#endif
  if ((HSTU_VQ_MERGED || kFunc) && NW == 4) {
    static bool attr_m = false;
    if (!attr_m) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_vq8_kernel<256, NW, kFunc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q);
      attr_m = true;
    }
    hipLaunchKernelGGL((hstu_bwd_vq8_kernel<256, NW, kFunc>), dim3(grid.x, grid.y, 2 * grid.z), dim3(64 * NW), smem_q, stream, g);
    return;
  }
  hipLaunchKernelGGL((hstu_bwd_v_p8_kernel<256, NW>), grid, dim3(64 * NW), smem_v, stream, g);
  hipLaunchKernelGGL((hstu_bwd_q_ds8_kernel<256, NW>), grid, dim3(64 * NW), smem_q, stream, g);
}

template <int D>
static int launch_bwd(BwdAttnArgs g, int B, int max_seqlen, hipStream_t stream) {
  dim3 grid(g.f.H, B, (max_seqlen + kBM - 1) / kBM);   // block rank slowest: see launch_fwd
  // (round 6) mask functions of up to two bands take the exchange backward at head dim 256 when the caller brought the scratch
  const bool func_x = D == 256 && g.f.func && !g.f.rab && g.f.n_func <= 5 && g.p_ws && g.ds_ws;
  if constexpr (D == 256) {
    if (func_x) {
      g.bq_kv = 32;
      const size_t smem_pc = (size_t)(6 * 32 * 256 + 2 * 4 * 2 * 64 * 8) * sizeof(uint16_t) + 2 * 192 * sizeof(int);
      static bool attr_pcf = false;
      if (!attr_pcf) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_kv_pc_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_pc);
        attr_pcf = true;
      }
      hipLaunchKernelGGL((hstu_bwd_kv_pc_kernel<256, true>), grid, dim3(512), smem_pc, stream, g);
      launch_bwd_x8<4, true>(g, B, max_seqlen, stream);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    }
  }
  if (g.f.rab || g.f.func) {   // attention bias / mask functions: the recomputing passes (S needs the bias in every pass), dS doubles as d rab
    g.ds_ws = g.p_ws = nullptr;
    g.bq_kv = 64;
    if constexpr (D >= 128) {
      launch_bwd_kv<D, 64, 1, false, false, true>(g, grid, stream);
      launch_bwd_kv<D, 32, 2, false, false, true>(g, grid, stream);
      launch_bwd_q<D, D >= 256 ? 32 : 64, false, true>(g, grid, stream);
    } else {
      launch_bwd_kv<D, 64, 0, false, false, true>(g, grid, stream);
      launch_bwd_q<D, 64, false, true>(g, grid, stream);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  if constexpr (D >= 256) {
    constexpr int var = 1;   // (the variant bits below were sweep switches of rounds 3-4; 1 is the measured best)
    g.bq_kv = (var & 1) ? 64 : 32;
    if (g.p_ws) {          // dK pass first (it writes P and dS), then the two one-GEMM passes
      // 32-row steps with the next step's Q / dO rows prefetched into registers (64-row steps leave no registers for it and
      // fetch synchronously): 0.153 -> 0.149 ms at C3, 1.33 -> 1.29 ms at L = 4096 once the elementwise phase was fixed
      if (var & 32) { g.bq_kv = 64; launch_bwd_kv<D, 64, 2, false, true>(g, grid, stream); }
      else {
        g.bq_kv = 32;
        constexpr int kvpc = 1;
        if (kvpc) {   // the S-wave / K-wave dK pass (two waves per SIMD)
          const size_t smem_pc = (size_t)(6 * 32 * 256 + 2 * 4 * 2 * 64 * 8) * sizeof(uint16_t);
          static bool attr_pc = false;
          if (!attr_pc) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_kv_pc_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_pc);
            attr_pc = true;
          }
          hipLaunchKernelGGL((hstu_bwd_kv_pc_kernel<256>), grid, dim3(512), smem_pc, stream, g);
        } else launch_bwd_kv<D, 32, 2, true, true>(g, grid, stream);
      }
      constexpr int x8 = 4;
      if (x8 == 8) launch_bwd_x8<8>(g, B, max_seqlen, stream);
      else if (x8) launch_bwd_x8<4>(g, B, max_seqlen, stream);
      else {
        launch_bwd_v_p<D>(g, grid, stream);
        launch_bwd_q_ds<D>(g, grid, stream);
      }
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    }
    if (var & 4) launch_bwd_kv<D, 64, 1, false>(g, grid, stream); else launch_bwd_kv<D, 64, 1, true>(g, grid, stream);
    if (var & 1) launch_bwd_kv<D, 64, 2, false>(g, grid, stream);
    else if (var & 8) launch_bwd_kv<D, 32, 2, false>(g, grid, stream);
    else launch_bwd_kv<D, 32, 2, true>(g, grid, stream);
    if (g.ds_ws) launch_bwd_q_ds<D>(g, grid, stream);
    else if (var & 2) launch_bwd_q<D, 64, false>(g, grid, stream);
    else if (var & 16) launch_bwd_q<D, 32, false>(g, grid, stream);
    else launch_bwd_q<D, 32, true>(g, grid, stream);
  } else if constexpr (D >= 128) {
    g.bq_kv = 32;
    if (g.p_ws) {
      launch_bwd_kv<D, 32, 2, false, true>(g, grid, stream);
      launch_bwd_v_p<D>(g, grid, stream);
      launch_bwd_q_ds<D>(g, grid, stream);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    }
    launch_bwd_kv<D, 64, 1, false>(g, grid, stream);
    launch_bwd_kv<D, 32, 2, false>(g, grid, stream);
    if (g.ds_ws) launch_bwd_q_ds<D>(g, grid, stream);
    else launch_bwd_q<D, 64, false>(g, grid, stream);
  } else {
    g.bq_kv = 64;
    g.p_ws = nullptr;          // (dV and dK come out of ONE pass at these head dims: only dS is handed on)
    launch_bwd_kv<D, 64, 0, false>(g, grid, stream);
    if (g.ds_ws) launch_bwd_q_ds<D>(g, grid, stream);
    else launch_bwd_q<D, 64, false>(g, grid, stream);
  }
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// the two-waves-per-SIMD forward (default at head dim 256; MI355_HSTU_PC=0 = the one-kind register-staged kernel)
template <int D>
static int launch_fwd_pc(const AttnArgs& a, int B, int max_seqlen, hipStream_t stream, bool dense_batch) {
  const size_t smem = (size_t)(4 * kBN * D + 2 * 4 * 4 * 64 * 8) * sizeof(uint16_t);   // K ring + V ring + P ring = 160 KB
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_pc_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_pc_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess) return MI355_ELAUNCH;
    attr_set = true;
  }
  // row blocks in (heavy, light) pairs per workgroup: dense batches only (every sequence max_seqlen rows: the caller said so with
  // mi355_hstu_attn_fwd_hint_tokens).  On a jagged batch the pairs of a long column are as heavy as before but half as many
  // workgroups share the machine and the tail grows (C4 shape 340 -> 355 us); MI355_HSTU_PAIR = 0 never, 2 always (A/B).
  // MI355_HSTU_FWD (a TEST hook, read once): 0 / unset = the rules below; 1 = 64-row waves at every length, 2 = ... in pairs on every
  // batch, 3 = 32-row waves at every length, 4 = ... in pairs on every batch (5 = the one-kind kernel: see mi355_hstu_attn_fwd_kv)
  static const int fwd_hook = getenv("MI355_HSTU_FWD") ? atoi(getenv("MI355_HSTU_FWD")) : 0;
  const int pair = (fwd_hook == 2 || fwd_hook == 4) ? 2 : 1;
  // 64 query rows per wave (two MFMAs per LDS fragment) from 1 025 rows: +4-6 % at L >= 2048, +1-4 % on jagged Zipf-to-4096 batches,
  // level at 768-1024, -2 % at C3 and -6 % at L = 256 (fewer, larger units per short column); 2 = always, 0 = never (A/B)
  const int q2 = (fwd_hook == 1 || fwd_hook == 2) ? 2 : ((fwd_hook == 3 || fwd_hook == 4) ? 0 : 1);
  if (q2 == 2 || (q2 == 1 && max_seqlen > 1024) || a.func) {
    static bool attr_q2 = false;
    if (!attr_q2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return MI355_ELAUNCH;
      attr_q2 = true;
    }
    const int nblk = (max_seqlen + kBM - 1) / kBM;
    const bool win = a.wl >= 0 || a.wr >= 0;
    if (a.func) {     // mask functions: the window-capable variants + the functions
      static bool attr_fn = false;
      if (!attr_fn) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_q2_kernel<D, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
          return MI355_ELAUNCH;
        attr_fn = true;
      }
      if (pair == 2 || (pair == 1 && dense_batch)) hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, true, true, true>), dim3(a.H, B, (nblk + 1) / 2), dim3(512), smem, stream, a);
      else hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, true, false, true>), dim3(a.H, B, nblk), dim3(512), smem, stream, a);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    }
    if (pair == 2 || (pair == 1 && dense_batch)) {
      dim3 grid(a.H, B, (nblk + 1) / 2);
      if (win) hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, true, true>), grid, dim3(512), smem, stream, a);
      else hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, false, true>), grid, dim3(512), smem, stream, a);
    } else {
      dim3 grid(a.H, B, nblk);
      if (win) hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, true, false>), grid, dim3(512), smem, stream, a);
      else hipLaunchKernelGGL((hstu_fwd_q2_kernel<D, false, false>), grid, dim3(512), smem, stream, a);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  if (pair == 2 || (pair == 1 && dense_batch)) {
    static bool attr_pair = false;
    if (!attr_pair) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_pair_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_pair_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess) return MI355_ELAUNCH;
      attr_pair = true;
    }
    const int nblk = (max_seqlen + kBM - 1) / kBM;
    dim3 grid(a.H, B, (nblk + 1) / 2);
    if (a.wl >= 0 || a.wr >= 0) hipLaunchKernelGGL((hstu_fwd_pair_kernel<D, true>), grid, dim3(512), smem, stream, a);
    else hipLaunchKernelGGL((hstu_fwd_pair_kernel<D, false>), grid, dim3(512), smem, stream, a);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  dim3 grid(a.H, B, (max_seqlen + kBM - 1) / kBM);
  if (a.wl >= 0 || a.wr >= 0) hipLaunchKernelGGL((hstu_fwd_pc_kernel<D, true>), grid, dim3(512), smem, stream, a);
  else hipLaunchKernelGGL((hstu_fwd_pc_kernel<D, false>), grid, dim3(512), smem, stream, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <int D>
static int launch_fwd(const AttnArgs& a, int B, int max_seqlen, hipStream_t stream) {
  const size_t vtile = HSTU_VTR ? (size_t)kBN * (D == 32 ? 32 : D + 32) : (size_t)D * (kBN + 8);
  const size_t smem = (size_t)((D >= HSTU_DB_MIN ? 2 : 1) * (kBN * (D + 8) + vtile) + (D >= HSTU_QLDS_MIN ? kBM * (D + 8) : 0)) * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_kernel<D, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess) return MI355_ELAUNCH;
    attr_set = true;
  }
  // Dispatch order is x, then y, then z: with the block rank in z, ALL sequences' heaviest (causal: latest) row blocks
  // are handed out first and the light ones fill in behind them.  With the rank in x (per-sequence order 8,6,4,2 key
  // tiles at L = 512) the CUs freed first drew heavy blocks again and the slowest CU did 16 tiles where 10 is the mean.
  dim3 grid(a.H, B, (max_seqlen + kBM - 1) / kBM);
  if (a.rab || a.func) hipLaunchKernelGGL((hstu_fwd_kernel<D, true, true>), grid, dim3(256), smem, stream, a);
  else if (a.wl >= 0 || a.wr >= 0) hipLaunchKernelGGL((hstu_fwd_kernel<D, true>), grid, dim3(256), smem, stream, a);
  else hipLaunchKernelGGL((hstu_fwd_kernel<D, false>), grid, dim3(256), smem, stream, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

HSTU_NS_END

using namespace mi355;

// local window of the call in flight on this thread (set by the *_window entry points around the plain ones)
static thread_local int tl_wl = -1, tl_wr = -1;
// rows of q of the NEXT forward call on this thread (mi355_hstu_attn_fwd_hint_tokens; 0 = unknown): B x max_seqlen rows = a dense
// batch, which takes the paired-row-block forward
static thread_local int64_t tl_fwd_tokens = 0;
// attention bias of the call in flight on this thread (set by the *_rab entry points)
struct RabCall { const uint16_t* rab = nullptr; int64_t rb = 0, rh = 0, rr = 0; uint16_t* drab = nullptr; int64_t db = 0, dh = 0, dr = 0;
                 const int32_t* func = nullptr; int64_t fh = 0, fp = 0; int nf = 0; float fneg = 0.f;
                 void* kvis = nullptr; int64_t kvis_bytes = 0; };   // (backward: room for the key-block table of the func masks)
static thread_local RabCall tl_rab;
static int block_rotation(int heads) {   // MI355_HSTU_ROT (A/B): see seq_head_of_block; default -H = by a sequence per rank, jagged batches only
  constexpr int v = 0x7fffffff;
  return v == 0x7fffffff ? -heads : v;
}
static int column_major() {   // MI355_HSTU_CM=0: dense batches keep the rank-major grid (A/B)
  constexpr int v = 1;
  return v;
}
static int window_skip() {   // MI355_HSTU_WSKIP=0: keep the full tile loops under a window (A/B tests of the band clipping)
  static const int v = [] { const char* e = getenv("MI355_HSTU_WSKIP"); return e ? atoi(e) != 0 : 1; }();
  return v;
}

// ---- sizes of the backward's optional P / dS exchange (shared by both translation units)
static int xch_regions(int64_t head_dim) {   // dS, and (head_dim >= 128, where dV and dK are separate passes) P behind it
  constexpr int envp = 1;
  return (envp && head_dim >= 128) ? 2 : 1;
}
static int64_t xch_plan_header(int64_t units) { return ((units * 8 + 255) / 256 + (units * 4 + 255) / 256 + 1) * 256; }
static int64_t xch_unit_tiles(int64_t ng, int tri) { return tri ? ng * (ng + 1) / 2 : ng * ng; }
// tiles per region of the whole batch, bounded without reading the lengths: sum_b f(ng_b) <= (sum_b ng_b) x f(ng_max) / ng_max,
// sum_b ng_b <= T / 32 + B
static int64_t xch_tiles_bound(int64_t batch, int64_t num_heads, int64_t ng, int64_t total_tokens, int tri) {
  const int64_t umax = xch_unit_tiles(ng, tri);
  const int64_t dense = batch * umax;
  const int64_t jag = total_tokens > 0 ? (((total_tokens + 31) / 32 + batch) * umax + ng - 1) / ng : dense;
  return num_heads * (jag < dense ? jag : dense);
}
extern "C" int64_t mi355_hstu_attn_bwd_take_hint_(void);

extern "C" {
#if HSTU_TIMING && !HSTU_F16
int mi355_hstu_dbg_dump(void* out, int64_t bytes) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mi355::g_hstu_dbg), (size_t)bytes);
}
#endif

// hstu_varlen_fwd (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:335-523).  q, k, v, out: bf16 [total, H, d] with
// explicit token / head strides (elements); cu_seqlens int32 [B+1] shared by q and k (self attention over
// jagged sequences); num_contexts / num_targets int32 [B] or NULL; window (-1, 0) = causal, (-1, -1) = full.
int HSTU_FN(mi355_hstu_attn_fwd)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                        int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                        int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens,
                        int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                        const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size, int causal,
                        float alpha, float scaling_seqlen, hipStream_t stream) {
  return HSTU_FN(mi355_hstu_attn_fwd_kv)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                k_head_stride, v_head_stride, o_head_stride, cu_seqlens, nullptr, batch, num_heads, head_dim,
                                max_seqlen, num_contexts, num_targets, target_group_size, causal, alpha, scaling_seqlen,
                                nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// Inference forward: queries may be the tail of a longer key sequence (cu_seqlens_k, "delta-q") and the keys / values
// of the history may live in a paged cache [num_pages, 2, page_size, H, d] (hstu_attn_varlen_func kv_cache /
// page_offsets / page_ids / last_page_lens, hstu_attn_interface.py; kernel hstu_fwd.h Paged_KV paths :104-131,516-545).
int HSTU_FN(mi355_hstu_attn_fwd_kv)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                           int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                           int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q,
                           const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads, int64_t head_dim,
                           int64_t max_seqlen_q, const int32_t* num_contexts, const int32_t* num_targets,
                           int64_t target_group_size, int causal, float alpha, float scaling_seqlen, const void* kv_cache,
                           const int32_t* page_offsets, const int32_t* page_ids, const int32_t* last_page_lens,
                           int64_t page_size, hipStream_t stream) {
  MI355_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128 || head_dim == 256,
                  "head_dim must be one of 32, 64, 128, 256 (hstu_api.cpp:391)");
  MI355_CHECK_ARG(target_group_size >= 1, "target_group_size must be >= 1");
  MI355_CHECK_ARG(causal || (!num_contexts && !num_targets), "contextual / target masks require causal attention");
  MI355_CHECK_ARG(scaling_seqlen > 0.f, "scaling_seqlen must be positive");
  MI355_CHECK_ARG(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0 &&
                      q_head_stride % 8 == 0 && k_head_stride % 8 == 0 && v_head_stride % 8 == 0 && o_head_stride % 4 == 0,
                  "q/k/v strides must be multiples of 8 elements (16-byte rows)");
  MI355_CHECK_ARG(!kv_cache || (cu_seqlens_k && page_offsets && page_ids && last_page_lens && page_size > 0),
                  "a paged cache needs cu_seqlens_k, page_offsets, page_ids, last_page_lens and page_size");
  if (batch == 0 || max_seqlen_q == 0) return MI355_OK;
  AttnArgs a{};
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = (uint16_t*)out;
  a.q_row = q_row_stride; a.k_row = k_row_stride; a.v_row = v_row_stride; a.o_row = o_row_stride;
  a.q_head = q_head_stride; a.k_head = k_head_stride; a.v_head = v_head_stride; a.o_head = o_head_stride;
  a.cu_seqlens = cu_seqlens_q; a.num_contexts = num_contexts; a.num_targets = num_targets;
  a.H = (int)num_heads; a.causal = causal; a.group = (int)target_group_size;
  a.wl = tl_wl; a.wr = tl_wr; a.wskip = window_skip(); a.rot = block_rotation((int)num_heads); a.colmajor = column_major(); a.max_len = (int)max_seqlen_q;
  a.rab = tl_rab.rab; a.rab_b = tl_rab.rb; a.rab_h = tl_rab.rh; a.rab_r = tl_rab.rr;
  a.func = tl_rab.func; a.func_h = tl_rab.fh; a.func_p = tl_rab.fp; a.n_func = tl_rab.nf; a.func_neg = tl_rab.fneg;
  a.alpha = alpha; a.inv_scale = 1.0f / scaling_seqlen;
  a.cu_seqlens_k = cu_seqlens_k; a.kv_cache = (const uint16_t*)kv_cache; a.page_offsets = page_offsets; a.page_ids = page_ids;
  a.last_page_lens = last_page_lens; a.page_size = (int)page_size;
  static const int use_pc = !(getenv("MI355_HSTU_FWD") && atoi(getenv("MI355_HSTU_FWD")) == 5);   // round 4: two waves per SIMD, S waves + O waves (hook 5: the one-kind kernel)
  const int64_t fwd_tokens = tl_fwd_tokens;
  tl_fwd_tokens = 0;
  // (`func` masks of up to two bands ride the 64-rows-per-wave forward at every length; longer function lists keep the one-kind kernel)
  if (use_pc && head_dim == 256 && !a.kv_cache && !a.rab && (!a.func || a.n_func <= 5))
    return launch_fwd_pc<256>(a, (int)batch, (int)max_seqlen_q, stream, !cu_seqlens_k && fwd_tokens == batch * max_seqlen_q);
  switch (head_dim) {
    case 32: return launch_fwd<32>(a, (int)batch, (int)max_seqlen_q, stream);
    case 64: return launch_fwd<64>(a, (int)batch, (int)max_seqlen_q, stream);
    case 128: return launch_fwd<128>(a, (int)batch, (int)max_seqlen_q, stream);
    default: return launch_fwd<256>(a, (int)batch, (int)max_seqlen_q, stream);
  }
}

// rows of q of the NEXT forward call on this thread (the reference's forward signature does not carry them): a batch of
// batch x max_seqlen rows is dense and takes the paired-row-block kernel.  Optional; without it the unpaired kernel runs.
void HSTU_FN(mi355_hstu_attn_fwd_hint_tokens)(int64_t total_tokens) { tl_fwd_tokens = total_tokens; }

#if !HSTU_F16   // (type-agnostic: once, in the bf16 translation unit)
// append_kvcache (examples/commons/ops/cuda_ops/csrc/paged_kvcache_ops_kernel.cu:106-140): token i of the new history
// (i < *nnz) of sequence batch_indices[i] goes to position positions[i] of that user's paged cache, NHD layout
// [num_pages, 2, page_size, H, d]; its source row in append_key / append_value is i + seqlen_offsets[batch] (the
// tensors hold [new history | candidates] per sequence; seqlen_offsets = candidate offsets).
namespace mi355 {
__global__ void __launch_bounds__(256)
append_kvcache_kernel(uint16_t* __restrict__ cache, const int* __restrict__ kv_indices, const int* __restrict__ kv_indptr,
                      int H, int D, int page_size, const uint16_t* __restrict__ key, const uint16_t* __restrict__ value,
                      int64_t k_row, int64_t v_row, int64_t k_head, int64_t v_head, const int* __restrict__ batch_indices,
                      const int* __restrict__ positions, const int* __restrict__ offsets, const int* __restrict__ nnz_dev,
                      int nnz_max) {
  int nnz = nnz_dev ? *nnz_dev : nnz_max;
  if (nnz_max > 0 && nnz > nnz_max) nnz = nnz_max;
  const int chunks = H * D / 8;   // 16-byte chunks per token
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < (int64_t)nnz * chunks; w += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(w / chunks), c = (int)(w % chunks);
    const int hh = (c * 8) / D, dd = (c * 8) % D;
    const int bidx = batch_indices[i], pos = positions[i];
    const int page = kv_indices[kv_indptr[bidx] + pos / page_size];
    const int64_t src = (int64_t)i + offsets[bidx];
    const int64_t dst = (((int64_t)page * 2) * page_size + pos % page_size) * H * D + (int64_t)hh * D + dd;
    *reinterpret_cast<uint4*>(cache + dst) = *reinterpret_cast<const uint4*>(key + src * k_row + (int64_t)hh * k_head + dd);
    *reinterpret_cast<uint4*>(cache + dst + (int64_t)page_size * H * D) =
        *reinterpret_cast<const uint4*>(value + src * v_row + (int64_t)hh * v_head + dd);
  }
}
}  // namespace mi355

int mi355_append_kvcache(void* kv_cache, const int32_t* kv_indices, const int32_t* kv_indptr, int64_t num_heads,
                         int64_t head_dim, int64_t page_size, const void* append_key, const void* append_value,
                         int64_t k_row_stride, int64_t v_row_stride, int64_t k_head_stride, int64_t v_head_stride,
                         const int32_t* batch_indices, const int32_t* positions, const int32_t* seqlen_offsets,
                         const int32_t* nnz_dev, int64_t max_nnz, int64_t nnz_upper, hipStream_t stream) {
  MI355_CHECK_ARG(head_dim % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && k_head_stride % 8 == 0 &&
                      v_head_stride % 8 == 0, "head_dim and strides must be multiples of 8 elements");
  MI355_CHECK_ARG(page_size > 0, "page_size must be positive");
  MI355_CHECK_ARG(nnz_dev || max_nnz > 0, "nnz (device) or max_nnz required");
  const int64_t upper = max_nnz > 0 ? max_nnz : nnz_upper;
  if (upper == 0) return MI355_OK;
  hipLaunchKernelGGL(append_kvcache_kernel, dim3(grid_for(upper * num_heads * head_dim / 8, 256)), dim3(256), 0, stream,
                     (uint16_t*)kv_cache, kv_indices, kv_indptr, (int)num_heads, (int)head_dim, (int)page_size,
                     (const uint16_t*)append_key, (const uint16_t*)append_value, k_row_stride, v_row_stride, k_head_stride,
                     v_head_stride, batch_indices, positions, seqlen_offsets, nnz_dev, (int)max_nnz);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int64_t mi355_hstu_attn_bwd_workspace_bytes(int64_t total_tokens, int64_t num_heads, int64_t head_dim) {
  (void)total_tokens; (void)num_heads; (void)head_dim;
  return 0;  // the backward needs no scratch (no fp32 dQ accumulator, no atomics); see mi355_hstu_attn_bwd_ds_bytes
}

// Optional scratch of the backward: with a workspace of at least this size the dK pass leaves dS (bf16, 32 x 32 sub-tiles
// in its register layout) for the dQ pass, which then skips the S / dP recomputation -- 2 of its 3 GEMMs and the SiLU.
// (head_dim < 128 runs dV and dK in one pass: only dS is exchanged there.)  0: the exchange is switched off.
int64_t mi355_hstu_attn_bwd_ds_bytes(int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen) {
  static const int env = getenv("MI355_HSTU_DS") ? atoi(getenv("MI355_HSTU_DS")) : 1;
  if (!env || batch <= 0 || max_seqlen <= 0) return 0;
  const int64_t ng = (max_seqlen + 31) / 32;
  return batch * num_heads * ng * ng * 2048 * xch_regions(head_dim);
}

// The same exchange under a byte cap (round 4): the buffer holds one CHUNK of (sequence, head) units at a time, each unit
// with the ceil(L_b / 32)^2 tiles of its own length, and the three passes run once per chunk -- scratch by the jagged sum of
// the batch (bounded on the host, without reading the lengths, by (T / 32 + B) x ceil(max_seqlen / 32) tiles per head),
// never more than `cap_bytes`.  Returns the workspace size to hand to mi355_hstu_attn_bwd (plan header included); 0 when even
// two units of the longest sequence do not fit under the cap (the recomputing passes then run: no scratch at all).
int64_t mi355_hstu_attn_bwd_ds_bytes_capped(int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                                            int64_t total_tokens, int64_t cap_bytes, int plain_causal) {
  if (mi355_hstu_attn_bwd_ds_bytes(batch, num_heads, head_dim, max_seqlen) == 0) return 0;
  const int64_t ng = (max_seqlen + 31) / 32, regions = xch_regions(head_dim), hdr = xch_plan_header(batch * num_heads);
  const int tri = plain_causal != 0;
  const int64_t want = hdr + regions * xch_tiles_bound(batch, num_heads, ng, total_tokens, tri) * 2048;
  const int64_t bytes = want < cap_bytes ? want : cap_bytes;
  return (bytes - hdr) / regions / 2048 >= 2 * xch_unit_tiles(ng, tri) ? bytes : 0;
}
// total tokens of the NEXT mi355_hstu_attn_bwd call on this thread (its signature, the reference's, does not carry them):
// tightens the number of chunk passes a capped workspace is walked in; 0 / not set = the dense bound
static thread_local int64_t tl_bwd_tokens = 0;
void mi355_hstu_attn_bwd_hint_tokens(int64_t total_tokens) { tl_bwd_tokens = total_tokens; }
int64_t mi355_hstu_attn_bwd_take_hint_(void) { const int64_t t = tl_bwd_tokens; tl_bwd_tokens = 0; return t; }   // (internal: both translation units)

#endif

// pinned host word the plan kernel writes when its plan needs more chunk passes than the host launched (see the kernel)
static int* plan_err_word() {
  static int* w = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    if (hipHostMalloc((void**)&w, sizeof(int), hipHostMallocMapped) == hipSuccess && w) *w = 0; else w = nullptr;
  }
  return w;
}

// hstu_varlen_bwd (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:525-719).  dq, dk, dv: contiguous bf16 [total, H, d].
int HSTU_FN(mi355_hstu_attn_bwd)(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                        int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                        int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                        const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                        const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size, int causal,
                        float alpha, float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128 || head_dim == 256,
                  "head_dim must be one of 32, 64, 128, 256 (hstu_api.cpp:391)");
  MI355_CHECK_ARG(target_group_size >= 1, "target_group_size must be >= 1");
  MI355_CHECK_ARG(causal || (!num_contexts && !num_targets), "contextual / target masks require causal attention");
  MI355_CHECK_ARG(scaling_seqlen > 0.f, "scaling_seqlen must be positive");
  MI355_CHECK_ARG(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && do_row_stride % 8 == 0 &&
                      q_head_stride % 8 == 0 && k_head_stride % 8 == 0 && v_head_stride % 8 == 0 && do_head_stride % 8 == 0,
                  "q/k/v/dout strides must be multiples of 8 elements (16-byte rows)");
  if (batch == 0 || max_seqlen == 0) return MI355_OK;
  if (int* ew = plan_err_word()) {
    if (*ew) {
      *ew = 0;
      mi355_set_error("an earlier hstu backward planned more exchange chunks than were launched (token hint of another batch?): "
                      "its gradients are incomplete");
      return MI355_EINVAL;
    }
  }
  BwdAttnArgs g{};
  AttnArgs& a = g.f;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = nullptr;
  a.q_row = q_row_stride; a.k_row = k_row_stride; a.v_row = v_row_stride; a.o_row = 0;
  a.q_head = q_head_stride; a.k_head = k_head_stride; a.v_head = v_head_stride; a.o_head = 0;
  a.cu_seqlens = cu_seqlens; a.num_contexts = num_contexts; a.num_targets = num_targets;
  a.H = (int)num_heads; a.causal = causal; a.group = (int)target_group_size;
  a.wl = tl_wl; a.wr = tl_wr; a.wskip = window_skip(); a.rot = block_rotation((int)num_heads); a.colmajor = column_major(); a.max_len = (int)max_seqlen;
  a.rab = tl_rab.rab; a.rab_b = tl_rab.rb; a.rab_h = tl_rab.rh; a.rab_r = tl_rab.rr;
  a.func = tl_rab.func; a.func_h = tl_rab.fh; a.func_p = tl_rab.fp; a.n_func = tl_rab.nf; a.func_neg = tl_rab.fneg;
  a.alpha = alpha; a.inv_scale = 1.0f / scaling_seqlen;
  g.dout = (const uint16_t*)dout; g.do_row = do_row_stride; g.do_head = do_head_stride;
  g.dq = (uint16_t*)dq; g.dk = (uint16_t*)dk; g.dv = (uint16_t*)dv;
  g.ds_ws = nullptr; g.ng = (int)((max_seqlen + 31) / 32); g.bq_kv = 32;
  g.drab = tl_rab.drab; g.drab_b = tl_rab.db; g.drab_h = tl_rab.dh; g.drab_r = tl_rab.dr;
  g.plan_base = nullptr; g.plan_chunk = nullptr; g.chunk = 0;
  g.func_kvis = nullptr; g.func_kvis_h = 0; g.func_gext = nullptr;
  const int64_t tokens_hint = mi355_hstu_attn_bwd_take_hint_();
  if (a.func && tl_rab.kvis && a.wskip) {
    // the query rows that reach every 128-key block, for the key-major passes (the query-major pass derives its reach in place)
    // (table of total_tokens / 128 + batch + 1 entries per function set; a key block whose entry lies beyond the buffer is simply
    //  not clipped -- both kernels check the index)
    // layout: [gext: nfh x 4 slots int4 | kvis: nfh x slots int2] = 72 bytes per slot and function set
    const int64_t nfh = a.func_h ? num_heads : 1;
    const int64_t slots = tl_rab.kvis_bytes / 72 / nfh;
    if (slots > 0 && ((uintptr_t)tl_rab.kvis & 15) == 0) {
      const size_t sm = 4 * (size_t)((max_seqlen + 31) / 32) * sizeof(int);
      int4* gext = (int4*)tl_rab.kvis;
      int2* kvis = (int2*)(gext + nfh * 4 * slots);
      hipLaunchKernelGGL(hstu_func_kvis_kernel, dim3((unsigned)batch, (unsigned)nfh), dim3(256), sm, stream, a, (int)nfh, kvis, slots, gext);
      g.func_kvis = kvis; g.func_kvis_h = slots; g.func_gext = gext;
    }
  }
  int nchunks = 1;
  {
    const int64_t need = mi355_hstu_attn_bwd_ds_bytes(batch, num_heads, head_dim, max_seqlen);
    const int64_t regions = xch_regions(head_dim), units = batch * num_heads, hdr = xch_plan_header(units);
    const int64_t udense = (int64_t)g.ng * g.ng;
    // plain causal mask: the sub-tiles above the diagonal do not exist in the chunked layout (see BwdAttnArgs::tri)
    g.tri = (causal && !num_contexts && tl_wl < 0 && tl_wr < 0 && !tl_rab.rab && !tl_rab.func) ? 1 : 0;
    const int64_t umax = xch_unit_tiles(g.ng, g.tri);
    g.p_ws = nullptr;
    if (need > 0 && workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0) {
      // the dense layout: everything in one pass
      g.ds_ws = (uint16_t*)workspace;
      const int64_t one = batch * num_heads * udense * 2048;
      if (need >= 2 * one && head_dim >= 128) g.p_ws = (uint16_t*)((uint8_t*)workspace + one);
    } else if (need > 0 && workspace && ((uintptr_t)workspace & 255) == 0 && !tl_rab.rab &&
               (!tl_rab.func || (head_dim == 256 && tl_rab.nf <= 5)) && workspace_bytes > hdr &&
               (workspace_bytes - hdr) / regions / 2048 >= 2 * umax) {
      // the jagged, chunked layout: [plan_base | plan_chunk | nchunks | dS region | P region]
      const int64_t cap_tiles = (workspace_bytes - hdr) / regions / 2048;
      uint8_t* w = (uint8_t*)workspace;
      int64_t* base = (int64_t*)w;
      int32_t* chunk = (int32_t*)(w + (units * 8 + 255) / 256 * 256);
      int32_t* nch = (int32_t*)(w + hdr - 256);
      g.plan_base = base; g.plan_chunk = chunk;
      g.ds_ws = (uint16_t*)(w + hdr);
      if (regions == 2) g.p_ws = (uint16_t*)(w + hdr + cap_tiles * 2048);
      // chunks the greedy cut can need at most: every chunk but the last holds more than cap_tiles - umax tiles
      const int64_t bound = xch_tiles_bound(batch, num_heads, g.ng, tokens_hint, g.tri);
      nchunks = (int)((bound + (cap_tiles - umax)) / (cap_tiles - umax + 1));
      // (round 5) a buffer as large as the bound itself holds every unit in ONE chunk: the formula above, which only knows that a
      // chunk but the last holds more than cap - umax tiles, still said two -- and every jagged backward under the cap launched a
      // second, empty set of three kernels (14 + 7 + 7 us at C4, profiles/r04_step_timeline.txt)
      if (bound <= cap_tiles) nchunks = 1;
      if (nchunks < 1) nchunks = 1;
      if (nchunks > units) nchunks = (int)units;
      hipLaunchKernelGGL(hstu_bwd_plan_kernel, dim3(1), dim3(256), 0, stream, cu_seqlens, (int)batch, (int)num_heads, cap_tiles, g.tri,
                         base, chunk, nch, nchunks, plan_err_word());
    }
  }
  for (int c = 0; c < nchunks; ++c) {
    g.chunk = c;
    int rc;
    switch (head_dim) {
      case 32: rc = launch_bwd<32>(g, (int)batch, (int)max_seqlen, stream); break;
      case 64: rc = launch_bwd<64>(g, (int)batch, (int)max_seqlen, stream); break;
      case 128: rc = launch_bwd<128>(g, (int)batch, (int)max_seqlen, stream); break;
      default: rc = launch_bwd<256>(g, (int)batch, (int)max_seqlen, stream); break;
    }
    if (rc != MI355_OK) return rc;
  }
  return MI355_OK;
}


// Local (sliding window) attention, window_size = (left, right) of hstu_attn_varlen_func (hstu_api.cpp:154-165: a negative
// side is unbounded; (-1, 0) is causal, (-1, -1) full): query i sees keys i - left .. i + right.  No contextual / target
// rows with a window (hstu_attn_interface.py:238-245), self attention only.
static int window_causal(int64_t wl, int64_t wr) { return wr == 0 ? 1 : 0; }   // right == 0: the causal tile loops apply

int HSTU_FN(mi355_hstu_attn_fwd_window)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                               int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                               int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens,
                               int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen, int64_t window_left,
                               int64_t window_right, float alpha, float scaling_seqlen, hipStream_t stream) {
  MI355_CHECK_ARG(window_left >= -1 && window_right >= -1 && window_left < (1 << 30) && window_right < (1 << 30), "bad window");
  tl_wl = (int)window_left; tl_wr = window_right == 0 ? -1 : (int)window_right;   // (right == 0 is what `causal` already says)
  const int rc = HSTU_FN(mi355_hstu_attn_fwd)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                     k_head_stride, v_head_stride, o_head_stride, cu_seqlens, batch, num_heads, head_dim,
                                     max_seqlen, nullptr, nullptr, 1, window_causal(window_left, window_right), alpha,
                                     scaling_seqlen, stream);
  tl_wl = tl_wr = -1;
  return rc;
}

// The inference forward (delta-q keys, paged cache) with a local attention window (hstu_fwd.h:104-131,463-470,516-545 compose
// Is_local with the query offset and Paged_KV): the window is taken over ABSOLUTE positions -- query r of a sequence sits at Lk - Lq + r.
int HSTU_FN(mi355_hstu_attn_fwd_kv_window)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                                  int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                                  int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q,
                                  const int32_t* cu_seqlens_k, int64_t batch, int64_t num_heads, int64_t head_dim,
                                  int64_t max_seqlen_q, int64_t window_left, int64_t window_right, float alpha,
                                  float scaling_seqlen, const void* kv_cache, const int32_t* page_offsets,
                                  const int32_t* page_ids, const int32_t* last_page_lens, int64_t page_size, hipStream_t stream) {
  MI355_CHECK_ARG(window_left >= -1 && window_right >= -1 && window_left < (1 << 30) && window_right < (1 << 30), "bad window");
  tl_wl = (int)window_left; tl_wr = window_right == 0 ? -1 : (int)window_right;
  const int rc = HSTU_FN(mi355_hstu_attn_fwd_kv)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                        k_head_stride, v_head_stride, o_head_stride, cu_seqlens_q, cu_seqlens_k, batch, num_heads,
                                        head_dim, max_seqlen_q, nullptr, nullptr, 1, window_causal(window_left, window_right), alpha,
                                        scaling_seqlen, kv_cache, page_offsets, page_ids, last_page_lens, page_size, stream);
  tl_wl = tl_wr = -1;
  return rc;
}

int HSTU_FN(mi355_hstu_attn_bwd_window)(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                               int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                               int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                               const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim,
                               int64_t max_seqlen, int64_t window_left, int64_t window_right, float alpha,
                               float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(window_left >= -1 && window_right >= -1 && window_left < (1 << 30) && window_right < (1 << 30), "bad window");
  tl_wl = (int)window_left; tl_wr = window_right == 0 ? -1 : (int)window_right;
  const int rc = HSTU_FN(mi355_hstu_attn_bwd)(dout, q, k, v, dq, dk, dv, q_row_stride, k_row_stride, v_row_stride, do_row_stride,
                                     q_head_stride, k_head_stride, v_head_stride, do_head_stride, cu_seqlens, batch, num_heads,
                                     head_dim, max_seqlen, nullptr, nullptr, 1, window_causal(window_left, window_right), alpha,
                                     scaling_seqlen, workspace, workspace_bytes, stream);
  tl_wl = tl_wr = -1;
  return rc;
}

// Relative attention bias (`rab` / `has_drab` of hstu_attn_varlen_func; hstu_api.cpp:100-111,253-263,417-430,659-667).
// rab: bf16 [batch][heads or 1][max_seqlen][max_seqlen] given by its batch / head / row strides in elements (last dim
// contiguous; head stride 0 = one matrix for all heads); added to q_i . k_j before alpha and SiLU.  The mask is given as
// in hstu_attn_varlen_func: window (-1, 0) causal (contextual / target rows allowed), (-1, -1) full, else a local window.
// Self attention over contiguous keys only.  drab (backward, nullable): bf16 with its own strides, one matrix per head,
// zero-filled by the caller; receives d loss / d rab (= dS) at every position inside the sequences.
static int rab_mask(int64_t wl, int64_t wr, const int32_t* nc, const int32_t* nt, int* causal) {
  MI355_CHECK_ARG(wl >= -1 && wr >= -1 && wl < (1 << 30) && wr < (1 << 30), "bad window");
  const bool local = !(wl == -1 && (wr == -1 || wr == 0));
  MI355_CHECK_ARG(!(nc || nt) || (wl == -1 && wr == 0), "contextual / target masks require the causal mask (-1, 0)");
  *causal = wr == 0 ? 1 : 0;
  tl_wl = local ? (int)wl : -1;
  tl_wr = (local && wr != 0) ? (int)wr : -1;
  return MI355_OK;
}

int HSTU_FN(mi355_hstu_attn_fwd_rab)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                            int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                            int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens, int64_t batch,
                            int64_t num_heads, int64_t head_dim, int64_t max_seqlen, const int32_t* num_contexts,
                            const int32_t* num_targets, int64_t target_group_size, int64_t window_left, int64_t window_right,
                            float alpha, float scaling_seqlen, const void* rab, int64_t rab_batch_stride,
                            int64_t rab_head_stride, int64_t rab_row_stride, hipStream_t stream) {
  MI355_CHECK_ARG(rab != nullptr && rab_row_stride >= max_seqlen, "rab must be [batch][heads or 1][max_seqlen][max_seqlen]");
  int causal = 0;
  if (const int rc = rab_mask(window_left, window_right, num_contexts, num_targets, &causal)) return rc;
  tl_rab = RabCall{(const uint16_t*)rab, rab_batch_stride, rab_head_stride, rab_row_stride, nullptr, 0, 0, 0};
  const int rc = HSTU_FN(mi355_hstu_attn_fwd)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                     k_head_stride, v_head_stride, o_head_stride, cu_seqlens, batch, num_heads, head_dim,
                                     max_seqlen, num_contexts, num_targets, target_group_size, causal, alpha, scaling_seqlen,
                                     stream);
  tl_rab = RabCall{};
  tl_wl = tl_wr = -1;
  return rc;
}

// The inference forward (delta-q keys, paged cache) with a relative attention bias: rab[b][h][i][j] is indexed by ABSOLUTE positions
// (query r of a sequence is row Lk - Lq + r), max_seqlen = the padded extent of the bias in i and j (>= every Lk).
int HSTU_FN(mi355_hstu_attn_fwd_kv_rab)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                               int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                               int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                               int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen_q, int64_t max_seqlen_k,
                               const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                               int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                               int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, const void* kv_cache,
                               const int32_t* page_offsets, const int32_t* page_ids, const int32_t* last_page_lens,
                               int64_t page_size, hipStream_t stream) {
  MI355_CHECK_ARG(rab != nullptr && rab_row_stride >= max_seqlen_k, "rab must be [batch][heads or 1][max_seqlen_k][max_seqlen_k]");
  int causal = 0;
  if (const int rc = rab_mask(window_left, window_right, num_contexts, num_targets, &causal)) return rc;
  tl_rab = RabCall{(const uint16_t*)rab, rab_batch_stride, rab_head_stride, rab_row_stride, nullptr, 0, 0, 0};
  const int rc = HSTU_FN(mi355_hstu_attn_fwd_kv)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                        k_head_stride, v_head_stride, o_head_stride, cu_seqlens_q, cu_seqlens_k, batch, num_heads,
                                        head_dim, max_seqlen_q, num_contexts, num_targets, target_group_size, causal, alpha,
                                        scaling_seqlen, kv_cache, page_offsets, page_ids, last_page_lens, page_size, stream);
  tl_rab = RabCall{};
  tl_wl = tl_wr = -1;
  return rc;
}

int HSTU_FN(mi355_hstu_attn_bwd_rab)(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                            int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                            int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                            const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                            const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                            int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const void* rab,
                            int64_t rab_batch_stride, int64_t rab_head_stride, int64_t rab_row_stride, void* drab,
                            int64_t drab_batch_stride, int64_t drab_head_stride, int64_t drab_row_stride, hipStream_t stream) {
  MI355_CHECK_ARG(rab != nullptr && rab_row_stride >= max_seqlen, "rab must be [batch][heads or 1][max_seqlen][max_seqlen]");
  MI355_CHECK_ARG(drab == nullptr || (drab_row_stride >= max_seqlen && drab_head_stride > 0),
                  "drab must hold one [max_seqlen][max_seqlen] matrix per head");
  int causal = 0;
  if (const int rc = rab_mask(window_left, window_right, num_contexts, num_targets, &causal)) return rc;
  tl_rab = RabCall{(const uint16_t*)rab, rab_batch_stride, rab_head_stride, rab_row_stride,
                   (uint16_t*)drab, drab_batch_stride, drab_head_stride, drab_row_stride};
  const int rc = HSTU_FN(mi355_hstu_attn_bwd)(dout, q, k, v, dq, dk, dv, q_row_stride, k_row_stride, v_row_stride, do_row_stride,
                                     q_head_stride, k_head_stride, v_head_stride, do_head_stride, cu_seqlens, batch, num_heads,
                                     head_dim, max_seqlen, num_contexts, num_targets, target_group_size, causal, alpha,
                                     scaling_seqlen, nullptr, 0, stream);
  tl_rab = RabCall{};
  tl_wl = tl_wr = -1;
  return rc;
}

// Arbitrary mask functions (`func` of hstu_attn_varlen_func; hstu_api.cpp:170-180, applied in hstu_fwd.h:139-145, 493-556) read INSIDE
// the kernels (AttnArgs::func): int32 func[heads or 1][n_func][>= total_q], n_func odd, given by its head stride (0 = one set for
// all heads) and the stride between the bounds of a token, last dimension contiguous.  Query token t sees key position j of its
// sequence iff j < func[0][t] or func[2p-1][t] <= j < func[2p][t]; func_neg is what a masked pair gets added to q.k (finite in the
// operand type: -1e9 bf16, -6e4 fp16).  The other masks apply on top, as for the bias entry points.  Forward: training keys,
// delta-q keys and the paged cache (as mi355_hstu_attn_fwd_kv); backward: self attention over contiguous keys.
int HSTU_FN(mi355_hstu_attn_fwd_kv_func)(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride, int64_t k_row_stride,
                                int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride, int64_t k_head_stride,
                                int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen_q, int64_t max_seqlen_k,
                                const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                                int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                                int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                                const void* kv_cache, const int32_t* page_offsets, const int32_t* page_ids,
                                const int32_t* last_page_lens, int64_t page_size, hipStream_t stream) {
  MI355_CHECK_ARG(func != nullptr && n_func >= 1 && (n_func & 1) == 1 && func_bound_stride > 0 && func_neg < 0.f,
                  "func must be int32 [heads or 1][n_func odd][tokens], func_neg negative");
  int causal = 0;
  if (const int rc = rab_mask(window_left, window_right, num_contexts, num_targets, &causal)) return rc;
  tl_rab = RabCall{};
  tl_rab.func = func; tl_rab.fh = func_head_stride; tl_rab.fp = func_bound_stride; tl_rab.nf = (int)n_func; tl_rab.fneg = func_neg;
  const int rc = HSTU_FN(mi355_hstu_attn_fwd_kv)(q, k, v, out, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_head_stride,
                                        k_head_stride, v_head_stride, o_head_stride, cu_seqlens_q, cu_seqlens_k, batch, num_heads,
                                        head_dim, max_seqlen_q, num_contexts, num_targets, target_group_size, causal, alpha,
                                        scaling_seqlen, kv_cache, page_offsets, page_ids, last_page_lens, page_size, stream);
  tl_rab = RabCall{};
  tl_wl = tl_wr = -1;
  return rc;
}

int HSTU_FN(mi355_hstu_attn_bwd_func)(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                             int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                             int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                             const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                             const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size,
                             int64_t window_left, int64_t window_right, float alpha, float scaling_seqlen, const int32_t* func,
                             int64_t func_head_stride, int64_t func_bound_stride, int64_t n_func, float func_neg,
                             void* func_workspace, int64_t func_workspace_bytes, void* workspace, int64_t workspace_bytes,
                             hipStream_t stream) {
  MI355_CHECK_ARG(func != nullptr && n_func >= 1 && (n_func & 1) == 1 && func_bound_stride > 0 && func_neg < 0.f,
                  "func must be int32 [heads or 1][n_func odd][tokens], func_neg negative");
  int causal = 0;
  if (const int rc = rab_mask(window_left, window_right, num_contexts, num_targets, &causal)) return rc;
  tl_rab = RabCall{};
  tl_rab.func = func; tl_rab.fh = func_head_stride; tl_rab.fp = func_bound_stride; tl_rab.nf = (int)n_func; tl_rab.fneg = func_neg;
  tl_rab.kvis = func_workspace; tl_rab.kvis_bytes = func_workspace_bytes;
  // the P / dS exchange (as mi355_hstu_attn_bwd's workspace) serves functions of up to two bands at head dim 256 when the tables
  // above are there (the passes that read the exchange replay the table to know which sub-tiles exist); otherwise: the recomputing passes
  const bool xch = head_dim == 256 && n_func <= 5 && func_workspace != nullptr && window_skip();
  const int rc = HSTU_FN(mi355_hstu_attn_bwd)(dout, q, k, v, dq, dk, dv, q_row_stride, k_row_stride, v_row_stride, do_row_stride,
                                     q_head_stride, k_head_stride, v_head_stride, do_head_stride, cu_seqlens, batch, num_heads,
                                     head_dim, max_seqlen, num_contexts, num_targets, target_group_size, causal, alpha,
                                     scaling_seqlen, xch ? workspace : nullptr, xch ? workspace_bytes : 0, stream);
  tl_rab = RabCall{};
  tl_wl = tl_wr = -1;
  return rc;
}

}  // extern "C"
