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

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int kBM = 128;  // query rows per workgroup (32 per wave)
constexpr int kBN = 64;   // keys per tile

struct AttnArgs {
  const uint16_t* q; const uint16_t* k; const uint16_t* v;
  uint16_t* out;
  int64_t q_row, k_row, v_row, o_row;   // element stride between tokens
  int64_t q_head, k_head, v_head, o_head;  // element stride between heads
  const int* cu_seqlens;
  const int* num_contexts;  // [B] or nullptr
  const int* num_targets;   // [B] or nullptr
  int H, causal, group;
  float alpha, inv_scale;
};

struct SeqInfo { int start, L, c, hlen; bool has_ctx, has_tgt; };

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
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ float silu_f(float x) {
  return x * __frcp_rn(1.0f + __expf(-x));
}

// Row-side mask state of one query (causal case).  M(i, j) of the reference collapses to
//   j <= jmax  and  (j < hlen  or  j >= jlo)
// with jmax = i (history / target rows) or hlen-1 (contextual rows: they see the whole history), and
// jlo = first key of the row's target group (0 for non-target rows).  Non causal: j < L.
struct RowMask { int jmax, jlo, hlen; };
__device__ __forceinline__ RowMask row_mask(int i, const SeqInfo& s, int causal, int group) {
  RowMask m;
  m.hlen = s.hlen;
  if (!causal) { m.jmax = s.L - 1; m.jlo = 0; m.hlen = s.L; return m; }
  m.jmax = (s.has_ctx && i < s.c) ? s.hlen - 1 : i;
  if (m.jmax > s.L - 1) m.jmax = s.L - 1;
  m.jlo = (s.has_tgt && i >= s.hlen) ? s.hlen + ((i - s.hlen) / group) * group : 0;
  return m;
}
__device__ __forceinline__ bool key_ok(int j, const RowMask& m) { return (j <= m.jmax) & ((j < m.hlen) | (j >= m.jlo)); }

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
template <int D>
__global__ void __launch_bounds__(256) hstu_fwd_kernel(AttnArgs a) {
  constexpr int KS = D + 8;    // padded K row (elements)
  constexpr int VS = kBN + 8;  // padded V^T row (elements)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Ks = smem;                 // [kBN][KS]
  uint16_t* Vt = smem + kBN * KS;      // [D][VS], key positions permuted inside every 16-group
  constexpr bool QLDS = D >= 256;      // large head dim: Q fragments live in LDS to free 64 VGPRs for prefetching
  uint16_t* Qs = Vt + D * VS;          // [kBM][KS] (QLDS only)

  const int b = blockIdx.z, h = blockIdx.y;
  SeqInfo s;
  s.start = a.cu_seqlens[b];
  s.L = a.cu_seqlens[b + 1] - s.start;
  const int nblk = (s.L + kBM - 1) / kBM;
  if ((int)blockIdx.x >= nblk) return;
  const int m0 = (nblk - 1 - (int)blockIdx.x) * kBM;  // heaviest (latest) row blocks first
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);

  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qrow0 = m0 + 32 * wv;
  const int qi = qrow0 + l31;
  const bool wave_live = qrow0 < s.L;

  // keys this row block can reach
  int last_row = m0 + kBM - 1 < s.L - 1 ? m0 + kBM - 1 : s.L - 1;
  int n_end = s.L;
  if (a.causal) {
    n_end = last_row + 1;
    if (s.has_ctx && m0 < s.c && s.hlen > n_end) n_end = s.hlen;
  }
  // the wave's own reach (skips MFMA work on tiles past it)
  int w_last = qrow0 + 31 < s.L - 1 ? qrow0 + 31 : s.L - 1;
  int w_end = s.L;
  if (a.causal) {
    w_end = w_last + 1;
    if (s.has_ctx && qrow0 < s.c && s.hlen > w_end) w_end = s.hlen;
  }

  // ---- Q fragments (B operand of GEMM 1): lane = (query l31, k half hi), 8 consecutive d per 16-slice
  bf16x8_t qf[QLDS ? 1 : D / 16];
  if constexpr (QLDS) {
    stage_rows<D, kBM>(Qs, a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head, a.q_row, m0, s.L);
  } else {
    const uint16_t* qp = a.q + (int64_t)(s.start + (qi < s.L ? qi : 0)) * a.q_row + (int64_t)h * a.q_head + 8 * hi;
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (qi < s.L) t = *reinterpret_cast<const uint4*>(qp + 16 * sl);
      qf[sl] = *reinterpret_cast<bf16x8_t*>(&t);
    }
  }

  f32x16_t acc_o[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;

  const float nal2e = -a.alpha * 1.44269504088896f, ais = a.alpha * a.inv_scale;
  const RowMask rm = row_mask(qi, s, a.causal, a.group);
  // ---- software-pipelined staging (issue-early / write-late): the K / V tile of step n+1 is fetched
  // into registers while step n computes; it is written to LDS (V transposed) after the barrier.
  constexpr int KCH = kBN * D / 8;            // 16-B chunks of the K tile
  constexpr int KPT = (KCH + 255) / 256;      // per thread
  constexpr int VCH = (kBN / 4) * (D / 8);    // (4 keys x 8 d) blocks of the V tile
  constexpr int VPT = (VCH + 255) / 256;
  uint4 kreg[KPT], vreg[VPT][4];
  const uint16_t* kbase = a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head;
  const uint16_t* vbase = a.v + (int64_t)s.start * a.v_row + (int64_t)h * a.v_head;
  auto fetch = [&](int n0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int key = ch / (D / 8), dc = ch % (D / 8);
      // rows past the sequence end are clamped to its last row: their P is masked to 0, so any finite data do
      const int row = n0 + key < s.L ? n0 + key : s.L - 1;
      if (KCH % 256 == 0 || ch < KCH) kreg[i] = *reinterpret_cast<const uint4*>(kbase + (int64_t)row * a.k_row + 8 * dc);
    }
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
        if (VCH % 256 == 0 || ch < VCH) vreg[i][kk] = *reinterpret_cast<const uint4*>(vbase + (int64_t)row * a.v_row + 8 * dc);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      const int key = ch / (D / 8), dc = ch % (D / 8);
      if (KCH % 256 == 0 || ch < KCH) *reinterpret_cast<uint4*>(Ks + key * KS + 8 * dc) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int ch = threadIdx.x + 256 * i;
      if (VCH % 256 != 0 && ch >= VCH) continue;
      const int kgpos = ch % (kBN / 4), dc = ch / (kBN / 4);
      const int g16 = kgpos >> 2, pg = kgpos & 3;
      const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&vreg[i][0]);
      const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&vreg[i][1]);
      const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&vreg[i][2]);
      const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&vreg[i][3]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
        uint2 o;
        o.x = __builtin_amdgcn_perm(w1[e >> 1], w0[e >> 1], sel);
        o.y = __builtin_amdgcn_perm(w3[e >> 1], w2[e >> 1], sel);
        *reinterpret_cast<uint2*>(Vt + (8 * dc + e) * VS + 16 * g16 + 4 * pg) = o;
      }
    }
  };

  if (n_end > 0) fetch(0);
  for (int n0 = 0; n0 < n_end; n0 += kBN) {
    __syncthreads();
    commit();
    __syncthreads();
    if (n0 + kBN < n_end) fetch(n0 + kBN);
    if (!wave_live || n0 >= w_end) continue;

    // ---- GEMM 1: S^T[64 keys x 32 q] = K Q^T, two 32-key tiles.  Operand fragments are fetched from LDS
    // in batches of 8 ahead of the 8 MFMAs that consume them (hipcc otherwise emits read-wait-mfma triples).
    f32x16_t acc_s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_s[t][r] = 0.f;
#pragma unroll
    for (int sl0 = 0; sl0 < D / 16; sl0 += 4) {
      bf16x8_t kfr[4][2], qfr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sl0 + u >= D / 16) continue;
        if constexpr (QLDS) qfr[u] = *reinterpret_cast<const bf16x8_t*>(Qs + (32 * wv + l31) * KS + 16 * (sl0 + u) + 8 * hi);
        else qfr[u] = qf[sl0 + u];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          kfr[u][t] = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + l31) * KS + 16 * (sl0 + u) + 8 * hi);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (sl0 + u < D / 16)
            acc_s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[u][t], qfr[u], acc_s[t], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, QLDS ? 12 : 8, 0);  // the DS reads of this batch first,
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);               // then its 8 MFMAs
    }
    // ---- P^T = mask * SiLU(alpha S^T) / scaling, packed straight into the B operand of GEMM 2.
    // Tiles strictly below the diagonal of every row of the wave need no per-element mask.
    const bool full = a.causal && (n0 + kBN - 1 <= qrow0) && (!s.has_ctx || qrow0 >= s.c) && (!s.has_tgt || n0 + kBN - 1 < s.hlen);
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
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int dt0 = 0; dt0 < NDT; dt0 += DB) {
        bf16x8_t vfr[DB];
#pragma unroll
        for (int u = 0; u < DB; ++u)
          vfr[u] = *reinterpret_cast<const bf16x8_t*>(Vt + (32 * (dt0 + u) + l31) * VS + 16 * ks + 8 * hi);
#pragma unroll
        for (int u = 0; u < DB; ++u)
          acc_o[dt0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[u], pf[ks], acc_o[dt0 + u], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, DB, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, DB, 0);
      }
    }
  }

  // ---- epilogue: O^T accumulator -> out[token][head][d] (bf16), 4 consecutive d per store
  if (qi < s.L) {
    uint16_t* op = a.out + (int64_t)(s.start + qi) * a.o_row + (int64_t)h * a.o_head;
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

// same rows transposed -> LDS [D][NR+8], row positions permuted inside every 16-group ({0-3,8-11,4-7,12-15})
template <int D, int NR>
__device__ __forceinline__ void stage_transposed(uint16_t* dst, const uint16_t* src, int64_t row_stride, int row0, int nvalid) {
  constexpr int NCH = (NR / 4) * (D / 8);
#pragma unroll
  for (int ch = threadIdx.x; ch < NCH; ch += 256) {
    const int gpos = ch % (NR / 4), dc = ch / (NR / 4);
    const int g16 = gpos >> 2, pg = gpos & 3;
    const int ak = pg == 1 ? 2 : (pg == 2 ? 1 : pg);
    const int r0 = 16 * g16 + 4 * ak;
    uint4 r[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      r[kk] = make_uint4(0, 0, 0, 0);
      if (row0 + r0 + kk < nvalid) r[kk] = *reinterpret_cast<const uint4*>(src + (int64_t)(row0 + r0 + kk) * row_stride + 8 * dc);
    }
    const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&r[0]);
    const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&r[1]);
    const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&r[2]);
    const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&r[3]);
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

struct BwdAttnArgs {
  AttnArgs f;                  // q, k, v (+ strides); f.out unused
  const uint16_t* dout; int64_t do_row, do_head;
  uint16_t* dq; uint16_t* dk; uint16_t* dv;   // contiguous [T, H, D]
};

// pass A: one workgroup = 128 keys (32 per wave) of one (sequence, head); loops over query tiles of BQ rows
template <int D, int BQ>
__global__ void __launch_bounds__(256) hstu_bwd_kv_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr int RS = D + 8, TS = BQ + 8;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Qs = smem;               // [BQ][RS]
  uint16_t* dOs = Qs + BQ * RS;      // [BQ][RS]
  uint16_t* Qt = dOs + BQ * RS;      // [D][TS]
  uint16_t* dOt = Qt + D * TS;       // [D][TS]

  const int b = blockIdx.z, h = blockIdx.y;
  SeqInfo s;
  s.start = a.cu_seqlens[b];
  s.L = a.cu_seqlens[b + 1] - s.start;
  const int n0 = blockIdx.x * kBM;
  if (n0 >= s.L) return;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
  const int lane = lane_id(), wv = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int key0 = n0 + 32 * wv;
  const int kj = key0 + l31;
  const bool wave_live = key0 < s.L;

  // K / V fragments of the wave's 32 keys (B operands: lane = key, 8 consecutive d per 16-slice)
  bf16x8_t kf[D / 16], vf[D / 16];
  {
    const int64_t tok = s.start + (kj < s.L ? kj : 0);
    const uint16_t* kp = a.k + tok * a.k_row + (int64_t)h * a.k_head + 8 * hi;
    const uint16_t* vp = a.v + tok * a.v_row + (int64_t)h * a.v_head + 8 * hi;
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
      uint4 t0 = make_uint4(0, 0, 0, 0), t1 = make_uint4(0, 0, 0, 0);
      if (kj < s.L) { t0 = *reinterpret_cast<const uint4*>(kp + 16 * sl); t1 = *reinterpret_cast<const uint4*>(vp + 16 * sl); }
      kf[sl] = *reinterpret_cast<bf16x8_t*>(&t0);
      vf[sl] = *reinterpret_cast<bf16x8_t*>(&t1);
    }
  }
  f32x16_t acc_dv[D / 32], acc_dk[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_dv[dt][r] = 0.f; acc_dk[dt][r] = 0.f; }

  const uint16_t* qbase = a.q + (int64_t)s.start * a.q_row + (int64_t)h * a.q_head;
  const uint16_t* dobase = g.dout + (int64_t)s.start * g.do_row + (int64_t)h * g.do_head;
  const int blk_last_key = n0 + kBM - 1 < s.L - 1 ? n0 + kBM - 1 : s.L - 1;
  for (int i0 = 0; i0 < s.L; i0 += BQ) {
    // does any row of this query tile see any key of the block?
    const int i1 = i0 + BQ - 1 < s.L - 1 ? i0 + BQ - 1 : s.L - 1;
    bool reach = true;
    if (a.causal) reach = (i1 >= n0) || (s.has_ctx && i0 < s.c && n0 < s.hlen);
    if (!reach) continue;   // block-uniform
    __syncthreads();
    stage_rows<D, BQ>(Qs, qbase, a.q_row, i0, s.L);
    stage_rows<D, BQ>(dOs, dobase, g.do_row, i0, s.L);
    stage_transposed<D, BQ>(Qt, qbase, a.q_row, i0, s.L);
    stage_transposed<D, BQ>(dOt, dobase, g.do_row, i0, s.L);
    __syncthreads();
    if (!wave_live) continue;
    (void)blk_last_key;
    // GEMM 1 / 2: S[q x keys] = Q K^T, dP[q x keys] = dO V^T (A from LDS rows, B = register fragments)
    f32x16_t acc_s[BQ / 32], acc_p[BQ / 32];
#pragma unroll
    for (int t = 0; t < BQ / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_s[t][r] = 0.f; acc_p[t][r] = 0.f; }
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
#pragma unroll
      for (int t = 0; t < BQ / 32; ++t) {
        const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qs + (32 * t + l31) * RS + 16 * sl + 8 * hi);
        const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(dOs + (32 * t + l31) * RS + 16 * sl + 8 * hi);
        acc_s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[sl], acc_s[t], 0, 0, 0);
        acc_p[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[sl], acc_p[t], 0, 0, 0);
      }
    }
    // P and dS packed as B operands (k = query rows held in the registers, lane = key)
    bf16x8_t pf[BQ / 16], sf[BQ / 16];
#pragma unroll
    for (int t = 0; t < BQ / 32; ++t) {
      uint32_t pk[8], sk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float p2[2], s2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rr = r + u;
          const int qi = i0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          const float x = acc_s[t][rr] * a.alpha;
          const bool ok = attn_allowed(qi, kj, s, a.causal, a.group);
          p2[u] = ok ? silu_f(x) * a.inv_scale : 0.f;
          s2[u] = ok ? acc_p[t][rr] * dsilu_f(x) * (a.inv_scale * a.alpha) : 0.f;
        }
        pk[r >> 1] = pack_bf16(p2[0], p2[1]);
        sk[r >> 1] = pack_bf16(s2[0], s2[1]);
      }
      uint4 x0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), x1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      uint4 y0 = make_uint4(sk[0], sk[1], sk[2], sk[3]), y1 = make_uint4(sk[4], sk[5], sk[6], sk[7]);
      pf[2 * t] = *reinterpret_cast<bf16x8_t*>(&x0); pf[2 * t + 1] = *reinterpret_cast<bf16x8_t*>(&x1);
      sf[2 * t] = *reinterpret_cast<bf16x8_t*>(&y0); sf[2 * t + 1] = *reinterpret_cast<bf16x8_t*>(&y1);
    }
    // GEMM 3 / 4: dV^T[D x keys] += dO^T P, dK^T[D x keys] += Q^T dS
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
      for (int ks = 0; ks < BQ / 16; ++ks) {
        const bf16x8_t dot = *reinterpret_cast<const bf16x8_t*>(dOt + (32 * dt + l31) * TS + 16 * ks + 8 * hi);
        const bf16x8_t qt = *reinterpret_cast<const bf16x8_t*>(Qt + (32 * dt + l31) * TS + 16 * ks + 8 * hi);
        acc_dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf[ks], acc_dv[dt], 0, 0, 0);
        acc_dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, sf[ks], acc_dk[dt], 0, 0, 0);
      }
    }
  }
  if (kj < s.L) {
    uint16_t* dvp = g.dv + ((int64_t)(s.start + kj) * a.H + h) * D;
    uint16_t* dkp = g.dk + ((int64_t)(s.start + kj) * a.H + h) * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 o;
        o.x = pack_bf16(acc_dv[dt][4 * g4 + 0], acc_dv[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_dv[dt][4 * g4 + 2], acc_dv[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g4 + 4 * hi) = o;
        o.x = pack_bf16(acc_dk[dt][4 * g4 + 0], acc_dk[dt][4 * g4 + 1]);
        o.y = pack_bf16(acc_dk[dt][4 * g4 + 2], acc_dk[dt][4 * g4 + 3]);
        *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g4 + 4 * hi) = o;
      }
  }
}

// pass B: one workgroup = 128 queries (32 per wave); loops over key tiles of BK keys -> dQ
template <int D, int BK>
__global__ void __launch_bounds__(256) hstu_bwd_q_kernel(BwdAttnArgs g) {
  const AttnArgs& a = g.f;
  constexpr int RS = D + 8, TS = BK + 8;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Ks = smem;               // [BK][RS]
  uint16_t* Vs = Ks + BK * RS;       // [BK][RS]
  uint16_t* Kt = Vs + BK * RS;       // [D][TS]

  const int b = blockIdx.z, h = blockIdx.y;
  SeqInfo s;
  s.start = a.cu_seqlens[b];
  s.L = a.cu_seqlens[b + 1] - s.start;
  const int nblk = (s.L + kBM - 1) / kBM;
  if ((int)blockIdx.x >= nblk) return;
  const int m0 = (nblk - 1 - (int)blockIdx.x) * kBM;
  s.has_ctx = a.num_contexts != nullptr;
  s.has_tgt = a.num_targets != nullptr;
  s.c = s.has_ctx ? a.num_contexts[b] : 0;
  s.hlen = s.L - (s.has_tgt ? a.num_targets[b] : 0);
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

  bf16x8_t qf[D / 16], dof[D / 16];
  {
    const int64_t tok = s.start + (qi < s.L ? qi : 0);
    const uint16_t* qp = a.q + tok * a.q_row + (int64_t)h * a.q_head + 8 * hi;
    const uint16_t* dp = g.dout + tok * g.do_row + (int64_t)h * g.do_head + 8 * hi;
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
      uint4 t0 = make_uint4(0, 0, 0, 0), t1 = make_uint4(0, 0, 0, 0);
      if (qi < s.L) { t0 = *reinterpret_cast<const uint4*>(qp + 16 * sl); t1 = *reinterpret_cast<const uint4*>(dp + 16 * sl); }
      qf[sl] = *reinterpret_cast<bf16x8_t*>(&t0);
      dof[sl] = *reinterpret_cast<bf16x8_t*>(&t1);
    }
  }
  f32x16_t acc_dq[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dq[dt][r] = 0.f;

  const uint16_t* kbase = a.k + (int64_t)s.start * a.k_row + (int64_t)h * a.k_head;
  const uint16_t* vbase = a.v + (int64_t)s.start * a.v_row + (int64_t)h * a.v_head;
  for (int n0 = 0; n0 < n_end; n0 += BK) {
    __syncthreads();
    stage_rows<D, BK>(Ks, kbase, a.k_row, n0, s.L);
    stage_rows<D, BK>(Vs, vbase, a.v_row, n0, s.L);
    stage_transposed<D, BK>(Kt, kbase, a.k_row, n0, s.L);
    __syncthreads();
    if (!wave_live || n0 >= w_end) continue;
    f32x16_t acc_s[BK / 32], acc_p[BK / 32];
#pragma unroll
    for (int t = 0; t < BK / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_s[t][r] = 0.f; acc_p[t][r] = 0.f; }
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
#pragma unroll
      for (int t = 0; t < BK / 32; ++t) {
        const bf16x8_t ka = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + l31) * RS + 16 * sl + 8 * hi);
        const bf16x8_t va = *reinterpret_cast<const bf16x8_t*>(Vs + (32 * t + l31) * RS + 16 * sl + 8 * hi);
        acc_s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[sl], acc_s[t], 0, 0, 0);    // S^T[keys x q]
        acc_p[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[sl], acc_p[t], 0, 0, 0);   // dP^T[keys x q]
      }
    }
    bf16x8_t sf[BK / 16];
#pragma unroll
    for (int t = 0; t < BK / 32; ++t) {
      uint32_t sk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float s2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rr = r + u;
          const int key = n0 + 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          const float x = acc_s[t][rr] * a.alpha;
          s2[u] = attn_allowed(qi, key, s, a.causal, a.group) ? acc_p[t][rr] * dsilu_f(x) * (a.inv_scale * a.alpha) : 0.f;
        }
        sk[r >> 1] = pack_bf16(s2[0], s2[1]);
      }
      uint4 y0 = make_uint4(sk[0], sk[1], sk[2], sk[3]), y1 = make_uint4(sk[4], sk[5], sk[6], sk[7]);
      sf[2 * t] = *reinterpret_cast<bf16x8_t*>(&y0); sf[2 * t + 1] = *reinterpret_cast<bf16x8_t*>(&y1);
    }
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const bf16x8_t kt = *reinterpret_cast<const bf16x8_t*>(Kt + (32 * dt + l31) * TS + 16 * ks + 8 * hi);
        acc_dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, sf[ks], acc_dq[dt], 0, 0, 0);  // dQ^T[D x q]
      }
    }
  }
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

template <int D>
static int launch_bwd(const BwdAttnArgs& g, int B, int max_seqlen, hipStream_t stream) {
  constexpr int BQ = D >= 256 ? 32 : 64;
  constexpr int BK = 64;
  const size_t smem_kv = (size_t)(2 * BQ * (D + 8) + 2 * D * (BQ + 8)) * sizeof(uint16_t);
  const size_t smem_q = (size_t)(2 * BK * (D + 8) + D * (BK + 8)) * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_kv_kernel<D, BQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_bwd_q_kernel<D, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q);
    attr_set = true;
  }
  dim3 grid((max_seqlen + kBM - 1) / kBM, g.f.H, B);
  hipLaunchKernelGGL((hstu_bwd_kv_kernel<D, BQ>), grid, dim3(256), smem_kv, stream, g);
  hipLaunchKernelGGL((hstu_bwd_q_kernel<D, BK>), grid, dim3(256), smem_q, stream, g);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <int D>
static int launch_fwd(const AttnArgs& a, int B, int max_seqlen, hipStream_t stream) {
  const size_t smem = (size_t)(kBN * (D + 8) + D * (kBN + 8) + (D >= 256 ? kBM * (D + 8) : 0)) * sizeof(uint16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(hstu_fwd_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid((max_seqlen + kBM - 1) / kBM, a.H, B);
  hipLaunchKernelGGL(hstu_fwd_kernel<D>, grid, dim3(256), smem, stream, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

// hstu_varlen_fwd (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:335-523).  q, k, v, out: bf16 [total, H, d] with
// explicit token / head strides (elements); cu_seqlens int32 [B+1] shared by q and k (self attention over
// jagged sequences); num_contexts / num_targets int32 [B] or NULL; window (-1, 0) = causal, (-1, -1) = full.
int mi355_hstu_attn_fwd(const void* q, const void* k, const void* v, void* out, int64_t q_row_stride,
                        int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int64_t q_head_stride,
                        int64_t k_head_stride, int64_t v_head_stride, int64_t o_head_stride, const int32_t* cu_seqlens,
                        int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                        const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size, int causal,
                        float alpha, float scaling_seqlen, hipStream_t stream) {
  MI355_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128 || head_dim == 256,
                  "head_dim must be one of 32, 64, 128, 256 (hstu_api.cpp:391)");
  MI355_CHECK_ARG(target_group_size >= 1, "target_group_size must be >= 1");
  MI355_CHECK_ARG(causal || (!num_contexts && !num_targets), "contextual / target masks require causal attention");
  MI355_CHECK_ARG(scaling_seqlen > 0.f, "scaling_seqlen must be positive");
  MI355_CHECK_ARG(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0 &&
                      q_head_stride % 8 == 0 && k_head_stride % 8 == 0 && v_head_stride % 8 == 0 && o_head_stride % 4 == 0,
                  "q/k/v strides must be multiples of 8 elements (16-byte rows)");
  if (batch == 0 || max_seqlen == 0) return MI355_OK;
  AttnArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = (uint16_t*)out;
  a.q_row = q_row_stride; a.k_row = k_row_stride; a.v_row = v_row_stride; a.o_row = o_row_stride;
  a.q_head = q_head_stride; a.k_head = k_head_stride; a.v_head = v_head_stride; a.o_head = o_head_stride;
  a.cu_seqlens = cu_seqlens; a.num_contexts = num_contexts; a.num_targets = num_targets;
  a.H = (int)num_heads; a.causal = causal; a.group = (int)target_group_size;
  a.alpha = alpha; a.inv_scale = 1.0f / scaling_seqlen;
  switch (head_dim) {
    case 32: return launch_fwd<32>(a, (int)batch, (int)max_seqlen, stream);
    case 64: return launch_fwd<64>(a, (int)batch, (int)max_seqlen, stream);
    case 128: return launch_fwd<128>(a, (int)batch, (int)max_seqlen, stream);
    default: return launch_fwd<256>(a, (int)batch, (int)max_seqlen, stream);
  }
}

int64_t mi355_hstu_attn_bwd_workspace_bytes(int64_t total_tokens, int64_t num_heads, int64_t head_dim) {
  (void)total_tokens; (void)num_heads; (void)head_dim;
  return 0;  // the two-pass backward needs no scratch (no fp32 dQ accumulator, no atomics)
}

// hstu_varlen_bwd (corelib/hstu/csrc/hstu_attn/hstu_api.cpp:525-719).  dq, dk, dv: contiguous bf16 [total, H, d].
int mi355_hstu_attn_bwd(const void* dout, const void* q, const void* k, const void* v, void* dq, void* dk, void* dv,
                        int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t do_row_stride,
                        int64_t q_head_stride, int64_t k_head_stride, int64_t v_head_stride, int64_t do_head_stride,
                        const int32_t* cu_seqlens, int64_t batch, int64_t num_heads, int64_t head_dim, int64_t max_seqlen,
                        const int32_t* num_contexts, const int32_t* num_targets, int64_t target_group_size, int causal,
                        float alpha, float scaling_seqlen, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  (void)workspace; (void)workspace_bytes;
  MI355_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128 || head_dim == 256,
                  "head_dim must be one of 32, 64, 128, 256 (hstu_api.cpp:391)");
  MI355_CHECK_ARG(target_group_size >= 1, "target_group_size must be >= 1");
  MI355_CHECK_ARG(causal || (!num_contexts && !num_targets), "contextual / target masks require causal attention");
  MI355_CHECK_ARG(scaling_seqlen > 0.f, "scaling_seqlen must be positive");
  MI355_CHECK_ARG(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && do_row_stride % 8 == 0 &&
                      q_head_stride % 8 == 0 && k_head_stride % 8 == 0 && v_head_stride % 8 == 0 && do_head_stride % 8 == 0,
                  "q/k/v/dout strides must be multiples of 8 elements (16-byte rows)");
  if (batch == 0 || max_seqlen == 0) return MI355_OK;
  BwdAttnArgs g;
  AttnArgs& a = g.f;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = nullptr;
  a.q_row = q_row_stride; a.k_row = k_row_stride; a.v_row = v_row_stride; a.o_row = 0;
  a.q_head = q_head_stride; a.k_head = k_head_stride; a.v_head = v_head_stride; a.o_head = 0;
  a.cu_seqlens = cu_seqlens; a.num_contexts = num_contexts; a.num_targets = num_targets;
  a.H = (int)num_heads; a.causal = causal; a.group = (int)target_group_size;
  a.alpha = alpha; a.inv_scale = 1.0f / scaling_seqlen;
  g.dout = (const uint16_t*)dout; g.do_row = do_row_stride; g.do_head = do_head_stride;
  g.dq = (uint16_t*)dq; g.dk = (uint16_t*)dk; g.dv = (uint16_t*)dv;
  switch (head_dim) {
    case 32: return launch_bwd<32>(g, (int)batch, (int)max_seqlen, stream);
    case 64: return launch_bwd<64>(g, (int)batch, (int)max_seqlen, stream);
    case 128: return launch_bwd<128>(g, (int)batch, (int)max_seqlen, stream);
    default: return launch_bwd<256>(g, (int)batch, (int)max_seqlen, stream);
  }
}

}  // extern "C"
