// Fused multi-head attention for the gst-visdial hot path on gfx950 (wave64, MFMA 16x16):
//   text self-attention (12h x 64, T<=256), vision self-attention (8h x 128, 37 regions), the two
//   directions of the co-attention connection layers (8h x 128, T x 37 and 37 x T), decoder causal
//   self-attention (25 x 25) and decoder cross-attention (25 x (37+T)).
// Reference arithmetic: softmax(Q K^T / sqrt(d) + additive mask) -> dropout -> V
//   (models/vilbert_dialog.py:389-405, 516-532, 670-710; transformers 4.16.2 BertSelfAttention).
//
// Design (all three kernels):
//   * one wave owns a 16-row tile; keys (forward, dQ) or queries (dK/dV) stream through LDS in chunks
//     of 64 rows, staged with 16-byte vector loads straight from the (possibly fused QKV) activations;
//   * the first product is issued so that the *contraction index of the second product* lands on the
//     accumulator rows: forward/dQ compute S^T = K Q^T (lane: one query column, 4 consecutive keys),
//     dK/dV computes S = Q K^T (lane: one key column, 4 consecutive queries).  The exponentiated tile
//     is therefore already a legal MFMA B operand (k = 4*(lane>>4)+j) and never visits LDS;
//   * the second product reads its A operand (V^T, K^T, dO^T, Q^T) with ds_read_b64_tr_b16 from the
//     row-major LDS image, i.e. the transpose is free;
//   * softmax statistics: online max/sum per query column in registers, updated once per 64-key chunk (d <= 64) or 32 keys
//     (d = 128) with two v_permlane swaps; forward saves only LSE; backward recomputes P = exp(s - LSE) and regenerates the
//     dropout mask;
//   * the kernels are bound by VALU issue and dependent-latency chains, not by MFMA or memory (rocprofv3 SQ counters of the
//     text shape: VALU active 0.21 of wave cycles x 3 waves per SIMD, MFMA pipe busy < 0.15, no LDS bank conflicts): tiles
//     are processed several at a time (independent first products in flight, one statistics update, pairs of tiles on one
//     16x16x32 second product) and the dropout draws use 32-bit index arithmetic when the launch allows it;
//   * fp32 parity mode runs the same skeleton on v_mfma_f32_16x16x4_f32.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

template <typename T, int D> struct Img {
  static constexpr bool BF = sizeof(T) == 2;
  static constexpr int RB = BF ? D * 2 : (D + 4) * 4;        // row bytes
  static constexpr int BYTES = 64 * RB;                       // 64-row chunk
  // bf16 row image: 16-byte slot s of row r at slot s ^ (r & mask)
  static DEVFN int row_off(int row, int slot) {
    constexpr int NS = D / 8, MASK = (NS < 16 ? NS : 16) - 1;
    return row * RB + ((slot ^ (row & MASK)) << 4);
  }
  // bf16 transposed-read image: 32-byte block b of row r at block b ^ (r / rows_per_bank_row)
  static DEVFN int tr_off(int row, int col) {
    constexpr int NB = D / 16, RPB = 128 / D >= 1 ? 128 / D : 1;
    int blk = ((col >> 4) ^ (row / RPB)) & (NB - 1);
    return row * RB + (blk << 5) + ((col & 15) << 1);
  }
  static DEVFN int f32_off(int row, int col) { return row * RB + col * 4; }
};

// copy rows [r0, r0+64) x D of a [rows, ld] matrix into LDS image(s); rows >= rmax are zero filled
template <typename T, int D>
DEVFN void stage64(const T* g, int64_t ld, int r0, int rmax, char* img_row, char* img_tr, int tid) {
  constexpr int VE = 16 / sizeof(T), VPR = D / VE, TOT = 64 * VPR;
  for (int v = tid; v < TOT; v += 256) {
    const int row = v / VPR, cv = v % VPR;
    u32x4 z = {0u, 0u, 0u, 0u};
    if (r0 + row < rmax) z = *(const u32x4*)(g + (int64_t)(r0 + row) * ld + cv * VE);
    if (Img<T, D>::BF) {
      if (img_row) *(u32x4*)(img_row + Img<T, D>::row_off(row, cv)) = z;
      if (img_tr) *(u32x4*)(img_tr + Img<T, D>::tr_off(row, cv * 8)) = z;
    } else {
      *(u32x4*)(img_row + row * Img<T, D>::RB + cv * 16) = z;
    }
  }
}

// Two-phase form of stage64 for software pipelining: `load` issues the chunk's global loads into registers (they stay in
// flight while the previous chunk is being consumed), `store` writes them into the LDS image(s) after the barrier.
template <typename T, int D> struct Stage64 {
  static constexpr int VE = 16 / sizeof(T), VPR = D / VE, TOT = 64 * VPR, NV = TOT / 256;
  static_assert(TOT % 256 == 0, "chunk must split evenly over 256 threads");
  u32x4 v[NV];
  DEVFN void load(const T* g, int64_t ld, int r0, int rmax, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256, row = idx / VPR, cv = idx % VPR;
      v[i] = (u32x4){0u, 0u, 0u, 0u};
      if (r0 + row < rmax) v[i] = *(const u32x4*)(g + (int64_t)(r0 + row) * ld + cv * VE);
    }
  }
  DEVFN void store(char* img_row, char* img_tr, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256, row = idx / VPR, cv = idx % VPR;
      if (Img<T, D>::BF) {
        if (img_row) *(u32x4*)(img_row + Img<T, D>::row_off(row, cv)) = v[i];
        if (img_tr) *(u32x4*)(img_tr + Img<T, D>::tr_off(row, cv * 8)) = v[i];
      } else {
        *(u32x4*)(img_row + row * Img<T, D>::RB + cv * 16) = v[i];
      }
    }
  }
};

// per-lane register copy of one row of a [rows, ld] matrix laid out as the MFMA operand that contracts over d:
//   bf16: NF = D/32 fragments of 8 (d = kk*32 + 8g + j);  f32: NF = D/4 scalars (d = g*(D/4) + ks)
template <typename T, int D> struct RowFrag;
template <int D> struct RowFrag<bf16, D> {
  static constexpr int NF = D / 32;
  bf16x8 f[NF];
  DEVFN void load(const bf16* rowp, bool valid, int g) {
#pragma unroll
    for (int kk = 0; kk < NF; ++kk) {
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      f[kk] = valid ? *(const bf16x8*)(rowp + kk * 32 + 8 * g) : __builtin_bit_cast(bf16x8, z);
    }
  }
  DEVFN float dot(const bf16* rowp, bool valid, int g) const {   // sum_j f * other (same positions)
    float a = 0.f;
    if (valid) {
#pragma unroll
      for (int kk = 0; kk < NF; ++kk) {
        bf16x8 o = *(const bf16x8*)(rowp + kk * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) a += (float)f[kk][j] * (float)o[j];
      }
    }
    return a;
  }
};
template <int D> struct RowFrag<float, D> {
  static constexpr int NF = D / 4;
  float f[NF];
  DEVFN void load(const float* rowp, bool valid, int g) {
#pragma unroll
    for (int v = 0; v < NF / 4; ++v) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (valid) t = *(const f32x4*)(rowp + g * NF + v * 4);
      f[v * 4 + 0] = t[0]; f[v * 4 + 1] = t[1]; f[v * 4 + 2] = t[2]; f[v * 4 + 3] = t[3];
    }
  }
  DEVFN float dot(const float* rowp, bool valid, int g) const {
    float a = 0.f;
    if (valid) {
#pragma unroll
      for (int ks = 0; ks < NF; ++ks) a += f[ks] * rowp[g * NF + ks];
    }
    return a;
  }
};

// first product: acc(16x16) = sum_d Arow[x = xb + (lane&15)][d] * frag[d]  (A from the LDS row image)
template <typename T, int D>
DEVFN f32x4 first_product(const char* img_row, int xb, const RowFrag<T, D>& fr, int lane) {
  const int g = lane >> 4, li = lane & 15;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) {
      bf16x8 a = *(const bf16x8*)(img_row + Img<T, D>::row_off(xb + li, kk * 4 + g));
      acc = mfma_bf16_k32(a, fr.f[kk], acc);
    }
  } else {
    constexpr int NF = D / 4;
#pragma unroll
    for (int ks = 0; ks < NF; ++ks) {
      float a = *(const float*)(img_row + Img<T, D>::f32_off(xb + li, g * NF + ks));
      acc = mfma_f32_k4(a, fr.f[ks], acc);
    }
  }
  return acc;
}

// second product: acc[i](16x16) += sum_{r<4} X[row = xb + 4g + r][col = i*16 + (lane&15)] * w[r]
//   (A = X^T read transposed from the row-major image, B = the in-register tile w)
template <typename T, int D>
DEVFN void second_product(f32x4 (&acc)[D / 16], const char* img_tr, int xb, const float (&w)[4], int lane) {
  const int g = lane >> 4, li = lane & 15;
  if constexpr (sizeof(T) == 2) {
    const s16x4 b = pack_bf16x4(w[0], w[1], w[2], w[3]);
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
      s16x4 a = lds_tr16(img_tr + Img<T, D>::tr_off(xb + 4 * g + (li >> 2), i * 16 + 4 * (lane & 3)));
      acc[i] = mfma_bf16_k16(a, b, acc[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = *(const float*)(img_tr + Img<T, D>::f32_off(xb + 4 * g + r, i * 16 + li));
        acc[i] = mfma_f32_k4(a, w[r], acc[i]);
      }
    }
  }
}

DEVFN int round4(int x) { return (x + 3) & ~3; }

// The second product of TWO 16-row tiles (rows xb.. and xb+16..).  bf16: ONE 16x16x32 MFMA per 16 output columns -- k-slot 8g+j
// of the instruction is row 4g+j of the first tile (j < 4) or of the second (j >= 4), i.e. the A operand is the concatenation
// of the two transposed reads the tiles would have issued separately and the B operand that of their in-register weights.
template <typename T, int D>
DEVFN void second_product_pair(f32x4 (&acc)[D / 16], const char* img_tr, int xb, const float (&w0)[4], const float (&w1)[4], int lane) {
  if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, li = lane & 15;
    const bf16x8 b = {(bf16)w0[0], (bf16)w0[1], (bf16)w0[2], (bf16)w0[3], (bf16)w1[0], (bf16)w1[1], (bf16)w1[2], (bf16)w1[3]};
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
      const s16x4 lo = lds_tr16(img_tr + Img<T, D>::tr_off(xb + 4 * g + (li >> 2), i * 16 + 4 * (lane & 3)));
      const s16x4 hi = lds_tr16(img_tr + Img<T, D>::tr_off(xb + 16 + 4 * g + (li >> 2), i * 16 + 4 * (lane & 3)));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[i] = mfma_bf16_k32(__builtin_bit_cast(bf16x8, v), b, acc[i]);
    }
  } else {
    second_product<T, D>(acc, img_tr, xb, w0, lane);
    second_product<T, D>(acc, img_tr, xb + 16, w1, lane);
  }
}

// Dropout draws.  E32: the launch has fewer than 2^33 score elements, so the high word of every pair index is zero and its
// term of drop_draw (a quarter-rate integer multiply per draw) vanishes -- same stream, cheaper arithmetic.
template <bool E32> DEVFN uint32_t draw_pair(const DropKey& k, uint64_t e2) {
  if (E32) return drop_hash((uint32_t)e2, k.key);
  return drop_draw(k, e2);
}
template <bool E32> DEVFN f32x4 drop_factor4e(const DropKey& k, uint64_t e) {      // four consecutive elements, e % 4 == 0
  f32x4 f = {1.f, 1.f, 1.f, 1.f};
  if (!k.on) return f;
  const uint32_t r0 = draw_pair<E32>(k, e >> 1), r1 = draw_pair<E32>(k, (e >> 1) + 1);
  f[0] = (r0 & 0xffffu) >= k.thr ? k.scale : 0.f;
  f[1] = (r0 >> 16) >= k.thr ? k.scale : 0.f;
  f[2] = (r1 & 0xffffu) >= k.thr ? k.scale : 0.f;
  f[3] = (r1 >> 16) >= k.thr ? k.scale : 0.f;
  return f;
}

// The three bodies below work on a whole 64-row chunk at a time: the first products of its (up to four) 16x16 tiles are issued
// together, the element-wise part runs on 16 values per lane, the softmax statistics are updated ONCE per chunk, and the second
// products pair tiles on 16x16x32 MFMAs.  The kernels are bound by VALU issue (exp, the dropout hash, bf16 packing -- not by
// MFMA or memory), so per-tile bookkeeping (two cross-lane maxima, a ballot, the accumulator rescale with its AGPR<->VGPR
// copies, loop and address arithmetic) was half of the instruction stream when it ran once per 16 keys.

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int D, bool E32>
DEVFN void attn_fwd_body(const gstvd_attn_t& a, char* smem) {
  constexpr int TF = D <= 64 ? 4 : 2;                         // 16-key tiles per softmax update
  char* sK = smem;
  char* sV = smem + Img<T, D>::BYTES;
  float* smask = (float*)(smem + 2 * Img<T, D>::BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 64 + wave * 16 + li;
  const bool qv = q < a.Lq;
  const int bk = a.kv_group > 1 ? b / a.kv_group : b;       // batch row of the (possibly shared) keys / values
  const int64_t qbs = a.q_bstride > 0 ? a.q_bstride : a.Lq, kbs = a.kv_bstride > 0 ? a.kv_bstride : a.Lk;
  const T* Qb = (const T*)a.Q + (int64_t)b * qbs * a.ldq + h * D;
  const T* Kb = (const T*)a.K + (int64_t)bk * kbs * a.ldk + h * D;
  const T* Vb = (const T*)a.V + (int64_t)bk * kbs * a.ldv + h * D;
  RowFrag<T, D> qf;
  qf.load(Qb + (int64_t)q * a.ldq, qv, g);
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const uint64_t ebase = ((uint64_t)(b * a.nh + h) * a.Lq + (uint64_t)(qv ? q : 0)) * (uint64_t)Lkp + (uint64_t)(4 * g);
  // keep-bit output (gstvd_attn_t.drop_bits): this wave's 16-query tile is tile `qtile` of [B, nh, ceil(Lq/16), ceil(Lk/16), 4]
  const int nkt16 = (a.Lk + 15) >> 4;
  unsigned long long* kbits = (dk.on && a.kv_group <= 1 && (int)(blockIdx.x * 4 + wave) < ((a.Lq + 15) >> 4)) ? (unsigned long long*)a.drop_bits : nullptr;
  const int64_t qtile = ((int64_t)(b * a.nh + h) * ((a.Lq + 15) >> 4)) + (blockIdx.x * 4 + wave);     // (a wave past the last query tile writes nothing)

  float m_run = -1e30f, l_part = 0.f;
  f32x4 accO[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) accO[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The K / V / mask loads of a chunk are issued PF chunks ahead of its use (PF register stages).  PF = 2 was measured in
  // round 3 (d = 64, 154 registers): 18.1 us against 17.1 us for the text shape, 10.6 against 10.5 us for the cross-attention
  // shape -- the kernel is not waiting for these loads; one stage it stays.
  constexpr int PF = 1;
  struct Stage { Stage64<T, D> k, v; float m; };
  Stage st[PF];
  auto prefetch = [&](Stage& sg, int c0) {
    sg.k.load(Kb, a.ldk, c0, a.Lk, tid);
    sg.v.load(Vb, a.ldv, c0, a.Lk, tid);
    if (tid < 64) {      // additive mask term of the key: 0 (valid), mask_neg (masked out), -inf (past the end)
      const int key = c0 + tid;
      sg.m = -INFINITY;
      if (key < a.Lk) sg.m = (a.key_mask == nullptr || a.key_mask[(int64_t)bk * a.Lk + key] != 0.f) ? 0.f : a.mask_neg;
    }
  };
  auto chunk = [&](Stage& sg, const int c0) {
    __syncthreads();
    sg.k.store(sK, nullptr, tid);
    sg.v.store(Img<T, D>::BF ? nullptr : sV, Img<T, D>::BF ? sV : nullptr, tid);
    if (tid < 64) smask[tid] = sg.m;
    __syncthreads();
    if (c0 + 64 * PF < a.Lk) prefetch(sg, c0 + 64 * PF);
    const int ntile = (a.Lk - c0 + 15) / 16 < 4 ? (a.Lk - c0 + 15) / 16 : 4;
    // TF tiles per softmax update: the whole chunk for d <= 64, half of it for d = 128 (register budget of two waves per SIMD).
    // FULL: all TF tiles present and no causal mask -- the body then has no control flow at all (the generic form tests
    // `t < nt` around every tile, which the compiler turns into a scalar branch per tile: basic blocks of a few instructions,
    // nothing for the scheduler to interleave with the MFMAs).
    auto tiles = [&](auto full_tag, const int nt, const int k0) {
      constexpr bool FULL = decltype(full_tag)::value;
      f32x4 s[TF];
#pragma unroll
      for (int t = 0; t < TF; ++t)
        if (FULL || t < nt) s[t] = first_product<T, D>(sK, k0 + t * 16, qf, lane);
      float val[TF][4], mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < TF; ++t) {
        if (FULL || t < nt) {
          f32x4 madd = *(const f32x4*)(smask + k0 + t * 16 + 4 * g);
          if (!FULL && a.causal) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + k0 + t * 16 + 4 * g + r > q && madd[r] == 0.f) madd[r] = a.mask_neg;   // causal x padding: the term is added once
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            val[t][r] = s[t][r] * a.scale + madd[r];
            mx = fmaxf(mx, val[t][r]);
          }
        }
      }
      mx = rows_max(mx);
      const float m_new = fmaxf(m_run, mx);
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {       // the running max moved for some query of this wave
        const float alpha = __expf(m_run - m_new);
        l_part *= alpha;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) accO[i] *= alpha;
        m_run = m_new;
      }
      float pd[TF][4], ps = 0.f;
#pragma unroll
      for (int t = 0; t < TF; ++t) {
        if (FULL || t < nt) {
          const f32x4 fac = drop_factor4e<E32>(dk, ebase + (uint64_t)(c0 + k0 + t * 16));
          if (D == 64 && sizeof(T) == 2 && kbits != nullptr) {
            // the keep bits of this 16 x 16 tile for the one-pass backward: word r = ballot of "element (query li, key 4g + r) kept"
            const unsigned long long w0 = __builtin_amdgcn_ballot_w64(fac[0] != 0.f), w1 = __builtin_amdgcn_ballot_w64(fac[1] != 0.f);
            const unsigned long long w2 = __builtin_amdgcn_ballot_w64(fac[2] != 0.f), w3 = __builtin_amdgcn_ballot_w64(fac[3] != 0.f);
            if (lane < 4) kbits[((int64_t)qtile * nkt16 + ((c0 + k0) >> 4) + t) * 4 + lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(val[t][r] - m_new);
            ps += p;
            pd[t][r] = p * fac[r];
          }
        }
      }
      l_part += ps;
#pragma unroll
      for (int pr = 0; pr < TF / 2; ++pr) {
        if (FULL || 2 * pr + 1 < nt) second_product_pair<T, D>(accO, sV, k0 + 32 * pr, pd[2 * pr], pd[2 * pr + 1], lane);
        else if (2 * pr < nt) second_product<T, D>(accO, sV, k0 + 32 * pr, pd[2 * pr], lane);
      }
    };
#pragma unroll 1
    for (int hf = 0; hf < 4 / TF; ++hf) {
      const int nt = ntile - TF * hf, k0 = 16 * TF * hf;
      if (nt <= 0) break;
      if (TF == 4 && nt >= TF && !a.causal) tiles(std::true_type{}, TF, k0);
      else tiles(std::false_type{}, nt, k0);
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (64 * j < a.Lk) prefetch(st[j], 64 * j);
  for (int c0 = 0; c0 < a.Lk; c0 += 64 * PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (c0 + 64 * j < a.Lk) chunk(st[j], c0 + 64 * j);
  }
  const float l_tot = rows_sum(l_part);
  const float inv = 1.f / l_tot;
  if (qv) {
    T* Op = (T*)a.O + ((int64_t)b * qbs + q) * a.ldo + h * D;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) st4(Op + i * 16 + 4 * g, accO[i] * inv);
    if (g == 0 && a.LSE) a.LSE[((int64_t)b * a.nh + h) * a.Lq + q] = m_run + __logf(l_tot);
  }
}

// =====================================================================================================
// forward, ONE query per (row, head): the KV-cached decode step (Lq = 1, no dropout)
// =====================================================================================================
// The tiled kernel above would run one useful lane per workgroup and walk the keys chunk after chunk (10.6 us for the
// 293 keys of the cross-attention: five dependent load -> LDS -> compute rounds).  Here the workgroup's 256 threads split the
// KEYS: groups of lanes score one key each against the query (16-byte pieces of its K row), one block-wide max / sum, then the
// threads split (key group, output column) for P.V with coalesced V reads, eight loads in flight per thread -- two memory
// round trips in all, fp32 arithmetic.
constexpr int DEC_NT = 1024;
template <typename T, int D>
__global__ __launch_bounds__(DEC_NT) void attn_decode_kernel(gstvd_attn_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = (float*)smem;                                   // [D] the query
  float* sp = sq + D;                                         // [Lk] scores, then probabilities
  constexpr int NWV = DEC_NT / 64;
  __shared__ float sred[NWV];
  __shared__ float sacc[DEC_NT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int bk = a.kv_group > 1 ? b / a.kv_group : b;
  const int64_t qbs = a.q_bstride > 0 ? a.q_bstride : a.Lq, kbs = a.kv_bstride > 0 ? a.kv_bstride : a.Lk;
  const T* Qr = (const T*)a.Q + (int64_t)b * qbs * a.ldq + h * D;
  const T* Kb = (const T*)a.K + (int64_t)bk * kbs * a.ldk + h * D;
  const T* Vb = (const T*)a.V + (int64_t)bk * kbs * a.ldv + h * D;
  if (tid < D) sq[tid] = to_f(Qr[tid]);
  __syncthreads();
  // scores: PPK lanes share one key (a 16-byte piece of its K row each), a wave scores 64 / PPK keys per round, the four waves
  // interleave rounds; every lane has one load per round in flight and the rounds are independent (unrolled)
  constexpr int VE = 16 / sizeof(T), PPK = D / VE, KPW = 64 / PPK;
  const int kr = lane / PPK, piece = lane % PPK;
  float qreg[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) qreg[j] = sq[piece * VE + j];
  float mx = -INFINITY;
  constexpr int RND = 4;                                      // rounds per iteration: their loads are issued together
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type vec_t;
  for (int k0 = wave * KPW; k0 < a.Lk; k0 += RND * NWV * KPW) {
    vec_t kv[RND];
#pragma unroll
    for (int r = 0; r < RND; ++r) {
      const int key = k0 + r * NWV * KPW + kr;
      const int kc = key < a.Lk ? key : a.Lk - 1;             // (clamped: the value is discarded below)
      kv[r] = *(const vec_t*)(Kb + (int64_t)kc * a.ldk + piece * VE);
    }
#pragma unroll
    for (int r = 0; r < RND; ++r) {
      const int key = k0 + r * NWV * KPW + kr;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < VE; ++j) s += (float)kv[r][j] * qreg[j];
#pragma unroll
      for (int o = 1; o < PPK; o <<= 1) s += __shfl_xor(s, o, 64);
      if (key < a.Lk) {
        const float madd = (a.key_mask == nullptr || a.key_mask[(int64_t)bk * a.Lk + key] != 0.f) ? 0.f : a.mask_neg;
        s = s * a.scale + madd;
        if (piece == 0) sp[key] = s;
        mx = fmaxf(mx, s);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) sred[wave] = mx;
  __syncthreads();
  mx = sred[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, sred[w]);
  float sum = 0.f;
  for (int key = tid; key < a.Lk; key += DEC_NT) {
    const float pv = __expf(sp[key] - mx);
    sp[key] = pv;
    sum += pv;
  }
  sum = wave_sum(sum);
  __syncthreads();                                            // (sred reads above are done; sp[] writes below are visible)
  if (lane == 0) sred[wave] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) sum += sred[w];
  // P.V: thread = (key group kg, output column dcol); a wave reads 64 consecutive columns of a V row
  constexpr int NG = DEC_NT / D;                              // key groups: 16 (d = 64), 8 (d = 128), 32 (d = 32)
  const int dcol = tid % D, kg = tid / D;
  float acc = 0.f;
  {
    constexpr int UN = 4;                                     // four independent V loads in flight per thread
    float part[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) part[u] = 0.f;
    int key = kg;
    for (; key + (UN - 1) * NG < a.Lk; key += UN * NG) {
#pragma unroll
      for (int u = 0; u < UN; ++u) part[u] += sp[key + u * NG] * to_f(Vb[(int64_t)(key + u * NG) * a.ldv + dcol]);
    }
    for (; key < a.Lk; key += NG) acc += sp[key] * to_f(Vb[(int64_t)key * a.ldv + dcol]);
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += part[u];
  }
  sacc[tid] = acc;
  __syncthreads();
  if (tid < D) {
    float o = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2) o += sacc[g2 * D + tid];
    T* Op = (T*)a.O + (int64_t)b * qbs * a.ldo + h * D;
    Op[tid] = from_f<T>(o / sum);
    if (tid == 0 && a.LSE) a.LSE[((int64_t)b * a.nh + h) * a.Lq] = mx + __logf(sum);
  }
}

DEVFN bool attn_small_index_space(const gstvd_attn_t& a) {
  return (uint64_t)a.B * (uint64_t)a.nh * (uint64_t)a.Lq * (uint64_t)round4(a.Lk) < (1ull << 33);
}

// Occupancy: the grids of the step give 2-3 waves per SIMD (d = 64) and ~2 (d = 128); the register budgets below keep exactly
// that many resident (left to itself the compiler takes up to 440 registers for the chunk-wide bodies: one wave per SIMD).
template <typename T, int D>
__global__ __launch_bounds__(256, (D <= 64 ? 3 : 2)) void attn_fwd_kernel(gstvd_attn_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (sizeof(T) == 2) {
    if (attn_small_index_space(a)) { attn_fwd_body<T, D, true>(a, smem); return; }
  }
  attn_fwd_body<T, D, false>(a, smem);
}

// =====================================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O)); same tiling as forward
// =====================================================================================================
template <typename T, int D, bool E32>
DEVFN void attn_bwd_dq_body(const gstvd_attn_t& a, const int bx, const int h, const int b, char* smem) {
  constexpr bool BF = Img<T, D>::BF;
  constexpr int TP = D <= 64 ? 2 : 1;                         // 16-key tiles per inner iteration
  char* sKr = smem;                                           // row image of K
  char* sKt = BF ? smem + Img<T, D>::BYTES : smem;            // transposed-read image of K
  char* sVr = smem + (BF ? 2 : 1) * Img<T, D>::BYTES;         // row image of V
  float* smask = (float*)(smem + (BF ? 3 : 2) * Img<T, D>::BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int q = bx * 64 + wave * 16 + li;
  const bool qv = q < a.Lq;
  const T* Qb = (const T*)a.Q + (int64_t)b * a.Lq * a.ldq + h * D;
  const T* Kb = (const T*)a.K + (int64_t)b * a.Lk * a.ldk + h * D;
  const T* Vb = (const T*)a.V + (int64_t)b * a.Lk * a.ldv + h * D;
  const T* dOb = (const T*)a.dO + (int64_t)b * a.Lq * a.lddo + h * D;
  const T* Ob = (const T*)a.O + (int64_t)b * a.Lq * a.ldo + h * D;
  RowFrag<T, D> qf, dof;
  qf.load(Qb + (int64_t)q * a.ldq, qv, g);
  dof.load(dOb + (int64_t)q * a.lddo, qv, g);
  float delta = rows_sum(dof.dot(Ob + (int64_t)q * a.ldo, qv, g));
  const int64_t stat = ((int64_t)b * a.nh + h) * a.Lq + q;
  if (qv && g == 0) a.delta[stat] = delta;
  const float lse = qv ? a.LSE[stat] : INFINITY;              // +inf => p = 0 for padded query rows
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const uint64_t ebase = ((uint64_t)(b * a.nh + h) * a.Lq + (uint64_t)(qv ? q : 0)) * (uint64_t)Lkp + (uint64_t)(4 * g);

  f32x4 acc[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Stage64<T, D> pk, pv;
  float pm = -INFINITY;
  auto prefetch = [&](int c0) {
    pk.load(Kb, a.ldk, c0, a.Lk, tid);
    pv.load(Vb, a.ldv, c0, a.Lk, tid);
    if (tid < 64) {      // additive mask term of the key: 0 (valid), mask_neg (masked out), -inf (past the end)
      const int key = c0 + tid;
      pm = -INFINITY;
      if (key < a.Lk) pm = (a.key_mask == nullptr || a.key_mask[(int64_t)b * a.Lk + key] != 0.f) ? 0.f : a.mask_neg;
    }
  };
  prefetch(0);
  for (int c0 = 0; c0 < a.Lk; c0 += 64) {
    __syncthreads();
    pk.store(sKr, BF ? sKt : nullptr, tid);
    pv.store(sVr, nullptr, tid);
    if (tid < 64) smask[tid] = pm;
    __syncthreads();
    if (c0 + 64 < a.Lk) prefetch(c0 + 64);
    const int ntile = (a.Lk - c0 + 15) / 16 < 4 ? (a.Lk - c0 + 15) / 16 : 4;
    // TP tiles at a time.  d <= 64: two (32 keys) -- enough to pair them on the 16x16x32 second product and to halve the
    // per-tile bookkeeping, few enough live values for three waves per SIMD (a whole chunk at once needed 296 registers in the
    // merged kernel: one wave per SIMD, 105 us instead of 61 for the text shape).  d = 128: one -- the accumulators, operand
    // fragments and prefetch registers of that width leave no room for a second tile at two waves per SIMD.
    // FULL: all TP tiles present, no causal mask -- a body without control flow (see attn_fwd_body)
    auto tiles = [&](auto full_tag, const int pr, const int nt) {
      constexpr bool FULL = decltype(full_tag)::value;
      f32x4 s[TP], dp[TP];
      float ds[TP][4];
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        if (FULL || tt < nt) {
          s[tt] = first_product<T, D>(sKr, 16 * TP * pr + tt * 16, qf, lane);
          dp[tt] = first_product<T, D>(sVr, 16 * TP * pr + tt * 16, dof, lane);
        }
      }
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        if (FULL || tt < nt) {
          const int k0 = 16 * TP * pr + tt * 16;
          const f32x4 fac = drop_factor4e<E32>(dk, ebase + (uint64_t)(c0 + k0));
          f32x4 madd = *(const f32x4*)(smask + k0 + 4 * g);
          if (!FULL && a.causal) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + k0 + 4 * g + r > q && madd[r] == 0.f) madd[r] = a.mask_neg;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(s[tt][r] * a.scale + madd[r] - lse);
            ds[tt][r] = p * (dp[tt][r] * fac[r] - delta) * a.scale;
          }
        }
      }
      if (TP > 1 && (FULL || nt > 1)) second_product_pair<T, D>(acc, sKt, 16 * TP * pr, ds[0], ds[TP - 1], lane);
      else second_product<T, D>(acc, sKt, 16 * TP * pr, ds[0], lane);
    };
#pragma unroll 1
    for (int pr = 0; pr < 4 / TP; ++pr) {
      const int nt = ntile - TP * pr;
      if (nt <= 0) break;
      if (TP > 1 && nt >= TP && !a.causal) tiles(std::true_type{}, pr, TP);      // (TP = 1: the generic body has no tile tests)
      else tiles(std::false_type{}, pr, nt);
    }
  }
  if (qv) {
    T* dQp = (T*)a.dQ + ((int64_t)b * a.Lq + q) * a.lddq + h * D;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) st4(dQp + i * 16 + 4 * g, acc[i]);
  }
}

// =====================================================================================================
// backward, part 2: dK and dV; one wave owns 16 keys, queries stream through LDS
// =====================================================================================================
template <typename T, int D, bool E32>
DEVFN void attn_bwd_dkv_body(const gstvd_attn_t& a, const int bx, const int h, const int b, char* smem) {
  constexpr bool BF = Img<T, D>::BF;
  constexpr int IB = Img<T, D>::BYTES;
  constexpr int TP = D <= 64 ? 2 : 1;                         // 16-query tiles per inner iteration
  char* sQr = smem;
  char* sQt = BF ? smem + IB : smem;
  char* sOr = smem + (BF ? 2 : 1) * IB;                       // dO row image
  char* sOt = BF ? smem + 3 * IB : sOr;                       // dO transposed-read image
  float* sLse = (float*)(smem + (BF ? 4 : 2) * IB);
  float* sDel = sLse + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int key = bx * 64 + wave * 16 + li;
  const bool kv = key < a.Lk;
  const T* Ob = (const T*)a.O + (int64_t)b * a.Lq * a.ldo + h * D;
  const T* Qb = (const T*)a.Q + (int64_t)b * a.Lq * a.ldq + h * D;
  const T* Kb = (const T*)a.K + (int64_t)b * a.Lk * a.ldk + h * D;
  const T* Vb = (const T*)a.V + (int64_t)b * a.Lk * a.ldv + h * D;
  const T* dOb = (const T*)a.dO + (int64_t)b * a.Lq * a.lddo + h * D;
  RowFrag<T, D> kf, vf;
  kf.load(Kb + (int64_t)key * a.ldk, kv, g);
  vf.load(Vb + (int64_t)key * a.ldv, kv, g);
  const bool kmasked = kv && a.key_mask != nullptr && a.key_mask[(int64_t)b * a.Lk + key] == 0.f;
  const float kadd = kv ? (kmasked ? a.mask_neg : 0.f) : -INFINITY;      // additive term of this lane's key (-inf: past the end => p = 0)
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const int64_t stat0 = ((int64_t)b * a.nh + h) * a.Lq;
  // pair index (element index >> 1) of (query 4g of the chunk at c0 = 0, this lane's key); Lkp is even, so a step of one query
  // is a step of Lkp / 2 pairs.  The two keys of a pair sit in neighbouring lanes (li, li ^ 1): the even lane draws for rows
  // r = 0, 1 of a tile, the odd lane for rows 2, 3, and one quad permute hands each its partner's draws.
  const uint64_t half = (uint64_t)(Lkp >> 1);
  const uint64_t e2lane = ((uint64_t)stat0 + (uint64_t)(4 * g)) * half + (uint64_t)(key >> 1);        // not clamped for keys past the end: the partner lane may be a valid key and takes our draws
  const bool odd = (key & 1) != 0;

  f32x4 accK[D / 16], accV[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) accK[i] = accV[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // delta[q] = rowsum(dO * O) is recomputed here for every query chunk (four threads per query row, a quarter of the head
  // dimension each) instead of being read from the dQ half: the two halves of the backward then share no data and run as ONE
  // launch, side by side (they used to be two dependent launches).
  Stage64<T, D> pq, po;
  float plse = INFINITY, pdel = 0.f;
  auto prefetch = [&](int c0) {
    pq.load(Qb, a.ldq, c0, a.Lq, tid);
    po.load(dOb, a.lddo, c0, a.Lq, tid);
    const int qq = c0 + (tid >> 2), part = tid & 3;
    float dsum = 0.f;
    if (qq < a.Lq) {
      const T* dr = dOb + (int64_t)qq * a.lddo + part * (D / 4);
      const T* orow = Ob + (int64_t)qq * a.ldo + part * (D / 4);
#pragma unroll
      for (int e = 0; e < D / 4; e += 4) {
        const f32x4 x = ld4(dr + e), y = ld4(orow + e);
        dsum += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      }
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    pdel = dsum;                                              // valid in the lanes with part == 0
    if (tid < 64) {
      const int q1 = c0 + tid;
      plse = q1 < a.Lq ? a.LSE[stat0 + q1] : INFINITY;        // +inf => p = 0 for padded query rows
    }
  };
  prefetch(0);
  for (int c0 = 0; c0 < a.Lq; c0 += 64) {
    __syncthreads();
    pq.store(sQr, BF ? sQt : nullptr, tid);
    po.store(sOr, BF ? sOt : nullptr, tid);
    if (tid < 64) sLse[tid] = plse;
    if ((tid & 3) == 0) sDel[tid >> 2] = pdel;
    __syncthreads();
    if (c0 + 64 < a.Lq) prefetch(c0 + 64);
    const int ntile = (a.Lq - c0 + 15) / 16 < 4 ? (a.Lq - c0 + 15) / 16 : 4;
    const uint64_t e2chunk = e2lane + (uint64_t)c0 * half;
    auto tiles = [&](auto full_tag, const int pr, const int nt) {
      constexpr bool FULL = decltype(full_tag)::value;
      f32x4 s[TP], dp[TP];
      float pd[TP][4], ds[TP][4];
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        if (FULL || tt < nt) {
          s[tt] = first_product<T, D>(sQr, 16 * TP * pr + tt * 16, kf, lane);     // [q = 4g+r][key = li]
          dp[tt] = first_product<T, D>(sOr, 16 * TP * pr + tt * 16, vf, lane);
        }
      }
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        if (FULL || tt < nt) {
          const int q0 = 16 * TP * pr + tt * 16;
          const f32x4 lse4 = *(const f32x4*)(sLse + q0 + 4 * g);
          const f32x4 del4 = *(const f32x4*)(sDel + q0 + 4 * g);
          float f[4] = {1.f, 1.f, 1.f, 1.f};
          if (dk.on) {
            const int r0 = odd ? 2 : 0;
            const uint32_t mine0 = draw_pair<E32>(dk, e2chunk + (uint64_t)(q0 + r0) * half);
            const uint32_t mine1 = draw_pair<E32>(dk, e2chunk + (uint64_t)(q0 + r0 + 1) * half);
            const uint32_t other0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine0, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
            const uint32_t other1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine1, 0xB1, 0xf, 0xf, true);
            const uint32_t d0 = odd ? other0 : mine0, d1 = odd ? other1 : mine1, d2 = odd ? mine0 : other0, d3 = odd ? mine1 : other1;
            const uint32_t sh = odd ? 16u : 0u;
            f[0] = ((d0 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
            f[1] = ((d1 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
            f[2] = ((d2 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
            f[3] = ((d3 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float add = kadd;
            if (!FULL && a.causal && key > c0 + q0 + 4 * g + r && add == 0.f) add = a.mask_neg;
            const float p = __expf(s[tt][r] * a.scale + add - lse4[r]);
            pd[tt][r] = p * f[r];
            ds[tt][r] = p * (dp[tt][r] * f[r] - del4[r]) * a.scale;
          }
        }
      }
      if (TP > 1 && (FULL || nt > 1)) {
        second_product_pair<T, D>(accV, sOt, 16 * TP * pr, pd[0], pd[TP - 1], lane);
        second_product_pair<T, D>(accK, sQt, 16 * TP * pr, ds[0], ds[TP - 1], lane);
      } else {
        second_product<T, D>(accV, sOt, 16 * TP * pr, pd[0], lane);
        second_product<T, D>(accK, sQt, 16 * TP * pr, ds[0], lane);
      }
    };
#pragma unroll 1
    for (int pr = 0; pr < 4 / TP; ++pr) {
      const int nt = ntile - TP * pr;
      if (nt <= 0) break;
      if (TP > 1 && nt >= TP && !a.causal) tiles(std::true_type{}, pr, TP);      // (TP = 1: the generic body has no tile tests)
      else tiles(std::false_type{}, pr, nt);
    }
  }
  if (kv) {
    T* dKp = (T*)a.dK + ((int64_t)b * a.Lk + key) * a.lddk + h * D;
    T* dVp = (T*)a.dV + ((int64_t)b * a.Lk + key) * a.lddv + h * D;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
      st4(dKp + i * 16 + 4 * g, accK[i]);
      st4(dVp + i * 16 + 4 * g, accV[i]);
    }
  }
}

// =====================================================================================================
// backward in ONE pass: bf16, d = 64, up to 256 keys, no causal mask (the text self-attention, vilbert_dialog.py:389-405)
// =====================================================================================================
// The two-part backward above computes S = Q K^T, dP = dO V^T, the exponentials and the dropout draws TWICE -- once in the blocks
// that own queries (dQ) and once in the blocks that own keys (dK, dV): 7 products and 2 x the element-wise work for 5 products'
// worth of results, and at L = 256 that element-wise work (exp, the counter hash, packing) is what the kernel is made of.
// Here ONE workgroup owns a (batch row, head): 16 waves x 16 keys.  A wave keeps the K and V rows of its keys as register
// fragments and dK^T / dV^T of its keys in accumulators (no sum across waves), and walks the queries in chunks of 64 exactly like
// attn_bwd_dkv_body (same lane layout, same pair-shared dropout draws).  What is new: the wave also writes its dS tile -- key on
// the row, four consecutive queries per lane -- into an LDS image [256 keys][64 queries]; after a barrier the 16 waves each take
// one 16 (d) x 16 (query) tile of dQ^T = K^T dS^T for the chunk (A = transposed reads of the resident K image, B = transposed
// reads of the dS image, 8 MFMAs deep) and store it: dS crosses LDS once, dQ needs no atomics and no second launch.
// LDS: Q and dO chunk images (row + transposed-read, 4 x 8 KB), K (32 KB), dS (32 KB), LSE / delta of all queries (8 KB): 104 KB;
// one workgroup per CU.
constexpr int ONEPASS_MAX_LQ = 1024;
template <bool E32, bool BITS>      // BITS: the dropout keep bits come from forward (gstvd_attn_t.drop_bits) instead of the counter hash
__global__ __launch_bounds__(1024) void attn_bwd_onepass_kernel(gstvd_attn_t a) {
  typedef bf16 T;
  constexpr int D = 64, IB = Img<T, D>::BYTES, TP = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQr = smem;
  char* sQt = smem + IB;
  char* sOr = smem + 2 * IB;
  char* sOt = smem + 3 * IB;
  char* sKt = smem + 4 * IB;                                  // [256 keys][64 d], transposed-read image
  char* sDs = smem + 8 * IB;                                  // [256 keys][64 queries of the chunk], transposed-read image
  float* sLse = (float*)(smem + 12 * IB);                     // [ONEPASS_MAX_LQ] LSE of every query (+inf past the end)
  float* sDel = sLse + ONEPASS_MAX_LQ;                        // [ONEPASS_MAX_LQ] delta of every query
  unsigned long long* sBits = (unsigned long long*)(sDel + ONEPASS_MAX_LQ);   // [4 query tiles][16 key tiles][4] keep bits of the chunk
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, b = blockIdx.y;
  const int key = wave * 16 + li;
  const bool kv = key < a.Lk;
  const T* Ob = (const T*)a.O + (int64_t)b * a.Lq * a.ldo + h * D;
  const T* Qb = (const T*)a.Q + (int64_t)b * a.Lq * a.ldq + h * D;
  const T* Kb = (const T*)a.K + (int64_t)b * a.Lk * a.ldk + h * D;
  const T* Vb = (const T*)a.V + (int64_t)b * a.Lk * a.ldv + h * D;
  const T* dOb = (const T*)a.dO + (int64_t)b * a.Lq * a.lddo + h * D;
  RowFrag<T, D> kf, vf;
  kf.load(Kb + (int64_t)key * a.ldk, kv, g);
  vf.load(Vb + (int64_t)key * a.ldv, kv, g);
  const bool kmasked = kv && a.key_mask != nullptr && a.key_mask[(int64_t)b * a.Lk + key] == 0.f;
  const float kadd = kv ? (kmasked ? a.mask_neg : 0.f) : -INFINITY;
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const int64_t stat0 = ((int64_t)b * a.nh + h) * a.Lq;
  const uint64_t half = (uint64_t)(Lkp >> 1);
  const uint64_t e2lane = ((uint64_t)stat0 + (uint64_t)(4 * g)) * half + (uint64_t)(key >> 1);
  const bool odd = (key & 1) != 0;
  // the K image for the dQ product: all (up to 256) keys, rows past the end zero
#pragma unroll
  for (int v = tid; v < 256 * 8; v += 1024) {
    const int row = v >> 3, cv = v & 7;
    u32x4 z = {0u, 0u, 0u, 0u};
    if (row < a.Lk) z = *(const u32x4*)(Kb + (int64_t)row * a.ldk + cv * 8);
    *(u32x4*)(sKt + Img<T, D>::tr_off(row, cv * 8)) = z;
  }

  f32x4 accK[D / 16], accV[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) accK[i] = accV[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // delta = rowsum(dO * O) and LSE of ALL queries once, in the prologue (four threads per query row; the loads sit beside the K / V
  // loads above): the chunk loop below then carries no global-memory result that arithmetic has to wait for
  for (int q0 = 0; q0 < a.Lq; q0 += 256) {
    const int qq = q0 + (tid >> 2), part = tid & 3;
    float dsum = 0.f;
    if (qq < a.Lq) {
      const T* dr = dOb + (int64_t)qq * a.lddo + part * (D / 4);
      const T* orow = Ob + (int64_t)qq * a.ldo + part * (D / 4);
      const bf16x8 x0 = *(const bf16x8*)dr, x1 = *(const bf16x8*)(dr + 8), y0 = *(const bf16x8*)orow, y1 = *(const bf16x8*)(orow + 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += (float)x0[e] * (float)y0[e] + (float)x1[e] * (float)y1[e];
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    if (part == 0 && qq < a.Lq) {
      sDel[qq] = dsum;
      if (a.delta) a.delta[stat0 + qq] = dsum;
    }
  }
  const int Lq64 = (a.Lq + 63) & ~63;
  for (int q = tid; q < Lq64; q += 1024) {
    sLse[q] = q < a.Lq ? a.LSE[stat0 + q] : INFINITY;         // +inf => p = 0 for padded query rows
    if (q >= a.Lq) sDel[q] = 0.f;
  }
  // staging of a 64-query chunk: threads [0, 512) carry one 16-byte piece of Q each, threads [512, 1024) one of dO; with stored keep
  // bits (forward's gstvd_attn_t.drop_bits) threads [0, 256) also carry one word of the chunk's 4 x 16 tiles
  u32x4 pv4 = {0u, 0u, 0u, 0u};
  unsigned long long pbits = ~0ull;
  const bool is_q = tid < 512;
  const int sidx = is_q ? tid : tid - 512, srow = sidx >> 3, scv = sidx & 7;
  const unsigned long long* kbits = (BITS && dk.on) ? (const unsigned long long*)a.drop_bits : nullptr;
  const int nqt16 = (a.Lq + 15) >> 4, nkt16 = (a.Lk + 15) >> 4;
  auto prefetch = [&](int c0) {
    pv4 = (u32x4){0u, 0u, 0u, 0u};
    if (c0 + srow < a.Lq) pv4 = is_q ? *(const u32x4*)(Qb + (int64_t)(c0 + srow) * a.ldq + scv * 8)
                                     : *(const u32x4*)(dOb + (int64_t)(c0 + srow) * a.lddo + scv * 8);
    if (BITS && tid < 256) {
      const int qt = (c0 >> 4) + (tid >> 6), kt = (tid >> 2) & 15;
      pbits = (kbits && qt < nqt16 && kt < nkt16) ? kbits[(((int64_t)(b * a.nh + h) * nqt16 + qt) * nkt16 + kt) * 4 + (tid & 3)] : ~0ull;
    }
  };
  prefetch(0);
  const int nkb = (a.Lk + 31) >> 5;                           // 32-key blocks of the dQ contraction
  const int dq_i = wave & 3, dq_t = wave >> 2;                // this wave's tile of dQ^T: d block, query block of the chunk
  for (int c0 = 0; c0 < a.Lq; c0 += 64) {
    __syncthreads();                                          // the previous chunk's images (Q, dO, dS) are no longer read
    if (is_q) {
      *(u32x4*)(sQr + Img<T, D>::row_off(srow, scv)) = pv4;
      *(u32x4*)(sQt + Img<T, D>::tr_off(srow, scv * 8)) = pv4;
    } else {
      *(u32x4*)(sOr + Img<T, D>::row_off(srow, scv)) = pv4;
      *(u32x4*)(sOt + Img<T, D>::tr_off(srow, scv * 8)) = pv4;
    }
    if (BITS && tid < 256) sBits[tid] = pbits;
    __syncthreads();
    if (c0 + 64 < a.Lq) prefetch(c0 + 64);
    const uint64_t e2chunk = e2lane + (uint64_t)c0 * half;
#pragma unroll 1
    for (int pr = 0; pr < 4 / TP; ++pr) {
      f32x4 s[TP], dp[TP];
      float pd[TP][4], ds[TP][4];
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        s[tt] = first_product<T, D>(sQr, 16 * TP * pr + tt * 16, kf, lane);       // [q = 4g + r][key = li]
        dp[tt] = first_product<T, D>(sOr, 16 * TP * pr + tt * 16, vf, lane);
      }
#pragma unroll
      for (int tt = 0; tt < TP; ++tt) {
        const int q0 = 16 * TP * pr + tt * 16;
        const f32x4 lse4 = *(const f32x4*)(sLse + c0 + q0 + 4 * g);
        const f32x4 del4 = *(const f32x4*)(sDel + c0 + q0 + 4 * g);
        float f[4] = {1.f, 1.f, 1.f, 1.f};
        if (BITS) {       // forward's keep bits: word (key & 3) of tile (query tile, this wave's key tile), bits 16 (li >> 2) + 4g + r
          const unsigned long long w = sBits[(((q0 >> 4) * 16) + wave) * 4 + (li & 3)];
          const unsigned m4 = (unsigned)(w >> (((li >> 2) << 4) + 4 * g)) & 0xfu;
#pragma unroll
          for (int r = 0; r < 4; ++r) f[r] = ((m4 >> r) & 1u) ? dk.scale : 0.f;
        } else if (!BITS && dk.on) {      // the two keys of a draw's pair sit in neighbouring lanes: each lane draws for two rows, one quad permute swaps
          const int r0 = odd ? 2 : 0;
          const uint32_t mine0 = draw_pair<E32>(dk, e2chunk + (uint64_t)(q0 + r0) * half);
          const uint32_t mine1 = draw_pair<E32>(dk, e2chunk + (uint64_t)(q0 + r0 + 1) * half);
          const uint32_t other0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine0, 0xB1, 0xf, 0xf, true);
          const uint32_t other1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine1, 0xB1, 0xf, 0xf, true);
          const uint32_t d0 = odd ? other0 : mine0, d1 = odd ? other1 : mine1, d2 = odd ? mine0 : other0, d3 = odd ? mine1 : other1;
          const uint32_t sh = odd ? 16u : 0u;
          f[0] = ((d0 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
          f[1] = ((d1 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
          f[2] = ((d2 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
          f[3] = ((d3 >> sh) & 0xffffu) >= dk.thr ? dk.scale : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[tt][r] * a.scale + kadd - lse4[r]);
          pd[tt][r] = p * f[r];
          ds[tt][r] = p * (dp[tt][r] * f[r] - del4[r]) * a.scale;
        }
        // dS^T for the dQ product: row = this lane's key, four consecutive queries (zero for keys / queries past the end: p = 0)
        *(s16x4*)(sDs + Img<T, D>::tr_off(key, q0 + 4 * g)) = pack_bf16x4(ds[tt][0], ds[tt][1], ds[tt][2], ds[tt][3]);
      }
      second_product_pair<T, D>(accV, sOt, 16 * TP * pr, pd[0], pd[1], lane);
      second_product_pair<T, D>(accK, sQt, 16 * TP * pr, ds[0], ds[1], lane);
    }
    __syncthreads();                                          // the chunk's dS is complete
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const int rr = 4 * g + (li >> 2), cc = 4 * (lane & 3);
      // two 32-key blocks at a time: their eight transposed reads are in flight together, then two MFMAs (rows past the last
      // key are zero in both images, so a partial group needs no test)
      for (int kb0 = 0; kb0 < nkb; kb0 += 2) {
        s16x4 alo[2], ahi[2], blo[2], bhi[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kb = kb0 + u;
          alo[u] = lds_tr16(sKt + Img<T, D>::tr_off(32 * kb + rr, dq_i * 16 + cc));
          ahi[u] = lds_tr16(sKt + Img<T, D>::tr_off(32 * kb + 16 + rr, dq_i * 16 + cc));
          blo[u] = lds_tr16(sDs + Img<T, D>::tr_off(32 * kb + rr, dq_t * 16 + cc));
          bhi[u] = lds_tr16(sDs + Img<T, D>::tr_off(32 * kb + 16 + rr, dq_t * 16 + cc));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const s16x8 av = {alo[u][0], alo[u][1], alo[u][2], alo[u][3], ahi[u][0], ahi[u][1], ahi[u][2], ahi[u][3]};
          const s16x8 bv = {blo[u][0], blo[u][1], blo[u][2], blo[u][3], bhi[u][0], bhi[u][1], bhi[u][2], bhi[u][3]};
          acc = mfma_bf16_k32(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc);
        }
      }
      const int q = c0 + dq_t * 16 + li;                      // acc: rows d = 16 dq_i + 4g + r, column = query li
      if (q < a.Lq) st4((T*)a.dQ + ((int64_t)b * a.Lq + q) * a.lddq + h * D + dq_i * 16 + 4 * g, acc);
    }
  }
  if (kv) {
    T* dKp = (T*)a.dK + ((int64_t)b * a.Lk + key) * a.lddk + h * D;
    T* dVp = (T*)a.dV + ((int64_t)b * a.Lk + key) * a.lddv + h * D;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
      st4(dKp + i * 16 + 4 * g, accK[i]);
      st4(dVp + i * 16 + 4 * g, accV[i]);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------
template <typename K> static int attn_lds_attr(K kernel, int bytes) {
  if (bytes <= 48 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? 0 : (int)e;
}

static int attn_check(const gstvd_attn_t* a, bool bwd) {
  if (!a || !a->Q || !a->K || !a->V || !a->O) return GSTVD_E_NULL;
  if (a->dtype != GSTVD_F32 && a->dtype != GSTVD_BF16) return GSTVD_E_DTYPE;
  if (a->d != 32 && a->d != 64 && a->d != 128) return GSTVD_E_UNSUPPORTED;
  if (a->B <= 0 || a->nh <= 0 || a->Lq <= 0 || a->Lk <= 0) return GSTVD_E_SHAPE;
  const int ve = a->dtype == GSTVD_BF16 ? 8 : 4;
  if ((a->ldq % ve) || (a->ldk % ve) || (a->ldv % ve) || (a->ldo % 4)) return GSTVD_E_ALIGN;
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) & 15) return GSTVD_E_ALIGN;
  if (bwd) {
    if (a->kv_group > 1 || a->q_bstride > 0 || a->kv_bstride > 0) return GSTVD_E_UNSUPPORTED;
    if (!a->dO || !a->dQ || !a->dK || !a->dV || !a->LSE || !a->delta) return GSTVD_E_NULL;
    if ((a->lddo % ve) || (a->lddq % 4) || (a->lddk % 4) || (a->lddv % 4)) return GSTVD_E_ALIGN;
  }
  return 0;
}

template <typename T, int D> static int attn_fwd_launch(const gstvd_attn_t& a, hipStream_t s) {
  if (a.Lq == 1 && !a.causal && !(a.dropout_p > 0.f && a.rng) && a.Lk <= 8192) {      // the decode step's shape
    const int ldsd = (D + a.Lk) * 4;
    hipLaunchKernelGGL((attn_decode_kernel<T, D>), dim3((unsigned)a.nh, (unsigned)a.B), dim3(DEC_NT), ldsd, s, a);
    GSTVD_LAUNCH_CHECK();
    return 0;
  }
  constexpr int lds = 2 * Img<T, D>::BYTES + 64 * 4;
  static int rc = attn_lds_attr(attn_fwd_kernel<T, D>, lds);
  if (rc) return rc;
  dim3 grid((unsigned)((a.Lq + 63) / 64), (unsigned)a.nh, (unsigned)a.B);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, dim3(256), lds, s, a);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
// One launch for the whole backward: per (row, head) nkb blocks own 64 keys each (dK, dV: they walk the queries in nqb chunks) and
// nqb blocks own 64 queries each (dQ: they walk the keys in nkb chunks).  A 1-D grid, ALL blocks of the longer-running class first
// (round 5): with the classes interleaved per (row, head) -- blockIdx.x over both -- the grid of the few-query / few-key shapes is
// 1.25-1.5 rounds of the chip's resident workgroups, so the long blocks of the last (row, head)s started when the first short ones
// retired and the launch took the SUM of the two classes' times (decoder cross-attention 25 x 293: 23.8 us for 14.1 us of dK/dV
// blocks alone and 12.5 us of dQ blocks alone; profiles/r05_attn_block_order.txt).  Long ones first, the short ones fill in behind.
template <typename T, int D>
__global__ __launch_bounds__(256, (D <= 64 ? 3 : 2)) void attn_bwd_kernel(gstvd_attn_t a, int nkb, int nqb, int dq_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nbh = a.nh * a.B;
  int id = blockIdx.x;
  const int nfirst = (dq_first ? nqb : nkb) * nbh;
  const bool second = id >= nfirst;
  if (second) id -= nfirst;
  const bool is_dq = (dq_first != 0) != second;
  const int per = is_dq ? nqb : nkb;
  const int bx = id % per, bh = id / per, h = bh % a.nh, b = bh / a.nh;
  if constexpr (sizeof(T) == 2) {
    if (attn_small_index_space(a)) {
      if (!is_dq) attn_bwd_dkv_body<T, D, true>(a, bx, h, b, smem);
      else attn_bwd_dq_body<T, D, true>(a, bx, h, b, smem);
      return;
    }
  }
  if (!is_dq) attn_bwd_dkv_body<T, D, false>(a, bx, h, b, smem);
  else attn_bwd_dq_body<T, D, false>(a, bx, h, b, smem);
}

static bool attn_small_index_space_host(const gstvd_attn_t& a) {
  return (uint64_t)a.B * (uint64_t)a.nh * (uint64_t)a.Lq * (uint64_t)((a.Lk + 3) & ~3) < (1ull << 33);
}

template <typename T, int D> static int attn_bwd_launch(const gstvd_attn_t& a, hipStream_t s) {
  constexpr bool BF = sizeof(T) == 2;
  if constexpr (BF && D == 64) {
    // one pass per (row, head) when a head's keys fit its 16 waves and there are enough queries to be worth a whole CU
    // (GSTVD_ATTN_ONEPASS=0: the two-part kernel everywhere, for A/B runs)
    static const int onepass = [] { const char* e = getenv("GSTVD_ATTN_ONEPASS"); return e ? atoi(e) : 1; }();
    if (onepass && !a.causal && a.Lk > 64 && a.Lk <= 256 && a.Lq >= 64 && a.Lq <= ONEPASS_MAX_LQ) {
      constexpr int lds1p = 12 * Img<T, D>::BYTES + 2 * ONEPASS_MAX_LQ * 4 + 256 * 8;
      static int rc1 = attn_lds_attr(attn_bwd_onepass_kernel<true, false>, lds1p) | attn_lds_attr(attn_bwd_onepass_kernel<false, false>, lds1p) |
                       attn_lds_attr(attn_bwd_onepass_kernel<true, true>, lds1p);
      if (rc1) return rc1;
      dim3 grid((unsigned)a.nh, (unsigned)a.B);
      const bool bits = a.drop_bits != nullptr && a.dropout_p > 0.f && a.rng != nullptr;     // forward left the keep bits of its draws
      if (bits) hipLaunchKernelGGL((attn_bwd_onepass_kernel<true, true>), grid, dim3(1024), lds1p, s, a);
      else if (attn_small_index_space_host(a)) hipLaunchKernelGGL((attn_bwd_onepass_kernel<true, false>), grid, dim3(1024), lds1p, s, a);
      else hipLaunchKernelGGL((attn_bwd_onepass_kernel<false, false>), grid, dim3(1024), lds1p, s, a);
      GSTVD_LAUNCH_CHECK();
      return 0;
    }
  }
  constexpr int lds1 = (BF ? 3 : 2) * Img<T, D>::BYTES + 64 * 4;
  constexpr int lds2 = (BF ? 4 : 2) * Img<T, D>::BYTES + 128 * 4;
  constexpr int lds = lds1 > lds2 ? lds1 : lds2;
  static int rc = attn_lds_attr(attn_bwd_kernel<T, D>, lds);
  if (rc) return rc;
  const int nkb = (a.Lk + 63) / 64, nqb = (a.Lq + 63) / 64;
  // a dQ block walks nkb key chunks, a dK/dV block nqb query chunks (with two second products per tile: the longer one at a tie);
  // the longer-running class goes first (round 5, profiles/r05_attn_block_order.txt; the forced orders of that A/B are gone)
  const int dq_first = nkb > nqb ? 1 : 0;
  dim3 grid((unsigned)((nkb + nqb) * a.nh * a.B));
  hipLaunchKernelGGL((attn_bwd_kernel<T, D>), grid, dim3(256), lds, s, a, nkb, nqb, dq_first);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

template <typename T> static int attn_fwd_d(const gstvd_attn_t& a, hipStream_t s) {
  if (a.d == 32) return attn_fwd_launch<T, 32>(a, s);
  if (a.d == 64) return attn_fwd_launch<T, 64>(a, s);
  return attn_fwd_launch<T, 128>(a, s);
}
template <typename T> static int attn_bwd_d(const gstvd_attn_t& a, hipStream_t s) {
  if (a.d == 32) return attn_bwd_launch<T, 32>(a, s);
  if (a.d == 64) return attn_bwd_launch<T, 64>(a, s);
  return attn_bwd_launch<T, 128>(a, s);
}

extern "C" int gstvd_attn_fwd(const gstvd_attn_t* a, gstvd_stream_t stream) {
  int rc = attn_check(a, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == GSTVD_BF16 ? attn_fwd_d<bf16>(*a, s) : attn_fwd_d<float>(*a, s);
}
extern "C" int gstvd_attn_bwd(const gstvd_attn_t* a, gstvd_stream_t stream) {
  int rc = attn_check(a, true);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == GSTVD_BF16 ? attn_bwd_d<bf16>(*a, s) : attn_bwd_d<float>(*a, s);
}
