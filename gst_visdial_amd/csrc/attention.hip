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
//   * softmax statistics: online max/sum per query column in registers, 2 shuffles per 16-key tile;
//     forward saves only LSE; backward recomputes P = exp(s - LSE) and regenerates the dropout mask;
//   * fp32 parity mode runs the same skeleton on v_mfma_f32_16x16x4_f32.
#include "common.h"
#include <math.h>

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

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(gstvd_attn_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const uint64_t ebase = ((uint64_t)(b * a.nh + h) * a.Lq + (uint64_t)(qv ? q : 0)) * (uint64_t)Lkp;

  float m_run = -1e30f, l_part = 0.f;
  f32x4 accO[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) accO[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // chunk c+1's K / V / mask loads are issued right after chunk c has been parked in LDS and fly during its compute
  Stage64<T, D> pk, pv;
  float pm = -INFINITY;
  auto prefetch = [&](int c0) {
    pk.load(Kb, a.ldk, c0, a.Lk, tid);
    pv.load(Vb, a.ldv, c0, a.Lk, tid);
    if (tid < 64) {      // additive mask term of the key: 0 (valid), mask_neg (masked out), -inf (past the end)
      const int key = c0 + tid;
      pm = -INFINITY;
      if (key < a.Lk) pm = (a.key_mask == nullptr || a.key_mask[(int64_t)bk * a.Lk + key] != 0.f) ? 0.f : a.mask_neg;
    }
  };
  prefetch(0);
  for (int c0 = 0; c0 < a.Lk; c0 += 64) {
    __syncthreads();
    pk.store(sK, nullptr, tid);
    pv.store(Img<T, D>::BF ? nullptr : sV, Img<T, D>::BF ? sV : nullptr, tid);
    if (tid < 64) smask[tid] = pm;
    __syncthreads();
    if (c0 + 64 < a.Lk) prefetch(c0 + 64);
    const int ntile = (a.Lk - c0 + 15) / 16 < 4 ? (a.Lk - c0 + 15) / 16 : 4;
    for (int t = 0; t < ntile; ++t) {
      f32x4 s = first_product<T, D>(sK, t * 16, qf, lane);
      f32x4 madd = *(const f32x4*)(smask + t * 16 + 4 * g);
      if (a.causal) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + t * 16 + 4 * g + r > q && madd[r] == 0.f) madd[r] = a.mask_neg;     // causal x padding: the term is added once
      }
      float val[4], mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        val[r] = s[r] * a.scale + madd[r];
        mx = fmaxf(mx, val[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const f32x4 fac = drop_factor4(dk, ebase + (uint64_t)(c0 + t * 16 + 4 * g));
      float pd[4], ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(val[r] - m_new);
        ps += p;
        pd[r] = p * fac[r];
      }
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {       // the running max moved for some query of this wave
        const float alpha = __expf(m_run - m_new);
        l_part *= alpha;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) accO[i] *= alpha;
        m_run = m_new;
      }
      l_part += ps;
      second_product<T, D>(accO, sV, t * 16, pd, lane);
    }
  }
  float l_tot = l_part + __shfl_xor(l_part, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.f / l_tot;
  if (qv) {
    T* Op = (T*)a.O + ((int64_t)b * qbs + q) * a.ldo + h * D;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) st4(Op + i * 16 + 4 * g, accO[i] * inv);
    if (g == 0 && a.LSE) a.LSE[((int64_t)b * a.nh + h) * a.Lq + q] = m_run + __logf(l_tot);
  }
}

// =====================================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O)); same tiling as forward
// =====================================================================================================
template <typename T, int D>
DEVFN void attn_bwd_dq_body(const gstvd_attn_t& a, const int bx, char* smem) {
  constexpr bool BF = Img<T, D>::BF;
  char* sKr = smem;                                           // row image of K
  char* sKt = BF ? smem + Img<T, D>::BYTES : smem;            // transposed-read image of K
  char* sVr = smem + (BF ? 2 : 1) * Img<T, D>::BYTES;         // row image of V
  float* smask = (float*)(smem + (BF ? 3 : 2) * Img<T, D>::BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
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
  float delta = dof.dot(Ob + (int64_t)q * a.ldo, qv, g);
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const int64_t stat = ((int64_t)b * a.nh + h) * a.Lq + q;
  if (qv && g == 0) a.delta[stat] = delta;
  const float lse = qv ? a.LSE[stat] : 0.f;
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const uint64_t ebase = ((uint64_t)(b * a.nh + h) * a.Lq + (uint64_t)(qv ? q : 0)) * (uint64_t)Lkp;

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
    for (int t = 0; t < ntile; ++t) {
      const f32x4 s = first_product<T, D>(sKr, t * 16, qf, lane);
      const f32x4 dp = first_product<T, D>(sVr, t * 16, dof, lane);
      const f32x4 fac = drop_factor4(dk, ebase + (uint64_t)(c0 + t * 16 + 4 * g));
      f32x4 madd = *(const f32x4*)(smask + t * 16 + 4 * g);
      if (a.causal) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + t * 16 + 4 * g + r > q && madd[r] == 0.f) madd[r] = a.mask_neg;
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float val = s[r] * a.scale + madd[r];
        const float p = qv ? __expf(val - lse) : 0.f;
        ds[r] = p * (dp[r] * fac[r] - delta) * a.scale;
      }
      second_product<T, D>(acc, sKt, t * 16, ds, lane);
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
template <typename T, int D>
DEVFN void attn_bwd_dkv_body(const gstvd_attn_t& a, const int bx, char* smem) {
  constexpr bool BF = Img<T, D>::BF;
  constexpr int IB = Img<T, D>::BYTES;
  char* sQr = smem;
  char* sQt = BF ? smem + IB : smem;
  char* sOr = smem + (BF ? 2 : 1) * IB;                       // dO row image
  char* sOt = BF ? smem + 3 * IB : sOr;                       // dO transposed-read image
  float* sLse = (float*)(smem + (BF ? 4 : 2) * IB);
  float* sDel = sLse + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
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
  const DropKey dk = make_drop(a.dropout_p, a.site, a.rng);
  const int Lkp = round4(a.Lk);
  const uint64_t erow = (uint64_t)(b * a.nh + h) * a.Lq;
  const int64_t stat0 = ((int64_t)b * a.nh + h) * a.Lq;

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
    for (int t = 0; t < ntile; ++t) {
      const f32x4 s = first_product<T, D>(sQr, t * 16, kf, lane);     // [q = 4g+r][key = li]
      const f32x4 dp = first_product<T, D>(sOr, t * 16, vf, lane);
      float pd[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = t * 16 + 4 * g + r, qq = c0 + ql;
        const bool masked = kmasked || (a.causal && key > qq);
        const float val = kv ? s[r] * a.scale + (masked ? a.mask_neg : 0.f) : -INFINITY;
        const float p = __expf(val - sLse[ql]);
        const float f = drop_factor(dk, (erow + (uint64_t)(qq < a.Lq ? qq : 0)) * (uint64_t)Lkp + (uint64_t)(kv ? key : 0));
        pd[r] = p * f;
        ds[r] = p * (dp[r] * f - sDel[ql]) * a.scale;
      }
      second_product<T, D>(accV, sOt, t * 16, pd, lane);
      second_product<T, D>(accK, sQt, t * 16, ds, lane);
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
  constexpr int lds = 2 * Img<T, D>::BYTES + 64 * 4;
  static int rc = attn_lds_attr(attn_fwd_kernel<T, D>, lds);
  if (rc) return rc;
  dim3 grid((unsigned)((a.Lq + 63) / 64), (unsigned)a.nh, (unsigned)a.B);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, dim3(256), lds, s, a);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
// One launch for the whole backward: blocks [0, nkb) own 64 keys each (dK, dV), blocks [nkb, nkb + nqb) own 64 queries each (dQ).
// The dK/dV blocks come first: they are the longer ones (two second products per tile).
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(gstvd_attn_t a, int nkb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bx = blockIdx.x;
  if (bx < nkb) attn_bwd_dkv_body<T, D>(a, bx, smem);
  else attn_bwd_dq_body<T, D>(a, bx - nkb, smem);
}

template <typename T, int D> static int attn_bwd_launch(const gstvd_attn_t& a, hipStream_t s) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int lds1 = (BF ? 3 : 2) * Img<T, D>::BYTES + 64 * 4;
  constexpr int lds2 = (BF ? 4 : 2) * Img<T, D>::BYTES + 128 * 4;
  constexpr int lds = lds1 > lds2 ? lds1 : lds2;
  static int rc = attn_lds_attr(attn_bwd_kernel<T, D>, lds);
  if (rc) return rc;
  const int nkb = (a.Lk + 63) / 64, nqb = (a.Lq + 63) / 64;
  dim3 grid((unsigned)(nkb + nqb), (unsigned)a.nh, (unsigned)a.B);
  hipLaunchKernelGGL((attn_bwd_kernel<T, D>), grid, dim3(256), lds, s, a, nkb);
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
