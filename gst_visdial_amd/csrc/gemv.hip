// Skinny GEMM for the KV-cached decode step (models/visual_dialog_model.py:74-120 feeds ONE new token per row through the
// decoder stack): C[M <= 16, N] = epi(A[M, K] . B[N, K]^T), bf16 in, fp32 accumulate.  Every nn.Linear of a decode step has
// this shape; the tiled kernels put 12-48 workgroups on it and spend 8-10 us per launch waiting for one CU to stream its
// weight slab through an LDS ring.  Here one workgroup owns 16 output columns (one 16x16 MFMA tile), its waves split K, every
// lane issues ALL of its 16-byte operand loads up front (global -> registers, no LDS staging: a weight element is used
// once), MFMA 16x16x32 accumulates, and the waves' partial tiles are added through LDS.  One memory round trip per launch.
#include "gemm_common.h"

template <int NW, bool F32OUT>
__global__ __launch_bounds__(NW * 64) void gemv16_kernel(GemmP p) {
  __shared__ f32x4 red[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;
  const bool av = li < p.M, bv = n0 + li < p.N;
  const bf16* arow = A + (int64_t)li * p.lda + 8 * g;
  const bf16* brow = B + (n0 + li) * p.ldb + 8 * g;
  // K steps of 32, dealt round-robin to the waves; K % 8 == 0 is guaranteed by gemm_params (16-byte units never straddle K)
  const int nk = (int)((p.K + 31) / 32);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const bf16x8 zero = __builtin_bit_cast(bf16x8, (s16x8){0, 0, 0, 0, 0, 0, 0, 0});
  constexpr int UNR = 8;
  for (int s0 = wave; s0 < nk; s0 += NW * UNR) {
    bf16x8 fa[UNR], fb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {                      // all loads of this batch in flight before the first MFMA
      const int s = s0 + u * NW;
      const int64_t k = (int64_t)s * 32 + 8 * g;
      const bool kin = s < nk && k < p.K;
      fa[u] = (kin && av) ? *(const bf16x8*)(arow + (int64_t)s * 32) : zero;
      fb[u] = (kin && bv) ? *(const bf16x8*)(brow + (int64_t)s * 32) : zero;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc = mfma_bf16_k32(fb[u], fa[u], acc);       // D[n = 4g + r][m = li] ... see below
  }
  // mfma(fb, fa): rows of the result index B's rows (n), columns index A's rows (m): lane holds C[m = li][n = n0 + 4g + r]
  red[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < NW; ++w) acc += red[w][lane];
  const int64_t m = li, n = n0 + 4 * g;
  if (m >= p.M || n >= p.N) return;
  f32x4 v = acc * p.alpha;
  if (p.epi & GSTVD_EPI_BIAS) v += *(const f32x4*)(p.bias + n);
  if (p.epi & GSTVD_EPI_ADD)        // the addend has the OUTPUT's type, as in the tiled kernels' epilogue (gemm_epilogue_tile)
    v += F32OUT ? ld4((const float*)p.addend + m * p.ldadd + n) : ld4((const bf16*)p.addend + m * p.ldadd + n);
  if (p.epi & GSTVD_EPI_GELU) {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float g_, d_; gelu_both<true>(v[e], g_, d_); v[e] = g_; d[e] = d_; }
    if (p.aux) st4((bf16*)p.aux + m * p.ldaux + n, d);
  }
  if (F32OUT) st4((float*)p.C + m * p.ldc + n, v);
  else st4((bf16*)p.C + m * p.ldc + n, v);
}

// The same product with a LayerNorm folded into the A operand: C = epi(LN(A; gamma, beta, eps) . B^T), K <= 1024 (a whole row of
// A sits in the registers of the workgroup's four waves).  In the decode step every LayerNorm is followed by a Linear that
// reads its output; as a launch of its own the LayerNorm costs as much as that Linear (both are at the ~4-5 us floor of a
// dependent launch).  Every workgroup normalises the M <= 16 rows itself (24 KB of bf16 from L2), workgroup 0 also writes the
// normalised rows to y_out -- the residual input of the sub-layer's closing Linear two launches later.
struct LnIn { const float* gamma; const float* beta; float eps; bf16* y_out; int64_t ldy; };

template <bool F32OUT>
__global__ __launch_bounds__(256) void gemv16_ln_kernel(GemmP p, LnIn ln) {
  constexpr int NW = 4, UNR = 8;                              // K <= NW * UNR * 32 = 1024
  __shared__ f32x4 red[NW][64];
  __shared__ float rsum[NW][16], rsq[NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;
  const bool av = li < p.M, bv = n0 + li < p.N;
  const bf16* arow = A + (int64_t)li * p.lda + 8 * g;
  const bf16* brow = B + (n0 + li) * p.ldb + 8 * g;
  const int nk = (int)((p.K + 31) / 32);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const bf16x8 zero = __builtin_bit_cast(bf16x8, (s16x8){0, 0, 0, 0, 0, 0, 0, 0});
  bf16x8 fa[UNR], fb[UNR];
  f32x4 gm[UNR][2], bt[UNR][2];                              // gamma / beta of this lane's k positions: loaded with the operands,
  bool kin[UNR];                                             // one memory round trip for everything
  const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int s = wave + u * NW;
    const int64_t k = (int64_t)s * 32 + 8 * g;
    kin[u] = s < nk && k < p.K;
    fa[u] = (kin[u] && av) ? *(const bf16x8*)(arow + (int64_t)s * 32) : zero;
    fb[u] = (kin[u] && bv) ? *(const bf16x8*)(brow + (int64_t)s * 32) : zero;
    gm[u][0] = kin[u] ? *(const f32x4*)(ln.gamma + k) : zf;
    gm[u][1] = kin[u] ? *(const f32x4*)(ln.gamma + k + 4) : zf;
    bt[u][0] = kin[u] ? *(const f32x4*)(ln.beta + k) : zf;
    bt[u][1] = kin[u] ? *(const f32x4*)(ln.beta + k + 4) : zf;
  }
  // row statistics in ONE reduction round (sum and sum of squares together; fp32, var = E[x^2] - mean^2 clamped at 0: the
  // rows are residual-stream activations, |mean| is far below the spread, nothing cancels) -- two barriers instead of four
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = (float)fa[u][j]; s1 += x; s2 += x * x; }     // (elements outside K / M are zero)
  s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
  s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);                                   // over the four k-groups of the wave
  if (g == 0) { rsum[wave][li] = s1; rsq[wave][li] = s2; }
  __syncthreads();
  const float mean = (rsum[0][li] + rsum[1][li] + rsum[2][li] + rsum[3][li]) / (float)p.K;
  const float ex2 = (rsq[0][li] + rsq[1][li] + rsq[2][li] + rsq[3][li]) / (float)p.K;
  const float rstd = 1.0f / sqrtf(fmaxf(ex2 - mean * mean, 0.f) + ln.eps);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    if (kin[u]) {                                             // (uniform per 16-lane group; MFMA below is outside the branch)
      const int64_t k = (int64_t)(wave + u * NW) * 32 + 8 * g;
      const f32x4 g0 = gm[u][0], g1 = gm[u][1], b0 = bt[u][0], b1 = bt[u][1];
      bf16x8 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (bf16)(((float)fa[u][j] - mean) * rstd * g0[j] + b0[j]);
        y[j + 4] = (bf16)(((float)fa[u][j + 4] - mean) * rstd * g1[j] + b1[j]);
      }
      fa[u] = av ? y : zero;
      if (blockIdx.x == 0 && ln.y_out && av) *(bf16x8*)(ln.y_out + (int64_t)li * ln.ldy + k) = y;
    }
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc = mfma_bf16_k32(fb[u], fa[u], acc);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < NW; ++w) acc += red[w][lane];
  const int64_t m = li, n = n0 + 4 * g;
  if (m >= p.M || n >= p.N) return;
  f32x4 v = acc * p.alpha;
  if (p.epi & GSTVD_EPI_BIAS) v += *(const f32x4*)(p.bias + n);
  if (p.epi & GSTVD_EPI_ADD)        // the addend has the OUTPUT's type, as in the tiled kernels' epilogue (gemm_epilogue_tile)
    v += F32OUT ? ld4((const float*)p.addend + m * p.ldadd + n) : ld4((const bf16*)p.addend + m * p.ldadd + n);
  if (p.epi & GSTVD_EPI_GELU) {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float g_, d_; gelu_both<true>(v[e], g_, d_); v[e] = g_; d[e] = d_; }
    if (p.aux) st4((bf16*)p.aux + m * p.ldaux + n, d);
  }
  if (F32OUT) st4((float*)p.C + m * p.ldc + n, v);
  else st4((bf16*)p.C + m * p.ldc + n, v);
}

int gemv16_ln_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, const float* gamma, const float* beta, float eps,
                       void* y_out, int64_t ldy, hipStream_t s) {
  if (p.M > 16 || p.K > 1024 || batch != 1 || akm || bkm) return GSTVD_E_UNSUPPORTED;
  if (p.epi & (GSTVD_EPI_DGELU | GSTVD_EPI_DROPOUT)) return GSTVD_E_UNSUPPORTED;
  if (!gamma || !beta) return GSTVD_E_NULL;
  if (y_out && ((ldy % 8) || ((uintptr_t)y_out & 15))) return GSTVD_E_ALIGN;
  LnIn ln{gamma, beta, eps, (bf16*)y_out, ldy};
  const dim3 grid((unsigned)((p.N + 15) / 16));
  if (out_f32) GSTVD_LAUNCH((gemv16_ln_kernel<true>), grid, dim3(256), 0, s, p, ln);
  else GSTVD_LAUNCH((gemv16_ln_kernel<false>), grid, dim3(256), 0, s, p, ln);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

// returns GSTVD_E_UNSUPPORTED when the problem is not of this shape (the caller falls through to the tiled kernels)
int gemv16_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s) {
  if (p.M > 16 || batch != 1 || akm || bkm) return GSTVD_E_UNSUPPORTED;
  if (p.epi & (GSTVD_EPI_DGELU | GSTVD_EPI_DROPOUT)) return GSTVD_E_UNSUPPORTED;
  const dim3 grid((unsigned)((p.N + 15) / 16));
  if (p.K >= 2048) {
    if (out_f32) GSTVD_LAUNCH((gemv16_kernel<8, true>), grid, dim3(512), 0, s, p);
    else GSTVD_LAUNCH((gemv16_kernel<8, false>), grid, dim3(512), 0, s, p);
  } else {
    if (out_f32) GSTVD_LAUNCH((gemv16_kernel<4, true>), grid, dim3(256), 0, s, p);
    else GSTVD_LAUNCH((gemv16_kernel<4, false>), grid, dim3(256), 0, s, p);
  }
  GSTVD_LAUNCH_CHECK();
  return 0;
}
