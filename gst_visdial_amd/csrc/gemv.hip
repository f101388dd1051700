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
  if (p.epi & GSTVD_EPI_ADD) v += ld4((const bf16*)p.addend + m * p.ldadd + n);
  if (p.epi & GSTVD_EPI_GELU) {
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float g_, d_; gelu_both<true>(v[e], g_, d_); v[e] = g_; d[e] = d_; }
    if (p.aux) st4((bf16*)p.aux + m * p.ldaux + n, d);
  }
  if (F32OUT) st4((float*)p.C + m * p.ldc + n, v);
  else st4((bf16*)p.C + m * p.ldc + n, v);
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
