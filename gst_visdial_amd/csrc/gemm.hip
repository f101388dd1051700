// MFMA GEMM for gfx950: C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)), every nn.Linear fwd/dgrad/wgrad
// on the gst-visdial enc_dec_a path (see include/gstvd_hip.h).
//
//  * bf16 inputs  -> v_mfma_f32_16x16x32_bf16, fp32 accumulate   (throughput mode)
//  * fp32 inputs  -> v_mfma_f32_16x16x4_f32, exact fp32 fma chain (parity mode, 1e-4 logits gate)
//  * operands are staged global -> registers -> LDS in 16-byte vectors, double buffered, one barrier
//    per K tile; the next tile's global loads are in flight while the current tile is multiplied;
//  * an operand may be row-major ([x][k], k contiguous) or k-major ([k][x]); the LDS image keeps the
//    memory layout (straight, coalesced copy) and the MFMA fragment is read either with ds_read_b128
//    (row-major, XOR-swizzled 16-byte slots) or with ds_read_b64_tr_b16 (k-major, XOR-swizzled 32-byte
//    blocks) -- no transposing pass anywhere;
//  * the MFMA is issued "swapped" (n on the accumulator rows) so each lane owns 4 consecutive n of one
//    output row and the epilogue stores 8/16 bytes per lane.
#include "gemm_common.h"

template <typename T, typename OT, int BM, int BN, int WM, int WN, bool AKM, bool BKM>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmP p) {
  constexpr int VE = TileCfg<T>::VE, BK = TileCfg<T>::BK, NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int A_BYTES = image_bytes<T, BM, AKM>(), B_BYTES = image_bytes<T, BN, BKM>();
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NVA = BM * BK / VE / NT, NVB = BN * BK / VE / NT;
  static_assert(NVA >= 1 && NVB >= 1, "tile too small for the thread count");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t z = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const char* gA = p.A + z * p.sA * (int64_t)sizeof(T);
  const char* gB = p.B + z * p.sB * (int64_t)sizeof(T);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[NVA], rb[NVB];
  const int64_t nkt = (p.K + BK - 1) / BK;
  load_tile<T, BM, AKM, NT, NVA>(ra, gA, p.lda, m0, p.M, 0, p.K, tid);
  load_tile<T, BN, BKM, NT, NVB>(rb, gB, p.ldb, n0, p.N, 0, p.K, tid);
  store_tile<T, BM, AKM, NT, NVA>(ra, smem, tid);
  store_tile<T, BN, BKM, NT, NVB>(rb, smem + A_BYTES, tid);
  __syncthreads();

  for (int64_t t = 0; t < nkt; ++t) {
    const char* cA = smem + (t & 1) * STAGE;
    const char* cB = cA + A_BYTES;
    const bool more = (t + 1 < nkt);
    if (more) {
      load_tile<T, BM, AKM, NT, NVA>(ra, gA, p.lda, m0, p.M, (t + 1) * BK, p.K, tid);
      load_tile<T, BN, BKM, NT, NVB>(rb, gB, p.ldb, n0, p.N, (t + 1) * BK, p.K, tid);
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        bf16x8 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = frag_bf16<BM, AKM>(cA, wm * WTM + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = frag_bf16<BN, BKM>(cB, wn * WTN + j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa[i], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = frag_f32<BM, AKM>(cA, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = frag_f32<BN, BKM>(cB, wn * WTN + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_f32_k4(fb[j], fa[i], acc[i][j]);
      }
    }
    if (more) {
      char* nA = smem + ((t + 1) & 1) * STAGE;
      store_tile<T, BM, AKM, NT, NVA>(ra, nA, tid);
      store_tile<T, BN, BKM, NT, NVB>(rb, nA + A_BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] ---------------------
  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<T, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li, n0 + wn * WTN + j * 16 + 4 * g);
}

template <typename T, typename OT, int BM, int BN, bool AKM, bool BKM>
static int launch_cfg(const GemmP& p, int64_t batch, hipStream_t s) {
  constexpr int WM = 2, WN = 2;
  constexpr int lds = 2 * (image_bytes<T, BM, AKM>() + image_bytes<T, BN, BKM>());
  auto k = gemm_kernel<T, OT, BM, BN, WM, WN, AKM, BKM>;
  static int attr_rc = ensure_lds(k, lds);
  if (attr_rc) return attr_rc;
  dim3 grid((unsigned)((p.N + BN - 1) / BN), (unsigned)((p.M + BM - 1) / BM), (unsigned)batch);
  GSTVD_LAUNCH(k, grid, dim3(WM * WN * 64), lds, s, p);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

template <typename T, typename OT, bool AKM, bool BKM>
static int launch_layout(const GemmP& p, int64_t batch, hipStream_t s) {
  // big tile when it still fills the chip, else the small one (M=592/400 streams, tiny test shapes)
  int64_t big = ((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  if (p.M >= 256 && p.N >= 128 && big >= 96) return launch_cfg<T, OT, 128, 128, AKM, BKM>(p, batch, s);
  return launch_cfg<T, OT, 64, 64, AKM, BKM>(p, batch, s);
}

template <typename T, typename OT>
static int launch_dtype(const GemmP& p, int64_t batch, int akm, int bkm, hipStream_t s) {
  if (!akm && !bkm) return launch_layout<T, OT, false, false>(p, batch, s);
  if (!akm && bkm) return launch_layout<T, OT, false, true>(p, batch, s);
  if (akm && bkm) return launch_layout<T, OT, true, true>(p, batch, s);
  return launch_layout<T, OT, true, false>(p, batch, s);
}

thread_local const void** gstvd_plan_capture = nullptr;

static int gemm_params(const gstvd_gemm_t* g, GemmP& p) {
  if (!g || !g->A || !g->B || !g->C) return GSTVD_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0 || g->batch <= 0) return GSTVD_E_SHAPE;
  const int ve = g->dtype_in == GSTVD_BF16 ? 8 : 4;
  if (g->N % 4) return GSTVD_E_SHAPE;
  if (!g->a_kmajor && (g->K % ve)) return GSTVD_E_SHAPE;
  if (!g->b_kmajor && (g->K % ve)) return GSTVD_E_SHAPE;
  if (g->a_kmajor && (g->M % ve)) return GSTVD_E_SHAPE;
  if (g->b_kmajor && (g->N % ve)) return GSTVD_E_SHAPE;
  if ((g->lda % ve) || (g->ldb % ve) || (g->ldc % 4)) return GSTVD_E_ALIGN;
  if (((uintptr_t)g->A | (uintptr_t)g->B | (uintptr_t)g->C) & 15) return GSTVD_E_ALIGN;
  if ((g->epilogue & GSTVD_EPI_BIAS) && !g->bias) return GSTVD_E_NULL;
  if ((g->epilogue & GSTVD_EPI_ADD) && !g->addend) return GSTVD_E_NULL;
  if ((g->epilogue & (GSTVD_EPI_GELU | GSTVD_EPI_DGELU)) && !g->aux) return GSTVD_E_NULL;
  p.A = (const char*)g->A; p.B = (const char*)g->B; p.C = (char*)g->C;
  p.bias = g->bias; p.addend = (const char*)g->addend; p.aux = (char*)g->aux;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc; p.ldadd = g->ldadd; p.ldaux = g->ldaux;
  p.sA = g->sA; p.sB = g->sB; p.sC = g->sC; p.sAdd = g->sAdd; p.sAux = g->sAux;
  p.epi = g->epilogue; p.alpha = g->alpha; p.p = g->dropout_p; p.site = g->site; p.rng = g->rng;
  return 0;
}

// Decode step: C = epi(LayerNorm(A; gamma, beta, eps) . B^T) for M <= 16 rows, K <= 1024, bf16 operands (gemv.hip).
extern "C" int gstvd_gemv_ln(const gstvd_gemm_t* g, const float* gamma, const float* beta, float eps, void* y_out, int64_t ldy,
                             gstvd_stream_t stream) {
  GemmP p;
  const int rc = gemm_params(g, p);
  if (rc) return rc;
  if (g->dtype_in != GSTVD_BF16) return GSTVD_E_UNSUPPORTED;
  if (g->dtype_out != GSTVD_BF16 && g->dtype_out != GSTVD_F32) return GSTVD_E_DTYPE;
  return gemv16_ln_dispatch(p, g->batch, g->a_kmajor, g->b_kmajor, g->dtype_out == GSTVD_F32, gamma, beta, eps, y_out, ldy,
                            (hipStream_t)stream);
}

// Skinny, deep problems (few output tiles, long K: the decoder's 400-row GEMMs, the LM-head input gradient): `splits`
// workgroups share one 64x64 output tile; see dma_tile in gemm_dma.hip.  `ws` is caller-owned scratch of at least
// gstvd_gemm_splitk_ws_bytes(M, N, splits) bytes, zero-filled once and never shared by launches that can overlap.
extern "C" int gstvd_gemm_splitk(const gstvd_gemm_t* g, int32_t splits, void* ws, int64_t ws_bytes, gstvd_stream_t stream) {
  GemmP p;
  const int rc = gemm_params(g, p);
  if (rc) return rc;
  if (splits < 2 || splits > 16 || !ws) return GSTVD_E_SHAPE;
  if (g->dtype_in != GSTVD_BF16 || g->batch != 1) return GSTVD_E_UNSUPPORTED;
  if (g->dtype_out != GSTVD_BF16 && g->dtype_out != GSTVD_F32) return GSTVD_E_DTYPE;
  return gemm_dma_splitk_dispatch(p, g->a_kmajor, g->b_kmajor, g->dtype_out == GSTVD_F32, splits, ws, ws_bytes, (hipStream_t)stream);
}

// Which kernel would gstvd_gemm (splits <= 1) / gstvd_gemm_splitk (splits >= 2) launch for this problem?  Writes the device
// function's (mangled) symbol name into buf.  Nothing is launched, no memory is touched.
extern "C" int gstvd_gemm_kernel_name(const gstvd_gemm_t* g, int32_t splits, char* buf, int32_t buf_len) {
  if (!g || !buf || buf_len <= 1) return GSTVD_E_NULL;
  const void* fn = nullptr;
  gstvd_plan_capture = &fn;
  static char dummy_ws[16];
  const int rc = splits >= 2 ? gstvd_gemm_splitk(g, splits, dummy_ws, (int64_t)1 << 40, nullptr) : gstvd_gemm(g, nullptr);
  gstvd_plan_capture = nullptr;
  if (rc) return rc;
  if (!fn) return GSTVD_E_UNSUPPORTED;
  const char* name = hipKernelNameRefByPtr(fn, nullptr);
  if (!name) return GSTVD_E_UNSUPPORTED;
  int i = 0;
  for (; name[i] && i < buf_len - 1; ++i) buf[i] = name[i];
  buf[i] = 0;
  return 0;
}

extern "C" int gstvd_gemm(const gstvd_gemm_t* g, gstvd_stream_t stream) {
  GemmP p;
  const int prc = gemm_params(g, p);
  if (prc) return prc;
  hipStream_t s = (hipStream_t)stream;
  if (g->dtype_in == GSTVD_BF16 && (g->dtype_out == GSTVD_BF16 || g->dtype_out == GSTVD_F32)) {
    int rc = gemv16_dispatch(p, g->batch, g->a_kmajor, g->b_kmajor, g->dtype_out == GSTVD_F32, s);     // decode step: M <= 16
    if (rc != GSTVD_E_UNSUPPORTED) return rc;
    rc = gemm_dma256_dispatch(p, g->batch, g->a_kmajor, g->b_kmajor, g->dtype_out == GSTVD_F32, s);
    if (rc != GSTVD_E_UNSUPPORTED) return rc;
    rc = gemm_dma_dispatch(p, g->batch, g->a_kmajor, g->b_kmajor, g->dtype_out == GSTVD_F32, s);
    if (rc != GSTVD_E_UNSUPPORTED) return rc;
  }
  if (g->dtype_in == GSTVD_BF16 && g->dtype_out == GSTVD_BF16) return launch_dtype<bf16, bf16>(p, g->batch, g->a_kmajor, g->b_kmajor, s);
  if (g->dtype_in == GSTVD_BF16 && g->dtype_out == GSTVD_F32) return launch_dtype<bf16, float>(p, g->batch, g->a_kmajor, g->b_kmajor, s);
  if (g->dtype_in == GSTVD_F32 && g->dtype_out == GSTVD_F32) return launch_dtype<float, float>(p, g->batch, g->a_kmajor, g->b_kmajor, s);
  return GSTVD_E_DTYPE;
}
