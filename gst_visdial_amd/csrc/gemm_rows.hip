// LayerNorm folded into the Linear next to it, for the latency-bound row counts of the decoder (M = rows x 25 <= 640):
//
//   forward   C = epi( LN(drop(x) + res) . B^T )          -- the LayerNorm that closes a sub-layer + the Linear that reads it
//                                                             (transformers-4.16.2 BertSelfOutput / BertOutput -> next dense)
//   backward  C = epi( dx . B ),  dx = drop'(LN'(dy))      -- the LayerNorm's backward + the input gradient of the Linear
//                                                             that PRODUCED the LayerNorm's input
//
// Why: at M = 400 every launch of the decoder chain costs 7-17 us whatever it computes (a dependent launch costs >= 4.5 us on
// this chip, profiles/r03_grid_barrier.txt), and a decoder layer is a chain of 11 forward + 11 backward launches.  A row
// block of the GEMM reads whole rows of its A operand anyway (K = H = 768), so the workgroup computes the LayerNorm (or its
// backward) of its 16 rows itself -- every column tile of a row block redoes it: ~50 KB of L2 reads per workgroup -- and keeps
// the normalised rows as the RESIDENT A image of the whole K range in LDS (12 K-tiles x 2 KB); only the B operand streams
// through an LDS-DMA ring.  One column tile per row block also writes what the rest of the step needs
// (y, mean / rstd; dres, dx, the [3][H] column partials for dgamma / dbeta / dbias), spread over different column tiles so
// that no workgroup does all of it.  3 of 11 launches per layer and direction disappear.
//
// Arithmetic is that of layernorm.hip (same order of operations per row: two-pass variance, wave_sum reductions, bf16
// rounding where ln_fwd / ln_bwd round), so y, mean, rstd, dx, dres are bit-identical to the unfused kernels; the column
// partials are summed over these 16-row blocks (two rows per wave) instead of 4 / 16-row blocks of one row per wave (fp32
// reassociation only).
#include "gemm_common.h"

static __device__ __attribute__((aligned(256))) char g_zero_page_rows[256];

struct RowPro {
  gstvd_ln_t f;
  const void* dy; int64_t lddy; void* dres; int64_t lddres; void* dx; int64_t lddx; float* partial;
};

// Geometry.  The prologue is VALU work (~80 instructions per 4 elements incl. the dropout hash) that EVERY column tile of a row
// block repeats, so its latency per workgroup decides: 64-row blocks measured 27 / 54 us per launch (forward / backward; 8 us /
// 14 us of it prologue arithmetic on one CU) against 17.7 / 21 us for the two separate kernels.  16 rows per workgroup (two per
// wave) make it ~2 us; 128-column tiles keep the B operand's L2 -> LDS traffic at 192 KB per workgroup and the grid at 150 / 450 /
// 600 workgroups for N = 768 / 2304 / 3072 (two fit a CU: 24 KB A image + 48 KB ring).
constexpr int ROWS_BM = 16, ROWS_BN = 128, ROWS_NT = 512, ROWS_NS = 3;
constexpr int ROWS_A_TILE = ROWS_BM * 128, ROWS_B_TILE = ROWS_BN * 128;
constexpr int ROWS_NPB = ROWS_B_TILE / (ROWS_NT * 16);                                          // LDS-DMA pieces per thread and stage
// A image + B ring = 72 KB (H <= 768, three 256-column vectors per row: NV = 3) or 80 KB (H <= 1024, NV = 4: the vision stream's
// width, round 5): two workgroups per CU (N = 3072: 600 workgroups = 1.2 rounds instead of 2.3).  The backward's 12 / 16 KB of
// column-partial scratch overlays the ring's last slot, which the prologue leaves unfilled.
constexpr int rows_kt_max(int NV) { return NV * 4; }
constexpr int rows_lds(int NV) { return rows_kt_max(NV) * ROWS_A_TILE + ROWS_NS * ROWS_B_TILE; }
static_assert(4 * 1024 * 4 <= ROWS_B_TILE, "partial scratch must fit the free ring slot");

// B operand: ROWS_NPB 16-byte units per thread and stage (128 columns x 64 k = 16 KB), lane-linear LDS image with the read
// swizzle applied to the source address (same images as gemm_dma.hip: rm_off / km_off_bf16)
template <bool KM>
struct DmaB {
  const char* ptr[ROWS_NPB]; int kofs[ROWS_NPB]; bool okx[ROWS_NPB]; int64_t kstep;
  DEVFN void init(const char* g, int64_t ld, int64_t x0, int64_t X, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS_NPB; ++i) {
      const int u = i * ROWS_NT + tid;
      int64_t x;
      if (!KM) {
        const int r = u >> 3, ls = ((u & 7) ^ r) & 7;
        x = x0 + r;
        kofs[i] = ls * 8;
        ptr[i] = g + (x * ld + ls * 8) * 2;
      } else {
        constexpr int VPR = ROWS_BN / 8, NB = ROWS_BN / 16;
        const int kr = u / VPR, cu = u % VPR;
        const int f = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int lb = ((cu >> 1) ^ f) & (NB - 1);
        x = x0 + lb * 16 + (cu & 1) * 8;
        kofs[i] = kr;
        ptr[i] = g + ((int64_t)kr * ld + x) * 2;
      }
      okx[i] = x < X;
    }
    kstep = KM ? 64 * ld * 2 : 128;
  }
  DEVFN void issue(int64_t kt, int64_t K, char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < ROWS_NPB; ++i) {
      const bool ok = okx[i] && (kt * 64 + kofs[i] < K);
      const char* src = ok ? ptr[i] + kt * kstep : (const char*)g_zero_page_rows;
      glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(img + (i * ROWS_NT + wave * 64) * 16)));
    }
  }
};

DEVFN f32x4 unpack4(const bf16x4& v) { return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
DEVFN bf16x4 pack4(const f32x4& v) { return (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }

// PRO = 1: LayerNorm forward prologue; PRO = 2: LayerNorm backward prologue.  H = K <= 256 NV, narrower rows handled by the
// `c < H` guards exactly like layernorm.hip (NV = 3 for H <= 768, NV = 4 for H <= 1024: ln_fwd_nv's / ln_bwd's choice).
template <typename OT, bool BKM, int PRO, int NV>
__global__ __launch_bounds__(ROWS_NT, 4) void gemm_rows_kernel(GemmP p, RowPro r, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RPW = ROWS_BM / 8;                      // rows per wave in the prologue (2)
  constexpr int HS = NV * 256;                          // stride of a per-wave column-partial slab
  const gstvd_ln_t& f = r.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = (int)f.H, KT = H / 64;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
  const int64_t m0 = (int64_t)mt * ROWS_BM, n0 = (int64_t)nt * ROWS_BN;
  char* A_img = smem;
  char* ring = smem + rows_kt_max(NV) * ROWS_A_TILE;
  float* scratch = (float*)(ring + (ROWS_NS - 1) * ROWS_B_TILE);       // the slot the prologue's DMAs do not touch

  // ---- B ring: the first stages fly while the prologue runs
  DmaB<BKM> ub;
  ub.init(p.B, p.ldb, n0, p.N, tid);
#pragma unroll
  for (int s = 0; s < ROWS_NS - 1; ++s) ub.issue(s, p.K, ring + s * ROWS_B_TILE, wave);

  // ---- prologue: two rows per wave, all loads first (one memory round trip), then row by row
  const DropKey dpre = make_drop(f.p_pre, f.site_pre, f.rng);
  bf16x4 px[RPW][NV], pr[RPW][NV], pd[RPW][NV];
  float mean_[RPW], rstd_[RPW];
  const bool has_res = f.res != nullptr;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int64_t row = m0 + wave * RPW + j;
    const bool rv = row < f.M;
    mean_[j] = 0.f; rstd_[j] = 0.f;
    if (PRO == 2 && rv) { mean_[j] = f.mean[row]; rstd_[j] = f.rstd[row]; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      const bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      px[j][i] = pr[j][i] = pd[j][i] = z;
      if (rv && c < H) {
        px[j][i] = *(const bf16x4*)((const bf16*)f.x + row * f.ldx + c);
        if (has_res) pr[j][i] = *(const bf16x4*)((const bf16*)f.res + row * f.ldres + c);
        if (PRO == 2) pd[j][i] = *(const bf16x4*)((const bf16*)r.dy + row * r.lddy + c);
      }
    }
  }
  f32x4 gam[NV], bet[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    gam[i] = bet[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H && (PRO == 1 || NV <= 3)) gam[i] = *(const f32x4*)(f.gamma + c);
    if (c < H && PRO == 1) bet[i] = *(const f32x4*)(f.beta + c);
  }
  // column partials: a column tile emits ONE of the three vectors (tile 0: sum dy * xhat, 1: sum dy, 2: sum dx), so one accumulator
  // (same sums in the same order as three would give; 8 NV registers less -- what lets NV = 4 stay inside 128)
  f32x4 mv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) mv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int rl = wave * RPW + j;                    // row inside the block
    const int64_t row = m0 + rl;
    const bool rv = row < f.M;
    f32x4 h[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      h[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rv && c < H) {
        h[i] = unpack4(px[j][i]);
        h[i] *= drop_factor4(dpre, (uint64_t)(row * f.H + c));
        if (has_res) h[i] += unpack4(pr[j][i]);
      }
    }
    bf16x4 aout[NV];
    if (PRO == 1) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) s += h[i][0] + h[i][1] + h[i][2] + h[i][3];
      const float mean = wave_sum(s) / (float)H;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < H) {
          const f32x4 d = h[i] - mean;
          v += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
      }
      const float var = wave_sum(v) / (float)H;
      const float rstd = 1.0f / sqrtf(var + f.eps);
      if (nt == 1 % ntn && rv && lane == 0) { f.mean[row] = mean; f.rstd[row] = rstd; }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        const f32x4 y = gam[i] * ((h[i] - mean) * rstd) + bet[i];
        aout[i] = pack4(rv ? y : (f32x4){0.f, 0.f, 0.f, 0.f});
        if (nt == 0 && rv && c < H) *(bf16x4*)((bf16*)f.y + row * f.ldy + c) = aout[i];
      }
    } else {
      f32x4 xh[NV], gy[NV], dyv[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        xh[i] = gy[i] = dyv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (rv && c < H) {
          xh[i] = (h[i] - mean_[j]) * rstd_[j];
          dyv[i] = unpack4(pd[j][i]);
          gy[i] = dyv[i] * (NV > 3 ? *(const f32x4*)(f.gamma + c) : gam[i]);     // (NV = 4: 16 registers the prologue does not have)
          s1 += gy[i][0] + gy[i][1] + gy[i][2] + gy[i][3];
          const f32x4 t = gy[i] * xh[i];
          s2 += t[0] + t[1] + t[2] + t[3];
        }
      }
      const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
        if (rv && c < H) {
          const f32x4 dh = (gy[i] - c1 - xh[i] * c2) * rstd_[j];
          dx = dh * drop_factor4(dpre, (uint64_t)(row * f.H + c));
          mv[i] += nt == 0 ? dyv[i] * xh[i] : (nt == 1 ? dyv[i] : dx);
          if (nt == 3 % ntn && r.dres) *(bf16x4*)((bf16*)r.dres + row * r.lddres + c) = pack4(dh);
          if (nt == 4 % ntn && r.dx && (r.dx != r.dres || dpre.on)) *(bf16x4*)((bf16*)r.dx + row * r.lddx + c) = pack4(dx);
        }
        aout[i] = pack4(dx);
      }
    }
    // row rl of the resident A image: columns c .. c+3 sit in K-tile c / 64, 16-byte slot (c % 64) / 8 (swizzled like rm_off)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < H) {
        const int kt = c >> 6, slot = (c & 63) >> 3;
        *(bf16x4*)(A_img + kt * ROWS_A_TILE + rm_off(rl, slot) + (c & 7) * 2) = aout[i];
      }
    }
  }
  if (PRO == 2 && nt < 3 && r.partial) {
    // column partials of this 64-row block: column tile 0 -> sum dy * xhat (dgamma), 1 -> sum dy (dbeta), 2 -> sum dx (dbias)
    // eight per-wave vectors through four 3 KB slabs, in a fixed order (no LDS atomics: run-to-run bit identity): waves 4-7
    // park theirs, waves 0-3 add them to their own and park the sums, then all threads add the four slabs
    float* mine = scratch + (wave & 3) * HS;
    if (wave >= 4) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { const int c = lane * 4 + i * 256; if (c < H) *(f32x4*)(mine + c) = mv[i]; }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { const int c = lane * 4 + i * 256; if (c < H) mv[i] += *(const f32x4*)(mine + c); }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { const int c = lane * 4 + i * 256; if (c < H) *(f32x4*)(mine + c) = mv[i]; }
    }
    __syncthreads();
    float* out = r.partial + ((int64_t)mt * 3 + nt) * H;
    for (int c = tid * 4; c < H; c += ROWS_NT * 4) {
      f32x4 a = *(const f32x4*)(scratch + c);
#pragma unroll
      for (int w = 1; w < 4; ++w) a += *(const f32x4*)(scratch + w * HS + c);
      *(f32x4*)(out + c) = a;
    }
  }
  // the prologue's global stores (y / dres / dx / partials) share vmcnt with the ring's LDS-DMA loads and may retire out of order
  // with them: drain everything once (the ring's first stages have had the whole prologue to land), then count DMAs only
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                      // the A image is complete

  // ---- K loop: A fragments from the resident image, B through the ring; wave w owns output columns 16 w .. 16 w + 15
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int slot = 0, fill = ROWS_NS - 1;
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROWS_NS - 2) * ROWS_NPB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ub.issue(kt + ROWS_NS - 1, p.K, ring + fill * ROWS_B_TILE, wave);
    const char* cA = A_img + kt * ROWS_A_TILE;
    const char* cB = ring + slot * ROWS_B_TILE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 fb = frag_bf16<ROWS_BN, BKM>(cB, wave * 16, kk, lane);
      const bf16x8 fa = frag_bf16<ROWS_BM, false>(cA, 0, kk, lane);
      acc = mfma_bf16_k32(fb, fa, acc);
    }
    slot = (slot + 1 == ROWS_NS) ? 0 : slot + 1;
    fill = (fill + 1 == ROWS_NS) ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  gemm_epilogue_tile<bf16, OT>(p, dk, acc, 0, m0 + li, n0 + wave * 16 + 4 * g);
}

template <typename OT, bool BKM, int PRO, int NV>
static int rows_launch_nv(const GemmP& p, const RowPro& r, hipStream_t s) {
  auto k = gemm_rows_kernel<OT, BKM, PRO, NV>;
  static int attr_rc = ensure_lds(k, rows_lds(NV));
  if (attr_rc) return attr_rc;
  const int ntm = (int)((p.M + ROWS_BM - 1) / ROWS_BM), ntn = (int)((p.N + ROWS_BN - 1) / ROWS_BN);
  GSTVD_LAUNCH(k, dim3((unsigned)(ntm * ntn)), dim3(ROWS_NT), rows_lds(NV), s, p, r, ntn);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
template <typename OT, bool BKM, int PRO>
static int rows_launch(const GemmP& p, const RowPro& r, hipStream_t s) {
  return r.f.H <= 768 ? rows_launch_nv<OT, BKM, PRO, 3>(p, r, s) : rows_launch_nv<OT, BKM, PRO, 4>(p, r, s);
}

static int rows_check(const gstvd_gemm_t* g, const gstvd_ln_t& f, GemmP& p) {
  if (!g || !g->B || !g->C) return GSTVD_E_NULL;
  if (g->dtype_in != GSTVD_BF16 || f.dtype != GSTVD_BF16 || f.mode != GSTVD_LN_RESID) return GSTVD_E_UNSUPPORTED;
  if (g->dtype_out != GSTVD_BF16 && g->dtype_out != GSTVD_F32) return GSTVD_E_DTYPE;
  if (g->batch != 1 || g->a_kmajor) return GSTVD_E_UNSUPPORTED;
  if (f.H != g->K || f.M != g->M || f.H % 64 || f.H > 64 * rows_kt_max(4) || f.M > 1024 || f.p_post > 0.f) return GSTVD_E_UNSUPPORTED;
  if (g->N < 5 * ROWS_BN || (g->N % 8)) return GSTVD_E_UNSUPPORTED;        // the side outputs are spread over five column tiles
  if (!f.x || !f.gamma || !f.mean || !f.rstd) return GSTVD_E_NULL;
  if ((f.ldx % 4) || (f.res && (f.ldres % 4)) || (g->ldb % 8) || (g->ldc % 4)) return GSTVD_E_ALIGN;
  if (((uintptr_t)g->B | (uintptr_t)g->C | (uintptr_t)f.x) & 15) return GSTVD_E_ALIGN;
  if ((g->epilogue & GSTVD_EPI_BIAS) && !g->bias) return GSTVD_E_NULL;
  if ((g->epilogue & GSTVD_EPI_ADD) && !g->addend) return GSTVD_E_NULL;
  if ((g->epilogue & (GSTVD_EPI_GELU | GSTVD_EPI_DGELU)) && !g->aux) return GSTVD_E_NULL;
  if (g->epilogue & (GSTVD_EPI_COLSUM | GSTVD_EPI_COLSUM_ACC)) return GSTVD_E_UNSUPPORTED;
  p.A = nullptr; p.B = (const char*)g->B; p.C = (char*)g->C;
  p.bias = g->bias; p.addend = (const char*)g->addend; p.aux = (char*)g->aux;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.lda = 0; p.ldb = g->ldb; p.ldc = g->ldc; p.ldadd = g->ldadd; p.ldaux = g->ldaux;
  p.sA = p.sB = p.sC = p.sAdd = p.sAux = 0;
  p.epi = g->epilogue; p.alpha = g->alpha; p.p = g->dropout_p; p.site = g->site; p.rng = g->rng;
  return 0;
}

extern "C" int gstvd_gemm_ln_fwd(const gstvd_gemm_t* g, const gstvd_ln_t* ln, gstvd_stream_t stream) {
  if (!ln) return GSTVD_E_NULL;
  GemmP p;
  const int rc = rows_check(g, *ln, p);
  if (rc) return rc;
  if (!ln->beta || !ln->y) return GSTVD_E_NULL;
  if ((ln->ldy % 4) || ((uintptr_t)ln->y & 7)) return GSTVD_E_ALIGN;
  RowPro r{};
  r.f = *ln;
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = g->dtype_out == GSTVD_F32;
  if (!g->b_kmajor) return f32 ? rows_launch<float, false, 1>(p, r, s) : rows_launch<bf16, false, 1>(p, r, s);
  return f32 ? rows_launch<float, true, 1>(p, r, s) : rows_launch<bf16, true, 1>(p, r, s);
}

extern "C" int gstvd_gemm_ln_bwd(const gstvd_gemm_t* g, const gstvd_ln_bwd_t* lb, gstvd_stream_t stream) {
  if (!lb) return GSTVD_E_NULL;
  GemmP p;
  const int rc = rows_check(g, lb->f, p);
  if (rc) return rc;
  if (!lb->dy || !lb->partial) return GSTVD_E_NULL;
  if ((lb->lddy % 4) || (lb->dres && (lb->lddres % 4)) || (lb->dx && (lb->lddx % 4))) return GSTVD_E_ALIGN;
  if (lb->nblk != (lb->f.M + ROWS_BM - 1) / ROWS_BM) return GSTVD_E_SHAPE;
  RowPro r{};
  r.f = lb->f;
  r.dy = lb->dy; r.lddy = lb->lddy; r.dres = lb->dres; r.lddres = lb->lddres; r.dx = lb->dx; r.lddx = lb->lddx; r.partial = lb->partial;
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = g->dtype_out == GSTVD_F32;
  if (!g->b_kmajor) return f32 ? rows_launch<float, false, 2>(p, r, s) : rows_launch<bf16, false, 2>(p, r, s);
  return f32 ? rows_launch<float, true, 2>(p, r, s) : rows_launch<bf16, true, 2>(p, r, s);
}

extern "C" int64_t gstvd_gemm_ln_rows_per_block(void) { return ROWS_BM; }
