// Fused dropout + residual + LayerNorm (forward/backward) and the two embedding front ends of the
// ViLBERT-dialog encoder / BERT-generation decoder.  HBM-bound: one wave64 per row, 4-element (8/16 B)
// vector accesses, row statistics by wave shuffles, dropout masks regenerated from the counter hash.
// Reference arithmetic: models/vilbert_dialog.py:283-296 (TF-style LN), :324-352 (BertEmbeddingsDialog),
// :1420-1427 (BertImageEmbeddings), :416-420 / :458-462 / :735-742 (dense -> dropout -> LN(x + residual)).
#include "common.h"

struct LnP {
  gstvd_ln_t f;
  // backward extras
  const void* dy; int64_t lddy; void* dres; int64_t lddres; void* dx; int64_t lddx; float* partial;
  float *dword, *dpos, *dtt, *dtt_ext;
  int64_t nblk_wide;       // > 0: the caller sized `partial` for the 16-wave geometry
};

// rows per wave in backward: 2 (both rows' loads in flight before any reduction) for large M; 1 for small M, where the
// grid cannot fill the chip anyway and the shorter per-wave chain is what counts (decoder / vision rows)
static inline int ln_bwd_rw(int64_t M) { return M <= 8192 ? 1 : 2; }
// waves per block in backward.  Every block writes one [3][H] slab of column partials; with 4 one-row waves per block that
// was 9.4 MB of partials for a 4096x768 LayerNorm -- more than any one of its operands -- read again by the column-sum launch
// (0.34 ms per step).  From 2048 rows up a block has 16 waves (one block per CU, 147 KB of LDS for the per-wave slabs, same
// number of rows in flight on the chip): a quarter of the partial bytes.  The LayerNorm kernel itself takes the same time.
static inline int ln_bwd_nw(int64_t M, int mode, int64_t H) { return (M >= 2048 && mode != GSTVD_LN_EMBED && H <= 768) ? 16 : 4; }

// h = pre-LayerNorm row, 4 elements starting at column c
template <typename T, int MODE>
DEVFN f32x4 ln_prologue(const gstvd_ln_t& f, int64_t row, int c, const DropKey& dpre,
                        int64_t id, int64_t tpos, int64_t seg, const float* locrow) {
  f32x4 h;
  if (MODE == GSTVD_LN_RESID) {
    h = ld4((const T*)f.x + row * f.ldx + c);
    h *= drop_factor4(dpre, (uint64_t)(row * f.H + c));
    if (f.res) h += ld4((const T*)f.res + row * f.ldres + c);
  } else if (MODE == GSTVD_LN_EMBED) {
    h = *(const f32x4*)(f.word + id * f.H + c) + *(const f32x4*)(f.pos + tpos * f.H + c);
    const float* trow = (seg < f.type_vocab) ? f.tt + seg * f.H : f.tt_ext + (seg - f.type_vocab) * f.H;
    h += *(const f32x4*)(trow + c);
  } else {
    h = ld4((const T*)f.x + row * f.ldx + c);
    f32x4 l = *(const f32x4*)(f.b_loc + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* w = f.w_loc + (int64_t)(c + e) * 5;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) a += locrow[j] * w[j];
      l[e] += a;
    }
    h += l;
  }
  return h;
}

template <typename T, int MODE, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(gstvd_ln_t f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= f.M) return;
  const int H = (int)f.H;
  const DropKey dpre = make_drop(MODE == GSTVD_LN_RESID ? f.p_pre : 0.f, f.site_pre, f.rng);
  const DropKey dpost = make_drop(f.p_post, f.site_post, f.rng);
  int64_t id = 0, tpos = 0, seg = 0;
  float locrow[5] = {0, 0, 0, 0, 0};
  if (MODE == GSTVD_LN_EMBED) { id = f.ids[row]; tpos = row % f.T + f.pos_offset; seg = f.segs ? f.segs[row] : 0; }
  if (MODE == GSTVD_LN_IMAGE) {
#pragma unroll
    for (int j = 0; j < 5; ++j) locrow[j] = f.loc[row * 5 + j];
  }
  f32x4 h[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    h[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      h[i] = ln_prologue<T, MODE>(f, row, c, dpre, id, tpos, seg, locrow);
      s += h[i][0] + h[i][1] + h[i][2] + h[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < H) {
      f32x4 d = h[i] - mean;
      v += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  const float var = wave_sum(v) / (float)H;
  const float rstd = 1.0f / sqrtf(var + f.eps);
  if (lane == 0) { f.mean[row] = mean; f.rstd[row] = rstd; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < H) {
      f32x4 y = *(const f32x4*)(f.gamma + c) * ((h[i] - mean) * rstd) + *(const f32x4*)(f.beta + c);
      y *= drop_factor4(dpost, (uint64_t)(row * f.H + c));
      st4((T*)f.y + row * f.ldy + c, y);
    }
  }
}

// Backward: a block owns 4*RW consecutive rows, each wave RW of them (RW = 1 up to 8192 rows: measured 17.1 -> 13.0 us
// at 4096x768 and 13.4 -> 8.0 us at 400x768; RW = 2 beyond).  With RW = 2 both rows' loads are issued
// before any reduction (memory-level parallelism), column partial sums stay in registers and are combined
// across the 4 waves through LDS with plain stores (no LDS atomics), one [3][H] slab per block goes to HBM.
template <typename T, int MODE, int NV, int RW, int NW = 4>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(LnP p) {
  constexpr int LN_BWD_RPB = NW * RW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;                   // [NW waves][3][H]
  const gstvd_ln_t& f = p.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = (int)f.H;
  const DropKey dpre = make_drop(MODE == GSTVD_LN_RESID ? f.p_pre : 0.f, f.site_pre, f.rng);
  const DropKey dpost = make_drop(f.p_post, f.site_post, f.rng);
  constexpr int NVP = (MODE == GSTVD_LN_EMBED) ? 4 : 3;      // partial vectors per block that go to HBM
  constexpr bool EXT = MODE == GSTVD_LN_EMBED && NV <= 4;    // embedding mode, H <= 1024: position slab + word strip fit the LDS
  constexpr int NVL = EXT ? 5 : NVP;                         // vectors per wave that go through LDS (+ the block's position row)
  f32x4 ag[NV], ab[NV], ax[NV], a4[NV], ap[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag[i] = ab[i] = ax[i] = a4[i] = ap[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Embedding mode, whole batches of rows per block (M = B * T, B a multiple of the block's rows): a block takes rows of ONE
  // position t -- batch rows b0 .. b0 + RPB - 1 -- so their position-embedding gradients are summed in the block and reach
  // dpos[t] as one atomic add per element and block instead of one per row (B-way contention on every address of a table of
  // T rows: the kernel is the last one of backward, 0.1 ms for 4096 rows).
  const int64_t nbat = (MODE == GSTVD_LN_EMBED && f.T > 0) ? f.M / f.T : 0;
  const bool pos_block = EXT && nbat > 0 && nbat * f.T == f.M && nbat % LN_BWD_RPB == 0;
  const int64_t bpt = pos_block ? nbat / LN_BWD_RPB : 1;     // blocks per position
  const int64_t row0 = pos_block ? ((int64_t)(blockIdx.x % bpt) * LN_BWD_RPB + wave * RW) * f.T + blockIdx.x / bpt
                                 : (int64_t)blockIdx.x * LN_BWD_RPB + wave * RW;
  const int64_t rstep = pos_block ? f.T : 1;                 // distance between a wave's consecutive rows
  float* wstrip = red + (int64_t)NW * NVL * H + (int64_t)wave * H;          // embedding mode: one row of dh per wave
  f32x4 xh[RW][NV], gy[RW][NV], dyv[RW][NV];
  float s1[RW] = {}, s2[RW] = {}, rstd[RW] = {};
  int64_t id[RW] = {}, tpos[RW] = {}, seg[RW] = {};
  bool rv[RW];
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int64_t row = row0 + j * rstep;
    rv[j] = row < f.M;
    float locrow[5] = {0, 0, 0, 0, 0};
    float mean = 0.f;
    if (rv[j]) {
      if (MODE == GSTVD_LN_EMBED) { id[j] = f.ids[row]; tpos[j] = row % f.T + f.pos_offset; seg[j] = f.segs ? f.segs[row] : 0; }
      if (MODE == GSTVD_LN_IMAGE) {
#pragma unroll
        for (int q = 0; q < 5; ++q) locrow[q] = f.loc[row * 5 + q];
      }
      mean = f.mean[row];
      rstd[j] = f.rstd[row];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      xh[j][i] = gy[j][i] = dyv[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rv[j] && c < H) {
        f32x4 h = ln_prologue<T, MODE>(f, row, c, dpre, id[j], tpos[j], seg[j], locrow);
        xh[j][i] = (h - mean) * rstd[j];
        dyv[j][i] = ld4((const T*)p.dy + row * p.lddy + c) * drop_factor4(dpost, (uint64_t)(row * f.H + c));
        gy[j][i] = dyv[j][i] * *(const f32x4*)(f.gamma + c);
        s1[j] += gy[j][i][0] + gy[j][i][1] + gy[j][i][2] + gy[j][i][3];
        f32x4 t = gy[j][i] * xh[j][i];
        s2[j] += t[0] + t[1] + t[2] + t[3];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int64_t row = row0 + j * rstep;
    const float c1 = wave_sum(s1[j]) / (float)H, c2 = wave_sum(s2[j]) / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      if (rv[j] && c < H) {
        f32x4 dh = (gy[j][i] - c1 - xh[j][i] * c2) * rstd[j];
        ag[i] += dyv[j][i] * xh[j][i];
        ab[i] += dyv[j][i];
        if (MODE == GSTVD_LN_RESID) {
          if (p.dres) st4((T*)p.dres + row * p.lddres + c, dh);
          f32x4 dx = dh * drop_factor4(dpre, (uint64_t)(row * f.H + c));
          if (p.dx && (p.dx != p.dres || dpre.on)) st4((T*)p.dx + row * p.lddx + c, dx);
          ax[i] += dx;
        } else if (MODE == GSTVD_LN_IMAGE) {
          st4((T*)p.dres + row * p.lddres + c, dh);
          ax[i] += dh;
        } else {
          // token-type rows 0 / 1 receive a contribution from (almost) every token: reduce them through the block
          // partials instead of ~M*H atomics onto two rows; rarer segment ids keep the atomic path
          if (seg[j] == 0) ax[i] += dh;
          else if (seg[j] == 1 && f.type_vocab > 1) a4[i] += dh;
          else {
            float* tg = (seg[j] < f.type_vocab) ? p.dtt + seg[j] * f.H : p.dtt_ext + (seg[j] - f.type_vocab) * f.H;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(tg + c + e, dh[e]);
          }
          // padded positions have an exactly-zero gradient (their keys are masked everywhere): skip their atomics,
          // which would all hit the [PAD] word row
          if (dh[0] != 0.f || dh[1] != 0.f || dh[2] != 0.f || dh[3] != 0.f) {
            if (!pos_block) {
#pragma unroll
              for (int e = 0; e < 4; ++e) atomicAdd(p.dpos + tpos[j] * f.H + c + e, dh[e]);
            } else {
              ap[i] += dh;
            }
          }
          // word-embedding row: through the wave's LDS strip, so that one atomic instruction covers 64 CONSECUTIVE columns
          // (two full 128-byte lines) instead of every fourth dword of a 1 KB span (eight lines a quarter used each)
          if (EXT) {
            *(f32x4*)(wstrip + c) = dh;
          } else if (dh[0] != 0.f || dh[1] != 0.f || dh[2] != 0.f || dh[3] != 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(p.dword + id[j] * f.H + c + e, dh[e]);
          }
        }
      }
    }
    if (EXT && rv[j]) {
      // (same wave wrote the strip: LDS operations of one wave complete in order)
      float* wrow = p.dword + id[j] * f.H;
      for (int c = lane; c < H; c += 64) {
        const float v = wstrip[c];
        if (v != 0.f) atomicAdd(wrow + c, v);
      }
    }
  }
  float* mine = red + (int64_t)wave * NVL * H;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < H) {
      *(f32x4*)(mine + c) = ag[i];
      *(f32x4*)(mine + H + c) = ab[i];
      *(f32x4*)(mine + 2 * H + c) = ax[i];
      if (NVP == 4) *(f32x4*)(mine + 3 * H + c) = a4[i];
      if (NVL == 5) *(f32x4*)(mine + 4 * H + c) = ap[i];
    }
  }
  __syncthreads();
  float* out = p.partial + (int64_t)blockIdx.x * NVP * H;
  for (int i = threadIdx.x * 4; i < NVP * H; i += NW * 256) {
    f32x4 a = *(const f32x4*)(red + i);
#pragma unroll
    for (int w = 1; w < NW; ++w) a += *(const f32x4*)(red + w * NVL * H + i);
    *(f32x4*)(out + i) = a;
  }
  if (NVL == 5 && pos_block) {                               // the block's position row: one atomic add per element
    const int64_t t = (int64_t)(blockIdx.x / bpt) + f.pos_offset;
    for (int i = threadIdx.x * 4; i < H; i += NW * 256) {
      f32x4 a = *(const f32x4*)(red + 4 * H + i);
#pragma unroll
      for (int w = 1; w < NW; ++w) a += *(const f32x4*)(red + w * NVL * H + 4 * H + i);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (a[e] != 0.f) atomicAdd(p.dpos + t * f.H + i + e, a[e]);
    }
  }
}

// out_j[c] (+)= sum_b partial[b][j][c]; block = 64 columns x 4 row groups
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* partial, int64_t nblk, int64_t nvec, int64_t H,
                                                              float* o0, float* o1, float* o2, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + cl;      // over nvec*H
  const int64_t W = nvec * H;
  float a = 0.f;
  if (col < W) {
    const int64_t j = col / H, c = col % H;
    const float* src = partial + j * H + c;
    for (int64_t b = rg; b < nblk; b += 4) a += src[b * 3 * H];
  }
  red[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && col < W) {
    a = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    const int64_t j = col / H, c = col % H;
    float* o = j == 0 ? o0 : (j == 1 ? o1 : o2);
    if (o) o[c] = accumulate ? o[c] + a : a;
  }
}

// stage 1 of a plain column sum: block = 64-row slab x 256 columns -> partial[slab][N] (stride 3*N like LN partials? no: N)
template <typename T>
__global__ __launch_bounds__(256) void colsum_slab_kernel(const T* x, int64_t ldx, int64_t M, int64_t N, float* partial) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * 64;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (c < N) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      int64_t r = r0 + wave + 4 * i;
      if (r < M) a += ld4(x + r * ldx + c);
    }
  }
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < N) {
    a = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    *(f32x4*)(partial + (int64_t)blockIdx.y * N + c) = a;
  }
}
// table-driven form of the slab stage: every pending bias-gradient column sum of a backward pass in one launch
template <typename T>
__global__ __launch_bounds__(256) void colsum_slab_batched_kernel(const gstvd_slab_entry_t* tab, int nent) {
  __shared__ f32x4 red[4][64];
  int lo = 0, hi = nent - 1;
  const int b = blockIdx.x;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (tab[mid].blk0 <= b) lo = mid; else hi = mid - 1; }
  const gstvd_slab_entry_t e = tab[lo];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncb = (int)((e.N + 255) / 256), local = b - e.blk0;
  const int64_t c = (int64_t)(local % ncb) * 256 + lane * 4, slab = local / ncb, r0 = slab * 64;
  const T* x = (const T*)e.x;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (c < e.N) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      int64_t r = r0 + wave + 4 * i;
      if (r < e.M) a += ld4(x + r * e.ldx + c);
    }
  }
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < e.N) {
    a = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    *(f32x4*)(e.scratch + slab * e.N + c) = a;
  }
}

// stage 2: out[c] (+)= sum_s partial[s][c]
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int64_t nslab, int64_t N, float* out, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < N) for (int64_t s = rg; s < nslab; s += 4) a += partial[s * N + c];
  red[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && c < N) {
    a = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    out[c] = accumulate ? out[c] + a : a;
  }
}

// dW_loc[h][j] += sum_m dh[m][h] loc[m][j]; grid (ceil(H/256), MS) with atomics
template <typename T>
__global__ __launch_bounds__(256) void locgrad_kernel(const T* dh, int64_t lddh, const float* loc, int64_t M, int64_t H, float* dw) {
  const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t m0 = blockIdx.y * per, m1 = (m0 + per < M) ? m0 + per : M;
  float a[5] = {0, 0, 0, 0, 0};
  for (int64_t m = m0; m < m1; ++m) {
    const float d = to_f(dh[m * lddh + h]);
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] += d * loc[m * 5 + j];
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) atomicAdd(dw + h * 5 + j, a[j]);
}

// All pending column reductions of a backward pass in ONE launch: block b serves the 64-column slice
// (b - blk0) of the entry that contains it (binary search over the entries' first-block offsets).
__global__ __launch_bounds__(256) void colsum_batched_kernel(const gstvd_colsum_entry_t* tab, int nent) {
  __shared__ float red[4][64];
  int lo = 0, hi = nent - 1;
  const int b = blockIdx.x;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (tab[mid].blk0 <= b) lo = mid; else hi = mid - 1; }
  const gstvd_colsum_entry_t e = tab[lo];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)(b - e.blk0) * 64 + cl, W = (int64_t)e.nvec * e.H;
  float a = 0.f;
  if (col < W) {
    // eight independent loads in flight per thread: the embedding LayerNorm's 1024 partial rows used to be 256 dependent
    // load -> add rounds per thread (118 us for a 48-block launch at the very end of backward)
    const float* src = e.partial + col;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t k = rg;
    for (; k + 28 < e.nblk; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p8[u] += src[(k + 4 * u) * e.stride];
    }
    for (; k < e.nblk; k += 4) a += src[k * e.stride];
    a += ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
  }
  red[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && col < W) {
    a = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    const int64_t j = col / e.H, c = col % e.H;
    // (selects, not e.out[j]: a dynamically indexed member put the whole entry on the stack -- 88 B/lane of scratch)
    float* o = j == 0 ? e.out[0] : (j == 1 ? e.out[1] : e.out[2]);
    const int acc = j == 0 ? e.accumulate[0] : (j == 1 ? e.accumulate[1] : e.accumulate[2]);
    if (o) o[c] = acc ? o[c] + a : a;
  }
}

// ---- host side -------------------------------------------------------------------------------------
static int ln_check(const gstvd_ln_t* p) {
  if (!p) return GSTVD_E_NULL;
  if (p->dtype != GSTVD_F32 && p->dtype != GSTVD_BF16) return GSTVD_E_DTYPE;
  if (p->M <= 0 || p->H <= 0 || (p->H % 4) || p->H > 2048) return GSTVD_E_SHAPE;
  if (!p->gamma || !p->beta || !p->mean || !p->rstd) return GSTVD_E_NULL;
  if (p->mode == GSTVD_LN_RESID) { if (!p->x) return GSTVD_E_NULL; }
  else if (p->mode == GSTVD_LN_EMBED) { if (!p->ids || !p->word || !p->pos || !p->tt || !p->tt_ext || p->T <= 0) return GSTVD_E_NULL; }
  else if (p->mode == GSTVD_LN_IMAGE) { if (!p->x || !p->loc || !p->w_loc || !p->b_loc) return GSTVD_E_NULL; }
  else return GSTVD_E_UNSUPPORTED;
  return 0;
}

template <typename T, int MODE>
static int ln_fwd_nv(const gstvd_ln_t& f, hipStream_t s) {
  dim3 grid((unsigned)((f.M + 3) / 4)), block(256);
  if (f.H <= 256) hipLaunchKernelGGL((ln_fwd_kernel<T, MODE, 1>), grid, block, 0, s, f);
  else if (f.H <= 768) hipLaunchKernelGGL((ln_fwd_kernel<T, MODE, 3>), grid, block, 0, s, f);
  else if (f.H <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<T, MODE, 4>), grid, block, 0, s, f);
  else hipLaunchKernelGGL((ln_fwd_kernel<T, MODE, 8>), grid, block, 0, s, f);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
template <typename T>
static int ln_fwd_mode(const gstvd_ln_t& f, hipStream_t s) {
  if (f.mode == GSTVD_LN_RESID) return ln_fwd_nv<T, GSTVD_LN_RESID>(f, s);
  if (f.mode == GSTVD_LN_EMBED) return ln_fwd_nv<T, GSTVD_LN_EMBED>(f, s);
  return ln_fwd_nv<T, GSTVD_LN_IMAGE>(f, s);
}
extern "C" int gstvd_ln_fwd(const gstvd_ln_t* p, gstvd_stream_t stream) {
  int rc = ln_check(p);
  if (rc) return rc;
  if (!p->y) return GSTVD_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  return p->dtype == GSTVD_BF16 ? ln_fwd_mode<bf16>(*p, s) : ln_fwd_mode<float>(*p, s);
}

extern "C" int64_t gstvd_ln_bwd_blocks(int64_t M) { const int rpb = 4 * ln_bwd_rw(M); return (M + rpb - 1) / rpb; }
extern "C" int64_t gstvd_ln_bwd_blocks_for(int64_t M, int64_t H, int32_t mode) {
  if (ln_bwd_nw(M, mode, H) == 16 && ln_bwd_rw(M) == 1) return (M + 15) / 16;
  return gstvd_ln_bwd_blocks(M);
}

template <typename T, int MODE>
static int ln_bwd_wide(const LnP& p, hipStream_t s) {       // 16 one-row waves per block, H <= 768
  constexpr int NW = 16;
  const size_t lds = (size_t)NW * 3 * p.f.H * sizeof(float);
  static int rc1 = (int)hipFuncSetAttribute((const void*)ln_bwd_kernel<T, MODE, 1, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 3 * 256 * 4);
  static int rc3 = (int)hipFuncSetAttribute((const void*)ln_bwd_kernel<T, MODE, 3, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 3 * 768 * 4);
  if (rc1 | rc3) return rc1 | rc3;
  dim3 grid((unsigned)((p.f.M + NW - 1) / NW)), block(NW * 64);
  if (p.f.H <= 256) hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 1, 1, NW>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 3, 1, NW>), grid, block, lds, s, p);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

template <typename T, int MODE, int RW>
static int ln_bwd_nv(const LnP& p, hipStream_t s) {
  dim3 grid((unsigned)gstvd_ln_bwd_blocks(p.f.M)), block(256);
  // embedding mode: 4 slabs, or (H <= 1024) 5 slabs + the word strip per wave
  size_t lds = (size_t)4 * (MODE == GSTVD_LN_EMBED ? (p.f.H <= 1024 ? 6 : 4) : 3) * p.f.H * sizeof(float);
  if (lds > 48 * 1024) {
    static int rc8 = (int)hipFuncSetAttribute((const void*)ln_bwd_kernel<T, MODE, 8, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * 2048 * 4)
                   | (int)hipFuncSetAttribute((const void*)ln_bwd_kernel<T, MODE, 4, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 6 * 1024 * 4)
                   | (int)hipFuncSetAttribute((const void*)ln_bwd_kernel<T, MODE, 3, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 6 * 768 * 4);
    if (rc8) return rc8;
  }
  if (p.f.H <= 256) hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 1, RW>), grid, block, lds, s, p);
  else if (p.f.H <= 768) hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 3, RW>), grid, block, lds, s, p);
  else if (p.f.H <= 1024) hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 4, RW>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((ln_bwd_kernel<T, MODE, 8, RW>), grid, block, lds, s, p);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
template <typename T, int MODE>
static int ln_bwd_rows(const LnP& p, hipStream_t s) {
  if constexpr (MODE != GSTVD_LN_EMBED) {
    if (p.nblk_wide > 0) return ln_bwd_wide<T, MODE>(p, s);
  }
  return ln_bwd_rw(p.f.M) == 1 ? ln_bwd_nv<T, MODE, 1>(p, s) : ln_bwd_nv<T, MODE, 2>(p, s);
}
template <typename T>
static int ln_bwd_mode(const LnP& p, hipStream_t s) {
  if (p.f.mode == GSTVD_LN_RESID) return ln_bwd_rows<T, GSTVD_LN_RESID>(p, s);
  if (p.f.mode == GSTVD_LN_EMBED) return ln_bwd_rows<T, GSTVD_LN_EMBED>(p, s);
  return ln_bwd_rows<T, GSTVD_LN_IMAGE>(p, s);
}
extern "C" int gstvd_ln_bwd(const gstvd_ln_bwd_t* b, gstvd_stream_t stream) {
  if (!b) return GSTVD_E_NULL;
  int rc = ln_check(&b->f);
  if (rc) return rc;
  if (!b->dy || !b->partial) return GSTVD_E_NULL;
  if (b->f.mode == GSTVD_LN_EMBED && (!b->dword || !b->dpos || !b->dtt || !b->dtt_ext)) return GSTVD_E_NULL;
  if (b->f.mode == GSTVD_LN_IMAGE && !b->dres) return GSTVD_E_NULL;
  LnP p;
  p.f = b->f; p.dy = b->dy; p.lddy = b->lddy; p.dres = b->dres; p.lddres = b->lddres;
  p.dx = b->dx; p.lddx = b->lddx; p.partial = b->partial;
  p.dword = b->dword; p.dpos = b->dpos; p.dtt = b->dtt; p.dtt_ext = b->dtt_ext;
  const int64_t wide = gstvd_ln_bwd_blocks_for(b->f.M, b->f.H, b->f.mode);
  p.nblk_wide = (b->nblk > 0 && b->nblk == wide && wide != gstvd_ln_bwd_blocks(b->f.M)) ? wide : 0;
  if (b->nblk > 0 && b->nblk != wide && b->nblk != gstvd_ln_bwd_blocks(b->f.M)) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  return b->f.dtype == GSTVD_BF16 ? ln_bwd_mode<bf16>(p, s) : ln_bwd_mode<float>(p, s);
}

extern "C" int gstvd_colsum_partials(const float* partial, int64_t nblk, int64_t nvec, int64_t H,
                                     float* out0, float* out1, float* out2, int32_t accumulate, gstvd_stream_t stream) {
  if (!partial) return GSTVD_E_NULL;
  if (nblk <= 0 || nvec <= 0 || nvec > 3 || H <= 0) return GSTVD_E_SHAPE;
  dim3 grid((unsigned)((nvec * H + 63) / 64));
  hipLaunchKernelGGL(colsum_partials_kernel, grid, dim3(256), 0, (hipStream_t)stream, partial, nblk, nvec, H, out0, out1, out2, accumulate);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_colsum_batched(const gstvd_colsum_entry_t* table_dev, int64_t nent, int64_t total_blocks, gstvd_stream_t stream) {
  if (!table_dev) return GSTVD_E_NULL;
  if (nent <= 0 || total_blocks <= 0) return GSTVD_E_SHAPE;
  hipLaunchKernelGGL(colsum_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, (int)nent);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

// stage 1 only of a plain column sum: scratch[slab][N] (slab = 64 rows); finish with gstvd_colsum_batched / _partials
extern "C" int gstvd_colsum_slabs(const void* x, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* scratch,
                                  int64_t scratch_elems, gstvd_stream_t stream) {
  if (!x || !scratch) return GSTVD_E_NULL;
  if (M <= 0 || N <= 0 || (N % 4)) return GSTVD_E_SHAPE;
  const int64_t nslab = (M + 63) / 64;
  if (scratch_elems < nslab * N) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)nslab);
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(colsum_slab_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, ldx, M, N, scratch);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(colsum_slab_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, M, N, scratch);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_colsum_slabs_batched(const gstvd_slab_entry_t* table_dev, int64_t nent, int64_t total_blocks, int32_t dtype,
                                          gstvd_stream_t stream) {
  if (!table_dev) return GSTVD_E_NULL;
  if (nent <= 0 || total_blocks <= 0) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(colsum_slab_batched_kernel<bf16>, dim3((unsigned)total_blocks), dim3(256), 0, s, table_dev, (int)nent);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(colsum_slab_batched_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, s, table_dev, (int)nent);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_colsum(const void* x, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* out,
                            float* scratch, int64_t scratch_elems, int32_t accumulate, gstvd_stream_t stream) {
  if (!x || !out || !scratch) return GSTVD_E_NULL;
  if (M <= 0 || N <= 0 || (N % 4)) return GSTVD_E_SHAPE;
  const int64_t nslab = (M + 63) / 64;
  if (scratch_elems < nslab * N) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)nslab);
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(colsum_slab_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, ldx, M, N, scratch);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(colsum_slab_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, M, N, scratch);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, scratch, nslab, N, out, accumulate);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_locgrad(const void* dh, int64_t lddh, const float* loc, int64_t M, int64_t H, int32_t dtype,
                             float* dw_loc, int32_t accumulate, gstvd_stream_t stream) {
  if (!dh || !loc || !dw_loc) return GSTVD_E_NULL;
  if (M <= 0 || H <= 0) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(dw_loc, 0, (size_t)H * 5 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid((unsigned)((H + 255) / 256), 16);
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(locgrad_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dh, lddh, loc, M, H, dw_loc);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(locgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)dh, lddh, loc, M, H, dw_loc);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}
