// LM-head cross entropy (visual_dialog_decoder.py:70-77), candidate scoring (evaluate_gen.py:94-106),
// casts, the dropout-mask probe used by tests, and the fused AdamW (pytorch_transformers 1.2.0 semantics).
// All HBM-bound: 16-byte vector accesses, one workgroup per logits row, deterministic reductions.
#include "common.h"
#include <math.h>

DEVFN float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* logits, int64_t ldl, const int64_t* labels, int64_t V,
                                                     int64_t ignore, float* row_loss, float* lse_out) {
  __shared__ float red[4];
  const int64_t m = blockIdx.x;
  const T* x = logits + m * ldl;
  const int64_t V4 = V & ~(int64_t)3;
  float mx = -INFINITY;
  for (int64_t c = threadIdx.x * 4; c < V4; c += 1024) {
    f32x4 v = ld4(x + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  for (int64_t c = V4 + threadIdx.x; c < V; c += 256) mx = fmaxf(mx, to_f(x[c]));
  mx = block_reduce(mx, red, true);
  float s = 0.f;
  for (int64_t c = threadIdx.x * 4; c < V4; c += 1024) {
    f32x4 v = ld4(x + c);
    s += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  for (int64_t c = V4 + threadIdx.x; c < V; c += 256) s += __expf(to_f(x[c]) - mx);
  s = block_reduce(s, red, false);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(s);
    lse_out[m] = lse;
    const int64_t lab = labels[m];
    row_loss[m] = (lab == ignore || lab < 0 || lab >= V) ? 0.f : lse - to_f(x[lab]);
  }
}

// stats[0] = sum(row_loss), stats[1] = #(label != ignore), stats[2] = mean; single block, fixed order => deterministic
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss, const int64_t* labels, int64_t M,
                                                        int64_t ignore, float* stats) {
  __shared__ float red[4];
  float s = 0.f, n = 0.f;
  for (int64_t i = threadIdx.x; i < M; i += 256) {
    s += row_loss[i];
    n += (labels[i] != ignore) ? 1.f : 0.f;
  }
  s = block_reduce(s, red, false);
  n = block_reduce(n, red, false);
  if (threadIdx.x == 0) { stats[0] = s; stats[1] = n; stats[2] = s / n; }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, int64_t ldl, const int64_t* labels, const float* lse,
                                                     const float* stats, const float* gscale, int mean, int64_t V,
                                                     int64_t ignore, T* dl, int64_t ldd) {
  const int64_t m = blockIdx.x;
  const T* x = logits + m * ldl;
  T* d = dl + m * ldd;
  const int64_t lab = labels[m];
  const bool keep = !(lab == ignore || lab < 0 || lab >= V);
  float gs = gscale ? gscale[0] : 1.f;
  if (mean) gs /= stats[1];
  const float l = lse[m];
  for (int64_t c = threadIdx.x * 4; c < ldd; c += 1024) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (keep) {
      if (c + 3 < V) {
        f32x4 v = ld4(x + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__expf(v[e] - l) - ((c + e) == lab ? 1.f : 0.f)) * gs;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < V) o[e] = (__expf(to_f(x[c + e]) - l) - ((c + e) == lab ? 1.f : 0.f)) * gs;
      }
    }
    st4(d + c, o);
  }
}

// score[row] = sum_u [tgt != 0] (logits[row,u,tgt] - lse[row,u]),  tgt = ids[row,u+1] (0 for the last u)
template <typename T>
__global__ __launch_bounds__(64) void answer_scores_kernel(const T* logits, int64_t ldl, const float* lse,
                                                           const int64_t* ids, int64_t U, float* scores) {
  const int64_t row = blockIdx.x;
  float a = 0.f;
  for (int64_t u = threadIdx.x; u < U; u += 64) {
    const int64_t tgt = (u + 1 < U) ? ids[row * U + u + 1] : 0;
    if (tgt != 0) a += to_f(logits[(row * U + u) * ldl + tgt]) - lse[row * U + u];
  }
  a = wave_sum(a);
  if (threadIdx.x == 0) scores[row] = a;
}

template <typename S, typename Dt>
__global__ __launch_bounds__(256) void cast_kernel(const S* src, Dt* dst, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) st4(dst + i, ld4(src + i));
  else for (int64_t j = i; j < n; ++j) dst[j] = from_f<Dt>(to_f(src[j]));
}

// fp32 -> bf16 over a LIST of ranges of two flat buffers with the same indexing (gstvd_cast_ranges): block b serves 1024 elements
// of the range whose block interval [blk0[i], blk0[i + 1]) holds it
__global__ __launch_bounds__(256) void cast_ranges_kernel(const float* src, bf16* dst, const int64_t* tab, const int32_t* blk0, int n) {
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (blk0[mid] <= b) lo = mid; else hi = mid - 1; }
  const int64_t start = tab[2 * lo], len = tab[2 * lo + 1];
  const int64_t i = ((int64_t)(b - blk0[lo]) * 256 + threadIdx.x) * 4;
  if (i + 3 < len) st4(dst + start + i, ld4(src + start + i));
  else for (int64_t j = i; j < len; ++j) dst[start + j] = (bf16)src[start + j];
}

__global__ void scale_kernel(float* x, const float* f, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= f[0];
}
__global__ void rng_advance_kernel(uint64_t* rng) { rng[1] += 1; }
__global__ void dropout_mask_kernel(float* out, int64_t n, float p, uint32_t site, const uint64_t* rng) {
  const DropKey dk = make_drop(p, site, rng);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = drop_factor(dk, (uint64_t)i);
}

// AdamW: pytorch_transformers==1.2.0 optimization.AdamW.step (train_gen.py:16,247).  Segments are 4-element aligned
// (the flat layout aligns every tensor to 64 elements), so one lookup serves a 16-byte vector.
// GT = float: grad is the flat fp32 gradient buffer (same indexing as param).  GT = bf16: grad points at the bf16 copy of
// the slice that was just all-reduced; element i of the flat buffers is grad[i - gorigin].
template <typename GT>
__global__ __launch_bounds__(256) void adamw_kernel(float* param, const GT* grad, float* m, float* v, bf16* shadow, int64_t n,
                                                    const int64_t* seg_end, const float* hp, int64_t nseg, float b1, float b2,
                                                    float eps, const float* step, float gscale, int64_t base, int64_t gorigin,
                                                    const int32_t* blocks, const uint8_t* seg_skip) {
  // the launch covers flat elements [base, n) of the buffers (pointers are the buffers' starts), or -- `blocks` given -- the
  // listed 1024-element blocks; segment ends are absolute
  __shared__ float s_bc;
  const int64_t i0 = blocks ? (int64_t)blocks[blockIdx.x] * 1024 : base + (int64_t)blockIdx.x * 1024;   // block-uniform: scalar loads below
  const int64_t i = i0 + threadIdx.x * 4;
  if (threadIdx.x == 0) s_bc = adamw_bias_correction(b1, b2, step[0]);     // once per block
  // streaming operands first (their latency overlaps the segment lookup); nothing here is re-read, keep it out of L2's way
  const bool vec = i + 3 < n;
  f32x4 g4 = {0.f, 0.f, 0.f, 0.f}, m4 = g4, v4 = g4, p4 = g4;
  if (vec) {
    g4 = ld4_stream(grad + (i - gorigin));
    m4 = __builtin_nontemporal_load((const f32x4*)(m + i));
    v4 = __builtin_nontemporal_load((const f32x4*)(v + i));
    p4 = __builtin_nontemporal_load((const f32x4*)(param + i));
  }
  int64_t lo = adamw_segment(seg_end, nseg, i0);        // first segment whose end > i0
  __syncthreads();
  if (i >= n || i < base) return;                    // (i < base: only with a block list, whose blocks may straddle the range's start)
  while (lo < nseg - 1 && seg_end[lo] <= i) ++lo;     // at most a few steps inside a 1024-element block
  const float lr = hp[2 * lo], wd = hp[2 * lo + 1];
  const float bc = s_bc;
  if (vec && seg_end[lo] >= i + 4) {
    if (lr == 0.f || (seg_skip && seg_skip[lo])) return;   // padding / frozen segment / updated by the weight-gradient launch
    adamw_update4(p4, m4, v4, g4, gscale, lr, wd, bc, b1, b2, eps);
    __builtin_nontemporal_store(m4, (f32x4*)(m + i));
    __builtin_nontemporal_store(v4, (f32x4*)(v + i));
    __builtin_nontemporal_store(p4, (f32x4*)(param + i));
    if (shadow) st4(shadow + i, p4);
    return;
  }
  int64_t seg = lo;
  for (int e = 0; e < 4 && i + e < n; ++e) {
    const int64_t k = i + e;
    while (seg < nseg - 1 && seg_end[seg] <= k) ++seg;
    const float lr_ = hp[2 * seg], wd_ = hp[2 * seg + 1];
    if (lr_ == 0.f || (seg_skip && seg_skip[seg])) continue;
    f32x4 g1 = {(float)grad[k - gorigin], 0.f, 0.f, 0.f}, m1 = {m[k], 0.f, 0.f, 0.f}, v1 = {v[k], 0.f, 0.f, 0.f}, p1 = {param[k], 0.f, 0.f, 0.f};
    adamw_update4(p1, m1, v1, g1, gscale, lr_, wd_, bc, b1, b2, eps);
    m[k] = m1[0]; v[k] = v1[0]; param[k] = p1[0];
    if (shadow) shadow[k] = (bf16)p1[0];
  }
}

// backward of VLFusion's concat + dropout (visual_dialog_model.py:132-133): split d_enc[B, R+T, H] into the
// vision rows [B*R, H] and the text rows [B*T, H], re-applying each half's dropout mask
template <typename T>
__global__ __launch_bounds__(256) void vl_split_kernel(const T* d, int64_t B, int64_t R, int64_t Tt, int64_t H, T* dv, T* dt_,
                                                       float p, uint32_t site_v, uint32_t site_t, const uint64_t* rng) {
  const DropKey kv = make_drop(p, site_v, rng), kt = make_drop(p, site_t, rng);
  const int64_t S = R + Tt, H4 = H / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * S * H4) return;
  const int64_t c = (i % H4) * 4, row = i / H4, b = row / S, r = row % S;
  f32x4 v = ld4(d + row * H + c);
  if (r < R) { const int64_t o = (b * R + r) * H + c; st4(dv + o, v * drop_factor4(kv, (uint64_t)o)); }
  else { const int64_t o = (b * Tt + (r - R)) * H + c; st4(dt_ + o, v * drop_factor4(kt, (uint64_t)o)); }
}

// ---- C ABI ----------------------------------------------------------------------------------------------
extern "C" int gstvd_vl_split(const void* d_enc, int64_t B, int64_t R, int64_t T, int64_t H, int32_t dtype, void* d_v, void* d_t,
                              float p, uint32_t site_v, uint32_t site_t, const uint64_t* rng, gstvd_stream_t stream) {
  if (!d_enc || !d_v || !d_t) return GSTVD_E_NULL;
  if (B <= 0 || R <= 0 || T <= 0 || H <= 0 || (H % 4)) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((B * (R + T) * (H / 4) + 255) / 256));
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(vl_split_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_enc, B, R, T, H, (bf16*)d_v, (bf16*)d_t, p, site_v, site_t, rng);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(vl_split_kernel<float>, grid, dim3(256), 0, s, (const float*)d_enc, B, R, T, H, (float*)d_v, (float*)d_t, p, site_v, site_t, rng);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

// 2: gstvd_ln_bwd_t.nblk (round 2); 3: gstvd_gemm_ln_fwd / _bwd, debug entry gone (round 3); 4: gstvd_gemm_grouped_adamw,
// gstvd_adamw_blocks (round 4); 5: block_map_dev / nblocks of the grouped launches (round 5)
extern "C" int gstvd_abi_version(void) { return 7; }
extern "C" const char* gstvd_build_arch(void) { return "gfx950"; }

extern "C" int gstvd_ce_fwd(const void* logits, int64_t ldl, const int64_t* labels, int64_t M, int64_t V, int64_t ignore_index,
                            int32_t dtype, float* row_loss, float* lse, float* stats, gstvd_stream_t stream) {
  if (!logits || !labels || !row_loss || !lse || !stats) return GSTVD_E_NULL;
  if (M <= 0 || V <= 0 || (ldl % 4)) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(ce_fwd_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (const bf16*)logits, ldl, labels, V, ignore_index, row_loss, lse);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)logits, ldl, labels, V, ignore_index, row_loss, lse);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, labels, M, ignore_index, stats);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_ce_bwd(const void* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* stats,
                            const float* gscale, int32_t mean, int64_t M, int64_t V, int64_t ignore_index, int32_t dtype,
                            void* dlogits, int64_t ldd, gstvd_stream_t stream) {
  if (!logits || !labels || !lse || !stats || !dlogits) return GSTVD_E_NULL;
  if (M <= 0 || V <= 0 || (ldl % 4) || (ldd % 4) || ldd < V) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(ce_bwd_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (const bf16*)logits, ldl, labels, lse, stats, gscale, mean, V, ignore_index, (bf16*)dlogits, ldd);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)logits, ldl, labels, lse, stats, gscale, mean, V, ignore_index, (float*)dlogits, ldd);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_answer_scores(const void* logits, int64_t ldl, const float* lse, const int64_t* dec_ids, int64_t rows,
                                   int64_t U, int32_t dtype, float* scores, gstvd_stream_t stream) {
  if (!logits || !lse || !dec_ids || !scores) return GSTVD_E_NULL;
  if (rows <= 0 || U <= 0) return GSTVD_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSTVD_BF16) hipLaunchKernelGGL(answer_scores_kernel<bf16>, dim3((unsigned)rows), dim3(64), 0, s, (const bf16*)logits, ldl, lse, dec_ids, U, scores);
  else if (dtype == GSTVD_F32) hipLaunchKernelGGL(answer_scores_kernel<float>, dim3((unsigned)rows), dim3(64), 0, s, (const float*)logits, ldl, lse, dec_ids, U, scores);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_cast(const void* src, int32_t sdt, void* dst, int32_t ddt, int64_t n, gstvd_stream_t stream) {
  if (!src || !dst) return GSTVD_E_NULL;
  if (n <= 0) return GSTVD_E_SHAPE;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return GSTVD_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((n + 1023) / 1024));
  if (sdt == GSTVD_F32 && ddt == GSTVD_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)src, (bf16*)dst, n);
  else if (sdt == GSTVD_BF16 && ddt == GSTVD_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, dim3(256), 0, s, (const bf16*)src, (float*)dst, n);
  else if (sdt == GSTVD_F32 && ddt == GSTVD_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (sdt == GSTVD_BF16 && ddt == GSTVD_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n);
  else return GSTVD_E_DTYPE;
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_cast_ranges(const float* src, void* dst_bf16, const int64_t* ranges_dev, const int32_t* blk0_dev, int64_t nranges,
                                 int64_t total_blocks, gstvd_stream_t stream) {
  if (!src || !dst_bf16 || !ranges_dev || !blk0_dev) return GSTVD_E_NULL;
  if (nranges <= 0 || total_blocks <= 0 || nranges > (1 << 24)) return GSTVD_E_SHAPE;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst_bf16 & 7)) return GSTVD_E_ALIGN;
  hipLaunchKernelGGL(cast_ranges_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst_bf16, ranges_dev, blk0_dev,
                     (int)nranges);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_scale(float* x, const float* factor, int64_t n, gstvd_stream_t stream) {
  if (!x || !factor) return GSTVD_E_NULL;
  if (n <= 0) return GSTVD_E_SHAPE;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, factor, n);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gstvd_rng_advance(uint64_t* rng, gstvd_stream_t stream) {
  if (!rng) return GSTVD_E_NULL;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gstvd_dropout_mask(float* out, int64_t n, float p, uint32_t site, const uint64_t* rng, gstvd_stream_t stream) {
  if (!out || !rng) return GSTVD_E_NULL;
  if (n <= 0) return GSTVD_E_SHAPE;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, p, site, rng);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_adamw(float* param, const float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                           const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2, float eps,
                           const float* step, float grad_scale, int64_t begin, gstvd_stream_t stream) {
  if (!param || !grad || !m || !v || !seg_end || !hp || !step) return GSTVD_E_NULL;
  if (n <= 0 || nseg <= 0 || begin < 0 || begin >= n || (begin % 4)) return GSTVD_E_SHAPE;
  hipLaunchKernelGGL(adamw_kernel<float>, dim3((unsigned)((n - begin + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     m, v, (bf16*)shadow_bf16, n, seg_end, hp, nseg, beta1, beta2, eps, step, grad_scale, begin, (int64_t)0,
                     (const int32_t*)nullptr, (const uint8_t*)nullptr);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_adamw_bf16grad(float* param, const void* grad_bf16, int64_t grad_origin, float* m, float* v, void* shadow_bf16,
                                    int64_t n, const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2,
                                    float eps, const float* step, float grad_scale, int64_t begin, gstvd_stream_t stream) {
  if (!param || !grad_bf16 || !m || !v || !seg_end || !hp || !step) return GSTVD_E_NULL;
  if (n <= 0 || nseg <= 0 || begin < 0 || begin >= n || (begin % 4) || grad_origin > begin || (grad_origin % 4)) return GSTVD_E_SHAPE;
  if ((uintptr_t)grad_bf16 & 7) return GSTVD_E_ALIGN;
  hipLaunchKernelGGL(adamw_kernel<bf16>, dim3((unsigned)((n - begin + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, param,
                     (const bf16*)grad_bf16, m, v, (bf16*)shadow_bf16, n, seg_end, hp, nseg, beta1, beta2, eps, step, grad_scale, begin,
                     grad_origin, (const int32_t*)nullptr, (const uint8_t*)nullptr);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_adamw_blocks(float* param, const float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                                  const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2, float eps,
                                  const float* step, float grad_scale, int64_t begin, const int32_t* block_list_dev, int64_t nblocks,
                                  const uint8_t* seg_skip_dev, gstvd_stream_t stream) {
  if (!param || !grad || !m || !v || !seg_end || !hp || !step || !block_list_dev) return GSTVD_E_NULL;
  if (n <= 0 || nseg <= 0 || begin < 0 || begin >= n || (begin % 4) || nblocks < 0 || nblocks > (n + 1023) / 1024) return GSTVD_E_SHAPE;
  if (nblocks == 0) return 0;
  hipLaunchKernelGGL(adamw_kernel<float>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                     (bf16*)shadow_bf16, n, seg_end, hp, nseg, beta1, beta2, eps, step, grad_scale, begin, (int64_t)0,
                     block_list_dev, seg_skip_dev);
  GSTVD_LAUNCH_CHECK();
  return 0;
}
