// Shared device helpers for the gfx950 kernels (wave64, MFMA fragments, counter-based dropout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gstvd_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define WAVE 64
#define DEVFN __device__ __forceinline__

DEVFN float to_f(float x) { return x; }
DEVFN float to_f(bf16 x) { return (float)x; }
template <typename T> DEVFN T from_f(float x);
template <> DEVFN float from_f<float>(float x) { return x; }
template <> DEVFN bf16 from_f<bf16>(float x) { return (bf16)x; }

// ---- 4-element vector load/store of T as fp32 -------------------------------------------------
DEVFN f32x4 ld4(const float* p) { return *(const f32x4*)p; }
DEVFN f32x4 ld4(const bf16* p) {
  bf16x4 v = *(const bf16x4*)p;
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
// streaming (non-temporal) variants for data that is read exactly once
DEVFN f32x4 ld4_stream(const float* p) { return __builtin_nontemporal_load((const f32x4*)p); }
DEVFN f32x4 ld4_stream(const bf16* p) {
  bf16x4 v = __builtin_nontemporal_load((const bf16x4*)p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
DEVFN void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
DEVFN void st4(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = r;
}

// ---- AdamW, one 4-vector: pytorch_transformers==1.2.0 optimization.AdamW.step (train_gen.py:16,247) -----------------------
// Shared by the stand-alone pass (loss.hip) and the weight-gradient launch that updates in its epilogue (gemm_dma256.hip): the
// two must agree bit for bit, so contraction is OFF inside and every fused multiply-add is spelled out, and the update term
// m / (sqrt(v) + eps) uses the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 in both (the IEEE sequences cost ~60 instructions per
// element: invisible in the HBM-bound pass, ~30 us per tile in a GEMM epilogue with two waves per SIMD); the term is scaled by
// lr before it meets the parameter, so its last ulp is ~1e-12 of the weight.
// g = stored gradient (alpha * acc), gscale = 1/world; bc = sqrt(1 - b2^t) / (1 - b1^t); decay applied after the update.
DEVFN void adamw_update4(f32x4& p4, f32x4& m4, f32x4& v4, f32x4 g4, float gscale, float lr, float wd, float bc, float b1, float b2,
                         float eps) {
#pragma clang fp contract(off)
  const float step_size = lr * bc, decay = -lr * wd, c1 = 1.f - b1, c2 = 1.f - b2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float g = g4[e] * gscale;
    const float mm = __builtin_fmaf(m4[e], b1, g * c1);
    const float vv = __builtin_fmaf(v4[e], b2, (g * g) * c2);
    const float den = __builtin_amdgcn_sqrtf(vv) + eps;
    float pp = __builtin_fmaf(-step_size, mm * __builtin_amdgcn_rcpf(den), p4[e]);
    if (wd > 0.f) pp = __builtin_fmaf(decay, pp, pp);
    m4[e] = mm; v4[e] = vv; p4[e] = pp;
  }
}
DEVFN float adamw_bias_correction(float b1, float b2, float t) { return sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t)); }
// first lr/wd segment whose (exclusive) end lies beyond flat index i: uniform arguments -> scalar loads
DEVFN int64_t adamw_segment(const int64_t* seg_end, int64_t nseg, int64_t i) {
  int64_t lo = 0, hi = nseg - 1;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (seg_end[mid] > i) hi = mid; else lo = mid + 1; }
  return lo;
}

// ---- wave reductions (64 lanes) ----------------------------------------------------------------
// Sum over the 64 lanes, result in every lane.  DPP moves inside the VALU (quad permutes, row mirrors, row broadcasts) instead
// of six dependent ds_bpermute round trips through the LDS crossbar (~100 cycles each): the reductions sit on the critical
// path of every LayerNorm wave.
DEVFN float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror: every lane = its row's sum
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: two gfx950 lane swaps
// inside the VALU instead of two dependent ds_bpermute round trips (~100 cycles each) through the LDS crossbar.
//   v_permlane16_swap a, b: rows 1, 3 of a <-> rows 0, 2 of b;   v_permlane32_swap a, b: rows 2, 3 of a <-> rows 0, 1 of b.
DEVFN void rows_swap16(float& x, float& y) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
DEVFN void rows_swap32(float& x, float& y) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
DEVFN float rows_max(float v) {
  float w = v;
  rows_swap16(v, w);
  v = fmaxf(v, w); w = v;
  rows_swap32(v, w);
  return fmaxf(v, w);
}
DEVFN float rows_sum(float v) {
  float w = v;
  rows_swap16(v, w);
  v += w; w = v;
  rows_swap32(v, w);
  return v + w;
}

// Max over the 64 lanes, result in every lane: DPP inside the 16-lane rows, lane swaps across them (no LDS crossbar trips)
DEVFN float wave_max_fast(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));  // row_half_mirror
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));  // row_mirror
  return rows_max(v);
}
DEVFN float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based dropout ----------------------------------------------------------------------
// lowbias32 integer hash (full avalanche); one 32-bit draw serves two elements (16 bits each).
// Counter hash of the dropout draws: xorshift folds around two 24-bit multiplies (v_mul_u32_u24: full rate; a 32-bit
// v_mul_lo_u32 issues at quarter rate and the attention kernels, which draw once per two score elements, are VALU bound).
// The fold in front of each multiply brings bits 16..31 down into the 24 bits the multiply reads.  Avalanche (every output
// bit flips with probability 0.497-0.503 for every input bit) and the keep statistics of sequential counters match the
// 32-bit-multiply mixer it replaced (round 3; numbers in DESIGN.md section 3, "Activations").
DEVFN uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x = __umul24(x, 0xeb352dU); x ^= x >> 15; x = __umul24(x, 0x6ca68bU); x ^= x >> 16;
  return x;
}
// Draw for 32-bit counter c under `key`.  mix32's first multiply reads 24 bits: after the xor-fold the counter's TOP BYTE meets
// them only linearly (in bits 8..15), so c and c ^ 0x01000100 gave the same draw for every key -- exact duplicates of the mask
// at a fixed distance as soon as a site has more than 2^24 draws (2^25 elements: attention probabilities at >= 43 rows x 12
// heads x 256 x 256).  The top byte is therefore hashed into the key first (8 bits x a 24-bit odd constant, a full-rate
// v_mul_u32_u24; it is zero -- and the stream unchanged -- for the first 2^24 counters of a site): counters that differ in bits
// 24..31 now collide only like any two unrelated counters (tests/test_host_logic.py restates this hash in numpy and checks
// both; tests/test_ops_gpu.py checks the device masks of a 2^26-element site).
DEVFN uint32_t drop_hash(uint32_t c, uint32_t key) {
  return mix32(c ^ key ^ __umul24(c >> 24, 0x9E3779u));
}
struct DropKey {
  uint32_t key, thr;   // keep iff u16 >= thr
  float scale;         // 1/(1-p)
  bool on;
};
DEVFN DropKey make_drop(float p, uint32_t site, const uint64_t* rng) {
  DropKey k;
  k.on = (p > 0.f) && (rng != nullptr);
  k.thr = 0; k.scale = 1.f; k.key = 0;
  if (k.on) {
    uint64_t seed = rng[0], off = rng[1];
    uint32_t a = mix32((uint32_t)seed ^ 0x9E3779B9u * (site + 1u));
    uint32_t b = mix32((uint32_t)(seed >> 32) + 0x85EBCA6Bu * (uint32_t)off + site);
    uint32_t c = mix32((uint32_t)(off >> 32) ^ (a + b));
    k.key = a ^ (b * 0xC2B2AE35u) ^ c;
    float t = p * 65536.f + 0.5f;
    k.thr = (uint32_t)t;
    k.scale = 1.f / (1.f - p);
  }
  return k;
}
DEVFN uint32_t drop_draw(const DropKey& k, uint64_t e2) {   // e2 = element index >> 1
  uint32_t lo = (uint32_t)e2, hi = (uint32_t)(e2 >> 32);
  return mix32((lo ^ k.key ^ __umul24(lo >> 24, 0x9E3779u)) + hi * 0x9E3779B1u);
}
DEVFN float drop_factor(const DropKey& k, uint64_t e) {     // scale if kept, 0 if dropped
  if (!k.on) return 1.f;
  uint32_t r = drop_draw(k, e >> 1);
  uint32_t u = (e & 1) ? (r >> 16) : (r & 0xffffu);
  return u >= k.thr ? k.scale : 0.f;
}
// four consecutive elements starting at e (e % 4 == 0)
DEVFN f32x4 drop_factor4(const DropKey& k, uint64_t e) {
  f32x4 f = {1.f, 1.f, 1.f, 1.f};
  if (!k.on) return f;
  uint32_t r0 = drop_draw(k, e >> 1), r1 = drop_draw(k, (e >> 1) + 1);
  f[0] = (r0 & 0xffffu) >= k.thr ? k.scale : 0.f;
  f[1] = (r0 >> 16) >= k.thr ? k.scale : 0.f;
  f[2] = (r1 & 0xffffu) >= k.thr ? k.scale : 0.f;
  f[3] = (r1 >> 16) >= k.thr ? k.scale : 0.f;
  return f;
}

// ---- erf GELU (vilbert_dialog.py:115-121): value and derivative from one exp ---------------------
// gelu(x) = x*Phi(x), gelu'(x) = Phi(x) + x*phi(x).  FAST (bf16 operands) evaluates erf with Abramowitz-Stegun 7.1.26
// (|err| <= 1.5e-7, far below a bf16 ulp) sharing exp(-x^2/2) with phi; the fp32 path keeps libm's erff.
template <bool FAST> DEVFN void gelu_both(float x, float& g, float& dg) {
  const float e = __expf(-0.5f * x * x);
  float erfv;
  if (FAST) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    erfv = copysignf(1.0f - poly * e, x);
  } else {
    erfv = erff(x * 0.70710678118654752440f);
  }
  const float cdf = 0.5f + 0.5f * erfv;
  g = x * cdf;
  dg = cdf + x * (0.39894228040143267794f * e);
}

// ---- MFMA wrappers: D(16x16) += A(16xK) * B(Kx16), fp32 accumulate ----------------------------
// lane l: A[row l&15][k-slice l>>4], B[k-slice l>>4][col l&15]; D[row (l>>4)*4 + r][col l&15].
DEVFN f32x4 mfma_bf16_k32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DEVFN f32x4 mfma_bf16_k16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
DEVFN f32x4 mfma_f32_k4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS transposed read: per 16-lane group a 4(row) x 16(col) block of 16-bit elements; lane 4q+p supplies
// the address of row q, cols 4p..4p+3; lane i receives column i of the four rows.
DEVFN s16x4 lds_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
DEVFN s16x4 pack_bf16x4(float a, float b, float c, float d) {
  bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  return __builtin_bit_cast(s16x4, v);
}

#define GSTVD_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
