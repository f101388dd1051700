// One sampling step of the decode loop on the device (models/visual_dialog_model.py:96-108, utils/decoding_utils.py:4-35):
//   z = logits / temperature  (banned tokens of the n-gram filter: -inf)
//   top-k: z < (k-th largest z) -> -inf   (ties with the k-th value stay, like the reference's `logits < topk(...)[..., -1]`)
//   top-p (utils/decoding_utils.py:22-34): a token stays when the probability mass of the tokens sorted in FRONT of it is <= top_p
//          (the reference removes `cumsum(softmax(sorted)) > top_p` shifted right by one, so the token that crosses top_p stays)
//   p = softmax(z);  id = first index whose cumulative probability reaches u * sum(p)   (inverse CDF, decoding.draw_from_uniform)
// One workgroup per dialog row, the row's V scaled logits live in LDS (V = 30522: 119 KB of the CU's 160 KB).  It replaces
// ~25 small library kernels per step (topk's multi-block radix passes, softmax, cumsum scans, compares, copies) by one launch,
// and keeps library kernels with their own temporary storage / memset nodes out of the captured token graph.
#include "common.h"
#include <math.h>

namespace {

constexpr int NT = 1024, NWV = NT / 64;
// up to here top-k walks the distinct values from the top (one block reduction per value: 33 us at k = 7, 120 us at k = 64 for
// 16 x 30522 bf16), beyond it bisects on the value (52 us whatever k; profiles/r05_sample_top_p.txt)
constexpr int TOPK_ITERATIVE_MAX = 16;

// block reduction of (max value, how many elements carry it); counts travel as floats (exact below 2^24)
DEVFN void reduce_maxcount(float& m, int& c, float* smf, float* smc, int tid) {
  const float wm = wave_max_fast(m);
  const float wc = wave_sum(m == wm ? (float)c : 0.f);
  const int w = tid >> 6;
  __syncthreads();
  if ((tid & 63) == 0) { smf[w] = wm; smc[w] = wc; }
  __syncthreads();
  float bm = -INFINITY, bc = 0.f;
#pragma unroll
  for (int i = 0; i < NWV; i += 4) {
    const f32x4 vm = *(const f32x4*)(smf + i), vc = *(const f32x4*)(smc + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (vm[j] > bm) { bm = vm[j]; bc = vc[j]; }
      else if (vm[j] == bm) bc += vc[j];
    }
  }
  m = bm; c = (int)bc;
}

// order-preserving integer key of a float (larger value <-> larger key; -inf is the smallest key of a non-NaN value)
DEVFN uint32_t fkey(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
DEVFN float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// block sum of one float per thread, the same value in every thread
DEVFN float block_sum(float x, float* smf, int tid) {
  const float w = wave_sum(x);
  __syncthreads();
  if ((tid & 63) == 0) smf[tid >> 6] = w;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NWV; ++i) t += smf[i];
  return t;
}

template <typename T>
__global__ __launch_bounds__(NT) void sample_topk_kernel(gstvd_sample_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* z = (float*)smem;                                   // [V]
  __shared__ __attribute__((aligned(16))) float smf[NWV];
  __shared__ __attribute__((aligned(16))) float smc[NWV];
  __shared__ int smi[NWV];
  __shared__ float swave[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, V = a.V;
  const T* row = (const T*)a.logits + (int64_t)b * a.ld;
  const uint8_t* ban = a.banned ? a.banned + (int64_t)b * a.banned_ld : nullptr;

  // the row: element i = tid + j * NT lives in register zr[j] of thread tid (for the top-k passes) and in z[i] (LDS, for the
  // ordered CDF pass below); all of a thread's loads are issued before the first use
  constexpr int SEG = 31;                                     // V <= 31 * 1024 (checked by the host entry)
  float zr[SEG];
  uint8_t bn[SEG];
#pragma unroll
  for (int j = 0; j < SEG; ++j) {
    const int i = tid + j * NT;
    zr[j] = i < V ? to_f(row[i]) : 0.f;
    bn[j] = (ban && i < V) ? ban[i] : (uint8_t)0;
  }
  float m = -INFINITY;
  int c = 0;
#pragma unroll
  for (int j = 0; j < SEG; ++j) {
    const int i = tid + j * NT;
    float v = -INFINITY;
    if (i < V) {
      v = zr[j] / a.temperature;                             // (a true division, like the reference's logits / temperature)
      if (bn[j]) v = -INFINITY;
      z[i] = v;
      if (v > m) { m = v; c = 1; }
      else if (v == m) ++c;
    }
    zr[j] = v;                                               // (-inf past the end: never selected, never counted below)
  }
  if (a.ngram > 0 && a.hist && a.ids_tm && a.cur_len >= a.ngram - 1 && a.hist_T >= a.ngram) {
    // The n-gram filter (utils/decoding_utils.py:38-77) in this launch: one window of the row's history per thread.  A window
    // bans its last token when it holds no special id and its first n-1 ids are the row's last n-1 generated ids.  The ban
    // goes into the LDS copy of the row; the register copies (which the top-k passes read) are refreshed from it afterwards.
    // (The torch-op form -- unfold, compares, a [B, V + 1] scatter -- was ten launches per token inside the captured loop.)
    const int n = a.ngram;
    const int64_t* hrow = a.hist + (int64_t)b * a.hist_ld;
    __syncthreads();                                         // every z[i] of the row is written
    for (int s0 = tid; s0 + n <= a.hist_T; s0 += NT) {
      bool hit = true;
      int64_t last = 0;
      for (int j = 0; j < n; ++j) {
        const int64_t t = hrow[s0 + j];
        for (int q = 0; q < a.n_special; ++q) hit = hit && (t != (int64_t)a.special[q]);
        if (j < n - 1) hit = hit && (t == a.ids_tm[(int64_t)(a.cur_len - (n - 1) + j) * a.ids_stride + b]);
        last = t;
      }
      if (hit && last >= 0 && last < V) z[last] = -INFINITY;    // (several windows may ban one token: same value, benign)
    }
    __syncthreads();
    m = -INFINITY; c = 0;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const int i = tid + j * NT;
      if (i < V) {
        const float v = z[i];
        zr[j] = v;
        if (v > m) { m = v; c = 1; }
        else if (v == m) ++c;
      }
    }
  }
  reduce_maxcount(m, c, smf, smc, tid);                      // (also orders the z[] writes before the reads below)
  const float zmax = m;
  float kth = -INFINITY;
  if (a.top_k > TOPK_ITERATIVE_MAX && a.top_k < V && c < a.top_k) {
    // large k: bisection over the float keys for the k-th largest value = the largest t with #{z >= t} >= k (at most 32 counts
    // of the row instead of up to k rounds of "next smaller distinct value"); fewer than k finite values -> -inf (nothing goes)
    uint32_t lo = fkey(-INFINITY), hi = fkey(zmax);            // #{z >= val(lo)} = V >= k;  #{z >= val(hi)} = c < k
    while (hi - lo > 1u) {                                     // uniform: every thread holds the same (lo, hi)
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const float t = fkey_inv(mid);
      float n = 0.f;
#pragma unroll
      for (int j = 0; j < SEG; ++j) n += (tid + j * NT < V && zr[j] >= t) ? 1.f : 0.f;
      n = block_sum(n, smf, tid);                              // (exact: counts stay below 2^24)
      if (n >= (float)a.top_k) lo = mid; else hi = mid;
    }
    kth = fkey_inv(lo);
  } else if (a.top_k > 0 && a.top_k < V) {                     // (top_k >= V filters nothing -- the reference clamps k to V and
    const int k = a.top_k;                                   //  keeps everything >= the row's minimum: kth stays -inf.  Walking
                                                             //  V distinct values, one block reduction each, did the same slowly)
    int have = c;
    float thr = m;
    while (have < k) {                                       // uniform: every thread holds the same (thr, have)
      float m2 = -INFINITY;
      int c2 = 0;
#pragma unroll
      for (int j = 0; j < SEG; ++j) {                      // branch-free: selects only (an all -inf remainder may count garbage
        const float v = zr[j] < thr ? zr[j] : -INFINITY;     // into c2 -- it is discarded below when m2 comes out as -inf)
        const bool gt = v > m2, eq = v == m2;
        c2 = gt ? 1 : (eq ? c2 + 1 : c2);
        m2 = gt ? v : m2;
      }
      // (entries that are -inf -- banned or past the end -- count as "nothing left": c2 of an all -inf remainder is dropped)
      reduce_maxcount(m2, c2, smf, smc, tid);
      if (c2 == 0 || m2 == -INFINITY) { thr = -INFINITY; break; }
      thr = m2;
      have += c2;
    }
    kth = thr;
  }
  if (a.top_p > 0.f && a.top_p < 1.f) {
    // top-p on what top-k left: with w_i = [z_i >= kth] exp(z_i - zmax) and S = sum w, token i stays iff the mass of the strictly
    // larger logits G(z_i) = sum_{z_j > z_i} w_j is <= top_p S.  G is a falling step function of the threshold, so the kept set
    // is {z >= t*} with t* the smallest value whose G is <= top_p S: bisection over the float keys, one masked sum of the row per
    // step (<= 32).  Equal logits stay or go together (the reference's sort leaves the order inside a tie to the sort).
    float sw = 0.f;
#pragma unroll
    for (int j = 0; j < SEG; ++j) sw += (zr[j] >= kth && zr[j] > -INFINITY) ? __expf(zr[j] - zmax) : 0.f;
    const float target = a.top_p * block_sum(sw, smf, tid);
    uint32_t lo = fkey(kth) - 1u, hi = fkey(zmax);             // G(val(hi)) = 0 <= target; below kth nothing is left to drop
    if (!(kth > -INFINITY)) lo = fkey(-INFINITY);
    while (hi - lo > 1u) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const float t = fkey_inv(mid);
      float g = 0.f;
#pragma unroll
      for (int j = 0; j < SEG; ++j) g += (zr[j] >= kth && zr[j] > t) ? __expf(zr[j] - zmax) : 0.f;
      g = block_sum(g, smf, tid);
      if (g <= target) hi = mid; else lo = mid;
    }
    const float pth = fkey_inv(hi);
    if (pth > kth) kth = pth;
  }
  // inverse CDF over e_i = [z_i >= kth] * exp(z_i - zmax): thread t owns the contiguous segment [t * seg, (t + 1) * seg)
  // (seg odd: the threads' LDS reads fall into different banks); the segment's weights stay in registers for the second pass
  constexpr int SEGMAX = SEG;
  const int seg = ((V + NT - 1) / NT) | 1;
  const int i0 = tid * seg, i1 = (i0 + seg < V) ? i0 + seg : V;
  float e[SEGMAX];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < SEGMAX; ++j) {
    const int i = i0 + j;
    float w = 0.f;
    if (j < seg && i < i1) {
      const float v = z[i];
      w = (v >= kth && v > -INFINITY) ? __expf(v - zmax) : 0.f;
    }
    e[j] = w;
    s += w;
  }
  // exclusive prefix of s over the 1024 threads: wave scan, then the 16 wave totals
  float inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) swave[wave] = inc;
  __syncthreads();
  float pre = inc - s, total = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    const float t = swave[w];
    if (w < wave) pre += t;
    total += t;
  }
  const float x = a.u[b] * total;
  int cnt = 0;
  float run = pre;
#pragma unroll
  for (int j = 0; j < SEGMAX; ++j) {
    run += e[j];
    cnt += (j < seg && i0 + j < i1 && run < x) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  __syncthreads();
  if (lane == 0) smi[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) idx += smi[i];
    if (idx > V - 1) idx = V - 1;
    a.out[(int64_t)b * a.out_stride] = idx;
  }
}

template <typename T> int launch(const gstvd_sample_t& a, hipStream_t s) {
  const int lds = (int)(((int64_t)a.V * 4 + 15) & ~15ll);
  static int attr_done = 0, attr_rc = 0;
  if (lds > 48 * 1024 && lds > attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)sample_topk_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_rc = e == hipSuccess ? 0 : (int)e;
    attr_done = lds;
  }
  if (attr_rc) return attr_rc;
  hipLaunchKernelGGL((sample_topk_kernel<T>), dim3((unsigned)a.B), dim3(NT), lds, s, a);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int gstvd_sample_topk(const gstvd_sample_t* a, gstvd_stream_t stream) {
  if (!a || !a->logits || !a->u || !a->out) return GSTVD_E_NULL;
  if (a->dtype != GSTVD_F32 && a->dtype != GSTVD_BF16) return GSTVD_E_DTYPE;
  if (a->B <= 0 || a->V <= 0 || a->ld < a->V || a->top_k < 0 || !(a->temperature > 0.f) || !(a->top_p >= 0.f)) return GSTVD_E_SHAPE;
  if (a->V > 31 * 1024) return GSTVD_E_UNSUPPORTED;                         // the row must fit the CU's LDS (and 31 weights per thread)
  if (a->ngram > 0 && (!a->hist || !a->ids_tm || a->hist_T < 0 || a->cur_len < 0 || a->n_special < 0 || a->n_special > 8 || a->ids_stride < a->B))
    return a->hist && a->ids_tm ? GSTVD_E_SHAPE : GSTVD_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == GSTVD_BF16 ? launch<bf16>(*a, s) : launch<float>(*a, s);
}
