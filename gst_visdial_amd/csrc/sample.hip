// One sampling step of the decode loop on the device (models/visual_dialog_model.py:96-108, utils/decoding_utils.py:4-35):
//   z = logits / temperature  (banned tokens of the n-gram filter: -inf)
//   top-k: z < (k-th largest z) -> -inf   (ties with the k-th value stay, like the reference's `logits < topk(...)[..., -1]`)
//   p = softmax(z);  id = first index whose cumulative probability reaches u * sum(p)   (inverse CDF, decoding.draw_from_uniform)
// One workgroup per dialog row, the row's V scaled logits live in LDS (V = 30522: 119 KB of the CU's 160 KB).  It replaces
// ~25 small library kernels per step (topk's multi-block radix passes, softmax, cumsum scans, compares, copies) by one launch,
// and keeps library kernels with their own temporary storage / memset nodes out of the captured token graph.
#include "common.h"
#include <math.h>

namespace {

constexpr int NT = 256;

// block reduction of (max value, how many elements carry it)
DEVFN void reduce_maxcount(float& m, int& c, float* smf, int* smi, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oc = __shfl_xor(c, o, 64);
    if (om > m) { m = om; c = oc; }
    else if (om == m) c += oc;
  }
  const int w = tid >> 6;
  __syncthreads();
  if ((tid & 63) == 0) { smf[w] = m; smi[w] = c; }
  __syncthreads();
  m = smf[0]; c = smi[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) {
    if (smf[i] > m) { m = smf[i]; c = smi[i]; }
    else if (smf[i] == m) c += smi[i];
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void sample_topk_kernel(gstvd_sample_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* z = (float*)smem;                                   // [V]
  __shared__ float smf[NT / 64];
  __shared__ int smi[NT / 64];
  __shared__ float spart[NT];
  __shared__ int scnt[NT / 64];
  const int tid = threadIdx.x, b = blockIdx.x, V = a.V;
  const T* row = (const T*)a.logits + (int64_t)b * a.ld;
  const uint8_t* ban = a.banned ? a.banned + (int64_t)b * a.banned_ld : nullptr;

  float m = -INFINITY;
  int c = 0;
  for (int i = tid; i < V; i += NT) {
    float v = to_f(row[i]) / a.temperature;
    if (ban && ban[i]) v = -INFINITY;
    z[i] = v;
    if (v > m) { m = v; c = 1; }
    else if (v == m) ++c;
  }
  reduce_maxcount(m, c, smf, smi, tid);                      // (also orders the z[] writes before the reads below)
  const float zmax = m;
  float kth = -INFINITY;
  if (a.top_k > 0) {
    const int k = a.top_k < V ? a.top_k : V;
    int have = c;
    float thr = m;
    while (have < k) {                                       // uniform: every thread holds the same (thr, have)
      float m2 = -INFINITY;
      int c2 = 0;
      for (int i = tid; i < V; i += NT) {
        const float v = z[i];
        if (v < thr) {
          if (v > m2) { m2 = v; c2 = 1; }
          else if (v == m2) ++c2;
        }
      }
      reduce_maxcount(m2, c2, smf, smi, tid);
      if (c2 == 0) { thr = -INFINITY; break; }               // nothing below thr (only possible when -inf entries are all that is left)
      thr = m2;
      have += c2;
    }
    kth = thr;
  }
  // inverse CDF over e_i = [z_i >= kth] * exp(z_i - zmax): thread t owns the contiguous segment [t * seg, (t + 1) * seg)
  const int seg = (V + NT - 1) / NT;
  const int i0 = tid * seg, i1 = (i0 + seg < V) ? i0 + seg : V;
  float s = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float v = z[i];
    s += (v >= kth && v > -INFINITY) ? expf(v - zmax) : 0.f;
  }
  spart[tid] = s;
  __syncthreads();
  float pre = 0.f, total = 0.f;
  for (int t = 0; t < NT; ++t) {                             // 256 LDS broadcasts: negligible next to the row passes
    const float v = spart[t];
    if (t < tid) pre += v;
    total += v;
  }
  const float x = a.u[b] * total;
  int cnt = 0;
  float run = pre;
  for (int i = i0; i < i1; ++i) {
    const float v = z[i];
    run += (v >= kth && v > -INFINITY) ? expf(v - zmax) : 0.f;
    cnt += run < x ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) scnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) idx += scnt[i];
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
  if (a->B <= 0 || a->V <= 0 || a->ld < a->V || a->top_k < 0 || !(a->temperature > 0.f)) return GSTVD_E_SHAPE;
  if ((int64_t)a->V * 4 > 150 * 1024) return GSTVD_E_UNSUPPORTED;          // the row must fit the CU's LDS
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == GSTVD_BF16 ? launch<bf16>(*a, s) : launch<float>(*a, s);
}
