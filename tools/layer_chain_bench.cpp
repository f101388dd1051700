// PREMISE CHECK for a two-chain schedule: the forward of L text layers of the encoder (QKV GEMM -> attention ->
// attention-out GEMM + residual -> LayerNorm -> FFN-up GEMM + GELU -> FFN-down GEMM + residual -> LayerNorm; bf16, the step's
// shapes: 16 rows x 256 tokens, H = 768, 12 heads) issued from a tight C++ loop on REAL streams (no Python, no hipGraph):
//   mode 0   ONE chain over all 16 rows                      (what the step does)
//   mode 1   TWO chains of 8 rows each on two streams        (the half-batch schedule: one half's attention / LayerNorm / launch gaps
//                                                             under the other half's GEMMs)
//   mode 2   the same two half chains one after the other on one stream   (control: what splitting alone costs)
// hipGraph cannot express mode 1 on this ROCm (its executor runs a forked branch only where the launch stream's branch waits for
// it -- round 3); a launch path of our own could.  This answers whether it would pay.
//   hipcc -O2 --offload-arch=gfx950 tools/layer_chain_bench.cpp -o build/layer_chain_bench -Lgst_visdial_amd/lib -lgstvd_hip -Wl,-rpath,$PWD/gst_visdial_amd/lib
//   build/layer_chain_bench [layers = 6] [dropout p = 0.1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "../include/gstvd_hip.h"

static const int64_t T = 256, H = 768, NH = 12, I = 3072;

struct Weights { void *qkv, *ao, *fi, *fo; float *bq, *ba, *bi, *bo, *g1, *b1, *g2, *b2; };
struct Chain {                       // activations of one chain (B rows)
  int64_t B, M;
  void *x, *qkv, *ctx, *ao, *y1, *a, *aux, *fo;
  float *lse, *mean, *rstd, *mask;
};

static void* dmalloc(size_t n) { void* p; if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); } hipMemset(p, 0, n); return p; }
static void fill_bf16(void* p, size_t n, float scale, unsigned seed) {
  std::vector<unsigned short> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    float v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale;
    unsigned u; memcpy(&u, &v, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice);
}
static void fill_f32(float* p, size_t n, float v) { std::vector<float> h(n, v); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); }

static gstvd_gemm_t gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, const float* bias, const void* add, void* aux, int epi) {
  gstvd_gemm_t g; memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.addend = add; g.aux = aux; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = N; g.ldadd = N; g.ldaux = N; g.batch = 1;
  g.dtype_in = GSTVD_BF16; g.dtype_out = GSTVD_BF16; g.alpha = 1.f;
  g.epilogue = epi | (bias ? GSTVD_EPI_BIAS : 0) | (add ? GSTVD_EPI_ADD : 0);
  return g;
}

static int layer(const Weights& w, const Chain& c, const uint64_t* rng, float p, hipStream_t s) {
  int rc = 0;
  gstvd_gemm_t g1 = gemm(c.x, w.qkv, c.qkv, c.M, 3 * H, H, w.bq, nullptr, nullptr, 0);
  rc |= gstvd_gemm(&g1, s);
  gstvd_attn_t a; memset(&a, 0, sizeof(a));
  a.Q = c.qkv; a.K = (char*)c.qkv + H * 2; a.V = (char*)c.qkv + 2 * H * 2; a.O = c.ctx; a.LSE = c.lse; a.key_mask = c.mask;
  a.ldq = a.ldk = a.ldv = 3 * H; a.ldo = H; a.B = (int)c.B; a.nh = NH; a.Lq = a.Lk = (int)T; a.d = 64; a.dtype = GSTVD_BF16;
  a.mask_neg = -10000.f; a.scale = 0.125f; a.dropout_p = p; a.site = 3; a.rng = rng;
  rc |= gstvd_attn_fwd(&a, s);
  gstvd_gemm_t g2 = gemm(c.ctx, w.ao, c.ao, c.M, H, H, w.ba, nullptr, nullptr, 0);
  rc |= gstvd_gemm(&g2, s);
  gstvd_ln_t l; memset(&l, 0, sizeof(l));
  l.mode = GSTVD_LN_RESID; l.dtype = GSTVD_BF16; l.M = c.M; l.H = H; l.x = c.ao; l.ldx = H; l.res = c.x; l.ldres = H;
  l.gamma = w.g1; l.beta = w.b1; l.eps = 1e-12f; l.y = c.y1; l.ldy = H; l.mean = c.mean; l.rstd = c.rstd; l.p_pre = p; l.site_pre = 4; l.rng = rng;
  rc |= gstvd_ln_fwd(&l, s);
  gstvd_gemm_t g3 = gemm(c.y1, w.fi, c.a, c.M, I, H, w.bi, nullptr, c.aux, GSTVD_EPI_GELU);
  rc |= gstvd_gemm(&g3, s);
  gstvd_gemm_t g4 = gemm(c.a, w.fo, c.fo, c.M, H, I, w.bo, nullptr, nullptr, 0);
  rc |= gstvd_gemm(&g4, s);
  l.x = c.fo; l.res = c.y1; l.gamma = w.g2; l.beta = w.b2; l.y = c.x; l.site_pre = 5;         // the layer's output feeds the next layer
  rc |= gstvd_ln_fwd(&l, s);
  return rc;
}

static Chain make_chain(int64_t B, unsigned seed) {
  Chain c; c.B = B; c.M = B * T;
  c.x = dmalloc(c.M * H * 2); c.qkv = dmalloc(c.M * 3 * H * 2); c.ctx = dmalloc(c.M * H * 2); c.ao = dmalloc(c.M * H * 2);
  c.y1 = dmalloc(c.M * H * 2); c.a = dmalloc(c.M * I * 2); c.aux = dmalloc(c.M * I * 2); c.fo = dmalloc(c.M * H * 2);
  c.lse = (float*)dmalloc(B * NH * T * 4); c.mean = (float*)dmalloc(c.M * 4); c.rstd = (float*)dmalloc(c.M * 4); c.mask = (float*)dmalloc(B * T * 4);
  fill_bf16(c.x, c.M * H, 1.f, seed); fill_f32(c.mask, B * T, 1.f);
  return c;
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 6;
  const float p = argc > 2 ? (float)atof(argv[2]) : 0.1f;
  std::vector<Weights> w(L);
  for (int i = 0; i < L; ++i) {
    w[i].qkv = dmalloc(3 * H * H * 2); w[i].ao = dmalloc(H * H * 2); w[i].fi = dmalloc(I * H * 2); w[i].fo = dmalloc(H * I * 2);
    fill_bf16(w[i].qkv, 3 * H * H, 0.05f, 10 + i); fill_bf16(w[i].ao, H * H, 0.05f, 20 + i); fill_bf16(w[i].fi, I * H, 0.05f, 30 + i); fill_bf16(w[i].fo, H * I, 0.03f, 40 + i);
    w[i].bq = (float*)dmalloc(3 * H * 4); w[i].ba = (float*)dmalloc(H * 4); w[i].bi = (float*)dmalloc(I * 4); w[i].bo = (float*)dmalloc(H * 4);
    w[i].g1 = (float*)dmalloc(H * 4); w[i].b1 = (float*)dmalloc(H * 4); w[i].g2 = (float*)dmalloc(H * 4); w[i].b2 = (float*)dmalloc(H * 4);
    fill_f32(w[i].g1, H, 1.f); fill_f32(w[i].g2, H, 1.f);
  }
  uint64_t* rng = (uint64_t*)dmalloc(16);
  uint64_t hrng[2] = {1234, 0}; hipMemcpy(rng, hrng, 16, hipMemcpyHostToDevice);
  Chain full = make_chain(16, 1), h0 = make_chain(8, 2), h1 = make_chain(8, 3);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1, ej; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  auto run = [&](int mode) {
    float best = 1e9f, host = 0.f; int rc = 0;
    for (int rep = 0; rep < 6; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::high_resolution_clock::now();
      hipEventRecord(e0, s0);
      if (mode == 1) { hipEventRecord(ej, s0); hipStreamWaitEvent(s1, ej, 0); }
      for (int i = 0; i < L; ++i) {
        if (mode == 0) rc |= layer(w[i], full, rng, p, s0);
        else { rc |= layer(w[i], h0, rng, p, s0); rc |= layer(w[i], h1, rng, p, mode == 1 ? s1 : s0); }
      }
      if (mode == 1) { hipEventRecord(ej, s1); hipStreamWaitEvent(s0, ej, 0); }
      hipEventRecord(e1, s0);
      auto t1 = std::chrono::high_resolution_clock::now();
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && ms < best) { best = ms; host = (float)std::chrono::duration<double, std::micro>(t1 - t0).count(); }
    }
    printf("  mode %d: %8.1f us for %d layers = %6.1f us per layer   (host issue %.0f us, rc %d)\n", mode, best * 1e3, L, best * 1e3 / L, host, rc);
    return best;
  };
  printf("text-layer forward chain, %d layers, 16 rows x 256 tokens, dropout %.2f\n", L, p);
  const float t0 = run(0), t1 = run(1), t2 = run(2);
  printf("two half-batch chains on two streams vs one full-batch chain: %.3f x   (split alone, one stream: %.3f x)\n", t1 / t0, t2 / t0);
  return 0;
}
