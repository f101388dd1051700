// Do the step's two tail kernels share the chip or take turns?  A batched weight-gradient GEMM (dW = dy^T x: 3072 x 768 x 4096,
// k-major operands, fp32 out, `nb` problems in one launch -- ingest/MFMA-bound) and the fused AdamW over `na` Mi parameters
// (HBM-bound), each alone and then together on two streams, in both issue orders.  If "together" ~ max(alone) the end of the step
// (1.45 ms of weight gradients, then 1.72 ms of AdamW, one after the other) can be cut into pieces that run side by side; if it is
// ~ the sum, it cannot.  (Measured: the sum -- AdamW's one-workgroup-per-block grid takes every wave slot, the GEMM makes no
// progress until it is through.  A resident-grid AdamW, GSTVD_ADAMW_BG=W workgroups per CU, existed for this probe and is gone
// again: profiles/r04_overlap_bench.txt, profiles/r04_bg_sweep.txt.  The variable is only echoed now.)
//   hipcc -O2 --offload-arch=gfx950 tools/overlap_bench.cpp -o build/overlap_bench -Lgst_visdial_amd/lib -lgstvd_hip -Wl,-rpath,$PWD/gst_visdial_amd/lib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/gstvd_hip.h"

static void* dmalloc(size_t n, int byte = 0) { void* p; if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", n); exit(1); } (void)hipMemset(p, byte, n); return p; }

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 24;
  const int64_t na = (int64_t)(argc > 2 ? atoi(argv[2]) : 96) << 20;
  const int64_t M = 3072, N = 768, K = 4096;
  // bf16 0x3c3c = 0.0115: finite operands (the clock under load depends on the data, zeros would flatter the GEMM)
  void* A = dmalloc((size_t)nb * K * M * 2, 0x3c); void* B = dmalloc((size_t)nb * K * N * 2, 0x3c); void* C = dmalloc((size_t)nb * M * N * 4);
  gstvd_gemm_t g; memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = M; g.ldb = N; g.ldc = N; g.batch = nb; g.sA = K * M; g.sB = K * N; g.sC = M * N;
  g.dtype_in = GSTVD_BF16; g.dtype_out = GSTVD_F32; g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = 1.f;
  const int64_t off = 1024;                                    // begin > 0: the launch the pipeline makes for every slice but the last
  float* P = (float*)dmalloc((na + off) * 4); float* G = (float*)dmalloc((na + off) * 4, 0x38); float* Mo = (float*)dmalloc((na + off) * 4);
  float* V = (float*)dmalloc((na + off) * 4); void* S = dmalloc((na + off) * 2);
  int64_t seg_h[1] = {na + off}; float hp_h[2] = {1e-4f, 0.01f}, step_h = 3.f;
  int64_t* seg = (int64_t*)dmalloc(8); float* hp = (float*)dmalloc(8); float* step = (float*)dmalloc(4);
  (void)hipMemcpy(seg, seg_h, 8, hipMemcpyHostToDevice); (void)hipMemcpy(hp, hp_h, 8, hipMemcpyHostToDevice); (void)hipMemcpy(step, &step_h, 4, hipMemcpyHostToDevice);
  hipStream_t s0, s1; (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
  hipEvent_t e0, e1, ea, eb, ej; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&ea); (void)hipEventCreate(&eb); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  auto gemm = [&](hipStream_t s) { return gstvd_gemm(&g, s); };
  auto adam = [&](hipStream_t s) { return gstvd_adamw(P, G, Mo, V, S, na + off, seg, hp, 1, 0.9f, 0.999f, 1e-8f, step, 1.f, off, s); };
  auto run = [&](const char* name, int mode) {
    float best = 1e9f, bg = 0, ba = 0; int rc = 0;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, s0);
      if (mode == 0) rc |= gemm(s0);
      else if (mode == 1) rc |= adam(s0);
      else if (mode == 2) { rc |= gemm(s0); rc |= adam(s0); }
      else {
        (void)hipEventRecord(ej, s0); (void)hipStreamWaitEvent(s1, ej, 0);
        if (mode == 3) { rc |= gemm(s0); (void)hipEventRecord(ea, s0); rc |= adam(s1); (void)hipEventRecord(eb, s1); }
        else { rc |= adam(s1); (void)hipEventRecord(eb, s1); rc |= gemm(s0); (void)hipEventRecord(ea, s0); }
        (void)hipEventRecord(ej, s1); (void)hipStreamWaitEvent(s0, ej, 0);
      }
      (void)hipEventRecord(e1, s0);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 1 && ms < best) {
        best = ms;
        if (mode >= 3) { (void)hipEventElapsedTime(&bg, e0, ea); (void)hipEventElapsedTime(&ba, e0, eb); }
      }
    }
    if (mode >= 3) printf("  %-44s %8.1f us   (GEMM done at %.1f, AdamW done at %.1f; rc %d)\n", name, best * 1e3, bg * 1e3, ba * 1e3, rc);
    else printf("  %-44s %8.1f us   (rc %d)\n", name, best * 1e3, rc);
    return best;
  };
  const char* e = getenv("GSTVD_ADAMW_BG");
  printf("weight-gradient GEMM x %d (%.0f GFLOP) and AdamW over %lld Mi parameters (%.2f GB), GSTVD_ADAMW_BG=%s\n", nb,
         2.0 * M * N * K * nb * 1e-9, (long long)(na >> 20), na * 30e-9, e ? e : "0");
  const float tg = run("GEMM alone", 0), ta = run("AdamW alone", 1), ts = run("one stream: GEMM then AdamW", 2);
  const float t3 = run("two streams, GEMM issued first", 3), t4 = run("two streams, AdamW issued first", 4);
  printf("  sum of alone %.1f, max of alone %.1f; together / sum = %.3f (GEMM first), %.3f (AdamW first); one stream / sum = %.3f\n",
         (tg + ta) * 1e3, (tg > ta ? tg : ta) * 1e3, t3 / (tg + ta), t4 / (tg + ta), ts / (tg + ta));
  return 0;
}
