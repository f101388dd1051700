// Two chains of dependent small GEMMs on one / two streams, launched from a tight C++ loop (no Python, no hipGraph):
// what does real stream concurrency buy for latency-bound launches?   hipcc -O2 --offload-arch=gfx950 tools/chain_bench.cpp -o build/chain_bench -Lgst_visdial_amd/lib -lgstvd_hip -Wl,-rpath,$PWD/gst_visdial_amd/lib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include "../include/gstvd_hip.h"   /* (tools/ -> include/) */
static void fill(gstvd_gemm_t& g, void* A, void* B, void* C, int64_t M, int64_t N, int64_t K) {
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.batch = 1;
  g.dtype_in = GSTVD_BF16; g.dtype_out = GSTVD_BF16; g.alpha = 1.f;
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoi(argv[1]) : 400, N = 768, K = argc > 3 ? atoi(argv[3]) : 768;
  const int n = argc > 2 ? atoi(argv[2]) : 96;
  void *A, *B, *C0, *C1;
  hipMalloc(&A, 4096 * K * 2 + 4096); hipMalloc(&B, N * K * 2 + 4096); hipMalloc(&C0, 4096 * N * 2); hipMalloc(&C1, 4096 * N * 2);
  hipMemset(A, 0, 4096 * K * 2); hipMemset(B, 0, N * K * 2);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1, ej; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  gstvd_gemm_t gf, gh0, gh1;
  fill(gf, A, B, C0, M, N, K);
  fill(gh0, A, B, C0, M / 2, N, K);
  fill(gh1, (char*)A + (M / 2) * K * 2, B, C1, M - M / 2, N, K);
  auto run = [&](int mode) {   // 0: one chain of n full launches; 1: two chains of n half launches on two streams; 2: the same 2n half launches on one stream
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::high_resolution_clock::now();
      hipEventRecord(e0, s0);
      if (mode == 1) { hipEventRecord(ej, s0); hipStreamWaitEvent(s1, ej, 0); }
      for (int i = 0; i < n; ++i) {
        if (mode == 0) gstvd_gemm(&gf, s0);
        else { gstvd_gemm(&gh0, s0); gstvd_gemm(&gh1, mode == 1 ? s1 : s0); }
      }
      if (mode == 1) { hipEventRecord(ej, s1); hipStreamWaitEvent(s0, ej, 0); }
      hipEventRecord(e1, s0);
      auto t1 = std::chrono::high_resolution_clock::now();
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("  mode %d: GPU %.1f us total = %.2f us per chain step; host issue %.1f us\n", mode, ms * 1e3, ms * 1e3 / n,
                           std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
  };
  printf("M = %ld, %d steps per chain\n", (long)M, n);
  run(0); run(1); run(2);
  // mode 3 / 4: the mode 0 / mode 1 launches captured into a hipGraph and replayed
  for (int mode = 3; mode <= 4; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal);
    if (mode == 4) { hipEventRecord(ej, s0); hipStreamWaitEvent(s1, ej, 0); }
    for (int i = 0; i < n; ++i) {
      if (mode == 3) gstvd_gemm(&gf, s0);
      else { gstvd_gemm(&gh0, s0); gstvd_gemm(&gh1, s1); }
    }
    if (mode == 4) { hipEventRecord(ej, s1); hipStreamWaitEvent(s0, ej, 0); }
    hipStreamEndCapture(s0, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 4; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::high_resolution_clock::now();
      hipEventRecord(e0, s0);
      hipGraphLaunch(ge, s0);
      hipEventRecord(e1, s0);
      auto t1 = std::chrono::high_resolution_clock::now();
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 3) printf("  mode %d (hipGraph replay of mode %d): GPU %.1f us total = %.2f us per chain step; host %.1f us\n", mode, mode - 3, ms * 1e3,
                           ms * 1e3 / n, std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
