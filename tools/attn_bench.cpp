// gstvd_attn_fwd / _bwd of one shape, back to back from a C++ loop (no Python between launches): us per launch.
//   build/attn_bench B nh Lq Lk d causal ldq ldk p [bwd]
//   hipcc -O2 --offload-arch=gfx950 tools/attn_bench.cpp -o build/attn_bench -Lgst_visdial_amd/lib -lgstvd_hip -Wl,-rpath,$PWD/gst_visdial_amd/lib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/gstvd_hip.h"
static void* dmalloc(size_t n, int byte = 0) { void* p; if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); } (void)hipMemset(p, byte, n); return p; }
int main(int argc, char** argv) {
  if (argc < 10) { fprintf(stderr, "usage: attn_bench B nh Lq Lk d causal ldq ldk p [bwd]\n"); return 2; }
  const int B = atoi(argv[1]), nh = atoi(argv[2]), Lq = atoi(argv[3]), Lk = atoi(argv[4]), d = atoi(argv[5]), causal = atoi(argv[6]);
  const int64_t ldq = atoll(argv[7]), ldk = atoll(argv[8]); const float p = (float)atof(argv[9]); const bool bwd = argc > 10;
  const int64_t H = (int64_t)nh * d;
  void* Q = dmalloc((size_t)B * Lq * ldq * 2, 0x3c); void* K = dmalloc((size_t)B * Lk * ldk * 2, 0x3c); void* V = dmalloc((size_t)B * Lk * ldk * 2, 0x3c);
  void* O = dmalloc((size_t)B * Lq * H * 2); float* LSE = (float*)dmalloc((size_t)B * nh * Lq * 4);
  void* dO = dmalloc((size_t)B * Lq * H * 2, 0x3c); void* dQ = dmalloc((size_t)B * Lq * ldq * 2); void* dK = dmalloc((size_t)B * Lk * ldk * 2); void* dV = dmalloc((size_t)B * Lk * ldk * 2);
  float* delta = (float*)dmalloc((size_t)B * nh * Lq * 4);
  std::vector<float> ones((size_t)B * Lk, 1.f); float* mask = (float*)dmalloc((size_t)B * Lk * 4); (void)hipMemcpy(mask, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
  uint64_t hr[2] = {77, 0}; uint64_t* rng = (uint64_t*)dmalloc(16); (void)hipMemcpy(rng, hr, 16, hipMemcpyHostToDevice);
  gstvd_attn_t a; memset(&a, 0, sizeof(a));
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE; a.key_mask = mask; a.ldq = ldq; a.ldk = ldk; a.ldv = ldk; a.ldo = H;
  a.B = B; a.nh = nh; a.Lq = Lq; a.Lk = Lk; a.d = d; a.causal = causal; a.dtype = GSTVD_BF16; a.mask_neg = -10000.f; a.scale = 0.125f; a.dropout_p = p; a.site = 3; a.rng = rng;
  a.dO = dO; a.lddo = H; a.dQ = dQ; a.dK = dK; a.dV = dV; a.lddq = ldq; a.lddk = ldk; a.lddv = ldk; a.delta = delta;
  hipStream_t s; (void)hipStreamCreate(&s); hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int rc = 0; float best = 1e9f; const int n = 50;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) rc |= bwd ? gstvd_attn_bwd(&a, s) : gstvd_attn_fwd(&a, s);
    (void)hipEventRecord(e1, s); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%s B %d nh %d Lq %d Lk %d d %d causal %d ldq %lld ldk %lld p %.2f: %.2f us per launch (rc %d)\n", bwd ? "bwd" : "fwd", B, nh, Lq, Lk, d, causal,
         (long long)ldq, (long long)ldk, p, best * 1e3 / n, rc);
  return 0;
}
