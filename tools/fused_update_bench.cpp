// Weight gradients + AdamW: grouped launch followed by the AdamW pass (what the step's tail does) against ONE launch whose tiles update
// their weights in the epilogue (gstvd_gemm_grouped_adamw).  `nb` problems dW[3072 x 768] = dY^T X over K = 4096 rows (the FFN-up
// weight gradient of a text layer), plus `ns` short-K problems (K = 592: the vision stream's 1024 x 1024 weights) -- the mix decides
// how much of the update hides under other tiles' K-loops.  Checks that both paths leave bit-identical param / m / v / shadow.
//   hipcc -O2 --offload-arch=gfx950 tools/fused_update_bench.cpp -o build/fused_update_bench -Lgst_visdial_amd/lib -lgstvd_hip -Wl,-rpath,$PWD/gst_visdial_amd/lib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../include/gstvd_hip.h"

static void* dmalloc(size_t n, int byte = 0) { void* p; if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", n); exit(1); } (void)hipMemset(p, byte, n); return p; }
static void fill_bf16(void* p, size_t n, float scale, unsigned seed) {
  std::vector<unsigned short> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    float v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale;
    unsigned u; memcpy(&u, &v, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  (void)hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice);
}
static void fill_f32(float* p, size_t n, float scale, unsigned seed, float bias = 0.f) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 999u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = bias + ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale; }
  (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 24, ns = argc > 2 ? atoi(argv[2]) : 12;
  const int order = argc > 3 ? atoi(argv[3]) : 0;       // 1: per-XCD queues of whole problems, long K first (round 5's block_map_dev)
  const int np = nb + ns;
  std::vector<int64_t> M(np), N(np), K(np), off(np + 1);
  int64_t tot = 1024;                                             // flat offset of the first weight (something sits in front of it)
  for (int i = 0; i < np; ++i) {
    const bool big = ((int64_t)i * ns) / np == ((int64_t)(i + 1) * ns) / np;          // the short-K problems spread evenly over the table
    if (big) { M[i] = 3072; N[i] = 768; K[i] = 4096; } else { M[i] = 1024; N[i] = 1024; K[i] = 592; }
    off[i] = tot; tot += M[i] * N[i] + 768;                       // a bias-sized gap between weights: the remainder pass's work
  }
  off[np] = tot;
  const int64_t n = tot;
  // operands: ONE dY / X pair per shape class is enough for timing the epilogue; separate buffers per problem keep HBM traffic honest
  std::vector<void*> A(np), B(np);
  for (int i = 0; i < np; ++i) {
    A[i] = dmalloc((size_t)K[i] * M[i] * 2); B[i] = dmalloc((size_t)K[i] * N[i] * 2);
    fill_bf16(A[i], (size_t)K[i] * M[i], 0.05f, 100 + i); fill_bf16(B[i], (size_t)K[i] * N[i], 1.f, 200 + i);
  }
  float* G = (float*)dmalloc(n * 4);
  float *P[2], *Mo[2], *V[2]; void* S[2];
  for (int c = 0; c < 2; ++c) {
    P[c] = (float*)dmalloc(n * 4); Mo[c] = (float*)dmalloc(n * 4); V[c] = (float*)dmalloc(n * 4); S[c] = dmalloc(n * 2);
    fill_f32(P[c], n, 0.05f, 1); fill_f32(Mo[c], n, 0.01f, 2); fill_f32(V[c], n, 1e-4f, 3, 2e-4f);
  }
  std::vector<float> g0(n, 0.f);
  for (int64_t i = 0; i < n; ++i) g0[i] = ((i * 2654435761u) % 1000) * 1e-5f;     // the gaps' gradients (bias-like)
  (void)hipMemcpy(G, g0.data(), n * 4, hipMemcpyHostToDevice);
  // segments: every weight and every gap its own segment, alternating decay
  std::vector<int64_t> seg; std::vector<float> hp; std::vector<uint8_t> skip;
  seg.push_back(1024); hp.push_back(1e-4f); hp.push_back(0.f); skip.push_back(0);
  for (int i = 0; i < np; ++i) {
    seg.push_back(off[i] + M[i] * N[i]); hp.push_back(1e-4f * (1 + i % 3)); hp.push_back(0.01f); skip.push_back(1);
    seg.push_back(off[i + 1]); hp.push_back(2e-4f); hp.push_back(0.f); skip.push_back(0);
  }
  const int64_t nseg = (int64_t)seg.size();
  int64_t* seg_d = (int64_t*)dmalloc(nseg * 8); float* hp_d = (float*)dmalloc(nseg * 8); uint8_t* skip_d = (uint8_t*)dmalloc(nseg);
  (void)hipMemcpy(seg_d, seg.data(), nseg * 8, hipMemcpyHostToDevice); (void)hipMemcpy(hp_d, hp.data(), nseg * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(skip_d, skip.data(), nseg, hipMemcpyHostToDevice);
  float step_h = 3.f; float* step = (float*)dmalloc(4); (void)hipMemcpy(step, &step_h, 4, hipMemcpyHostToDevice);
  // the remainder pass's block list: every 1024-element block that holds an element outside the weights
  std::vector<int32_t> blocks;
  {
    std::vector<uint8_t> need((n + 1023) / 1024, 0);
    auto mark = [&](int64_t a, int64_t b) { for (int64_t k = a / 1024; k <= (b - 1) / 1024; ++k) need[k] = 1; };
    mark(0, 1024);
    for (int i = 0; i < np; ++i) mark(off[i] + M[i] * N[i], off[i + 1]);
    for (size_t k = 0; k < need.size(); ++k) if (need[k]) blocks.push_back((int32_t)k);
  }
  int32_t* blocks_d = (int32_t*)dmalloc(blocks.size() * 4); (void)hipMemcpy(blocks_d, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice);
  // tables
  std::vector<gstvd_gemm_t> tab(np), tabf(np); std::vector<int32_t> toff(np); int32_t tiles = 0; double flops = 0;
  for (int i = 0; i < np; ++i) {
    gstvd_gemm_t g; memset(&g, 0, sizeof(g));
    g.A = A[i]; g.B = B[i]; g.C = G + off[i]; g.M = M[i]; g.N = N[i]; g.K = K[i]; g.lda = M[i]; g.ldb = N[i]; g.ldc = N[i]; g.batch = 1;
    g.dtype_in = GSTVD_BF16; g.dtype_out = GSTVD_F32; g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = 1.f;
    tab[i] = g; g.epilogue = GSTVD_EPI_ADAMW; g.addend = hp_d + 2 * (1 + 2 * i); tabf[i] = g;
    toff[i] = tiles; tiles += (int32_t)(((M[i] + 255) / 256) * ((N[i] + 255) / 256)); flops += 2.0 * M[i] * N[i] * K[i];
  }
  gstvd_gemm_t* tab_d = (gstvd_gemm_t*)dmalloc(np * sizeof(gstvd_gemm_t)); gstvd_gemm_t* tabf_d = (gstvd_gemm_t*)dmalloc(np * sizeof(gstvd_gemm_t));
  int32_t* toff_d = (int32_t*)dmalloc(np * 4);
  (void)hipMemcpy(tab_d, tab.data(), np * sizeof(gstvd_gemm_t), hipMemcpyHostToDevice); (void)hipMemcpy(tabf_d, tabf.data(), np * sizeof(gstvd_gemm_t), hipMemcpyHostToDevice);
  (void)hipMemcpy(toff_d, toff.data(), np * 4, hipMemcpyHostToDevice);
  // round 5: the placement ops.xcd_block_map builds, restated for this table -- whole problems dealt to the least-loaded XCD queue,
  // long-K problems first; workgroup b runs tile bmap[b] (entries b, b + 8, ... = one XCD's queue), -1 = idle
  int32_t* bmap_d = nullptr; int64_t nblocks = 0;
  if (order) {
    std::vector<int> idx(np);
    for (int i = 0; i < np; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return K[a] > K[b]; });
    std::vector<std::vector<int32_t>> q(8); double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : idx) {
      int x = 0; for (int j = 1; j < 8; ++j) if (load[j] < load[x]) x = j;
      const int32_t nt = (int32_t)(((M[i] + 255) / 256) * ((N[i] + 255) / 256));
      for (int32_t t = 0; t < nt; ++t) q[x].push_back(toff[i] + t);
      load[x] += nt * (35.0 + 0.76 * ((K[i] + 31) / 32));
    }
    size_t depth = 0; for (auto& v : q) depth = v.size() > depth ? v.size() : depth;
    std::vector<int32_t> bm(depth * 8, -1);
    for (int x = 0; x < 8; ++x) for (size_t j = 0; j < q[x].size(); ++j) bm[j * 8 + x] = q[x][j];
    nblocks = (int64_t)bm.size();
    bmap_d = (int32_t*)dmalloc(bm.size() * 4); (void)hipMemcpy(bmap_d, bm.data(), bm.size() * 4, hipMemcpyHostToDevice);
    printf("block map: %lld workgroups for %d tiles, queue depths", (long long)nblocks, tiles);
    for (auto& v : q) printf(" %zu", v.size());
    printf("\n");
  }
  hipStream_t s0; (void)hipStreamCreate(&s0);
  hipEvent_t e0, e1, em; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&em);
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-6f, gs = 1.f;
  auto plain = [&](int c) {
    int rc = gstvd_gemm_grouped(tab_d, toff_d, np, tiles, GSTVD_BF16, GSTVD_F32, 1, 1, bmap_d, nblocks, s0);
    (void)hipEventRecord(em, s0);
    rc |= gstvd_adamw(P[c], G, Mo[c], V[c], S[c], n, seg_d, hp_d, nseg, b1, b2, eps, step, gs, 0, s0);
    return rc;
  };
  auto fused = [&](int c) {
    gstvd_adamw_fuse_t f; memset(&f, 0, sizeof(f));
    f.grad_base = G; f.param = P[c]; f.m = Mo[c]; f.v = V[c]; f.shadow_bf16 = S[c]; f.step = step;
    f.beta1 = b1; f.beta2 = b2; f.eps = eps; f.grad_scale = gs; f.write_grad = 0;
    int rc = gstvd_gemm_grouped_adamw(tabf_d, toff_d, np, tiles, &f, bmap_d, nblocks, s0);
    (void)hipEventRecord(em, s0);
    rc |= gstvd_adamw_blocks(P[c], G, Mo[c], V[c], S[c], n, seg_d, hp_d, nseg, b1, b2, eps, step, gs, 0, blocks_d, (int64_t)blocks.size(), skip_d, s0);
    return rc;
  };
  // correctness first: one update each from identical states
  int rc = plain(0); rc |= fused(1); (void)hipDeviceSynchronize();
  std::vector<float> h0(n), h1(n); std::vector<unsigned short> s0h(n), s1h(n);
  const char* names[3] = {"param", "m", "v"}; float* bufs[2][3] = {{P[0], Mo[0], V[0]}, {P[1], Mo[1], V[1]}};
  int bad = 0;
  for (int k = 0; k < 3; ++k) {
    (void)hipMemcpy(h0.data(), bufs[0][k], n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), bufs[1][k], n * 4, hipMemcpyDeviceToHost);
    int64_t nd = 0, first = -1; for (int64_t i = 0; i < n; ++i) if (memcmp(&h0[i], &h1[i], 4)) { if (first < 0) first = i; ++nd; }
    printf("  %-6s: %lld of %lld elements differ%s", names[k], (long long)nd, (long long)n, nd ? "" : "  (bit-identical)\n");
    if (nd) { printf("  first at %lld: %.9g vs %.9g\n", (long long)first, h0[first], h1[first]); bad = 1; }
  }
  (void)hipMemcpy(s0h.data(), S[0], n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(s1h.data(), S[1], n * 2, hipMemcpyDeviceToHost);
  { int64_t nd = 0; for (int64_t i = 0; i < n; ++i) nd += s0h[i] != s1h[i]; printf("  shadow: %lld differ\n", (long long)nd); bad |= nd != 0; }
  printf("launch rc %d, parity %s\n", rc, bad ? "FAILED" : "ok");
  auto time = [&](const char* name, bool f) {
    float best = 1e9f, mid = 0;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, s0);
      if (f) fused(1); else plain(0);
      (void)hipEventRecord(e1, s0);
      (void)hipDeviceSynchronize();
      float ms, m2; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventElapsedTime(&m2, e0, em);
      if (rep >= 1 && ms < best) { best = ms; mid = m2; }
    }
    printf("  %-46s %8.1f us  (first launch %.1f us, second %.1f us)\n", name, best * 1e3, mid * 1e3, (best - mid) * 1e3);
    return best;
  };
  printf("%d long-K + %d short-K weight gradients: %.0f GFLOP, %d tiles, %.1f Mi weights (%.2f GB of AdamW traffic), remainder %zu blocks\n",
         nb, ns, flops * 1e-9, tiles, n / 1048576.0, n * 30e-9, blocks.size());
  const float tp = time("grouped weight gradients, then AdamW", false), tf = time("one launch with the update in its epilogue", true);
  printf("  fused / plain = %.3f\n", tf / tp);
  return bad;
}
