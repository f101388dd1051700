// Which XCD does workgroup b of a large grid run on, and in what order do an XCD's workgroups start?  (round 5: the placement of
// the grouped weight-gradient launch assumes b % 8 names the XCD also beyond the first round of the chip, with one 139 KB-LDS
// workgroup per CU and workgroups of very different running times.)
//   hipcc --offload-arch=gfx950 -O2 -o build/xcd_probe tools/xcd_probe.cpp && build/xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(768) void probe(unsigned* xcc, unsigned long long* t0, unsigned long long* t1, const int* dur_us) {
  extern __shared__ char smem[];
  const int b = blockIdx.x;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const unsigned long long start = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  if (threadIdx.x == 0) { xcc[b] = id & 0xf; t0[b] = start; smem[0] = 1; }
  const unsigned long long until = start + (unsigned long long)dur_us[b] * 100ull;
  while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) t1[b] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 5352;
  std::vector<int> dur(n);
  srand(3);
  for (int i = 0; i < n; ++i) dur[i] = (i * 7 / n) % 2 ? 20 + rand() % 40 : 100 + rand() % 60;    // blocks of long and short "tiles", jittered
  unsigned* xcc; unsigned long long *t0, *t1; int* d;
  hipMalloc(&xcc, n * 4); hipMalloc(&t0, n * 8); hipMalloc(&t1, n * 8); hipMalloc(&d, n * 4);
  hipMemcpy(d, dur.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 139 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(768), 139 * 1024, 0, xcc, t0, t1, d);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> hx(n); std::vector<unsigned long long> h0(n), h1(n);
  hipMemcpy(hx.data(), xcc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h0.data(), t0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), t1, n * 8, hipMemcpyDeviceToHost);
  // (1) is the XCD of block b the XCD of block b % 8 ?
  int bad = 0, first_bad = -1;
  for (int b = 8; b < n; ++b) if (hx[b] != hx[b % 8]) { if (first_bad < 0) first_bad = b; ++bad; }
  printf("blocks %d: XCD ids of blocks 0..7 = ", n);
  for (int b = 0; b < 8; ++b) printf("%u ", hx[b]);
  printf("\nblocks whose XCD differs from block (b %% 8)'s: %d (first at %d)\n", bad, first_bad);
  // (2) within an XCD: do blocks start in id order?  how many start before an earlier-numbered block of the same XCD has started?
  for (unsigned x = 0; x < 8; ++x) {
    std::vector<int> ids;
    for (int b = 0; b < n; ++b) if (hx[b] == hx[x]) ids.push_back(b);
    int inv = 0; unsigned long long last = 0;
    for (int b : ids) { if (h0[b] < last) ++inv; last = std::max(last, h0[b]); }
    // concurrency: at the start of each block, how many blocks of this XCD are running
    long conc = 0;
    for (int b : ids) { int c = 0; for (int o : ids) if (h0[o] <= h0[b] && h1[o] > h0[b]) ++c; conc += c; }
    printf("XCD slot %u: %zu blocks, start-order inversions %d, mean concurrency at block start %.1f, span %.0f us\n", x, ids.size(), inv,
           (double)conc / ids.size(), (double)(h1[ids.back()] - h0[ids.front()]) / 100.0);
  }
  return 0;
}
