// bf16 MFMA GEMM main loop with direct-to-LDS loads (global_load_lds_dwordx4) and an NS-deep LDS ring.
//
// Why: the GEMMs of this path are short-K (768..3072) and often skinny (M = 400/592 rows), so a workgroup
// spends its life in the load->use latency chain.  Here every thread keeps (NS-1) K-tiles of LDS-DMA in
// flight, waits with a COUNTED s_waitcnt vmcnt (never 0 inside the loop), and there is one raw s_barrier per
// K-tile.  No staging registers, no ds_write pass.
//
//  * LDS-DMA writes wave-uniform base + lane*16, i.e. the LDS image is lane-linear; the XOR swizzle that
//    makes ds_read_b128 / ds_read_b64_tr_b16 conflict-free is therefore applied to the per-lane SOURCE
//    address (and, identically, to the read address) -- cdna guide rule 21;
//  * tails (M, N, K not tile multiples) never mask lanes: an out-of-range 16-byte unit reads a device-side
//    zero page instead, so contraction tails are zero filled and the DMA count per stage stays constant;
//  * blockIdx is remapped so that each XCD (private L2) walks a contiguous range of output tiles.
#include "gemm_common.h"
#include <stdlib.h>

__device__ __attribute__((aligned(256))) char g_zero_page[256];


DEVFN void glds16(const char* src, char* lds_wave_base) {
  glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(lds_wave_base)));
}

// Per-thread description of the 16-byte units this thread feeds into one operand image, fixed over the K loop.
template <int BX, bool KM, int NP, int NT>
struct DmaUnits {
  const char* ptr[NP];   // address of the unit in K-tile 0 (meaningless when !okx)
  int kofs[NP];          // k index of the unit inside a K-tile
  bool okx[NP];
  int64_t kstep;         // byte advance per K-tile

  DEVFN void init(const char* g, int64_t ld, int64_t x0, int64_t X, int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int u = i * NT + tid;
      int64_t x;
      if (!KM) {
        const int r = u >> 3, ls = ((u & 7) ^ r) & 7;
        x = x0 + r;
        kofs[i] = ls * 8;
        ptr[i] = g + (x * ld + ls * 8) * 2;
      } else {
        constexpr int VPR = BX / 8, NB = BX / 16;
        const int kr = u / VPR, cu = u % VPR;
        const int f = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int lb = ((cu >> 1) ^ f) & (NB - 1);
        x = x0 + lb * 16 + (cu & 1) * 8;
        kofs[i] = kr;
        ptr[i] = g + ((int64_t)kr * ld + x) * 2;
      }
      okx[i] = x < X;
    }
    kstep = KM ? 64 * ld * 2 : 128;
  }
  DEVFN void issue(int64_t kt, int64_t K, char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool ok = okx[i] && (kt * 64 + kofs[i] < K);
      const char* src = ok ? ptr[i] + kt * kstep : (const char*)g_zero_page;
      glds16(src, img + (i * NT + wave * 64) * 16);
    }
  }
};

// One output tile: the whole K loop + epilogue.  `wg` is the tile's index in the problem's L2-friendly order.
// Split-K (S > 1): the tile's K range is cut into S contiguous runs of K-tiles, one workgroup each.  Every workgroup
// parks its fp32 accumulators in `ws`, the last one to arrive (per-tile counter) adds the S partials in split order --
// so the sum does not depend on arrival order -- and runs the epilogue; it also re-arms the counter.
// NIU > 0: only NIU of a wave's BN/WN/16 column tiles are used -- the output tile is WN*NIU*16 wide inside the same BN-wide
// LDS image (96 of 128 columns with WN = 2, NIU = 3: N = 768 then makes 256 tiles, one per CU, instead of 192).
template <typename OT, int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS, int NIU = 0>
DEVFN void dma_tile(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem, int S = 1, int split = 0,
                    float* ws = nullptr, int* cnt = nullptr) {
  constexpr int NT = WM * WN * 64, WTM = BM / WM, MI = WTM / 16, NI = NIU > 0 ? NIU : BN / WN / 16, WTN = NI * 16, BNU = WN * WTN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NPA = A_BYTES / (NT * 16), NPB = B_BYTES / (NT * 16), LPS = NPA + NPB;
  static_assert(NPA >= 1 && NPB >= 1, "tile too small for the thread count");
  static_assert((NS - 2) * LPS <= 63, "vmcnt immediate out of range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // tiles are numbered in column strips 4 n-tiles wide, so ~32 consecutive ids form an (8 m) x (4 n) block
  const int ntm = nwg / ntn;
  const int strip = wg / (4 * ntm), sw = (ntn - strip * 4) < 4 ? (ntn - strip * 4) : 4;
  const int within = wg - strip * 4 * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * 4 + within % sw) * BNU;

  DmaUnits<BM, AKM, NPA, NT> ua;
  DmaUnits<BN, BKM, NPB, NT> ub;
  ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, tid);
  ub.init(p.B + z * p.sB * 2, p.ldb, n0, (n0 + BNU < p.N) ? n0 + BNU : p.N, tid);    // columns past the tile read the zero page

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t nkt_all = (p.K + 63) / 64, per = (nkt_all + S - 1) / S;
  const int64_t kt0 = (int64_t)split * per, kt1 = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;
  const int64_t Kend = (kt1 * 64 < p.K) ? kt1 * 64 : p.K;      // units past this split's K range read the zero page
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    ua.issue(kt0 + s, Kend, smem + s * STAGE, wave);
    ub.issue(kt0 + s, Kend, smem + s * STAGE + A_BYTES, wave);
  }
  int slot = 0, fill = NS - 1;
  for (int64_t t = kt0; t < kt1; ++t) {
    // K-tile t has landed for this wave's own DMA once at most (NS-2) younger stages are still in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();      // ... and for everybody's; everybody is also done reading slot `fill`
    asm volatile("" ::: "memory");     // s_barrier is IntrNoMem: keep the LDS reads / DMA issue below it
    ua.issue(t + NS - 1, Kend, smem + fill * STAGE, wave);
    ub.issue(t + NS - 1, Kend, smem + fill * STAGE + A_BYTES, wave);
    const char* cA = smem + slot * STAGE;
    const char* cB = cA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = frag_bf16<BM, AKM>(cA, wm * WTM + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = frag_bf16<BN, BKM>(cB, wn * WTN + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa[i], acc[i][j]);
    }
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    fill = (fill + 1 == NS) ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing (zero page) DMAs must land before LDS is released

  if (S > 1) {
    // Workgroups of one tile may sit on different XCDs (private, mutually non-coherent L2s).  Partials and the arrival
    // counter therefore move with agent-scope relaxed atomics (sc1: performed at the device coherence point) and the
    // ordering is explicit -- stores retired (vmcnt 0) by every thread before the counter is bumped -- instead of
    // agent-scope fences, which write back / invalidate the whole L2 per workgroup.
    constexpr int APT = MI * NI * 4;                       // fp32 accumulator elements per thread
    float* mine = ws + (int64_t)(wg * S + split) * NT * APT + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __hip_atomic_store(mine + ((i * NI + j) * 4 + e) * NT, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)smem;
    if (tid == 0) *flag = __hip_atomic_fetch_add(cnt + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(cnt + wg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {
      const float* part = ws + (int64_t)(wg * S + q) * NT * APT + tid;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][j][e] += __hip_atomic_load(part + ((i * NI + j) * 4 + e) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  constexpr int EPI_LPR = NI <= 1 ? 2 : (NI <= 2 ? 4 : 8);           // lanes per row of the row-wise epilogue
  if constexpr (sizeof(OT) == 2 && (MI < 4 ? MI : 4) * 16 >= 64 / EPI_LPR) {      // (a 16x16 wave tile has too few rows for it)
    if (epilogue_rows_ok(p)) {
      constexpr int HB = MI < 4 ? MI : 4;
      static_assert(WM * WN * epi_wave_bytes<NI, HB>() <= NS * STAGE, "epilogue parking must fit in the ring");
      __builtin_amdgcn_s_barrier();            // every wave is done reading the ring: its LDS is free for the row-wise epilogue
      gemm_epilogue_rows<MI, NI, HB>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, HB>(), lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<bf16, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li, n0 + wn * WTN + j * 16 + 4 * g);
}

// XCD-aware block order: blocks b and b+8 share an XCD (private 4 MiB L2); give each XCD a contiguous run of ids
DEVFN int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename OT, int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS, int NIU = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_kernel(GemmP p, int ntn, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dma_tile<OT, BM, BN, WM, WN, AKM, BKM, NS, NIU>(p, blockIdx.y, xcd_remap(blockIdx.x, nwg), ntn, nwg, smem);
}

template <typename OT, int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_splitk_kernel(GemmP p, int ntn, int ntiles, int S, float* ws, int* cnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = xcd_remap(blockIdx.x, ntiles * S);
  dma_tile<OT, BM, BN, WM, WN, AKM, BKM, NS>(p, 0, id / S, ntn, ntiles, smem, S, id % S, ws, cnt);
}

template <typename OT, int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS, int NIU = 0>
static int dma_launch(const GemmP& p, int64_t batch, hipStream_t s) {
  constexpr int lds = NS * (BM + BN) * 128, BNU = NIU > 0 ? WN * NIU * 16 : BN;
  auto k = gemm_dma_kernel<OT, BM, BN, WM, WN, AKM, BKM, NS, NIU>;
  static int attr_rc = ensure_lds(k, lds);
  if (attr_rc) return attr_rc;
  const int ntm = (int)((p.M + BM - 1) / BM), ntn = (int)((p.N + BNU - 1) / BNU);
  GSTVD_LAUNCH(k, dim3((unsigned)(ntm * ntn), (unsigned)batch), dim3(WM * WN * 64), lds, s, p, ntn, ntm * ntn);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

template <typename OT, bool AKM, bool BKM>
static int dma_layout(const GemmP& p, int64_t batch, hipStream_t s) {
  const int64_t big = ((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  if (p.M >= 256 && p.N >= 128 && big >= 96) {
    // 96-wide tiles when they turn a partially filled round of CUs into a full one (N = 768: 192 -> 256 tiles)
    const int64_t ntm = (p.M + 127) / 128, t128 = ntm * ((p.N + 127) / 128) * batch, t96 = ntm * ((p.N + 95) / 96) * batch;
    const double nkt = (double)((p.K + 63) / 64);
    const double c128 = (double)((t128 + 255) / 256) * (8.0 + 0.45 * nkt), c96 = (double)((t96 + 255) / 256) * (8.0 + 0.40 * nkt);
    // Ring depth 3 (96 KB), not 5 (160 KB = the whole LDS of a CU): the K-step rate is the same (0.52-0.65 against
    // 0.56-0.62 us per 64-deep step -- the loop is bound by the L2 -> LDS feed rate, not by the bytes in flight), and the
    // 64 KB left free let a workgroup of another stream's kernel share the CU: 1174-1178 against 1167-1172 rounds/s in
    // the step (round 3).
    if (c96 < c128) return dma_launch<OT, 128, 128, 4, 2, AKM, BKM, 3, 3>(p, batch, s);
    return dma_launch<OT, 128, 128, 2, 4, AKM, BKM, 3>(p, batch, s);
  }
  // (32x32 / 32x64 tiles for the skinny M = 400 / 592 problems were measured in round 2: no faster -- 400x768x768 9.8-11.0 us
  // against 10.2 us, 592x1024x1024 16.5 us against 13.1 us -- the ring of a workgroup is latency bound, 7 stages x stage
  // bytes in flight per ~2 us round trip, so halving the tile halves the bytes in flight along with the bytes needed.
  // 64x32 tiles -- half the weight bytes per workgroup, twice the workgroups: 8.7-8.8 us against 9.1-9.5 us at M = 400, 18 us
  // against 13 us at M = 592.  Not adopted.)
  // (ring depth 8 / 6 / 5 / 4 measured inside the step in round 3: 1198 / 1202 / 1201 / 1200 rounds/s, i.e. no difference)
  // Round 5: THREE stages (48 KB of LDS) instead of eight (128 KB).  Alone the kernel runs the same (round 3's depth sweep), but a
  // 128 KB workgroup of the vision / decoder chains could only start on a CU that a text-chain workgroup (96 KB, 128-tile kernels)
  // had just left -- 8.6-11.9 us alone against 25.8-33.8 us whenever a text kernel was in flight (profiles/r04_overlap_stats.txt);
  // with 48 KB it fits beside one: whole step 12.40 / 12.45 (8 stages) vs 12.37 / 12.37 (4) vs 12.23 / 12.24 ms (3),
  // profiles/r05_gemm64_ns_ab.txt.
  return dma_launch<OT, 64, 64, 2, 2, AKM, BKM, 3>(p, batch, s);
}

template <typename OT>
static int dma_out(const GemmP& p, int64_t batch, int akm, int bkm, hipStream_t s) {
  if (!akm && !bkm) return dma_layout<OT, false, false>(p, batch, s);
  if (!akm && bkm) return dma_layout<OT, false, true>(p, batch, s);
  if (akm && bkm) return dma_layout<OT, true, true>(p, batch, s);
  return dma_layout<OT, true, false>(p, batch, s);
}

constexpr int SPLITK_MAX_TILES = 1024;      // counters occupy the first 4 KiB of the scratch, partials follow

template <typename OT, bool AKM, bool BKM, int NS>
static int splitk_launch_ns(const GemmP& p, int S, void* ws, int64_t ws_bytes, hipStream_t s);

template <typename OT, bool AKM, bool BKM>
static int splitk_launch(const GemmP& p, int S, void* ws, int64_t ws_bytes, hipStream_t s) {
  // ring depth of the split-K form: 8 (four stages / 64 KB measured no better inside the step, profiles/r05_gemm64_sk_ns_ab.txt)
  return splitk_launch_ns<OT, AKM, BKM, 8>(p, S, ws, ws_bytes, s);
}

template <typename OT, bool AKM, bool BKM, int NS>
static int splitk_launch_ns(const GemmP& p, int S, void* ws, int64_t ws_bytes, hipStream_t s) {
  constexpr int BM = 64, BN = 64, lds = NS * (BM + BN) * 128;
  auto k = gemm_dma_splitk_kernel<OT, BM, BN, 2, 2, AKM, BKM, NS>;
  static int attr_rc = ensure_lds(k, lds);
  if (attr_rc) return attr_rc;
  const int ntm = (int)((p.M + BM - 1) / BM), ntn = (int)((p.N + BN - 1) / BN), tiles = ntm * ntn;
  if (ws_bytes < gstvd_gemm_splitk_ws_bytes(p.M, p.N, S)) return GSTVD_E_SHAPE;
  if (tiles > SPLITK_MAX_TILES) return GSTVD_E_SHAPE;
  int* cnt = (int*)ws;                                                   // [tiles] arrival counters (zero between launches)
  float* part = (float*)((char*)ws + SPLITK_MAX_TILES * 4);             // fixed layout: a scratch serves launches of any shape
  GSTVD_LAUNCH(k, dim3((unsigned)(tiles * S)), dim3(256), lds, s, p, ntn, tiles, S, part, cnt);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t gstvd_gemm_splitk_ws_bytes(int64_t M, int64_t N, int32_t splits) {
  const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
  return SPLITK_MAX_TILES * 4 + tiles * splits * 64 * 64 * 4;
}

int gemm_dma_splitk_dispatch(const GemmP& p, int akm, int bkm, int out_f32, int S, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (akm && !bkm) return GSTVD_E_UNSUPPORTED;
  if (out_f32) {
    if (!akm && !bkm) return splitk_launch<float, false, false>(p, S, ws, ws_bytes, s);
    if (!akm && bkm) return splitk_launch<float, false, true>(p, S, ws, ws_bytes, s);
    return splitk_launch<float, true, true>(p, S, ws, ws_bytes, s);
  }
  if (!akm && !bkm) return splitk_launch<bf16, false, false>(p, S, ws, ws_bytes, s);
  if (!akm && bkm) return splitk_launch<bf16, false, true>(p, S, ws, ws_bytes, s);
  return splitk_launch<bf16, true, true>(p, S, ws, ws_bytes, s);
}

int gemm_dma_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s) {
  return out_f32 ? dma_out<float>(p, batch, akm, bkm, s) : dma_out<bf16>(p, batch, akm, bkm, s);
}
