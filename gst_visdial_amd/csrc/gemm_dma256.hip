// 256x256x32 bf16 MFMA GEMM tile, 8 waves (each 128x64), LDS-DMA ring of NS256 stages (3 since round 3; 5 = the whole 160 KB LDS
// of a CU in rounds 1-2).
//
// Measured on MI355X (profiles/r01_pmc_gemm_l2_fetch.json): with 128x128x64 tiles the main loop is bound by the rate
// at which a CU can fill LDS (~35-50 GB/s per CU, L2 hit 70-82 %), i.e. by bytes staged per FLOP (1 B / 64 FLOP).
// This tile stages the same 32 KB per K-step but does 4x the MFMA work per step over half the K depth:
// 1 B / 128 FLOP.  Same ring / counted-vmcnt / one-barrier structure and the same zero-page tails as gemm_dma.hip.
//
// LDS images for a 32-deep K step:
//   row-major operand  : 256 rows x 64 B; 16-byte slot s of row r lives at slot s ^ (2*((r>>3)&1))   (conflict-free
//                        ds_read_b128 for the 16x16x32 fragment pattern; found by exhaustive search over the b128 lane groups)
//   k-major operand    : 32 k-rows x 512 B; 32-byte block b of k-row r lives at block b ^ ((r&3) | ((r>>3)&1)<<2)
#include "gemm_common.h"
#include <stdlib.h>

static __device__ __attribute__((aligned(256))) char g_zero_page256[256];
static constexpr int g_strip_w = 8;      // column-strip width of the tile order (measured 2 / 4 / 8: 31.3 / 30.7 / 30.5 us at 4096x3072x768)
// Ring depth.  Rounds 1-2 used 5 stages (the whole 160 KB LDS of a CU).  Round 3 measured the K-step slope at 5 / 4 / 3 stages
// (tools/nt_study.py on three builds): 0.669 / 0.686 / 0.684 us (producer / consumer form), 0.775 / 0.736 / 0.778 us (8-wave form)
// -- the K loop is bound by the RATE at which a CU takes operands in, not by bytes in flight -- while every stage less takes
// ~0.8 us off the fill in front of the first MFMA: 29.0 -> 27.8 us at 4096x3072x768, step +0.8 % (1190 vs 1181 rounds/s, three
// A/B pairs).  The LDS allocation is the larger of the ring and the epilogue's parking space.
constexpr int NS256 = 3;
// Timing ablations and in-kernel clock stamps exist only in the DIAGNOSTIC build of this file (-DGSTVD_DIAG ->
// lib/libgstvd_hip_diag.so, `make diag`; loaded by tools/ only, never by gst_visdial_amd/_lib.py): several of them compute
// wrong results on purpose, and no environment variable may be able to make the product library do that.
#ifdef GSTVD_DIAG
constexpr bool kDiag = true;
// shader-clock and 100 MHz wall-clock stamps around the K loop of the first 512 workgroups (ST = 3), written to a buffer of their
// own -- in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6);
// gstvd_debug_gemm_clock() copies them out
static __device__ unsigned long long g_clk256[512 * 4];
#else
constexpr bool kDiag = false;
#endif


DEVFN int opaque_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }
DEVFN int rm32_off(int row, int slot) { return row * 64 + ((slot ^ (((row >> 3) & 1) << 1)) << 4); }
template <int BX> DEVFN int km32_off(int krow, int col) {
  constexpr int NB = BX / 16;
  const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * (BX * 2) + ((((col >> 4) ^ f) & (NB - 1)) << 5) + ((col & 15) << 1);
}

template <int BX, bool KM> DEVFN bf16x8 frag32(const char* img, int xb, int lane) {
  const int g = lane >> 4, li = lane & 15;
  if (!KM) {
    return *(const bf16x8*)(img + rm32_off(xb + li, g));
  } else {
    const int kr = 8 * g + (li >> 2), col = xb + 4 * (lane & 3);
    s16x4 lo = lds_tr16(img + km32_off<BX>(kr, col));
    s16x4 hi = lds_tr16(img + km32_off<BX>(kr + 4, col));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int BX, bool KM, int NP, int NT>
struct Dma32 {
  const char* ptr[NP];
  int kofs[NP];
  bool okx[NP];
  int64_t kstep;
  DEVFN void init(const char* g, int64_t ld, int64_t x0, int64_t X, int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int u = i * NT + tid;
      int64_t x;
      if (!KM) {
        const int r = u >> 2, ls = (u & 3) ^ (((r >> 3) & 1) << 1);
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
    kstep = KM ? 32 * ld * 2 : 64;
  }
  DEVFN void issue(int64_t kt, int64_t K, char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool ok = okx[i] && (kt * 32 + kofs[i] < K);
      const char* src = ok ? ptr[i] + kt * kstep : (const char*)g_zero_page256;
      glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(img + (i * NT + wave * 64) * 16)));
    }
  }
};

// NIU = 16-column accumulator tiles per wave actually used (4: the full 256-wide tile; 3: a 192-wide tile inside the same
// 256-wide LDS image -- N = 3072 then gives 16 x 16 = 256 tiles, one per CU, instead of 192 tiles on 256 CUs).
// ST = K-step schedule: 0 the compiler's own order (DMA issue, then fragment reads two A fragments at a time in front of the
// MFMAs that use them); 4 all twelve fragment reads first, then the DMA issue under their latency (diagnostic build, GSTVD_DIAG_ST=4; correct
// results, measured -4 % per step in isolation, nothing inside the step).  Staggered-issue schedules (round 2's ST = 1 / 2 / 5)
// measured no gain and are gone (profiles/r02_gemm_kloop_study.txt).
// PF < 0 (except -5, the ping-pong tile) and ST = 3 are timing-only ablations of the diagnostic build (kDiag).
template <typename OT, bool AKM, bool BKM, int PF, int NIU = 4, int ST = 0>
DEVFN void dma_tile256(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NS = NS256, NT = 512;
  constexpr int WTM = BM / WM, WTN = NIU * 16, MI = WTM / 16, NI = NIU, BNU = WN * WTN;   // 8 x NIU accumulator tiles per wave
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int NPA = A_BYTES / (NT * 16), NPB = B_BYTES / (NT * 16), LPS = NPA + NPB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // tiles are numbered in column strips g_strip_w n-tiles wide, row-major inside a strip: the ~32 tiles an XCD runs
  // concurrently form a compact block that shares A rows and B columns in its L2
  const int ntm = nwg / ntn;
  const int SWD = g_strip_w;
  const int strip = wg / (SWD * ntm), sw = (ntn - strip * SWD) < SWD ? (ntn - strip * SWD) : SWD;
  const int within = wg - strip * SWD * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * SWD + within % sw) * BNU;

  Dma32<BM, AKM, NPA, NT> ua;
  Dma32<BN, BKM, NPB, NT> ub;
  ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, tid);
  ub.init(p.B + z * p.sB * 2, p.ldb, n0, (n0 + BNU < p.N) ? n0 + BNU : p.N, tid);     // columns past the tile read the zero page

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t nkt = (p.K + 31) / 32;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    ua.issue(s, p.K, smem + s * STAGE, wave);
    ub.issue(s, p.K, smem + s * STAGE + A_BYTES, wave);
  }
  int slot = 0, fill = NS - 1;
  static_assert(kDiag || (PF == 0 && ST == 0), "schedule variants and timing ablations belong to the -DGSTVD_DIAG build");
#ifdef GSTVD_DIAG
  unsigned long long clk0 = 0, rt0 = 0;
  if (ST == 3) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
  for (int64_t t = 0; t < nkt; ++t) {
    if (PF != -7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // diagnostic build only -- PF = -1 / -2 / -6 / -7 / -8 are timing-only ablations (wrong results): -1 = zero-page DMAs
    // inside the loop, -7 = no DMA instruction at all, -2 = no LDS reads / MFMA, -8 = MFMAs on stale registers, -6 = no epilogue
    if (ST == 4 && PF != -2) {
      // fragment reads first, all twelve of them (48 VGPRs): the compiler's own schedule reads two A fragments at a time right
      // before the eight MFMAs that use them, which leaves the matrix pipe waiting on LDS latency eight times per step
      // (in-kernel stamps: 1617 cycles per step for 1024 cycles of MFMA work with the DMA switched off).  Then the DMA issue,
      // under the reads' latency; then 32 MFMAs behind counted lgkmcnt waits.
      const char* cA = smem + slot * STAGE;
      const char* cB = cA + A_BYTES;
      bf16x8 fb[NI], fa[MI];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = frag32<BN, BKM>(cB, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = frag32<BM, AKM>(cA, wm * WTM + i * 16, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (PF == -1) {
        ua.issue(1 << 20, p.K, smem + fill * STAGE, wave);
        ub.issue(1 << 20, p.K, smem + fill * STAGE + A_BYTES, wave);
      } else {
        ua.issue(t + NS - 1, p.K, smem + fill * STAGE, wave);
        ub.issue(t + NS - 1, p.K, smem + fill * STAGE + A_BYTES, wave);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa[i], acc[i][j]);
      slot = (slot + 1 == NS) ? 0 : slot + 1;
      fill = (fill + 1 == NS) ? 0 : fill + 1;
      continue;
    }
    if (PF == -7) {
    } else if (PF == -1) {
      ua.issue(1 << 20, p.K, smem + fill * STAGE, wave);      // zero-page DMAs keep the vmcnt bookkeeping identical
      ub.issue(1 << 20, p.K, smem + fill * STAGE + A_BYTES, wave);
    } else {
      ua.issue(t + NS - 1, p.K, smem + fill * STAGE, wave);
      ub.issue(t + NS - 1, p.K, smem + fill * STAGE + A_BYTES, wave);
    }
    const char* cA = smem + slot * STAGE;
    const char* cB = cA + A_BYTES;
    if (PF == -8) {       // timing-only: real DMA, 32 MFMAs per wave on registers that are never reloaded (no LDS fragment reads)
      bf16x8 fx = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)lane, 1u, 2u, (unsigned)t});
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fx, fx, acc[i][j]);
    } else if (PF != -2) {
      bf16x8 fb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = frag32<BN, BKM>(cB, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const bf16x8 fa = frag32<BM, AKM>(cA, wm * WTM + i * 16, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa, acc[i][j]);
      }
    }
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    fill = (fill + 1 == NS) ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GSTVD_DIAG
  if (ST == 3 && tid == 0 && wg < 512) {
    g_clk256[wg * 4 + 0] = __builtin_amdgcn_s_memtime() - clk0;
    g_clk256[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    g_clk256[wg * 4 + 2] = (unsigned long long)nkt;
  }
#endif

  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  if (PF == -6) {      // timing-only ablation: no epilogue (one dependent store keeps the accumulators alive)
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) t += acc[i][j];
    if (t[0] + t[1] + t[2] + t[3] == 12345.678f) ((float*)p.C)[0] = t[0];
    return;
  }
  if constexpr (sizeof(OT) == 2) {
    if (epilogue_rows_ok(p)) {
      __builtin_amdgcn_s_barrier();            // every wave is done reading the ring: its LDS is free for the row-wise epilogue
      gemm_epilogue_rows<MI, NI, 4>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, 4>(), lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<bf16, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li, n0 + wn * WTN + j * 16 + 4 * g);
}

// Ping-pong schedule of the same tile: the two wave groups (wm = 0 / 1, one wave of each per SIMD) run one slot apart, so
// in every slot one group issues 16 MFMAs from registers while the other reads its next fragments from LDS and issues its
// share of the LDS-DMA.  Per 32-deep stage and group: L0 (A pieces, 4 B + 4 A fragments) | M0 (rows 0-3) | L1 (B pieces,
// 4 A fragments, counted vmcnt for stage t+1) | M1 (rows 4-7); every slot ends in a workgroup barrier.
// Ordering: a stage is read from the slot after the barrier that follows every wave's vmcnt wait for it; a ring slot is
// refilled only after the barrier that follows the lgkmcnt(0) retiring its last fragment read.
template <typename OT, bool AKM, bool BKM>
DEVFN void pp_tile256(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem) {
  constexpr int BM = 256, BN = 256, WN = 4, NS = NS256, NT = 512;
  constexpr int WTM = 128, WTN = 64, NI = 4, HI = 4;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int NPA = A_BYTES / (NT * 16), NPB = B_BYTES / (NT * 16), LPS = NPA + NPB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntm = nwg / ntn;
  const int strip = wg / (2 * ntm), sw = (ntn - strip * 2) < 2 ? (ntn - strip * 2) : 2;
  const int within = wg - strip * 2 * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * 2 + within % sw) * BN;

  Dma32<BM, AKM, NPA, NT> ua;
  Dma32<BN, BKM, NPB, NT> ub;
  ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, tid);
  ub.init(p.B + z * p.sB * 2, p.ldb, n0, p.N, tid);

  f32x4 acc[2 * HI][NI];
#pragma unroll
  for (int i = 0; i < 2 * HI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t nkt = (p.K + 31) / 32;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    ua.issue(s, p.K, smem + s * STAGE, wave);
    ub.issue(s, p.K, smem + s * STAGE + A_BYTES, wave);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one slot behind group 0
  __builtin_amdgcn_sched_barrier(0);

  int slot = 0, fill = NS - 1;
  for (int64_t t = 0; t < nkt; ++t) {
    const char* cA = smem + slot * STAGE;
    const char* cB = cA + A_BYTES;
    bf16x8 fb[NI], fa0[HI], fa1[HI];
    // ---- L0
    ua.issue(t + NS - 1, p.K, smem + fill * STAGE, wave);
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[j] = frag32<BN, BKM>(cB, wn * WTN + j * 16, lane);
#pragma unroll
    for (int i = 0; i < HI; ++i) fa0[i] = frag32<BM, AKM>(cA, wm * WTM + i * 16, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M0
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa0[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- L1
    ub.issue(t + NS - 1, p.K, smem + fill * STAGE + A_BYTES, wave);
#pragma unroll
    for (int i = 0; i < HI; ++i) fa1[i] = frag32<BM, AKM>(cA, wm * WTM + (HI + i) * 16, lane);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");     // own pieces of stage t+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M1
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[HI + i][j] = mfma_bf16_k32(fb[j], fa1[i], acc[HI + i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    fill = (fill + 1 == NS) ? 0 : fill + 1;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  if constexpr (sizeof(OT) == 2) {
    if (epilogue_rows_ok(p)) {
      __builtin_amdgcn_s_barrier();
      gemm_epilogue_rows<2 * HI, NI, 4>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, 4>(), lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 2 * HI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<bf16, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li, n0 + wn * WTN + j * 16 + 4 * g);
}

// ---- weight-gradient tile whose epilogue IS the optimizer step (gstvd_gemm_grouped_adamw) --------------------------------------
// The consumer waves hold 128 accumulator registers each and have nothing to spare for a stream of optimizer state (every
// in-consumer form of this epilogue spilled 100+ registers per lane); the four PRODUCER waves are idle after the K-loop and own
// nearly all of their 168 registers.  So: the consumers park their accumulators -- 4 x NI tiles = 64 rows at a time -- in their
// slices of the idle ring, exactly as gemm_epilogue_rows does, and after a barrier each producer wave runs AdamW on the parked
// rows of two consumer waves: a lane owns 4 consecutive columns of one row (16 lanes = one 256-byte row segment, whole lines per
// instruction), 16-byte loads of param / m / v, the update of common.h, 16-byte stores of param / m / v and 8 bytes of bf16
// shadow.  The state rows are fetched D row groups ahead of their use, IN PROGRAM ORDER in front of the stores of the groups in
// between (the compiler may not move a load across a store it cannot tell apart from it): 3 D 16-byte loads in flight per lane.
// Rows / columns outside the matrix: loads are clamped onto a valid element (no branch around a load), stores are masked.
// Addresses: wave-uniform 64-bit bases + one 32-bit lane offset per row group.
#ifndef ADAMW_PARK_DEPTH
#define ADAMW_PARK_DEPTH 5
#endif
template <int NI, int ROWS, int D>
DEVFN void adamw_rows_pass(const gstvd_adamw_fuse_t& af, float alpha, float lr, float wd, float bc, const float* park, int64_t flat0,
                           int rows_left, int cols_left, unsigned ldc, int lane) {
  static_assert(NI == 4, "16 lanes x 4 columns = the wave tile's 64 columns");
  constexpr int S = epi_row_floats<NI>(), RPI = 4, R = ROWS / RPI;
  if (rows_left <= 0 || cols_left <= 0) return;           // wave-uniform: the rows lie below / beside the matrix
  const int rl = lane >> 4, c4 = (lane & 15) * 4;
  const bool n_ok = c4 < cols_left;                       // N % 4 == 0: a 4-column piece is all in or all out
  float* Pt = af.param + flat0; float* Mt = af.m + flat0; float* Vt = af.v + flat0;
  bf16* St = (bf16*)af.shadow_bf16 + flat0;
  float* Gt = const_cast<float*>(af.grad_base) + flat0;
  const unsigned cofs = n_ok ? (unsigned)c4 : 0u;
  const int last = rows_left - 1;
  f32x4 bm[D + 1], bv[D + 1], bp[D + 1];
  auto fetch = [&](int rr, int b) {
    const int row = rr * RPI + rl;
    const unsigned o = (unsigned)(row < last ? row : last) * ldc + cofs;
    bm[b] = __builtin_nontemporal_load((const f32x4*)(Mt + o)); bv[b] = __builtin_nontemporal_load((const f32x4*)(Vt + o));
    bp[b] = __builtin_nontemporal_load((const f32x4*)(Pt + o));
  };
#pragma unroll
  for (int k = 0; k < D; ++k) fetch(k, k);
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    if (rr + D < R) fetch(rr + D, (rr + D) % (D + 1));
    const int b = rr % (D + 1);
    const int row = rr * RPI + rl;
    f32x4 g4 = *(const f32x4*)(park + row * S + c4);
    g4 = g4 * alpha;                                     // what the plain epilogue stores as dW
    f32x4 p4 = bp[b], m4 = bm[b], v4 = bv[b];
    adamw_update4(p4, m4, v4, g4, af.grad_scale, lr, wd, bc, af.beta1, af.beta2, af.eps);
    if (row < rows_left && n_ok) {
      const unsigned o = (unsigned)row * ldc + (unsigned)c4;
      if (af.write_grad) st4(Gt + o, g4);
      __builtin_nontemporal_store(m4, (f32x4*)(Mt + o));
      __builtin_nontemporal_store(v4, (f32x4*)(Vt + o));
      __builtin_nontemporal_store(p4, (f32x4*)(Pt + o));
      if (af.shadow_bf16) st4(St + o, p4);
    }
  }
}

// Producer / consumer form of the same tile (the default; GSTVD_GEMM_PC=0 switches it off): 12 waves.  Waves 0-7 are the MFMA consumers of dma_tile256 (128x64 each)
// and never touch the memory pipeline; waves 8-11 are LDS-DMA producers (8 pieces of 1 KB per wave and stage) and never touch
// the matrix pipe.  One s_barrier per K-step still orders everything: before barrier t every producer has waited for its pieces
// of stage t (counted vmcnt) and every consumer has retired its fragment reads of slot t-1, so after it the consumers read slot t
// and the producers refill slot t-1 with stage t+NS-1.  Why: in the 8-wave kernel every wave spends ~260 cycles of each K-step
// issuing DMA and both waves of a SIMD do so at the same time (section 5 of DESIGN.md: 1358 cycles per step without any DMA
// instruction, 1725 with); here the consumers' step IS that DMA-free loop and the DMA issue runs beside it on a third wave of
// the SIMD.  Needs <= 168 VGPRs (three waves per SIMD).
template <typename OT, bool AKM, bool BKM, int NIU = 4, bool CS = false, bool ADAM = false>
DEVFN void pc_tile256(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem, const gstvd_adamw_fuse_t af = gstvd_adamw_fuse_t{}) {
  constexpr int BM = 256, BN = 256, WN = 4, NS = NS256, NTP = 256;
  constexpr int WTM = 128, WTN = NIU * 16, MI = 8, NI = NIU, BNU = WN * WTN;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int NPA = A_BYTES / (NTP * 16), NPB = B_BYTES / (NTP * 16), LPS = NPA + NPB;      // 4 + 4 pieces per producer wave
  static_assert((NS - 2) * LPS <= 63, "vmcnt immediate out of range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 8;
  const int ntm = nwg / ntn;
  const int SWD = g_strip_w;
  const int strip = wg / (SWD * ntm), sw = (ntn - strip * SWD) < SWD ? (ntn - strip * SWD) : SWD;
  const int within = wg - strip * SWD * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * SWD + within % sw) * BNU;
  const int64_t nkt = (p.K + 31) / 32;

  if (producer) {
    const int ptid = tid - 512, pw = wave - 8;
    Dma32<BM, AKM, NPA, NTP> ua;
    Dma32<BN, BKM, NPB, NTP> ub;
    ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, ptid);
    ub.init(p.B + z * p.sB * 2, p.ldb, n0, (n0 + BNU < p.N) ? n0 + BNU : p.N, ptid);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
      ua.issue(s, p.K, smem + s * STAGE, pw);
      ub.issue(s, p.K, smem + s * STAGE + A_BYTES, pw);
    }
    int fill = NS - 1, slot = 0;
    // CS (grouped weight gradients, k-major A = dY): the producers of the tiles in column block 0 also sum the A image over k
    // while it sits in the ring -- bias[m] = sum over the batch rows of dY[row][m], the bias gradient -- instead of a separate
    // pass over dY (colsum_slab_batched: 0.25 ms per step, a 0.7 GB re-read).  Lane = (row group rg = lane >> 3, column group
    // cg = 8 * wave + (lane & 7)): thread reads the 16-byte unit of columns 8cg..8cg+7 in k-rows 4rg..4rg+3 of every stage;
    // the eight row groups of a column group sit in one wave, so the final reduction is three shuffles, no LDS, no barrier.
    const bool do_cs = CS && AKM && (p.epi & GSTVD_EPI_COLSUM) && n0 == 0 && p.bias != nullptr;
    const int rg = lane >> 3, cg = pw * 8 + (lane & 7);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t t = 0; t < nkt; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ua.issue(t + NS - 1, p.K, smem + fill * STAGE, pw);
      ub.issue(t + NS - 1, p.K, smem + fill * STAGE + A_BYTES, pw);
      if (CS && do_cs) {
        const char* cA = smem + slot * STAGE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16x8 v = *(const bf16x8*)(cA + km32_off<BM>(rg * 4 + r, cg * 8));
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += (float)v[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before this wave releases the slot at the next barrier
      }
      fill = (fill + 1 == NS) ? 0 : fill + 1;
      slot = (slot + 1 == NS) ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's trailing (zero page) pieces
    __builtin_amdgcn_s_barrier();                          // pairs with the consumers' barrier in front of the epilogue
    if (CS && do_cs) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        cs[e] = v;
      }
      if (rg == 0) {
        const bool accb = (p.epi & GSTVD_EPI_COLSUM_ACC) != 0;
        float* out = const_cast<float*>(p.bias);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int64_t m = m0 + cg * 8 + e;
          if (m < p.M) out[m] = accb ? out[m] + cs[e] : cs[e];
        }
      }
    }
    if constexpr (ADAM) {
      if (p.epi & GSTVD_EPI_ADAMW) {
        // the tile's optimizer constants: uniform -> scalar loads.  addend = this weight's (lr, wd) pair in the device table
        const float* hp2 = (const float*)p.addend;
        const float ad_lr = hp2[0], ad_wd = hp2[1];
        const float ad_bc = adamw_bias_correction(af.beta1, af.beta2, af.step[0]);
        constexpr int HB = MI / 2, ROWS = HB * 16;
        const int64_t cflat = (const float*)p.C - af.grad_base;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          __builtin_amdgcn_s_barrier();                    // the consumers have parked half h (they waited lgkmcnt(0) before arriving)
          asm volatile("" ::: "memory");                   // no park read may be scheduled above the barrier
          if (ad_lr != 0.f) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {                  // this producer wave serves consumer waves 2 pw and 2 pw + 1
              const int cw = pw * 2 + c, cwm = cw / WN, cwn = cw % WN;
              const int64_t mw = m0 + cwm * WTM + h * ROWS, nw = n0 + cwn * WTN;
              const int rows_left = (int)(p.M - mw < ROWS ? p.M - mw : ROWS), cols_left = (int)(p.N - nw < WTN ? p.N - nw : WTN);
              adamw_rows_pass<NI, ROWS, ADAMW_PARK_DEPTH>(af, p.alpha, ad_lr, ad_wd, ad_bc, (const float*)(smem + cw * epi_wave_bytes<NI, HB>()),
                                           cflat + mw * p.ldc + nw, rows_left, cols_left, (unsigned)p.ldc, lane);
            }
          }
          if (h == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every park read of this wave has returned
            __builtin_amdgcn_s_barrier();                  // the parks may be overwritten
          }
        }
      }
    }
    return;
  }

  const int wm = wave / WN, wn = wave % WN;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int slot = 0;
  for (int64_t t = 0; t < nkt; ++t) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* cA = smem + slot * STAGE;
    const char* cB = cA + A_BYTES;
    bf16x8 fb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[j] = frag32<BN, BKM>(cB, wn * WTN + j * 16, lane);
    if constexpr (NIU <= 3) {          // 96 accumulator registers: room for all eight A fragments up front (no spill at 168 VGPRs)
      bf16x8 fa[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = frag32<BM, AKM>(cA, wm * WTM + i * 16, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa[i], acc[i][j]);
    } else {                           // 128 accumulator registers: A fragments two at a time, like dma_tile256
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const bf16x8 fa = frag32<BM, AKM>(cA, wm * WTM + i * 16, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa, acc[i][j]);
      }
    }
    slot = (slot + 1 == NS) ? 0 : slot + 1;
  }
  __builtin_amdgcn_s_barrier();       // every producer has drained its DMAs, every consumer is done with the ring: LDS is free
  // The epilogue's per-lane addresses are derived from an OPAQUE copy of the lane id: derived from `lane` the compiler computes
  // them in front of the K loop and, at the 168-VGPR cap of three waves per SIMD, spills them across it (8-28 B of scratch per
  // lane, which also makes the dispatch set up scratch memory) -- VERDICT r5 item 1b.
  const int lane_e = (sizeof(OT) == 4 || ADAM) ? opaque_lane(lane) : lane;     // (the row-wise bf16 epilogue is better off without it)
  const int g = lane_e >> 4, li = lane_e & 15;
  if constexpr (ADAM) {
    if (p.epi & GSTVD_EPI_ADAMW) {
      // park 64 rows, let the producer waves update them (adamw_rows_pass), twice
      constexpr int S = epi_row_floats<NI>(), HB = MI / 2;
      float* park = (float*)(smem + wave * epi_wave_bytes<NI, HB>());
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h) {
          __builtin_amdgcn_s_barrier();                    // the producers are done with the first half
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) *(f32x4*)(park + (i * 16 + li) * S + j * 16 + 4 * g) = acc[h * HB + i][j];
        // The park crosses waves (the producers read it), and s_barrier on gfx950 waits for no counter: the LDS writes must
        // have retired before this wave arrives, and the compiler may not move them across the barrier either.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // parked
        asm volatile("" ::: "memory");
      }
      return;
    }
  }
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  if constexpr (sizeof(OT) == 2) {
    if (epilogue_rows_ok(p)) {
      gemm_epilogue_rows<MI, NI, 4, (NIU >= 4)>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, 4>(), lane_e);
      return;
    }
  }
  // (tile-wise fallback for unaligned operands: fresh opaque lane values per row block keep its 64-bit row / column indices from
  // being computed up front and spilled at the 168-VGPR cap)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int li_i = opaque_lane(li), g_i = opaque_lane(g);
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<bf16, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li_i, n0 + wn * WTN + j * 16 + 4 * g_i);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row-major A and B (the forward Linears with N >= 2304: QKV, FFN-up, cross-K/V, LM head), FULL-LINE staging (round 4).
// The 32-deep stages above fetch 64 bytes of every operand row per K-step: half of a 128-byte line, whose other half is fetched
// again one step later -- by then the 32 KB L1 has long turned over.  The LDS-DMA-only loop of such operands runs at 0.57 us per
// 32 KB against 0.46 us for k-major operands, whose rows are whole lines (round 2), and the producer / consumer K-step is
// producer bound.  Here an operand is filled 64 deep -- 128 bytes = one line per row and fill, 8 rows per wave instruction --
// and the fills ALTERNATE: an odd 32-deep step issues A(j) (K range [64j, 64j+64), used in steps 2j, 2j+1; slot j % 3), an even
// step B(j) (slot j % 2): still 32 KB of LDS-DMA per step, every byte of a fetched line used.  A(j) is issued in step 2j - 3 and
// B(j) in step 2j - 2, i.e. three resp. two steps ahead of their first use; a slot is refilled two resp. one barrier after its
// last read.  LDS: 3 x 32 KB + 2 x (32 | 24) KB.
//   image of a fill: row r at 128 r, 16-byte slot s of the row at slot s ^ ((r >> 1) & 7) -- conflict-free for the b128 lane
//   groups with the 16x16x32 fragment pattern (checked exhaustively); the swizzle is applied to the
//   per-lane SOURCE address, the LDS side of the DMA stays lane-linear.
// Requires K % 64 == 0 (every K of the step); other shapes take pc_tile256.
// Measured (round 4, tools/r04_nt64_check.sh): K-step slope 0.702 -> 0.630 us at 4096 x 3072, 0.552 -> 0.498 at 4096 x 2304 (vendor
// BLAS 0.495 / 0.399), fixed part +0.7 .. 1.0 us (three fills in front of the first MFMA); 4096x3072x768 30.4 -> 29.6 us.
// An L2 PREFETCH of the lines of fill j + 2 .. 6 (consumer waves 0 / 1 touching this workgroup's share of the rows it shares with
// the 8 resp. 4 workgroups of its XCD, one dword per line) changed nothing: slope 0.647 / 0.644 / 0.641 / 0.638 us for 0 / 2 / 4 / 6
// fills ahead -- the ring is not waiting on beyond-L2 latency (removed again, with its sweep script).
DEVFN int rm64_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int ROWS, int NP>            // NP = 1 KB pieces per producer wave and fill (ROWS / 32)
struct Dma64 {
  const char* ptr[NP];
  bool okx[NP];
  DEVFN void init(const char* g, int64_t ld, int64_t x0, int64_t X, int ptid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int u = i * 256 + ptid, r = u >> 3, ls = (u & 7) ^ ((r >> 1) & 7);
      const int64_t x = x0 + r;
      okx[i] = x < X;
      ptr[i] = g + (x * ld + ls * 8) * 2;
    }
  }
  DEVFN void issue(int64_t j, int64_t J, char* img, int pw) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const char* src = (okx[i] && j < J) ? ptr[i] + j * 128 : (const char*)g_zero_page256;
      glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(img + (i * 256 + pw * 64) * 16)));
    }
  }
};

template <typename OT, int NIU>
DEVFN void pc_tile256_nt64(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem) {
  constexpr int BM = 256, WN = 4, WTM = 128, WTN = NIU * 16, MI = 8, NI = NIU, BNU = WN * WTN;
  constexpr int A_SLOT = BM * 128, B_SLOT = BNU * 128, B_BASE = 3 * A_SLOT;
  constexpr int NPA = BM / 32, NPB = BNU / 32;
  static_assert(NPA + NPB <= 63, "vmcnt immediate out of range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntm = nwg / ntn;
  const int SWD = g_strip_w;
  const int strip = wg / (SWD * ntm), sw = (ntn - strip * SWD) < SWD ? (ntn - strip * SWD) : SWD;
  const int within = wg - strip * SWD * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * SWD + within % sw) * BNU;
  const int64_t nkt = p.K / 32, J = p.K / 64;

  if (wave >= 8) {                     // ---- producers
    const int ptid = tid - 512, pw = wave - 8;
    Dma64<BM, NPA> ua;
    Dma64<BNU, NPB> ub;
    ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, ptid);
    ub.init(p.B + z * p.sB * 2, p.ldb, n0, p.N, ptid);
    ua.issue(0, J, smem, pw);
    ub.issue(0, J, smem + B_BASE, pw);
    ua.issue(1, J, smem + A_SLOT, pw);
    int sa = 2, sb = 1;                // slots of the next A / B fill
    int64_t ja = 2, jb = 1;
    for (int64_t t = 0; t < nkt; t += 2) {
      // even step t = 2j: A(j) and B(j) must have landed; the youngest fill in flight is A(j+1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ub.issue(jb, J, smem + B_BASE + sb * B_SLOT, pw);          // B(j+1): its slot held B(j-1), last read in step t - 1
      ++jb; sb ^= 1;
      // odd step: nothing new is needed (A(j), B(j) are in use); in flight: A(j+1), B(j+1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA + NPB) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ua.issue(ja, J, smem + sa * A_SLOT, pw);                   // A(j+2): its slot held A(j-1), last read in step t - 1
      ++ja; sa = (sa + 1 == 3) ? 0 : sa + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing (zero page) fills
    __builtin_amdgcn_s_barrier();                                 // pairs with the consumers' barrier in front of the epilogue
    return;
  }

  const int wm = wave / WN, wn = wave % WN;                        // ---- consumers
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int sa = 0, sb = 0;
  {
  const int g = lane >> 4, li = lane & 15;
  for (int64_t t = 0; t < nkt; t += 2) {
    const char* cA = smem + sa * A_SLOT;
    const char* cB = smem + B_BASE + sb * B_SLOT;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      bf16x8 fb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *(const bf16x8*)(cB + rm64_off(wn * WTN + j * 16 + li, h * 4 + g));
      if constexpr (NIU <= 3) {
        bf16x8 fa[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(cA + rm64_off(wm * WTM + i * 16 + li, h * 4 + g));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa[i], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const bf16x8 fa = *(const bf16x8*)(cA + rm64_off(wm * WTM + i * 16 + li, h * 4 + g));
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16_k32(fb[j], fa, acc[i][j]);
        }
      }
    }
    sa = (sa + 1 == 3) ? 0 : sa + 1;
    sb ^= 1;
  }
  }
  __builtin_amdgcn_s_barrier();        // every producer has drained its DMAs, every consumer is done with the ring: LDS is free
  const int lane_e = sizeof(OT) == 4 ? opaque_lane(lane) : lane;        // (see pc_tile256)
  const int g = lane_e >> 4, li = lane_e & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  if constexpr (sizeof(OT) == 2) {
    if (epilogue_rows_ok(p)) {
      gemm_epilogue_rows<MI, NI, 4, (NIU >= 4)>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, 4>(), lane_e);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int li_i = opaque_lane(li), g_i = opaque_lane(g);
#pragma unroll
    for (int j = 0; j < NI; ++j)
      gemm_epilogue_tile<bf16, OT>(p, dk, acc[i][j], z, m0 + wm * WTM + i * 16 + li_i, n0 + wn * WTN + j * 16 + 4 * g_i);
  }
}

DEVFN int xcd_remap256(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename OT, bool AKM, bool BKM, int PF, int NIU = 4, int ST = 0>
__global__ __launch_bounds__(512) void gemm_dma256_kernel(GemmP p, int ntn, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (PF == -5) pp_tile256<OT, AKM, BKM>(p, blockIdx.y, xcd_remap256(blockIdx.x, nwg), ntn, nwg, smem);
  else dma_tile256<OT, AKM, BKM, PF, NIU, ST>(p, blockIdx.y, xcd_remap256(blockIdx.x, nwg), ntn, nwg, smem);
}

template <typename OT, bool AKM, bool BKM, int NIU = 4>
__global__ __launch_bounds__(768) void gemm_pc256_kernel(GemmP p, int ntn, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pc_tile256<OT, AKM, BKM, NIU>(p, blockIdx.y, xcd_remap256(blockIdx.x, nwg), ntn, nwg, smem);
}

template <typename OT, int NIU>
__global__ __launch_bounds__(768) void gemm_pc256_nt64_kernel(GemmP p, int ntn, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pc_tile256_nt64<OT, NIU>(p, blockIdx.y, xcd_remap256(blockIdx.x, nwg), ntn, nwg, smem);
}

// Which tile of the table a workgroup of a grouped launch runs.
//   bmap != nullptr (the host's order, ops.GemmGroup: gstvd_gemm_grouped*'s `block_map_dev`): block b runs tile bmap[b], or
//     nothing when that is negative (padding of the shorter XCD queues).  Blocks are dealt to the XCDs round-robin (observed, speed
//     only), so entries b = x, x + 8, x + 16, ... are XCD x's queue: the host puts whole problems -- tiles that share operand
//     panels and run for the same time -- back to back into ONE queue, so that the ~32 tiles an XCD runs at a time stream the
//     same few panels through its L2 in step.
//   bmap == nullptr: XCD x walks chunks x, x+8, x+16, ... of 2^chs consecutive tiles (rounds 1-4).
// A map entry's tile id is its bits 0-20; the bits above are ignored (reserved).  (Round 5 tried a START BARRIER in them -- position of
// the tile in its unit and the unit's size, the workgroups of a unit waiting for each other before their K loops: 7.5 % fewer operand
// fetches, 2.3 % slower, and the call cost the fused kernel 256 B/lane of scratch; removed, profiles/r05_group_sync_ab.txt.)
constexpr int GROUP_TILE_MASK = 0x1fffff;
DEVFN int grouped_tile_id(const int* bmap, int total, int chs) {
  const int bid = blockIdx.x;
  if (bmap) return bmap[bid];
  const int full = total - total % (8 << chs);
  return bid < full ? (((bid >> 3) >> chs) * 8 + (bid & 7)) * (1 << chs) + ((bid >> 3) & ((1 << chs) - 1)) : bid;
}

template <typename OT, bool AKM, bool BKM, int PF, int ST = 0>
__global__ __launch_bounds__(512) void gemm_dma256_grouped_kernel(const gstvd_gemm_t* tab, const int* tile_off, int nprob, int total, int chs,
                                                                  const int* bmap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int entry = grouped_tile_id(bmap, total, chs);
  if (entry < 0) return;
  const int gid = entry & GROUP_TILE_MASK;
  if (gid >= total) return;                  // a map entry that names no tile of the table (the host checks the map it builds; this keeps a bad one from writing)
  int lo = 0, hi = nprob - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_off[mid] <= gid) lo = mid; else hi = mid - 1; }
  const gstvd_gemm_t& g = tab[lo];
  GemmP p;
  p.A = (const char*)g.A; p.B = (const char*)g.B; p.C = (char*)g.C;
  p.bias = g.bias; p.addend = (const char*)g.addend; p.aux = (char*)g.aux;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldadd = g.ldadd; p.ldaux = g.ldaux;
  p.sA = p.sB = p.sC = p.sAdd = p.sAux = 0;
  p.epi = g.epilogue; p.alpha = g.alpha; p.p = g.dropout_p; p.site = g.site; p.rng = g.rng;
  const int ntn = (int)((g.N + 255) / 256), ntm = (int)((g.M + 255) / 256);
  dma_tile256<OT, AKM, BKM, PF, 4, ST>(p, 0, gid - tile_off[lo], ntn, ntm * ntn, smem);
}

constexpr int RING256 = NS256 * (256 + 256) * 64;
constexpr int PARK256 = 8 * epi_wave_bytes<4, 4>();        // eight waves park 4 x 4 accumulator tiles each (row-wise epilogue)
constexpr int LDS256 = RING256 > PARK256 ? RING256 : PARK256;

template <typename OT, bool AKM, bool BKM>
static int launch256(const GemmP& p, int64_t batch, int niu, hipStream_t s) {
  auto k0 = gemm_dma256_kernel<OT, AKM, BKM, 0>;
  auto k3 = gemm_dma256_kernel<OT, AKM, BKM, 0, 3>;
  static int attr_rc = ensure_lds(k0, LDS256) | ensure_lds(k3, LDS256);
  if (attr_rc) return attr_rc;
  // The K-step schedule variants that rounds 2-4 measured and rejected (ST = 4: all fragment reads first; the ping-pong tile;
  // the timing ablations) exist only in the diagnostic build: GSTVD_DIAG_ST, GSTVD_DIAG_PP, GSTVD_DIAG_ABLATE.
  int abl = 0, st = 0;
#ifdef GSTVD_DIAG
  auto p0 = gemm_dma256_kernel<OT, AKM, BKM, 0, 4, 4>;
  auto p3 = gemm_dma256_kernel<OT, AKM, BKM, 0, 3, 4>;
  auto ke = gemm_dma256_kernel<OT, AKM, BKM, -5>;
  auto c0 = gemm_dma256_kernel<OT, AKM, BKM, 0, 4, 3>;
  auto c1 = gemm_dma256_kernel<OT, AKM, BKM, -1, 4, 3>;
  auto c2 = gemm_dma256_kernel<OT, AKM, BKM, -2, 4, 3>;
  auto c7 = gemm_dma256_kernel<OT, AKM, BKM, -7, 4, 3>;
  auto c8 = gemm_dma256_kernel<OT, AKM, BKM, -8, 4, 3>;
  auto p1 = gemm_dma256_kernel<OT, AKM, BKM, -1, 4, 4>;
  auto ka = gemm_dma256_kernel<OT, AKM, BKM, -1>;
  auto kb = gemm_dma256_kernel<OT, AKM, BKM, -2>;
  auto kf = gemm_dma256_kernel<OT, AKM, BKM, -6>;
  static int diag_rc = ensure_lds(c0, LDS256) | ensure_lds(c1, LDS256) | ensure_lds(c2, LDS256) | ensure_lds(c7, LDS256) |
                       ensure_lds(c8, LDS256) | ensure_lds(p1, LDS256) | ensure_lds(ka, LDS256) | ensure_lds(kb, LDS256) |
                       ensure_lds(kf, LDS256) | ensure_lds(ke, LDS256) | ensure_lds(p0, LDS256) | ensure_lds(p3, LDS256);
  if (diag_rc) return diag_rc;
  static const int st_env = [] { const char* e = getenv("GSTVD_DIAG_ST"); return e ? atoi(e) : 0; }();
  static const int pp = [] { const char* e = getenv("GSTVD_DIAG_PP"); return e ? atoi(e) : 0; }();
  static const int diag_abl = [] { const char* e = getenv("GSTVD_DIAG_ABLATE"); return e ? atoi(e) : 0; }();
  abl = diag_abl ? diag_abl : (pp ? 5 : 0);
  st = (st_env == 3 || st_env == 4) ? st_env : 0;
#endif
  const int bnu = (niu == 3 && abl == 0) ? 192 : 256;
  const int ntm = (int)((p.M + 255) / 256), ntn = (int)((p.N + bnu - 1) / bnu);
  // producer / consumer form: measured -7 % per launch on single-round grids (4096x3072x768 33.8 -> 31.3 us, per K-step 0.82 ->
  // 0.76 us), +7 % on the 1368-tile cross-K/V projection (its extra ~1.5 us of fixed cost is paid 5.3 times): up to 2 rounds of tiles
  static const int pc = [] { const char* e = getenv("GSTVD_GEMM_PC"); return e ? atoi(e) : 1; }();
  if (pc && abl == 0 && st == 0 && (pc == 2 || (int64_t)ntm * ntn * batch <= 512)) {
    if constexpr (!AKM && !BKM) {
      // full-line (64-deep, alternating) staging of two row-major operands; GSTVD_GEMM_NT64=0 keeps the 32-deep stages (A/B)
      static const int nt64 = [] { const char* e = getenv("GSTVD_GEMM_NT64"); return e ? atoi(e) : 1; }();
      if (nt64 && p.K % 64 == 0 && p.K >= 128) {
        auto n4 = gemm_pc256_nt64_kernel<OT, 4>;
        auto n3 = gemm_pc256_nt64_kernel<OT, 3>;
        constexpr int L4 = 3 * 256 * 128 + 2 * 256 * 128, L3 = 3 * 256 * 128 + 2 * 192 * 128;
        constexpr int P4 = 8 * epi_wave_bytes<4, 4>(), P3 = 8 * epi_wave_bytes<3, 4>();
        constexpr int lds4 = L4 > P4 ? L4 : P4, lds3 = L3 > P3 ? L3 : P3;
        static_assert(lds4 <= 160 * 1024 && lds3 <= 160 * 1024, "LDS budget");
        static int nt_rc = ensure_lds(n4, lds4) | ensure_lds(n3, lds3);
        if (nt_rc) return nt_rc;
        GSTVD_LAUNCH(bnu == 192 ? n3 : n4, dim3((unsigned)(ntm * ntn), (unsigned)batch), dim3(768), bnu == 192 ? lds3 : lds4, s, p, ntn, ntm * ntn);
        GSTVD_LAUNCH_CHECK();
        return 0;
      }
    }
    auto c4 = gemm_pc256_kernel<OT, AKM, BKM, 4>;
    auto c3 = gemm_pc256_kernel<OT, AKM, BKM, 3>;
    static int pc_rc = ensure_lds(c4, LDS256) | ensure_lds(c3, LDS256);
    if (pc_rc) return pc_rc;
    GSTVD_LAUNCH(bnu == 192 ? c3 : c4, dim3((unsigned)(ntm * ntn), (unsigned)batch), dim3(768), LDS256, s, p, ntn, ntm * ntn);
    GSTVD_LAUNCH_CHECK();
    return 0;
  }
  auto kern = bnu == 192 ? k3 : k0;
#ifdef GSTVD_DIAG
  if (st == 4) kern = abl == 1 ? p1 : (bnu == 192 ? p3 : p0);
  else if (st == 3) kern = abl == 1 ? c1 : abl == 2 ? c2 : abl == 7 ? c7 : abl == 8 ? c8 : c0;
  else if (abl == 5) kern = ke;
  else if (abl == 1) kern = ka;
  else if (abl == 2) kern = kb;
  else if (abl == 6) kern = kf;
#endif
  GSTVD_LAUNCH(kern, dim3((unsigned)(ntm * ntn), (unsigned)batch), dim3(512), LDS256, s, p, ntn, ntm * ntn);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

template <typename OT>
static int launch256_layout(const GemmP& p, int64_t batch, int akm, int bkm, int niu, hipStream_t s) {
  if (!akm && !bkm) return launch256<OT, false, false>(p, batch, niu, s);
  if (!akm && bkm) return launch256<OT, false, true>(p, batch, niu, s);
  if (akm && bkm) return launch256<OT, true, true>(p, batch, niu, s);
  return GSTVD_E_UNSUPPORTED;
}

// Tile width: 256 columns, or 192 when that needs fewer rounds of 256 CUs for the work it does.  Cost model from the
// measured kernel: ~14 us fixed + 0.95 us per 32-deep K step for the full tile, ~0.9x that per step for the 192 one.
static int pick_niu(const GemmP& p, int64_t batch) {
  const double nkt = (double)((p.K + 31) / 32), ntm = (double)((p.M + 255) / 256);
  const double t4 = ntm * (double)((p.N + 255) / 256) * batch, t3 = ntm * (double)((p.N + 191) / 192) * batch;
  const double r4 = (double)(int64_t)((t4 + 255) / 256), r3 = (double)(int64_t)((t3 + 255) / 256);
  return r3 * (14.0 + 0.84 * nkt) < r4 * (14.0 + 0.95 * nkt) ? 3 : 4;     // measured 0.73 vs 0.82 us per step at K = 768
}

// Single-problem policy: the big tile only pays when its (4x smaller) grid still covers most of the chip.
int gemm_dma256_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s) {
  static const int min_tiles = [] { const char* e = getenv("GSTVD_GEMM256_MIN_TILES"); return e ? atoi(e) : 120; }();
  const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
  if (p.M < 256 || p.N < 256 || tiles < min_tiles) return GSTVD_E_UNSUPPORTED;
  const int niu = pick_niu(p, batch);
  return out_f32 ? launch256_layout<float>(p, batch, akm, bkm, niu, s) : launch256_layout<bf16>(p, batch, akm, bkm, niu, s);
}

template <typename OT, bool AKM, bool BKM>
__global__ __launch_bounds__(768) void gemm_pc256_grouped_kernel(const gstvd_gemm_t* tab, const int* tile_off, int nprob, int total, int chs,
                                                                 const int* bmap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int entry = grouped_tile_id(bmap, total, chs);
  if (entry < 0) return;
  const int gid = entry & GROUP_TILE_MASK;
  if (gid >= total) return;                  // a map entry that names no tile of the table (the host checks the map it builds; this keeps a bad one from writing)
  int lo = 0, hi = nprob - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_off[mid] <= gid) lo = mid; else hi = mid - 1; }
  const gstvd_gemm_t& g = tab[lo];
  GemmP p;
  p.A = (const char*)g.A; p.B = (const char*)g.B; p.C = (char*)g.C;
  p.bias = g.bias; p.addend = (const char*)g.addend; p.aux = (char*)g.aux;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldadd = g.ldadd; p.ldaux = g.ldaux;
  p.sA = p.sB = p.sC = p.sAdd = p.sAux = 0;
  p.epi = g.epilogue; p.alpha = g.alpha; p.p = g.dropout_p; p.site = g.site; p.rng = g.rng;
  const int ntn = (int)((g.N + 255) / 256), ntm = (int)((g.M + 255) / 256);
  pc_tile256<OT, AKM, BKM, 4, true>(p, 0, gid - tile_off[lo], ntn, ntm * ntn, smem);
}

__global__ __launch_bounds__(768) void gemm_pc256_grouped_adamw_kernel(const gstvd_gemm_t* tab, const int* tile_off, int nprob, int total, int chs,
                                                                       const int* bmap, gstvd_adamw_fuse_t af) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int entry = grouped_tile_id(bmap, total, chs);
  if (entry < 0) return;
  const int gid = entry & GROUP_TILE_MASK;
  if (gid >= total) return;                  // a map entry that names no tile of the table (the host checks the map it builds; this keeps a bad one from writing)
  int lo = 0, hi = nprob - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_off[mid] <= gid) lo = mid; else hi = mid - 1; }
  const gstvd_gemm_t& g = tab[lo];
  GemmP p;
  p.A = (const char*)g.A; p.B = (const char*)g.B; p.C = (char*)g.C;
  p.bias = g.bias; p.addend = (const char*)g.addend; p.aux = (char*)g.aux;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldadd = g.ldadd; p.ldaux = g.ldaux;
  p.sA = p.sB = p.sC = p.sAdd = p.sAux = 0;
  p.epi = g.epilogue; p.alpha = g.alpha; p.p = g.dropout_p; p.site = g.site; p.rng = g.rng;
  const int ntn = (int)((g.N + 255) / 256), ntm = (int)((g.M + 255) / 256);
  pc_tile256<float, true, true, 4, true, true>(p, 0, gid - tile_off[lo], ntn, ntm * ntn, smem, af);
}

template <typename OT, bool AKM, bool BKM>
static int grouped256(const gstvd_gemm_t* tab, const int* off, int nprob, int total, const int* bmap, int nblocks, hipStream_t s) {
  auto k0 = gemm_dma256_grouped_kernel<OT, AKM, BKM, 0>;
  static int attr_rc = ensure_lds(k0, LDS256);
  if (attr_rc) return attr_rc;
  // tiles per XCD chunk = 2^chs (default 8: measured best of 1..128 inside the step); GSTVD_GROUP_CHUNK_LOG2 overrides for tuning runs
  static const int chs = [] { const char* e = getenv("GSTVD_GROUP_CHUNK_LOG2"); const int v = e ? atoi(e) : 3; return v < 0 ? 0 : (v > 10 ? 10 : v); }();
  // weight gradients are long-K problems (K = rows of the batch): the producer / consumer tile gains ~0.08 us on every one of
  // their ~128 K-steps (18432x768x4688: 160 -> 143 us)
  static const int pc = [] { const char* e = getenv("GSTVD_GEMM_PC"); return e ? atoi(e) : 1; }();
  if (pc) {
    auto kp = gemm_pc256_grouped_kernel<OT, AKM, BKM>;
    static int pc_rc = ensure_lds(kp, LDS256);
    if (pc_rc) return pc_rc;
    GSTVD_LAUNCH(kp, dim3((unsigned)(bmap ? nblocks : total)), dim3(768), LDS256, s, tab, off, nprob, total, chs, bmap);
    GSTVD_LAUNCH_CHECK();
    return 0;
  }
  GSTVD_LAUNCH(k0, dim3((unsigned)(bmap ? nblocks : total)), dim3(512), LDS256, s, tab, off, nprob, total, chs, bmap);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_gemm_group_tile(void) { return 256; }

extern "C" int32_t gstvd_gemm_group_caps(void) {
  const char* e = getenv("GSTVD_GEMM_PC");
  return (e ? atoi(e) : 1) != 0 ? 1 : 0;
}

// symbol of the grouped kernel for a (dtype, layout) combination -- same plan-only mechanism as gstvd_gemm_kernel_name
extern "C" int gstvd_gemm_grouped_kernel_name(int32_t dtype_in, int32_t dtype_out, int32_t a_kmajor, int32_t b_kmajor, char* buf,
                                              int32_t buf_len) {
  if (!buf || buf_len <= 1) return GSTVD_E_NULL;
  const void* fn = nullptr;
  static gstvd_gemm_t dummy_tab;
  static int32_t dummy_off;
  gstvd_plan_capture = &fn;
  const int rc = gstvd_gemm_grouped(&dummy_tab, &dummy_off, 1, 1, dtype_in, dtype_out, a_kmajor, b_kmajor, nullptr, 0, nullptr);
  gstvd_plan_capture = nullptr;
  if (rc) return rc;
  const char* name = fn ? hipKernelNameRefByPtr(fn, nullptr) : nullptr;
  if (!name) return GSTVD_E_UNSUPPORTED;
  int i = 0;
  for (; name[i] && i < buf_len - 1; ++i) buf[i] = name[i];
  buf[i] = 0;
  return 0;
}

extern "C" int gstvd_gemm_grouped_adamw_kernel_name(char* buf, int32_t buf_len) {
  if (!buf || buf_len <= 1) return GSTVD_E_NULL;
  const char* name = hipKernelNameRefByPtr((const void*)gemm_pc256_grouped_adamw_kernel, nullptr);
  if (!name) return GSTVD_E_UNSUPPORTED;
  int i = 0;
  for (; name[i] && i < buf_len - 1; ++i) buf[i] = name[i];
  buf[i] = 0;
  return 0;
}

#ifdef GSTVD_DIAG
extern "C" int gstvd_debug_gemm_clock(uint64_t* out_host, int32_t n_words) {
  if (!out_host || n_words <= 0 || n_words > 512 * 4) return GSTVD_E_SHAPE;
  hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_clk256), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
  return e == hipSuccess ? 0 : (int)e;
}
#endif

extern "C" int gstvd_gemm_grouped_adamw(const gstvd_gemm_t* table_dev, const int32_t* tile_off_dev, int64_t nprob, int64_t total_tiles,
                                        const gstvd_adamw_fuse_t* f, const int32_t* block_map_dev, int64_t nblocks, gstvd_stream_t stream) {
  if (!table_dev || !tile_off_dev || !f) return GSTVD_E_NULL;
  if (!f->grad_base || !f->param || !f->m || !f->v || !f->step) return GSTVD_E_NULL;
  if (nprob <= 0 || total_tiles <= 0) return GSTVD_E_SHAPE;
  if (block_map_dev && (nblocks < total_tiles || total_tiles > GROUP_TILE_MASK)) return GSTVD_E_SHAPE;     // a map that cannot name every tile
  if (((uintptr_t)f->grad_base | (uintptr_t)f->param | (uintptr_t)f->m | (uintptr_t)f->v | (uintptr_t)f->shadow_bf16) & 15) return GSTVD_E_ALIGN;
  if (!(gstvd_gemm_group_caps() & 1)) return GSTVD_E_UNSUPPORTED;     // the producer / consumer tile is switched off (tuning runs)
  auto kp = gemm_pc256_grouped_adamw_kernel;
  static int rc = ensure_lds(kp, LDS256);
  if (rc) return rc;
  static const int chs = [] { const char* e = getenv("GSTVD_GROUP_CHUNK_LOG2"); const int v = e ? atoi(e) : 3; return v < 0 ? 0 : (v > 10 ? 10 : v); }();
  GSTVD_LAUNCH(kp, dim3((unsigned)(block_map_dev ? nblocks : total_tiles)), dim3(768), LDS256, (hipStream_t)stream, table_dev, tile_off_dev,
               (int)nprob, (int)total_tiles, chs, block_map_dev, *f);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gstvd_gemm_grouped(const gstvd_gemm_t* table_dev, const int32_t* tile_off_dev, int64_t nprob, int64_t total_tiles,
                                  int32_t dtype_in, int32_t dtype_out, int32_t a_kmajor, int32_t b_kmajor,
                                  const int32_t* block_map_dev, int64_t nblocks, gstvd_stream_t stream) {
  if (!table_dev || !tile_off_dev) return GSTVD_E_NULL;
  if (nprob <= 0 || total_tiles <= 0) return GSTVD_E_SHAPE;
  if (block_map_dev && (nblocks < total_tiles || total_tiles > GROUP_TILE_MASK)) return GSTVD_E_SHAPE;     // a map that cannot name every tile
  if (dtype_in != GSTVD_BF16) return GSTVD_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)nprob, t = (int)total_tiles, nb = (int)nblocks;
  const int* bm = block_map_dev;
  if (dtype_out == GSTVD_F32) {
    if (a_kmajor && b_kmajor) return grouped256<float, true, true>(table_dev, tile_off_dev, n, t, bm, nb, s);
    if (!a_kmajor && b_kmajor) return grouped256<float, false, true>(table_dev, tile_off_dev, n, t, bm, nb, s);
    if (!a_kmajor && !b_kmajor) return grouped256<float, false, false>(table_dev, tile_off_dev, n, t, bm, nb, s);
  } else if (dtype_out == GSTVD_BF16) {
    if (a_kmajor && b_kmajor) return grouped256<bf16, true, true>(table_dev, tile_off_dev, n, t, bm, nb, s);
    if (!a_kmajor && b_kmajor) return grouped256<bf16, false, true>(table_dev, tile_off_dev, n, t, bm, nb, s);
    if (!a_kmajor && !b_kmajor) return grouped256<bf16, false, false>(table_dev, tile_off_dev, n, t, bm, nb, s);
  }
  return GSTVD_E_UNSUPPORTED;
}
