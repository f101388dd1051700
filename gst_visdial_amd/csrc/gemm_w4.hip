// 256x256x32 bf16 MFMA tile on FOUR waves (each 128x128 = 64 accumulator tiles, 256 accumulator registers), LDS-DMA ring.
//
// Why a second 256-tile kernel (round 3).  The 8-wave tile of gemm_dma256.hip (each wave 128x64) reads per 32-deep K-step and
// wave 8 A + 4 B fragments for 32 MFMAs: 12 ds_read_b128 of 1 KB each = 96 KB of LDS reads per step and workgroup; at the LDS
// port's 128 B / clk that is 768 cycles, the ring's 32 KB of LDS-DMA writes another 256 -- together exactly the 1024 cycles the
// step's 256 MFMAs (64 per SIMD x 16 cycles) take: the LDS port is saturated, nothing may ever be late (measured: 1358 cycles
// per step without any DMA, 1725 with; the vendor BLAS runs the same problems with a 30 % shorter K loop --
// profiles/r03_gemm_vs_vendor_blas.txt, tools/nt_study.py: 0.49 vs 0.70 us per step at 4096 x 3072).  With 128x128 per wave a
// wave reads 8 + 8 fragments for 64 MFMAs: 64 KB per step and workgroup = 512 cycles, + 256 of DMA = 75 % of the MFMA time.
// One wave per SIMD then has to hide everything itself: the B fragments of step t+1 are read into a second register set while
// the 64 MFMAs of step t run (256 accumulator registers in the AGPR half of the file, 3 x 32 fragment registers + addressing
// in the VGPR half).
//
// Ring protocol (NS slots, stage s lives in slot s % NS): iteration t holds stage t's B fragments in registers (read during
// iteration t-1), reads stage t's A fragments and stage t+1's B fragments, and refills the slot of stage t-1 (every wave
// retires its reads of it -- lgkmcnt(0) -- before the barrier that opens iteration t) with stage t + NS - 1.
#include "gemm_common.h"
#include <stdlib.h>

static __device__ __attribute__((aligned(256))) char g_zero_page_w4[256];

DEVFN int w4_rm_off(int row, int slot) { return row * 64 + ((slot ^ (((row >> 3) & 1) << 1)) << 4); }
template <int BX> DEVFN int w4_km_off(int krow, int col) {
  constexpr int NB = BX / 16;
  const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * (BX * 2) + ((((col >> 4) ^ f) & (NB - 1)) << 5) + ((col & 15) << 1);
}
template <int BX, bool KM> DEVFN bf16x8 w4_frag(const char* img, int xb, int lane) {
  const int g = lane >> 4, li = lane & 15;
  if (!KM) {
    return *(const bf16x8*)(img + w4_rm_off(xb + li, g));
  } else {
    const int kr = 8 * g + (li >> 2), col = xb + 4 * (lane & 3);
    s16x4 lo = lds_tr16(img + w4_km_off<BX>(kr, col));
    s16x4 hi = lds_tr16(img + w4_km_off<BX>(kr + 4, col));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

// MFMA with the accumulator tile pinned in the AGPR half of the register file.  Left to the compiler, a kernel whose 256
// accumulator registers cannot all be VGPRs keeps the loop-carried values in VGPRs and copies every tile to AGPRs and back around
// each v_mfma (4 v_accvgpr_write + wait states per MFMA, 192 v_accvgpr_read per K-step in the first build of this file).  As an
// asm operand of class "a" the tile lives in AGPRs for the whole K loop; consecutive MFMAs here never touch the same tile (a tile
// is revisited 48-64 MFMAs later), so no software wait states are needed between them.
DEVFN void mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the first K-step: C = 0, so the tile is DEFINED by an asm of class "a" -- the compiler never sees an accumulator value in a
// VGPR (an explicit zero initialisation made it keep the loop-carried values in VGPRs and copy around every MFMA)
DEVFN void mfma_init(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

template <int BX, bool KM, int NP, int NT>
struct DmaW4 {
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
      const char* src = ok ? ptr[i] + kt * kstep : (const char*)g_zero_page_w4;
      glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(img + (i * NT + wave * 64) * 16)));
    }
  }
  DEVFN void issue1(int i, int64_t kt, int64_t K, char* img, int wave) const {       // piece i only (i is a compile-time constant)
    const bool ok = okx[i] && (kt * 32 + kofs[i] < K);
    const char* src = ok ? ptr[i] + kt * kstep : (const char*)g_zero_page_w4;
    glds16_asm(src, __builtin_amdgcn_readfirstlane(lds_addr(img + (i * NT + wave * 64) * 16)));
  }
};

// 256 x 192 output tile (NI = 6 accumulator tiles of 16 columns per wave), row-major A and B (the forward Linears).
// Ring protocol.  Stage s = the 32-deep K slice s.  A(s) lives in A-slot s % 5 (16 KB each), B(s) in B-slot s % 6 (192 rows x 64 B =
// 12 KB each): 152 KB.  Step t computes stage t: its B fragments are already in registers (read during step t-1), its A fragments
// are read during the step (two regions ahead of their MFMAs), and the B fragments of stage t+1 are read for the next step.  So
// A(t) and B(t+1) must have landed when step t opens; both were issued together in step t-4 ("group t-4" = A(t) then B(t+1)), and a
// step issues A(t+4) into the slot of A(t-1) and B(t+5) into the slot of B(t-1) -- four stages of each operand in flight, the same
// bytes in flight as the 8-wave kernel, which a single five-slot ring cannot offer once the B fragments are read a step early
// (first version of this file: three stages in flight, 0.96 us per step -- the loop waited for memory, not for MFMAs).
template <typename OT, int ABL = 0>       // ABL (diagnosis, wrong results): 1 no DMA in the loop, 2 no fragment reads in the loop, 3 both
DEVFN void w4_tile256(const GemmP& p, int64_t z, int wg, int ntn, int nwg, char* smem) {
  constexpr int BM = 256, BN = 192, WN = 2, NSA = 5, NSB = 6, NT = 256, SWD = 8;
  constexpr int WTM = 128, MI = 8, NI = 6, WTN = NI * 16, BNU = WN * WTN;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int NPA = A_BYTES / (NT * 16), NPB = B_BYTES / (NT * 16), LPS = NPA + NPB;      // 4 + 3 pieces per thread and step
  constexpr int B_BASE = NSA * A_BYTES;
  static_assert(BNU == BN && NPA == 4 && NPB == 3 && 3 * LPS <= 63, "geometry");
  constexpr bool AKM = false, BKM = false;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntm = nwg / ntn;
  const int strip = wg / (SWD * ntm), sw = (ntn - strip * SWD) < SWD ? (ntn - strip * SWD) : SWD;
  const int within = wg - strip * SWD * ntm;
  const int64_t m0 = (int64_t)(within / sw) * BM, n0 = (int64_t)(strip * SWD + within % sw) * BNU;

  DmaW4<BM, AKM, NPA, NT> ua;
  DmaW4<BN, BKM, NPB, NT> ub;
  ua.init(p.A + z * p.sA * 2, p.lda, m0, p.M, tid);
  ub.init(p.B + z * p.sB * 2, p.ldb, n0, p.N, tid);

  f32x4 acc[MI][NI];                            // defined by the first K-step's MFMAs (mfma_init), AGPRs throughout
  const int64_t nkt = (p.K + 31) / 32;
  // prologue: B(0), then the groups A(0) B(1) | A(1) B(2) | A(2) B(3) | A(3) B(4)
  ub.issue(0, p.K, smem + B_BASE, wave);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ua.issue(s, p.K, smem + s * A_BYTES, wave);
    ub.issue(s + 1, p.K, smem + B_BASE + (s + 1) * B_BYTES, wave);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPS) : "memory");        // B(0) has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 fb0[NI], fb1[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) fb0[j] = w4_frag<BN, BKM>(smem + B_BASE, wn * WTN + j * 16, lane);

  // The step is written as eight REGIONS, one per 16-row block of the wave's tile, each closed by a scheduling barrier: region i
  // issues one LDS-DMA piece, reads the A fragment needed two regions later and one B fragment of the NEXT stage, then its six
  // MFMAs.  One wave per SIMD hides nothing by itself: with the compiler's own order (all reads, then the whole DMA issue block,
  // then the MFMAs) the step took 1.00 us; here every non-MFMA instruction sits in the shadow of its region's MFMAs.
#define W4_REGION(i, fb_c, fb_n, MF)                                                                             \
  {                                                                                                              \
    if (!(ABL & 1)) {                                                                                            \
    if ((i) < NPA) ua.issue1((i) < NPA ? (i) : 0, t + 4, p.K, smem + asl_f * A_BYTES, wave);                     \
    else if ((i) < LPS) ub.issue1(((i) >= NPA && (i) < LPS) ? (i) - NPA : 0, t + 5, p.K, smem + B_BASE + bsl_f * B_BYTES, wave); \
    }                                                                                                            \
    if (!(ABL & 2)) {                                                                                            \
    if ((i) + 2 < MI) fa[(i) + 2 < MI ? (i) + 2 : 0] = w4_frag<BM, AKM>(cA, wm * WTM + ((i) + 2) * 16, lane);   \
    if ((i) < NI) fb_n[(i) < NI ? (i) : 0] = w4_frag<BN, BKM>(nB, wn * WTN + ((i) < NI ? (i) : 0) * 16, lane);   \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) MF(acc[i][j], fb_c[j], fa[i]);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
#define W4_STEP(fb_c, fb_n, MF)                                                                                  \
  {                                                                                                              \
    if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");   /* group t-4 = A(t), B(t+1): own pieces landed */ \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                /* my reads of A(t-1) / B(t) are done */      \
    __builtin_amdgcn_s_barrier();                                                                                \
    asm volatile("" ::: "memory");                                                                               \
    const char* cA = smem + asl * A_BYTES;                                                                       \
    const char* nB = smem + B_BASE + bsl_n * B_BYTES;                                                            \
    bf16x8 fa[MI];                                                                                               \
    if (ABL & 2) { _Pragma("unroll") for (int q = 0; q < MI; ++q) fa[q] = fb_c[q % NI]; }                        \
    else {                                                                                                       \
    fa[0] = w4_frag<BM, AKM>(cA, wm * WTM, lane);                                                                \
    fa[1] = w4_frag<BM, AKM>(cA, wm * WTM + 16, lane);                                                           \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    W4_REGION(0, fb_c, fb_n, MF) W4_REGION(1, fb_c, fb_n, MF) W4_REGION(2, fb_c, fb_n, MF) W4_REGION(3, fb_c, fb_n, MF) \
    W4_REGION(4, fb_c, fb_n, MF) W4_REGION(5, fb_c, fb_n, MF) W4_REGION(6, fb_c, fb_n, MF) W4_REGION(7, fb_c, fb_n, MF) \
    asl_f = asl;                                  /* next step refills the slot this step consumed */             \
    asl = (asl + 1 == NSA) ? 0 : asl + 1;                                                                        \
    bsl_f = (bsl_f + 1 == NSB) ? 0 : bsl_f + 1;                                                                  \
    bsl_n = (bsl_n + 1 == NSB) ? 0 : bsl_n + 1;                                                                  \
  }

  // step t: A(t) in slot asl = t % 5; B(t+1) in slot bsl_n = (t+1) % 6; refills: A(t+4) -> slot asl_f = (t+4) % 5 = (t-1) % 5,
  // B(t+5) -> slot bsl_f = (t+5) % 6 = (t-1) % 6
  int asl = 0, asl_f = 4, bsl_n = 1, bsl_f = 5;
  int64_t t = 0;
  W4_STEP(fb0, fb1, mfma_init)                   // (K >= 1: there is always a first step)
  ++t;
  while (t < nkt) {
    W4_STEP(fb1, fb0, mfma_acc)
    if (++t >= nkt) break;
    W4_STEP(fb0, fb1, mfma_acc)
    ++t;
  }
#undef W4_STEP
#undef W4_REGION
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring's trailing (zero page) pieces
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results are in the AGPRs before anything reads them

  const int g = lane >> 4, li = lane & 15;
  const DropKey dk = make_drop((p.epi & GSTVD_EPI_DROPOUT) ? p.p : 0.f, p.site, p.rng);
  if constexpr (sizeof(OT) == 2) {
    if (epilogue_rows_ok(p)) {
      __builtin_amdgcn_s_barrier();            // every wave is done with the ring: its LDS is free for the row-wise epilogue
      gemm_epilogue_rows<MI, NI, 4, true>(p, dk, acc, z, m0 + wm * WTM, n0 + wn * WTN, smem + wave * epi_wave_bytes<NI, 4>(), lane);
      return;
    }
  }
  // tile-wise epilogue (fp32 outputs, unaligned rows): every accumulator tile makes a round trip through the wave's own LDS
  // scratch (asm ds_write from the AGPRs, ordinary read back in the same MFMA layout)
  __builtin_amdgcn_s_barrier();
  float* scr = (float*)(smem + wave * 1024) + lane * 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      epi_park_tile<true>(scr, acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const f32x4 v = *(const volatile f32x4*)scr;
      gemm_epilogue_tile<bf16, OT>(p, dk, v, z, m0 + wm * WTM + i * 16 + li, n0 + wn * WTN + j * 16 + 4 * g);
    }
}

DEVFN int xcd_remap_w4(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename OT, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmP p, int ntn, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w4_tile256<OT, ABL>(p, blockIdx.y, xcd_remap_w4(blockIdx.x, nwg), ntn, nwg, smem);
}

constexpr int LDS_W4 = 5 * 256 * 64 + 6 * 192 * 64;

template <typename OT>
static int launch_w4(const GemmP& p, int64_t batch, hipStream_t s) {
  auto k = gemm_w4_kernel<OT>;
  auto k1 = gemm_w4_kernel<OT, 1>;
  auto k2 = gemm_w4_kernel<OT, 2>;
  auto k3 = gemm_w4_kernel<OT, 3>;
  static int attr_rc = ensure_lds(k, LDS_W4) | ensure_lds(k1, LDS_W4) | ensure_lds(k2, LDS_W4) | ensure_lds(k3, LDS_W4);
  if (attr_rc) return attr_rc;
  static const int abl = [] { const char* e = getenv("GSTVD_W4_ABL"); return e ? atoi(e) : 0; }();
  if (abl == 1) k = k1; else if (abl == 2) k = k2; else if (abl == 3) k = k3;
  const int ntm = (int)((p.M + 255) / 256), ntn = (int)((p.N + 191) / 192);
  GSTVD_LAUNCH(k, dim3((unsigned)(ntm * ntn), (unsigned)batch), dim3(256), LDS_W4, s, p, ntn, ntm * ntn);
  GSTVD_LAUNCH_CHECK();
  return 0;
}

// row-major A and B, 192-wide tiles only (niu == 3 in the caller's terms); everything else stays on the 8-wave kernels
int gemm_w4_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, int niu, hipStream_t s) {
  if (akm || bkm || niu != 3) return GSTVD_E_UNSUPPORTED;
  return out_f32 ? launch_w4<float>(p, batch, s) : launch_w4<bf16>(p, batch, s);
}
