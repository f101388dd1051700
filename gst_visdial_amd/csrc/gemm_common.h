// Shared pieces of the two GEMM main loops (gemm.hip: register-staged, any dtype; gemm_dma.hip: LDS-DMA ring, bf16).
#pragma once
#include "common.h"

struct GemmP {
  const char* A; const char* B; char* C;
  const float* bias; const char* addend; char* aux;
  int64_t M, N, K, lda, ldb, ldc, ldadd, ldaux, sA, sB, sC, sAdd, sAux;
  int epi; float alpha, p; uint32_t site; const uint64_t* rng;
};

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16> { static constexpr int VE = 8, BK = 64; };
template <> struct TileCfg<float> { static constexpr int VE = 4, BK = 32; };

// ---- LDS images -------------------------------------------------------------------------------
// row-major image: BX rows of 128 bytes (BK elements); 16-byte slot s of row r lives at slot s ^ (r & 7)
DEVFN int rm_off(int row, int slot) { return row * 128 + (((slot ^ row) & 7) << 4); }
// k-major bf16 image: BK rows of BX*2 bytes; 32-byte block b of k-row r lives at block b ^ f(r)
template <int BX> DEVFN int km_off_bf16(int krow, int col) {
  constexpr int NB = BX / 16;
  int f = (krow & 3) | (((krow >> 3) & 1) << 2);
  int blk = ((col >> 4) ^ f) & (NB - 1);
  return krow * (BX * 2) + (blk << 5) + ((col & 15) << 1);
}
template <int BX> constexpr int km_row_bytes_f32() { return (BX + 4) * 4; }

template <typename T, int BX, bool KM> constexpr int image_bytes() {
  return KM ? (sizeof(T) == 2 ? TileCfg<T>::BK * BX * 2 : TileCfg<T>::BK * km_row_bytes_f32<BX>()) : BX * 128;
}

// ---- global -> registers ----------------------------------------------------------------------
template <typename T, int BX, bool KM, int NT, int NV>
DEVFN void load_tile(u32x4 (&reg)[NV], const char* g, int64_t ld, int64_t x0, int64_t X, int64_t k0, int64_t K, int tid) {
  constexpr int VE = TileCfg<T>::VE, BK = TileCfg<T>::BK;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int v = tid + i * NT;
    int64_t x, k;
    if (KM) { constexpr int VPR = BX / VE; k = k0 + v / VPR; x = x0 + (v % VPR) * VE; }
    else    { constexpr int VPR = BK / VE; x = x0 + v / VPR; k = k0 + (v % VPR) * VE; }
    u32x4 z = {0u, 0u, 0u, 0u};
    if (x < X && k < K) {
      const char* ptr = g + (KM ? (k * ld + x) : (x * ld + k)) * (int64_t)sizeof(T);
      z = *(const u32x4*)ptr;
    }
    reg[i] = z;
  }
}
// ---- registers -> LDS image ---------------------------------------------------------------------
template <typename T, int BX, bool KM, int NT, int NV>
DEVFN void store_tile(const u32x4 (&reg)[NV], char* img, int tid) {
  constexpr int VE = TileCfg<T>::VE, BK = TileCfg<T>::BK;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int v = tid + i * NT;
    int off;
    if (KM) {
      constexpr int VPR = BX / VE;
      int kr = v / VPR, cv = v % VPR;
      off = (sizeof(T) == 2) ? km_off_bf16<BX>(kr, cv * VE) : kr * km_row_bytes_f32<BX>() + cv * 16;
    } else {
      constexpr int VPR = BK / VE;
      off = rm_off(v / VPR, v % VPR);
    }
    *(u32x4*)(img + off) = reg[i];
  }
}

// ---- LDS image -> MFMA fragment ----------------------------------------------------------------
// bf16: 8 elements k = kk*32 + 8g + j for x = xb + (lane & 15)
template <int BX, bool KM> DEVFN bf16x8 frag_bf16(const char* img, int xb, int kk, int lane) {
  int g = lane >> 4, li = lane & 15;
  if (!KM) {
    return *(const bf16x8*)(img + rm_off(xb + li, kk * 4 + g));
  } else {
    int kr = kk * 32 + 8 * g + (li >> 2), col = xb + 4 * (lane & 3);
    s16x4 lo = lds_tr16(img + km_off_bf16<BX>(kr, col));
    s16x4 hi = lds_tr16(img + km_off_bf16<BX>(kr + 4, col));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}
// f32: one element k = ks*4 + g for x = xb + (lane & 15)
template <int BX, bool KM> DEVFN float frag_f32(const char* img, int xb, int ks, int lane) {
  int g = lane >> 4, li = lane & 15;
  if (!KM) return *(const float*)(img + rm_off(xb + li, ks) + g * 4);
  return *(const float*)(img + (ks * 4 + g) * km_row_bytes_f32<BX>() + (xb + li) * 4);
}


// ---- epilogue shared by both main loops: lane owns C[m0 + li][n0 + 4g .. +3] of a 16x16 accumulator tile ----
template <typename T, typename OT>
DEVFN void gemm_epilogue_tile(const GemmP& p, const DropKey& dk, f32x4 acc, int64_t z, int64_t m, int64_t n) {
  if (m >= p.M || n >= p.N) return;
  OT* C = (OT*)p.C + z * p.sC;
  f32x4 v = acc * p.alpha;
  if (p.epi & GSTVD_EPI_BIAS) v += *(const f32x4*)(p.bias + n);
  if (p.epi & GSTVD_EPI_ADD) v += ld4((const OT*)p.addend + z * p.sAdd + m * p.ldadd + n);
  if (p.epi & GSTVD_EPI_GELU) {
    // aux keeps gelu'(pre-activation): the backward epilogue is then a plain multiply (no erf / exp there)
    constexpr bool FAST = sizeof(T) == 2;
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float g_, d_; gelu_both<FAST>(v[e], g_, d_); v[e] = g_; d[e] = d_; }
    st4((T*)p.aux + z * p.sAux + m * p.ldaux + n, d);
  }
  if (p.epi & GSTVD_EPI_DGELU) v *= ld4((const T*)p.aux + z * p.sAux + m * p.ldaux + n);
  if (dk.on) v *= drop_factor4(dk, (uint64_t)((z * p.M + m) * p.N + n));
  st4(C + m * p.ldc + n, v);
}

// ---- row-wise epilogue through LDS (bf16 outputs of the LDS-DMA kernels) ---------------------------------------------
// In the MFMA layout a lane owns 4 columns of 16 different rows, so every C store / addend / aux access is an 8-byte piece
// and one wave instruction touches 16 rows.  Here a wave parks its fp32 accumulators -- HB 16-row blocks at a time -- in
// its own slice of the (now idle) ring, reads them back with each lane owning 8 consecutive columns of ONE row, and runs
// the whole epilogue in that layout: 16-byte accesses, full 64/128-byte row segments per instruction.  Same arithmetic
// per element in the same order as gemm_epilogue_tile (fp32 until the final rounding), so results are bit-identical.
// The caller guarantees (uniformly) 16-byte alignment of every operand row: see epilogue_rows_ok().
template <int NI> constexpr int epi_row_floats() { return NI * 16 + 4; }          // +4 floats: conflict-free b128 parking
template <int NI, int HB> constexpr int epi_wave_bytes() { return HB * 16 * epi_row_floats<NI>() * 4; }

DEVFN bool epilogue_rows_ok(const GemmP& p) {
  bool ok = (p.ldc % 8 == 0) && (p.sC % 8 == 0);
  if (p.epi & GSTVD_EPI_ADD) ok = ok && (p.ldadd % 8 == 0) && (p.sAdd % 8 == 0) && (((uintptr_t)p.addend & 15) == 0);
  if (p.epi & (GSTVD_EPI_GELU | GSTVD_EPI_DGELU)) ok = ok && (p.ldaux % 8 == 0) && (p.sAux % 8 == 0) && (((uintptr_t)p.aux & 15) == 0);
  if (p.epi & GSTVD_EPI_BIAS) ok = ok && (((uintptr_t)p.bias & 15) == 0);
  return ok;
}

DEVFN void ld8(const bf16* q, f32x4& lo, f32x4& hi) {
  const bf16x8 v = *(const bf16x8*)q;
  lo = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  hi = (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
DEVFN void st8(bf16* q, const f32x4& lo, const f32x4& hi) {
  const bf16x8 v = {(bf16)lo[0], (bf16)lo[1], (bf16)lo[2], (bf16)lo[3], (bf16)hi[0], (bf16)hi[1], (bf16)hi[2], (bf16)hi[3]};
  *(bf16x8*)q = v;
}

// LATE: the addend / aux rows of a pass are fetched AFTER its accumulators are parked instead of in front of them.  The
// producer / consumer tiles run three waves per SIMD (168 VGPRs): with all 128 accumulator registers of a 256-wide tile still
// live, 32 registers of prefetched rows spilled ~100 B per lane to scratch; once half of the accumulators
// sit in LDS there is room.  One load latency per pass is exposed instead (the rows are consumed in issue order).
template <int MI, int NI, int HB, bool LATE = false>
DEVFN void gemm_epilogue_rows(const GemmP& p, const DropKey& dk, const f32x4 (&acc)[MI][NI], int64_t z, int64_t mw, int64_t nw,
                              char* lds_wave, int lane) {
  static_assert(MI % HB == 0, "row blocks per pass must divide the wave tile");
  // lanes per row: 8 columns each, rounded up to a power of two (NI = 3: 8 lanes, the last two idle)
  constexpr int S = epi_row_floats<NI>(), LPR = NI <= 1 ? 2 : (NI <= 2 ? 4 : 8), RPI = 64 / LPR, ROWS = HB * 16;
  const int g = lane >> 4, li = lane & 15;
  const int rl = lane / LPR, c8 = (lane % LPR) * 8;
  float* park = (float*)lds_wave;
  bf16* C = (bf16*)p.C + z * p.sC;
  const int64_t n = nw + c8;
  const bool in_tile = c8 < NI * 16;
  const bool n_lo = in_tile && n < p.N, n_hi = in_tile && n + 4 < p.N;   // N % 4 == 0: a 4-column half is all in or all out
  f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
  if (p.epi & GSTVD_EPI_BIAS) {
    if (n_lo) b_lo = *(const f32x4*)(p.bias + n);
    if (n_hi) b_hi = *(const f32x4*)(p.bias + n + 4);
  }
  // the pass's addend (or, without an addend, aux) rows are fetched up front -- one 16-byte load per row in flight per lane
  // while the accumulators make their LDS round trip -- instead of one exposed load latency per row
  const bool pre_add = (p.epi & GSTVD_EPI_ADD) != 0, pre_aux = !pre_add && (p.epi & GSTVD_EPI_DGELU);
  const bf16* pre_base = pre_add ? (const bf16*)p.addend + z * p.sAdd : (const bf16*)p.aux + z * p.sAux;
  const int64_t pre_ld = pre_add ? p.ldadd : p.ldaux;
  // LATE: two sub-batches of row groups per pass, each fetched right in front of its use (16 instead of 32 registers of rows)
  constexpr int NR = ROWS / RPI, SUBN = (LATE && NR % 2 == 0) ? 2 : 1, NRS = NR / SUBN;
#pragma unroll
  for (int pb = 0; pb < MI / HB; ++pb) {
    bf16x8 pre[NRS];
    auto fetch_pre = [&](int sb) {
      if (pre_add || pre_aux) {
#pragma unroll
        for (int rr = 0; rr < NRS; ++rr) {
          const int64_t m = mw + pb * ROWS + (sb * NRS + rr) * RPI + rl;
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
          pre[rr] = __builtin_bit_cast(bf16x8, zero);
          if (m < p.M && n_hi) pre[rr] = *(const bf16x8*)(pre_base + m * pre_ld + n);
          else if (m < p.M && n_lo) {
            const bf16x4 h = *(const bf16x4*)(pre_base + m * pre_ld + n);
            pre[rr][0] = h[0]; pre[rr][1] = h[1]; pre[rr][2] = h[2]; pre[rr][3] = h[3];
          }
        }
      }
    };
    if (!LATE) fetch_pre(0);
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) *(f32x4*)(park + (i * 16 + li) * S + j * 16 + 4 * g) = acc[pb * HB + i][j];
#pragma unroll
    for (int sb = 0; sb < SUBN; ++sb) {
      if (LATE) {
        __builtin_amdgcn_sched_barrier(0);      // the parking stores / the previous sub-batch are issued before these loads
        fetch_pre(sb);
      }
#pragma unroll
      for (int rr = 0; rr < NRS; ++rr) {
        const int row = (sb * NRS + rr) * RPI + rl;
        const int64_t m = mw + pb * ROWS + row;
        if (m >= p.M || !n_lo) continue;
        f32x4 lo = *(const f32x4*)(park + row * S + c8), hi = *(const f32x4*)(park + row * S + c8 + 4);
        lo = lo * p.alpha + b_lo;
        hi = hi * p.alpha + b_hi;
        const bool full = n_hi;
        const f32x4 q_lo = {(float)pre[rr][0], (float)pre[rr][1], (float)pre[rr][2], (float)pre[rr][3]};
        const f32x4 q_hi = {(float)pre[rr][4], (float)pre[rr][5], (float)pre[rr][6], (float)pre[rr][7]};
        if (pre_add) { lo += q_lo; hi += q_hi; }
        if (p.epi & GSTVD_EPI_GELU) {
          f32x4 d0, d1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float g_, d_;
            gelu_both<true>(lo[e], g_, d_); lo[e] = g_; d0[e] = d_;
            gelu_both<true>(hi[e], g_, d_); hi[e] = g_; d1[e] = d_;
          }
          bf16* xp = (bf16*)p.aux + z * p.sAux + m * p.ldaux + n;
          if (full) st8(xp, d0, d1); else st4(xp, d0);
        }
        if (p.epi & GSTVD_EPI_DGELU) {
          if (pre_aux) { lo *= q_lo; hi *= q_hi; }
          else {
            const bf16* xp = (const bf16*)p.aux + z * p.sAux + m * p.ldaux + n;
            if (full) { f32x4 a0, a1; ld8(xp, a0, a1); lo *= a0; hi *= a1; }
            else lo *= ld4(xp);
          }
        }
        if (dk.on) {
          const uint64_t e0 = (uint64_t)((z * p.M + m) * p.N + n);
          lo *= drop_factor4(dk, e0);
          if (full) hi *= drop_factor4(dk, e0 + 4);
        }
        if (full) st8(C + m * p.ldc + n, lo, hi); else st4(C + m * p.ldc + n, lo);
      }
    }
  }
}

// Plan-only mode of the GEMM entry points (gstvd_gemm_kernel_name): the dispatch runs exactly as for a launch, but the launch
// site records the host-side kernel handle it WOULD have launched instead of launching it.  bench.py reports the launched
// symbol from this -- the dispatch itself is the single source of truth, not a host-side re-derivation of its rule.
extern thread_local const void** gstvd_plan_capture;
#define GSTVD_LAUNCH(kern, grid, block, lds, stream, ...)                                              \
  do {                                                                                                 \
    auto k_ = (kern);                                                                                  \
    if (gstvd_plan_capture) *gstvd_plan_capture = (const void*)k_;                                     \
    else hipLaunchKernelGGL(k_, grid, block, lds, stream, __VA_ARGS__);                                \
  } while (0)

template <typename K> static int ensure_lds(K kernel, int bytes) {
  if (bytes <= 48 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? 0 : (int)e;
}

// gemm_dma.hip: bf16 inputs through the LDS-DMA pipelined main loop; returns GSTVD_E_UNSUPPORTED when not applicable
int gemm_dma_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s);
int gemm_dma_splitk_dispatch(const GemmP& p, int akm, int bkm, int out_f32, int S, void* ws, int64_t ws_bytes, hipStream_t s);
// gemv.hip: M <= 16 (one new token per row of a KV-cached decode step), operands straight from global memory to MFMA fragments
int gemv16_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s);
int gemv16_ln_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, const float* gamma, const float* beta, float eps,
                       void* y_out, int64_t ldy, hipStream_t s);
// gemm_dma256.hip: 256x256x32 tile for problems whose grid still fills the chip
int gemm_dma256_dispatch(const GemmP& p, int64_t batch, int akm, int bkm, int out_f32, hipStream_t s);

// LDS-DMA issue in inline asm.  The compiler's waitcnt pass treats a *builtin* global_load_lds as a pending LDS write and
// puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 that follows (it cannot see that the ring slot being
// read is not the one being filled) -- which serialises the ring for k-major operands.  Issued from asm the DMA is
// invisible to that pass; ordering is ours: a counted `s_waitcnt vmcnt(N)` + s_barrier before a slot is read, and
// `s_waitcnt vmcnt(0)` before the epilogue.  m0 (the LDS destination base) is saved/restored around the instruction.
DEVFN void glds16_asm(const char* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
DEVFN unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
