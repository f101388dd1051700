/*
 * gstvd_hip.h -- C ABI of the MI355X (gfx950) kernels behind the gst-visdial enc_dec_a hot path.
 *
 * The reference (gicheonkang/gst-visdial) has NO native/FFI boundary: its hot path is eager
 * PyTorch inside models/vilbert_dialog.py, models/visual_dialog_{encoder,decoder,model}.py
 * (SURVEY.md section 8b).  This header is therefore the build's own boundary; every entry
 * point names the reference code it replaces.  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - the caller owns every buffer (no allocation inside), passes the hipStream_t to launch on;
 *   - return value: 0 = ok, <0 = invalid argument (GSTVD_E_*), >0 = hipError_t of the launch;
 *   - re-entrant, no global mutable state; safe under hipGraph stream capture
 *     (no sync / malloc / memcpy inside);
 *   - dtype: GSTVD_F32 (exact-fp32 parity mode, f32-input MFMA) or GSTVD_BF16 (bf16 storage,
 *     fp32 accumulate / statistics) -- the arithmetic type of activations;
 *   - dropout is counter based: keep(e) = hash(rng[0] (seed), rng[1] (step offset), site, e);
 *     `rng` points to two uint64 in device memory so a captured graph sees a fresh offset
 *     per replay; backward regenerates the mask instead of storing it.
 */
#ifndef GSTVD_HIP_H
#define GSTVD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gstvd_stream_t; /* hipStream_t */

enum { GSTVD_F32 = 0, GSTVD_BF16 = 1 };
enum {
  GSTVD_E_DTYPE = -1, GSTVD_E_SHAPE = -2, GSTVD_E_ALIGN = -3, GSTVD_E_NULL = -4, GSTVD_E_UNSUPPORTED = -5
};

/* ---- library info ----------------------------------------------------------------------- */
int gstvd_abi_version(void);            /* bumps on any signature change */
const char* gstvd_build_arch(void);     /* "gfx950" */

/* ---- GEMM: every nn.Linear forward / dgrad / wgrad on the path ---------------------------
 * C[m,n] = epi( alpha * sum_k A(m,k) * B(n,k) ),   batch = grid.z with element strides s*.
 *   A(m,k) = a_kmajor ? A[k*lda + m] : A[m*lda + k];   B(n,k) = b_kmajor ? B[k*ldb + n] : B[n*ldb + k]
 * forward  y = x W^T + b        : A=x,  B=W            (vilbert_dialog.py:381-383,417,446,459 ...)
 * dgrad    dx = dy W            : A=dy, B=W  b_kmajor
 * wgrad    dW = dy^T x          : A=dy a_kmajor, B=x b_kmajor, fp32 output, optional accumulate
 * Epilogue flags (bitmask), applied in this order:
 *   GSTVD_EPI_BIAS      += bias[n]                       (bias is always fp32)
 *   GSTVD_EPI_ADD       += addend[m,n]                   (dtype_out; residual / grad accumulate)
 *   GSTVD_EPI_GELU      aux[m,n] = gelu_erf'(v) (dtype_in); v = gelu_erf(v)   (vilbert_dialog.py:115-121)
 *   GSTVD_EPI_DGELU     v *= aux[m,n]   (aux as written by the forward GELU epilogue)
 *   GSTVD_EPI_DROPOUT   v *= keep(site, (z*M+m)*N+n) / (1-p)      (VLFusion dropout, visual_dialog_model.py:133)
 * Constraints: N % 4 == 0; row-major operands need K % (16/sizeof(T)) == 0 and 16-byte aligned rows;
 * k-major operands need their M (resp. N) extent % (16/sizeof(T)) == 0.  M and the k extent of
 * k-major operands are arbitrary (predicated / zero filled).
 */
enum { GSTVD_EPI_BIAS = 1, GSTVD_EPI_ADD = 2, GSTVD_EPI_GELU = 4, GSTVD_EPI_DGELU = 8, GSTVD_EPI_DROPOUT = 16,
       /* grouped weight-gradient launches only (gstvd_gemm_grouped with a k-major A, see gstvd_gemm_group_caps):
        * bias[m] (=|+=) sum_k A[k][m] -- the bias gradient dY^T.1 that autograd computes next to dW = dY^T X, taken from the
        * dY tiles the launch stages anyway (the tiles of column block 0 do it); COLSUM_ACC adds to bias instead of overwriting */
       GSTVD_EPI_COLSUM = 32, GSTVD_EPI_COLSUM_ACC = 64,
       /* gstvd_gemm_grouped_adamw only: C of this problem is a weight's slot of the flat gradient buffer and nothing else adds
        * to it this step -- the launch applies the AdamW update to the weight in its epilogue (see there) */
       GSTVD_EPI_ADAMW = 128 };

typedef struct {
  const void* A; const void* B; void* C;
  const float* bias; const void* addend; void* aux;
  int64_t M, N, K;
  int64_t lda, ldb, ldc, ldadd, ldaux;
  int64_t batch, sA, sB, sC, sAdd, sAux;
  int32_t dtype_in, dtype_out, a_kmajor, b_kmajor, epilogue;
  float alpha, dropout_p;
  uint32_t site;
  const uint64_t* rng;
} gstvd_gemm_t;
int gstvd_gemm(const gstvd_gemm_t* g, gstvd_stream_t s);

/* Split-K form of the same problem (bf16 inputs, batch = 1): for skinny, deep problems -- the decoder's 400-row
 * GEMMs with K >= 2304 and the LM-head input gradient (K = 30528) -- whose few 64x64 output tiles cannot keep the chip's
 * LDS fill paths busy.  `splits` workgroups share a tile, park fp32 partials in `ws`, the last to arrive adds them in
 * split order (result independent of arrival order) and runs the epilogue.  `ws`: caller-owned device scratch of at
 * least gstvd_gemm_splitk_ws_bytes() bytes (4 KiB of per-tile arrival counters, then the partials; at most 1024 tiles),
 * zero-filled once (the kernel leaves its counters zero; one scratch serves launches of any shape), never shared by
 * launches that may run concurrently (one per stream). */
int64_t gstvd_gemm_splitk_ws_bytes(int64_t M, int64_t N, int32_t splits);
int gstvd_gemm_splitk(const gstvd_gemm_t* g, int32_t splits, void* ws, int64_t ws_bytes, gstvd_stream_t s);

/* Grouped form: `table_dev` is a DEVICE array of nprob independent problems (batch ignored, = 1) that share dtypes
 * and operand layouts; they run as ONE launch over all their T x T tiles, T = gstvd_gemm_group_tile() (256).
 * tile_off_dev[i] = first tile id of problem i (ceil(M/T)*ceil(N/T) tiles each), device int32[nprob].  The engine uses it
 * for the deferred weight-gradient GEMMs of a whole backward pass (dW = dy^T x: a_kmajor = b_kmajor = 1, fp32 out, EPI_ADD
 * to accumulate), whose individual grids are too small to fill the chip.  bf16 inputs only.
 * block_map_dev (ABI 5; NULL = the library's own order): device int32[nblocks], nblocks >= total_tiles -- workgroup b runs
 * tile block_map_dev[b] (a tile id as in tile_off_dev), or nothing when the entry is negative; every tile id must appear
 * exactly once.  Workgroups b, b + 8, b + 16, ... share one XCD and its L2 (a speed fact, never a correctness one): the
 * caller that knows which problems share operand panels and run equally long (same K) queues them back to back on one XCD,
 * so that the ~32 tiles an XCD runs at a time stream the same panels in step -- autograd has no counterpart, it is pure
 * placement.
 * ABI 6: an entry's tile id is its bits 0-20 (total_tiles < 2^21); bits 21-30 of a non-negative entry are reserved and
 * ignored (round 5 measured a start barrier between the workgroups of a unit in them: fewer fetches, slower launch;
 * profiles/r05_group_sync_ab.txt). */
int gstvd_gemm_group_tile(void);
int gstvd_gemm_grouped(const gstvd_gemm_t* table_dev, const int32_t* tile_off_dev, int64_t nprob, int64_t total_tiles,
                       int32_t dtype_in, int32_t dtype_out, int32_t a_kmajor, int32_t b_kmajor,
                       const int32_t* block_map_dev, int64_t nblocks, gstvd_stream_t s);

/* ---- fused (bias-free) dropout + residual + LayerNorm, and the two embedding front ends -----
 * mode GSTVD_LN_RESID : h = drop_pre(x) + res ; y = drop_post(LN(h))   (BertSelfOutput/BertOutput/
 *        BertBiOutput, vilbert_dialog.py:416-420,458-462,735-742; HF BertSelfOutput/BertOutput)
 * mode GSTVD_LN_EMBED : h = word[ids] + pos[t] + type(seg)             (BertEmbeddingsDialog, :324-352)
 * mode GSTVD_LN_IMAGE : h = x + loc W_loc^T + b_loc                    (BertImageEmbeddings, :1420-1427)
 * LayerNorm is the TF-style (x-u)/sqrt(var+eps) of vilbert_dialog.py:283-296.
 * mean / rstd (fp32 [M]) are saved for backward.
 */
enum { GSTVD_LN_RESID = 0, GSTVD_LN_EMBED = 1, GSTVD_LN_IMAGE = 2 };

typedef struct {
  int32_t mode, dtype;
  int64_t M, H;                 /* rows, hidden (H % 4 == 0, H <= 2048) */
  const void* x; int64_t ldx;   /* RESID / IMAGE: [M,H] */
  const void* res; int64_t ldres; /* RESID: residual or NULL */
  const float* gamma; const float* beta; float eps;
  void* y; int64_t ldy;
  float* mean; float* rstd;
  float p_pre, p_post; uint32_t site_pre, site_post; const uint64_t* rng;
  /* EMBED */
  const int64_t* ids; const int64_t* segs; /* [M]; segs may be NULL (=0) */
  int64_t T; int32_t type_vocab;           /* position = row % T + pos_offset */
  const float* word; const float* pos; const float* tt; const float* tt_ext; /* fp32 tables, row stride H */
  /* IMAGE */
  const float* loc; const float* w_loc; const float* b_loc; /* loc [M,5] fp32, w_loc [H,5], b_loc [H] */
  int64_t pos_offset;                      /* EMBED: added to the position (KV-cached decode feeds one token per row) */
} gstvd_ln_t;
int gstvd_ln_fwd(const gstvd_ln_t* p, gstvd_stream_t s);

/* backward: dh = LN'(dy) ; dres = dh ; dx = drop_pre'(dh).
 * partial: fp32 [nblk, 3, H] scratch (nblk = gstvd_ln_bwd_blocks(M)) receiving per-block column sums of
 * (dy*xhat, dy, dx); reduce them with gstvd_colsum_partials.
 * EMBED mode scatters dh into dword/dpos (and dtt/dtt_ext for segment ids >= 2) with fp32 atomic adds instead of
 * writing dres/dx; its partial is [nblk, 4, H]: (dy*xhat, dy, dh of segment-0 rows, dh of segment-1 rows) -- the
 * caller adds vectors 2 and 3 to dtt rows 0 and 1 (no atomics onto those two hot rows). */
typedef struct {
  gstvd_ln_t f;                 /* the forward descriptor (y unused) */
  const void* dy; int64_t lddy;
  void* dres; int64_t lddres;   /* RESID: gradient wrt res (NULL allowed if res NULL); IMAGE: dh */
  void* dx; int64_t lddx;       /* RESID: gradient wrt x (may alias dres when p_pre == 0) */
  float* partial;
  float* dword; float* dpos; float* dtt; float* dtt_ext; /* EMBED */
  int64_t nblk;   /* number of [3|4][H] slabs `partial` holds = gstvd_ln_bwd_blocks_for(M, H, mode); 0 = gstvd_ln_bwd_blocks(M) */
} gstvd_ln_bwd_t;
int64_t gstvd_ln_bwd_blocks(int64_t M);                                    /* upper bound for every (H, mode) */
int64_t gstvd_ln_bwd_blocks_for(int64_t M, int64_t H, int32_t mode);       /* the geometry gstvd_ln_bwd uses when told so via nblk */
int gstvd_ln_bwd(const gstvd_ln_bwd_t* p, gstvd_stream_t s);

/* LayerNorm folded into the Linear next to it, for latency-bound row counts (the decoder's M = rows x 25; csrc/gemm_rows.hip).
 * Both take a gstvd_gemm_t whose A operand is NOT read (A is produced in the kernel; M, K = the LayerNorm's M, H; bf16 operands,
 * batch 1, row-major or k-major B, every gstvd_gemm epilogue flag except COLSUM) and a RESID-mode LayerNorm descriptor:
 *   gstvd_gemm_ln_fwd   C = epi( LN(drop(x) + res) . B^T );  also writes ln->y / mean / rstd exactly as gstvd_ln_fwd would
 *                       (BertSelfOutput / BertOutput LayerNorm + the dense that reads it: transformers 4.16.2 modeling_bert,
 *                       call sites models/visual_dialog_decoder.py:300-311)
 *   gstvd_gemm_ln_bwd   C = epi( dx . B ), dx = what gstvd_ln_bwd(lb) writes to lb->dx (the LayerNorm's backward + the input
 *                       gradient of the Linear that produced the LayerNorm's input; autograd of the same modules); also writes
 *                       lb->dres, lb->dx and lb->partial = [ceil(M/R), 3, H] column partials, R = gstvd_gemm_ln_rows_per_block();
 *                       lb->nblk must be ceil(M/R)
 * Constraints: H = K a multiple of 64 and <= 768, M <= 1024, N >= 640 and N % 8 == 0.  GSTVD_E_UNSUPPORTED otherwise (the caller
 * runs the two kernels separately).  M <= 1024 is what the KERNEL accepts; the host policy (ops.gemm_ln_ok) only sends M <= 640
 * and at most 512 workgroups (one round of the chip) -- beyond that the two plain kernels are faster. */
int gstvd_gemm_ln_fwd(const gstvd_gemm_t* g, const gstvd_ln_t* ln, gstvd_stream_t s);
int gstvd_gemm_ln_bwd(const gstvd_gemm_t* g, const gstvd_ln_bwd_t* lb, gstvd_stream_t s);
int64_t gstvd_gemm_ln_rows_per_block(void);

/* out_j[c] (+)= sum_blk partial[blk, j, c]  for j in 0..nvec-1 ; out_j may be NULL (skipped) */
int gstvd_colsum_partials(const float* partial, int64_t nblk, int64_t nvec, int64_t H,
                          float* out0, float* out1, float* out2, int32_t accumulate, gstvd_stream_t s);

/* Batched form: every pending column reduction of a backward pass in ONE launch.  `table_dev` is a device
 * array of entries; entry i owns blocks [blk0_i, blk0_{i+1}) of the grid, one block per 64 output columns
 * of its nvec*H wide row; out_j[c] (+)= sum_{k<nblk} partial[k*stride + j*H + c]. */
typedef struct {
  const float* partial; float* out[3];
  int64_t nblk, stride, H;
  int32_t nvec, accumulate[3], blk0;
} gstvd_colsum_entry_t;
int gstvd_colsum_batched(const gstvd_colsum_entry_t* table_dev, int64_t nent, int64_t total_blocks, gstvd_stream_t s);
/* table-driven stage 1: entry i owns blocks [blk0_i, blk0_{i+1}), ceil(M/64) * ceil(N/256) blocks each;
 * scratch[slab, N] = sums over rows [64*slab, 64*slab+64) of x[M, N] (row stride ldx, all entries of one dtype) */
typedef struct { const void* x; float* scratch; int64_t ldx, M, N; int32_t blk0, pad_; } gstvd_slab_entry_t;
int gstvd_colsum_slabs_batched(const gstvd_slab_entry_t* table_dev, int64_t nent, int64_t total_blocks, int32_t dtype,
                               gstvd_stream_t s);
/* stage 1 of a plain column sum: scratch[slab, N] = per-64-row-slab sums of x[M, N] (reduce with the batched form) */
int gstvd_colsum_slabs(const void* x, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* scratch,
                       int64_t scratch_elems, gstvd_stream_t s);

/* out[c] (+)= sum_m x[m, c]   (bias gradients of QKV / FFN-up projections) */
int gstvd_colsum(const void* x, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* out,
                 float* scratch, int64_t scratch_elems, int32_t accumulate, gstvd_stream_t s);

/* dW_loc[h, j] (+)= sum_m dh[m,h] * loc[m,j]   (image_location_embeddings.weight grad) */
int gstvd_locgrad(const void* dh, int64_t lddh, const float* loc, int64_t M, int64_t H, int32_t dtype,
                  float* dw_loc, int32_t accumulate, gstvd_stream_t s);

/* ---- fused attention: softmax(Q K^T * scale + mask) V with dropout on the probabilities ------
 * (BertSelfAttention / BertImageSelfAttention / BertBiAttention, vilbert_dialog.py:380-407,507-534,
 *  646-712; HF 4.16.2 BertSelfAttention incl. the causal decoder mask and cross attention.)
 * Element (b, i, h, c) of Q lives at Q[(b*Lq + i)*ldq + h*d + c]; K, V, O, dO, dQ, dK, dV alike, so the
 * operands can be column slices of fused QKV buffers.
 * additive mask(b, i, j) = (key_mask[b, j] != 0 && (!causal || j <= i)) ? 0 : mask_neg;
 * LSE [B, nh, Lq] fp32 is saved by forward; backward recomputes P from it (nothing of size Lq x Lk is stored).
 * d in {32, 64, 128}.
 */
typedef struct {
  const void *Q, *K, *V; void* O; float* LSE;
  const float* key_mask;            /* [B, Lk] 1/0, or NULL = all ones */
  int64_t ldq, ldk, ldv, ldo;
  int32_t B, nh, Lq, Lk, d, causal, dtype;
  float mask_neg, scale, dropout_p; uint32_t site; const uint64_t* rng;
  /* backward only */
  const void* dO; int64_t lddo;
  void *dQ, *dK, *dV; int64_t lddq, lddk, lddv;
  float* delta;                      /* [B, nh, Lq] fp32 scratch: rowsum(dO * O) */
  int32_t kv_group;                  /* forward only: K, V and key_mask are shared by kv_group consecutive batch rows
                                        (their batch index is b / kv_group); 0 or 1 = one K/V per row.  Used to score the
                                        100 answer candidates of a dialog round against ONE encoder pass (evaluate_gen.py:45-92
                                        re-encodes the identical context 100 times). */
  int32_t q_bstride, kv_bstride;     /* forward only: rows between consecutive batch elements of Q/O resp. K/V (0 = Lq / Lk).
                                        Lets one decode step (Lq = 1) read a [B, Umax, H] K/V cache of which Lk rows are filled. */
  uint64_t* drop_bits;               /* (ABI 5) NULL, or the dropout keep bits of the probabilities: uint64
                                        [B, nh, ceil(Lq/16), ceil(Lk/16), 4] -- word r of tile (qt, kt): bit 16 g + i = "keep" of
                                        (query 16 qt + i, key 16 kt + 4 g + r).  Forward WRITES them when the pointer is set and
                                        dropout is on (the same draws it applies: nn.Dropout's mask, vilbert_dialog.py:398-401);
                                        the one-pass backward READS them instead of hashing every draw again (40 % of its vector
                                        instructions); the other backward kernels ignore them.  bf16, d = 64 only. */
} gstvd_attn_t;
int gstvd_attn_fwd(const gstvd_attn_t* a, gstvd_stream_t s);
int gstvd_attn_bwd(const gstvd_attn_t* a, gstvd_stream_t s);  /* dQ (+delta) then dK,dV */

/* ---- LM head loss: CrossEntropyLoss(ignore_index) of visual_dialog_decoder.py:70-77 ----------
 * logits [M, ldl >= V]; row_loss [M] (0 for ignored rows); stats (fp32[3], written by the call):
 * stats[0] = sum of row losses, stats[1] = number of non-ignored rows, stats[2] = mean loss
 * (= stats[0] / stats[1], NaN when every row is ignored, like torch); lse [M] saved. */
int gstvd_ce_fwd(const void* logits, int64_t ldl, const int64_t* labels, int64_t M, int64_t V,
                 int64_t ignore_index, int32_t dtype, float* row_loss, float* lse, float* stats,
                 gstvd_stream_t s);
/* dlogits[m, v] = (softmax - onehot) * gscale[0] / (mean ? stats[1] : 1) for kept rows, 0 otherwise;
 * columns V..ldd-1 are zero filled (they are the zero padded vocabulary rows of the LM head). */
int gstvd_ce_bwd(const void* logits, int64_t ldl, const int64_t* labels, const float* lse,
                 const float* stats, const float* gscale, int32_t mean, int64_t M, int64_t V,
                 int64_t ignore_index, int32_t dtype, void* dlogits, int64_t ldd, gstvd_stream_t s);
/* evaluate_gen.py:94-106: score[m] = sum_u [tgt != 0] * (logits[m,u,tgt] - lse[m,u]) with tgt = ids shifted left */
int gstvd_answer_scores(const void* logits, int64_t ldl, const float* lse, const int64_t* dec_ids,
                        int64_t rows, int64_t U, int32_t dtype, float* scores, gstvd_stream_t s);

/* One sampling step of the decode loop (models/visual_dialog_model.py:96-108 after the n-gram filter has produced `banned`;
 * utils/decoding_utils.py:4-35 for the top-k rule): z = logits / temperature (banned -> -inf); top_k > 0: z below the k-th
 * largest z -> -inf (ties with it stay); out[b * out_stride] = first index whose cumulative softmax probability reaches
 * u[b] * total (inverse CDF -- the draw the oracle substitutes for the reference's torch.multinomial, whose stream is device
 * specific).  logits [B, ld >= V] fp32 or bf16; banned: NULL or uint8 [B, banned_ld >= V]; u [B] in (0, 1).
 * ngram > 0 (ABI 5): the n-gram filter itself, utils/decoding_utils.py:38-77 (batch_ngram_blocking + _get_generated_ngrams), runs
 * in the same launch -- a token is banned when it would complete an n-gram that occurs in the row's history hist[b, 0..hist_T)
 * (int64 ids, row stride hist_ld; n-grams that contain one of the n_special ids in `special` are ignored) and whose first n-1
 * tokens are the row's last n-1 generated ids: ids_tm[(cur_len - (n-1) + j) * ids_stride + b], j < n-1, read from the TIME-MAJOR
 * id buffer [positions, ids_stride >= B].  Nothing is banned while cur_len < n-1 or hist_T < n (the reference's slice
 * semantics).  It composes with `banned` (either bans).
 * top_p in (0, 1) (ABI 6): nucleus filtering after top-k, utils/decoding_utils.py:22-34 -- a token stays when the softmax mass of
 * the tokens sorted in front of it (strictly larger logits) is <= top_p, i.e. the token that crosses top_p stays; equal logits
 * stay or go together (the reference leaves the order inside a tie to torch.sort).  0 or >= 1: off.  top_k is unbounded since
 * ABI 6 (k <= 16 walks the distinct values from the top, larger k bisects on the value). */
typedef struct {
  const void* logits; int64_t ld; int32_t dtype; int32_t B; int32_t V; int32_t top_k; float temperature;
  const float* u; int64_t* out; int64_t out_stride; const uint8_t* banned; int64_t banned_ld;
  const int64_t* hist; int64_t hist_ld; int32_t hist_T; int32_t ngram;
  const int64_t* ids_tm; int64_t ids_stride; int32_t cur_len; int32_t n_special; int32_t special[8];
  float top_p; int32_t reserved_;                                       /* ABI 6 */
} gstvd_sample_t;
int gstvd_sample_topk(const gstvd_sample_t* a, gstvd_stream_t s);

/* backward of VLFusion's concat + dropout (visual_dialog_model.py:132-133): d_enc [B, R+T, H] ->
 * d_v [B*R, H] (vision rows first) and d_t [B*T, H], each multiplied by the dropout mask its forward GEMM
 * epilogue applied (element index (b*R + r)*H + n under site_v, (b*T + t)*H + n under site_t). */
int gstvd_vl_split(const void* d_enc, int64_t B, int64_t R, int64_t T, int64_t H, int32_t dtype, void* d_v, void* d_t,
                   float p, uint32_t site_v, uint32_t site_t, const uint64_t* rng, gstvd_stream_t s);

/* ---- element-wise plumbing --------------------------------------------------------------------*/
int gstvd_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, gstvd_stream_t s);
/* (ABI 7) dst[e] = bf16(src[e]) for e in the union of `nranges` ranges of two flat buffers that share their indexing: ranges_dev =
 * device int64[nranges][2] (start, length; start % 4 == 0), blk0_dev = device int32[nranges + 1], the exclusive prefix sum of the
 * ranges' block counts ceil(length / 1024); total_blocks = blk0_dev[nranges].  The N > 1 gradient path: the weight-gradient
 * launch writes the bf16 all-reduce payload of every GEMM weight itself, this casts the REST of a slice (biases, LayerNorm, embedding
 * tables: ~5 % of it) instead of a pass over the whole fp32 gradient buffer (train_gen.py:324: the reduce-add of DataParallel). */
int gstvd_cast_ranges(const float* src, void* dst_bf16, const int64_t* ranges_dev, const int32_t* blk0_dev, int64_t nranges,
                      int64_t total_blocks, gstvd_stream_t s);
int gstvd_scale(float* x, const float* factor, int64_t n, gstvd_stream_t s);   /* x *= factor[0] */
int gstvd_rng_advance(uint64_t* rng, gstvd_stream_t s);                        /* rng[1] += 1 */
/* materialise a dropout mask (1/(1-p) or 0) for tests: out[e] for e in [0, n) */
int gstvd_dropout_mask(float* out, int64_t n, float p, uint32_t site, const uint64_t* rng, gstvd_stream_t s);

/* fused AdamW over flat fp32 buffers with pytorch_transformers-1.2.0 semantics (train_gen.py:16,247:
 * eps inside sqrt(v)+eps, bias correction, decoupled decay applied after the update) and, optionally,
 * refresh of the bf16 shadow weights in the same pass.  lr/wd are per-element-segment tables:
 * seg_end[i] is the exclusive end offset of segment i; hp[2*i] = lr, hp[2*i+1] = weight decay (lr 0 = skip).
 * The call updates flat elements [begin, n): the backward pipeline applies the optimizer slice by slice, as soon
 * as a slice's gradients are final, overlapped with the rest of backward. */
int gstvd_adamw(float* param, const float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2, float eps,
                const float* step /* device scalar, 1-based */, float grad_scale, int64_t begin, gstvd_stream_t s);
/* Same update with the gradient taken from a bf16 buffer: flat element i reads grad_bf16[i - grad_origin].  Used at N>1
 * when the gradient slice travels over RCCL in bf16: the all-reduced bf16 slice feeds AdamW directly instead of being
 * expanded back into the fp32 gradient buffer first (saves 8 B/param of HBM traffic per step). */
int gstvd_adamw_bf16grad(float* param, const void* grad_bf16, int64_t grad_origin, float* m, float* v, void* shadow_bf16,
                         int64_t n, const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2, float eps,
                         const float* step, float grad_scale, int64_t begin, gstvd_stream_t s);

/* The update of a chosen set of 1024-element blocks of the flat buffers: block_list_dev[b] = index of a block (elements
 * [1024 k, 1024 k + 1024), absolute), nblocks of them; of those, elements outside [begin, n) and elements of segments with
 * seg_skip_dev[segment] != 0 are left alone (seg_skip_dev NULL = none).  It is the remainder pass behind
 * gstvd_gemm_grouped_adamw: biases, LayerNorm and embedding parameters, and any weight whose gradient was accumulated from
 * several producers. */
int gstvd_adamw_blocks(float* param, const float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                       const int64_t* seg_end, const float* hp, int64_t nseg, float beta1, float beta2, float eps,
                       const float* step, float grad_scale, int64_t begin, const int32_t* block_list_dev, int64_t nblocks,
                       const uint8_t* seg_skip_dev, gstvd_stream_t s);

/* Weight gradients and their AdamW update as ONE launch (single-GPU training: no all-reduce stands between a gradient and its
 * update; train_gen.py:324-329 loss.backward(); optimizer.step(); optimizer.zero_grad()).  A grouped launch like
 * gstvd_gemm_grouped (bf16 operands, both k-major, fp32 accumulators); for every problem that carries GSTVD_EPI_ADAMW the
 * tile's epilogue does not store dW but runs the update of gstvd_adamw on its 256 x 256 weights straight from the
 * accumulators: g = alpha * acc * grad_scale; reads param / m / v, writes param / m / v and the bf16 shadow weights -- 26
 * bytes per weight instead of 4 (dW out) + 30 (AdamW pass), and the HBM-bound update runs under the MFMA-bound K-loops of the
 * other workgroups instead of after them.  Such a problem's C must point INTO the flat gradient buffer `grad_base` (its flat
 * index addresses param / m / v / shadow), with ldc % 4 == 0 and a flat index % 4 == 0; its `addend` field carries the DEVICE
 * address of the weight's (lr, weight decay) pair -- two consecutive floats, e.g. &hp[2 * segment] of gstvd_adamw's table, read at
 * run time so that a schedule step needs no new table; it carries no other epilogue flag than COLSUM / COLSUM_ACC.  write_grad
 * != 0 also stores dW (the `.grad` the caller may want to look at).  Problems without the flag get the plain epilogue.  Same
 * arithmetic per element as gstvd_gemm_grouped followed by gstvd_adamw: results are bit-identical. */
typedef struct {
  const float* grad_base; float* param; float* m; float* v; void* shadow_bf16;
  const float* step; float beta1, beta2, eps, grad_scale;
  int32_t write_grad;
} gstvd_adamw_fuse_t;
int gstvd_gemm_grouped_adamw(const gstvd_gemm_t* table_dev, const int32_t* tile_off_dev, int64_t nprob, int64_t total_tiles,
                             const gstvd_adamw_fuse_t* f, const int32_t* block_map_dev, int64_t nblocks, gstvd_stream_t s);
/* measurement support: the (mangled) symbol of the kernel gstvd_gemm_grouped_adamw launches (rocprofv3 traces key on it) */
int gstvd_gemm_grouped_adamw_kernel_name(char* buf, int32_t buf_len);

/* Decode step (one token per row, models/visual_dialog_model.py:87-92 with the KV cache of this build): the LayerNorm that
 * closes a BERT sub-layer (transformers 4.16.2 BertSelfOutput / BertOutput, eps 1e-12) folded into the Linear that consumes it:
 * C = epi(LN(A; gamma, beta, eps) . B^T) for M <= 16 rows, K <= 1024, bf16 operands; epilogue flags BIAS / ADD / GELU as in
 * gstvd_gemm.  y_out (or NULL): bf16 [M, ldy >= K] receives LN(A), the residual input of the sub-layer's closing Linear.
 * GSTVD_E_UNSUPPORTED for any other shape (the caller then runs gstvd_ln_fwd + gstvd_gemm). */
int gstvd_gemv_ln(const gstvd_gemm_t* g, const float* gamma, const float* beta, float eps, void* y_out, int64_t ldy, gstvd_stream_t s);
/* bit 0: gstvd_gemm_grouped honours GSTVD_EPI_COLSUM (the producer / consumer kernel is the one it launches) */
int32_t gstvd_gemm_group_caps(void);
/* Measurement support: the (mangled) symbol of the device kernel that gstvd_gemm (splits <= 1) or gstvd_gemm_splitk
 * (splits >= 2) would launch for this descriptor -- the dispatch runs, the launch is replaced by recording its target.
 * bench.py's roofline.kernel comes from here.  Nothing is launched; pointers in the descriptor are not dereferenced. */
int gstvd_gemm_kernel_name(const gstvd_gemm_t* g, int32_t splits, char* buf, int32_t buf_len);
int gstvd_gemm_grouped_kernel_name(int32_t dtype_in, int32_t dtype_out, int32_t a_kmajor, int32_t b_kmajor, char* buf, int32_t buf_len);

#ifdef GSTVD_DIAG
/* DIAGNOSTIC BUILD ONLY (lib/libgstvd_hip_diag.so, `make -C gst_visdial_amd/csrc diag`; the product library neither exports this
 * symbol nor contains the timing ablations it serves): copies the in-kernel clock stamps that the GSTVD_GEMM_ST=3 variant of the
 * 256x256 GEMM tile leaves behind -- per workgroup {shader-clock ticks, 100 MHz wall ticks, K steps, 0} around its K loop -- to
 * host memory.  Evidence for DESIGN.md's "what clock does an MFMA-dense loop hold" (MI355X_MICROARCH.md, DVFS give-back item 6);
 * replaces nothing in the reference. */
int gstvd_debug_gemm_clock(uint64_t* out_host, int32_t n_words);
#endif

#ifdef __cplusplus
}
#endif
#endif /* GSTVD_HIP_H */
