/* evo_mi355x.h -- C ABI of libevo_mi355x.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * StripedHyena forward that evo-design/evo runs through `stripedhyena==0.2.2` + FlashAttention-2.
 *
 * The reference has no C FFI: its plugin surface is the Python API of `stripedhyena` as consumed by
 * `evo`.  Each entry point below replaces the vendor/third-party kernel(s) that the
 * reference reaches for one step of that forward; the reference-side call site that pins the step is
 * cited per function.  `evo_amd/ops.py` is the ctypes binding; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions (every function):
 *   - raw DEVICE pointers (tensor.data_ptr()); bf16 = uint16 storage; "c64" = interleaved {re,im} f32;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no allocation, no host sync, no global state: safe to capture in a hipGraph; workspace is
 *     passed in by the caller;
 *   - return value is a hipError_t cast to int (0 = hipSuccess); -1 = bad argument.  The Python
 *     binding raises on non-zero.
 */
#ifndef EVO_MI355X_H
#define EVO_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version; bumped when a signature changes (the binding refuses a library whose version differs).
 *   2: evo_embed_bf16 gained `bad_flag`; evo_hyena_seg_state / evo_hyena_apply gained `mask`;
 *      evo_unembed_logprob_bf16 and evo_hyena_mfma added.
 *   3: evo_rope_append_decode_bf16 added; the fused decode launches take up to 8 rows at K = 4096.
 *   4: evo_hyena_mfma gained the carry-in state `s0`, the end state `s_out` and `poles`; evo_hyena_mfma_state added.
 *   5: evo_mlp_gate_mfma_bf16 (GELU * gate in the dense layer's epilogue), evo_linear_zg_mfma_bf16 (group-major result) and
 *      evo_hyena_mfma_zg (the single-pass operator on group-major z) added.
 *   6: evo_hyena_cs_zg (the single-pass operator with channel-stationary waves: outputs, end state or state-only walk; blocked y)
 *      and evo_linear_xblk_mfma_bf16 (the output projection on that blocked y) added.
 *   7: evo_hyena_ct (the same operator on channel-major z^T: no input window in LDS), evo_linear_t_mfma_bf16 (the projection with
 *      a transposed result) and evo_rmsnorm_rows_bf16 (RMSNorm with padded batch rows) added.
 *   8: evo_attn_fwd_causal_bf16 gained `vt_ws` (workspace for V^T: the round-5 prefill kernel of csrc/attn_w64.hip reads its V
 *      fragments from a transposed copy); evo_hyena_mfma, evo_hyena_mfma_state, evo_hyena_mfma_zg, evo_hyena_cs_zg and
 *      evo_linear_zg_mfma_bf16 REMOVED (the earlier forms of the single-pass Hyena operator and the group-major projection that
 *      fed them: every caller is on evo_hyena_ct).
 *   9: the RMSNorm passes folded into the dense layers around them: evo_linear_mfma_nf_bf16, evo_linear_xblk_mfma_nf_bf16,
 *      evo_mlp_gate_mfma_nf_bf16, evo_linear_t_mfma_nf_bf16 (the same launches with a per-row factor in / the rows' sums of squares
 *      out) and evo_rms_finalize_f32 added; no signature changed.
 *  10: evo_probe_copy_f4 and evo_probe_mfma_bf16 added (the box-calibration probes of bench.py's `box` block); evo_mlp_gate_small_m_bf16 and
 *      evo_norm_mlp_gate_small_m_bf16 gained `grouped` (the decode launches read l1 | l2 in the gated MFMA launch's row order: one weight set), evo_mlp_gate_small_m_bf16
 *      takes 5-64 rows on an MFMA form;
 *      evo_rope_qk_bf16 and evo_rope_append_decode_bf16 gained `q_scale`, evo_attn_fwd_causal_bf16 / evo_attn_decode_bf16 accept
 *      softmax_scale <= 0 = "queries pre-scaled" (the prefill attention kernel without its per-score multiply); evo_linear_small_m_bf16
 *      takes up to 64 rows and an optional workspace (`ws`, `ws_bytes`: split over K across workgroups for the narrow layers); evo_hyena_ct gained `y_row_pitch` (rows of y between two batch rows: the scoring path runs the 512 k main tokens
 *      of every row through the operator and the one token behind them through the single-token launch, see below). */
#define EVO_ABI_VERSION 10
int evo_abi_version(void);

/* ---- embedding gather ------------------------------------------------------------------------
 * replaces VocabParallelEmbedding.embed (ATen gather)    [REF evo/scoring.py:81; evo/models.py:136]
 * ids [n_tok] int64, weight [vocab, D] bf16 -> out [n_tok, D] bf16.  An id outside [0, vocab) never
 * indexes the table: its output row is zeros and *bad_flag (device int, may be NULL; the caller zeroes
 * it) is set to 1 -- the binding raises IndexError where F.embedding would device-assert. */
int evo_embed_bf16(const int64_t* ids, const void* weight, void* out,
                   int64_t n_tok, int64_t D, int64_t vocab, int* bad_flag, void* stream);

/* ---- RMSNorm -----------------------------------------------------------------------------------
 * replaces the eager RMSNorm chain (norm, div, mul)      [REF evo/configs/evo-1-8k-base_inference.yml:13,31]
 *   out = scale * x / (||x||_2 * D^-1/2 + eps)            (eps OUTSIDE the root)
 * x [M, D] bf16 -> out [M, D] bf16, fp32 accumulation, single output rounding.
 * If `bias` != NULL the row is first updated in place, x <- bf16(x + bias), and the norm is taken of
 * the updated row (this folds the out_proj / out_filter_dense bias add and the residual write). */
int evo_rmsnorm_bf16(void* x, const void* bias, const void* scale, void* out,
                     int64_t M, int64_t D, float eps, void* stream);

/* ---- Hyena operator, parallel (prefill / scoring) form ------------------------------------------
 * replaces HyenaInferenceEngine.parallel_fir + ParallelHyenaFilter.compute_filter + parallel_iir
 * (+ prefill_via_modal_fft)                               [REF evo/configs/evo-1-8k-base_inference.yml:8,10,14,33,37;
 *                                                          evo/generation.py:111-114]
 * Three launches over a (batch, head, time-segment) decomposition; the long convolution
 *   y_t = sum_{j<=t} h_{t-j} x1v_j,  h_k = Re sum_s R_s p_s^k
 * is evaluated EXACTLY through its 8 complex modes (S_t = p S_{t-1} + x1v_t; y_t = Re sum R_s S_t),
 * so no filter and no FFT buffer ever touches HBM.
 *
 *   z        [B, T, 3*D] bf16   projections output, channel-last; channel c = h*3*hd + g*hd + j,
 *                               g in {0:x2, 1:x1, 2:v}  (column split, hd = D / n_heads = 128)
 *   z_halo   [B, 2, 3*D] bf16 or NULL: rows t=-2,-1 (sequence-parallel halo; NULL = zeros)
 *   fir_w    [3*D, 3] bf16      short_filter_weight (cross-correlation taps; tap 2 hits z_t)
 *   fir_b    [3*D]    bf16      short_filter_bias
 *   poles, residues [D, 8, 2] f32
 *   dskip    [D] bf16           filter.D
 *   seg_len                     time-segment length C (multiple of 4); n_seg = ceil(T / C)
 *   agg      [B, n_seg, D, 8] c64 workspace:  (1) seg_state writes each segment's end state from a
 *                               zero start; (2) carry_scan rewrites it in place into the state ENTERING
 *                               each segment; (3) apply reads it.
 *   s0       [B, D, 8] c64 or NULL: state entering t=0 (sequence-parallel carry-in / resumed prefill)
 *   s_final  [B, D, 8] c64 or NULL: state after t=T-1  (== upstream prefill_via_modal_fft)
 *   y        [B, T, D] bf16     (y_conv + x1v * dskip) * x2, channel-last
 *   mask     [B, T] uint8 or NULL: upstream's `padding_mask` (1 = token, 0 = pad): the FIR output (x2, x1, v) of
 *                               a padded position is zero, as engine.parallel_fir multiplies it.  evo never passes one.
 */
int evo_hyena_seg_state(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                        const float* poles, float* agg, const uint8_t* mask,
                        int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len, void* stream);
int evo_hyena_carry_scan(float* agg, const float* poles, const float* s0, float* s_final,
                         int64_t B, int64_t T, int64_t D, int64_t seg_len, void* stream);
/* sequence-parallel fix-up (new; the reference has no multi-GPU path): `agg` already scanned with a zero
 * carry-in gets p^(k*seg_len) * s0 added to segment k once the state s0 entering this shard is known. */
int evo_hyena_carry_add(float* agg, const float* poles, const float* s0,
                        int64_t B, int64_t T, int64_t D, int64_t seg_len, void* stream);
int evo_hyena_apply(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                    const float* poles, const float* residues, const void* dskip,
                    const float* agg, void* y, const uint8_t* mask,
                    int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len, void* stream);

/* ---- Hyena operator, single-pass matrix-core form (scoring, cached prefill, sequence-parallel shards) -------
 * replaces the same reference functions as the three launches above (parallel_fir + compute_filter +
 * parallel_iir, and prefill_via_modal_fft for `s_out`)     [REF evo/configs/evo-1-8k-base_inference.yml:8,10,14,33,37;
 *                                                            evo/generation.py:111-117,152 for the cached form]
 * One launch, z read once, y written once (csrc/hyena_ct.hip; rounds 2-4 shipped three earlier forms of it -- token-major,
 * group-major with channel-stationary waves -- which round 5 retired: this is the only single-pass kernel).  A wave owns two
 * channels for a 512-step tile: its lanes' FIR outputs ARE the B operands of the block-Toeplitz MFMAs
 * (v_mfma_f32_16x16x32_bf16, operands split into bf16 hi + lo terms, fp32 accumulation), the 16 blocks' modal states meet in a
 * DPP scan in fp32, the carry product runs on the matrix cores with hi/lo-split states.
 *   zt     CHANNEL-MAJOR z: [zt_pitch / 256][3 D][256] bf16, the projection's result TRANSPOSED and stored in blocks of 256
 *          positions -- column c = h*3*hd + g*hd + j of z (the reference's order, no regrouping), position p at element
 *          ((p / 256) * 3 D + c) * 256 + p % 256; written by evo_linear_t_mfma_bf16.  Batch row b, token t sits at position
 *          zt_row0 + b * row_pitch + t.  A lane's eight steps of one channel are 16 consecutive bytes, loaded straight into the
 *          registers the FIR reads: no window in LDS, no DMA, no bank conflicts.  row_pitch % 8 == 0, zt_row0 % 8 == 0,
 *          zt_pitch % 256 == 0, zt_pitch * 3 D * 2 < 4 GiB, zt 16-byte aligned; positions between T and row_pitch may hold anything.
 *   tail_T != 0 (the "tail form", T = 512 k + r with r <= 8: tail_T = 512 k): tokens t >= tail_T of batch row b sit at position
 *          tail_pos0 + 8 b + (t - tail_T) instead (a tail block behind the main area, filled by the weight-streaming dense layer:
 *          rows of 512 k positions need no padding and the projection no extra round of tiles; evo_amd/ops.py zt_layout).
 *   z_halo [B, 2, 3 D] bf16 in the reference's column order (rows = steps -2, -1) or NULL
 *   table  [D, 52, 64] u32: per-channel MFMA operand constants (evo_amd/hyena_tables.py mfma_operand_table; filter.D folded into
 *          the block-Toeplitz diagonal)
 *   s0     [B, D, 8] c64 or NULL: modal state entering t = 0 (resumed prefill / sequence-parallel carry-in)
 *   s_out  [B, D, 8] c64 or NULL: state after t = T-1; needs `poles` [D, 8] c64 (fp32 pairs).  T % 512 == 0: the carry of the last tile, stored
 *          from registers; otherwise the last block's steps are walked from the state entering it (fp32 recurrence, ~half a tile's time per row)
 *   state_only != 0: no y (may be NULL), only s_out -- stage 1 of a sequence-parallel shard, whose end state from a zero carry-in
 *          goes to the other ranks before anybody can finish its outputs (new; the reference has no multi-GPU path)
 *   y      [B, T, D] bf16, or with y_blocked_rows != 0 BLOCKED: [ceil(y_blocked_rows / 128)][D / 16][128][16] bf16 -- the
 *          [y_blocked_rows, D] matrix with a group's 16 channels of 128 consecutive rows kept together (whole cache lines per
 *          store; evo_linear_xblk_mfma_bf16 reads it); batch row b, token t is row y_row0 + b y_row_pitch + t of that matrix (y_row0 > 0:
 *          the row groups of a sequence-parallel shard write into one tensor)
 *   y_row_pitch (ABI 10): rows of y between two batch rows, 0 = T.  > T: the caller owns rows T .. y_row_pitch - 1 behind every batch
 *          row (T = 512 k + 1 scoring batches: the operator walks the 512 k tokens in whole tiles and returns s_out, the last token of every
 *          row is one evo_hyena_decode_fused_small_m step from that state -- a ragged tile with ONE valid step costs a full tile's issue
 *          time, 8 of 136 tile steps at 8 x 8,193)
 *   no mask (padding_mask shapes take the three-launch form).  D == n_heads * 128. */
int evo_hyena_ct(const void* zt, const void* z_halo, const void* fir_w, const void* fir_b, const void* table, void* y,
                 const float* s0, float* s_out, const float* poles, int64_t B, int64_t T, int64_t D, int64_t n_heads,
                 int64_t zt_pitch, int64_t row_pitch, int64_t zt_row0, int64_t tail_T, int64_t tail_pos0, int64_t state_only,
                 int64_t y_blocked_rows, int64_t y_row0, int64_t y_row_pitch, void* stream);

/* The Hyena block's output projection on the blocked y of evo_hyena_ct               [REF stripedhyena/model.py ParallelGatedConvBlock:
 * out_filter_dense]:  y [M, N] = x . w^T (+ bias [N]) (+ residual [M, N], may alias y), x = [M / 128][K / 16][128][16] bf16.  The persistent dense
 * layer of evo_linear_mfma_bf16 with other source addresses for its X tiles; M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128. */
int evo_linear_xblk_mfma_bf16(const void* x_blk, const void* w, const void* bias, const void* residual, void* y,
                              int64_t M, int64_t N, int64_t K, void* stream);

/* The Hyena projection with a transposed result              [REF stripedhyena/model.py ParallelGatedConvBlock.forward: projections]:
 * zt [Mp / 256][N][256] bf16 = (x [Mp, K] . w [N, K]^T + bias [N])^T in blocks of 256 positions (an output tile of the kernel = one
 * contiguous 128 KiB piece) -- the persistent dense layer of evo_linear_mfma_bf16 launched with its operands swapped (rows of the
 * result = output features, columns = tokens; the bias runs along the rows; the tile raster mirrored), bit-identical to that
 * layer's result, transposed.  Mp % 256 == 0 (the caller pads x: see evo_rmsnorm_rows_bf16), N % 256 == 0, K % 64 == 0, K >= 128. */
int evo_linear_t_mfma_bf16(const void* x, const void* w, const void* bias, void* zt, int64_t Mp, int64_t N, int64_t K, void* stream);

/* evo_rmsnorm_bf16 with the output rows in the order of a channel-major z^T: token t < Tm of batch row b (row b * T + t of x [M = B T, D])
 * -> row b * Tp + t of out (Tp >= Tm; the pad rows are not written), the last T - Tm tokens of a row -> the compact tail rows
 * tail0 + b * (T - Tm) + (t - Tm).  Tm = T: no tail.  Same arithmetic, same bits per row. */
int evo_rmsnorm_rows_bf16(void* x, const void* bias, const void* scale, void* out, int64_t M, int64_t D, float eps,
                          int64_t T, int64_t Tp, int64_t Tm, int64_t tail0, void* stream);

/* ---- Hyena operator, recurrent (decode) form -----------------------------------------------------
 * replaces step_fir + step_iir                             [REF evo/generation.py:111-114,138-155]
 *   z_t [B, 3D] bf16; fir_state [B, 3D, 2] bf16 (in/out, oldest first); iir_state [B, D, 8] c64 (in/out)
 *   y [B, D] bf16 */
int evo_hyena_step(const void* z_t, void* fir_state, float* iir_state,
                   const void* fir_w, const void* fir_b, const float* poles, const float* residues,
                   const void* dskip, void* y, int64_t B, int64_t D, int64_t n_heads, void* stream);

/* ---- rotary embedding -----------------------------------------------------------------------------
 * replaces flash_attn's Triton rotary kernel               [REF evo/configs/evo-1-131k-base_inference.yml:39-40]
 * NeoX (non-interleaved) pairs (i, i+hd/2), in place on the q and k thirds of a packed
 * qkv [B, T, 3, H, hd] bf16.  cos/sin [T, hd/2] f32 are host-built for absolute positions
 * pos0..pos0+T-1 (already divided by the interpolation factor, already rounded to bf16 values).
 * q_scale (ABI 10; > 0): the rotated QUERY rows are multiplied by it before their one rounding -- the caller passes
 * softmax_scale * log2(e) and then calls evo_attn_fwd_causal_bf16 / evo_attn_decode_bf16 with softmax_scale = 0 ("queries pre-scaled":
 * a score is an exponent, the prefill kernel's per-score multiply disappears); 1.0 = plain rotary (exact: the bits of ABI 9). */
int evo_rope_qk_bf16(void* qkv, const float* cos_t, const float* sin_t,
                     int64_t B, int64_t T, int64_t H, int64_t hd, float q_scale, void* stream);

/* decode form: rotary on the q and k rows of ONE token per stream, each at its own position, and the append of (k, v) to
 * the KV cache, in one launch                                  [REF evo/generation.py:138-155; flash_attn_with_kvcache's
 *                                                              rotary + cache-update arguments]
 *   qkv [B, 3, H, hd] bf16 contiguous (q and k rotated in place); kv = the cache [.., cap, 2, H, hd] bf16 with element strides
 *   (batch, token, k|v, head), each a multiple of 8; pos [B] device int64: row b is written at token pos[b] (< cap: the caller's
 *   bound); inv_freq [hd/2] f32; angle = (pos / scaling) * inv_freq, cos / sin rounded to bf16 values as the cached tables are.
 *   Bit-identical to evo_rope_qk_bf16 with a table for those positions followed by the indexed copy. */
int evo_rope_append_decode_bf16(void* qkv, void* kv, const int64_t* pos, const float* inv_freq, float scaling,
                                int64_t B, int64_t H, int64_t hd, int64_t kv_sb, int64_t kv_st, int64_t kv_sw, int64_t kv_sh,
                                float q_scale, void* stream);

/* ---- causal multi-head attention forward ------------------------------------------------------------
 * replaces flash_attn_2_cuda fwd / flash_attn_with_kvcache  [REF README.md:47-50; evo/configs/evo-1-8k-base_inference.yml:9,30]
 * MFMA 32x32x16 bf16 tiles, online softmax in fp32, head dim 128 only.
 *   q [B, Tq, H, 128], k/v [B, Tk, H, 128] bf16 with explicit element strides (batch, token, head);
 *   o [B, Tq, H, 128] bf16 contiguous.  Query i may see key j iff j <= i + q_pos0 (q_pos0 = absolute
 *   position of query 0 minus absolute position of key 0).
 *   vt_ws: caller-owned workspace of B * H * 128 * (Tk rounded up to 64) bf16, or NULL.  Query ranges longer than 128 rows run
 *   the 64-rows-per-wave kernel (csrc/attn_w64.hip: one wave per SIMD, K / V tiles through LDS-DMA rings), which reads V^T:
 *   a pre-pass launch writes it into vt_ws.  With vt_ws == NULL (or Tq <= 128) the kernels of rounds 1-4 run (csrc/attn.hip).
 *   softmax_scale <= 0 (ABI 10): the queries carry softmax_scale * log2(e) already (evo_rope_qk_bf16's q_scale) -- scores are taken as
 *   exponents in the log2 domain; same for evo_attn_decode_bf16. */
int evo_attn_fwd_causal_bf16(const void* q, const void* k, const void* v, void* o,
                             int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t q_pos0,
                             int64_t q_sb, int64_t q_st, int64_t q_sh,
                             int64_t k_sb, int64_t k_st, int64_t k_sh,
                             int64_t v_sb, int64_t v_st, int64_t v_sh,
                             float softmax_scale, void* vt_ws, void* stream);

/* decode form (one query per sequence, Tq = 1): split-K over the key range ("flash-decoding") + combine.
 * replaces flash_attn_with_kvcache                           [REF evo/generation.py:109-110,138-155]
 *   q [B, 1, H, 128] (q_sb, q_sh strides), k/v views of the KV cache [B, cap, H, 128];
 *   dyn_pos: NULL -> the query sits at position Tk-1 and sees keys [0, Tk);  non-NULL -> device int64 [B]:
 *            row b's query sits at p_b = dyn_pos[b] and sees keys [0, p_b] (Tk is then only the cache capacity
 *            bound).  The launch no longer depends on the positions, so a captured hipGraph can be replayed for
 *            every token, and rows may be at DIFFERENT positions (continuous batching of decode streams);
 *   part_o [B, H, n_splits, 128] f32 and part_ml [B, H, n_splits, 2] f32: caller-owned workspace;
 *   n_splits <= 1024.  A bandwidth kernel (the KV cache is read once, 512 * H bytes per key): a WAVE owns a split = every
 *   n_splits-th 64-key block, requests the 32 KiB of a block at once and needs neither LDS nor barriers. */
int evo_attn_decode_bf16(const void* q, const void* k, const void* v, void* o,
                         int64_t B, int64_t H, int64_t Tk,
                         int64_t q_sb, int64_t q_sh,
                         int64_t k_sb, int64_t k_st, int64_t k_sh,
                         int64_t v_sb, int64_t v_st, int64_t v_sh,
                         const int64_t* dyn_pos, float* part_o, float* part_ml, int64_t n_splits,
                         float softmax_scale, void* stream);

/* ---- skinny dense layer (decode) ----------------------------------------------------------------------
 * replaces cuBLAS GEMV-shaped nn.Linear calls of the single-token forward   [REF evo/generation.py:151-155]
 * y [M, N] = x [M, K] . w [N, K]^T (+ bias [N]) (+ residual [M, N]);  1 <= M <= 64 (ABI 10; 16 before), K % 8 == 0 (K % 32 == 0
 * for M > 8), all bf16, fp32 accumulate, one rounding.  Weight-streaming (HBM-bound) forms: dot2 on the VALU up to
 * M = 4, v_mfma_f32_16x16x32_bf16 with the batch rows on the MFMA's N side from M = 5 -- one to four tiles of 16 rows per weight pass
 * (17-64 rows: the pooled decode step of 17-64 live streams); `residual` may alias `y`.  From 5 rows on layers with N >= 8192 and K % 256 == 0 the
 * weight is requested in whole 512-byte row pieces and the x rows of a 256-k chunk are shared by the workgroup's waves through LDS (csrc/gemv.hip
 * skinny_nw_kernel).  `ws` (ABI 10; may be NULL): a 16-byte aligned fp32 workspace of `ws_bytes` >= 8 M N 4 bytes lets the narrow layers (N < 8192) at
 * 17-64 rows split K over workgroups -- partial sums through ws, added in a fixed order by a second small launch; without it they run the k-split form. */
int evo_linear_small_m_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                            int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes, void* stream);

/* ---- dense layer on the MFMA pipe (prefill) --------------------------------------------------------------
 * replaces the cuBLAS nn.Linear GEMMs of the attention block: Wqkv (with bias) and out_proj (+ residual)
 *                                                     [REF stripedhyena/model.py:52-59 (MHA projections), :84-92]
 * y [M, N] = x [M, K] . w [N, K]^T (+ bias [N]) (+ residual [M, N]); all bf16, fp32 accumulate on
 * v_mfma_f32_32x32x16_bf16, one rounding.  N % 256 == 0, K % 64 == 0, any M >= 1 (ragged last row tile);
 * `residual` may alias `y`; `x` must not alias `y`.  Returns -1 for an unsupported shape. */
int evo_linear_mfma_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                         int64_t M, int64_t N, int64_t K, void* stream);

/* ---- gated MLP, first half, on the MFMA pipe (prefill) ----------------------------------------------------------
 * replaces l1 / l2 (two nn.Linear GEMMs) + F.gelu + the elementwise product of ParallelGatedMLP.forward
 *                                                     [REF stripedhyena/layers.py ParallelGatedMLP: l3(gelu(l1 x) * l2 x)]
 * a [M, I] = gelu(x [M, K] . W1^T) * (x . W2^T), all bf16; z1 = x W1^T and z2 = x W2^T are rounded to bf16 (the dense layers'
 * outputs in the reference), the exact-erf gate is evaluated in fp32 and rounded once -- the arithmetic of evo_gelu_gate_bf16
 * behind evo_linear_mfma_bf16, bit for bit, without the [M, 2 I] intermediate.  `w12g` = the 2 I rows of [W1; W2] regrouped in
 * blocks of 64: rows 32 q .. 32 q + 31 of W1 followed by the same rows of W2 (q = 0 .. I / 32 - 1).
 * (2 I) % 256 == 0, K % 64 == 0, K >= 128, any M >= 1.  Returns -1 for an unsupported shape. */
int evo_mlp_gate_mfma_bf16(const void* x, const void* w12g, void* a, int64_t M, int64_t I, int64_t K, void* stream);

/* ---- RMSNorm folded into the dense layers around it (prefill-sized batches) ---------------------------------------------
 * replaces the RMSNorm pass between two dense layers          [REF stripedhyena/model.py: pre_norm / post_norm of every block;
 *                                                              stripedhyena/layers.py RMSNorm.forward]
 * The reference writes n = bf16(g * x / (rms(x) + eps)) and multiplies it by the next layer's weight W.  Here the dense layer that
 * WRITES the residual stream x also emits each row's sum of squares, and the layer that consumes the norm reads x itself:
 *     W n  =  r_m * ((W diag(g)) x_m),   r_m = 1 / (rms(x_m) + eps)
 * with W diag(g) a bf16 copy of the weight the caller folds once (one rounding of the weight where the reference rounds the
 * activation) and r_m applied to the fp32 accumulators before bias / gate / the one output rounding.
 *   sumsq  [N / 128][ss_ld] fp32 out: per 128-column strip of the stored rows, the sum of squares of the ROUNDED values of row m
 *          (ss_ld >= M rounded up to 256; rows beyond M are scratch).  Needs `residual` (the launches that write the stream).
 *   row_scale [M rounded up to 256] fp32 in: r_m.  Launches without a residual.
 *   evo_rms_finalize_f32: rstd[m] = 1 / (sqrt(sum over strips) / sqrt(D) + eps) for m < M_main (strip order: bit-reproducible), and
 *          computed from the rows of x [M, D] bf16 themselves for M_main <= m < M (the rows a weight-streaming launch wrote).
 *   evo_linear_t_mfma_nf_bf16: additionally reads its token rows from the stream in (batch row, token) order: position p = b Tm + t
 *          of z^T (Mp = B Tm, Tm % 256 == 0) is row p + b row_skip of x [x_rows, K] -- the tail form of z^T; row_scale has x_rows
 *          entries.  row_scale == NULL: the plain launch (x_rows = Tm = Mp, row_skip = 0).
 * Shape contracts as the plain entries'; K >= 128 and operands below 4 GiB (the persistent kernel). */
int evo_linear_mfma_nf_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                            const float* row_scale, float* sumsq, int64_t ss_ld, int64_t M, int64_t N, int64_t K, void* stream);
int evo_linear_xblk_mfma_nf_bf16(const void* x_blk, const void* w, const void* bias, const void* residual, void* y,
                                 float* sumsq, int64_t ss_ld, int64_t M, int64_t N, int64_t K, void* stream);
int evo_mlp_gate_mfma_nf_bf16(const void* x, const float* row_scale, const void* w12g, void* a, int64_t M, int64_t I, int64_t K, void* stream);
int evo_linear_t_mfma_nf_bf16(const void* x, const float* row_scale, const void* w, const void* bias, void* zt, int64_t Mp, int64_t N,
                              int64_t K, int64_t x_rows, int64_t Tm, int64_t row_skip, void* stream);
int evo_rms_finalize_f32(const float* sumsq, int64_t n_strips, int64_t ss_ld, const void* x, int64_t M_main, int64_t M, int64_t D,
                         float eps, float* rstd, void* stream);

/* ---- Hyena mixer input of one decode step, fused ---------------------------------------------------------------
 * replaces pre-norm + projections GEMV + step_fir + step_iir of the single-token forward   [REF evo/generation.py:111-114,138-155]
 * x [M, D] bf16 residual rows (M = batch, 1 <= M <= 4; up to 8 at D = 4096), norm_scale [D], proj_w [3D, D], proj_b [3D] -> y [M, D] bf16;
 * fir_state [M, 3D, 2] bf16 and iir_state [M, D, 8] complex64 are updated in place.  Bit-identical to
 * evo_norm_linear_small_m_bf16 followed by evo_hyena_step. */
int evo_hyena_decode_fused_small_m(const void* x, const void* norm_scale, const void* proj_w, const void* proj_b,
                                   void* fir_state, float* iir_state, const void* fir_w, const void* fir_b,
                                   const float* poles, const float* residues, const void* dskip, void* y,
                                   int64_t M, int64_t D, int64_t n_heads, float eps, void* stream);

/* ---- RMSNorm + dense layer, decode form ----------------------------------------------------------------------
 * replaces the pre-mixer RMSNorm and the projection GEMV of the single-token forward     [REF evo/generation.py:151-155;
 *                                                                                   evo/configs/evo-1-8k-base_inference.yml:13]
 * y [M, N] = bf16(scale * x / (rms(x) + eps)) . w [N, K]^T (+ bias [N]);  1 <= M <= 4 (up to 8 at K = 4096), K % 8 == 0,
 * all bf16.  The
 * normalised row is bit-identical to what evo_rmsnorm_bf16 stores (same reduction order). */
int evo_norm_linear_small_m_bf16(const void* x, const void* scale, const void* w, const void* bias, void* y,
                                 int64_t M, int64_t N, int64_t K, float eps, void* stream);

/* ---- gated MLP input, decode form ---------------------------------------------------------------------------
 * replaces the l1 / l2 GEMV pair + gelu * mul of the single-token forward   [REF evo/configs/evo-1-8k-base_inference.yml:38;
 *                                                                          evo/generation.py:151-155]
 * a [M, I] bf16 = gelu_erf(x . W1^T) * (x . W2^T) with w12 [2I, K] = [W1; W2] bf16, x [M, K] bf16; 1 <= M <= 4,
 * I % 2 == 0, K % 8 == 0.  Both products are rounded to bf16 before the gate, as the unfused layers store them.
 * `grouped` != 0 (ABI 10): w12 is given in the row order of evo_mlp_gate_mfma_bf16's w12g -- blocks of 64 rows = 32 rows of W1 followed
 * by the same 32 rows of W2 (I % 32 == 0) -- so that ONE copy of l1 | l2 serves the prefill launch and the decode launches.
 * 5 <= M <= 64 (ABI 10; I % 32 == 0, K % 256 == 0, either layout): the MFMA weight-streaming form with the gate in its epilogue (csrc/gemv.hip
 * skinny_nw_kernel GATE) -- the pooled decode step of 5-64 live streams and the BOS sliver rows of a scoring batch; bit for bit
 * evo_linear_small_m_bf16 on w12 followed by evo_gelu_gate_bf16. */
int evo_mlp_gate_small_m_bf16(const void* x, const void* w12, void* a, int64_t M, int64_t I, int64_t K, int64_t grouped, void* stream);
/* same with the post-mixer RMSNorm folded in: x is the residual row, `scale` [K] the norm weight (bit-identical to
 * evo_rmsnorm_bf16 followed by evo_mlp_gate_small_m_bf16); this form also takes 5 <= M <= 8 at K = 4096. */
int evo_norm_mlp_gate_small_m_bf16(const void* x, const void* scale, const void* w12, void* a, int64_t M, int64_t I,
                                   int64_t K, float eps, int64_t grouped, void* stream);

/* ---- gated MLP activation ---------------------------------------------------------------------------
 * replaces ATen gelu + mul                                  [REF evo/configs/evo-1-8k-base_inference.yml:38]
 * g [M, 2*I] bf16 = [l1 x | l2 x]  ->  a [M, I] bf16 = gelu_erf(g[:, :I]) * g[:, I:]. */
int evo_gelu_gate_bf16(const void* g, void* a, int64_t M, int64_t I, void* stream);

/* ---- scoring tail ---------------------------------------------------------------------------------------
 * replaces torch.log_softmax + gather (+ entropy)            [REF evo/scoring.py:47-57,119-121]
 * logits [M, V] bf16 (logits_f32 = 0) or f32 (logits_f32 = 1; evo.generation keeps its score buffer in
 * f32 [REF evo/generation.py:97-103,287]), target [M] int64 (0 is written where target < 0 or >= V)
 * -> logprob [M] f32 (may be NULL), entropy [M] f32 (may be NULL).  fp32 log-softmax. */
int evo_logprob_entropy(const void* logits, int64_t logits_f32, const int64_t* target, float* logprob,
                        float* entropy, int64_t M, int64_t V, void* stream);

/* ---- fused scoring tail: unembed + log_softmax + gather (+ entropy) ---------------------------------
 * replaces  logits = x @ E^T ; log_softmax(logits) ; gather(next token) ; -sum p log p
 *                                                         [REF evo/scoring.py:47-57,81-84,119-121]
 * hidden [M, K] bf16 (final-norm output), emb [V = 512, K] bf16 (tied embedding), target [M] int64 or
 * NULL (negative = masked -> log-prob 0), logprob / entropy [M] f32 or NULL.  The [M, V] logits are
 * never written to HBM; every logit is rounded to bf16 once (the reference's logits tensor is bf16)
 * and the softmax statistics are taken in fp32 on the rounded values.  V must be 512, K % 32 == 0. */
int evo_unembed_logprob_bf16(const void* hidden, const void* emb, const int64_t* target,
                             float* logprob, float* entropy, int64_t M, int64_t V, int64_t K, void* stream);

/* ---- box-calibration probes (measurement infrastructure; no reference counterpart) -----------------------------------
 * Two FIXED kernels whose rates depend on the box (HBM, the clocks its power cap allows) and on nothing else in this library:
 * bench.py times them in the same process right before the headline so that a driver-timed number can be compared across
 * boxes and rounds.
 *   evo_probe_copy_f4:   dst[i] = src[i], 16 bytes per lane, grid-stride (nbytes % 16 == 0): the guide's "float4 copy".
 *   evo_probe_mfma_bf16: n_blocks workgroups of 4 waves (one per SIMD), each wave `iters` trips of 16 independent
 *                        v_mfma_f32_16x16x32_bf16 on register-resident pseudo-random bf16 operands (never zeros);
 *                        flop = n_blocks * 4 * iters * 16 * 16384.  out [n_blocks * 256] fp32 sink or NULL. */
int evo_probe_copy_f4(const void* src, void* dst, int64_t nbytes, void* stream);
int evo_probe_mfma_bf16(float* out, int64_t n_blocks, int64_t iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVO_MI355X_H */
