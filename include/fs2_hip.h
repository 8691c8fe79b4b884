/* libfs2_gfx950.so — C ABI of the MI355X (gfx950) FastSpeech2 forward-path kernels.
 *
 * The reference (francissfy/fastspeech2) is pure Python/PyTorch and has no FFI; each entry
 * point below replaces the stock-ATen op sequence of the reference lines cited, and is what a
 * ctypes binding in the reference's `network/` modules would call.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (FS2_ERR_*); `fs2_last_error()` gives the
 *    message of the calling thread's last failure.
 *  - all pointers are DEVICE pointers (hipMalloc'ed / torch CUDA tensors), 16-byte aligned;
 *    nothing is allocated, freed or synchronised inside; launches go to `stream`.
 *  - activations are PACKED ragged matrices [rows, channels] (channel-last, no padding between
 *    utterances); `cu` arrays hold B+1 int32 row offsets; `seg` tables hold, per row, the int32
 *    pair (start, end) of its utterance.
 *  - dtype: FS2_F32 (0) or FS2_BF16 (1) selects the operand type "T" of the MFMA kernels
 *    (accumulation, LayerNorm / softmax statistics and the residual stream are always fp32).
 */
#ifndef FS2_HIP_H
#define FS2_HIP_H

#include <hip/hip_runtime_api.h>
#include "fs2_pack_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FS2_F32 0
#define FS2_BF16 1
#define FS2_F16X3_QKV 3 /* fs2_attention only: q | k | v and ctx in split-fp16 storage */
#define FS2_F16X3 2   /* split-fp16 storage (see fs2_conv_gemm_f16x3): T copies of fs2_add_layernorm / fs2_cast, f16x3 GEMMs */

#define FS2_OK 0
#define FS2_ERR_ARG (-1)
#define FS2_ERR_UNSUPPORTED (-2)
#define FS2_ERR_HIP (-3)

const char* fs2_last_error(void);
int fs2_abi_version(void);
int fs2_device_arch(char* buf, int n);

/* seg[m] = (cu[b], cu[b+1]) and pos[m] = m - cu[b] for the utterance b owning row m.
 * Replaces make_non_pad_mask/_source_mask (network/utils.py:59-86, network/fastspeech2.py:174-176):
 * kernels consume row ranges instead of (B,T,T) boolean masks. pos may be NULL. */
int fs2_build_seg(const int* cu, int B, int* seg /*[M][2]*/, int* pos /*[M]*/, int M, hipStream_t stream);

/* out[m,:] = emb[ids[m],:] * xscale + alpha * pe[pos[m],:]
 * nn.Embedding(padding_idx=0) + PositionalEncoding (xscale=sqrt(d), alpha=1) or
 * ScaledPositionalEncoding (xscale=1): network/fastspeech2.py:116-120, network/encoder.py:125-128, :142-145.
 * out_f32 / out_t: either may be NULL. */
int fs2_embed_posenc(const long long* ids, const int* pos, const float* emb, const float* pe, float xscale,
                     float alpha, float* out_f32, void* out_t, int M, int d, int idim, int dtype,
                     hipStream_t stream);

/* fs2_embed_posenc with alpha read from device memory (training: the learnable nn.Parameter, no host read). */
int fs2_embed_posenc_dev(const long long* ids, const int* pos, const float* emb, const float* pe, float xscale,
                         const float* alpha_dev, float* out_f32, void* out_t, int M, int d, int idim, int dtype,
                         hipStream_t stream);

/* LayerNorm over channels, eps as given (reference: 1e-12), network/utils.py:39-47.
 * x fp32 [M,C]; writes fp32 and/or T copies. C % 4 == 0, C <= 1024. */
int fs2_layernorm(const float* x, const float* w, const float* b, float* out_f32, void* out_t, int M, int C,
                  float eps, int dtype, hipStream_t stream);

/* y = x + sum_p parts[p] (x fp32 [M,C] or NULL; parts fp32 [nparts][part_stride], the outputs of
 * fs2_conv_gemm incl. split-K partials), optionally stored to sum_out (may alias x); then LayerNorm(y)
 * when w != NULL.  The `residual + sublayer(x)` adds of EncoderLayer.forward fused with norm1 / norm2 /
 * after_norm (network/encoder.py:171-187, :252-253). */
int fs2_add_layernorm(const float* x, const float* parts, int nparts, long long part_stride, const float* w,
                      const float* b, float* sum_out, float* out_f32, void* out_t, int M, int C, float eps,
                      int dtype, hipStream_t stream);

/* Conv1d over time ("same" zero padding inside each utterance) or Linear (taps == 1) as an MFMA
 * implicit GEMM:  out = act(sum_tap X[m+tap-pad,:] . W[tap]^T + bias) [+ res],  act = `relu` code:
 * 0 none, 1 ReLU, 2 tanh (PostNet)
 * X [M,ldx] T; W packed [taps][N][Cin] T; bias fp32 [N] or NULL; res fp32 [M,ldr] or NULL;
 * out [M,ldo] fp32 (out_f32=1) or T.  seg is required when taps > 1.
 * ksplit > 1: the channel chunks are split into `ksplit` ranges whose fp32 partial results go to
 * out + p * split_stride (no activation); bias and res are added into partial 0, so the sum of the partials —
 * taken by fs2_add_layernorm's `parts` — is the full result.
 * Replaces nn.Conv1d in MultiLayeredConv1d (network/encoder.py:16-30) and ConvNorm
 * (network/utils.py:22-36), and nn.Linear in PositionwiseFeedForward / MultiHeadedAttention /
 * feat_out (network/encoder.py:39-44, :57-60, :89; network/fastspeech2.py:158-159, :219). */
int fs2_conv_gemm(const void* X, const void* W, const float* bias, const float* res, void* out, const int* seg,
                  int M, int N, int Cin, int taps, int ldx, int ldr, int ldo, int relu, int out_f32, int dtype,
                  int ksplit, long long split_stride, hipStream_t stream);

/* fs2_conv_gemm at (nearly) fp32 accuracy on the fp16 matrix pipe — the same nn.Conv1d / nn.Linear layers
 * (network/encoder.py:16-30, :39-44, :57-60, :89; network/utils.py:22-36) for the phone-level half, whose output
 * decides the INTEGER durations (network/variance_adaptor.py:128-138): a bf16 GEMM flips ~1 % of them, this mode none
 * on the fixtures (operand error < 2^-21, the level of fp32 summation-order noise).
 * Split-fp16 storage (dtype FS2_F16X3 of fs2_cast / fs2_add_layernorm, and this function's X / W / non-fp32 out):
 * 4 bytes per element; every aligned run of 32 consecutive elements holds 32 fp16 high halves hi = fp16(e), then 32
 * low halves lo = fp16(e - hi).  Buffers 128-byte aligned, row lengths multiples of 32.
 * X [M, ldx] split-fp16; W from fs2_pack_weight_f16x3 (w * w_scale); accumulates wl.xh + wh.xl + wh.xh in fp32
 * (3 MFMAs instead of 8 exact-fp32 ones of twice the cycles); out = acc / w_scale + bias, act, + res as
 * fs2_conv_gemm, stored as fp32 [M, ldo] / split-K partials (out_f32 != 0) or as split-fp16 (out_f32 == 0: no
 * residual, no split-K).  |x| and |w * w_scale| must stay inside the fp16 range (65504). */
int fs2_conv_gemm_f16x3(const void* X, const void* W, float w_scale, const float* bias, const float* res, void* out,
                        const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr, int ldo, int relu, int out_f32,
                        int ksplit, long long split_stride, hipStream_t stream);

/* Weight operand of fs2_conv_gemm_f16x3 from the parameter's own layout src (N, C, k) fp32:
 * dst[t][n][C / 32][2][32] fp16 (split-fp16 storage of w * scale along C).
 * scale: a power of two that brings max |w| near 2^13 (the fp16 low halves then stay normal numbers). */
int fs2_pack_weight_f16x3(const float* src, void* dst, int N, int C, int k, float scale, hipStream_t stream);

/* fs2_conv_gemm for the big frame-level FFN convs (nn.Conv1d x 2 of MultiLayeredConv1d, network/encoder.py:16-30) on
 * 256 x 256 tiles: one persistent 8-wave workgroup per CU, the W tile streamed through a ring of LDS half-stages by
 * LDS-DMA from PRE-TILED weights (fs2_pack_weight_tiled), the two wave groups of a workgroup alternating between an MFMA
 * segment and a load segment (csrc/conv_gemm256.hip).  bf16 only, taps 9 or 3, N % 256 == 0 (N / 256 in {1, 2, 4, 8}),
 * Cin % 64 == 0, activation 0 / 1; same arguments, semantics and split-K convention as fs2_conv_gemm otherwise
 * (N / 256 * ksplit must divide 8).  Same results as fs2_conv_gemm up to fp32 summation order. */
int fs2_conv_gemm256(const void* X, const void* Wt, const float* bias, const float* res, void* out, const int* seg, int M,
                     int N, int Cin, int taps, int ldx, int ldr, int ldo, int relu, int out_f32, int dtype, int ksplit,
                     long long split_stride, hipStream_t stream);

/* Weight operand of fs2_conv_gemm256 from fs2_conv_gemm's W [taps][N][Cin] (bf16): for every (256-channel tile, 64-channel
 * chunk, tap, k half) one contiguous 16-KB block in the order of the kernel's LDS image (64-byte rows, 16-byte slots
 * XOR-swizzled).  dst holds taps * N * Cin elements. */
int fs2_pack_weight_tiled(const void* W, void* dst, int N, int Cin, int taps, int dtype, hipStream_t stream);

/* 0 where fs2_conv_gemm is the kernel to use for this shape; else the `ksplit` to call fs2_conv_gemm256 with (1 unless
 * want_split != 0, i.e. an fs2_add_layernorm call sums the partials).  M = 1 << 20 asks whether the shape is served at
 * all (so that the caller packs tiled weights).  No launch; part of the kernel dispatch, no counterpart in the reference. */
int fs2_conv_gemm256_plan(int M, int N, int Cin, int taps, int dtype, int want_split);

/* out = X . W^T + bias for the short-K projections of the frame-level half - the fused q | k | v nn.Linear(d, d) x 3 of
 * MultiHeadedAttention (network/encoder.py:57-59, :73-76) - with the WEIGHTS IN REGISTERS (csrc/gemm_regw.hip): a wave
 * keeps its 64 output channels x all Cin = 256 input channels of W in VGPRs for the whole launch and the launch streams
 * the token rows through LDS in 16-row pieces (persistent workgroups, balanced row ranges).
 * X [M, ldx] bf16, W [N][256] bf16 (fs2_conv_gemm's layout for taps == 1), bias fp32 [N] or NULL, out [M, ldo] bf16
 * (out_f32 == 0) or fp32; N a multiple of 256.  Same results as fs2_conv_gemm with taps == 1 up to fp32 summation
 * order.  fs2_gemm_regw_plan: 1 where this kernel serves the shape and is the one to use, else 0 (no launch). */
int fs2_gemm_regw(const void* X, const void* W, const float* bias, void* out, int M, int N, int Cin, int ldx, int ldo,
                  int out_f32, int dtype, hipStream_t stream);
int fs2_gemm_regw_plan(int M, int N, int Cin, int dtype, int has_res, int relu, int out_f32);

/* out = LayerNorm(res + X . W^T + bias) (eps as given; gamma / beta [N]) for N = Cin = 256, bf16 operands: the attention
 * output projection, the residual add and norm1 of a post-LN EncoderLayer (network/encoder.py:89, :171-180 with
 * normalize_before = False; LayerNorm: network/utils.py:39-47) as ONE stream over the token rows with W in registers
 * (csrc/proj_ln.hip).  X [M, ldx] bf16, W [N][Cin] bf16, res [M, ldr] fp32, out_f32 [M, ldo] fp32 and out_t [M, ldo]
 * bf16 (both written).  fs2_proj_add_layernorm_plan: 1 where this kernel serves the shape and is the path to take
 * (else fs2_conv_gemm with `res` + fs2_add_layernorm); no launch. */
int fs2_proj_add_layernorm(const void* X, const void* W, const float* bias, const float* res, const float* gamma,
                           const float* beta, float* out_f32, void* out_t, int M, int N, int Cin, int ldx, int ldr, int ldo,
                           float eps, int dtype, hipStream_t stream);
int fs2_proj_add_layernorm_plan(int M, int N, int Cin, int dtype);

/* Split-K factor (>= 1) the launch schedules pass as `ksplit` for a GEMM of this shape whose partial outputs
 * an fs2_add_layernorm call sums: 1 unless the tiling would leave most CUs idle.  No launch; no counterpart in
 * the reference (it is part of the kernel dispatch). */
int fs2_conv_gemm_ksplit(int M, int N, int Cin, int taps, int dtype);

/* The same for a GEMM whose activation needs the full sum (ReLU of the first FFN conv, network/encoder.py:28-30, and of
 * the predictor convs, network/variance_adaptor.py:72-75): the launch then runs with `relu` = 0 and `ksplit` partials and
 * fs2_sum_act_layernorm applies the ReLU to their sum.  extra_us: cost of that call when it is an extra launch (0 where
 * a LayerNorm follows anyway).  Serves the latency regime only (one utterance, batch-1 decoding); 1 = do not split. */
int fs2_conv_gemm_ksplit_act(int M, int N, int Cin, int taps, int dtype, float extra_us);

/* LayerNorm(act(sum_p parts[p])) -> fp32 and/or T copies [M, C]; with w = b = NULL: act(sum_p parts[p]) alone.
 * parts fp32 [nparts][part_stride] (split-K partials of fs2_conv_gemm, bias in partial 0), relu: 0 none, 1 ReLU.
 * Conv1d -> ReLU -> LayerNorm of the predictors (network/variance_adaptor.py:72-75, :108-111) and the ReLU between the
 * FFN convs (network/encoder.py:28-30) when the conv ran split-K.  C % 4 == 0, C <= 1024, nparts <= 16. */
int fs2_sum_act_layernorm(const float* parts, int nparts, long long part_stride, int relu, const float* w, const float* b,
                          float* out_f32, void* out_t, int M, int C, float eps, int dtype, hipStream_t stream);

/* Fused varlen multi-head self-attention (scores, softmax, P.V) over packed qkv [M, ld] T laid out
 * q | k | v with each part [n_heads][dk]; ctx [M, ldo] T.  dk in {32, 64, 96, 128}.
 * dtype 2 (FS2_F16X3): fp32 q | k | v and fp32 arithmetic, ctx written in split-fp16 storage (the operand form of
 * fs2_conv_gemm_f16x3, as fs2_cast would convert it; ldo % 32 == 0, ctx 128-byte aligned).
 * dtype 3 (FS2_F16X3_QKV): q | k | v in split-fp16 storage as well (what fs2_conv_gemm_f16x3 writes with out_f32 = 0;
 * ld % 32 == 0, qkv 128-byte aligned, dk in {64, 128}); both products as three fp16 MFMAs (hi.hi + hi.lo + lo.hi, fp32
 * accumulation and softmax), ctx in split-fp16 storage: fp32-grade results at a fifth of the exact-fp32 MFMA time.
 * Replaces MultiHeadedAttention.forward between the projections (network/encoder.py:73-88). */
int fs2_attention(const void* qkv, void* ctx, const int* cu, int B, int max_len, int n_heads, int dk, int ld,
                  int ldo, int dtype, hipStream_t stream);

/* fs2_attention for launches that would leave the chip to a few long serial chains (one long utterance, a handful of
 * them: BASELINE config 4): the keys of every (utterance, head, query block) are split over `ksplit` workgroups, each
 * leaves its unnormalised context and (max, sum) in `workspace`, a second kernel merges them (same softmax,
 * network/encoder.py:78-88; fp32 merge).  bf16 (dtype 1) only.  fs2_attention_ksplit: the factor to use (1 = call
 * fs2_attention; no launch); fs2_attention_keysplit_workspace: bytes of `workspace` (16-byte aligned); M = cu[B]. */
int fs2_attention_ksplit(int B, int max_len, int n_heads, int dk, int dtype);
long long fs2_attention_keysplit_workspace(int M, int n_heads, int dk, int ksplit);
int fs2_attention_keysplit(const void* qkv, void* ctx, const int* cu, int B, int max_len, int M, int n_heads, int dk, int ld,
                           int ldo, int dtype, int ksplit, float* workspace, hipStream_t stream);

/* Debug: materialise softmax(QK^T/sqrt(dk)) of one utterance (rows row0..row0+len) as fp32
 * atten[n_heads][len][len] — what MultiHeadedAttention.atten holds (network/encoder.py:82-84),
 * consumed by calculate_all_attentions (network/fastspeech2.py:305-341). */
int fs2_attention_dump(const void* qkv, float* atten, int row0, int len, int n_heads, int dk, int ld, int dtype,
                       hipStream_t stream);

/* out[m,j] = x[m,:] . W[j,:] + b[j], nout <= 8, fp32 — the predictors' final nn.Linear
 * (network/variance_adaptor.py:78, :115 with repair D5). */
int fs2_rowdot(const float* x, const float* W, const float* b, float* out, int M, int C, int nout,
               hipStream_t stream);

/* out[m,j] = LayerNorm(x[m,:]) . W[j,:] + hb[j]: fs2_layernorm followed by fs2_rowdot in one pass, the LayerNorm output
 * is not stored — the last LayerNorm + `linear` of the pitch / energy predictors at inference
 * (network/variance_adaptor.py:107-115 with repair D5). */
int fs2_layernorm_rowdot(const float* x, const float* w, const float* b, const float* W, const float* hb, float* out,
                         int M, int C, int nout, float eps, hipStream_t stream);

/* dur[m] = (int) clamp(round_half_even(exp(logd[m]) - 1), 0) — DurationPredictor.inference,
 * network/variance_adaptor.py:135. */
int fs2_duration_round(const float* logd, int* dur, int M, hipStream_t stream);

/* LengthRegulator planning (network/variance_adaptor.py:13-31): per utterance, optional
 * round(d * control) (:18-19), all-zero row -> ones (:28-30), inclusive scan `cum`, frame counts
 * `olens` and frame offsets `cu_frames` [B+1].  dur_out receives the durations actually used. */
int fs2_lr_plan(const int* dur_in, const int* cu_phone, float control, int* dur_out, int* cum, int* olens,
                int* cu_frames, int B, hipStream_t stream);

/* cu[0] = 0, cu[b + 1] = lens[0] + .. + lens[b] (int32, n utterances): the frame offsets of a sub-batch rebased to its own
 * first frame - the `olens` -> padded-batch bookkeeping of pad_list / make_pad_mask (network/utils.py:50-86) for one
 * slice of a grouped phone phase. */
int fs2_cu_from_lens(const int* lens, int* cu, int n, hipStream_t stream);

/* LengthRegulator expansion + pad_list (network/variance_adaptor.py:31, network/utils.py:50-56) into
 * the packed frame matrix: out[f,:] = hs[cu_phone[b] + lr_index[f], :].  Also emits the frame-level
 * row tables (seg, pos) and the index map lr_index (phone index inside the utterance).  out_t: the T copy (dtype 1: bf16,
 * 2: split-fp16 storage, d % 32 == 0 and 128-byte aligned). */
int fs2_lr_gather(const float* hs, const int* cum, const int* cu_phone, const int* cu_frames, int B,
                  float* out_f32, void* out_t, int* frame_seg, int* frame_pos, int* lr_index, int Mf, int d,
                  int dtype, hipStream_t stream);

/* hs_va = hs + (energy_embed(e*escale) + pitch_embed(p*pscale)) for kernel-size-1 VarianceEmbedding
 * (network/variance_adaptor.py:35-51, :203-206), then the decoder positional encoding
 * out = hs_va * xscale + alpha * pe[pos] (network/encoder.py:216-219).  hs_va / out_f32 / out_t may be NULL
 * (at least one of out_f32, out_t is required; out_t in T: dtype 1 bf16, 2 split-fp16 storage with d % 32 == 0). */
int fs2_variance_add_posenc(const float* hs, const float* pitch, const float* energy, const float* Wp,
                            const float* bp, const float* We, const float* be, float pscale, float escale,
                            const float* pe, const int* pos, float xscale, float alpha, float* hs_va,
                            float* out_f32, void* out_t, int M, int d, int pdim, int edim, int dtype,
                            hipStream_t stream);

/* fs2_variance_add_posenc with the decoder's alpha read from device memory (training). */
int fs2_variance_add_posenc_dev(const float* hs, const float* pitch, const float* energy, const float* Wp,
                                const float* bp, const float* We, const float* be, float pscale, float escale,
                                const float* pe, const int* pos, float xscale, const float* alpha_dev, float* hs_va,
                                float* out_f32, void* out_t, int M, int d, int pdim, int edim, int dtype,
                                hipStream_t stream);

/* packed [M,C] <-> zero-padded [B,Tmax,C] (pad_list, network/utils.py:50-56). */
int fs2_unpack_padded(const float* src, const int* cu, float* dst, int B, int Tmax, int C, hipStream_t stream);
int fs2_pack_padded(const float* src, const int* cu, float* dst, int B, int Tmax, int C, int M,
                    hipStream_t stream);

/* fp32 -> T copy (weight packing; dtype FS2_F16X3: split-fp16 storage, n % 32 == 0). */
int fs2_cast(const float* src, void* dst, long long n, int dtype, hipStream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training path: what loss.backward() (run/train_decode.py:94-97) computes through the modules of
 * network/encoder.py, network/variance_adaptor.py and network/fastspeech2.py:232-295, restated
 * over the packed layout.  Gradients of GEMM operands travel as T matrices, the residual-stream
 * gradient and every parameter gradient as fp32.  `accumulate` != 0 adds into the output
 * (gradient accumulation, run/train_decode.py:94-102).
 * ------------------------------------------------------------------------------------------ */

/* Dropout (nn.Dropout in training mode; network/encoder.py:30,:44,:86,:128,:145,:178,:185, network/variance_adaptor.py:75,:111)
 * is counter based: element (row, col) of dropout site `stream_id` is kept iff
 * hash(seed, stream_id, row, col) >= rate * 2^32 and then scaled by 1 / (1 - rate).  Backward kernels evaluate the
 * same function again, so no mask is stored; a rate of 0 disables the site.  The random stream differs from
 * torch's, so dropout results are statistically, not bit-wise, comparable with the reference ("parity unpinned"). */

/* fs2_attention that also saves lse[h][m] (log2-domain log-sum-exp of the scaled scores,
 * [n_heads][M] fp32, M = total packed rows) for fs2_attention_bwd, with dropout of the probabilities
 * (site of head h = stream_id + h; row = packed query row, col = key index inside the utterance). */
int fs2_attention_fwd(const void* qkv, void* ctx, float* lse, const int* cu, int B, int max_len, int M, int n_heads,
                      int dk, int ld, int ldo, int dtype, unsigned seed, unsigned stream_id, float rate,
                      hipStream_t stream);

/* Backward of MultiHeadedAttention's core (network/encoder.py:73-88): from qkv, ctx (forward output),
 * dctx and lse computes dqkv [M, ld] T (dq | dk | dv).  delta: fp32 scratch [n_heads][M].
 * Probabilities are recomputed per tile; every output element has one writer (deterministic). */
int fs2_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv,
                      const int* cu, int B, int max_len, int M, int n_heads, int dk, int ld, int ldo, int dtype,
                      unsigned seed, unsigned stream_id, float rate, hipStream_t stream);

/* fs2_conv_gemm (no activation, no split-K) whose result is gate[m, n] > 0 ? out * gate_scale : 0: the input
 * gradient of a conv / linear fed by dropout(ReLU(.)) (`self.dropout(torch.relu(self.w1(x)))`,
 * network/encoder.py:29-30, :43-44), computed as the same implicit GEMM over the flipped-transposed weights
 * (fs2_pack_weight `bwd`); gate = the saved activation (dropped elements are 0), gate_scale = 1 / (1 - rate). */
int fs2_conv_gemm_gated(const void* X, const void* W, const float* bias, const float* res, const void* gate,
                        float gate_scale, void* out, const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr,
                        int ldg, int ldo, int out_f32, int dtype, hipStream_t stream);

/* out = dropout(act(conv(X) + bias)): training forward of the first FFN conv / linear (same references). */
int fs2_conv_gemm_dropout(const void* X, const void* W, const float* bias, void* out, const int* seg, int M, int N,
                          int Cin, int taps, int ldx, int ldo, int relu, int out_f32, int dtype, unsigned seed,
                          unsigned stream_id, float rate, hipStream_t stream);

/* fs2_add_layernorm with dropout of the summed sub-layer output before the residual add
 * (`residual + self.dropout(sublayer(x))`, network/encoder.py:178, :185) and / or of the LayerNorm output
 * (predictors: LayerNorm -> Dropout, network/variance_adaptor.py:74-75, :110-111). */
int fs2_add_layernorm_dropout(const float* x, const float* parts, int nparts, long long part_stride, const float* w,
                              const float* b, float* sum_out, float* out_f32, void* out_t, int M, int C, float eps,
                              int dtype, unsigned seed, unsigned parts_stream, float parts_rate, unsigned out_stream,
                              float out_rate, hipStream_t stream);

/* x = dropout(x) in place (fp32 [M, C]) with an optional T copy: positional-encoding dropout
 * (network/encoder.py:128, :145). */
int fs2_dropout(float* x, void* out_t, int M, int C, unsigned seed, unsigned stream_id, float rate, int dtype,
                hipStream_t stream);

/* Weight gradient of nn.Conv1d / nn.Linear (same layers as fs2_conv_gemm):
 *   grad[n][c][tap] (the parameter's own layout, fp32) = scale * sum_m dY[m, n] * X[m + tap - pad, c]
 * over token pairs inside one utterance, and (dbias != NULL) the bias gradient dbias[n] = scale * sum_m dY[m, n]
 * from the same pass.  MFMA GEMM over the token axis, `ksplit` token ranges summed in a fixed order from
 * `workspace` (fs2_conv_wgrad_workspace bytes). */
long long fs2_conv_wgrad_workspace(int N, int Cin, int taps, int ksplit);
int fs2_conv_wgrad(const void* dY, const void* X, float* grad, float* dbias, float* workspace, const int* seg, int M,
                   int N, int Cin, int taps, int ldy, int ldx, int dtype, int ksplit, float scale, int accumulate,
                   hipStream_t stream);

/* LayerNorm backward (network/utils.py:39-47) fused with the residual-gradient adds of
 * EncoderLayer.forward (network/encoder.py:171-187):  g = dy + sum_p parts[p];
 * dx = LayerNormBackward(g; x, w) (w == NULL: dx = g) + dres; relu_gate != 0 additionally zeroes dx
 * where x <= 0 (x is the ReLU output feeding the predictors' LayerNorm, network/variance_adaptor.py:66-75).
 * dw / db receive the affine gradients.  workspace: fs2_layernorm_bwd_workspace bytes.
 * Dropout: site `in_stream` masks the incoming gradient g (the forward dropped this LayerNorm's output, or —
 * w == NULL — the tensor whose gradient is summed here); site `t_stream` masks only the T copy dx_t, which then
 * is the gradient of a dropped-out sub-layer output while dx_f32 stays the residual-stream gradient. */
long long fs2_layernorm_bwd_workspace(int M, int C);
int fs2_layernorm_bwd(const float* x, const float* dy, const float* parts, int nparts, long long part_stride,
                      const float* w, const float* dres, int relu_gate, float* dx_f32, void* dx_t, float* dw, float* db,
                      float* workspace, int M, int C, float eps, int accumulate, int dtype, unsigned seed,
                      unsigned in_stream, float in_rate, unsigned t_stream, float t_rate, hipStream_t stream);

/* out[r][c] = scale * sum_m S[m][r] * G[m][c]  (S NULL: column sums, R = 1) — bias gradients, the
 * predictors' final-Linear and the k=1 VarianceEmbedding weight gradients.  pe != NULL:
 * out[c] = scale * sum_m G[m][c] * pe[pos[m]][c] (d alpha of ScaledPositionalEncoding summed over c
 * by the caller, network/encoder.py:142-145).  G fp32 (g_dtype 0) or bf16 (1).  R <= 8.
 * workspace: fs2_colreduce_workspace bytes. */
long long fs2_colreduce_workspace(int M, int C, int R);
int fs2_colreduce(const void* G, int g_dtype, int ldg, const float* S, int lds, int R, const float* pe, const int* pos,
                  float* out, float* workspace, int M, int C, float scale, int accumulate, hipStream_t stream);

/* fs2_colreduce without a caller workspace and without any allocation.  NO counterpart in the reference: it exists
 * for callers that cannot allocate device memory around a call (code under HIP graph capture, tight serving /
 * training loops, harnesses that forbid hipMalloc).  The stage-1 partials go to a static device array of
 * fs2_colreduce_static_max_bytes() = 8 MiB that every GPU the library runs on holds until the process ends; a
 * shape whose fs2_colreduce_workspace(M, C, R) exceeds that is rejected with FS2_ERR_ARG.  Same kernels, same
 * summation order, bit-identical results to fs2_colreduce.  All calls on one GPU share that array: calls that
 * overlap on one GPU (different streams or threads) race — serialise them. */
long long fs2_colreduce_static_max_bytes(void);
/* Resolves the static scratch array's address on the current device (hipGetSymbolAddress; once per device and process,
 * idempotent; returns 0 or FS2_ERR_HIP).  fs2_colreduce_static resolves it lazily on its first call per device, which
 * is a runtime query: callers that capture HIP graphs, or that must not touch the runtime inside their loop, call this
 * once per device beforehand - after it every fs2_colreduce_static call on that device is kernel launches only.
 * No counterpart in the reference (part of the no-allocation entry point above). */
int fs2_colreduce_static_init(void);
int fs2_colreduce_static(const void* G, int g_dtype, int ldg, const float* S, int lds, int R, const float* pe,
                         const int* pos, float* out, int M, int C, float scale, int accumulate, hipStream_t stream);

/* dx[m, :] = sum_j dy[m, j] * W[j, :] — input gradient of fs2_rowdot. */
int fs2_rowdot_bwd(const float* dy, const float* W, float* dx, int M, int C, int nout, hipStream_t stream);

/* LengthRegulator backward (network/variance_adaptor.py:13-31): out[p, :] = base[p, :] (or 0) +
 * scale * sum of dframes over the dur[p] frames phone p was repeated into. */
int fs2_lr_scatter(const float* dframes, const int* dur, const int* cum, const int* cu_phone, const int* cu_frames, int B,
                   const float* base, float scale, float* out, int Mp, int d, hipStream_t stream);

/* nn.Embedding(padding_idx=0) backward: demb[v, :] = xscale * sum_{m: ids[m] == v} dx[m, :], row 0 zero. */
int fs2_embed_bwd(const long long* ids, const float* dx, float xscale, float* demb, int M, int d, int idim,
                  int accumulate, hipStream_t stream);

/* Loss gradients (network/fastspeech2.py:259-279, network/loss.py:5-14): mode 0 L1, 1 MSE, 2 MSE against
 * log(dur + 1); scale = 1 / element count, times *scale_dev (the upstream gradient as a device
 * scalar, may be NULL) so that no host read is needed.  Writes fp32 [M, C] and/or T [M, ldt]
 * (pad columns zero-filled so the matrix can feed the MFMA kernels). */
int fs2_loss_grad(const float* pred, const float* tgt, const int* dur, int mode, float scale, const float* scale_dev,
                  float* g_f32, void* g_t, int M, int C, int ldt, int dtype, hipStream_t stream);

/* out = a * x (fp32). */
int fs2_scale(const float* x, float a, float* out, long long n, hipStream_t stream);

/* Packs a parameter in nn.Conv1d / nn.Linear layout src (N, C, k) fp32 into the GEMM operand layouts:
 * fwd[t][n][c] (fs2_conv_gemm) and bwd[k-1-t][c][n] (input gradient); either may be NULL. */
int fs2_pack_weight(const float* src, void* fwd, void* bwd, int N, int C, int k, int ld_fwd, int ld_bwd,
                    long long tap_stride_fwd, long long tap_stride_bwd, int dtype, hipStream_t stream);

/* fs2_pack_weight for n weights in ONE launch (a training step re-packs every GEMM operand after the optimiser
 * update: ~55 parameters, two launches each otherwise).  desc [n] and start [n + 1] live in DEVICE memory:
 * desc[j] = the arguments of fs2_pack_weight (fs2_pack_desc.h), start = running sum of
 * 2 * ceil(N_j * C_j / 256), total_blocks = start[n]. */
int fs2_pack_weights(const fs2_pack_desc* desc, const int* start, int n, int total_blocks, int dtype,
                     hipStream_t stream);

/* Training-mode nn.BatchNorm1d of the repaired PostNet (network/fastspeech2.py:13-54: Conv1d(bias=False) -> BatchNorm1d
 * -> Tanh (not on the last layer) -> Dropout; applied at :222-225) over packed frames: y [M, C] fp32 is a conv output
 * of the whole batch, statistics per channel over its M rows.
 * fs2_bn_stats: stats[3][C] = mean, 1 / sqrt(var + eps), biased variance (the caller folds them into running_mean /
 *   running_var).  workspace: fs2_bn_workspace(M, C) bytes.
 * fs2_bn_apply: out = dropout(act(gamma * (y - mean) * rstd + beta)) (+ res [M, ldr] fp32), act 0 none / 1 tanh, written
 *   as `dtype` [M, ldt] and / or fp32 [M, ldf]; dropout element (site, row, col) as in the other training kernels;
 *   stats NULL = no BatchNorm (use_batch_norm=False).
 * fs2_bn_bwd: backward of fs2_bn_apply from y and the statistics alone: dout [M, ldd] fp32 is the gradient of its output;
 *   dgamma = sum dz * xhat, dbeta = sum dz (dz = dout through dropout and tanh), dy_t [M, ldt] (`dtype`) =
 *   gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)): the operand of the conv's gradient GEMMs. */
long long fs2_bn_workspace(int M, int C);
int fs2_bn_stats(const float* y, float* stats, float* workspace, int M, int C, float eps, hipStream_t stream);
int fs2_bn_apply(const float* y, const float* stats, const float* gamma, const float* beta, int act, const float* res,
                 int ldr, float* out_f32, int ldf, void* out_t, int ldt, int M, int C, int dtype, unsigned seed,
                 unsigned site, float rate, hipStream_t stream);
int fs2_bn_bwd(const float* y, const float* stats, const float* gamma, const float* beta, int act, const float* dout,
               int ldd, float* dgamma, float* dbeta, void* dy_t, int ldt, float* workspace, int M, int C, int dtype,
               unsigned seed, unsigned site, float rate, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FS2_HIP_H */
