/*
 * mvit_hip.h -- C-ABI of the MI355X (gfx950) MViTv2 hot path.
 *
 * The reference (JunweiLiang/aicity_action) is pure Python: it has no FFI of its own; every
 * operator below replaces an implicit ATen dispatch on the MViT path, cited per entry point as
 * reference file:line (paths relative to the reference root).  The host side
 * (aicity_action_amd/, Python on PyTorch-ROCm) binds this library with ctypes -- see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions ("C-ABI layer"):
 *   - plain pointers + sizes; all pointers are DEVICE pointers owned by the caller;
 *   - the library never allocates DEVICE MEMORY, never synchronises, never throws.  The one piece of state it owns is a
 *     lazily created side stream + two events per device (mvit_side_stream / mvit_side_fork / mvit_side_join below, also used
 *     inside mvit_attention_bwd): created on the first call that needs it, from the calling thread, with no lock -- the
 *     threading contract of the reference applies (one process per GPU, the model touched by that process's main thread
 *     only, slowfast/utils/misc.py:307-320); set MVIT_NO_SIDE_STREAM to keep every launch on the caller's stream;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     0 (MVIT_OK) or a negative MVIT_E* code;
 *   - token-major activations: [batch][token][channel], channel contiguous;
 *   - `act_dtype`: MVIT_F32 or MVIT_BF16 = storage type of intermediate activations (and
 *     the MFMA operand type); the residual stream and all parameters are always fp32 unless a
 *     parameter is documented as "act-typed" (pre-converted by the host at load time).
 */
#ifndef MVIT_HIP_H
#define MVIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points declared between this push and the pop at the end of the file are
 * its ONLY dynamic exports (kernel handles, launch helpers and device stubs stay internal). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

enum { MVIT_F32 = 0, MVIT_BF16 = 1 };

enum {
    MVIT_OK = 0,
    MVIT_EINVAL = -1,      /* bad shape / null pointer */
    MVIT_EDTYPE = -2,      /* unsupported dtype combination */
    MVIT_ELAUNCH = -3,     /* hipGetLastError() != hipSuccess after a launch */
    MVIT_EUNSUPPORTED = -4 /* shape outside the compiled specialisations (e.g. head_dim != 96) */
};

/* GEMM epilogue flags */
enum {
    MVIT_EPI_BIAS = 1,     /* + bias[n] (fp32) */
    MVIT_EPI_GELU = 2,     /* exact erf GELU after bias (reference slowfast/models/common.py:28) */
    MVIT_EPI_RESIDUAL = 4  /* + residual[m][n] (fp32), applied last */
};

const char* mvit_version(void);
const char* mvit_strerror(int code);

/* LayerNorm over the last dim, biased variance, affine; x fp32 [rows][C] -> y act-typed [rows][C].
 * Replaces nn.LayerNorm at slowfast/models/attention.py:421,436 (eps 1e-6) and
 * slowfast/models/video_model_builder.py:1248-1249.  C in {96,192,384,768}. */
int mvit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                       int64_t rows, int C, float eps, int act_dtype, void* stream);

/* y[M][N] = epilogue( a[M][K] . w[N][K]^T ).  Replaces nn.Linear (addmm) at
 * slowfast/models/attention.py:231 (qkv), :281 (proj), :426 (proj_max_pool),
 * slowfast/models/common.py:27-31 (fc1/fc2).
 *   a:  a_dtype-typed (MVIT_F32 or MVIT_BF16), row stride lda elements
 *   w:  act-typed [N][K] (host pre-converts fp32 parameters when act_dtype == MVIT_BF16)
 *   bias fp32 [N] or NULL; residual fp32 [M][N] (row stride ldr) or NULL
 *   row_scale: NULL, or fp32 [M / rows_per_scale]: drop-path factor per sample applied to
 *              (acc + bias) before the residual add (slowfast/models/common.py:46-59)
 *   y:  out_dtype-typed, row stride ldy.
 * Constraints: K % 96 == 0 or K % 32 == 0; N % 32 == 0 for the bf16 MFMA path. */
int mvit_linear_fwd(const void* a, int a_dtype, int64_t lda, const void* w, const float* bias,
                    const float* residual, int64_t ldr, const float* row_scale, int64_t rows_per_scale,
                    void* y, int out_dtype, int64_t ldy, int64_t M, int N, int K, int epilogue,
                    int act_dtype, void* stream);

/* The block tail in one kernel (inference, 16-bit builds):  out = x + fc2(GELU_erf(fc1(LayerNorm(x))))
 * -- slowfast/models/attention.py:436-445 (x + drop_path(mlp(norm2(x))), drop-path = identity in eval mode) with Mlp.forward of
 * slowfast/models/common.py:26-34.  The [M][hidden] activation never reaches HBM (csrc/mlp_fused.hip).
 *   x, out: fp32 [M][C] (out may alias x: a workgroup reads and writes its own rows only); C in {96, 192, 384}, hidden = 4 C;
 *   packed: the weights re-laid once per weight version by mvit_mlp_fused_pack into mvit_mlp_fused_pack_bytes(C, hidden) bytes:
 *           per chunk of 32 hidden units the 16-bit fc1 rows (LayerNorm's gamma folded in) and fc2 columns in the kernel's LDS
 *           layout, then b1' = b1 + W1 beta (fp32 [hidden]);  w1 fp32 [hidden][C], w2 fp32 [C][hidden], b2 fp32 [C].
 * Returns MVIT_EUNSUPPORTED for other shapes / act_dtype == MVIT_F32: callers keep mvit_layernorm_fwd + two mvit_linear_fwd. */
int64_t mvit_mlp_fused_pack_bytes(int C, int hidden);
int mvit_mlp_fused_pack(const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2,
                        void* packed, int C, int hidden, void* stream);
int mvit_mlp_fused_fwd(const float* x, const void* packed, const float* b2, float* out, int64_t M, int C, int hidden,
                       float eps, int act_dtype, void* stream);
/* The same kernel with the attention output projection in front (slowfast/models/attention.py:281 proj, :434 x = x_res + x_block,
 * :436-445 the MLP branch; eval mode):   out = y + fc2(GELU_erf(fc1(LayerNorm(y)))),   y = resid + o . Wproj^T + bproj.
 *   o: attention output, act-typed [M][C]; resid: fp32 [M][C] (the pooled skip path); out fp32 [M][C] (may alias resid).
 * y never reaches HBM: the accumulators start at resid, the projection accumulates onto them, LayerNorm is taken from them.
 *   packed: mvit_block_tail_pack_bytes(C, hidden) bytes written by mvit_block_tail_pack: the proj weight as C/32 chunk images, the
 *           mvit_mlp_fused_pack image, the proj bias.  wproj fp32 [C][C].  Same shape limits as mvit_mlp_fused_fwd.
 */
int64_t mvit_block_tail_pack_bytes(int C, int hidden);
int mvit_block_tail_pack(const float* wproj, const float* bproj, const float* w1, const float* b1, const float* gamma,
                         const float* beta, const float* w2, void* packed, int C, int hidden, void* stream);
int mvit_block_tail_fwd(const void* o, const float* resid, const void* packed, const float* b2, float* out, int64_t M, int C,
                        int hidden, float eps, int act_dtype, void* stream);

/* Pooling conv + LayerNorm of one of q/k/v for all heads (attention_pool, conv variant:
 * slowfast/models/attention.py:12-83 with the Conv3d of :172-212 and LayerNorm(eps 1e-5) of
 * :185,199,213).  Input is the fused qkv activation [B][T*H*W][ld] (act-typed); channel of
 * (which, head g, d) = chan_off + g*96 + d.  Depthwise 3x3x3, zero pad 1, stride (1,s,s), the same
 * w[96][27] (fp32, reference layout [96,1,3,3,3]) for every head.  Output [B][heads][T*Ho*Wo][96]
 * act-typed.  head_dim is fixed at 96 (every block of every configs/Aicity model). */
int mvit_pool_conv_ln_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                          const float* beta, void* out, int B, int heads, int T, int H, int W,
                          int stride_hw, float eps, int act_dtype, void* stream);

/* out[b][lq][g*96+d] = softmax_k(q.k^T * scale) . v (+ q if add_q)  -- fused, scores never stored.
 * Replaces bmm/softmax/bmm + residual at slowfast/models/attention.py:267-279.
 * q [B][heads][Lq][96], k,v [B][heads][Lk][96], out [B][Lq][heads*96], all act-typed.
 * lse: NULL, or fp32 [B][heads][Lq] receiving log2(sum_k exp(score)) (saved for the backward pass). */
int mvit_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int heads,
                       int Lq, int Lk, float scale, int add_q, int act_dtype, void* stream);
/* Skip-path MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1) on the token grid (slowfast/models/attention.py:
 * 316-318,389-395,427-432); x fp32 [B][T*H*W][C] -> y fp32 [B][T*Ho*Wo][C]. */
int mvit_maxpool_skip_fwd(const float* x, float* y, int B, int T, int H, int W, int C, void* stream);

/* Cube embedding + separable position embedding (slowfast/models/stem_helper.py:335-338,
 * slowfast/models/video_model_builder.py:1196-1223): Conv3d(3->96, k(3,7,7), s(2,4,4), p(1,3,3)) +
 * bias, output token-major, + pos_spatial[h*W'+w] + pos_temporal[t].
 * clip fp32 [B][3][T][S][S]; w fp32 [96][3][3][7][7]; x fp32 [B][T/2*S/4*S/4][96]. */
int mvit_stem_fwd(const float* clip, const float* w, const float* bias, const float* pos_spatial,
                  const float* pos_temporal, float* x, int B, int T, int S, int act_dtype, void* stream);

/* Final LayerNorm(eps) + token mean + Linear(C,num_classes) + optional softmax
 * (slowfast/models/video_model_builder.py:1248-1249,1310; slowfast/models/head_helper.py:409-417).
 * x fp32 [B][N][C]; workspace fp32, at least mvit_head_workspace_bytes(B,N,C) bytes;
 * logits/probs fp32 [B][num_classes] (either may be NULL). */
int64_t mvit_head_workspace_bytes(int B, int N, int C);
int mvit_head_fwd(const float* x, const float* gamma, const float* beta, const float* w_head,
                  const float* b_head, float* workspace, float* logits, float* probs, int B, int N,
                  int C, int num_classes, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path (backward of every operator above + the optimizer step).  The reference gets these from
 * torch.autograd (2128 backward dispatches per step); here each is one entry point.
 * Gradients of act-typed activations are act-typed; gradients of the fp32 residual stream and of all
 * parameters are fp32.  "accumulate" parameters select (+=) vs (=).
 * ------------------------------------------------------------------------------------------------ */

/* LayerNorm backward (native_layer_norm_backward): statistics are recomputed from x.
 * dy: dy_dtype-typed [rows][C]; or, when rows_per_dy > 1, fp32 [rows/rows_per_dy][C] broadcast over
 * rows_per_dy consecutive rows and multiplied by dy_scale (backward of the token mean in the head,
 * slowfast/models/video_model_builder.py:1310).  workspace >= mvit_layernorm_bwd_workspace_bytes(C). */
int64_t mvit_layernorm_bwd_workspace_bytes(int C);
/* dx = (dx_base ? dx_base : 0) + LN-backward(dy); dx_base may alias dx (in-place accumulate).  When dx16 != NULL the resulting dx
 * also leaves as the 16-bit operand of the GEMMs that consume it next: dx16[r][:] = (16-bit)(dx[r][:] * (dx16_row_scale ?
 * dx16_row_scale[r / dx16_rows_per_scale] : 1)) -- bit for bit what mvit_cast_rows_f32_to_bf16 would make of dx (the
 * residual-stream gradient times the next branch's drop-path factor), without reading dx back.  dgamma / dbeta leave through the
 * library's ordered column reduction (no float atomics). */
int mvit_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_dtype, int64_t rows_per_dy, float dy_scale,
                       const float* dx_base, float* dx, float* dgamma, float* dbeta, int accumulate_param, float* workspace,
                       int64_t rows, int C, float eps, void* dx16, const float* dx16_row_scale, int64_t dx16_rows_per_scale,
                       void* stream);

/* erf-GELU as separate elementwise passes (training keeps the pre-activation; slowfast/models/common.py:28). */
int mvit_gelu_fwd(const void* x, void* y, int64_t n, int act_dtype, void* stream);
/* fc1 of the MLP in a training step (mlp.fc1 + GELU, reference slowfast/models/common.py:27-31 inside the autograd graph of
 * tools/train_net.py): pre = a . w^T + bias AND y = GELU(pre) from one pass over the accumulators; pre is what the backward
 * keeps.  16-bit operands/outputs only ([M][K] a with row stride lda, [N][K] w, [M][N] pre and y). */
int mvit_linear_gelu_fwd(const void* a, int64_t lda, const void* w, const float* bias, void* pre, void* y, int64_t M, int N,
                         int K, int act_dtype, void* stream);

/* Backward twin: y = GELU'(pre) * row_scale[m / rows_per_scale] * (a . w^T) -- the data gradient of mlp.fc2 fused with the GELU
 * backward (a = d_out rows, w = fc2.weight^T [N][K], pre = what mvit_linear_gelu_fwd kept); 16-bit operands / outputs. */
int mvit_linear_dgelu_fwd(const void* a, int64_t lda, const void* w, const float* row_scale, int64_t rows_per_scale,
                          const void* pre, void* y, int64_t M, int N, int K, int act_dtype, void* stream);

/* The same pair with the DERIVATIVE kept instead of the pre-activation: the forward writes y = GELU(pre) and dact = GELU'(pre)
 * from one erf / exp evaluation, the backward multiplies the fc2 data gradient by dact (no transcendental in its epilogue).
 * Shapes of the 128x192 kernels only (N % 192 == 0, K % 64 == 0), MVIT_EUNSUPPORTED otherwise. */
int mvit_linear_gelu_fwd_dsave(const void* a, int64_t lda, const void* w, const float* bias, void* dact, void* y, int64_t M, int N,
                               int K, int act_dtype, void* stream);
int mvit_linear_dact_fwd(const void* a, int64_t lda, const void* w, const float* row_scale, int64_t rows_per_scale, const void* dact,
                         void* y, int64_t M, int N, int K, int act_dtype, void* stream);
int mvit_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int act_dtype, void* stream);

/* dW[N][K] += sum_m dy[m][n] * a[m][k]  (mm backward wrt the weight) and, when db != NULL, db[n] += sum_m dy[m][n]
 * (bias gradient, fused: the dy tile is already on chip).  dy rows may carry the per-sample drop-path factor
 * row_scale[m / rows_per_scale].  dW / db are accumulated into (zero them once per step).  Every M chunk writes its partial dW / db
 * into its own slab of `workspace` (>= mvit_linear_wgrad_workspace_bytes(...) bytes) and a second kernel adds the slabs in chunk
 * order, so the result is bit-reproducible (the reference's fp32 step is).  N and K multiples of 4. */
int64_t mvit_linear_wgrad_workspace_bytes(int a_dtype, int64_t lda, int dy_dtype, int64_t ldd, int has_row_scale, int64_t M, int N,
                                          int K, int act_dtype);
int mvit_linear_wgrad(const void* a, int a_dtype, int64_t lda, const void* dy, int dy_dtype, int64_t ldd,
                      const float* row_scale, int64_t rows_per_scale, float* dW, float* db, int64_t M, int N,
                      int K, int act_dtype, float* workspace, int64_t workspace_bytes, void* stream);

/* out[n] (+)= sum_m row_scale[m/rps] * a[m][n]  (bias gradients).  workspace >= mvit_colsum_workspace_bytes(N). */
int64_t mvit_colsum_workspace_bytes(int N);
int mvit_colsum(const void* a, int a_dtype, int64_t M, int N, const float* row_scale, int64_t rows_per_scale,
                float* out, int accumulate, float* workspace, void* stream);

/* Backward of mvit_attention_fwd (recompute from LSE; deterministic, no atomics).
 * out / dout: [B][Lq][heads*96]; dq [B][heads][Lq][96]; dk, dv [B][heads][Lk][96]; all act-typed.
 * workspace >= mvit_attention_bwd_workspace_bytes(B, heads, Lq, Lk)  (delta + fp32 dK/dV partial sums for the
 * query-split variant used when B*heads*ceil(Lk/128) workgroups cannot fill the chip). */
int64_t mvit_attention_bwd_workspace_bytes(int B, int heads, int Lq, int Lk);
int mvit_attention_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                       const void* dout, void* dq, void* dk, void* dv, float* workspace, int B, int heads,
                       int Lq, int Lk, float scale, int add_q, int act_dtype, void* stream);
/* Backward of mvit_pool_conv_ln_fwd: dout [B][heads][T*Ho*Wo][96] -> the (which) slice of dqkv [B][T*H*W][ld]
 * (fully overwritten), dw [96][27] (accumulated), dgamma/dbeta.  dconv: scratch shaped like dout.
 * workspace >= mvit_pool_bwd_workspace_bytes(B, heads, T, H, W, stride_hw). */
int64_t mvit_pool_bwd_workspace_bytes(int B, int heads, int T, int H, int W, int stride_hw);
int mvit_pool_conv_ln_bwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                          const void* dout, void* dconv, void* dqkv, float* dw, float* dgamma, float* dbeta,
                          int accumulate_param, float* workspace, int B, int heads, int T, int H, int W,
                          int stride_hw, float eps, int act_dtype, void* stream);

/* Training pair that avoids the second convolution in the backward: the forward also writes xhat = (conv - mean) * rstd
 * ([B][heads][T*Ho*Wo][96], act-typed) and rstd (fp32 per output token) -- what torch's native_layer_norm keeps for
 * slowfast/models/attention.py:66-67 --; the backward takes them (or NULL, NULL = recompute as mvit_pool_conv_ln_bwd). */
int mvit_pool_conv_ln_fwd_train(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta,
                                void* out, void* xhat, float* rstd, int B, int heads, int T, int H, int W, int stride_hw, float eps,
                                int act_dtype, void* stream);
int mvit_pool_conv_ln_bwd_saved(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const void* xhat,
                                const float* rstd, const void* dout, void* dconv, void* dqkv, float* dw, float* dgamma,
                                float* dbeta, int accumulate_param, float* workspace, int B, int heads, int T, int H, int W,
                                int stride_hw, float eps, int act_dtype, void* stream);

/* The k and the v pooling conv (+ LayerNorm) of a block in ONE set of launches (attention.py:196-213: pool_k / norm_k and pool_v /
 * norm_v read adjacent head groups of the same fused qkv buffer -- channel offsets chan_off_k and chan_off_k + heads*96 -- with the
 * same geometry).  Spatial stride 2 only (the eleven blocks whose k / v grid is half the token grid); any other stride returns
 * MVIT_EUNSUPPORTED and the caller uses the single-tensor form twice.  Tensors of the pair are laid out back to back:
 * out_kv / xhat_kv / dout_kv / dconv_kv = [2][B][heads][T*Ho*Wo][96] (k then v), rstd_kv = [2][B*heads*T*Ho*Wo].  The backward
 * needs workspace >= 2 * mvit_pool_bwd_workspace_bytes(B, heads, T, H, W, 2).  Results are bit-identical to the single form
 * (each tensor's partial sums are reduced over its own rows, in the same order). */
int mvit_pool_conv_ln_fwd_train_kv(const void* qkv, int64_t ld, int chan_off_k, const float* w_k, const float* gamma_k,
                                   const float* beta_k, const float* w_v, const float* gamma_v, const float* beta_v, void* out_kv,
                                   void* xhat_kv, float* rstd_kv, int B, int heads, int T, int H, int W, int stride_hw, float eps,
                                   int act_dtype, void* stream);
int mvit_pool_conv_ln_bwd_saved_kv(const void* qkv, int64_t ld, int chan_off_k, const float* w_k, const float* gamma_k,
                                   const float* w_v, const float* gamma_v, const void* xhat_kv, const float* rstd_kv,
                                   const void* dout_kv, void* dconv_kv, void* dqkv, float* dw_k, float* dgamma_k, float* dbeta_k,
                                   float* dw_v, float* dgamma_v, float* dbeta_v, int accumulate_param, float* workspace, int B,
                                   int heads, int T, int H, int W, int stride_hw, int act_dtype, void* stream);

/* Backward of mvit_maxpool_skip_fwd: gradient goes to the first maximum of each window (ATen semantics). */
int mvit_maxpool_skip_bwd(const float* x, const float* dy, float* dx, int B, int T, int H, int W, int C, void* stream);

/* Training pair of the skip max-pool (attention.py:316-318,427-432): the forward also writes idx [B][T][Ho][Wo][C] bytes =
 * window position (ky*3+kx) of the first maximum; the backward routes dy by that index without re-reading x. */
int mvit_maxpool_skip_fwd_idx(const float* x, float* y, void* idx, int B, int T, int H, int W, int C, void* stream);
int mvit_maxpool_skip_bwd_idx(const void* idx, const float* dy, float* dx, int B, int T, int H, int W, int C, void* stream);

/* The widening skip path of a stage-transition block in one pass (slowfast/models/attention.py:424-432: x = proj(x) then
 * attention_pool(x, pool_skip), MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1)):
 *   fwd: y[B*T*Ho*Wo][Cout] = maxpool(x[B*T*H*W][Cin] . w[Cout][Cin]^T + bias); idx (may be NULL) = mvit_maxpool_skip_fwd_idx's bytes;
 *        x16 (may be NULL; written only together with idx) = x rounded to the act type [B*T*H*W][Cin], the operand of the weight gradient.
 *   bwd: dx[B*T*H*W][Cin] = g . w with g = mvit_maxpool_skip_bwd_idx(idx, dy) built on chip; wt = w^T [Cin][Cout] act-typed;
 *        d16 (may be NULL) receives g [B*T*H*W][Cout] in the act type, the dy operand of the weight-gradient GEMM.
 * The widened full-resolution tensor never reaches HBM; results are bit-identical to mvit_linear_fwd (fp32 in/out) followed by
 * mvit_maxpool_skip_fwd[_idx], resp. mvit_maxpool_skip_bwd_idx followed by mvit_linear_fwd.  act_dtype MVIT_BF16 only (the exact
 * path keeps the separate calls); Cin, Cout multiples of 96; MVIT_EUNSUPPORTED otherwise. */
int mvit_proj_maxpool_fwd(const float* x, const void* w, const float* bias, float* y, void* idx, void* x16, int B, int T, int H, int W,
                          int Cin, int Cout, int act_dtype, void* stream);
int mvit_proj_maxpool_bwd(const void* idx, const float* dy, const void* wt, float* dx, void* d16, int B, int T, int H, int W, int Cin,
                          int Cout, int act_dtype, void* stream);

/* Stem backward: dW [96][3][3][7][7], dpos_spatial, dpos_temporal accumulated from dx [B][N][96] (the input clip needs no
 * gradient; the bias gradient is mvit_colsum).  Kernel family by act_dtype (MVIT_F32: exact fp32 VALU; MVIT_BF16: matrix-core weight
 * gradient on 16-bit operands, fp32 accumulate).  Per-workgroup partial slabs of dW and of dpos_temporal in `workspace`
 * (>= mvit_stem_bwd_workspace_bytes) are added in a fixed order: the three gradients are bit-reproducible. */
int64_t mvit_stem_bwd_workspace_bytes(int B, int T, int S, int act_dtype);
int mvit_stem_bwd(const float* clip, const float* dx, float* dW, float* dpos_spatial, float* dpos_temporal, int B, int T, int S,
                  int act_dtype, float* workspace, int64_t workspace_bytes, void* stream);

/* Head, training variant.  mvit_head_ln_partial = stage 1 of mvit_head_fwd (workspace [B][ceil(N/32)][C]);
 * mvit_head_project_train: z = mean * mask (dropout mask holding 0 or 1/(1-p), or NULL), logits = z W^T + b;
 * mvit_head_bwd: dW, db, dz = mask * (dlogits W); the final-LN backward is mvit_layernorm_bwd(broadcast). */
int mvit_head_ln_partial(const float* x, const float* gamma, const float* beta, float* workspace, int B, int N, int C,
                         float eps, void* stream);
int mvit_head_project_train(const float* partials, const float* w_head, const float* b_head, const float* mask,
                            float* z_out, float* logits, int B, int N, int nchunks, int C, int num_classes,
                            void* stream);
int mvit_head_bwd(const float* dlogits, const float* z, const float* w_head, const float* mask, float* dW, float* db,
                  float* dz, int B, int C, int num_classes, int accumulate, void* stream);

/* SoftTargetCrossEntropy, reduction mean (slowfast/models/losses.py:133-142): loss scalar + dlogits*grad_scale. */
int mvit_soft_ce(const float* logits, const float* labels, float* loss, float* dlogits, int B, int num_classes,
                 float grad_scale, void* stream);

/* Multi-tensor optimizer step over a device table of chunk descriptors {float* p,g,m,v; int n; float wd} (size
 * mvit_mt_chunk_bytes()).  mvit_grad_norm: out2[0] = global L2 norm, out2[1] = min(1, max_norm/(norm+1e-6))
 * (torch clip_grad_norm_, tools/train_net.py:239-243); mvit_adamw_step: torch.optim.AdamW semantics with the
 * gradients scaled by norm_coef[1] (slowfast/models/optimizer.py:200-206). */
int mvit_mt_chunk_bytes(void);
int mvit_grad_norm(const void* chunk_table, int nchunks, float max_norm, float* partials, float* out2, void* stream);
int mvit_adamw_step(const void* chunk_table, int nchunks, const float* norm_coef, float lr, float beta1, float beta2,
                    float eps, int step, void* stream);
/* The same with {lr, 1 - beta1^step, sqrt(1 - beta2^step)} read from 3 floats of device memory: nothing in the launch changes
 * from iteration to iteration, so the whole train step can be captured in a hipGraph and replayed (the host refreshes the 3
 * floats before each replay; tools/train_net.py:113-115 sets the LR every iteration). */
int mvit_adamw_step_dev(const void* chunk_table, int nchunks, const float* norm_coef, const float* hyper, float beta1, float beta2,
                        float eps, void* stream);

/* Sliding-window front end (scripts/module_wrapper.py:304-370,384-397; scripts/utils.py:172-211): gather frame_length frames
 * per window by index from the decoded uint8 stream [N][H][W][3], resize to SxS with OpenCV's 8-bit INTER_LINEAR arithmetic
 * (aspect ignored), /255, (x-mean)/std, write fp32 [nclips][3][frame_length][S][S]. frame_idx: device int32. */
int mvit_window_preprocess(const void* frames, const int* frame_idx, float* out, int H, int W, int S, int nclips,
                           int frame_length, float mean, float std, void* stream);

/* fp32 -> bf16 (round to nearest even) conversion of parameters, n elements. */
int mvit_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* dst[r][c] = (16-bit) (row_scale ? row_scale[r / rows_per_scale] : 1) * src[r][c]; cols % 8 == 0.  Training only: the fp32
 * residual-stream gradient times its drop-path factor (reference: slowfast/models/common.py:46-59 backward) as the
 * 16-bit operand of the weight- and data-gradient GEMMs of proj / fc2 (attention.py:281,445 backward). */
int mvit_cast_rows_f32_to_bf16(const float* src, void* dst, int64_t rows, int cols, const float* row_scale,
                               int64_t rows_per_scale, void* stream);

/* Side stream of the library (one per device, created on first use) with a fork / join pair: independent operators may be issued
 * concurrently -- mvit_side_fork(stream); op(..., mvit_side_stream()); op(..., stream); mvit_side_join(stream) -- so that each
 * fills the other's partially occupied last wave of workgroups.  After the join everything is ordered on `stream` again.
 * mvit_side_fork / _join return MVIT_EUNSUPPORTED when the side stream is disabled (env MVIT_NO_SIDE_STREAM): issue on `stream`. */
void* mvit_side_stream(void);
int mvit_side_fork(void* stream);
int mvit_side_join(void* stream);

/* Deferred parameter-gradient reductions.  mvit_layernorm_bwd and mvit_pool_conv_ln_bwd_saved each end in small column-sum
 * launches over a partial table in their workspace (d_gamma / d_beta, conv d_w): eight per block backward, ~11 us apiece for a few
 * microseconds of work.  Between mvit_reduce_queue_begin() and mvit_reduce_queue_flush(stream) those launches are queued instead
 * (up to 16; a full queue flushes itself) and go out as ONE launch on `stream`, which must be ordered after every producer.
 * Contract while a queue is open: every workspace handed to those two entry points stays untouched (not freed, not passed to
 * another call) until the flush -- the partial tables live there.  The outputs (d_gamma, d_beta, d_w) are complete after the flush.
 * Same threading contract as the side stream (one queue per process, main thread).  The sums are bit-identical to the immediate
 * form: the same units of work in the same order, only launched together. */
int mvit_reduce_queue_begin(void);
int mvit_reduce_queue_flush(void* stream);

/* Query path of blocks WITHOUT a pooling conv (MVIT.Q_POOL_ALL off -> pool_q is None, attention.py:14-15,131-134,239-246):
 * out[b][g][n][:] = qkv[b][n][chan_off + g*96 : +96] (head split only, no LayerNorm); _bwd copies dout back into the slice of the
 * fused gradient buffer. */
int mvit_head_split_fwd(const void* qkv, int64_t ld, int chan_off, void* out, int B, int heads, int64_t N, int act_dtype, void* stream);
int mvit_head_split_bwd(const void* dout, void* dqkv, int64_t ld, int chan_off, int B, int heads, int64_t N, int act_dtype, void* stream);

/* dst[r][c] (optional, may be NULL) and dst_t[c][r] = (16-bit) src[r][c]: the forward and the data-gradient GEMM operands of one
 * nn.Linear weight (attention.py:231,281, common.py:27-31 and their backward), refreshed together after an optimizer step. */
int mvit_cast_transpose_f32_to_bf16(const float* src, void* dst, void* dst_t, int rows, int cols, void* stream);

/* The same for a whole set of weights in one launch.  desc_table: device array of ntensors records of mvit_cast_desc_bytes() bytes
 * {const float* src; void* dst; void* dst_t; int rows, cols; int first_tile, pad;} with first_tile = running sum of
 * ceil(rows/64)*ceil(cols/64); total_tiles = that sum over all tensors (one workgroup per 64x64 tile). */
int mvit_cast_desc_bytes(void);
int mvit_cast_transpose_multi(const void* desc_table, int ntensors, int total_tiles, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* MVIT_HIP_H */
