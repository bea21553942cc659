/*
 * devqa.h -- C ABI of libdevqa_hip.so: the MI355X (gfx950) kernels behind the
 * DE-VQA edit-then-evaluate hot path (BLIP-2 + FT_VL).
 *
 * The reference (sev777/DE-VQA) is pure Python with no FFI of its own; every
 * entry point below replaces a PyTorch op sequence that the reference reaches
 * through the call site cited on it ("R/" = /root/reference/DE-VQA/).  The host
 * side (de-vqa_amd/, Python, mirrors the reference's plugin API) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DEVQA_E_* code; it never
 *     throws and never synchronises the device; devqa_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - tensors are raw DEVICE pointers, row-major, caller-owned.  The op-level
 *     entry points allocate nothing, with ONE exception: the split-K GEMM rule for
 *     skinny problems keeps a fixed 36 MiB fp32 partial-sum buffer per (device,
 *     stream), allocated at first use and never freed or regrown (no hipFree /
 *     hipMalloc / synchronisation ever happens inside a launch path).
 *     bf16 is passed as uint16_t bit patterns.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - shapes are checked on the host before any launch (a kernel is never
 *     launched with operands its indexing does not cover).
 */
#ifndef DEVQA_H_
#define DEVQA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEVQA_OK 0
#define DEVQA_E_ARG (-1)   /* null pointer / bad flag */
#define DEVQA_E_SHAPE (-2) /* dimension outside what the kernel supports */
#define DEVQA_E_HIP (-3)   /* launch / runtime call failed */
#define DEVQA_E_OOM (-4)   /* an allocation inside devqa_ctx_create / devqa_ctx_bind_edit_target / devqa_comm_create failed */
#define DEVQA_E_STATE (-5) /* invalid handle, missing weight-table entry, call out of order */

typedef uint16_t devqa_bf16;

const char* devqa_last_error(void);
int devqa_abi_version(void);

/* ---- GEMM (K2, K3, K4, K5, K7, K8, K10: every nn.Linear on the path) --------------------
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  A/W bf16 K-contiguous (nn.Linear layout),
 * fp32 accumulate on MFMA.  Replaces F.linear at e.g. HF Blip2Attention.qkv/projection,
 * Blip2MLP, QFormer dense layers, OPT q/k/v/out_proj/fc1/fc2 and lm_head, reached from
 * R/editor/vllms_for_edit/blip2/blip2.py:25-31,35-45,69-74.
 *   bias      : fp32 [N] or NULL
 *   alpha     : result = (acc + bias) * alpha   (OPT q-scaling, modeling_opt q_proj*scaling)
 *   act       : DEVQA_ACT_*
 *   residual  : fp32 [M,ldc] or NULL, added after the activation (may alias out_f32)
 *   out_bf16 / out_f32 : at least one non-NULL; both are written when both are given
 * Requirements: K % 8 == 0, lda/ldw % 8 == 0, N % 4 == 0.
 */
#define DEVQA_ACT_NONE 0
#define DEVQA_ACT_RELU 1
#define DEVQA_ACT_GELU 2 /* exact erf GELU (HF "gelu") */
#define DEVQA_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x) (HF "quick_gelu", CLIP ViT in LLaVA) */
/* Fused SwiGLU (LLaMA FFN): W is the [gate | up] projection with its rows INTERLEAVED in blocks of 16 -- rows [32 b, 32 b + 16) = gate rows
 * [16 b, 16 b + 16), rows [32 b + 16, 32 b + 32) = the matching up rows -- and the output is out_bf16 [M, N / 2], out[m, 16 b + c] =
 * silu(acc[m, 32 b + c]) * acc[m, 32 b + 16 + c], from the fp32 accumulators (no [M, N] intermediate, no second pass).  bf16 output only,
 * no bias / residual / alpha; N % 256 == 0, ldc = row stride of the [M, N / 2] output (% 8 == 0); served by the 256 x 256 kernel only:
 * ask devqa_gemm_bf16_swiglu_supported(M, N, K) first (1 = this call shape takes that kernel under the current mode). */
#define DEVQA_ACT_SWIGLU_IL16 4
int devqa_gemm_bf16_swiglu_supported(int M, int N, int K);
int devqa_gemm_bf16(const devqa_bf16* A, int64_t lda, const devqa_bf16* W, int64_t ldw, const float* bias,
                    int M, int N, int K, float alpha, int act, const float* residual, devqa_bf16* out_bf16,
                    float* out_f32, int64_t ldc, void* stream);

/* Split-K form for skinny problems with a very long K (K10: dH = dlogits[rows,V] . E[V,d], rows <= 256; one row tile of 64 / 128 / 256 so that W streams once):
 * the K range is cut into `splits` slices (one workgroup column each), raw fp32 partial slabs go to
 * partial_ws [splits][M][N] and a second launch sums them in slice order (deterministic).  No epilogue. */
int devqa_gemm_bf16_splitk(const devqa_bf16* A, int64_t lda, const devqa_bf16* W, int64_t ldw, int M, int N, int K,
                           int splits, float* partial_ws, float* out_f32, void* stream);

/* Kernel selection for A/B measurements: 0 = default (LDS-DMA staged tiles when K % 64 == 0 -- 256x256 for
 * large problems, 128x128 / 64x128 below -- register-staged otherwise), 1 = always register-staged,
 * 2 = LDS-DMA staged without the 256x256 tile, 10..17 = experimental ring variants (csrc/gemm_bf16_pipe.hip), 20..28 = the 256x256
 * ping-pong kernel forced, with its tile orders / epilogue A/B switches (csrc/gemm_bf16_pp.hip: launch_gemm_pp).
 * Every variant issues the same MFMA sequence per output element, so results are bit-identical. */
int devqa_gemm_set_mode(int mode);

/* Measurement hook for bench.py's roofline lines (csrc/profile.hip): while enabled, every launch of an instrumented kernel is
 * bracketed by HIP events on its own stream (up to 98304 launches; thread-safe).  devqa_profile_read synchronises on the events
 * of one slot and returns its summed kernel time (ms), summed work and launch count.  Slots: 0..3 GEMM tile variants (32x128,
 * 64x128, 128x128, 256x256 ping-pong; work = 2 M N K FLOPs), 4 attention_mfma (FLOPs as launched), 5 ft_adamw_step (bytes if every
 * edit updates: 24 E Dout Din), 6 cosine top-k call (corpus bytes 4 N D), 7 layernorm (bytes).  devqa_profile_gemm / _gemm_read are
 * the round-1 names for slots 0..3 (arrays of 4).  Not part of the data path.  devqa_profile(1) starts a new recording, (0) pauses / ends it,
 * (2) resumes a paused one without dropping its records (bench.py brackets a SAMPLE of its timed steps: the event pairs cost ~2 % of throughput). */
#define DEVQA_PROF_SLOT_GEMM0 0
#define DEVQA_PROF_SLOT_ATTENTION 4
#define DEVQA_PROF_SLOT_FT_ADAMW 5
#define DEVQA_PROF_SLOT_COSINE 6
#define DEVQA_PROF_SLOT_LAYERNORM 7
int devqa_profile(int enable);
int devqa_profile_read(int slot, double* ms, double* work, int64_t* launches);
/* launches that were NOT recorded since devqa_profile(1) because the event pool was full: a reader reports it beside the figures */
int devqa_profile_dropped(int64_t* dropped);
int devqa_profile_gemm(int enable);
int devqa_profile_gemm_read(double* ms, double* flops, int64_t* launches);

/* fp32 ("faithful") compute mode: same contract with fp32 operands on the exact-fp32 MFMA
 * (v_mfma_f32_16x16x4_f32).  Used to pin the HIP path to the reference's fp32 results at 1e-3;
 * K/lda/ldw % 4 == 0. */
int devqa_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int M, int N, int K,
                   float alpha, int act, const float* residual, float* out_f32, int64_t ldc, void* stream);

/* ---- LayerNorm (every nn.LayerNorm on the path) --------------------------------------------
 * y = (x - mean) * rsqrt(var + eps) * gamma + beta over the last dim, fp32 statistics.
 * x fp32 [M,D]; optional add: x := x + add (fp32 [M,D], BERT-style post-LN residual,
 * HF Blip2QFormerSelfOutput/Output); outputs bf16 and/or fp32.  D % 4 == 0, D <= 16384.
 */
int devqa_layernorm(const float* x, const float* add, const float* gamma, const float* beta, int M, int D, float eps,
                    devqa_bf16* out_bf16, float* out_f32, void* stream);

/* ---- LLaMA-family row ops (LLaVA / Vicuna decoder; csrc/llama_ops.hip) -----------------------
 * devqa_rmsnorm        : y = (x [+ add]) * rsqrt(mean((x+add)^2) + eps) * w   (HF LlamaRMSNorm), bf16 and/or fp32 out
 * devqa_rmsnorm_bwd_dx : gradient of the above w.r.t. its input
 * devqa_rope_*         : rotary embedding (HF rotate_half convention, angle = pos * theta^(-2j/dh)), in place on the
 *                        first n_heads*dh columns of each row (q heads then k heads of a fused QKV buffer)
 * devqa_swiglu_*       : out[r,j] = silu(gu[r,j]) * gu[r,F+j] for the fused [gate|up] projection output [R,2F]
 */
int devqa_rmsnorm(const float* x, const float* add, const float* w, int M, int D, float eps, devqa_bf16* out_bf16,
                  float* out_f32, void* stream);
int devqa_rmsnorm_bwd_dx(const float* x, const float* add, const float* w, const float* dy, int M, int D, float eps, float* dx,
                         void* stream);
int devqa_rope_bf16(devqa_bf16* x, int64_t ld, int R, const int32_t* pos, int n_heads, int dh, float theta, void* stream);
int devqa_rope_f32(float* x, int64_t ld, int R, const int32_t* pos, int n_heads, int dh, float theta, void* stream);
int devqa_swiglu_bf16(const devqa_bf16* gu, int R, int F, devqa_bf16* out, void* stream);
int devqa_swiglu_f32(const float* gu, int R, int F, float* out, void* stream);

/* ---- attention (K3 ViT self-attn, K4 Q-Former self/cross-attn, K7 OPT causal attn) ----------
 * Packed varlen softmax attention: for sequence s (0..n_seq), query rows
 * [q_start[s], q_start[s]+q_len[s]) attend to
 *   - a fully visible key range  [kp_start[s], kp_start[s]+kp_len[s])   (shared image-token
 *     prefix for OPT probes; the whole key set for ViT / Q-Former), then
 *   - an own key range [ko_start[s], ko_start[s]+ko_len[s]) that is causal when `causal` != 0
 *     (query i sees own keys 0..i+ko_len-q_len; OPT) or fully visible otherwise.
 * q/k/v bf16 rows with row strides ldq/ldk/ldv (elements) and head h at column h*dh;
 * scores = (q.k) * scale; softmax in fp32; out bf16 [rows, H*dh] row stride ldo.
 * seq_desc: int32 [n_seq][6] = {q_start,q_len,kp_start,kp_len,ko_start,ko_len} (device).
 * max_q_len is the host-known max of q_len (grid sizing).  dh % 8 == 0, dh <= 128.
 * `causal`: bit 0 = the causal rule above; bit 2 (value 4) = a promise by the caller that EVERY sequence is plain non-causal
 * self-attention (kp_len == 0, ko_len == q_len, the ViT case), which lets short sequences take a kernel that keeps the whole
 * K / V of a sequence in LDS (descriptors live on the device, so the library cannot check this itself).
 * Replaces eager/sdpa attention in HF Blip2Attention, Blip2QFormerMultiHeadAttention and
 * OPTAttention (same call sites as above).
 */
int devqa_attention(const devqa_bf16* q, int64_t ldq, const devqa_bf16* k, int64_t ldk, const devqa_bf16* v,
                    int64_t ldv, devqa_bf16* out, int64_t ldo, const int32_t* seq_desc, int n_seq, int max_q_len,
                    int H, int dh, float scale, int causal, void* stream);

/* The kernel-variant switches of devqa_attention (DEVQA_ATTENTION_DMA / _NW / _QB / _DBUF / _SHORT / _RESIDENT: A/B measurements and
 * the variant tests) are read from the environment once; a host that changes them afterwards calls this to have them read again. */
int devqa_attention_reload_env(void);

int devqa_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                        int64_t ldo, const int32_t* seq_desc, int n_seq, int max_q_len, int H, int dh, float scale,
                        int causal, void* stream);

/* SwiGLU backward (MEND_VL edit path through LLaMA FFNs): gu [R,2F] (gate | up, bf16 or fp32), da fp32 [R,F] ->
 * dgu fp32 [R,2F] = (da u sig(g)(1 + g(1 - sig(g))) | da silu(g)).  Autograd of LlamaMLP's act_fn(gate) * up. */
int devqa_swiglu_bwd_bf16(const devqa_bf16* gu, const float* da, int R, int F, float* dgu, void* stream);
int devqa_swiglu_bwd_f32(const float* gu, const float* da, int R, int F, float* dgu, void* stream);

/* ---- attention backward (MEND_VL edit path) ---------------------------------------------------
 * Gradients of devqa_attention for descriptors WITHOUT a visible prefix (kp_len == 0; every sequence attends to its own
 * rows, causal or full): given q,k,v, the forward output o and d_out = dL/do, writes dq, dk, dv (same dtype/layout as
 * the inputs).  stats: fp32 workspace [rows * H * 2] (logsumexp and dO.O per (row, head)).  fp32 arithmetic.
 * Replaces the autograd backward of OPTAttention that R/editor/vllm_editors/mend_vl/mend_vl.py:177-186
 * (`torch.autograd.grad(edit_loss, self.autograd_params)`) runs through the edited layers.
 */
int devqa_attention_bwd(const devqa_bf16* q, int64_t ldq, const devqa_bf16* k, int64_t ldk, const devqa_bf16* v, int64_t ldv,
                        const devqa_bf16* o, int64_t ldo, const devqa_bf16* d_out, int64_t lddo, devqa_bf16* dq, int64_t lddq,
                        devqa_bf16* dk, int64_t lddk, devqa_bf16* dv, int64_t lddv, float* stats, const int32_t* seq_desc,
                        int n_seq, int max_len, int H, int dh, float scale, int causal, void* stream);
int devqa_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o,
                            int64_t ldo, const float* d_out, int64_t lddo, float* dq, int64_t lddq, float* dk, int64_t lddk,
                            float* dv, int64_t lddv, float* stats, const int32_t* seq_desc, int n_seq, int max_len, int H, int dh,
                            float scale, int causal, void* stream);

/* ---- MEND_VL row-wise pieces --------------------------------------------------------------------
 * relu_bwd: grad_in = grad_out where act_out > 0 else 0 -- the hooked fc1 output gradient (mend_vl.py:68-71).
 * gelu_f32 / gelu_bwd_f32: HF "gelu" (x * 0.5 * (1 + erf(x / sqrt 2))) of fp32 pre-activations that a backward pass keeps (out_bf16 and / or out_f32)
 *   and grad_in = grad_out * (Phi(x) + x phi(x)): the Q-Former FFN when FT_VL's substring rule selects Q-Former parameters
 *   (R/editor/vllm_editors/ft_vl/ft_vl.py:31-36 with the template "qformer", R/configs/ft_vl/blip2-opt-2.7b.yaml:9; autograd of
 *   transformers' Blip2QFormerIntermediate there).
 * mend_normalize_concat: out[r] = [(u[idx[r]] - u_mean)/(u_std + eps) | (v[idx[r]] - v_mean)/(v_std + eps)]  fp32
 *   [n_rows, du+dv]; idx NULL = identity, mean/std NULL = no normalisation (auxiliary_networks.py:118-148; the caller
 *   builds idx from the nz_mask rule of :118-120).
 * mend_lrlinear_epilogue: out = max((pre + bias) * scale + shift, 0) + x  (auxiliary_networks.py:70-83, init 'id').
 * logit_kl_rows: kl[r] = sum_v softmax(l1[r])_v (log_softmax(l1[r])_v - log_softmax(l2[r])_v)  (mend_vl.py:355-366, K18).
 */
int devqa_relu_bwd(const devqa_bf16* act_out, const devqa_bf16* grad_out, devqa_bf16* grad_in, int64_t n, void* stream);
int devqa_relu_bwd_f32(const float* act_out, const float* grad_out, float* grad_in, int64_t n, void* stream);
int devqa_gelu_f32(const float* x, devqa_bf16* out_bf16, float* out_f32, int64_t n, void* stream);
int devqa_gelu_bwd_f32(const float* x, const float* grad_out, float* grad_in, int64_t n, void* stream);
int devqa_mend_normalize_concat(const float* u, const float* v, const int32_t* idx, const float* u_mean, const float* u_std,
                                const float* v_mean, const float* v_std, float eps, int n_rows, int du, int dv, float* out,
                                void* stream);
int devqa_mend_lrlinear_epilogue(const float* pre, const float* bias, const float* scale, const float* shift, const float* x,
                                 float* out, int n_rows, int D, void* stream);
int devqa_logit_kl_rows(const float* logits1, int64_t ld1, const float* logits2, int64_t ld2, int R, int V, float* kl,
                        void* stream);

/* ---- MEND_VL training step (R/editor/vllm_editors/mend_vl/mend_vl.py:301-341) -----------------------------------
 * kl_dlogits: kl[r] as devqa_logit_kl_rows and dlogits2[r,:] = coef[r] * (softmax(l2[r]) - softmax(l1[r])), the gradient of
 *   coef-weighted KL(l1 || l2) w.r.t. l2 (autograd of logit_KL_loss, :355-366); dlogits2 bf16 or fp32, row stride ldd.
 * welford_rows: the running mean / variance update of GradientTransform in training mode, row by row in order
 *   (auxiliary_networks.py:88-91,122-136); reset != 0 = the `norm_init == False` branch (first row initialises).
 * mend_lrlinear_bwd: backward of devqa_mend_lrlinear_epilogue for one mode row: dpre [n,D] and += into the scale / shift /
 *   bias gradients.   sumsq_f32: out[0] += sum x^2 (gradient-norm clipping, :336-337).
 * adam_step: torch.optim.Adam defaults (no weight decay): grad scaled by grad_scale[0] (the clip coefficient), step >= 1.
 */
int devqa_kl_dlogits(const float* logits1, int64_t ld1, const float* logits2, int64_t ld2, int R, int V, const float* coef,
                     float* kl, void* dlogits2, int64_t ldd, int dlogits_bf16, void* stream);
int devqa_welford_rows(const float* x, const int32_t* idx, int n_rows, int D, int reset, float* mean, float* s, float* stdv,
                       const float* k_in, float* k_out, void* stream);   /* k_out != k_in: the sample counter before / after */
int devqa_mend_lrlinear_bwd(const float* pre, const float* bias, const float* scale, const float* shift, const float* dout,
                            int n_rows, int D, float* dpre, float* g_scale, float* g_shift, float* g_bias, void* stream);
int devqa_sumsq_f32(const float* x, int64_t n, float* out, void* stream);
int devqa_adam_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, const float* grad_scale, void* stream);

/* ---- TP_VL (T-Patcher) patch-neuron step (R/editor/vllm_editors/tp_vl/tp_vl.py:155-192) --------------------------------
 * One new FFN neuron (key k [d], bias b [1], value v [d_out]) trained on FROZEN activations of the edited layer.
 * tp_neuron_fwd: pre[t] = h[t].k + b for the T rows of the edit-role sequence; y[r] = ybase[r] + relu(pre[lab[r]]) * v for
 *   its L label rows (ybase = the layer's output rows without the new neuron).
 * tp_neuron_bwd: given dy = dLoss_e/dy [L,d_out] and the memory text's rows hm [Tm,d]: gradients gk, gb, gv of
 *   loss_e + lambda_a * mean_t exp(-pre[t]) + lambda_m * mean_t exp(pm[t] * (pm[t] > 0)) (+ weight_decay * p, Adam's L2),
 *   losses[0..1] = (loss_a, loss_m).  scratch: fp32 [T + Tm].  All fp32; one workgroup (T, Tm are tens of rows).
 */
int devqa_tp_neuron_fwd(const float* h, int T, int d, const float* k, const float* b, const int32_t* lab, int L, const float* v,
                        const float* ybase, int d_out, float* pre, float* y, void* stream);
int devqa_tp_neuron_bwd(const float* h, const float* pre, int T, int d, const int32_t* lab, int L, const float* dy, int d_out,
                        const float* hm, int Tm, const float* k, const float* b, const float* v, float lambda_a, float lambda_m,
                        float weight_decay, float* scratch, float* gk, float* gb, float* gv, float* losses, void* stream);

/* Gated variant for the LLaMA FFN, down(silu(gate(x)) * up(x)) -- R/configs/tp_vl/llava-v1.5-7b.yaml:8-12 and
 * minigpt-4-vicuna-7b.yaml patch gate_proj AND up_proj: the new neuron has keys K2 = [k_gate; k_up] [2,d], biases B2 [2], value
 * v [d_out]; act = silu(h.k_gate + b_gate) * (h.k_up + b_up); loss_a / loss_m are summed over the two in-layers
 * (tp_vl.py:164-177).  pre: fp32 [2,T] (gate row, up row); scratch: fp32 [2 * (T + Tm)]; GK2 [2,d], GB2 [2].
 */
int devqa_tp_gated_neuron_fwd(const float* h, int T, int d, const float* K2, const float* B2, const int32_t* lab, int L,
                              const float* v, const float* ybase, int d_out, float* pre, float* y, void* stream);
int devqa_tp_gated_neuron_bwd(const float* h, const float* pre, int T, int d, const int32_t* lab, int L, const float* dy, int d_out,
                              const float* hm, int Tm, const float* K2, const float* B2, const float* v, float lambda_a,
                              float lambda_m, float weight_decay, float* scratch, float* GK2, float* GB2, float* gv, float* losses,
                              void* stream);

/* ---- K2 patch-embed staging ---------------------------------------------------------------
 * im2col for Conv2d(3->D, k=P, s=P): pixels fp32 [B,3,S,S] -> bf16 [B*(S/P)^2, Kpad] with
 * column (c*P+py)*P+px, zero padded to Kpad (Kpad % 8 == 0).  HF Blip2VisionEmbeddings,
 * reached from blip2.py:25-31.
 */
int devqa_im2col_patches(const float* pixels, int B, int S, int P, int Kpad, devqa_bf16* out, void* stream);
int devqa_im2col_patches_f32(const float* pixels, int B, int S, int P, int Kpad, float* out, void* stream);
/* x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patches[b*np+p,:] + pos[1+p]  (fp32 [B,np+1,D]) */
int devqa_vit_assemble(const float* patches, const float* cls, const float* pos, int B, int np, int D, float* out,
                       void* stream);

/* ---- K6 token embedding + OPT learned positions ---------------------------------------------
 * out[r,:] = (src_row[r] >= 0 ? table_f32rows[src_row[r]] : embed[token[r]]) + pos_table[pos[r]+2]
 * i.e. rows are either gathered from a bf16 embedding table [V,D] by token id or copied from an
 * fp32 row buffer (the projected image tokens), then the OPT positional embedding (offset 2,
 * HF OPTLearnedPositionalEmbedding) is added.  out fp32 [R,D].  blip2.py:45-52,63; OPTDecoder.
 */
int devqa_embed_rows(const int32_t* token, const int32_t* src_row, const int32_t* pos, const devqa_bf16* embed,
                     const float* rows_f32, const devqa_bf16* pos_table, int R, int D, int V, int n_rows_f32,
                     int n_pos, float* out, void* stream);
int devqa_embed_rows_f32(const int32_t* token, const int32_t* src_row, const int32_t* pos, const float* embed,
                         const float* rows_f32, const float* pos_table, int R, int D, int V, int n_rows_f32, int n_pos,
                         float* out, void* stream);

/* gather rows: out[r,:] = in[idx[r],:] (fp32 or bf16 by elem_bytes 4/2) */
int devqa_gather_rows(const void* in, const int32_t* idx, int R, int D, int elem_bytes, void* out, void* stream);
/* fp32 -> bf16 (round to nearest even), n elements */
int devqa_cast_f32_bf16(const float* in, devqa_bf16* out, int64_t n, void* stream);
/* x = hi + lo, hi = bf16(x), lo = bf16(x - hi): the operands of the three-product bf16 form of an fp32 GEMM (devqa_mend_transform, DEVQA_MEND_SPLIT_BF16) */
int devqa_split_f32_bf16x2(const float* in, devqa_bf16* hi, devqa_bf16* lo, int64_t n, void* stream);
/* out = act(in) over n fp32 values (act: DEVQA_ACT_NONE or DEVQA_ACT_RELU), written as bf16 and / or fp32 (either may be NULL, out_f32
 * may alias in).  Finishes fp32 pre-activations that received a low-rank term before the activation (MEND_VL's edited fc1,
 * R/editor/vllm_editors/mend_vl/mend_vl.py:72-79). */
int devqa_act_cast(const float* in, int act, devqa_bf16* out_bf16, float* out_f32, int64_t n, void* stream);

/* ---- K9/K14 rows over the vocabulary ------------------------------------------------------
 * For each logits row r (fp32 [R,V], row stride ldl): argmax (first max wins, as torch.argmax),
 * and when labels != NULL: nll[r] = logsumexp(row) - row[label[r]];  when dlogits != NULL:
 * dlogits[r,:] = (softmax(row) - onehot(label[r])) * coef[r]  (bf16 [R,V]; coef = mask/mask.sum,
 * the gradient of R/editor/vllm_editors/ft_vl/ft_vl.py:191-199 w.r.t. the logits).
 * Replaces log_softmax+gather (ft_vl.py:193-195) and softmax+argmax
 * (R/evaluation/vllm_editor_eval.py:111,147).
 */
int devqa_vocab_rows(const float* logits, int64_t ldl, int R, int V, const int32_t* labels, const float* coef,
                     int32_t* argmax_out, float* nll_out, devqa_bf16* dlogits, int64_t ldd, void* stream);
int devqa_vocab_rows_f32(const float* logits, int64_t ldl, int R, int V, const int32_t* labels, const float* coef,
                         int32_t* argmax_out, float* nll_out, float* dlogits, int64_t ldd, void* stream);

/* LayerNorm backward w.r.t. the input only: dx from dy, for rows (x [+ add]) fp32 [M,D], gamma fp32 [D]. */
int devqa_layernorm_bwd_dx(const float* x, const float* add, const float* gamma, const float* dy, int M, int D, float eps,
                           float* dx, void* stream);

/* LayerNorm backward w.r.t. gamma / beta over M rows (autograd through nn.LayerNorm in full fine-tuning: LTE_VL training,
 * R/editor/vllm_editors/lte_vl/lte_vl.py:207-233): dgamma[c] (+)= sum_r dy[r,c] * xhat[r,c], dbeta[c] (+)= sum_r dy[r,c], xhat from
 * x (+ add).  rms != 0: LlamaRMSNorm (xhat = x * rsqrt(mean(x^2) + eps); dbeta may be nullptr).  stats_ws: 2 * M + D floats of scratch.
 * Deterministic.  devqa_colsum_f32: out[c] (+)= sum_r x[r,c] (bias gradients). */
int devqa_layernorm_bwd_params(const float* x, const float* add, const float* dy, int M, int D, float eps, int rms, int accumulate,
                               float* dgamma, float* dbeta, float* stats_ws, void* stream);
int devqa_colsum_f32(const float* x, int M, int D, int accumulate, float* out, void* stream);

/* ---- K10/K11/K12: fused FT_VL inner step on the edited matrix ---------------------------------
 * One launch per optimiser step, batched over E concurrent edits that each own a private
 * fp32 copy of the edited matrix W_e [Dout,Din] and AdamW moments m_e, v_e:
 *   g_e      = sum_{r<L} dy[e,r,:]^T (x) a[e,r,:]            (rank-L gradient, never materialised)
 *   m,v,w    = torch.optim.AdamW update (decoupled weight decay; bias correction with
 *              step t = adam_t[e]), optional L-inf clamp of w around w0 (norm_constraint)
 *   y[e,r,:] = W_e_new . a[e,r,:]                           (next step's fc2 output rows, no bias)
 * Edits with do_update[e] == 0 are skipped (their state and y are left unchanged).  On the first
 * update (adam_t[e]==1) w is read from w0 (shared, read-only) and m,v are taken as zero, so no
 * initialisation pass is needed.  Algorithmic HBM traffic per edit-step: read w,m,v + write
 * w,m,v = 6*4*Dout*Din bytes (the reference additionally round-trips g: 7 tensors, 734 MB at
 * 2560x10240).
 * Replaces loss.backward() onto the weight + opt.step() + clamp (ft_vl.py:131-141).
 *   w,m,v : fp32 [E][Dout][Din]      w0 : fp32 [Dout][Din] shared (w0_stride_e = 0) or per edit (= Dout*Din)
 *   a     : fp32 [E][Lmax][Din]      dy : fp32 [E][Lmax][Dout]     y : fp32 [E][Lmax][Dout]
 *   do_update, adam_t : int32 [E] (device, from devqa_ft_step_control)
 *   clamp_eps < 0 disables the clamp.  1 <= Lmax <= DEVQA_FT_MAX_ROWS, Din % 4 == 0 (Lmax <= 16: the a-rows stay in registers;
 *   17..64: re-read from cache per column step -- long targets such as captions).
 */
#define DEVQA_FT_MAX_ROWS 64
int devqa_ft_adamw_step(float* w, float* m, float* v, const float* w0, const float* a, const float* dy, float* y,
                        const int32_t* do_update, const int32_t* adam_t, int E, int Lmax, int Dout, int Din,
                        float lr, float beta1, float beta2, float eps, float weight_decay, float clamp_eps,
                        int64_t w0_stride_e, void* stream);

/* The same step without a first-moment MATRIX ("factored momentum"; what the path's FT_VL loops call).  Inside one edit's loop the rows a[e]
 * are constant -- every layer below the edited matrix is frozen (ft_vl.py:111-146 re-runs the same forward) -- so every gradient is
 * dy_t^T (x) a and the first moment is m_t = D_t^T (x) a with D_t = lerp(D_{t-1}, dy_t, 1 - beta1): a state of [Lmax][Dout] floats per edit
 * instead of [Dout][Din].  The kernel keeps D in `dstate`, rebuilds m_t per element from D_t and the a-values it holds anyway (L fmas) and
 * streams only w and v: 4*4*Dout*Din bytes per edit-step (first update: read w0, write w, v = 3*4) instead of 6*4.  Exact in exact
 * arithmetic; in fp32 m_t differs from the recurrence by rounding only.  The second moment is a sum of squares of sums and does NOT factor
 * safely in fp32 -- except for an edit with ONE loss row (single[e] != 0: dy[e][r] == 0 and a[e][r] == 0 for every r >= 1, at every step):
 * then v_t[i][j] = e_t[i] * a[e][0][j]^2 with e_t = beta2 e_{t-1} + (1 - beta2) dy_t[0][i]^2, one term, nothing cancels; e is kept in row
 * Lmax of `dstate`, v[e] is neither read nor written and only w crosses HBM (2*4*Dout*Din bytes per step).
 * CONTRACT: a[e] must hold the same values at every step between two first updates (adam_t[e] == 1) of edit e.
 *   dstate : fp32 [E][Lmax + 1][Dout] (rows 0..Lmax-1 laid out like dy, row Lmax = e; contents ignored at adam_t[e] == 1)
 *   single : int32 [E] or NULL (no edit takes the one-row form); everything else as devqa_ft_adamw_step.
 */
int devqa_ft_adamw_step_fm(float* w, float* dstate, float* v, const float* w0, const float* a, const float* dy, float* y,
                           const int32_t* do_update, const int32_t* adam_t, const int32_t* single, int E, int Lmax, int Dout, int Din,
                           float lr, float beta1, float beta2, float eps, float weight_decay, float clamp_eps,
                           int64_t w0_stride_e, void* stream);

/* ---- column compaction of the FT loop (csrc/ft_compact.hip) ------------------------------------------
 * With a = relu(.) constant over the loop, a column j of the edited matrix whose a[e,r,j] == 0 for every
 * loss row r has zero gradient, zero moments and zero update at every step (weight_decay == 0), so the loop
 * only carries the active columns of each edit.  devqa_active_columns lists them (ascending) per edit;
 * devqa_gather_cols_* builds out[e,row,c] = c < count[e] ? src[e*src_stride_e + row*ld_src + idx[e*idx_stride_e + c]] : 0;
 * devqa_scatter_cols_add_f32 adds a compacted [rows,npad] block back into a dense matrix (one edit).
 * devqa_ft_adamw_step then runs on [E,Dout,npad] with w0_stride_e = Dout*npad.
 */
int devqa_active_columns(const float* a, int E, int L, int Din, int32_t* idx, int32_t* count, void* stream);
int devqa_gather_cols_f32(const float* src, int64_t src_stride_e, int64_t ld_src, int rows, const int32_t* idx,
                          int64_t idx_stride_e, const int32_t* count, int E, int npad, float* out, void* stream);
int devqa_gather_cols_bf16(const devqa_bf16* src, int64_t src_stride_e, int64_t ld_src, int rows, const int32_t* idx,
                           int64_t idx_stride_e, const int32_t* count, int E, int npad, devqa_bf16* out, void* stream);
int devqa_scatter_cols_add_f32(const float* comp, int rows, const int32_t* idx, const int32_t* count, int npad, float* dense,
                               int64_t ld_dense, void* stream);

/* y[e,r,:] = W[e or shared] . a[e,r,:] (+ bias) (+ resid[e,r,:]) : fc2 on a few cached rows with an
 * fp32 matrix (pre-/post-edit probe tails, step-0 forward).  w_stride_e = 0 shares one matrix.  1 <= L <= DEVQA_FT_MAX_ROWS. */
int devqa_rows_matvec_f32(const float* w, int64_t w_stride_e, const float* a, const float* bias, const float* resid,
                          float* y, int E, int L, int Dout, int Din, void* stream);

/* ---- K13 delta / restore / apply ---------------------------------------------------------
 * mode 0: delta = w - w0            (ft_vl.py:148)
 * mode 1: w += delta                (ft_vl.py:60-61)
 * mode 2: w = w0                    (ft_vl.py:151-153, :44-45)
 */
int devqa_delta_op(int mode, float* w, const float* w0, float* delta, int64_t n, void* stream);

/* ---- FT loop control on device (ft_vl.py:125-133,145-146) ------------------------------------
 * For every edit e with active[e] != 0:
 *   loss = sum_r nll[e,r]*mask[e,r] / sum_r mask[e,r];  losses[e][step] = loss;  n_steps[e] = step+1;
 *   do_update[e] = (loss >= floor); if so adam_t[e] += 1 (the 1-based AdamW step index);
 *   if (loss < floor) active[e] = 0   (the reference skips backward/step below the floor, then breaks).
 * Inactive edits get do_update[e] = 0.  int32/fp32 device arrays of length E (losses: [E][max_steps]).
 */
int devqa_ft_step_control(const float* nll, const float* mask, int E, int Lmax, int step, int max_steps, float floor,
                          int32_t* active, int32_t* do_update, int32_t* n_steps, int32_t* adam_t, float* losses,
                          void* stream);

/* ---- K19 cosine top-k (dynamic-eval retrieval / IKE) ----------------------------------------
 * scores = normalise?(Q) . normalise?(C)^T ; per query the k best corpus ids sorted by
 * descending score, ties -> lowest id.  fp32 scan of the corpus (HBM-bound: N*D*4 bytes per
 * query batch) keeps k+8 candidates per query which are re-scored in fp64 for exact ordering.
 * corpus fp32 [N,D], queries fp32 [Q,D]; workspace bytes from devqa_cosine_topk_workspace.
 * Replaces util.normalize_embeddings + util.semantic_search(dot_score)
 * (R/dataset/vllm.py:65-70,104,117; R/easyeditor/models/ike/ike_main.py:193-202).
 * k <= 32, D % 4 == 0, D <= 1024.
 */
int64_t devqa_cosine_topk_workspace(int N, int Q, int k);
int devqa_cosine_topk(const float* corpus, const float* queries, int N, int Q, int D, int k, int normalize_corpus,
                      int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream);
/* The reference normalises its stored embeddings ONCE, when the corpus is loaded (R/dataset/vllm.py:104,117:
 * util.normalize_embeddings), and every search is a dot product against them.  Same division of work here: devqa_row_inv_norm fills
 * out[r] = 1 / ||rows[r]|| (0 for a zero row) once per corpus, devqa_cosine_topk_cached is devqa_cosine_topk(normalize_corpus = 1)
 * reading those inverse norms instead of accumulating them again (corpus_inv_norm == NULL: accumulate, as devqa_cosine_topk).  The
 * fp64 re-score of the k + 8 candidates recomputes their norms in fp64 either way, so the reported order and scores do not change.
 * Q <= 4 and N <= 20480 run as ONE launch (scores, then selection by the last-arriving workgroup). */
int devqa_row_inv_norm(const float* rows, int R, int D, float* out, void* stream);
int devqa_cosine_topk_cached(const float* corpus, const float* corpus_inv_norm, const float* queries, int N, int Q, int D, int k,
                             int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream);

/* =====================================================================================================================
 * PATH LEVEL (csrc/path_ctx.hip): a model context and the launch schedules of the hot path over the kernels above -- the
 * surface SURVEY.md 8(b) lists, so that a host in any language drives the edit-then-evaluate path without re-implementing
 * the schedules.  The Python host (de-vqa_amd/engine.py, batched.py) calls exactly these for BLIP-2.
 *
 *   - a context is created once per (process, GPU) from the model dims and a TABLE of named device weights.  The table
 *     stores POINTERS: the caller keeps ownership and may update weights in place (editors do).  Names are the HF parameter
 *     names of SURVEY.md Appendix D (R/configs/ft_vl/blip2-opt-2.7b.yaml:8 addresses them), 2-D weights in the compute dtype
 *     ([out, in], K-contiguous), 1-D parameters / ViT class + position embeddings / query_tokens fp32, plus the derived
 *     GEMM operands:  "derived.patch_w_gemm" [v_hidden, Kpad] (conv weight flattened, K = 3 P^2 zero-padded to a multiple of
 *     64), "derived.dec_qkv.<layer>.weight" [3d, d] + ".bias" fp32 [3d] (q, k, v projections of an OPT layer fused),
 *     "derived.embed_T" [d, V] (transposed tied embedding, dH = dlogits . E).  An fp32 edit target "<name>" may come with its
 *     compute-dtype shadow "<name>#shadow", which devqa_apply_delta / devqa_restore keep in step.
 *   - calls on one context are serialised by the caller; every call takes the stream; scratch memory is a caller-provided,
 *     256-byte aligned `workspace` of at least the bytes the matching *_workspace query returns (-1: bad handle / dims).
 *     The context itself owns only the pristine copy of the bound edit target.
 *   - status codes and devqa_last_error() as above; no call synchronises the device.
 * ===================================================================================================================== */
typedef uint64_t devqa_ctx_t;
typedef uint64_t devqa_comm_t;
#define DEVQA_FAMILY_BLIP2_OPT 1 /* ViT-g + Q-Former + projection, OPT decoder (R/editor/vllms_for_edit/blip2/blip2.py) */
#define DEVQA_FAMILY_LLAVA 2     /* CLIP ViT (hidden state -2, CLS dropped) + 2-layer GELU projector, LLaMA decoder (.../llava/llava.py:25-68) */
#define DEVQA_FAMILY_MINIGPT4 3  /* EVA ViT-g + Q-Former + llama_proj, LLaMA decoder (.../minigpt4/minigpt4.py:33-69) */
#define DEVQA_DTYPE_BF16 1
#define DEVQA_DTYPE_F32 2
#define DEVQA_SCORE_COLS 16   /* [sample_id, reliability, text_rephrase, image_rephrase, 9 locality accs, edit_time, steps, final_loss] */

#define DEVQA_DESC_FUSE_SWIGLU 1
typedef struct devqa_model_desc {
    int32_t family;        /* DEVQA_FAMILY_* */
    int32_t compute_dtype; /* DEVQA_DTYPE_BF16 (bf16 operands, fp32 accumulate / residual stream) or DEVQA_DTYPE_F32 ("faithful") */
    int32_t image_size, patch_size, v_hidden, v_layers, v_heads, v_ffn;             /* ViT (HF Blip2VisionConfig) */
    int32_t q_hidden, q_layers, q_heads, q_ffn, q_cross_freq, num_query_tokens;     /* Q-Former */
    int32_t t_hidden, t_layers, t_heads, t_ffn, t_vocab, t_max_pos;                 /* decoder (t_max_pos: OPT's learned positions) */
    float v_ln_eps, q_ln_eps, t_ln_eps;
    /* LLaMA-family decoders (DEVQA_FAMILY_LLAVA / _MINIGPT4): RMSNorm epsilon, rotary base; LLAVA: number of CLIP encoder layers that
     * run (HF vision_feature_layer = -2 -> v_layers - 1).  Ignored by DEVQA_FAMILY_BLIP2_OPT. */
    float t_rms_eps, t_rope_theta;
    int32_t v_run_layers, t_flags;        /* t_flags bit 0 (DEVQA_DESC_FUSE_SWIGLU; LLaMA decoders, bf16): when the weight table holds
                                             derived.llama_gu_il.<i>.weight (the [gate | up] operand with rows interleaved in blocks of 16) and the call
                                             shape takes the 256 x 256 kernel, SwiGLU runs inside that GEMM's epilogue (DEVQA_ACT_SWIGLU_IL16) */
} devqa_model_desc;

/* Weight-table names per family.  The table carries ONE canonical naming; a host whose checkpoint names differ registers its tensors
 * under these (the Python host does: MiniGPT4Native.weight_table).
 *   BLIP2_OPT : the HF Blip2ForConditionalGeneration parameter names + derived.dec_qkv.<i>.{weight,bias} (fused
 *               q|k|v rows), derived.patch_w_gemm [v_hidden, Kpad], derived.embed_T [t_hidden, vocab]; an fp32 edit target additionally
 *               as "<name>#shadow" (its compute-dtype copy).
 *   LLAVA     : vision_tower.vision_model.{embeddings.{class_embedding, position_embedding.weight}, pre_layrnorm.*, encoder.layers.<i>.
 *               {layer_norm1, layer_norm2, self_attn.out_proj, mlp.fc1, mlp.fc2}.*}, derived.clip_qkv.<i>.{weight,bias},
 *               multi_modal_projector.linear_{1,2}.*, language_model.model.{embed_tokens.weight, norm.weight, layers.<i>.{input_layernorm,
 *               post_attention_layernorm}.weight, layers.<i>.self_attn.o_proj.weight, layers.<i>.mlp.down_proj.weight},
 *               language_model.lm_head.weight, derived.llama_qkv.<i>.weight [3d, d], derived.llama_gu.<i>.weight [2F, d] (gate rows, then
 *               up rows), derived.patch_w_gemm, derived.embed_T (= lm_head^T).
 *   MINIGPT4  : the vision side under the BLIP2_OPT names (visual_encoder.* -> vision_model.*, ln_vision -> vision_model.post_layernorm,
 *               Qformer.bert.* -> qformer.*, llama_proj -> language_projection), the decoder under the LLAVA names (llama_model.* ->
 *               language_model.*). */

typedef struct devqa_weight {
    const char* name;   /* HF parameter name or "derived.*" */
    const void* ptr;    /* device pointer, caller-owned */
    int32_t dtype;      /* DEVQA_DTYPE_* */
    int32_t ndim;
    int64_t shape[4];
} devqa_weight;

int devqa_ctx_create(int device, const devqa_model_desc* desc, const devqa_weight* table, int n_weights, devqa_ctx_t* out);
int devqa_ctx_destroy(devqa_ctx_t ctx);
int devqa_ctx_set_weight(devqa_ctx_t ctx, const char* name, const void* ptr);   /* re-point one table entry */

/* K2-K5 (R/editor/vllms_for_edit/blip2/blip2.py:25-45): pixel_values fp32 [B,3,S,S] -> projected query tokens fp32
 * [B, num_query_tokens, t_hidden]: im2col + patch GEMM + CLS/positions, v_layers pre-LN ViT layers (fused QKV GEMM, MFMA
 * attention, GELU FFN), post-LN, the Q-Former over the learned queries (self-attention, cross-attention to the image tokens
 * every q_cross_freq layers, query FFN; BERT post-LN residuals), language projection. */
int64_t devqa_vision_encode_workspace(devqa_ctx_t ctx, int B);
int devqa_vision_encode(devqa_ctx_t ctx, const float* pixel_values, int B, float* out_embeds, void* workspace, int64_t ws_bytes,
                        void* stream);

/* K7 (blip2.py:68-75, HF OPTDecoder): `n_layers` decoder layers (-1 = all) IN PLACE on the packed fp32 rows x [R, t_hidden]
 * (token / image-token embeddings + learned positions, devqa_embed_rows), sequences given by the attention descriptors of
 * devqa_attention (causal own range + optional visible prefix).  dense != 0: every row is a query row of some sequence
 * (otherwise the attention output is zero-filled first).  With stop_before_fc2 the LAST processed layer stops at its fc2 INPUT:
 * x then holds the residual stream before that layer's FFN add and out_fc2_in [R, t_ffn] (compute dtype) = relu(fc1(LN(x))) --
 * the frozen prefix of FT_VL (devqa_llm_prefix in SURVEY.md 8(b)): only fc2.weight of that layer changes during an edit. */
int64_t devqa_llm_layers_workspace(devqa_ctx_t ctx, int R, int stop_before_fc2);
int devqa_llm_layers(devqa_ctx_t ctx, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense, int n_layers,
                     int stop_before_fc2, void* out_fc2_in, void* workspace, int64_t ws_bytes, void* stream);
/* The general form: layers [first_layer, first_layer + n_layers) (n_layers = -1: up to the last), and `positions` = int32 [R] rotary
 * position of every packed row -- REQUIRED by the LLaMA-family decoders (RoPE is applied per layer; OPT's learned positions are already
 * in x, so it takes NULL).  LLaMA layers: RMSNorm -> fused q|k|v GEMM -> rotary q, k -> causal attention -> o_proj (+ residual) -> RMSNorm
 * -> fused gate|up GEMM -> SiLU(gate) * up -> down_proj (+ residual); stop_before_fc2 stops the last layer at its down_proj INPUT
 * (out_fc2_in [R, t_ffn]).  first_layer > 0 continues a forward whose earlier layers already ran on x (e.g. the layers an editor
 * modifies, after the frozen ones). */
int devqa_llm_layers_ex(devqa_ctx_t ctx, float* x, const int32_t* positions, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                        int first_layer, int n_layers, int stop_before_fc2, void* out_fc2_in, void* workspace, int64_t ws_bytes, void* stream);
/* devqa_llm_prefix (SURVEY.md 8(b)'s name): devqa_llm_layers over ALL layers with stop_before_fc2 = 1 */
int64_t devqa_llm_prefix_workspace(devqa_ctx_t ctx, int R);
int devqa_llm_prefix(devqa_ctx_t ctx, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense, void* out_fc2_in,
                     void* workspace, int64_t ws_bytes, void* stream);
/* K8: logits fp32 [R, t_vocab] = lm_head(final LayerNorm(rows [+ add])), rows / add fp32 [R, t_hidden] (tied embedding, no bias) */
int64_t devqa_llm_head_workspace(devqa_ctx_t ctx, int R);
int devqa_llm_head(devqa_ctx_t ctx, const float* rows, const float* add, int R, float* out_logits, void* workspace, int64_t ws_bytes,
                   void* stream);
/* K7 + K8: all layers on x (in place), then logits of the rows listed in want_rows (int32 [R_want], device) */
int64_t devqa_llm_forward_workspace(devqa_ctx_t ctx, int R, int R_want);
int devqa_llm_forward(devqa_ctx_t ctx, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                      const int32_t* want_rows, int R_want, float* out_logits, void* workspace, int64_t ws_bytes, void* stream);
/* the same with the rows' rotary positions (LLaMA-family decoders) */
int devqa_llm_forward_ex(devqa_ctx_t ctx, float* x, const int32_t* positions, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                         const int32_t* want_rows, int R_want, float* out_logits, void* workspace, int64_t ws_bytes, void* stream);

/* K9-K13 (R/editor/vllm_editors/ft_vl/ft_vl.py:111-158): the FT_VL inner loop for E concurrent edits of the last decoder
 * layer's fc2 matrix, control flow on the device (no host sync): per step  final LN + lm_head on the loss rows -> masked NLL +
 * dlogits -> loop control (skip the update under loss_floor, stop after it) -> dH = dlogits . E -> LN backward -> fused
 * rank-k gradient + AdamW + next fc2 rows (devqa_ft_adamw_step).
 *   w0        : pristine matrix, shared [t_hidden, npad] (w0_stride_e = 0) or per edit [E, t_hidden, npad] (= t_hidden * npad) --
 *               npad = t_ffn for the dense loop, or the padded count of ACTIVE columns (devqa_active_columns / devqa_gather_cols_*)
 *   a_rows    : fp32 [E, kmax, npad] fc2 inputs of the loss rows (padding rows zero);  resid_rows fp32 [E*kmax, t_hidden] residual
 *               (+ fc2 bias);  labels int32 [E*kmax];  mask fp32 [E, kmax] (1 = loss row);  1 <= kmax <= DEVQA_FT_MAX_ROWS
 *   out_delta : fp32 [E, t_hidden, npad] = W_e - w0 (0 for an edit that never updated);  out_losses fp32 [E, num_steps];
 *               out_steps int32 [E] executed steps;  out_updates int32 [E] AdamW updates taken */
typedef struct devqa_ft_cfg {
    int32_t num_steps;
    float lr, weight_decay, beta1, beta2, eps;
    float loss_floor;   /* 1e-2 in the reference (ft_vl.py:131,145) */
    float clamp_eps;    /* norm_constraint; < 0 disables */
} devqa_ft_cfg;
int64_t devqa_ft_edit_workspace(devqa_ctx_t ctx, int E, int kmax, int npad);
int devqa_ft_edit(devqa_ctx_t ctx, const float* w0, int64_t w0_stride_e, const float* a_rows, const float* resid_rows,
                  const int32_t* labels, const float* mask, int E, int kmax, int npad, const devqa_ft_cfg* cfg, float* out_delta,
                  float* out_losses, int32_t* out_steps, int32_t* out_updates, void* workspace, int64_t ws_bytes, void* stream);

/* K13 on a bound edit target (an fp32 master in the table): bind copies the pristine matrix into the context (the only device
 * memory a context owns); apply_delta: w += delta (ft_vl.py:56-61); restore: w = pristine (ft_vl.py:44-45); both refresh the
 * "<name>#shadow" entry when the table has one. */
int devqa_ctx_bind_edit_target(devqa_ctx_t ctx, const char* name, void* stream);
int devqa_apply_delta(devqa_ctx_t ctx, const float* delta, void* stream);
int devqa_restore(devqa_ctx_t ctx, void* stream);

/* K14 (R/evaluation/vllm_editor_eval.py:111,147-150): out_pred[r] = argmax logits_rows[r] (first max wins), out_acc[0] =
 * sum((pred == labels) * mask) / sum(mask).  logits fp32 [R, V] row stride ldl; labels int32 [R]; mask fp32 [R]. */
int devqa_token_acc(const float* logits_rows, int64_t ldl, int R, int V, const int32_t* labels, const float* mask, float* out_acc,
                    int32_t* out_pred, void* stream);

/* The single collective of a sharded run: every rank contributes n_rows x DEVQA_SCORE_COLS fp32 (blocks
 * padded to the same n_rows), every rank receives world x n_rows x DEVQA_SCORE_COLS in rank order -- one RCCL all-gather
 * over xGMI.  Rank 0 obtains the 128-byte id (devqa_comm_unique_id) and hands it to the other ranks by any host channel. */
int devqa_comm_unique_id(void* id128);
int devqa_comm_create(int rank, int world, const void* id128, int device, devqa_comm_t* out);
int devqa_comm_destroy(devqa_comm_t comm);
int devqa_gather_scores(devqa_comm_t comm, const float* local, int n_rows, float* out, void* stream);

/* ---- K16 / K17: MEND_VL's GradientTransform and the application of its low-rank deltas (devqa_mend_transform /
 * devqa_mend_apply).  R/editor/vllm_editors/mend_vl/auxiliary_networks.py:112-151 (GradientTransform.forward, inference mode), :62-83
 * (LRLinear), :20-24 (IDMLP); R/editor/vllm_editors/mend_vl/mend_vl.py:73-80 (forward_edit_hook).
 *   transform: rows idx[0..n) (nullptr: rows 0..n) of x [R, du] and delta [R, dv] (fp32) -> (x - u_mean) / (u_std + 1e-7) | same for
 *     delta (statistics nullptr: no normalisation) -> n_layers times  out = in + relu((in v^T) u^T + bias) * mode_scale + mode_shift
 *     with v [rank, D], u [D, rank], bias / mode_scale / mode_shift [D] (the mode row of the edited module already selected; D = du +
 *     dv) on the exact-fp32 GEMM (net->flags & DEVQA_MEND_SPLIT_BF16: three bf16 products of split operands) -> out_x [n, du], out_d [n, dv].
 *     D % 4 == 0, rank % 4 == 0.
 *   apply: y [R, dout] fp32 += (h [R, din] . xt^T) . dt with the factors xt [npad, din] and dtT [dout, npad] in the compute dtype
 *     (delta_W = xt^T dt is never materialised), npad % 64 == 0 (zero rows). */
#define DEVQA_MEND_MAX_LAYERS 4
typedef struct devqa_mend_layer {
    const float *u, *v, *bias, *mode_scale, *mode_shift;
} devqa_mend_layer;
#define DEVQA_MEND_SPLIT_BF16 1   /* flags: the two GEMMs of a layer as three bf16 MFMA products each of split operands (x = hi + lo, fp32 accumulation;
                                    ~2e-5 relative instead of the exact-fp32 GEMM's 1e-7, 3x its speed): what the bf16 compute mode asks for.  Needs D % 8 == 0
                                    and rank % 8 == 0, otherwise the exact form runs */
typedef struct devqa_mend_net {
    int32_t n_layers, rank;
    const float *u_mean, *u_std, *v_mean, *v_std;   /* nullptr: aux_model.norm = False */
    devqa_mend_layer layers[DEVQA_MEND_MAX_LAYERS];
    int32_t flags, reserved;
} devqa_mend_net;
int64_t devqa_mend_transform_workspace(int n, int du, int dv, int rank);
int devqa_mend_transform(const float* x, const float* delta, const int32_t* idx, int n, int du, int dv, const devqa_mend_net* net,
                         float* out_x, float* out_d, void* workspace, int64_t ws_bytes, void* stream);
int64_t devqa_mend_apply_workspace(int R, int npad, int compute_dtype);
int devqa_mend_apply(const void* h, const void* xt, const void* dtT, float* y, int R, int din, int dout, int npad, int compute_dtype,
                     void* workspace, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEVQA_H_ */
