// Path-level entry points of libdevqa_hip.so (include/devqa.h, "path level"): a model CONTEXT (device, dims, a table of named device
// weights) and the launch schedules of the BLIP-2 + FT_VL hot path over the op-level kernels of this library -- what a host in any
// language binds to drive the path without re-implementing the schedules:
//   devqa_vision_encode   K2-K5   ViT -> Q-Former -> language projection            (R/editor/vllms_for_edit/blip2/blip2.py:25-45)
//   devqa_llm_layers      K7      OPT decoder layers on packed rows, optionally up to the edited layer's fc2 INPUT (frozen prefix)
//   devqa_llm_head        K8      final LayerNorm + tied lm_head on given rows                                       (blip2.py:68-75)
//   devqa_llm_forward     K7+K8   layers -> gather rows -> head
//   devqa_ft_edit         K9-K13  the FT_VL inner loop for E concurrent edits, device-side control flow (ft_vl.py:111-158)
//   devqa_token_acc       K14     argmax + masked token accuracy                               (vllm_editor_eval.py:111,147-150)
//   devqa_apply_delta / devqa_restore   K13 on the context's edited matrix                      (ft_vl.py:44-45,56-61)
//   devqa_comm_* / devqa_gather_scores   the single collective of the sharded run (RCCL all-gather of score rows)
// No function here allocates device memory except devqa_ctx_create (the pristine copy of the edited matrix) and
// devqa_comm_create (RCCL's own); scratch is a caller-provided workspace sized by the matching *_workspace query.
#include <dlfcn.h>
#include <math.h>
#include <string.h>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

namespace {

struct Weight {
    const void* ptr;
    int dtype;        // DEVQA_DTYPE_*
    int ndim;
    int64_t shape[4];
};

struct Ctx {
    int device = 0;
    devqa_model_desc d{};
    bool bf16 = true;
    std::unordered_map<std::string, Weight> w;
    // edited matrix (devqa_apply_delta / devqa_restore)
    std::string edit_name;
    float* edit_w0 = nullptr;   // pristine copy, owned by the context
    int64_t edit_numel = 0;
};

std::mutex g_ctx_mu;
std::set<Ctx*> g_ctx_live;

// Validates the handle against the live set.  The pointer is used after the lock is dropped: destroying a context while another
// thread is inside a call on it is outside the contract (include/devqa.h: calls on a context are serialised by the caller).
Ctx* ctx_of(devqa_ctx_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(static_cast<uintptr_t>(h));
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    return g_ctx_live.count(c) ? c : nullptr;
}

inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// bump allocator over the caller's workspace; with base == nullptr it only measures
struct Arena {
    char* base;
    int64_t cap, off = 0;
    bool overflow = false;
    Arena(void* b, int64_t c) : base((char*)b), cap(c) {}
    void* take(int64_t bytes) {
        off = al256(off);
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (base && off > cap) overflow = true;
        return p;
    }
};

#define CTX_OR_FAIL(h)                                                     \
    Ctx* cp = ctx_of(h);                                                   \
    if (!cp) return devqa_fail(DEVQA_E_STATE, "invalid or destroyed context handle"); \
    Ctx& c = *cp

#define RC(expr)                       \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != DEVQA_OK) return rc_; \
    } while (0)

const Weight* find(const Ctx& c, const std::string& name) {
    auto it = c.w.find(name);
    return it == c.w.end() ? nullptr : &it->second;
}

// ---- dtype-generic wrappers over the op-level entry points (bf16 operands or the exact-fp32 "faithful" mode) -------------------
inline int esz(const Ctx& c) { return c.bf16 ? 2 : 4; }

int need(const Ctx& c, const std::string& name, const Weight** out, int dtype) {
    const Weight* w = find(c, name);
    if (!w) return devqa_fail(DEVQA_E_STATE, "weight table has no entry '%s'", name.c_str());
    if (w->dtype != dtype) return devqa_fail(DEVQA_E_STATE, "weight '%s' has dtype %d, the path needs %d", name.c_str(), w->dtype, dtype);
    *out = w;
    return DEVQA_OK;
}
int op_dtype(const Ctx& c) { return c.bf16 ? DEVQA_DTYPE_BF16 : DEVQA_DTYPE_F32; }

// out_act: operand-dtype output (bf16 or fp32), out_f32: fp32 output (bf16 mode may write both)
int gemm(const Ctx& c, const void* A, int64_t lda, const std::string& wname, const char* bname, int M, int N, int K, float alpha, int act,
         const float* residual, void* out_act, float* out_f32, hipStream_t st) {
    const Weight *W = nullptr, *B = nullptr;
    if (c.bf16) W = find(c, wname + "#shadow");      // an fp32 edit target: GEMMs read its compute-dtype shadow
    if (!W) RC(need(c, wname, &W, op_dtype(c)));
    if (bname) RC(need(c, bname, &B, DEVQA_DTYPE_F32));
    if (W->ndim != 2 || W->shape[0] != N || W->shape[1] != K)
        return devqa_fail(DEVQA_E_SHAPE, "weight '%s' is [%lld,%lld], the path needs [%d,%d]", wname.c_str(), (long long)W->shape[0],
                          (long long)W->shape[1], N, K);
    const float* bias = B ? (const float*)B->ptr : nullptr;
    if (c.bf16)
        return devqa_gemm_bf16((const devqa_bf16*)A, lda, (const devqa_bf16*)W->ptr, K, bias, M, N, K, alpha, act, residual,
                               (devqa_bf16*)out_act, out_f32, N, st);
    float* o = out_f32 ? out_f32 : (float*)out_act;
    return devqa_gemm_f32((const float*)A, lda, (const float*)W->ptr, K, bias, M, N, K, alpha, act, residual, o, N, st);
}

int layernorm(const Ctx& c, const float* x, const float* add, const std::string& prefix, int M, int D, float eps, void* out_act,
              float* out_f32, hipStream_t st) {
    const Weight *g = nullptr, *b = nullptr;
    RC(need(c, prefix + ".weight", &g, DEVQA_DTYPE_F32));
    RC(need(c, prefix + ".bias", &b, DEVQA_DTYPE_F32));
    if (c.bf16) return devqa_layernorm(x, add, (const float*)g->ptr, (const float*)b->ptr, M, D, eps, (devqa_bf16*)out_act, out_f32, st);
    float* o = out_f32 ? out_f32 : (float*)out_act;
    return devqa_layernorm(x, add, (const float*)g->ptr, (const float*)b->ptr, M, D, eps, nullptr, o, st);
}

inline bool llama_dec(const Ctx& c) { return c.d.family != DEVQA_FAMILY_BLIP2_OPT; }

int rmsnorm(const Ctx& c, const float* x, const float* add, const std::string& wname, int M, int D, float eps, void* out_act, float* out_f32,
            hipStream_t st) {
    const Weight* g = nullptr;
    RC(need(c, wname, &g, DEVQA_DTYPE_F32));
    if (c.bf16) return devqa_rmsnorm(x, add, (const float*)g->ptr, M, D, eps, (devqa_bf16*)out_act, out_f32, st);
    float* o = out_f32 ? out_f32 : (float*)out_act;
    return devqa_rmsnorm(x, add, (const float*)g->ptr, M, D, eps, nullptr, o, st);
}

int attention(const Ctx& c, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo,
              const int32_t* desc, int n_seq, int max_q, int H, int dh, int causal, hipStream_t st) {
    const float scale = (float)pow((double)dh, -0.5);   // dh ** -0.5 evaluated in double, then rounded: what the Python host passes
    if (c.bf16)
        return devqa_attention((const devqa_bf16*)q, ldq, (const devqa_bf16*)k, ldk, (const devqa_bf16*)v, ldv, (devqa_bf16*)out, ldo, desc,
                               n_seq, max_q, H, dh, scale, causal, st);
    return devqa_attention_f32((const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)out, ldo, desc, n_seq, max_q, H,
                               dh, scale, causal & 1, st);
}

// fp32 rows -> operand dtype (bf16: cast kernel; fp32: the rows themselves)
int to_act(const Ctx& c, const float* x32, void* out, int64_t n, const void** res, hipStream_t st) {
    if (!c.bf16) { *res = x32; return DEVQA_OK; }
    *res = out;
    return devqa_cast_f32_bf16(x32, (devqa_bf16*)out, n, st);
}

// n_seq independent, fully visible sequences.  Self-attention (self_rows): the keys are the sequence's own rows -- the form the
// self_full promise of devqa_attention describes; cross-attention: the keys are a visible range of another row set.
__global__ void full_desc_kernel(int32_t* desc, int n_seq, int q_len, int kv_len, int self_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seq) return;
    int32_t* d = desc + 6 * i;
    d[0] = i * q_len; d[1] = q_len;
    if (self_rows) { d[2] = 0; d[3] = 0; d[4] = i * q_len; d[5] = q_len; }
    else { d[2] = i * kv_len; d[3] = kv_len; d[4] = 0; d[5] = 0; }
}
__global__ void repeat_rows_kernel(const float* __restrict__ src, int rows, int D, int reps, float* __restrict__ dst) {
    const int64_t n = (int64_t)rows * D, total = n * reps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i % n];
}
unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

// =================================================================================================================================
// context
// =================================================================================================================================
extern "C" int devqa_ctx_create(int device, const devqa_model_desc* desc, const devqa_weight* table, int n_weights, devqa_ctx_t* out) {
    DEVQA_CHECK_ARG(desc && table && out && n_weights > 0, "ctx_create: null argument");
    DEVQA_CHECK_ARG(desc->family == DEVQA_FAMILY_BLIP2_OPT || desc->family == DEVQA_FAMILY_LLAVA || desc->family == DEVQA_FAMILY_MINIGPT4,
                    "ctx_create: unknown model family %d", desc->family);
    DEVQA_CHECK_ARG(desc->compute_dtype == DEVQA_DTYPE_BF16 || desc->compute_dtype == DEVQA_DTYPE_F32, "ctx_create: bad compute dtype");
    DEVQA_CHECK_SHAPE(desc->v_heads > 0 && desc->t_heads > 0 && desc->v_hidden % desc->v_heads == 0 && desc->t_hidden % desc->t_heads == 0,
                      "ctx_create: hidden sizes must be multiples of the head counts");
    DEVQA_CHECK_SHAPE(desc->patch_size > 0 && desc->image_size % desc->patch_size == 0, "ctx_create: bad vision geometry");
    if (desc->family != DEVQA_FAMILY_LLAVA)     /* a Q-Former sits between the ViT and the decoder */
        DEVQA_CHECK_SHAPE(desc->q_heads > 0 && desc->q_hidden % desc->q_heads == 0 && desc->q_cross_freq >= 1, "ctx_create: bad Q-Former geometry");
    if (desc->family != DEVQA_FAMILY_BLIP2_OPT) /* LLaMA decoder */
        DEVQA_CHECK_SHAPE(desc->t_rms_eps > 0.f && desc->t_rope_theta > 0.f && (desc->t_hidden / desc->t_heads) % 2 == 0,
                          "ctx_create: the LLaMA decoder needs t_rms_eps, t_rope_theta and an even head size");
    if (desc->family == DEVQA_FAMILY_LLAVA)
        DEVQA_CHECK_SHAPE(desc->v_run_layers >= 1 && desc->v_run_layers <= desc->v_layers, "ctx_create: v_run_layers=%d", desc->v_run_layers);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return devqa_fail(DEVQA_E_ARG, "ctx_create: no device %d", device);
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return devqa_fail(DEVQA_E_OOM, "ctx_create: host allocation failed");
    c->device = device;
    c->d = *desc;
    c->bf16 = desc->compute_dtype == DEVQA_DTYPE_BF16;
    for (int i = 0; i < n_weights; ++i) {
        const devqa_weight& t = table[i];
        if (!t.name || !t.ptr || t.ndim < 1 || t.ndim > 4) {
            delete c;
            return devqa_fail(DEVQA_E_ARG, "ctx_create: bad weight table entry %d", i);
        }
        Weight w{t.ptr, t.dtype, t.ndim, {1, 1, 1, 1}};
        for (int k = 0; k < t.ndim; ++k) w.shape[k] = t.shape[k];
        c->w[t.name] = w;
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        g_ctx_live.insert(c);
    }
    *out = (devqa_ctx_t)(uintptr_t)c;
    return DEVQA_OK;
}

extern "C" int devqa_ctx_destroy(devqa_ctx_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(static_cast<uintptr_t>(h));
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        if (!g_ctx_live.erase(c)) return devqa_fail(DEVQA_E_STATE, "ctx_destroy: invalid or destroyed context handle");
    }
    if (c->edit_w0) (void)hipFree(c->edit_w0);
    delete c;
    return DEVQA_OK;
}

extern "C" int devqa_ctx_set_weight(devqa_ctx_t h, const char* name, const void* ptr) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(name && ptr, "ctx_set_weight: null argument");
    auto it = c.w.find(name);
    if (it == c.w.end()) return devqa_fail(DEVQA_E_STATE, "ctx_set_weight: no entry '%s'", name);
    it->second.ptr = ptr;
    return DEVQA_OK;
}

// =================================================================================================================================
// K2-K5: images -> projected query tokens
// =================================================================================================================================
namespace {
struct VisionPlan {
    void *cols, *h, *qkv, *att, *f, *qq, *kk, *vv, *qatt, *hb, *qf;
    float *patches, *x, *h32a, *h32b, *o32, *qln;
    int32_t *d_vit, *d_self, *d_cross;
};
int64_t plan_vision(const Ctx& c, int B, Arena& a, VisionPlan& p) {
    const auto& d = c.d;
    const int G = d.image_size / d.patch_size, NP = G * G, N = NP + 1, D = d.v_hidden, F = d.v_ffn, Q = d.num_query_tokens, dq = d.q_hidden;
    const int kpad = (3 * d.patch_size * d.patch_size + 63) / 64 * 64;
    const int64_t e = esz(c), R = (int64_t)B * N, RQ = (int64_t)B * Q;
    p.cols = a.take((int64_t)B * NP * kpad * e);
    p.patches = (float*)a.take((int64_t)B * NP * D * 4);
    p.x = (float*)a.take(R * D * 4);
    p.h = a.take(R * D * e);
    p.qkv = a.take(R * 3 * D * e);
    p.att = a.take(R * D * e);
    p.f = a.take(R * F * e);
    p.qln = (float*)a.take((int64_t)Q * dq * 4);
    p.h32a = (float*)a.take(RQ * dq * 4);
    p.h32b = (float*)a.take(RQ * dq * 4);
    p.o32 = (float*)a.take(RQ * dq * 4);
    p.hb = a.take(RQ * dq * e);
    p.qq = a.take(RQ * dq * e);
    const int64_t RK = R > RQ ? R : RQ;   // keys / values: the Q queries themselves (self-attention) or the N image tokens (cross)
    p.kk = a.take(RK * dq * e);
    p.vv = a.take(RK * dq * e);
    p.qatt = a.take(RQ * dq * e);
    p.qf = a.take(RQ * d.q_ffn * e);
    p.d_vit = (int32_t*)a.take((int64_t)B * 24);
    p.d_self = (int32_t*)a.take((int64_t)B * 24);
    p.d_cross = (int32_t*)a.take((int64_t)B * 24);
    return al256(a.off);
}

// BERT-style attention block of the Q-Former (HF Blip2QFormerAttention): h32 <- LN(dense(attn(q(h), k(src), v(src))) + h32)
int bert_attention(const Ctx& c, const std::string& p, float* h32_in, float* h32_out, const void* kv_src, int kv_rows, int kv_dim,
                   const int32_t* desc, int B, VisionPlan& w, hipStream_t st) {
    const auto& d = c.d;
    const int Q = d.num_query_tokens, dq = d.q_hidden, H = d.q_heads, dh = dq / H;
    const int RQ = B * Q;
    const void* hb = nullptr;
    RC(to_act(c, h32_in, w.hb, (int64_t)RQ * dq, &hb, st));
    const void* src = kv_src ? kv_src : hb;
    const int srows = kv_src ? kv_rows : RQ, sdim = kv_src ? kv_dim : dq;
    RC(gemm(c, hb, dq, p + "attention.query.weight", (p + "attention.query.bias").c_str(), RQ, dq, dq, 1.f, DEVQA_ACT_NONE, nullptr, w.qq, nullptr, st));
    RC(gemm(c, src, sdim, p + "attention.key.weight", (p + "attention.key.bias").c_str(), srows, dq, sdim, 1.f, DEVQA_ACT_NONE, nullptr, w.kk, nullptr, st));
    RC(gemm(c, src, sdim, p + "attention.value.weight", (p + "attention.value.bias").c_str(), srows, dq, sdim, 1.f, DEVQA_ACT_NONE, nullptr, w.vv, nullptr, st));
    RC(attention(c, w.qq, dq, w.kk, dq, w.vv, dq, w.qatt, dq, desc, B, Q, H, dh, 0, st));
    RC(gemm(c, w.qatt, dq, p + "output.dense.weight", (p + "output.dense.bias").c_str(), RQ, dq, dq, 1.f, DEVQA_ACT_NONE, nullptr, nullptr, w.o32, st));
    return layernorm(c, w.o32, h32_in, p + "output.LayerNorm", RQ, dq, d.q_ln_eps, nullptr, h32_out, st);
}
}  // namespace

// ---- LLaVA-1.5 (R/editor/vllms_for_edit/llava/llava.py:25-51): CLIP ViT hidden state of layer v_run_layers (HF hidden_states[-2]) without
// the CLS row -> multi_modal_projector (Linear, GELU, Linear) -> [B, NP, t_hidden] ------------------------------------------------------
namespace {
struct ClipPlan {
    void *cols, *h, *qkv, *att, *f, *g, *p1;
    float *patches, *x0, *x, *rows;
    int32_t *d_vit, *idx;
};
int64_t plan_clip(const Ctx& c, int B, Arena& a, ClipPlan& p) {
    const auto& d = c.d;
    const int G = d.image_size / d.patch_size, NP = G * G, N = NP + 1, D = d.v_hidden, F = d.v_ffn;
    const int kpad = (3 * d.patch_size * d.patch_size + 63) / 64 * 64;
    const int64_t e = esz(c), R = (int64_t)B * N, RP = (int64_t)B * NP;
    p.cols = a.take(RP * kpad * e);
    p.patches = (float*)a.take(RP * D * 4);
    p.x0 = (float*)a.take(R * D * 4);
    p.x = (float*)a.take(R * D * 4);
    p.h = a.take(R * D * e);
    p.qkv = a.take(R * 3 * D * e);
    p.att = a.take(R * D * e);
    p.f = a.take(R * F * e);
    p.rows = (float*)a.take(RP * D * 4);
    p.g = a.take(RP * D * e);
    p.p1 = a.take(RP * d.t_hidden * e);
    p.d_vit = (int32_t*)a.take((int64_t)B * 24);
    p.idx = (int32_t*)a.take(RP * 4);
    return al256(a.off);
}
__global__ void drop_cls_idx_kernel(int32_t* idx, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (N - 1)) return;
    idx[i] = (i / (N - 1)) * N + 1 + i % (N - 1);
}
int clip_vision_encode(const Ctx& c, const float* pixel_values, int B, float* out_embeds, void* workspace, int64_t ws_bytes, hipStream_t st) {
    const auto& d = c.d;
    const int P = d.patch_size, G = d.image_size / P, NP = G * G, N = NP + 1, D = d.v_hidden, F = d.v_ffn, H = d.v_heads, dh = D / H;
    const int R = B * N, RP = B * NP;
    const int kpad = (3 * P * P + 63) / 64 * 64;
    Arena a(workspace, ws_bytes);
    ClipPlan w;
    plan_clip(c, B, a, w);
    if (a.overflow) return devqa_fail(DEVQA_E_SHAPE, "vision_encode: workspace of %lld bytes is too small (need %lld)", (long long)ws_bytes, (long long)al256(a.off));
    const std::string pre = "vision_tower.vision_model.";
    if (c.bf16) RC(devqa_im2col_patches(pixel_values, B, d.image_size, P, kpad, (devqa_bf16*)w.cols, st));
    else RC(devqa_im2col_patches_f32(pixel_values, B, d.image_size, P, kpad, (float*)w.cols, st));
    RC(gemm(c, w.cols, kpad, "derived.patch_w_gemm", nullptr, RP, D, kpad, 1.f, DEVQA_ACT_NONE, nullptr, nullptr, w.patches, st));     // CLIP: no conv bias
    const Weight *cls = nullptr, *pos = nullptr;
    RC(need(c, pre + "embeddings.class_embedding", &cls, DEVQA_DTYPE_F32));
    RC(need(c, pre + "embeddings.position_embedding.weight", &pos, DEVQA_DTYPE_F32));
    RC(devqa_vit_assemble(w.patches, (const float*)cls->ptr, (const float*)pos->ptr, B, NP, D, w.x0, st));
    {
        const bool keep = c.bf16;      // (the fp32 LayerNorm wrapper writes to out_f32 either way)
        (void)keep;
        const Weight *g = nullptr, *b = nullptr;
        RC(need(c, pre + "pre_layrnorm.weight", &g, DEVQA_DTYPE_F32));
        RC(need(c, pre + "pre_layrnorm.bias", &b, DEVQA_DTYPE_F32));
        RC(devqa_layernorm(w.x0, nullptr, (const float*)g->ptr, (const float*)b->ptr, R, D, d.v_ln_eps, nullptr, w.x, st));
    }
    hipLaunchKernelGGL(full_desc_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.d_vit, B, N, N, 0);    // keys = a visible range (the own rows)
    hipLaunchKernelGGL(drop_cls_idx_kernel, dim3((RP + 255) / 256), dim3(256), 0, st, w.idx, B, N);
    DEVQA_LAUNCH_CHECK("clip descriptors");
    char buf[200];
    const int64_t e = esz(c);
    for (int i = 0; i < d.v_run_layers; ++i) {
        snprintf(buf, sizeof(buf), "%sencoder.layers.%d.", pre.c_str(), i);
        const std::string p(buf);
        snprintf(buf, sizeof(buf), "derived.clip_qkv.%d", i);
        const std::string fq(buf);
        RC(layernorm(c, w.x, nullptr, p + "layer_norm1", R, D, d.v_ln_eps, w.h, nullptr, st));
        RC(gemm(c, w.h, D, fq + ".weight", (fq + ".bias").c_str(), R, 3 * D, D, 1.f, DEVQA_ACT_NONE, nullptr, w.qkv, nullptr, st));
        const char* qkv = (const char*)w.qkv;
        RC(attention(c, qkv, 3 * D, qkv + (int64_t)D * e, 3 * D, qkv + (int64_t)2 * D * e, 3 * D, w.att, D, w.d_vit, B, N, H, dh, 0, st));
        RC(gemm(c, w.att, D, p + "self_attn.out_proj.weight", (p + "self_attn.out_proj.bias").c_str(), R, D, D, 1.f, DEVQA_ACT_NONE, w.x, nullptr, w.x, st));
        RC(layernorm(c, w.x, nullptr, p + "layer_norm2", R, D, d.v_ln_eps, w.h, nullptr, st));
        RC(gemm(c, w.h, D, p + "mlp.fc1.weight", (p + "mlp.fc1.bias").c_str(), R, F, D, 1.f, DEVQA_ACT_QUICK_GELU, nullptr, w.f, nullptr, st));
        RC(gemm(c, w.f, F, p + "mlp.fc2.weight", (p + "mlp.fc2.bias").c_str(), R, D, F, 1.f, DEVQA_ACT_NONE, w.x, nullptr, w.x, st));
    }
    RC(devqa_gather_rows(w.x, w.idx, RP, D, 4, w.rows, st));
    const void* g = nullptr;
    RC(to_act(c, w.rows, w.g, (int64_t)RP * D, &g, st));
    RC(gemm(c, g, D, "multi_modal_projector.linear_1.weight", "multi_modal_projector.linear_1.bias", RP, d.t_hidden, D, 1.f, DEVQA_ACT_GELU, nullptr, w.p1,
            nullptr, st));
    return gemm(c, w.p1, d.t_hidden, "multi_modal_projector.linear_2.weight", "multi_modal_projector.linear_2.bias", RP, d.t_hidden, d.t_hidden, 1.f,
                DEVQA_ACT_NONE, nullptr, nullptr, out_embeds, st);
}
}  // namespace

extern "C" int64_t devqa_vision_encode_workspace(devqa_ctx_t h, int B) {
    Ctx* c = ctx_of(h);
    if (!c || B <= 0) return -1;
    Arena a(nullptr, 0);
    if (c->d.family == DEVQA_FAMILY_LLAVA) {
        ClipPlan p;
        return plan_clip(*c, B, a, p);
    }
    VisionPlan p;
    return plan_vision(*c, B, a, p);
}

extern "C" int devqa_vision_encode(devqa_ctx_t h, const float* pixel_values, int B, float* out_embeds, void* workspace, int64_t ws_bytes,
                                   void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(pixel_values && out_embeds && workspace, "vision_encode: null pointer");
    if (B == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(B > 0, "vision_encode: B=%d", B);
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "vision_encode: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (c.d.family == DEVQA_FAMILY_LLAVA) return clip_vision_encode(c, pixel_values, B, out_embeds, workspace, ws_bytes, st);
    const auto& d = c.d;
    const int P = d.patch_size, G = d.image_size / P, NP = G * G, N = NP + 1, D = d.v_hidden, F = d.v_ffn, H = d.v_heads, dh = D / H;
    const int Q = d.num_query_tokens, dq = d.q_hidden, R = B * N, RQ = B * Q;
    const int kpad = (3 * P * P + 63) / 64 * 64;
    Arena a(workspace, ws_bytes);
    VisionPlan w;
    plan_vision(c, B, a, w);
    if (a.overflow) return devqa_fail(DEVQA_E_SHAPE, "vision_encode: workspace of %lld bytes is too small (need %lld)", (long long)ws_bytes, (long long)al256(a.off));
    // ---- K2: patch embedding = im2col + GEMM, CLS + positions ----
    if (c.bf16) RC(devqa_im2col_patches(pixel_values, B, d.image_size, P, kpad, (devqa_bf16*)w.cols, st));
    else RC(devqa_im2col_patches_f32(pixel_values, B, d.image_size, P, kpad, (float*)w.cols, st));
    RC(gemm(c, w.cols, kpad, "derived.patch_w_gemm", "vision_model.embeddings.patch_embedding.bias", B * NP, D, kpad, 1.f, DEVQA_ACT_NONE, nullptr,
            nullptr, w.patches, st));
    const Weight *cls = nullptr, *pos = nullptr;
    RC(need(c, "vision_model.embeddings.class_embedding", &cls, DEVQA_DTYPE_F32));
    RC(need(c, "vision_model.embeddings.position_embedding", &pos, DEVQA_DTYPE_F32));
    RC(devqa_vit_assemble(w.patches, (const float*)cls->ptr, (const float*)pos->ptr, B, NP, D, w.x, st));
    hipLaunchKernelGGL(full_desc_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.d_vit, B, N, N, 1);
    hipLaunchKernelGGL(full_desc_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.d_self, B, Q, Q, 1);
    hipLaunchKernelGGL(full_desc_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.d_cross, B, Q, N, 0);
    DEVQA_LAUNCH_CHECK("full_desc");
    // ---- K3: ViT encoder (pre-LN) ----
    char buf[160];
    const int64_t e = esz(c);
    for (int i = 0; i < d.v_layers; ++i) {
        snprintf(buf, sizeof(buf), "vision_model.encoder.layers.%d.", i);
        const std::string p(buf);
        RC(layernorm(c, w.x, nullptr, p + "layer_norm1", R, D, d.v_ln_eps, w.h, nullptr, st));
        RC(gemm(c, w.h, D, p + "self_attn.qkv.weight", (p + "self_attn.qkv.bias").c_str(), R, 3 * D, D, 1.f, DEVQA_ACT_NONE, nullptr, w.qkv, nullptr, st));
        const char* qkv = (const char*)w.qkv;
        RC(attention(c, qkv, 3 * D, qkv + (int64_t)D * e, 3 * D, qkv + (int64_t)2 * D * e, 3 * D, w.att, D, w.d_vit, B, N, H, dh, 4, st));
        RC(gemm(c, w.att, D, p + "self_attn.projection.weight", (p + "self_attn.projection.bias").c_str(), R, D, D, 1.f, DEVQA_ACT_NONE, w.x, nullptr, w.x, st));
        RC(layernorm(c, w.x, nullptr, p + "layer_norm2", R, D, d.v_ln_eps, w.h, nullptr, st));
        RC(gemm(c, w.h, D, p + "mlp.fc1.weight", (p + "mlp.fc1.bias").c_str(), R, F, D, 1.f, DEVQA_ACT_GELU, nullptr, w.f, nullptr, st));
        RC(gemm(c, w.f, F, p + "mlp.fc2.weight", (p + "mlp.fc2.bias").c_str(), R, D, F, 1.f, DEVQA_ACT_NONE, w.x, nullptr, w.x, st));
    }
    RC(layernorm(c, w.x, nullptr, "vision_model.post_layernorm", R, D, d.v_ln_eps, w.h, nullptr, st));   // image tokens, operand dtype
    // ---- K4: Q-Former over the learned queries ----
    const Weight* qt = nullptr;
    RC(need(c, "query_tokens", &qt, DEVQA_DTYPE_F32));
    RC(layernorm(c, (const float*)qt->ptr, nullptr, "qformer.layernorm", Q, dq, d.q_ln_eps, nullptr, w.qln, st));
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(grid_for((int64_t)RQ * dq)), dim3(256), 0, st, w.qln, Q, dq, B, w.h32a);
    DEVQA_LAUNCH_CHECK("repeat_rows");
    float *cur = w.h32a, *nxt = w.h32b;
    for (int i = 0; i < d.q_layers; ++i) {
        snprintf(buf, sizeof(buf), "qformer.encoder.layer.%d.", i);
        const std::string p(buf);
        RC(bert_attention(c, p + "attention.", cur, nxt, nullptr, 0, 0, w.d_self, B, w, st));
        std::swap(cur, nxt);
        if (i % d.q_cross_freq == 0) {
            RC(bert_attention(c, p + "crossattention.", cur, nxt, w.h, R, D, w.d_cross, B, w, st));
            std::swap(cur, nxt);
        }
        const void* hb = nullptr;
        RC(to_act(c, cur, w.hb, (int64_t)RQ * dq, &hb, st));
        RC(gemm(c, hb, dq, p + "intermediate_query.dense.weight", (p + "intermediate_query.dense.bias").c_str(), RQ, d.q_ffn, dq, 1.f, DEVQA_ACT_GELU,
                nullptr, w.qf, nullptr, st));
        RC(gemm(c, w.qf, d.q_ffn, p + "output_query.dense.weight", (p + "output_query.dense.bias").c_str(), RQ, dq, d.q_ffn, 1.f, DEVQA_ACT_NONE, nullptr,
                nullptr, w.o32, st));
        RC(layernorm(c, w.o32, cur, p + "output_query.LayerNorm", RQ, dq, d.q_ln_eps, nullptr, nxt, st));
        std::swap(cur, nxt);
    }
    // ---- K5: language projection ----
    const void* hb = nullptr;
    RC(to_act(c, cur, w.hb, (int64_t)RQ * dq, &hb, st));
    return gemm(c, hb, dq, "language_projection.weight", "language_projection.bias", RQ, d.t_hidden, dq, 1.f, DEVQA_ACT_NONE, nullptr, nullptr,
                out_embeds, st);
}

// =================================================================================================================================
// K7 / K8: decoder on packed rows
// =================================================================================================================================
namespace {
struct LlmPlan {
    void *h, *qkv, *att, *a, *gu;
};
int64_t plan_llm(const Ctx& c, int R, bool own_a, Arena& ar, LlmPlan& p) {
    const int64_t e = esz(c), d = c.d.t_hidden;
    p.h = ar.take((int64_t)R * d * e);
    p.qkv = ar.take((int64_t)R * 3 * d * e);
    p.att = ar.take((int64_t)R * d * e);
    p.a = own_a ? ar.take((int64_t)R * c.d.t_ffn * e) : nullptr;
    p.gu = llama_dec(c) ? ar.take((int64_t)R * 2 * c.d.t_ffn * e) : nullptr;     // fused [gate | up] projection output
    return al256(ar.off);
}
// LLaMA / Vicuna decoder layers [first, first + n_layers) (HF LlamaDecoderLayer: RMSNorm, rotary q / k, causal attention, SwiGLU FFN; no
// biases) in place on x; positions: int32 [R] rotary position of every row.  R/editor/vllms_for_edit/llava/llava.py:63-68,
// .../minigpt4/minigpt4.py:63-68.
int llama_layers(const Ctx& c, float* x, const int32_t* pos, const int32_t* desc, int n_seq, int max_len, int R, int dense, int first, int n_layers,
                 int stop_before_fc2, void* a_out, LlmPlan& w, hipStream_t st) {
    const auto& d = c.d;
    const int D = d.t_hidden, H = d.t_heads, dh = D / H, F = d.t_ffn;
    const int64_t e = esz(c);
    char buf[160];
    for (int i = first; i < first + n_layers; ++i) {
        snprintf(buf, sizeof(buf), "language_model.model.layers.%d.", i);
        const std::string p(buf);
        RC(rmsnorm(c, x, nullptr, p + "input_layernorm.weight", R, D, d.t_rms_eps, w.h, nullptr, st));
        snprintf(buf, sizeof(buf), "derived.llama_qkv.%d.weight", i);
        RC(gemm(c, w.h, D, buf, nullptr, R, 3 * D, D, 1.f, DEVQA_ACT_NONE, nullptr, w.qkv, nullptr, st));
        if (c.bf16) RC(devqa_rope_bf16((devqa_bf16*)w.qkv, 3 * D, R, pos, 2 * H, dh, d.t_rope_theta, st));    // q heads then k heads
        else RC(devqa_rope_f32((float*)w.qkv, 3 * D, R, pos, 2 * H, dh, d.t_rope_theta, st));
        if (!dense && hipMemsetAsync(w.att, 0, (size_t)R * D * e, st) != hipSuccess) return devqa_fail(DEVQA_E_HIP, "llm_layers: memset failed");
        const char* qkv = (const char*)w.qkv;
        RC(attention(c, qkv, 3 * D, qkv + (int64_t)D * e, 3 * D, qkv + (int64_t)2 * D * e, 3 * D, w.att, D, desc, n_seq, max_len, H, dh, 1, st));
        RC(gemm(c, w.att, D, p + "self_attn.o_proj.weight", nullptr, R, D, D, 1.f, DEVQA_ACT_NONE, x, nullptr, x, st));
        RC(rmsnorm(c, x, nullptr, p + "post_attention_layernorm.weight", R, D, d.t_rms_eps, w.h, nullptr, st));
        const bool last_stop = stop_before_fc2 && i == first + n_layers - 1;
        void* a = last_stop ? a_out : w.a;
        // SwiGLU inside the [gate | up] GEMM's epilogue when the host asked for it (t_flags), registered the row-interleaved operand and the
        // call shape takes the 256 x 256 kernel: no [R, 2F] intermediate, no second pass (2.7 GB per layer at R = 40 k, F = 11008)
        snprintf(buf, sizeof(buf), "derived.llama_gu_il.%d.weight", i);
        const Weight* wil = (c.bf16 && (d.t_flags & DEVQA_DESC_FUSE_SWIGLU)) ? find(c, buf) : nullptr;
        if (wil && wil->dtype == DEVQA_DTYPE_BF16 && devqa_gemm_bf16_swiglu_supported(R, 2 * F, D)) {
            RC(devqa_gemm_bf16((const devqa_bf16*)w.h, D, (const devqa_bf16*)wil->ptr, D, nullptr, R, 2 * F, D, 1.f, DEVQA_ACT_SWIGLU_IL16, nullptr,
                               (devqa_bf16*)a, nullptr, F, st));
        } else {
            snprintf(buf, sizeof(buf), "derived.llama_gu.%d.weight", i);
            RC(gemm(c, w.h, D, buf, nullptr, R, 2 * F, D, 1.f, DEVQA_ACT_NONE, nullptr, w.gu, nullptr, st));
            if (c.bf16) RC(devqa_swiglu_bf16((const devqa_bf16*)w.gu, R, F, (devqa_bf16*)a, st));
            else RC(devqa_swiglu_f32((const float*)w.gu, R, F, (float*)a, st));
        }
        if (last_stop) return DEVQA_OK;
        RC(gemm(c, a, F, p + "mlp.down_proj.weight", nullptr, R, D, F, 1.f, DEVQA_ACT_NONE, x, nullptr, x, st));
    }
    return DEVQA_OK;
}
int llm_layers(const Ctx& c, float* x, const int32_t* pos, const int32_t* desc, int n_seq, int max_len, int R, int dense, int first, int n_layers,
               int stop_before_fc2, void* a_out, LlmPlan& w, hipStream_t st) {
    if (llama_dec(c)) return llama_layers(c, x, pos, desc, n_seq, max_len, R, dense, first, n_layers, stop_before_fc2, a_out, w, st);
    const auto& d = c.d;
    const int D = d.t_hidden, H = d.t_heads, dh = D / H, F = d.t_ffn;
    const int64_t e = esz(c);
    char buf[160];
    for (int i = first; i < first + n_layers; ++i) {
        snprintf(buf, sizeof(buf), "language_model.model.decoder.layers.%d.", i);
        const std::string p(buf);
        RC(layernorm(c, x, nullptr, p + "self_attn_layer_norm", R, D, d.t_ln_eps, w.h, nullptr, st));
        snprintf(buf, sizeof(buf), "derived.dec_qkv.%d", i);
        const std::string fq(buf);
        RC(gemm(c, w.h, D, fq + ".weight", (fq + ".bias").c_str(), R, 3 * D, D, 1.f, DEVQA_ACT_NONE, nullptr, w.qkv, nullptr, st));
        if (!dense && hipMemsetAsync(w.att, 0, (size_t)R * D * e, st) != hipSuccess) return devqa_fail(DEVQA_E_HIP, "llm_layers: memset failed");
        const char* qkv = (const char*)w.qkv;
        RC(attention(c, qkv, 3 * D, qkv + (int64_t)D * e, 3 * D, qkv + (int64_t)2 * D * e, 3 * D, w.att, D, desc, n_seq, max_len, H, dh, 1, st));
        RC(gemm(c, w.att, D, p + "self_attn.out_proj.weight", (p + "self_attn.out_proj.bias").c_str(), R, D, D, 1.f, DEVQA_ACT_NONE, x, nullptr, x, st));
        RC(layernorm(c, x, nullptr, p + "final_layer_norm", R, D, d.t_ln_eps, w.h, nullptr, st));
        const bool last_stop = stop_before_fc2 && i == first + n_layers - 1;
        void* a = last_stop ? a_out : w.a;
        RC(gemm(c, w.h, D, p + "fc1.weight", (p + "fc1.bias").c_str(), R, F, D, 1.f, DEVQA_ACT_RELU, nullptr, a, nullptr, st));
        if (last_stop) return DEVQA_OK;
        RC(gemm(c, a, F, p + "fc2.weight", (p + "fc2.bias").c_str(), R, D, F, 1.f, DEVQA_ACT_NONE, x, nullptr, x, st));
    }
    return DEVQA_OK;
}
int llm_head(const Ctx& c, const float* rows, const float* add, int R, float* logits, void* h_ws, hipStream_t st) {
    if (llama_dec(c)) {      // RMSNorm + the (untied) lm_head
        RC(rmsnorm(c, rows, add, "language_model.model.norm.weight", R, c.d.t_hidden, c.d.t_rms_eps, h_ws, nullptr, st));
        return gemm(c, h_ws, c.d.t_hidden, "language_model.lm_head.weight", nullptr, R, c.d.t_vocab, c.d.t_hidden, 1.f, DEVQA_ACT_NONE, nullptr, nullptr,
                    logits, st);
    }
    RC(layernorm(c, rows, add, "language_model.model.decoder.final_layer_norm", R, c.d.t_hidden, c.d.t_ln_eps, h_ws, nullptr, st));
    return gemm(c, h_ws, c.d.t_hidden, "language_model.model.decoder.embed_tokens.weight", nullptr, R, c.d.t_vocab, c.d.t_hidden, 1.f, DEVQA_ACT_NONE,
                nullptr, nullptr, logits, st);
}
}  // namespace

extern "C" int64_t devqa_llm_layers_workspace(devqa_ctx_t h, int R, int stop_before_fc2) {
    Ctx* c = ctx_of(h);
    if (!c || R <= 0) return -1;
    Arena a(nullptr, 0);
    LlmPlan p;
    // the FFN activation buffer is the caller's `out_fc2_in` on the stopping layer, the workspace's on all others
    (void)stop_before_fc2;
    return plan_llm(*c, R, true, a, p);
}

extern "C" int devqa_llm_layers_ex(devqa_ctx_t h, float* x, const int32_t* positions, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                                   int first_layer, int n_layers, int stop_before_fc2, void* out_fc2_in, void* workspace, int64_t ws_bytes, void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(x && seq_desc && workspace, "llm_layers: null pointer");
    DEVQA_CHECK_ARG(!stop_before_fc2 || out_fc2_in, "llm_layers: stop_before_fc2 needs out_fc2_in");
    DEVQA_CHECK_ARG(!llama_dec(c) || positions, "llm_layers: the LLaMA decoder needs the rotary positions of the rows (devqa_llm_layers_ex)");
    if (R == 0 || n_seq == 0) return DEVQA_OK;
    if (n_layers < 0) n_layers = c.d.t_layers - first_layer;
    DEVQA_CHECK_SHAPE(R > 0 && n_seq > 0 && max_len > 0 && first_layer >= 0 && n_layers >= 1 && first_layer + n_layers <= c.d.t_layers,
                      "llm_layers: bad dims R=%d n_seq=%d first=%d layers=%d", R, n_seq, first_layer, n_layers);
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "llm_layers: workspace must be 256-byte aligned");
    Arena a(workspace, ws_bytes);
    LlmPlan w;
    plan_llm(c, R, true, a, w);
    if (a.overflow) return devqa_fail(DEVQA_E_SHAPE, "llm_layers: workspace of %lld bytes is too small (need %lld)", (long long)ws_bytes, (long long)al256(a.off));
    return llm_layers(c, x, positions, seq_desc, n_seq, max_len, R, dense, first_layer, n_layers, stop_before_fc2, out_fc2_in, w, (hipStream_t)stream);
}

extern "C" int devqa_llm_layers(devqa_ctx_t h, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense, int n_layers,
                                int stop_before_fc2, void* out_fc2_in, void* workspace, int64_t ws_bytes, void* stream) {
    return devqa_llm_layers_ex(h, x, nullptr, seq_desc, n_seq, max_len, R, dense, 0, n_layers, stop_before_fc2, out_fc2_in, workspace, ws_bytes, stream);
}

// SURVEY.md 8(b)'s name for the frozen prefix of FT_VL: all layers, the last one stopping at its fc2 input
extern "C" int64_t devqa_llm_prefix_workspace(devqa_ctx_t h, int R) { return devqa_llm_layers_workspace(h, R, 1); }
extern "C" int devqa_llm_prefix(devqa_ctx_t h, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense, void* out_fc2_in,
                                void* workspace, int64_t ws_bytes, void* stream) {
    return devqa_llm_layers(h, x, seq_desc, n_seq, max_len, R, dense, -1, 1, out_fc2_in, workspace, ws_bytes, stream);
}

extern "C" int64_t devqa_llm_head_workspace(devqa_ctx_t h, int R) {
    Ctx* c = ctx_of(h);
    if (!c || R <= 0) return -1;
    return al256((int64_t)R * c->d.t_hidden * esz(*c));
}

extern "C" int devqa_llm_head(devqa_ctx_t h, const float* rows, const float* add, int R, float* out_logits, void* workspace, int64_t ws_bytes,
                              void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(rows && out_logits && workspace, "llm_head: null pointer");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && ws_bytes >= (int64_t)R * c.d.t_hidden * esz(c), "llm_head: bad R=%d or workspace too small", R);
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "llm_head: workspace must be 256-byte aligned");
    return llm_head(c, rows, add, R, out_logits, workspace, (hipStream_t)stream);
}

extern "C" int64_t devqa_llm_forward_workspace(devqa_ctx_t h, int R, int R_want) {
    Ctx* c = ctx_of(h);
    if (!c || R <= 0 || R_want <= 0) return -1;
    Arena a(nullptr, 0);
    LlmPlan p;
    plan_llm(*c, R, true, a, p);
    a.take((int64_t)R_want * c->d.t_hidden * 4);              // gathered rows
    a.take((int64_t)R_want * c->d.t_hidden * esz(*c));         // their LayerNorm output
    return al256(a.off);
}

extern "C" int devqa_llm_forward(devqa_ctx_t h, float* x, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                                 const int32_t* want_rows, int R_want, float* out_logits, void* workspace, int64_t ws_bytes, void* stream) {
    return devqa_llm_forward_ex(h, x, nullptr, seq_desc, n_seq, max_len, R, dense, want_rows, R_want, out_logits, workspace, ws_bytes, stream);
}

extern "C" int devqa_llm_forward_ex(devqa_ctx_t h, float* x, const int32_t* positions, const int32_t* seq_desc, int n_seq, int max_len, int R, int dense,
                                    const int32_t* want_rows, int R_want, float* out_logits, void* workspace, int64_t ws_bytes, void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(x && seq_desc && want_rows && out_logits && workspace, "llm_forward: null pointer");
    DEVQA_CHECK_ARG(!llama_dec(c) || positions, "llm_forward: the LLaMA decoder needs the rotary positions of the rows (devqa_llm_forward_ex)");
    if (R == 0 || R_want == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && R_want > 0 && n_seq > 0 && max_len > 0, "llm_forward: bad dims");
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "llm_forward: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    Arena a(workspace, ws_bytes);
    LlmPlan w;
    plan_llm(c, R, true, a, w);
    float* rows = (float*)a.take((int64_t)R_want * c.d.t_hidden * 4);
    void* hws = a.take((int64_t)R_want * c.d.t_hidden * esz(c));
    if (a.overflow) return devqa_fail(DEVQA_E_SHAPE, "llm_forward: workspace of %lld bytes is too small (need %lld)", (long long)ws_bytes, (long long)al256(a.off));
    RC(llm_layers(c, x, positions, seq_desc, n_seq, max_len, R, dense, 0, c.d.t_layers, 0, nullptr, w, st));
    RC(devqa_gather_rows(x, want_rows, R_want, c.d.t_hidden, 4, rows, st));
    return llm_head(c, rows, nullptr, R_want, out_logits, hws, st);
}

// =================================================================================================================================
// K9-K13: the FT_VL inner loop for E concurrent edits
// =================================================================================================================================
namespace {
__global__ void ft_init_kernel(const float* __restrict__ mask, int E, int Lmax, int max_steps, float* coef, int32_t* active, int32_t* do_update,
                               int32_t* n_steps, int32_t* adam_t, float* losses, int32_t* single) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float s = 0.f;
    for (int l = 0; l < Lmax; ++l) s += mask[e * Lmax + l];
    for (int l = 0; l < Lmax; ++l) coef[e * Lmax + l] = mask[e * Lmax + l] / s;     // mask / mask.sum (ft_vl.py:191-199)
    active[e] = 1; do_update[e] = 0; n_steps[e] = 0; adam_t[e] = 0;
    single[e] = (s == 1.f && mask[e * Lmax] == 1.f) ? 1 : 0;        // one loss row, in slot 0: the second moment factors too (devqa_ft_adamw_step_fm)
    for (int t = 0; t < max_steps; ++t) losses[e * max_steps + t] = 0.f;
}
// delta[e] = w[e] - w0 (shared or per edit) for edits that took at least one update, else 0 (their w was never written)
__global__ void ft_delta_kernel(const float* __restrict__ w, const float* __restrict__ w0, int64_t w0_stride_e, const int32_t* __restrict__ adam_t,
                                int64_t per, float* __restrict__ delta) {
    const int e = blockIdx.y;
    const bool upd = adam_t[e] > 0;
    const float* we = w + (int64_t)e * per;
    const float* w0e = w0 + (int64_t)e * w0_stride_e;
    float* de = delta + (int64_t)e * per;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < per; i += (int64_t)gridDim.x * blockDim.x * 4) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (upd) {
            const float4 a = *reinterpret_cast<const float4*>(we + i), b = *reinterpret_cast<const float4*>(w0e + i);
            r = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        }
        *reinterpret_cast<float4*>(de + i) = r;
    }
}

struct FtPlan {
    float *w, *var, *dstate, *y, *logits, *nll, *coef, *dH, *dy, *skws;
    void *h, *dlog;
    int32_t *active, *do_update, *single;
    int splits, g0, g1;
};
void longk_groups(const Ctx& c, int M, int N, int K, FtPlan& p) {    // the row grouping of lib.gemm_rows_longk (tools/splitk_rows_bench.py)
    p.splits = 0; p.g0 = M; p.g1 = 0;
    if (!c.bf16 || M > 256) return;
    const int nk = (K + 63) / 64, tn = (N + 127) / 128;
    int s = 640 / (tn > 0 ? tn : 1);
    if (s > nk) s = nk;
    if (s < 1) s = 1;
    p.splits = s;
    if (M > 128 && M <= 192) { p.g0 = 128; p.g1 = M - 128; }
}
int64_t plan_ft(const Ctx& c, int E, int kmax, int npad, Arena& a, FtPlan& p) {
    const int64_t Dout = c.d.t_hidden, V = c.d.t_vocab, R = (int64_t)E * kmax, e = esz(c);
    p.w = (float*)a.take((int64_t)E * Dout * npad * 4);
    p.var = (float*)a.take((int64_t)E * Dout * npad * 4);
    p.dstate = (float*)a.take((R + E) * Dout * 4);    // EMA of dy (+ one row per edit: EMA of dy[0]^2 of one-row edits): devqa_ft_adamw_step_fm
    p.y = (float*)a.take(R * Dout * 4);
    p.h = a.take(R * Dout * e);
    p.logits = (float*)a.take(R * V * 4);
    p.dlog = a.take(R * V * e);
    p.nll = (float*)a.take(R * 4);
    p.coef = (float*)a.take(R * 4);
    p.dH = (float*)a.take(R * Dout * 4);
    p.dy = (float*)a.take(R * Dout * 4);
    p.active = (int32_t*)a.take((int64_t)E * 4);
    p.do_update = (int32_t*)a.take((int64_t)E * 4);
    p.single = (int32_t*)a.take((int64_t)E * 4);
    longk_groups(c, (int)R, (int)Dout, (int)V, p);
    p.skws = p.splits ? (float*)a.take((int64_t)p.splits * (p.g0 > p.g1 ? p.g0 : p.g1) * Dout * 4) : nullptr;
    return al256(a.off);
}
}  // namespace

extern "C" int64_t devqa_ft_edit_workspace(devqa_ctx_t h, int E, int kmax, int npad) {
    Ctx* c = ctx_of(h);
    if (!c || E <= 0 || kmax <= 0 || npad <= 0) return -1;
    Arena a(nullptr, 0);
    FtPlan p;
    return plan_ft(*c, E, kmax, npad, a, p);
}

extern "C" int devqa_ft_edit(devqa_ctx_t h, const float* w0, int64_t w0_stride_e, const float* a_rows, const float* resid_rows,
                             const int32_t* labels, const float* mask, int E, int kmax, int npad, const devqa_ft_cfg* cfg, float* out_delta,
                             float* out_losses, int32_t* out_steps, int32_t* out_updates, void* workspace, int64_t ws_bytes, void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(w0 && a_rows && resid_rows && labels && mask && cfg && out_delta && out_losses && out_steps && out_updates && workspace,
                    "ft_edit: null pointer");
    if (E == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && kmax >= 1 && kmax <= DEVQA_FT_MAX_ROWS && npad > 0 && npad % 4 == 0 && cfg->num_steps >= 1, "ft_edit: bad dims E=%d kmax=%d npad=%d", E, kmax, npad);
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "ft_edit: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int Dout = c.d.t_hidden, V = c.d.t_vocab, R = E * kmax;
    Arena ar(workspace, ws_bytes);
    FtPlan p;
    plan_ft(c, E, kmax, npad, ar, p);
    if (ar.overflow) return devqa_fail(DEVQA_E_SHAPE, "ft_edit: workspace of %lld bytes is too small (need %lld)", (long long)ws_bytes, (long long)al256(ar.off));
    const Weight *fg = nullptr, *et = nullptr;
    RC(need(c, llama_dec(c) ? "language_model.model.norm.weight" : "language_model.model.decoder.final_layer_norm.weight", &fg, DEVQA_DTYPE_F32));
    if (c.bf16) RC(need(c, "derived.embed_T", &et, DEVQA_DTYPE_BF16));
    hipLaunchKernelGGL(ft_init_kernel, dim3((E + 63) / 64), dim3(64), 0, st, mask, E, kmax, cfg->num_steps, p.coef, p.active, p.do_update, out_steps,
                       out_updates, out_losses, p.single);
    static const bool one_row_form = !(getenv("DEVQA_FT_SINGLE") && atoi(getenv("DEVQA_FT_SINGLE")) == 0);     // A/B: 0 = every edit keeps its v matrix
    if (!one_row_form) (void)hipMemsetAsync(p.single, 0, (size_t)E * 4, st);
    DEVQA_LAUNCH_CHECK("ft_init");
    // step-0 fc2 rows with the pristine matrix (the active columns carry all of W.a)
    RC(devqa_rows_matvec_f32(w0, w0_stride_e, a_rows, nullptr, nullptr, p.y, E, kmax, Dout, npad, st));
    for (int it = 0; it < cfg->num_steps; ++it) {
        RC(llm_head(c, p.y, resid_rows, R, p.logits, p.h, st));
        if (c.bf16) RC(devqa_vocab_rows(p.logits, V, R, V, labels, p.coef, nullptr, p.nll, (devqa_bf16*)p.dlog, V, st));
        else RC(devqa_vocab_rows_f32(p.logits, V, R, V, labels, p.coef, nullptr, p.nll, (float*)p.dlog, V, st));
        RC(devqa_ft_step_control(p.nll, mask, E, kmax, it, cfg->num_steps, cfg->loss_floor, p.active, p.do_update, out_steps, out_updates, out_losses, st));
        // dH = dlogits . E   (split-K over the vocabulary on the bf16 MFMA kernel; exact-fp32 GEMM in faithful mode)
        if (p.splits) {
            const devqa_bf16* dl = (const devqa_bf16*)p.dlog;
            RC(devqa_gemm_bf16_splitk(dl, V, (const devqa_bf16*)et->ptr, V, p.g0, Dout, V, p.splits, p.skws, p.dH, st));
            if (p.g1) RC(devqa_gemm_bf16_splitk(dl + (int64_t)p.g0 * V, V, (const devqa_bf16*)et->ptr, V, p.g1, Dout, V, p.splits, p.skws,
                                                p.dH + (int64_t)p.g0 * Dout, st));
        } else {
            RC(gemm(c, p.dlog, V, "derived.embed_T", nullptr, R, Dout, V, 1.f, DEVQA_ACT_NONE, nullptr, nullptr, p.dH, st));
        }
        if (llama_dec(c)) RC(devqa_rmsnorm_bwd_dx(p.y, resid_rows, (const float*)fg->ptr, p.dH, R, Dout, c.d.t_rms_eps, p.dy, st));
        else RC(devqa_layernorm_bwd_dx(p.y, resid_rows, (const float*)fg->ptr, p.dH, R, Dout, c.d.t_ln_eps, p.dy, st));
        // (a_rows is the same at every step: the first moment is kept as its rank-kmax factors, no [Dout, npad] matrix; DEVQA_FT_FACTORED=0: the
        // form with the matrix, in out_delta's storage -- the delta is written there only after the loop)
        static const bool factored = !(getenv("DEVQA_FT_FACTORED") && atoi(getenv("DEVQA_FT_FACTORED")) == 0);
        if (factored)
            RC(devqa_ft_adamw_step_fm(p.w, p.dstate, p.var, w0, a_rows, p.dy, p.y, p.do_update, out_updates, p.single, E, kmax, Dout, npad, cfg->lr, cfg->beta1,
                                      cfg->beta2, cfg->eps, cfg->weight_decay, cfg->clamp_eps, w0_stride_e, st));
        else
            RC(devqa_ft_adamw_step(p.w, out_delta, p.var, w0, a_rows, p.dy, p.y, p.do_update, out_updates, E, kmax, Dout, npad, cfg->lr, cfg->beta1,
                                   cfg->beta2, cfg->eps, cfg->weight_decay, cfg->clamp_eps, w0_stride_e, st));
    }
    const int64_t per = (int64_t)Dout * npad;
    hipLaunchKernelGGL(ft_delta_kernel, dim3((unsigned)((per / 4 + 255) / 256 < 1024 ? (per / 4 + 255) / 256 : 1024), E), dim3(256), 0, st, p.w, w0,
                       w0_stride_e, out_updates, per, out_delta);
    DEVQA_LAUNCH_CHECK("ft_delta");
    return DEVQA_OK;
}

// =================================================================================================================================
// K13 on the context's edited matrix, K14
// =================================================================================================================================
extern "C" int devqa_ctx_bind_edit_target(devqa_ctx_t h, const char* name, void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(name, "bind_edit_target: null name");
    const Weight* w = find(c, name);
    if (!w) return devqa_fail(DEVQA_E_STATE, "bind_edit_target: weight table has no entry '%s'", name);
    if (w->dtype != DEVQA_DTYPE_F32 || w->ndim != 2) return devqa_fail(DEVQA_E_STATE, "bind_edit_target: '%s' must be an fp32 master matrix", name);
    const int64_t n = w->shape[0] * w->shape[1];
    if (c.edit_w0 && c.edit_numel != n) { (void)hipFree(c.edit_w0); c.edit_w0 = nullptr; }
    if (!c.edit_w0 && hipMalloc((void**)&c.edit_w0, (size_t)n * 4) != hipSuccess) {
        (void)hipGetLastError();
        return devqa_fail(DEVQA_E_OOM, "bind_edit_target: cannot allocate the pristine copy (%lld bytes)", (long long)n * 4);
    }
    c.edit_numel = n;
    c.edit_name = name;
    if (hipMemcpyAsync(c.edit_w0, w->ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return devqa_fail(DEVQA_E_HIP, "bind_edit_target: copy failed");
    return DEVQA_OK;
}

namespace {
int refresh_shadow(Ctx& c, hipStream_t st) {   // GEMMs read the bf16 shadow "<name>#shadow" of an fp32 master when the table has one
    const Weight* sh = find(c, c.edit_name + "#shadow");
    const Weight* w = find(c, c.edit_name);
    if (sh && c.bf16) return devqa_cast_f32_bf16((const float*)w->ptr, (devqa_bf16*)sh->ptr, c.edit_numel, st);
    return DEVQA_OK;
}
}  // namespace

extern "C" int devqa_apply_delta(devqa_ctx_t h, const float* delta, void* stream) {
    CTX_OR_FAIL(h);
    DEVQA_CHECK_ARG(delta, "apply_delta: null pointer");
    if (!c.edit_w0) return devqa_fail(DEVQA_E_STATE, "apply_delta: no edit target bound (devqa_ctx_bind_edit_target)");
    const Weight* w = find(c, c.edit_name);
    RC(devqa_delta_op(1, (float*)w->ptr, nullptr, const_cast<float*>(delta), c.edit_numel, stream));    // w += delta (ft_vl.py:60-61)
    return refresh_shadow(c, (hipStream_t)stream);
}

extern "C" int devqa_restore(devqa_ctx_t h, void* stream) {
    CTX_OR_FAIL(h);
    if (!c.edit_w0) return devqa_fail(DEVQA_E_STATE, "restore: no edit target bound (devqa_ctx_bind_edit_target)");
    const Weight* w = find(c, c.edit_name);
    RC(devqa_delta_op(2, (float*)w->ptr, c.edit_w0, nullptr, c.edit_numel, stream));                     // w = w0 (ft_vl.py:44-45)
    return refresh_shadow(c, (hipStream_t)stream);
}

namespace {
__global__ void token_acc_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ labels, const float* __restrict__ mask, int R,
                                 float* __restrict__ acc) {
    float hit = 0.f, tot = 0.f;
    for (int i = threadIdx.x; i < R; i += 64) {
        hit += (pred[i] == labels[i] ? 1.f : 0.f) * mask[i];
        tot += mask[i];
    }
    hit = wave_sum(hit);
    tot = wave_sum(tot);
    if (threadIdx.x == 0) acc[0] = hit / tot;
}
}  // namespace

extern "C" int devqa_token_acc(const float* logits_rows, int64_t ldl, int R, int V, const int32_t* labels, const float* mask, float* out_acc,
                               int32_t* out_pred, void* stream) {
    DEVQA_CHECK_ARG(logits_rows && labels && mask && out_acc && out_pred, "token_acc: null pointer");
    DEVQA_CHECK_SHAPE(R > 0 && V > 0, "token_acc: bad dims");
    RC(devqa_vocab_rows(logits_rows, ldl, R, V, nullptr, nullptr, out_pred, nullptr, nullptr, V, stream));   // argmax(softmax(.)) == argmax(.)
    hipLaunchKernelGGL(token_acc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_pred, labels, mask, R, out_acc);
    DEVQA_LAUNCH_CHECK("token_acc");
    return DEVQA_OK;
}

// =================================================================================================================================
// the single collective: RCCL all-gather of [n, 16] fp32 score rows.  librccl is opened at run time (the same
// library object a host runtime may already have loaded), so libdevqa_hip.so has no link-time dependency on it.
// =================================================================================================================================
namespace {
typedef struct { char internal[128]; } rccl_uid;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_uid*) = nullptr;
    int (*CommInitRank)(void**, int, rccl_uid, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return DEVQA_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* l = nullptr;
    for (const char* n : names)
        if ((l = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!l) return devqa_fail(DEVQA_E_STATE, "comm: cannot open librccl (%s)", dlerror());
    g_rccl.GetUniqueId = (int (*)(rccl_uid*))dlsym(l, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, rccl_uid, int))dlsym(l, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(l, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(l, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(l, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return devqa_fail(DEVQA_E_STATE, "comm: librccl lacks an expected symbol");
    g_rccl.lib = l;
    return DEVQA_OK;
}
struct Comm { void* nccl; int rank, world; };
// live handles: a stale / foreign handle (or a call before librccl was ever loaded) is an error, not a dereference
std::mutex g_comm_mu;
std::set<Comm*> g_comm_live;
Comm* comm_of(devqa_comm_t h) {
    Comm* cm = reinterpret_cast<Comm*>(static_cast<uintptr_t>(h));
    std::lock_guard<std::mutex> lock(g_comm_mu);
    return (g_rccl.lib && g_comm_live.count(cm)) ? cm : nullptr;
}
int rccl_fail(const char* what, int rc) {
    return devqa_fail(DEVQA_E_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error");
}
}  // namespace

extern "C" int devqa_comm_unique_id(void* id128) {
    DEVQA_CHECK_ARG(id128, "comm_unique_id: null pointer");
    RC(rccl_load());
    rccl_uid u;
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, &u, sizeof(u));
    return DEVQA_OK;
}

extern "C" int devqa_comm_create(int rank, int world, const void* id128, int device, devqa_comm_t* out) {
    DEVQA_CHECK_ARG(id128 && out && world >= 1 && rank >= 0 && rank < world, "comm_create: bad argument");
    RC(rccl_load());
    if (hipSetDevice(device) != hipSuccess) return devqa_fail(DEVQA_E_ARG, "comm_create: no device %d", device);
    rccl_uid u;
    memcpy(&u, id128, sizeof(u));
    Comm* cm = new (std::nothrow) Comm{nullptr, rank, world};
    if (!cm) return devqa_fail(DEVQA_E_OOM, "comm_create: host allocation failed");
    const int rc = g_rccl.CommInitRank(&cm->nccl, world, u, rank);
    if (rc != 0) { delete cm; return rccl_fail("ncclCommInitRank", rc); }
    {
        std::lock_guard<std::mutex> lock(g_comm_mu);
        g_comm_live.insert(cm);
    }
    *out = (devqa_comm_t)(uintptr_t)cm;
    return DEVQA_OK;
}

extern "C" int devqa_comm_destroy(devqa_comm_t h) {
    Comm* cm = comm_of(h);
    if (!cm) return devqa_fail(DEVQA_E_STATE, "comm_destroy: invalid or destroyed communicator handle");
    {
        std::lock_guard<std::mutex> lock(g_comm_mu);
        g_comm_live.erase(cm);
    }
    const int rc = g_rccl.CommDestroy(cm->nccl);
    delete cm;
    return rc == 0 ? DEVQA_OK : rccl_fail("ncclCommDestroy", rc);
}

extern "C" int devqa_gather_scores(devqa_comm_t h, const float* local, int n_rows, float* out, void* stream) {
    Comm* cm = comm_of(h);
    if (!cm) return devqa_fail(DEVQA_E_STATE, "gather_scores: invalid or destroyed communicator handle");
    DEVQA_CHECK_ARG(local && out && n_rows > 0, "gather_scores: bad argument");
    const int rc = g_rccl.AllGather(local, out, (size_t)n_rows * DEVQA_SCORE_COLS, /*ncclFloat32*/ 7, cm->nccl, (hipStream_t)stream);
    return rc == 0 ? DEVQA_OK : rccl_fail("ncclAllGather", rc);
}


// ---- K16 / K17: MEND_VL's hyper-network transform and low-rank delta application as two calls ------------------------------------
// K16 (R/editor/vllm_editors/mend_vl/auxiliary_networks.py:112-151, 62-83, inference mode): rows idx[0..n) of (x [R, du], delta [R, dv])
// -> normalise with the stored statistics and concatenate [n, D = du + dv] -> n_layers low-rank residual layers
// out = x + relu((x v^T) u^T + bias) * mode_scale + mode_shift (exact-fp32 GEMMs; DEVQA_MEND_SPLIT_BF16: three bf16 products of split operands, what
// the bf16 compute mode asks for -- 1284 -> ~450 us per GEMM pair of the BLIP-2 hyper-network) -> split into (x~ [n, du], delta~ [n, dv]).
extern "C" int64_t devqa_mend_transform_workspace(int n, int du, int dv, int rank) {
    if (n <= 0 || du <= 0 || dv <= 0 || rank <= 0) return 256;
    const int64_t D = (int64_t)du + dv;
    // + the split-bf16 form's operands: (hi, lo) of the activations [n, D], of the low-rank product [n, rank] and of one weight [rank, D]
    return 3 * al256((int64_t)n * D * 4) + al256((int64_t)n * rank * 4) + 2 * al256((int64_t)n * D * 2) + 2 * al256((int64_t)n * rank * 2) +
           2 * al256((int64_t)rank * D * 2);
}

namespace {
// C [M, N] fp32 = A . W^T on the bf16 MFMA from split operands: A_hi.W_lo + A_lo.W_hi + A_hi.W_hi (small terms first), fp32 accumulation
int gemm_split3(const devqa_bf16* a_hi, const devqa_bf16* a_lo, const devqa_bf16* w_hi, const devqa_bf16* w_lo, int M, int N, int K, float* out,
                void* stream) {
    int rc = devqa_gemm_bf16(a_hi, K, w_lo, K, nullptr, M, N, K, 1.f, DEVQA_ACT_NONE, nullptr, nullptr, out, N, stream);
    if (rc != DEVQA_OK) return rc;
    rc = devqa_gemm_bf16(a_lo, K, w_hi, K, nullptr, M, N, K, 1.f, DEVQA_ACT_NONE, out, nullptr, out, N, stream);
    if (rc != DEVQA_OK) return rc;
    return devqa_gemm_bf16(a_hi, K, w_hi, K, nullptr, M, N, K, 1.f, DEVQA_ACT_NONE, out, nullptr, out, N, stream);
}
}  // namespace

extern "C" int devqa_mend_transform(const float* x, const float* delta, const int32_t* idx, int n, int du, int dv,
                                    const devqa_mend_net* net, float* out_x, float* out_d, void* workspace, int64_t ws_bytes, void* stream) {
    DEVQA_CHECK_ARG(x && delta && net && out_x && out_d && workspace, "mend_transform: null pointer");
    if (n == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(n > 0 && du > 0 && dv > 0 && net->rank > 0 && net->n_layers >= 1 && net->n_layers <= DEVQA_MEND_MAX_LAYERS,
                      "mend_transform: bad dims n=%d du=%d dv=%d rank=%d layers=%d", n, du, dv, net->rank, net->n_layers);
    DEVQA_CHECK_SHAPE((du + dv) % 4 == 0 && net->rank % 4 == 0, "mend_transform: D=%d and rank=%d must be multiples of 4 (exact-fp32 GEMM)", du + dv, net->rank);
    DEVQA_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "mend_transform: workspace must be 256-byte aligned");
    DEVQA_CHECK_SHAPE(ws_bytes >= devqa_mend_transform_workspace(n, du, dv, net->rank), "mend_transform: workspace too small");
    const int D = du + dv, rank = net->rank;
    char* ws = (char*)workspace;
    float* a = (float*)ws;
    float* b = (float*)(ws + al256((int64_t)n * D * 4));
    float* pre = (float*)(ws + 2 * al256((int64_t)n * D * 4));
    float* low = (float*)(ws + 3 * al256((int64_t)n * D * 4));
    char* sp = ws + 3 * al256((int64_t)n * D * 4) + al256((int64_t)n * rank * 4);
    devqa_bf16* a_hi = (devqa_bf16*)sp;
    devqa_bf16* a_lo = (devqa_bf16*)(sp + al256((int64_t)n * D * 2));
    devqa_bf16* l_hi = (devqa_bf16*)(sp + 2 * al256((int64_t)n * D * 2));
    devqa_bf16* l_lo = (devqa_bf16*)(sp + 2 * al256((int64_t)n * D * 2) + al256((int64_t)n * rank * 2));
    devqa_bf16* w_hi = (devqa_bf16*)(sp + 2 * al256((int64_t)n * D * 2) + 2 * al256((int64_t)n * rank * 2));
    devqa_bf16* w_lo = (devqa_bf16*)((char*)w_hi + al256((int64_t)rank * D * 2));
    const bool split = (net->flags & DEVQA_MEND_SPLIT_BF16) && D % 8 == 0 && rank % 8 == 0;
    int rc = devqa_mend_normalize_concat(x, delta, idx, net->u_mean, net->u_std, net->v_mean, net->v_std, 1e-7f, n, du, dv, a, stream);
    if (rc != DEVQA_OK) return rc;
    float* cur = a;
    float* nxt = b;
    for (int l = 0; l < net->n_layers; ++l) {
        const devqa_mend_layer& L = net->layers[l];
        DEVQA_CHECK_ARG(L.u && L.v && L.bias && L.mode_scale && L.mode_shift, "mend_transform: layer %d has a null pointer", l);
        if (split) {
            rc = devqa_split_f32_bf16x2(cur, a_hi, a_lo, (int64_t)n * D, stream);
            if (rc == DEVQA_OK) rc = devqa_split_f32_bf16x2(L.v, w_hi, w_lo, (int64_t)rank * D, stream);
            if (rc == DEVQA_OK) rc = gemm_split3(a_hi, a_lo, w_hi, w_lo, n, rank, D, low, stream);                                   // x v^T
            if (rc == DEVQA_OK) rc = devqa_split_f32_bf16x2(low, l_hi, l_lo, (int64_t)n * rank, stream);
            if (rc == DEVQA_OK) rc = devqa_split_f32_bf16x2(L.u, w_hi, w_lo, (int64_t)rank * D, stream);
            if (rc == DEVQA_OK) rc = gemm_split3(l_hi, l_lo, w_hi, w_lo, n, D, rank, pre, stream);                                   // . u^T
            if (rc != DEVQA_OK) return rc;
        } else {
            rc = devqa_gemm_f32(cur, D, L.v, D, nullptr, n, rank, D, 1.f, DEVQA_ACT_NONE, nullptr, low, rank, stream);      // x v^T
            if (rc != DEVQA_OK) return rc;
            rc = devqa_gemm_f32(low, rank, L.u, rank, nullptr, n, D, rank, 1.f, DEVQA_ACT_NONE, nullptr, pre, D, stream);  // . u^T
            if (rc != DEVQA_OK) return rc;
        }
        rc = devqa_mend_lrlinear_epilogue(pre, L.bias, L.mode_scale, L.mode_shift, cur, nxt, n, D, stream);
        if (rc != DEVQA_OK) return rc;
        float* t = cur; cur = nxt; nxt = t;
    }
    // split [n, D] -> [n, du], [n, dv]
    hipError_t e = hipMemcpy2DAsync(out_x, (size_t)du * 4, cur, (size_t)D * 4, (size_t)du * 4, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(out_d, (size_t)dv * 4, cur + du, (size_t)D * 4, (size_t)dv * 4, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return devqa_fail(DEVQA_E_HIP, "mend_transform: split copy: %s", hipGetErrorString(e));
    return DEVQA_OK;
}

// K17 (mend_vl.py:73-80, the forward hook `output + input @ delta_W` with delta_W = x~^T delta~ * lr / n kept in FACTORS):
// y [R, dout] fp32 += (h [R, din] . xt^T [din, npad]) . dt [npad, dout], operands in the compute dtype (bf16 or fp32), npad % 64 == 0;
// dtT = dt transposed [dout, npad] (the TN GEMM operand).
extern "C" int64_t devqa_mend_apply_workspace(int R, int npad, int compute_dtype) {
    if (R <= 0 || npad <= 0) return 256;
    return al256((int64_t)R * npad * (compute_dtype == DEVQA_DTYPE_F32 ? 4 : 2));
}

extern "C" int devqa_mend_apply(const void* h, const void* xt, const void* dtT, float* y, int R, int din, int dout, int npad, int compute_dtype,
                                void* workspace, int64_t ws_bytes, void* stream) {
    DEVQA_CHECK_ARG(h && xt && dtT && y && workspace, "mend_apply: null pointer");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && din > 0 && dout > 0 && npad > 0 && npad % 64 == 0, "mend_apply: bad dims R=%d din=%d dout=%d npad=%d", R, din, dout, npad);
    DEVQA_CHECK_ARG(compute_dtype == DEVQA_DTYPE_BF16 || compute_dtype == DEVQA_DTYPE_F32, "mend_apply: unknown dtype %d", compute_dtype);
    DEVQA_CHECK_SHAPE(ws_bytes >= devqa_mend_apply_workspace(R, npad, compute_dtype), "mend_apply: workspace too small");
    if (compute_dtype == DEVQA_DTYPE_F32) {
        float* coef = (float*)workspace;
        int rc = devqa_gemm_f32((const float*)h, din, (const float*)xt, din, nullptr, R, npad, din, 1.f, DEVQA_ACT_NONE, nullptr, coef, npad, stream);
        if (rc != DEVQA_OK) return rc;
        return devqa_gemm_f32(coef, npad, (const float*)dtT, npad, nullptr, R, dout, npad, 1.f, DEVQA_ACT_NONE, y, y, dout, stream);
    }
    devqa_bf16* coef = (devqa_bf16*)workspace;
    int rc = devqa_gemm_bf16((const devqa_bf16*)h, din, (const devqa_bf16*)xt, din, nullptr, R, npad, din, 1.f, DEVQA_ACT_NONE, nullptr, coef, nullptr,
                             npad, stream);
    if (rc != DEVQA_OK) return rc;
    return devqa_gemm_bf16(coef, npad, (const devqa_bf16*)dtT, npad, nullptr, R, dout, npad, 1.f, DEVQA_ACT_NONE, y, nullptr, y, dout, stream);
}
