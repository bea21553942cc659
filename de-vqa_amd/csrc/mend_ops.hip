// Small row-wise kernels of the MEND_VL edit path (R/editor/vllm_editors/mend_vl): ReLU backward for the hooked fc1
// gradient, the GradientTransform input normalisation + concat (auxiliary_networks.py:138-148), the LRLinear epilogue
// (auxiliary_networks.py:70-83) and logit_KL_loss rows (mend_vl.py:355-366).  All HBM-bound, one pass.
#include "common.h"

template <typename T> __device__ __forceinline__ float mo_ld(const T* p);
template <> __device__ __forceinline__ float mo_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float mo_ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void mo_st(T* p, float v);
template <> __device__ __forceinline__ void mo_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void mo_st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// grad_in = grad_out where act_out > 0 else 0  (act_out = relu(pre): act_out > 0 <=> pre > 0)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ act_out, const T* __restrict__ grad_out, T* __restrict__ grad_in, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mo_st<T>(grad_in + i, mo_ld<T>(act_out + i) > 0.f ? mo_ld<T>(grad_out + i) : 0.f);
}

// GELU (erf form, HF "gelu": x * 0.5 * (1 + erf(x / sqrt 2))) on fp32 pre-activations kept for a backward pass, and its derivative
// Phi(x) + x * phi(x): the Q-Former / ViT FFN activation when FT_VL trains Q-Former parameters (ft_vl.py:_execute_ft_general)
__global__ void gelu_fwd_kernel(const float* __restrict__ x, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float y = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (out_bf16) out_bf16[i] = f32_to_bf16(y);
        if (out_f32) out_f32[i] = y;
    }
}

__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ grad_out, float* __restrict__ grad_in, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
        grad_in[i] = grad_out[i] * (cdf + v * pdf);
    }
}

// out[r, 0:du] = (u[idx[r], :] - u_mean) / (u_std + eps);  out[r, du:du+dv] = (v[idx[r], :] - v_mean) / (v_std + eps)
// (mean/std NULL: plain gather-concat, cfg.norm == False)
__global__ void mend_normalize_concat_kernel(const float* __restrict__ u, const float* __restrict__ v, const int32_t* __restrict__ idx,
                                             const float* __restrict__ u_mean, const float* __restrict__ u_std,
                                             const float* __restrict__ v_mean, const float* __restrict__ v_std, float eps, int du,
                                             int dv, float* __restrict__ out) {
    const int r = blockIdx.x;
    const int64_t src = idx ? idx[r] : r;
    const int D = du + dv;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float x;
        if (c < du) {
            x = u[src * du + c];
            if (u_mean) x = (x - u_mean[c]) / (u_std[c] + eps);
        } else {
            const int cc = c - du;
            x = v[src * dv + cc];
            if (v_mean) x = (x - v_mean[cc]) / (v_std[cc] + eps);
        }
        out[(int64_t)r * D + c] = x;
    }
}

// out = max((pre + bias) * scale + shift, 0) + x      (LRLinear with init == 'id'; scale/shift = the mode's rows)
__global__ void mend_lrlinear_epilogue_kernel(const float* __restrict__ pre, const float* __restrict__ bias, const float* __restrict__ scale,
                                              const float* __restrict__ shift, const float* __restrict__ x, float* __restrict__ out,
                                              int64_t n, int D) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        float p = pre[i] + (bias ? bias[c] : 0.f);
        if (scale) p = p * scale[c] + shift[c];
        out[i] = fmaxf(p, 0.f) + x[i];
    }
}

// kl[r] = sum_v softmax(l1[r])_v * (log_softmax(l1[r])_v - log_softmax(l2[r])_v)   (one workgroup per row)
__global__ __launch_bounds__(256) void logit_kl_rows_kernel(const float* __restrict__ l1, const float* __restrict__ l2, int64_t ld1,
                                                            int64_t ld2, int V, float* __restrict__ kl) {
    __shared__ float red[8];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* a = l1 + (int64_t)r * ld1;
    const float* b = l2 + (int64_t)r * ld2;
    auto block_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float ma = -INFINITY, mb = -INFINITY;
    for (int c = tid; c < V; c += 256) { ma = fmaxf(ma, a[c]); mb = fmaxf(mb, b[c]); }
    ma = block_max(ma);
    mb = block_max(mb);
    float sa = 0.f, sb = 0.f;
    for (int c = tid; c < V; c += 256) { sa += expf(a[c] - ma); sb += expf(b[c] - mb); }
    sa = block_sum(sa);
    sb = block_sum(sb);
    const float lza = ma + logf(sa), lzb = mb + logf(sb);
    float acc = 0.f;
    for (int c = tid; c < V; c += 256) {
        const float lp1 = a[c] - lza, lp2 = b[c] - lzb;
        acc += expf(lp1) * (lp1 - lp2);
    }
    acc = block_sum(acc);
    if (tid == 0) kl[r] = acc;
}

static inline unsigned mo_grid(int64_t n) { return (unsigned)((n + 255) / 256 < 65535 * 4 ? (n + 255) / 256 : 65535 * 4); }

extern "C" int devqa_relu_bwd(const devqa_bf16* act_out, const devqa_bf16* grad_out, devqa_bf16* grad_in, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(act_out && grad_out && grad_in && n > 0, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, act_out, grad_out, grad_in, n);
    DEVQA_LAUNCH_CHECK("relu_bwd");
    return DEVQA_OK;
}

extern "C" int devqa_gelu_f32(const float* x, devqa_bf16* out_bf16, float* out_f32, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(x && (out_bf16 || out_f32) && n > 0, "gelu_f32: bad arguments");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out_bf16, out_f32, n);
    DEVQA_LAUNCH_CHECK("gelu_f32");
    return DEVQA_OK;
}

extern "C" int devqa_gelu_bwd_f32(const float* x, const float* grad_out, float* grad_in, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(x && grad_out && grad_in && n > 0, "gelu_bwd_f32: bad arguments");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, x, grad_out, grad_in, n);
    DEVQA_LAUNCH_CHECK("gelu_bwd_f32");
    return DEVQA_OK;
}

extern "C" int devqa_relu_bwd_f32(const float* act_out, const float* grad_out, float* grad_in, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(act_out && grad_out && grad_in && n > 0, "relu_bwd_f32: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, act_out, grad_out, grad_in, n);
    DEVQA_LAUNCH_CHECK("relu_bwd_f32");
    return DEVQA_OK;
}

extern "C" int devqa_mend_normalize_concat(const float* u, const float* v, const int32_t* idx, const float* u_mean, const float* u_std,
                                           const float* v_mean, const float* v_std, float eps, int n_rows, int du, int dv, float* out,
                                           void* stream) {
    DEVQA_CHECK_ARG(u && v && out && n_rows > 0 && du > 0 && dv > 0, "mend_normalize_concat: bad arguments");
    DEVQA_CHECK_ARG((u_mean == nullptr) == (u_std == nullptr) && (v_mean == nullptr) == (v_std == nullptr),
                    "mend_normalize_concat: mean and std must be given together");
    hipLaunchKernelGGL(mend_normalize_concat_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, u, v, idx, u_mean, u_std, v_mean,
                       v_std, eps, du, dv, out);
    DEVQA_LAUNCH_CHECK("mend_normalize_concat");
    return DEVQA_OK;
}

extern "C" int devqa_mend_lrlinear_epilogue(const float* pre, const float* bias, const float* scale, const float* shift, const float* x,
                                            float* out, int n_rows, int D, void* stream) {
    DEVQA_CHECK_ARG(pre && x && out && n_rows > 0 && D > 0, "mend_lrlinear_epilogue: bad arguments");
    DEVQA_CHECK_ARG((scale == nullptr) == (shift == nullptr), "mend_lrlinear_epilogue: scale and shift must be given together");
    const int64_t n = (int64_t)n_rows * D;
    hipLaunchKernelGGL(mend_lrlinear_epilogue_kernel, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, pre, bias, scale, shift, x, out,
                       n, D);
    DEVQA_LAUNCH_CHECK("mend_lrlinear_epilogue");
    return DEVQA_OK;
}

extern "C" int devqa_logit_kl_rows(const float* logits1, int64_t ld1, const float* logits2, int64_t ld2, int R, int V, float* kl,
                                   void* stream) {
    DEVQA_CHECK_ARG(logits1 && logits2 && kl && R > 0 && V > 0, "logit_kl_rows: bad arguments");
    hipLaunchKernelGGL(logit_kl_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits1, logits2, ld1, ld2, V, kl);
    DEVQA_LAUNCH_CHECK("logit_kl_rows");
    return DEVQA_OK;
}

// ---- MEND_VL training (train_a_batch, R/editor/vllm_editors/mend_vl/mend_vl.py:301-341) ---------------------------------

// d/d(logits2) of coef[r] * KL(softmax(l1[r]) || softmax(l2[r])) = coef[r] * (softmax(l2[r]) - softmax(l1[r]));  kl[r] too
template <typename T>
__global__ __launch_bounds__(256) void kl_dlogits_kernel(const float* __restrict__ l1, const float* __restrict__ l2, int64_t ld1,
                                                         int64_t ld2, int V, const float* __restrict__ coef, float* __restrict__ kl,
                                                         T* __restrict__ dl2, int64_t ldd) {
    __shared__ float red[8];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* a = l1 + (int64_t)r * ld1;
    const float* b = l2 + (int64_t)r * ld2;
    auto block_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float ma = -INFINITY, mb = -INFINITY;
    for (int c = tid; c < V; c += 256) { ma = fmaxf(ma, a[c]); mb = fmaxf(mb, b[c]); }
    ma = block_max(ma);
    mb = block_max(mb);
    float sa = 0.f, sb = 0.f;
    for (int c = tid; c < V; c += 256) { sa += expf(a[c] - ma); sb += expf(b[c] - mb); }
    sa = block_sum(sa);
    sb = block_sum(sb);
    const float lza = ma + logf(sa), lzb = mb + logf(sb);
    const float cf = coef[r];
    float acc = 0.f;
    for (int c = tid; c < V; c += 256) {
        const float lp1 = a[c] - lza, lp2 = b[c] - lzb;
        const float p1 = expf(lp1);
        acc += p1 * (lp1 - lp2);
        mo_st<T>(dl2 + (int64_t)r * ldd + c, cf * (expf(lp2) - p1));
    }
    acc = block_sum(acc);
    if (tid == 0) kl[r] = acc;
}

// running mean / sum of squared deviations over rows, one row at a time (auxiliary_networks.py:88-91,122-136):
//   k += 1; new_m = m + (x - m) / k; s += (x - m) * (x - new_m); m = new_m        (reset: m = x_0, s = 0, k = 1)
// then std = sqrt(s / (k - 1)).  One thread per feature; k (a scalar) is updated by thread 0 after all features used it.
__global__ void welford_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, int n_rows, int D, int reset,
                                    float* __restrict__ mean, float* __restrict__ s, float* __restrict__ stdv,
                                    const float* __restrict__ k_in, float* __restrict__ k_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float kk = reset ? 0.f : k_in[0];
    float m = reset ? 0.f : mean[c], ss = reset ? 0.f : s[c];
    for (int r = 0; r < n_rows; ++r) {
        const float v = x[(int64_t)(idx ? idx[r] : r) * D + c];
        if (kk == 0.f) { m = v; ss = 0.f; kk = 1.f; continue; }
        kk += 1.f;
        const float nm = m + (v - m) / kk;
        ss += (v - m) * (v - nm);
        m = nm;
    }
    mean[c] = m;
    s[c] = ss;
    stdv[c] = sqrtf(ss / (kk - 1.f));
    if (c == 0) k_out[0] = kk;   // every thread computed the same kk; k_out != k_in (late blocks still read k_in)
}

// backward of out = max(z, 0) + x, z = (pre + bias) * scale + shift for one mode row:
//   dz = dout * (z >= 0);  dpre = dz * scale;  g_scale += sum_r dz * (pre + bias);  g_shift += sum_r dz;  g_bias += sum_r dpre
__global__ void mend_lrlinear_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ bias, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const float* __restrict__ dout, int n_rows, int D,
                                         float* __restrict__ dpre, float* __restrict__ g_scale, float* __restrict__ g_shift,
                                         float* __restrict__ g_bias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    const float b = bias[c], sc = scale[c], sh = shift[c];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int r = 0; r < n_rows; ++r) {
        const int64_t i = (int64_t)r * D + c;
        const float pb = pre[i] + b;
        const float z = pb * sc + sh;
        const float dz = z >= 0.f ? dout[i] : 0.f;   // clamp(min=0): the gradient passes AT 0 (auxiliary_networks.py:78-79)
        const float dp = dz * sc;
        dpre[i] = dp;
        a0 += dz * pb;
        a1 += dz;
        a2 += dp;
    }
    g_scale[c] += a0;
    g_shift[c] += a1;
    g_bias[c] += a2;
}

// sum of squares -> out[0] += sum x^2 (fp32 atomics; caller zeroes out)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.Adam (no weight decay, amsgrad off): g = grad * grad_scale[0]; m, v moments; step t (1-based)
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                 int64_t n, float lr, float b1, float b2, float eps, int t, const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    const float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = grad[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    }
}

extern "C" int devqa_kl_dlogits(const float* logits1, int64_t ld1, const float* logits2, int64_t ld2, int R, int V, const float* coef,
                                float* kl, void* dlogits2, int64_t ldd, int dlogits_bf16, void* stream) {
    DEVQA_CHECK_ARG(logits1 && logits2 && coef && kl && dlogits2 && R > 0 && V > 0, "kl_dlogits: bad arguments");
    if (dlogits_bf16)
        hipLaunchKernelGGL(kl_dlogits_kernel<bf16_t>, dim3(R), dim3(256), 0, (hipStream_t)stream, logits1, logits2, ld1, ld2, V, coef, kl,
                           (bf16_t*)dlogits2, ldd);
    else
        hipLaunchKernelGGL(kl_dlogits_kernel<float>, dim3(R), dim3(256), 0, (hipStream_t)stream, logits1, logits2, ld1, ld2, V, coef, kl,
                           (float*)dlogits2, ldd);
    DEVQA_LAUNCH_CHECK("kl_dlogits");
    return DEVQA_OK;
}

extern "C" int devqa_welford_rows(const float* x, const int32_t* idx, int n_rows, int D, int reset, float* mean, float* s, float* stdv,
                                  const float* k_in, float* k_out, void* stream) {
    DEVQA_CHECK_ARG(x && mean && s && stdv && k_in && k_out && k_in != k_out && n_rows >= 0 && D > 0, "welford_rows: bad arguments");
    hipLaunchKernelGGL(welford_rows_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, idx, n_rows, D, reset, mean, s,
                       stdv, k_in, k_out);
    DEVQA_LAUNCH_CHECK("welford_rows");
    return DEVQA_OK;
}

extern "C" int devqa_mend_lrlinear_bwd(const float* pre, const float* bias, const float* scale, const float* shift, const float* dout,
                                       int n_rows, int D, float* dpre, float* g_scale, float* g_shift, float* g_bias, void* stream) {
    DEVQA_CHECK_ARG(pre && bias && scale && shift && dout && dpre && g_scale && g_shift && g_bias && n_rows > 0 && D > 0,
                    "mend_lrlinear_bwd: bad arguments");
    hipLaunchKernelGGL(mend_lrlinear_bwd_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, pre, bias, scale, shift, dout,
                       n_rows, D, dpre, g_scale, g_shift, g_bias);
    DEVQA_LAUNCH_CHECK("mend_lrlinear_bwd");
    return DEVQA_OK;
}

extern "C" int devqa_sumsq_f32(const float* x, int64_t n, float* out, void* stream) {
    DEVQA_CHECK_ARG(x && out && n > 0, "sumsq: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, dim3(mo_grid(n) > 1024 ? 1024 : mo_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    DEVQA_LAUNCH_CHECK("sumsq");
    return DEVQA_OK;
}

extern "C" int devqa_adam_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               int step, const float* grad_scale, void* stream) {
    DEVQA_CHECK_ARG(p && grad && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    hipLaunchKernelGGL(adam_step_kernel, dim3(mo_grid(n)), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, n, lr, beta1, beta2, eps, step,
                       grad_scale);
    DEVQA_LAUNCH_CHECK("adam_step");
    return DEVQA_OK;
}

// ---- TP_VL (T-Patcher) patch-neuron step (R/editor/vllm_editors/tp_vl/tp_vl.py:155-192) ------------------------------------
// One new FFN neuron (key k [d], bias b, value v [d_out]) on frozen activations.  forward: pre[t] = h[t].k + b for the rows
// of the "edit-role" sequence, y[r] = ybase[r] + relu(pre[lab[r]]) * v for its label rows.  backward (one workgroup):
//   loss_a = mean_t exp(-pre[t]);  loss_m = mean_t exp(pm[t] * (pm[t] > 0)),  pm = hm.k + b  (memory text)
//   dpre[t]  = -la * exp(-pre[t]) / T   (+ (dy[r].v) * (pre[t] > 0) on label rows);   dpm[t] = lm * exp(pm[t]) * (pm[t] > 0) / Tm
//   gk = h^T dpre + hm^T dpm + wd k;  gb = sum dpre + sum dpm + wd b;  gv = sum_r relu(pre[lab[r]]) dy[r] + wd v
__global__ __launch_bounds__(256) void tp_neuron_fwd_kernel(const float* __restrict__ h, int T, int d, const float* __restrict__ k,
                                                            const float* __restrict__ b, const int32_t* __restrict__ lab, int L,
                                                            const float* __restrict__ v, const float* __restrict__ ybase, int d_out,
                                                            float* __restrict__ pre, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += 4) {
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc += h[(int64_t)t * d + c] * k[c];
        acc = wave_sum(acc);
        if (lane == 0) pre[t] = acc + b[0];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * d_out; i += 256) {
        const int r = i / d_out, c = i - r * d_out;
        y[i] = ybase[i] + fmaxf(pre[lab[r]], 0.f) * v[c];
    }
}

__global__ __launch_bounds__(256) void tp_neuron_bwd_kernel(const float* __restrict__ h, const float* __restrict__ pre, int T, int d,
                                                            const int32_t* __restrict__ lab, int L, const float* __restrict__ dy,
                                                            int d_out, const float* __restrict__ hm, int Tm, const float* __restrict__ k,
                                                            const float* __restrict__ b, const float* __restrict__ v, float la, float lm,
                                                            float wd, float* __restrict__ scratch, float* __restrict__ gk,
                                                            float* __restrict__ gb, float* __restrict__ gv, float* __restrict__ losses) {
    // scratch: [T + Tm] fp32 (dpre | dpm)
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* dpre = scratch;
    float* dpm = scratch + T;
    float la_acc = 0.f, lm_acc = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float e = expf(-pre[t]);
        la_acc += e;
        dpre[t] = -la * e / (float)T;
    }
    for (int t = wave; t < Tm; t += 4) {
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc += hm[(int64_t)t * d + c] * k[c];
        acc = wave_sum(acc) + b[0];
        if (lane == 0) {
            const float e = expf(acc > 0.f ? acc : 0.f);
            dpm[t] = acc > 0.f ? lm * e / (float)Tm : 0.f;
            lm_acc += e;
        }
    }
    __syncthreads();
    for (int r = wave; r < L; r += 4) {     // label rows: dpre += (dy[r] . v) * (pre > 0)
        float acc = 0.f;
        for (int c = lane; c < d_out; c += 64) acc += dy[(int64_t)r * d_out + c] * v[c];
        acc = wave_sum(acc);
        if (lane == 0 && pre[lab[r]] > 0.f) atomicAdd(&dpre[lab[r]], acc);
    }
    la_acc = wave_sum(la_acc);
    lm_acc = wave_sum(lm_acc);
    if (lane == 0) red[wave] = la_acc;
    __syncthreads();
    const float la_tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = lm_acc;
    __syncthreads();
    const float lm_tot = red[0] + red[1] + red[2] + red[3];
    if (tid == 0) {
        losses[0] = la_tot / (float)T;
        losses[1] = lm_tot / (float)Tm;
    }
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float acc = wd * k[c];
        for (int t = 0; t < T; ++t) acc += h[(int64_t)t * d + c] * dpre[t];
        for (int t = 0; t < Tm; ++t) acc += hm[(int64_t)t * d + c] * dpm[t];
        gk[c] = acc;
    }
    for (int c = tid; c < d_out; c += 256) {
        float acc = wd * v[c];
        for (int r = 0; r < L; ++r) acc += fmaxf(pre[lab[r]], 0.f) * dy[(int64_t)r * d_out + c];
        gv[c] = acc;
    }
    if (tid == 0) {
        float acc = wd * b[0];
        for (int t = 0; t < T; ++t) acc += dpre[t];
        for (int t = 0; t < Tm; ++t) acc += dpm[t];
        gb[0] = acc;
    }
}

// ---- gated variant (LLaMA FFN: down(silu(gate(x)) * up(x)); R/configs/tp_vl/llava-v1.5-7b.yaml patches gate_proj AND up_proj) ----
// The new neuron has two keys K2 = [k_gate; k_up] [2,d], biases B2 [2] and a value v [d_out]:
//   pg = h.k_gate + b_gate, pu = h.k_up + b_up, act = silu(pg) * pu, y[r] = ybase[r] + act[lab[r]] * v.
// The reference sums loss_a and loss_m over the in-layers (tp_vl.py:166-177): loss_a = mean exp(-pg) + mean exp(-pu), same for loss_m.
// pre: [2,T] (gate row, up row); scratch: [2,T + Tm] laid out as dpg[T] | dpu[T] | dpmg[Tm] | dpmu[Tm].
__device__ __forceinline__ float tp_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void tp_gated_fwd_kernel(const float* __restrict__ h, int T, int d, const float* __restrict__ K2,
                                                           const float* __restrict__ B2, const int32_t* __restrict__ lab, int L,
                                                           const float* __restrict__ v, const float* __restrict__ ybase, int d_out,
                                                           float* __restrict__ pre, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += 4) {
        float ag = 0.f, au = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float hv = h[(int64_t)t * d + c];
            ag += hv * K2[c];
            au += hv * K2[d + c];
        }
        ag = wave_sum(ag);
        au = wave_sum(au);
        if (lane == 0) {
            pre[t] = ag + B2[0];
            pre[T + t] = au + B2[1];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * d_out; i += 256) {
        const int r = i / d_out, c = i - r * d_out;
        const float pg = pre[lab[r]], pu = pre[T + lab[r]];
        y[i] = ybase[i] + pg * tp_sigmoid(pg) * pu * v[c];
    }
}

__global__ __launch_bounds__(256) void tp_gated_bwd_kernel(const float* __restrict__ h, const float* __restrict__ pre, int T, int d,
                                                           const int32_t* __restrict__ lab, int L, const float* __restrict__ dy,
                                                           int d_out, const float* __restrict__ hm, int Tm, const float* __restrict__ K2,
                                                           const float* __restrict__ B2, const float* __restrict__ v, float la, float lm,
                                                           float wd, float* __restrict__ scratch, float* __restrict__ GK2,
                                                           float* __restrict__ GB2, float* __restrict__ gv, float* __restrict__ losses) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* dpg = scratch;
    float* dpu = scratch + T;
    float* dpmg = scratch + 2 * T;
    float* dpmu = scratch + 2 * T + Tm;
    float la_acc = 0.f, lm_acc = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float eg = expf(-pre[t]), eu = expf(-pre[T + t]);
        la_acc += eg + eu;
        dpg[t] = -la * eg / (float)T;
        dpu[t] = -la * eu / (float)T;
    }
    for (int t = wave; t < Tm; t += 4) {
        float ag = 0.f, au = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float hv = hm[(int64_t)t * d + c];
            ag += hv * K2[c];
            au += hv * K2[d + c];
        }
        ag = wave_sum(ag) + B2[0];
        au = wave_sum(au) + B2[1];
        if (lane == 0) {
            const float eg = expf(ag > 0.f ? ag : 0.f), eu = expf(au > 0.f ? au : 0.f);
            dpmg[t] = ag > 0.f ? lm * eg / (float)Tm : 0.f;
            dpmu[t] = au > 0.f ? lm * eu / (float)Tm : 0.f;
            lm_acc += eg + eu;
        }
    }
    __syncthreads();
    for (int r = wave; r < L; r += 4) {     // label rows: dact = dy[r] . v;  dpg += dact * silu'(pg) * pu;  dpu += dact * silu(pg)
        float acc = 0.f;
        for (int c = lane; c < d_out; c += 64) acc += dy[(int64_t)r * d_out + c] * v[c];
        acc = wave_sum(acc);
        if (lane == 0) {
            const float pg = pre[lab[r]], pu = pre[T + lab[r]], sg = tp_sigmoid(pg);
            atomicAdd(&dpg[lab[r]], acc * sg * (1.f + pg * (1.f - sg)) * pu);
            atomicAdd(&dpu[lab[r]], acc * pg * sg);
        }
    }
    la_acc = wave_sum(la_acc);
    lm_acc = wave_sum(lm_acc);
    if (lane == 0) red[wave] = la_acc;
    __syncthreads();
    const float la_tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = lm_acc;
    __syncthreads();
    const float lm_tot = red[0] + red[1] + red[2] + red[3];
    if (tid == 0) {
        losses[0] = la_tot / (float)T;
        losses[1] = lm_tot / (float)Tm;
    }
    __syncthreads();
    for (int c = tid; c < 2 * d; c += 256) {
        const int which = c >= d, cc = c - which * d;
        const float* dp = which ? dpu : dpg;
        const float* dm = which ? dpmu : dpmg;
        float acc = wd * K2[c];
        for (int t = 0; t < T; ++t) acc += h[(int64_t)t * d + cc] * dp[t];
        for (int t = 0; t < Tm; ++t) acc += hm[(int64_t)t * d + cc] * dm[t];
        GK2[c] = acc;
    }
    for (int c = tid; c < d_out; c += 256) {
        float acc = wd * v[c];
        for (int r = 0; r < L; ++r) {
            const float pg = pre[lab[r]];
            acc += pg * tp_sigmoid(pg) * pre[T + lab[r]] * dy[(int64_t)r * d_out + c];
        }
        gv[c] = acc;
    }
    if (tid < 2) {
        const float* dp = tid ? dpu : dpg;
        const float* dm = tid ? dpmu : dpmg;
        float acc = wd * B2[tid];
        for (int t = 0; t < T; ++t) acc += dp[t];
        for (int t = 0; t < Tm; ++t) acc += dm[t];
        GB2[tid] = acc;
    }
}

extern "C" int devqa_tp_gated_neuron_fwd(const float* h, int T, int d, const float* K2, const float* B2, const int32_t* lab, int L,
                                         const float* v, const float* ybase, int d_out, float* pre, float* y, void* stream) {
    DEVQA_CHECK_ARG(h && K2 && B2 && lab && v && ybase && pre && y && T > 0 && d > 0 && L > 0 && d_out > 0,
                    "tp_gated_neuron_fwd: bad arguments");
    hipLaunchKernelGGL(tp_gated_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, h, T, d, K2, B2, lab, L, v, ybase, d_out, pre, y);
    DEVQA_LAUNCH_CHECK("tp_gated_neuron_fwd");
    return DEVQA_OK;
}

extern "C" int devqa_tp_gated_neuron_bwd(const float* h, const float* pre, int T, int d, const int32_t* lab, int L, const float* dy,
                                         int d_out, const float* hm, int Tm, const float* K2, const float* B2, const float* v,
                                         float lambda_a, float lambda_m, float weight_decay, float* scratch, float* GK2, float* GB2,
                                         float* gv, float* losses, void* stream) {
    DEVQA_CHECK_ARG(h && pre && lab && dy && hm && K2 && B2 && v && scratch && GK2 && GB2 && gv && losses && T > 0 && Tm > 0 && L > 0,
                    "tp_gated_neuron_bwd: bad arguments");
    hipLaunchKernelGGL(tp_gated_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, h, pre, T, d, lab, L, dy, d_out, hm, Tm, K2, B2, v,
                       lambda_a, lambda_m, weight_decay, scratch, GK2, GB2, gv, losses);
    DEVQA_LAUNCH_CHECK("tp_gated_neuron_bwd");
    return DEVQA_OK;
}

extern "C" int devqa_tp_neuron_fwd(const float* h, int T, int d, const float* k, const float* b, const int32_t* lab, int L,
                                   const float* v, const float* ybase, int d_out, float* pre, float* y, void* stream) {
    DEVQA_CHECK_ARG(h && k && b && lab && v && ybase && pre && y && T > 0 && d > 0 && L > 0 && d_out > 0, "tp_neuron_fwd: bad arguments");
    hipLaunchKernelGGL(tp_neuron_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, h, T, d, k, b, lab, L, v, ybase, d_out, pre, y);
    DEVQA_LAUNCH_CHECK("tp_neuron_fwd");
    return DEVQA_OK;
}

extern "C" int devqa_tp_neuron_bwd(const float* h, const float* pre, int T, int d, const int32_t* lab, int L, const float* dy, int d_out,
                                   const float* hm, int Tm, const float* k, const float* b, const float* v, float lambda_a,
                                   float lambda_m, float weight_decay, float* scratch, float* gk, float* gb, float* gv, float* losses,
                                   void* stream) {
    DEVQA_CHECK_ARG(h && pre && lab && dy && hm && k && b && v && scratch && gk && gb && gv && losses && T > 0 && Tm > 0 && L > 0,
                    "tp_neuron_bwd: bad arguments");
    hipLaunchKernelGGL(tp_neuron_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, h, pre, T, d, lab, L, dy, d_out, hm, Tm, k, b, v,
                       lambda_a, lambda_m, weight_decay, scratch, gk, gb, gv, losses);
    DEVQA_LAUNCH_CHECK("tp_neuron_bwd");
    return DEVQA_OK;
}
