// Attention backward for the MEND_VL edit path (one backward pass per edit through the last few decoder layers;
// R/editor/vllm_editors/mend_vl/mend_vl.py:177-186 reaches it through torch.autograd).  Not an MFMA kernel: the path
// runs once per edit on <= a few hundred rows, so it is written for exactness and simplicity -- fp32 arithmetic on
// the VALU, K/V (resp. Q/dO) staged through LDS in 64-row chunks, lane = key (resp. query) for the score phase and
// lane = channel for the accumulation phase.
//
// Supports the own-key part of the packed descriptor only (kp_len == 0: every sequence attends to its own rows,
// causal or full) -- what the edit path packs.  Two kernels:
//   dq   : per query row i: L_i = logsumexp_j(s_ij), D_i = dO_i . O_i, dq_i = scale * sum_j P_ij (dP_ij - D_i) k_j
//   dkdv : per key row j  : dv_j = sum_i P_ij dO_i,  dk_j = scale * sum_i P_ij (dP_ij - D_i) q_i
// with s_ij = scale * q_i . k_j, P = softmax(s), dP_ij = dO_i . v_j.  stats[(row, head)] = (L_i, D_i) is written by dq
// and read by dkdv.
#include "common.h"

#define AB_DH 128   // max head dim
#define AB_CH 64    // rows per staged chunk
#define AB_RB 16    // rows per workgroup (4 waves x 4 rows)

template <typename T> __device__ __forceinline__ float ab_ld(const T* p);
template <> __device__ __forceinline__ float ab_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ab_ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void ab_st(T* p, float v);
template <> __device__ __forceinline__ void ab_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void ab_st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// stage `n` rows (row r -> global row g0 + r, zero beyond n) of head h into fp32 LDS [AB_CH][AB_DH + 1]
template <typename T>
__device__ __forceinline__ void ab_stage(float (*dst)[AB_DH + 1], const T* src, int64_t ld, int64_t g0, int n, int h, int dh) {
    for (int i = threadIdx.x; i < AB_CH * dh; i += 256) {
        const int r = i / dh, c = i - r * dh;
        dst[r][c] = r < n ? ab_ld<T>(src + (g0 + r) * ld + h * dh + c) : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                          const T* __restrict__ v, int64_t ldv, const T* __restrict__ o, int64_t ldo,
                                                          const T* __restrict__ dO, int64_t lddo, T* __restrict__ dq, int64_t lddq,
                                                          float2* __restrict__ stats, const int32_t* __restrict__ seq_desc, int H,
                                                          int dh, float scale, int causal, int q_blocks) {
    __shared__ float Ks[AB_CH][AB_DH + 1];
    __shared__ float Vs[AB_CH][AB_DH + 1];
    __shared__ float qs[AB_RB][AB_DH];
    __shared__ float dOs[AB_RB][AB_DH];
    const int bid = blockIdx.x;
    const int qb = bid % q_blocks, h = (bid / q_blocks) % H, s = bid / (q_blocks * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], ko_start = d[4], ko_len = d[5];
    const int i0 = qb * AB_RB;
    if (i0 >= q_len) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int causal_off = ko_len - q_len;
    // this block's query rows and their dO, D_i
    for (int i = threadIdx.x; i < AB_RB * dh; i += 256) {
        const int r = i / dh, c = i - r * dh;
        const bool ok = i0 + r < q_len;
        qs[r][c] = ok ? ab_ld<T>(q + (int64_t)(q_start + i0 + r) * ldq + h * dh + c) : 0.f;
        dOs[r][c] = ok ? ab_ld<T>(dO + (int64_t)(q_start + i0 + r) * lddo + h * dh + c) : 0.f;
    }
    __syncthreads();
    float Dv[4], mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wave * 4 + r;
        float acc = 0.f;
        if (i0 + row < q_len)
            for (int c = lane; c < dh; c += 64) acc += dOs[row][c] * ab_ld<T>(o + (int64_t)(q_start + i0 + row) * ldo + h * dh + c);
        Dv[r] = wave_sum(acc);
        mrun[r] = -INFINITY;
        lrun[r] = 0.f;
    }
    const int hi = causal ? min(ko_len, i0 + AB_RB + causal_off) : ko_len;   // keys any row of this block can see
    // ---- pass A: softmax statistics
    for (int c0 = 0; c0 < hi; c0 += AB_CH) {
        __syncthreads();
        ab_stage<T>(Ks, k, ldk, ko_start + c0, min(AB_CH, ko_len - c0), h, dh);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r, i = i0 + row;
            const int j = c0 + lane;
            bool ok = i < q_len && j < ko_len;
            if (causal) ok = ok && j <= i + causal_off;
            float sv = 0.f;
            for (int c = 0; c < dh; ++c) sv += qs[row][c] * Ks[lane][c];
            sv = ok ? sv * scale : -INFINITY;
            const float mloc = wave_max(sv);
            const float mnew = fmaxf(mrun[r], mloc);
            if (mnew != -INFINITY) {
                const float p = ok ? __expf(sv - mnew) : 0.f;
                lrun[r] = lrun[r] * (mrun[r] == -INFINITY ? 0.f : __expf(mrun[r] - mnew)) + wave_sum(p);
                mrun[r] = mnew;
            }
        }
    }
    float Lv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Lv[r] = lrun[r] > 0.f ? mrun[r] + __logf(lrun[r]) : INFINITY;   // no visible key: every P is 0
        const int i = i0 + wave * 4 + r;
        if (lane == 0 && i < q_len) stats[(int64_t)(q_start + i) * H + h] = make_float2(Lv[r], Dv[r]);
    }
    // ---- pass B: dq
    float dqa[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) dqa[r][0] = dqa[r][1] = 0.f;
    for (int c0 = 0; c0 < hi; c0 += AB_CH) {
        __syncthreads();
        ab_stage<T>(Ks, k, ldk, ko_start + c0, min(AB_CH, ko_len - c0), h, dh);
        ab_stage<T>(Vs, v, ldv, ko_start + c0, min(AB_CH, ko_len - c0), h, dh);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r, i = i0 + row;
            const int j = c0 + lane;
            bool ok = i < q_len && j < ko_len;
            if (causal) ok = ok && j <= i + causal_off;
            float sv = 0.f, dp = 0.f;
            for (int c = 0; c < dh; ++c) {
                sv += qs[row][c] * Ks[lane][c];
                dp += dOs[row][c] * Vs[lane][c];
            }
            const float p = ok ? __expf(sv * scale - Lv[r]) : 0.f;
            const float ds = p * (dp - Dv[r]) * scale;
            for (int jj = 0; jj < AB_CH; ++jj) {
                const float dsj = __shfl(ds, jj, 64);
                dqa[r][0] += dsj * Ks[jj][lane];             // lane = channel (columns >= dh hold stale data; never stored)
                if (dh > 64) dqa[r][1] += dsj * Ks[jj][lane + 64];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 4 + r;
        if (i >= q_len) continue;
        T* row = dq + (int64_t)(q_start + i) * lddq + h * dh;
        if (lane < dh) ab_st<T>(row + lane, dqa[r][0]);
        if (lane + 64 < dh) ab_st<T>(row + lane + 64, dqa[r][1]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                            const T* __restrict__ v, int64_t ldv, const T* __restrict__ dO, int64_t lddo,
                                                            T* __restrict__ dk, int64_t lddk, T* __restrict__ dv, int64_t lddv,
                                                            const float2* __restrict__ stats, const int32_t* __restrict__ seq_desc,
                                                            int H, int dh, float scale, int causal, int k_blocks) {
    __shared__ float Qs[AB_CH][AB_DH + 1];
    __shared__ float Gs[AB_CH][AB_DH + 1];   // dO
    __shared__ float ks[AB_RB][AB_DH];
    __shared__ float vs[AB_RB][AB_DH];
    __shared__ float2 st[AB_CH];
    const int bid = blockIdx.x;
    const int kb = bid % k_blocks, h = (bid / k_blocks) % H, s = bid / (k_blocks * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], ko_start = d[4], ko_len = d[5];
    const int j0 = kb * AB_RB;
    if (j0 >= ko_len) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int causal_off = ko_len - q_len;
    for (int i = threadIdx.x; i < AB_RB * dh; i += 256) {
        const int r = i / dh, c = i - r * dh;
        const bool ok = j0 + r < ko_len;
        ks[r][c] = ok ? ab_ld<T>(k + (int64_t)(ko_start + j0 + r) * ldk + h * dh + c) : 0.f;
        vs[r][c] = ok ? ab_ld<T>(v + (int64_t)(ko_start + j0 + r) * ldv + h * dh + c) : 0.f;
    }
    float dka[4][2], dva[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) dka[r][0] = dka[r][1] = dva[r][0] = dva[r][1] = 0.f;
    const int lo = causal ? max(0, j0 - causal_off) : 0;   // first query that can see a key of this block
    for (int c0 = (lo / AB_CH) * AB_CH; c0 < q_len; c0 += AB_CH) {
        __syncthreads();
        const int n = min(AB_CH, q_len - c0);
        ab_stage<T>(Qs, q, ldq, q_start + c0, n, h, dh);
        ab_stage<T>(Gs, dO, lddo, q_start + c0, n, h, dh);
        if (threadIdx.x < AB_CH)
            st[threadIdx.x] = (int)threadIdx.x < n ? stats[(int64_t)(q_start + c0 + threadIdx.x) * H + h] : make_float2(INFINITY, 0.f);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r, j = j0 + row;
            const int i = c0 + lane;
            bool ok = j < ko_len && i < q_len;
            if (causal) ok = ok && j <= i + causal_off;
            float sv = 0.f, dp = 0.f;
            for (int c = 0; c < dh; ++c) {
                sv += Qs[lane][c] * ks[row][c];
                dp += Gs[lane][c] * vs[row][c];
            }
            const float2 sd = st[lane];
            const float p = ok ? __expf(sv * scale - sd.x) : 0.f;
            const float ds = p * (dp - sd.y) * scale;
            for (int ii = 0; ii < AB_CH; ++ii) {
                const float pi = __shfl(p, ii, 64), dsi = __shfl(ds, ii, 64);
                dva[r][0] += pi * Gs[ii][lane];
                dka[r][0] += dsi * Qs[ii][lane];
                if (dh > 64) {
                    dva[r][1] += pi * Gs[ii][lane + 64];
                    dka[r][1] += dsi * Qs[ii][lane + 64];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + wave * 4 + r;
        if (j >= ko_len) continue;
        T* rk = dk + (int64_t)(ko_start + j) * lddk + h * dh;
        T* rv = dv + (int64_t)(ko_start + j) * lddv + h * dh;
        if (lane < dh) { ab_st<T>(rk + lane, dka[r][0]); ab_st<T>(rv + lane, dva[r][0]); }
        if (lane + 64 < dh) { ab_st<T>(rk + lane + 64, dka[r][1]); ab_st<T>(rv + lane + 64, dva[r][1]); }
    }
}

template <typename T>
static int attention_bwd_impl(const T* q, int64_t ldq, const T* k, int64_t ldk, const T* v, int64_t ldv, const T* o, int64_t ldo,
                              const T* dO, int64_t lddo, T* dq, int64_t lddq, T* dk, int64_t lddk, T* dv, int64_t lddv, float* stats,
                              const int32_t* seq_desc, int n_seq, int max_len, int H, int dh, float scale, int causal, void* stream) {
    DEVQA_CHECK_ARG(q && k && v && o && dO && dq && dk && dv && stats && seq_desc, "attention_bwd: null pointer");
    DEVQA_CHECK_SHAPE(n_seq > 0 && max_len > 0 && H > 0 && dh > 0 && dh <= AB_DH, "attention_bwd: bad sizes (dh <= 128)");
    const int blocks = (max_len + AB_RB - 1) / AB_RB;
    const long grid = (long)n_seq * H * blocks;
    DEVQA_CHECK_SHAPE(grid < 2147483647L, "attention_bwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, dO, lddo, dq, lddq,
                       reinterpret_cast<float2*>(stats), seq_desc, H, dh, scale, causal, blocks);
    DEVQA_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<T>, dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, dO, lddo, dk, lddk, dv,
                       lddv, reinterpret_cast<const float2*>(stats), seq_desc, H, dh, scale, causal, blocks);
    DEVQA_LAUNCH_CHECK("attn_bwd_dkdv");
    return DEVQA_OK;
}

extern "C" int devqa_attention_bwd(const devqa_bf16* q, int64_t ldq, const devqa_bf16* k, int64_t ldk, const devqa_bf16* v, int64_t ldv,
                                   const devqa_bf16* o, int64_t ldo, const devqa_bf16* d_out, int64_t lddo, devqa_bf16* dq, int64_t lddq,
                                   devqa_bf16* dk, int64_t lddk, devqa_bf16* dv, int64_t lddv, float* stats, const int32_t* seq_desc,
                                   int n_seq, int max_len, int H, int dh, float scale, int causal, void* stream) {
    return attention_bwd_impl<bf16_t>(q, ldq, k, ldk, v, ldv, o, ldo, d_out, lddo, dq, lddq, dk, lddk, dv, lddv, stats, seq_desc, n_seq,
                                      max_len, H, dh, scale, causal, stream);
}

extern "C" int devqa_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       const float* o, int64_t ldo, const float* d_out, int64_t lddo, float* dq, int64_t lddq, float* dk,
                                       int64_t lddk, float* dv, int64_t lddv, float* stats, const int32_t* seq_desc, int n_seq,
                                       int max_len, int H, int dh, float scale, int causal, void* stream) {
    return attention_bwd_impl<float>(q, ldq, k, ldk, v, ldv, o, ldo, d_out, lddo, dq, lddq, dk, lddk, dv, lddv, stats, seq_desc, n_seq,
                                     max_len, H, dh, scale, causal, stream);
}
