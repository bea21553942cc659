// Cosine / dot-product top-k retrieval (dynamic-eval `finds_sim`, IKE `semantic_search`).
//
//   1. row_inv_norm : 1/||row|| for corpus and queries (fp32 accumulate, one wave per row)
//   2. score_tile   : S[Q,N] = Qm . C^T on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32; bit-equal to
//                     an fmaf chain), 64x64 tile per workgroup, K streamed through LDS in 64-float
//                     chunks (row stride 66 floats -> conflict-free ds_read_b32 fragment reads)
//   3. select       : one workgroup per query: k+8 successive arg-max sweeps over its score row
//                     (L2-resident), candidates re-scored in fp64, sorted by (score desc, id asc)
//
// Exactness: the fp32 scan only has to place the true top-k inside the top-(k+8); the final order
// and the reported scores come from the fp64 re-score, so indices match a float64 brute force.
#include "common.h"

#define CT_MARGIN 8
#define CT_MAXK 32

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, int R, int D, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float s = 0.f;
    for (int c = lane; c < (D >> 2); c += 64) {
        const float4 v = xr[c];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum(s);
    if (lane == 0) inv[row] = s > 0.f ? 1.f / sqrtf(s) : 0.f;
}

#define ST_LD 66
__global__ __launch_bounds__(256) void score_tile_kernel(const float* __restrict__ corpus, const float* __restrict__ queries,
                                                         int N, int Q, int D, const float* __restrict__ inv_c,
                                                         const float* __restrict__ inv_q, float* __restrict__ scores) {
    __shared__ float Qs[64 * ST_LD];
    __shared__ float Cs[64 * ST_LD];
    const int n0 = blockIdx.x * 64, q0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    float4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += 64) {
        __syncthreads();
        // stage 64 rows x 64 floats of each operand (float4 global loads, scalar LDS stores)
        for (int i = tid; i < 64 * 16; i += 256) {
            const int r = i >> 4, c4 = i & 15;
            const int kk = k0 + c4 * 4;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), cv = qv;
            if (kk < D) {
                if (q0 + r < Q) qv = *reinterpret_cast<const float4*>(queries + (int64_t)(q0 + r) * D + kk);
                if (n0 + r < N) cv = *reinterpret_cast<const float4*>(corpus + (int64_t)(n0 + r) * D + kk);
            }
            float* qd = Qs + r * ST_LD + c4 * 4;
            float* cd = Cs + r * ST_LD + c4 * 4;
            qd[0] = qv.x; qd[1] = qv.y; qd[2] = qv.z; qd[3] = qv.w;
            cd[0] = cv.x; cd[1] = cv.y; cd[2] = cv.z; cd[3] = cv.w;
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float af = Qs[(wave * 16 + fr) * ST_LD + ks * 4 + fq];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bf = Cs[(j * 16 + fr) * ST_LD + ks * 4 + fq];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0);
            }
        }
    }
    // C/D map: col = lane&15 (corpus row within the 16-tile), row = (lane>>4)*4 + reg (query)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + fr;
        if (n >= N) continue;
        const float ic = inv_c ? inv_c[n] : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + wave * 16 + fq * 4 + r;
            if (qi >= Q) continue;
            const float iq = inv_q ? inv_q[qi] : 1.f;
            scores[(int64_t)qi * N + n] = acc[j][r] * ic * iq;
        }
    }
}

__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ scores, const float* __restrict__ corpus,
                                                          const float* __restrict__ queries, int N, int D, int k,
                                                          int norm_c, int norm_q, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_score) {
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    __shared__ int cand_i[CT_MAXK + CT_MARGIN];
    __shared__ double cand_s[CT_MAXK + CT_MARGIN];
    __shared__ float prev_s;
    __shared__ int prev_i;
    const int qi = blockIdx.x;
    const float* row = scores + (int64_t)qi * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = min(k + CT_MARGIN, N);
    if (tid == 0) {
        prev_s = INFINITY;
        prev_i = -1;
    }
    __syncthreads();
    for (int p = 0; p < nc; ++p) {
        const float ps = prev_s;
        const int pi = prev_i;
        float bs = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < N; i += 256) {
            const float s = row[i];
            // strictly after (ps, pi) in (score desc, id asc) order
            const bool after = (s < ps) || (s == ps && i > pi);
            if (after && (s > bs || (s == bs && i < bi))) {
                bs = s;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bs, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (s2 > bs || (s2 == bs && i2 < bi)) {
                bs = s2;
                bi = i2;
            }
        }
        if (lane == 0) {
            red_s[wave] = bs;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red_s[w] > bs || (red_s[w] == bs && red_i[w] < bi)) {
                    bs = red_s[w];
                    bi = red_i[w];
                }
            prev_s = bs;
            prev_i = bi;
            cand_i[p] = bi;
        }
        __syncthreads();
    }
    // fp64 re-score (dot and norms), one wave per candidate round-robin
    const float* qv = queries + (int64_t)qi * D;
    double qn = 0.0;
    if (norm_q) {
        for (int c = lane; c < D; c += 64) qn += (double)qv[c] * (double)qv[c];
        qn = wave_sum_d(qn);
    }
    for (int p = wave; p < nc; p += 4) {
        const int ci = cand_i[p];
        double dot = 0.0, cn = 0.0;
        if (ci >= 0 && ci < N) {
            const float* cv = corpus + (int64_t)ci * D;
            for (int c = lane; c < D; c += 64) {
                const double x = (double)cv[c];
                dot += x * (double)qv[c];
                cn += x * x;
            }
        }
        dot = wave_sum_d(dot);
        cn = wave_sum_d(cn);
        if (lane == 0) {
            double s = dot;
            if (norm_c) s = cn > 0.0 ? s / sqrt(cn) : 0.0;
            if (norm_q) s = qn > 0.0 ? s / sqrt(qn) : 0.0;
            cand_s[p] = s;
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int a = 1; a < nc; ++a) {  // insertion sort by (score desc, id asc)
            const double s = cand_s[a];
            const int id = cand_i[a];
            int b = a - 1;
            while (b >= 0 && (cand_s[b] < s || (cand_s[b] == s && cand_i[b] > id))) {
                cand_s[b + 1] = cand_s[b];
                cand_i[b + 1] = cand_i[b];
                --b;
            }
            cand_s[b + 1] = s;
            cand_i[b + 1] = id;
        }
        for (int a = 0; a < k; ++a) {
            out_idx[(int64_t)qi * k + a] = a < nc ? (int64_t)cand_i[a] : (int64_t)-1;
            out_score[(int64_t)qi * k + a] = a < nc ? (float)cand_s[a] : -INFINITY;
        }
    }
}

// Register-resident selection: the query's score row (N <= 256*NPT) is loaded ONCE, NPT values per thread;
// every thread caches its local (max, index); each of the k+8 rounds is one workgroup arg-max over the cached
// pairs, after which only the winning thread clears its element and rescans its NPT registers.  Same
// (score desc, id asc) order as the multi-pass kernel it replaces; candidates are re-scored in fp64 as before.
template <int NPT>
__global__ __launch_bounds__(256) void topk_select_reg_kernel(const float* __restrict__ scores, const float* __restrict__ corpus,
                                                              const float* __restrict__ queries, int N, int D, int k,
                                                              int norm_c, int norm_q, int64_t* __restrict__ out_idx,
                                                              float* __restrict__ out_score) {
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    __shared__ int cand_i[CT_MAXK + CT_MARGIN];
    __shared__ double cand_s[CT_MAXK + CT_MARGIN];
    __shared__ int win_i;
    const int qi = blockIdx.x;
    const float* row = scores + (int64_t)qi * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = min(k + CT_MARGIN, N);
    float v[NPT];
    float lmax = -INFINITY;
    int lidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
        const int i = r * 256 + tid;  // strided: coalesced loads, ascending ids per thread
        v[r] = i < N ? row[i] : -INFINITY;
        if (i < N && v[r] > lmax) {
            lmax = v[r];
            lidx = i;
        }
    }
    for (int p = 0; p < nc; ++p) {
        float bs = lmax;
        int bi = lidx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bs, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (s2 > bs || (s2 == bs && i2 < bi)) {
                bs = s2;
                bi = i2;
            }
        }
        if (lane == 0) {
            red_s[wave] = bs;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red_s[w] > bs || (red_s[w] == bs && red_i[w] < bi)) {
                    bs = red_s[w];
                    bi = red_i[w];
                }
            win_i = bi;
            cand_i[p] = bi;
        }
        __syncthreads();
        const int w = win_i;
        if (w < N && (w & 255) == tid) {  // the owner clears it and rescans its registers
            lmax = -INFINITY;
            lidx = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < NPT; ++r) {
                const int i = r * 256 + tid;
                if (i == w) v[r] = -INFINITY;
                if (i < N && i != w && v[r] > lmax) {
                    lmax = v[r];
                    lidx = i;
                }
            }
            if (lmax == -INFINITY) {  // exhausted (or only -inf left): keep ids increasing among the rest
                lidx = 0x7fffffff;
            }
        }
    }
    // fp64 re-score (dot and norms), one wave per candidate round-robin
    const float* qv = queries + (int64_t)qi * D;
    double qn = 0.0;
    if (norm_q) {
        for (int c = lane; c < D; c += 64) qn += (double)qv[c] * (double)qv[c];
        qn = wave_sum_d(qn);
    }
    for (int p = wave; p < nc; p += 4) {
        const int ci = cand_i[p];
        double dot = 0.0, cn = 0.0;
        if (ci >= 0 && ci < N) {
            const float* cv = corpus + (int64_t)ci * D;
            for (int c = lane; c < D; c += 64) {
                const double x = (double)cv[c];
                dot += x * (double)qv[c];
                cn += x * x;
            }
        }
        dot = wave_sum_d(dot);
        cn = wave_sum_d(cn);
        if (lane == 0) {
            double s = dot;
            if (norm_c) s = cn > 0.0 ? s / sqrt(cn) : 0.0;
            if (norm_q) s = qn > 0.0 ? s / sqrt(qn) : 0.0;
            cand_s[p] = (ci >= 0 && ci < N) ? s : -INFINITY;
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int a = 1; a < nc; ++a) {
            const double s = cand_s[a];
            const int id = cand_i[a];
            int b = a - 1;
            while (b >= 0 && (cand_s[b] < s || (cand_s[b] == s && cand_i[b] > id))) {
                cand_s[b + 1] = cand_s[b];
                cand_i[b + 1] = cand_i[b];
                --b;
            }
            cand_s[b + 1] = s;
            cand_i[b + 1] = id;
        }
        for (int a = 0; a < k; ++a) {
            out_idx[(int64_t)qi * k + a] = a < nc ? (int64_t)cand_i[a] : (int64_t)-1;
            out_score[(int64_t)qi * k + a] = a < nc ? (float)cand_s[a] : -INFINITY;
        }
    }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t devqa_cosine_topk_workspace(int N, int Q, int k) {
    (void)k;
    if (N <= 0 || Q <= 0) return 256;
    return align256((int64_t)Q * N * 4) + align256((int64_t)N * 4) + align256((int64_t)Q * 4);
}

extern "C" int devqa_cosine_topk(const float* corpus, const float* queries, int N, int Q, int D, int k, int normalize_corpus,
                                 int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream) {
    DEVQA_CHECK_ARG(corpus && queries && out_idx && out_score && workspace, "cosine_topk: null pointer");
    if (Q == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(N > 0 && Q > 0 && D > 0 && D % 4 == 0 && D <= 4096, "cosine_topk: bad dims N=%d Q=%d D=%d", N, Q, D);
    DEVQA_CHECK_SHAPE(k >= 1 && k <= CT_MAXK, "cosine_topk: k=%d unsupported (1..%d)", k, CT_MAXK);
    DEVQA_CHECK_SHAPE((((uintptr_t)workspace) & 255) == 0, "cosine_topk: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int ph = devqa_prof_begin(DEVQA_PROF_COSINE, st);     // the whole call: norms + score tiles + selection
    char* ws = (char*)workspace;
    float* scores = (float*)ws;
    float* inv_c = (float*)(ws + align256((int64_t)Q * N * 4));
    float* inv_q = (float*)((char*)inv_c + align256((int64_t)N * 4));
    if (normalize_corpus) {
        hipLaunchKernelGGL(row_inv_norm_kernel, dim3((N + 3) / 4), dim3(256), 0, st, corpus, N, D, inv_c);
        DEVQA_LAUNCH_CHECK("row_inv_norm(corpus)");
    }
    if (normalize_queries) {
        hipLaunchKernelGGL(row_inv_norm_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, queries, Q, D, inv_q);
        DEVQA_LAUNCH_CHECK("row_inv_norm(queries)");
    }
    hipLaunchKernelGGL(score_tile_kernel, dim3((N + 63) / 64, (Q + 63) / 64), dim3(256), 0, st, corpus, queries, N, Q, D,
                       normalize_corpus ? inv_c : nullptr, normalize_queries ? inv_q : nullptr, scores);
    DEVQA_LAUNCH_CHECK("score_tile");
    if (N <= 256 * 16)
        hipLaunchKernelGGL(topk_select_reg_kernel<16>, dim3(Q), dim3(256), 0, st, scores, corpus, queries, N, D, k,
                           normalize_corpus, normalize_queries, out_idx, out_score);
    else if (N <= 256 * 80)
        hipLaunchKernelGGL(topk_select_reg_kernel<80>, dim3(Q), dim3(256), 0, st, scores, corpus, queries, N, D, k,
                           normalize_corpus, normalize_queries, out_idx, out_score);
    else
        hipLaunchKernelGGL(topk_select_kernel, dim3(Q), dim3(256), 0, st, scores, corpus, queries, N, D, k, normalize_corpus,
                           normalize_queries, out_idx, out_score);
    devqa_prof_end(ph, 4.0 * (double)N * D, st);
    DEVQA_LAUNCH_CHECK("topk_select");
    return DEVQA_OK;
}
