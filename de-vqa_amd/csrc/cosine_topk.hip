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
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include <rocprim/warp/warp_reduce.hpp>

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

// ---- few queries (IKE_VL / LTE_VL retrieval: Q = 1 per probe) ------------------------------------------------------------------
// The tiled path above is four launches and a selection by ONE workgroup whose k + 8 rounds each cross two barriers: 91 us at k = 5
// and 205 us at k = 32 over a 15000 x 384 corpus that streams in 3 us -- launch latency and a serial selection, not bytes.  For Q <= 4,
// two launches:
//   score_select : a wave owns 64 corpus rows: each row is read once for its norm and its <= 4 dot products, lane r keeps row r's
//                  scores; then k + 8 arg-max rounds over the 64 lanes (shuffles only) leave a SORTED candidate list per wave and query
//   merge_final  : one workgroup per query: a 64-way merge of sorted lists per wave (lane = list, one head each; the winner advances
//                  its pointer), the four waves' results merged once more, then the fp64 re-score and (score desc, id asc) order of
//                  the kernels above.
// The query's norm scales all of its scores alike and is left to the fp64 re-score.
// arg-max of (score desc, id asc) over a wave, id 0x7fffffff = nothing.  Two DPP all-reductions (rocPRIM warp_reduce: max of the
// scores, then min of the ids that hold it) instead of a 6-step butterfly of ds_bpermute pairs: the selection rounds are dependent
// chains, and 12 LDS-crossbar round trips per round were what the first version of this path spent its time on.
typedef rocprim::warp_reduce<float, 64, true> ct_wr_f;
typedef rocprim::warp_reduce<int, 64, true> ct_wr_i;
struct ct_wr_storage {
    ct_wr_f::storage_type f;
    ct_wr_i::storage_type i;
};
__device__ __forceinline__ void ct_wave_argmax(float& bs, int& bi, ct_wr_storage& st) {
    float m;
    ct_wr_f().reduce(bi == 0x7fffffff ? -INFINITY : bs, m, st.f, rocprim::maximum<float>());
    int mi;
    ct_wr_i().reduce((bi != 0x7fffffff && bs == m) ? bi : 0x7fffffff, mi, st.i, rocprim::minimum<int>());
    bs = m;
    bi = mi;
}

template <int QMAX>
__global__ __launch_bounds__(256) void score_select_kernel(const float* __restrict__ corpus, const float* __restrict__ queries, int N, int Q,
                                                           int D, int norm_c, int nc, int n_lists, float* __restrict__ list_s,
                                                           int* __restrict__ list_i) {
    __shared__ ct_wr_storage wr_st[4];
    ct_wr_storage& wst = wr_st[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_lists) return;
    const int row0 = w * 64;
    float mine[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) mine[q] = -INFINITY;
    const int nrows = min(64, N - row0);
    // rows in groups of RG with all of a group's loads issued before the first use: one row at a time the loop was a chain of 64
    // global-memory round trips (59 us for 64 rows)
    constexpr int RG = 8;
    const int nv = D >> 2;
    for (int r0 = 0; r0 < nrows; r0 += RG) {
        float nrm[RG], acc[RG][QMAX];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            nrm[g] = 0.f;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) acc[g][q] = 0.f;
        }
        for (int c = lane; c < nv; c += 64) {
            float4 v[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g)
                v[g] = reinterpret_cast<const float4*>(corpus + (int64_t)(row0 + min(r0 + g, nrows - 1)) * D)[c];
            float4 u[QMAX];
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
                u[q] = q < Q ? reinterpret_cast<const float4*>(queries + (int64_t)q * D)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                nrm[g] += (v[g].x * v[g].x + v[g].y * v[g].y) + (v[g].z * v[g].z + v[g].w * v[g].w);
#pragma unroll
                for (int q = 0; q < QMAX; ++q)
                    acc[g][q] += (v[g].x * u[q].x + v[g].y * u[q].y) + (v[g].z * u[q].z + v[g].w * u[q].w);
            }
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int r = r0 + g;
            float n2;
            ct_wr_f().reduce(nrm[g], n2, wst.f, rocprim::plus<float>());       // DPP all-reduce
            const float ic = norm_c ? (n2 > 0.f ? 1.f / sqrtf(n2) : 0.f) : 1.f;
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
                if (q < Q) {
                    float sc;
                    ct_wr_f().reduce(acc[g][q], sc, wst.f, rocprim::plus<float>());
                    if (lane == r && r < nrows) mine[q] = sc * ic;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        if (q >= Q) continue;
        float v = mine[q];
        int id = lane < nrows ? row0 + lane : 0x7fffffff;
        float os = -INFINITY;
        int oi = -1;
        for (int p = 0; p < nc; ++p) {
            float bs = id == 0x7fffffff ? -INFINITY : v;
            int bi = id;
            ct_wave_argmax(bs, bi, wst);
            if (lane == p) {
                os = bs;
                oi = bi == 0x7fffffff ? -1 : bi;
            }
            if (bi != 0x7fffffff && id == bi) id = 0x7fffffff;      // the owner retires
        }
        if (lane < nc) {
            list_s[((int64_t)q * n_lists + w) * 64 + lane] = os;
            list_i[((int64_t)q * n_lists + w) * 64 + lane] = oi;
        }
    }
}

__global__ __launch_bounds__(256) void merge_final_kernel(const float* __restrict__ list_s, const int* __restrict__ list_i, int n_lists,
                                                          const float* __restrict__ corpus, const float* __restrict__ queries, int N, int D,
                                                          int k, int nc, int norm_c, int norm_q, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_score) {
    __shared__ float run_s[4][2][64];      // per wave: running merged list (ping-pong)
    __shared__ int run_i[4][2][64];
    __shared__ float fin_s[64];
    __shared__ int cand_i[CT_MAXK + CT_MARGIN];
    __shared__ double cand_s[CT_MAXK + CT_MARGIN];
    __shared__ ct_wr_storage wr_st[4];
    const int qi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ct_wr_storage& wst = wr_st[wave];
    const float* ls = list_s + (int64_t)qi * n_lists * 64;
    const int* li = list_i + (int64_t)qi * n_lists * 64;
    // wave w merges lists w, w + 4, ... in batches of 31 + its running list, each batch staged in LDS first (a winner advancing its
    // pointer then costs an LDS read, not a global-memory round trip inside the dependent chain of rounds)
    __shared__ float st_s[4][32][CT_MAXK + CT_MARGIN];
    __shared__ int st_i[4][32][CT_MAXK + CT_MARGIN];
    int cur = 0;
    bool have = false;
    for (int j0 = wave; j0 < n_lists; j0 += 4 * 31) {
        const int nb = min(31, (n_lists - j0 + 3) / 4);
        for (int e = lane; e < nb * nc; e += 64) {          // list e / nc, entry e % nc
            const int l = e / nc, p = e - l * nc;
            st_s[wave][l][p] = ls[(int64_t)(j0 + 4 * l) * 64 + p];
            st_i[wave][l][p] = li[(int64_t)(j0 + 4 * l) * 64 + p];
        }
        if (have)
            for (int p = lane; p < nc; p += 64) {
                st_s[wave][31][p] = run_s[wave][cur][p];
                st_i[wave][31][p] = run_i[wave][cur][p];
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS stores are visible to its lanes
        int ptr = 0;
        float hv = -INFINITY;
        int hi = 0x7fffffff;
        const int my = lane < nb ? lane : (lane == 31 && have ? 31 : -1);
        if (my >= 0) {
            hv = st_s[wave][my][0];
            hi = st_i[wave][my][0];
            if (hi < 0) hi = 0x7fffffff;
        }
        float* ds = run_s[wave][cur ^ 1];
        int* di = run_i[wave][cur ^ 1];
        for (int p = 0; p < nc; ++p) {
            float bs = hi == 0x7fffffff ? -INFINITY : hv;
            int bi = hi;
            ct_wave_argmax(bs, bi, wst);
            if (lane == 0) {
                ds[p] = bs;
                di[p] = bi == 0x7fffffff ? -1 : bi;
            }
            if (bi != 0x7fffffff && hi == bi) {
                ++ptr;
                if (ptr < nc) {
                    hv = st_s[wave][my][ptr];
                    hi = st_i[wave][my][ptr];
                    if (hi < 0) hi = 0x7fffffff;
                } else {
                    hi = 0x7fffffff;
                }
            }
        }
        cur ^= 1;
        have = true;
        __builtin_amdgcn_wave_barrier();
    }
    if (!have && lane < nc) {       // a wave without lists
        run_s[wave][cur][lane] = -INFINITY;
        run_i[wave][cur][lane] = -1;
    }
    __shared__ int cur_of[4];
    if (lane == 0) cur_of[wave] = cur;
    __syncthreads();
    if (wave == 0) {
        // the four waves' lists -> the final candidates
        int ptr = 0;
        float hv = -INFINITY;
        int hi = 0x7fffffff;
        const float* ps = nullptr;
        const int* pi = nullptr;
        if (lane < 4) {
            ps = run_s[lane][cur_of[lane]];
            pi = run_i[lane][cur_of[lane]];
            hv = ps[0];
            hi = pi[0];
            if (hi < 0) hi = 0x7fffffff;
        }
        for (int p = 0; p < nc; ++p) {
            float bs = hi == 0x7fffffff ? -INFINITY : hv;
            int bi = hi;
            ct_wave_argmax(bs, bi, wst);
            if (lane == 0) {
                fin_s[p] = bs;
                cand_i[p] = bi == 0x7fffffff ? -1 : bi;
            }
            if (bi != 0x7fffffff && hi == bi) {
                ++ptr;
                if (ptr < nc) {
                    hv = ps[ptr];
                    hi = pi[ptr];
                    if (hi < 0) hi = 0x7fffffff;
                } else {
                    hi = 0x7fffffff;
                }
            }
        }
    }
    __syncthreads();
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
        int n_ok = 0;
        for (int a = 0; a < nc; ++a) n_ok += cand_i[a] >= 0;
        for (int a = 1; a < nc; ++a) {
            const double s = cand_s[a];
            const int id = cand_i[a];
            int b = a - 1;
            while (b >= 0 && (cand_s[b] < s || (cand_s[b] == s && (unsigned)cand_i[b] > (unsigned)id))) {
                cand_s[b + 1] = cand_s[b];
                cand_i[b + 1] = cand_i[b];
                --b;
            }
            cand_s[b + 1] = s;
            cand_i[b + 1] = id;
        }
        for (int a = 0; a < k; ++a) {
            out_idx[(int64_t)qi * k + a] = a < n_ok ? (int64_t)cand_i[a] : (int64_t)-1;
            out_score[(int64_t)qi * k + a] = a < n_ok ? (float)cand_s[a] : -INFINITY;
        }
    }
}


// ---- few queries (Q <= 4: finds_sim, IKE_VL, LTE_VL retrieval), ONE launch -------------------------------------------------------
// Phase A: every workgroup (8 waves) scores 64 corpus rows (a wave: 8 rows with all their loads in flight) for all Q queries, fp32,
// the corpus norm either from the caller's cache (corpus_inv_norm) or accumulated in the same pass, and publishes the scores.
// Phase B: the workgroup whose arrival on a counter came LAST selects per query the nc = k + 8 best fp32 scores, re-scores those
// candidates in fp64 and emits them by (score desc, id asc): the same contract as the tiled path.  Selection on the order-preserving
// integer image of the scores, held in LDS:
//   * the nc-th largest of the 64 group maxima (a group = the keys of 8 lanes) is a LOWER BOUND T0 of the nc-th largest key: at least
//     nc keys (those maxima) reach it, so {key >= T0} contains the nc best; it is typically 1.1-2.5 nc keys;
//   * if that set fits the candidate buffer (256) it is ranked by (key desc, id asc) and the first nc are kept;
//   * else (thousands of equal scores: degenerate corpora) the exact nc-th key is found by bisection over the 32 key bits with
//     workgroup-wide counts, and the ties at that key are admitted by ascending id (bisection over the id).
// (A 256-bin LDS histogram radix select was measured first: the top byte of a cosine score takes 2-4 values, so its 15000 LDS
// atomics serialise on a handful of addresses -- 55-70 us per query.)
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility: hand-offs measured with sc1 accesses in place of release / acquire,
// first row): the scores are written with agent-scope (sc1) stores, every storing wave drains them (s_waitcnt vmcnt(0)), the
// workgroup's barrier, ONE lane's agent-scope atomic add on ONE counter; the workgroup whose add returned the last ticket reads the
// scores with agent-scope (sc1) loads only, after a barrier behind that add.  The last workgroup resets the counter for the next call.
#define CT_COUNTERS 64
#define CT_FB 512             // threads of a cosine_fused workgroup (8 waves: 256 VGPRs per lane, no spills; 1024 threads spilled)
#define CT_FW (CT_FB / 64)
#define CT_NPT 40             // keys per thread in phase B: N <= 20480 (10 x 16-byte loads)
#define CT_CCAP 256           // candidate buffer of the fast selection
__device__ unsigned g_ct_counter[CT_COUNTERS];
#ifdef CT_TIMING      /* tools/debug/cosine_phase_timing.sh: wall-clock stamps of the last workgroup's thread 0 (debug builds only) */
#define CT_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(scores + (((int64_t)Q * (N + 4) + 63) / 64 + 1) * 64)[i] = wall_clock64(); } while (0)
#else
#define CT_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned ct_key(float f) {       // larger float <-> larger unsigned; -0 < +0 is harmless (fp64 re-score decides)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// workgroup-wide sum of a per-thread count (two barriers; every thread gets the total)
__device__ __forceinline__ int ct_block_sum(int c, ct_wr_storage& wst, int* s_part, int lane, int wave) {
    int cs;
    ct_wr_i().reduce(c, cs, wst.i, rocprim::plus<int>());
    __syncthreads();
    if (lane == 0) s_part[wave] = cs;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < CT_FW; ++w) tot += s_part[w];
    return tot;
}

template <int QMAX>
__global__ __launch_bounds__(CT_FB) void cosine_fused_kernel(const float* __restrict__ corpus, const float* __restrict__ inv_cached,
                                                             const float* __restrict__ queries, int N, int Q, int D, int norm_c, int norm_q,
                                                             int k, int nc, float* scores, unsigned* counter,
                                                             int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    extern __shared__ unsigned lkeys[];                    // phase B: the keys of one query (dynamic LDS, 4 * roundup(N, 4) bytes)
    __shared__ ct_wr_storage wr_st[CT_FW];
    __shared__ unsigned s_gmax[64];
    __shared__ unsigned long long s_c64[CT_CCAP + 4];
    __shared__ unsigned s_T;
    __shared__ int s_last, s_n, s_part[CT_FW];
    __shared__ int cand_i[CT_MAXK + CT_MARGIN];
    __shared__ double cand_s[CT_MAXK + CT_MARGIN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ct_wr_storage& wst = wr_st[wave];
    const int Np = (N + 3) & ~3;           // row stride of the score buffer: 16-byte aligned rows for phase B's loads
    // ---------------- phase A ----------------
    CT_STAMP(0);
    {
        constexpr int RG = 8;
        const int row0 = blockIdx.x * (CT_FW * RG) + wave * RG;
        const int nrows = min(RG, N - row0);
        if (nrows > 0) {
            const int nv = D >> 2;
            float nrm[RG], acc[RG][QMAX];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                nrm[g] = 0.f;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) acc[g][q] = 0.f;
            }
            const bool need_norm = norm_c && inv_cached == nullptr;
            constexpr int CH = 1;                       // 16-byte pieces per row and lane in flight
            for (int c0 = lane; c0 < nv; c0 += 64 * CH) {
                float4 v[RG][CH];
#pragma unroll
                for (int g = 0; g < RG; ++g)
#pragma unroll
                    for (int h = 0; h < CH; ++h)
                        v[g][h] = c0 + 64 * h < nv ? reinterpret_cast<const float4*>(corpus + (int64_t)(row0 + min(g, nrows - 1)) * D)[c0 + 64 * h]
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 u[QMAX][CH];
#pragma unroll
                for (int q = 0; q < QMAX; ++q)
#pragma unroll
                    for (int h = 0; h < CH; ++h)
                        u[q][h] = (q < Q && c0 + 64 * h < nv) ? reinterpret_cast<const float4*>(queries + (int64_t)q * D)[c0 + 64 * h]
                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int g = 0; g < RG; ++g)
#pragma unroll
                    for (int h = 0; h < CH; ++h) {
                        const float4 x = v[g][h];
                        // explicit fma chains: every row goes through the SAME instruction sequence whatever slot g it sits in, so identical
                        // rows get identical fp32 scores (ties are then broken by id, as the contract says)
                        if (need_norm) nrm[g] = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, nrm[g]))));
#pragma unroll
                        for (int q = 0; q < QMAX; ++q)
                            acc[g][q] = fmaf(x.x, u[q][h].x, fmaf(x.y, u[q][h].y, fmaf(x.z, u[q][h].z, fmaf(x.w, u[q][h].w, acc[g][q]))));
                    }
            }
            float mine[QMAX];
#pragma unroll
            for (int q = 0; q < QMAX; ++q) mine[q] = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                float ic = 1.f;
                if (need_norm) {
                    float n2;
                    ct_wr_f().reduce(nrm[g], n2, wst.f, rocprim::plus<float>());
                    ic = n2 > 0.f ? 1.f / sqrtf(n2) : 0.f;
                } else if (norm_c) {
                    ic = inv_cached[row0 + min(g, nrows - 1)];
                }
#pragma unroll
                for (int q = 0; q < QMAX; ++q)
                    if (q < Q) {
                        float sc;
                        ct_wr_f().reduce(acc[g][q], sc, wst.f, rocprim::plus<float>());
                        if (lane == g) mine[q] = sc * ic;
                    }
            }
            if (lane < nrows) {
#pragma unroll
                for (int q = 0; q < QMAX; ++q)
                    if (q < Q) __hip_atomic_store(scores + (int64_t)q * Np + row0 + lane, mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores ...
        __syncthreads();                                        // ... before the ONE lane that signals for the workgroup
        CT_STAMP(1);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == gridDim.x - 1) ? 1 : 0;
            if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
        }
        __syncthreads();                                        // every wave of the last workgroup loads behind the add's return
        if (!s_last) return;
    }
    CT_STAMP(2);
    // ---------------- phase B: the last workgroup ----------------
    for (int qi = 0; qi < Q; ++qi) {
        // keys -> LDS: 10 x 16-byte agent-scope (sc1) loads per thread, all in flight (thread t, piece j: ids 4 (512 j + t) .. + 3); every
        // later step walks the LDS copy in ROLLED loops -- this code runs once per launch on one workgroup, straight from a cold
        // instruction cache: a fully unrolled register-resident version of it measured slower
        unsigned mx = 0u;
        {
            const float* srow = scores + (int64_t)qi * Np;
            const float* pp[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) pp[j] = srow + 4 * min(j * CT_FB + tid, (Np >> 2) - 1);
            float4_t r0, r1, r2, r3, r4, r5, r6, r7, r8, r9;
            asm volatile("global_load_dwordx4 %0, %10, off sc1\n\tglobal_load_dwordx4 %1, %11, off sc1\n\tglobal_load_dwordx4 %2, %12, off sc1\n\t"
                         "global_load_dwordx4 %3, %13, off sc1\n\tglobal_load_dwordx4 %4, %14, off sc1\n\tglobal_load_dwordx4 %5, %15, off sc1\n\t"
                         "global_load_dwordx4 %6, %16, off sc1\n\tglobal_load_dwordx4 %7, %17, off sc1\n\tglobal_load_dwordx4 %8, %18, off sc1\n\t"
                         "global_load_dwordx4 %9, %19, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9)
                         : "v"(pp[0]), "v"(pp[1]), "v"(pp[2]), "v"(pp[3]), "v"(pp[4]), "v"(pp[5]), "v"(pp[6]), "v"(pp[7]), "v"(pp[8]), "v"(pp[9])
                         : "memory");
            const float4_t rr[10] = {r0, r1, r2, r3, r4, r5, r6, r7, r8, r9};
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int i4 = j * CT_FB + tid;
                if (4 * i4 < Np) {
                    uint4 kv;
                    kv.x = 4 * i4 + 0 < N ? ct_key(rr[j][0]) : 0u;
                    kv.y = 4 * i4 + 1 < N ? ct_key(rr[j][1]) : 0u;
                    kv.z = 4 * i4 + 2 < N ? ct_key(rr[j][2]) : 0u;
                    kv.w = 4 * i4 + 3 < N ? ct_key(rr[j][3]) : 0u;
                    reinterpret_cast<uint4*>(lkeys)[i4] = kv;
                    mx = max(max(mx, max(kv.x, kv.y)), max(kv.z, kv.w));
                }
            }
        }
        CT_STAMP(3);
        // 64 group maxima (8 waves x 8 groups of 8 lanes) -> T0 = the nc-th largest of them (nc <= 40 <= 64), found by wave 0 in registers
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
        if ((lane & 7) == 0) s_gmax[wave * 8 + (lane >> 3)] = mx;
        if (tid == 0) s_n = 0;
        if (tid < CT_CCAP + 4) s_c64[tid] = 0ull;
        __syncthreads();                                        // (also publishes lkeys)
        if (wave == 0) {
            const unsigned g = s_gmax[lane];
            int rank = 0;
            for (int j = 0; j < 64; ++j) {
                const unsigned o = s_gmax[j];
                rank += (o > g || (o == g && j < lane)) ? 1 : 0;
            }
            if (rank == nc - 1) s_T = g;
        }
        __syncthreads();
        const unsigned T0 = s_T;
        // candidates {key >= T0} straight into the buffer (a 64-bit image: key, then the complemented id, so that ONE comparison orders
        // by key desc, id asc); the count decides afterwards whether the buffer held them all
        for (int id = tid; id < N; id += CT_FB) {
            const unsigned key = lkeys[id];
            if (key >= T0) {
                const int p = atomicAdd(&s_n, 1);
                if (p < CT_CCAP) s_c64[p] = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - id);
            }
        }
        __syncthreads();
        const int C = s_n;
        CT_STAMP(4);
        if (C <= CT_CCAP) {
            // fast path: rank the candidates, keep the first nc (entries past C are 0: never above a real one)
            if (tid < C) {
                const unsigned long long mine = s_c64[tid];
                int rank = 0;
                for (int j = 0; j < C; j += 4)
                    rank += (s_c64[j] > mine ? 1 : 0) + (s_c64[j + 1] > mine ? 1 : 0) + (s_c64[j + 2] > mine ? 1 : 0) + (s_c64[j + 3] > mine ? 1 : 0);
                if (rank < nc) cand_i[rank] = 0x7fffffff - (int)(unsigned)(mine & 0xffffffffull);
            }
        } else {
            // exact path: the nc-th largest key by bisection over its bits, ties at it by ascending id
            __syncthreads();
            if (tid == 0) s_n = 0;
            unsigned T = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned cand = T | (1u << bit);
                int cc = 0;
                for (int id = tid; id < N; id += CT_FB) cc += lkeys[id] >= cand ? 1 : 0;
                if (ct_block_sum(cc, wst, s_part, lane, wave) >= nc) T = cand;      // largest T with count(key >= T) >= nc
            }
            int cg = 0, ce = 0;
            for (int id = tid; id < N; id += CT_FB) {
                const unsigned key = lkeys[id];
                cg += key > T ? 1 : 0;
                ce += key == T ? 1 : 0;
            }
            const int n_gt = ct_block_sum(cg, wst, s_part, lane, wave);
            const int n_eq = ct_block_sum(ce, wst, s_part, lane, wave);
            const int need = nc - n_gt;                                            // 1 <= need <= n_eq
            int id_cap = N - 1;
            if (n_eq > need) {
                int lo = 0, hi = N - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    int cc = 0;
                    for (int id = tid; id <= mid; id += CT_FB) cc += lkeys[id] == T ? 1 : 0;
                    if (ct_block_sum(cc, wst, s_part, lane, wave) >= need) hi = mid; else lo = mid + 1;
                }
                id_cap = lo;
            }
            for (int id = tid; id < N; id += CT_FB) {
                const unsigned key = lkeys[id];
                if (key > T || (key == T && id <= id_cap)) cand_i[atomicAdd(&s_n, 1)] = id;
            }
        }
        __syncthreads();
        CT_STAMP(5);
        // fp64 re-score of the nc candidates, rank by (score desc, id asc), emit the first k.  One wave per candidate; the row and the
        // query come in 16-byte pieces, FOUR pieces per lane issued before the first use (a scalar loop over D was a chain of D / 64
        // dependent memory round trips: ~5 us per candidate round)
        const float4* qv4 = reinterpret_cast<const float4*>(queries + (int64_t)qi * D);
        const int nv = D >> 2;
        constexpr int PC = 3;                // candidates of a wave in flight (8 waves x 3 per round)
        for (int p0 = wave; p0 < nc; p0 += CT_FW * PC) {
            double dot[PC], cn[PC], qn = 0.0;
            const float4* cv4[PC];
#pragma unroll
            for (int u = 0; u < PC; ++u) {
                dot[u] = 0.0;
                cn[u] = 0.0;
                cv4[u] = reinterpret_cast<const float4*>(corpus + (int64_t)cand_i[min(p0 + u * CT_FW, nc - 1)] * D);
            }
            for (int c0 = lane; c0 < nv; c0 += 128) {
                float4 xv[PC][2], yv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cc = c0 + 64 * h;
                    yv[h] = cc < nv ? qv4[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < PC; ++u) xv[u][h] = cc < nv ? cv4[u][cc] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double y0 = yv[h].x, y1 = yv[h].y, y2 = yv[h].z, y3 = yv[h].w;
                    qn += (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3);
#pragma unroll
                    for (int u = 0; u < PC; ++u) {
                        const double x0 = xv[u][h].x, x1 = xv[u][h].y, x2 = xv[u][h].z, x3 = xv[u][h].w;
                        dot[u] += (x0 * y0 + x1 * y1) + (x2 * y2 + x3 * y3);
                        cn[u] += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
                    }
                }
            }
            qn = wave_sum_d(qn);
#pragma unroll
            for (int u = 0; u < PC; ++u) {
                const int p = p0 + u * CT_FW;
                if (p >= nc) break;
                const double d_ = wave_sum_d(dot[u]), c_ = wave_sum_d(cn[u]);
                if (lane == 0) {
                    double sc = d_;
                    if (norm_c) sc = c_ > 0.0 ? sc / sqrt(c_) : 0.0;
                    if (norm_q) sc = qn > 0.0 ? sc / sqrt(qn) : 0.0;
                    cand_s[p] = sc;
                }
            }
        }
        __syncthreads();
        CT_STAMP(6);
        if (tid < nc) {
            const double sc = cand_s[tid];
            const int id = cand_i[tid];
            int rank = 0;
            for (int b = 0; b < nc; ++b) rank += (cand_s[b] > sc || (cand_s[b] == sc && cand_i[b] < id)) ? 1 : 0;
            if (rank < k) {
                out_idx[(int64_t)qi * k + rank] = (int64_t)id;
                out_score[(int64_t)qi * k + rank] = (float)sc;
            }
        }
        CT_STAMP(7);
        for (int a = nc + tid; a < k; a += CT_FB) {             // a corpus smaller than k: pad (as the tiled path does)
            out_idx[(int64_t)qi * k + a] = (int64_t)-1;
            out_score[(int64_t)qi * k + a] = -INFINITY;
        }
        __syncthreads();
    }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t devqa_cosine_topk_workspace(int N, int Q, int k) {
    (void)k;
    if (N <= 0 || Q <= 0) return 256;
    // few-query path: [Q][ceil(N / 64)][64] candidate scores live in the score region (<= Q * (N + 63) floats), the ids behind it
    const int64_t ids = Q <= 4 ? align256((int64_t)Q * ((N + 63) / 64) * 64 * 4) : 0;
    return align256((int64_t)Q * (N + 64) * 4) + align256((int64_t)N * 4) + align256((int64_t)Q * 4) + ids;
}

static int cosine_topk_impl(const float* corpus, const float* corpus_inv_norm, const float* queries, int N, int Q, int D, int k,
                            int normalize_corpus, int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream) {
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
    {   // few queries: two launches (score + per-wave selection, merge + fp64 re-score); DEVQA_COSINE_FEWQ=0 keeps the tiled path
        static const int fewq = getenv("DEVQA_COSINE_FEWQ") ? atoi(getenv("DEVQA_COSINE_FEWQ")) : 1;
        const int nc = min(k + CT_MARGIN, N);
        const int n_lists = (N + 63) / 64;                                  // one sorted candidate list of <= 64 entries per wave
        if (fewq == 1 && Q <= 4 && N <= CT_FB * CT_NPT) {      // ONE launch: scores + last-workgroup selection (cosine_fused_kernel)
            static std::atomic<unsigned> ticket{0};
            unsigned* counter = nullptr;
            if (hipGetSymbolAddress((void**)&counter, HIP_SYMBOL(g_ct_counter)) != hipSuccess)
                return devqa_fail(DEVQA_E_HIP, "cosine_topk: counter symbol");
            counter += ticket.fetch_add(1) % CT_COUNTERS;        // calls in flight on different streams do not share a counter
            const dim3 grid((N + CT_FW * 8 - 1) / (CT_FW * 8));
            const float* inv = normalize_corpus ? corpus_inv_norm : nullptr;
            const size_t lds = (size_t)((N + 3) & ~3) * 4;       // the keys of one query (<= 80 KiB)
            static std::atomic<unsigned> a1{0}, a2{0}, a4{0};
            if (Q == 1) {
                devqa_set_max_smem(cosine_fused_kernel<1>, 96 * 1024, a1);
                hipLaunchKernelGGL(cosine_fused_kernel<1>, grid, dim3(CT_FB), lds, st, corpus, inv, queries, N, Q, D, normalize_corpus, normalize_queries,
                                   k, nc, scores, counter, out_idx, out_score);
            } else if (Q == 2) {
                devqa_set_max_smem(cosine_fused_kernel<2>, 96 * 1024, a2);
                hipLaunchKernelGGL(cosine_fused_kernel<2>, grid, dim3(CT_FB), lds, st, corpus, inv, queries, N, Q, D, normalize_corpus, normalize_queries,
                                   k, nc, scores, counter, out_idx, out_score);
            } else {
                devqa_set_max_smem(cosine_fused_kernel<4>, 96 * 1024, a4);
                hipLaunchKernelGGL(cosine_fused_kernel<4>, grid, dim3(CT_FB), lds, st, corpus, inv, queries, N, Q, D, normalize_corpus, normalize_queries,
                                   k, nc, scores, counter, out_idx, out_score);
            }
            devqa_prof_end(ph, 4.0 * (double)N * D, st);
            DEVQA_LAUNCH_CHECK("cosine_fused");
            return DEVQA_OK;
        }
        if (fewq && Q <= 4) {
            float* ls = scores;                                             // [Q][n_lists][64] scores, then ids: inside the Q * N floats
            int* li = (int*)(ws + align256((int64_t)Q * (N + 64) * 4));     // ids: behind the (padded) score region
            hipLaunchKernelGGL(score_select_kernel<4>, dim3((n_lists + 3) / 4), dim3(256), 0, st, corpus, queries, N, Q, D, normalize_corpus,
                               nc, n_lists, ls, li);
            DEVQA_LAUNCH_CHECK("score_select");
            hipLaunchKernelGGL(merge_final_kernel, dim3(Q), dim3(256), 0, st, ls, li, n_lists, corpus, queries, N, D, k, nc,
                               normalize_corpus, normalize_queries, out_idx, out_score);
            devqa_prof_end(ph, 4.0 * (double)N * D, st);
            DEVQA_LAUNCH_CHECK("merge_final");
            return DEVQA_OK;
        }
    }
    if (normalize_corpus && corpus_inv_norm) {
        inv_c = const_cast<float*>(corpus_inv_norm);        // the caller's cache (computed once at load, as the reference normalises once)
    } else if (normalize_corpus) {
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

extern "C" int devqa_cosine_topk(const float* corpus, const float* queries, int N, int Q, int D, int k, int normalize_corpus,
                                 int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream) {
    return cosine_topk_impl(corpus, nullptr, queries, N, Q, D, k, normalize_corpus, normalize_queries, out_idx, out_score, workspace, stream);
}

// The same search with the corpus' inverse row norms supplied by the caller (devqa_row_inv_norm, computed ONCE when the corpus is loaded --
// the reference normalises its stored embeddings once at load, R/dataset/vllm.py:104,117): the per-call pass over the corpus then
// accumulates dot products only.  corpus_inv_norm == NULL is devqa_cosine_topk(normalize_corpus = 1).
extern "C" int devqa_cosine_topk_cached(const float* corpus, const float* corpus_inv_norm, const float* queries, int N, int Q, int D, int k,
                                        int normalize_queries, int64_t* out_idx, float* out_score, void* workspace, void* stream) {
    return cosine_topk_impl(corpus, corpus_inv_norm, queries, N, Q, D, k, 1, normalize_queries, out_idx, out_score, workspace, stream);
}

extern "C" int devqa_row_inv_norm(const float* rows, int R, int D, float* out, void* stream) {
    DEVQA_CHECK_ARG(rows && out, "row_inv_norm: null pointer");
    DEVQA_CHECK_SHAPE(R > 0 && D > 0 && D % 4 == 0, "row_inv_norm: bad dims R=%d D=%d", R, D);
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, R, D, out);
    DEVQA_LAUNCH_CHECK("row_inv_norm");
    return DEVQA_OK;
}

#ifdef CT_TIMING
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCT_TIMING cosine_topk.hip -o build/ct_timing   (stand-alone phase timing of cosine_fused_kernel)
#include <cstdarg>
#include <cstdio>
#include <vector>
// stand-alone: the three library hooks this file uses (linking libdevqa_hip.so would interpose ITS copy of the kernels)
int devqa_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
int devqa_prof_begin(int, hipStream_t) { return -1; }
void devqa_prof_end(int, double, hipStream_t) {}
int main() {
    const int N = 15000, D = 384;
    std::vector<float> hc((size_t)N * D), hq(4 * D);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hc) v = rnd();
    for (auto& v : hq) v = rnd();
    float *dc, *dq, *dsc;
    int64_t* di;
    void* ws;
    hipMalloc(&dc, hc.size() * 4); hipMalloc(&dq, hq.size() * 4); hipMalloc(&dsc, 4 * 32 * 4); hipMalloc(&di, 4 * 32 * 8);
    hipMalloc(&ws, devqa_cosine_topk_workspace(N, 4, 32));
    hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    for (int k : {5, 32}) {
        for (int rep = 0; rep < 4; ++rep) {
            const int rc = devqa_cosine_topk(dc, dq, N, 1, D, k, 1, 1, di, dsc, ws, nullptr);
            const hipError_t e = hipDeviceSynchronize();
            if (rc != 0 || e != hipSuccess) printf("rc %d hip %s\n", rc, hipGetErrorString(e));
        }
        unsigned long long st[16];
        hipMemcpy(st, (float*)ws + (((int64_t)1 * (N + 4) + 63) / 64 + 1) * 64, sizeof(st), hipMemcpyDeviceToHost);
        int64_t hi[4]; hipMemcpy(hi, di, sizeof(hi), hipMemcpyDeviceToHost);
        printf("idx %ld %ld %ld  stamp0 %llu\n", (long)hi[0], (long)hi[1], (long)hi[2], st[0]);
        printf("k=%d (100 MHz ticks -> us): phaseA %.2f  arrive->B %.2f  keys %.2f  T0+count %.2f  candidates %.2f  rescore %.2f  emit %.2f  (last block total %.2f)\n", k,
               (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0, (st[4] - st[3]) / 100.0, (st[5] - st[4]) / 100.0,
               (st[6] - st[5]) / 100.0, (st[7] - st[6]) / 100.0, (st[7] - st[0]) / 100.0);
    }
    return 0;
}
#endif
