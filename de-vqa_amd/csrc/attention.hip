// Packed varlen softmax attention (ViT self-attn, Q-Former self/cross-attn, OPT causal attn with an
// optional shared image-token prefix).  See include/devqa.h for the sequence descriptor.
//
// v1 structure (correctness-first, VALU):  one workgroup = one (sequence, head, 16-query tile);
// 4 waves x 4 query rows.  Keys/values stream through LDS in 64-key chunks (bf16 -> fp32, row
// stride dh+1 so that lane=key reads are bank-conflict free); scores use lane = key, the PV
// product uses lane = output channel; online softmax in fp32 (running max / sum per query row).
#include <stdlib.h>
#include "common.h"

#define ATT_QT 16     // query rows per workgroup
#define ATT_KC 64     // keys per chunk
#define ATT_MAXDH 128

__device__ __forceinline__ void load8(const bf16_t* p, float* o) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        o[2 * t] = bf16_to_f32(w[t] & 0xffff);
        o[2 * t + 1] = bf16_to_f32(w[t] >> 16);
    }
}
__device__ __forceinline__ void load8(const float* p, float* o) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, int64_t ldq,
                                                        const T* __restrict__ k, int64_t ldk,
                                                        const T* __restrict__ v, int64_t ldv,
                                                        T* __restrict__ out, int64_t ldo,
                                                        const int32_t* __restrict__ seq_desc, int H, int dh, float scale,
                                                        int causal, int q_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int dhp = dh + 1;
    float* Ks = smf;                      // [ATT_KC][dhp]
    float* Vs = Ks + ATT_KC * dhp;        // [ATT_KC][dhp]
    float* Qs = Vs + ATT_KC * dhp;        // [ATT_QT][dh]   (pre-scaled)
    float* Ps = Qs + ATT_QT * dh;         // [ATT_QT][ATT_KC]

    const int bid = blockIdx.x;
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % H;
    const int s = bid / (q_tiles * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], kp_start = d[2], kp_len = d[3], ko_start = d[4], ko_len = d[5];
    const int q0 = qt * ATT_QT;
    if (q0 >= q_len) return;  // uniform per block
    const int nq = min(ATT_QT, q_len - q0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // stage the query tile (scaled)
    for (int i = tid; i < ATT_QT * dh; i += 256) {
        const int r = i / dh, c = i % dh;
        float val = 0.f;
        if (r < nq) val = ld1(q + (int64_t)(q_start + q0 + r) * ldq + h * dh + c) * scale;
        Qs[r * dh + c] = val;
    }

    // keys visible to this tile: prefix keys [0,kp_len) then own keys [0, own_hi)
    const int causal_off = ko_len - q_len;  // query i sees own keys 0..i+causal_off
    const int own_hi = causal ? min(ko_len, q0 + nq - 1 + causal_off + 1) : ko_len;
    const int n_keys = kp_len + (own_hi > 0 ? own_hi : 0);

    float m_run[4], l_run[4], o0[4], o1[4];  // rows wave*4+j ; o0: channel lane, o1: channel lane+64
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m_run[j] = -INFINITY;
        l_run[j] = 0.f;
        o0[j] = 0.f;
        o1[j] = 0.f;
    }

    for (int c0 = 0; c0 < n_keys; c0 += ATT_KC) {
        const int nk = min(ATT_KC, n_keys - c0);
        __syncthreads();  // previous chunk fully consumed (also covers the Qs staging on the first pass)
        // ---- stage K/V chunk: 8 bf16 per access ----
        const int vec_per_row = dh >> 3;
        for (int i = tid; i < ATT_KC * vec_per_row; i += 256) {
            const int r = i / vec_per_row, cv = i % vec_per_row;
            float kv[8], vv[8];
            if (r < nk) {
                const int kidx = c0 + r;
                const int64_t grow = (kidx < kp_len) ? (int64_t)(kp_start + kidx) : (int64_t)(ko_start + kidx - kp_len);
                load8(k + grow * ldk + h * dh + cv * 8, kv);
                load8(v + grow * ldv + h * dh + cv * 8, vv);
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    kv[t] = 0.f;
                    vv[t] = 0.f;
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                Ks[r * dhp + cv * 8 + t] = kv[t];
                Vs[r * dhp + cv * 8 + t] = vv[t];
            }
        }
        __syncthreads();

        // ---- scores: lane = key; 4 query rows per wave ----
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* krow = Ks + lane * dhp;
        const float* qb = Qs + (wave * 4) * dh;
        for (int c = 0; c < dh; ++c) {
            const float kvv = krow[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] += qb[j * dh + c] * kvv;
        }
        const int kidx = c0 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qi = q0 + wave * 4 + j;  // index within the sequence
            bool ok = (lane < nk) && (wave * 4 + j < nq);
            if (ok && causal && kidx >= kp_len) ok = (kidx - kp_len) <= qi + causal_off;
            const float sv = ok ? sc[j] : -INFINITY;
            const float cm = wave_max(sv);
            const float m_new = fmaxf(m_run[j], cm);
            float p = 0.f, alpha = 1.f;
            if (m_new != -INFINITY) {
                p = ok ? __expf(sv - m_new) : 0.f;
                alpha = (m_run[j] == -INFINITY) ? 0.f : __expf(m_run[j] - m_new);
            }
            const float ps = wave_sum(p);
            l_run[j] = l_run[j] * alpha + ps;
            o0[j] *= alpha;
            o1[j] *= alpha;
            m_run[j] = m_new;
            Ps[(wave * 4 + j) * ATT_KC + lane] = p;
        }
        // Ps rows of this wave are written and read by this wave only; make the LDS writes visible
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- PV: lane = channel (lane, lane+64) ----
        const float* pb = Ps + (wave * 4) * ATT_KC;
        const bool c0ok = lane < dh, c1ok = lane + 64 < dh;
        for (int kk = 0; kk < nk; ++kk) {
            const float v0 = c0ok ? Vs[kk * dhp + lane] : 0.f;
            const float v1 = c1ok ? Vs[kk * dhp + lane + 64] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = pb[j * ATT_KC + kk];
                o0[j] += p * v0;
                o1[j] += p * v1;
            }
        }
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 4 + j;
        if (r < nq) {
            const float inv = l_run[j] > 0.f ? 1.f / l_run[j] : 0.f;
            T* orow = out + (int64_t)(q_start + q0 + r) * ldo + h * dh;
            if (lane < dh) st1(orow + lane, o0[j] * inv);
            if (lane + 64 < dh) st1(orow + lane + 64, o1[j] * inv);
        }
    }
}

static inline bool h_aligned(int H, int dh) { (void)H; return (dh * 2) % 16 == 0; }

template <typename T>
static int launch_attention(const T* q, int64_t ldq, const T* k, int64_t ldk, const T* v, int64_t ldv, T* out, int64_t ldo,
                            const int32_t* seq_desc, int n_seq, int max_q_len, int H, int dh, float scale, int causal,
                            void* stream) {
    DEVQA_CHECK_ARG(q && k && v && out && seq_desc, "attention: null pointer");
    causal &= 1;      // bit 2 (the caller's self-attention promise, include/devqa.h) is a hint for the MFMA kernels only
    if (n_seq == 0 || max_q_len == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(n_seq > 0 && max_q_len > 0 && H > 0, "attention: bad dims");
    DEVQA_CHECK_SHAPE(dh % 8 == 0 && dh > 0 && dh <= ATT_MAXDH, "attention: dh=%d unsupported", dh);
    DEVQA_CHECK_SHAPE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "attention: row strides must be multiples of 8");
    DEVQA_CHECK_SHAPE(ldq >= (int64_t)H * dh && ldk >= (int64_t)H * dh && ldv >= (int64_t)H * dh && ldo >= (int64_t)H * dh,
                      "attention: row strides smaller than H*dh");
    const int q_tiles = (max_q_len + ATT_QT - 1) / ATT_QT;
    const size_t smem = sizeof(float) * (2 * ATT_KC * (dh + 1) + ATT_QT * dh + ATT_QT * ATT_KC);
    const long grid = (long)n_seq * H * q_tiles;
    DEVQA_CHECK_SHAPE(grid < 2147483647L, "attention: grid too large");
    static std::atomic<unsigned> attr_done{0};
    devqa_set_max_smem(attention_kernel<T>, sizeof(float) * (2 * ATT_KC * (ATT_MAXDH + 1) + ATT_QT * ATT_MAXDH + ATT_QT * ATT_KC), attr_done);
    hipLaunchKernelGGL(attention_kernel<T>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, q, ldq, k, ldk, v,
                       ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles);
    DEVQA_LAUNCH_CHECK("attention");
    return DEVQA_OK;
}

int launch_attention_mfma(const bf16_t* q, int64_t ldq, const bf16_t* k, int64_t ldk, const bf16_t* v, int64_t ldv, bf16_t* out,
                          int64_t ldo, const int32_t* seq_desc, int n_seq, int max_q_len, int H, int dh, float scale,
                          int causal, void* stream);  // attention_mfma.hip

// bf16: MFMA flash attention (attention_mfma.hip).  DEVQA_ATTENTION_VALU=1 selects the v1 VALU kernel
// (kept as an in-library cross-check; same results up to fp reassociation).
extern "C" int devqa_attention(const devqa_bf16* q, int64_t ldq, const devqa_bf16* k, int64_t ldk, const devqa_bf16* v,
                               int64_t ldv, devqa_bf16* out, int64_t ldo, const int32_t* seq_desc, int n_seq,
                               int max_q_len, int H, int dh, float scale, int causal, void* stream) {
    static const bool use_valu = getenv("DEVQA_ATTENTION_VALU") != nullptr;
    if (use_valu)
        return launch_attention<bf16_t>(q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, n_seq, max_q_len, H, dh, scale, causal,
                                        stream);
    DEVQA_CHECK_ARG(q && k && v && out && seq_desc, "attention: null pointer");
    if (n_seq == 0 || max_q_len == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(n_seq > 0 && max_q_len > 0 && H > 0, "attention: bad dims");
    DEVQA_CHECK_SHAPE(dh % 8 == 0 && dh > 0 && dh <= ATT_MAXDH, "attention: dh=%d unsupported", dh);
    DEVQA_CHECK_SHAPE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "attention: row strides must be multiples of 8");
    DEVQA_CHECK_SHAPE(ldo % 4 == 0 && (((uintptr_t)out) & 7) == 0, "attention: out must be 8-byte aligned with ldo %% 4 == 0 (packed stores)");
    DEVQA_CHECK_SHAPE(ldq >= (int64_t)H * dh && ldk >= (int64_t)H * dh && ldv >= (int64_t)H * dh && ldo >= (int64_t)H * dh,
                      "attention: row strides smaller than H*dh");
    DEVQA_CHECK_SHAPE((((uintptr_t)q) & 15) == 0 && (((uintptr_t)k) & 15) == 0 && (((uintptr_t)v) & 15) == 0 && (h_aligned(H, dh)),
                      "attention: q/k/v must be 16-byte aligned");
    return launch_attention_mfma(q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, n_seq, max_q_len, H, dh, scale, causal, stream);
}
extern "C" int devqa_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                   float* out, int64_t ldo, const int32_t* seq_desc, int n_seq, int max_q_len, int H, int dh,
                                   float scale, int causal, void* stream) {
    return launch_attention<float>(q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, n_seq, max_q_len, H, dh, scale, causal,
                                   stream);
}
