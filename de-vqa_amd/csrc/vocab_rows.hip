// Rows over the vocabulary: argmax, NLL of a label, and d(NLL)/d(logits) in one kernel.
// One workgroup (1024 threads) per logits row; the row (fp32, V ~ 50k = 200 KB) is read once for
// the online (max, sum-exp, argmax) reduction and once more only when dlogits is requested
// (second read is L2-resident).  Reductions use wavefront shuffles + one LDS hop across 16 waves.
#include "common.h"

struct RedT {
    float m;   // running max
    float s;   // sum exp(x - m)
    int idx;   // index of first max
};
__device__ __forceinline__ RedT red_combine(RedT a, RedT b) {
    RedT r;
    if (b.m > a.m || (b.m == a.m && b.idx < a.idx)) {
        r.m = b.m;
        r.idx = b.idx;
    } else {
        r.m = a.m;
        r.idx = a.idx;
    }
    const float sa = (a.m == -INFINITY) ? 0.f : a.s * __expf(a.m - r.m);
    const float sb = (b.m == -INFINITY) ? 0.f : b.s * __expf(b.m - r.m);
    r.s = sa + sb;
    return r;
}

template <typename T>
__global__ __launch_bounds__(1024) void vocab_rows_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                          const int32_t* __restrict__ labels, const float* __restrict__ coef,
                                                          int32_t* __restrict__ argmax_out, float* __restrict__ nll_out,
                                                          T* __restrict__ dlogits, int64_t ldd) {
    __shared__ float sm_m[16], sm_s[16];
    __shared__ int sm_i[16];
    __shared__ float bc_m, bc_lse;
    const int r = blockIdx.x;
    const float* row = logits + (int64_t)r * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RedT acc = {-INFINITY, 0.f, 0x7fffffff};
    const int nv = V >> 2;
    for (int i = tid; i < nv; i += 1024) {
        const float4 x = reinterpret_cast<const float4*>(row)[i];
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            RedT e = {xs[t], 1.f, i * 4 + t};
            if (!(xs[t] == xs[t])) e.m = INFINITY;  // NaN compares as the maximum like torch.argmax
            acc = red_combine(acc, e);
        }
    }
    for (int i = nv * 4 + tid; i < V; i += 1024) {
        RedT e = {row[i], 1.f, i};
        acc = red_combine(acc, e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        RedT b;
        b.m = __shfl_xor(acc.m, o, 64);
        b.s = __shfl_xor(acc.s, o, 64);
        b.idx = __shfl_xor(acc.idx, o, 64);
        acc = red_combine(acc, b);
    }
    if (lane == 0) {
        sm_m[wave] = acc.m;
        sm_s[wave] = acc.s;
        sm_i[wave] = acc.idx;
    }
    __syncthreads();
    if (wave == 0) {
        RedT b = {-INFINITY, 0.f, 0x7fffffff};
        if (lane < 16) {
            b.m = sm_m[lane];
            b.s = sm_s[lane];
            b.idx = sm_i[lane];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            RedT c;
            c.m = __shfl_xor(b.m, o, 64);
            c.s = __shfl_xor(b.s, o, 64);
            c.idx = __shfl_xor(b.idx, o, 64);
            b = red_combine(b, c);
        }
        if (lane == 0) {
            const float lse = b.m + __logf(b.s);
            if (argmax_out) argmax_out[r] = b.idx;
            if (nll_out && labels) nll_out[r] = lse - row[labels[r]];
            bc_m = b.m;
            bc_lse = lse;
        }
    }
    if (!dlogits) return;
    __syncthreads();
    const float lse = bc_lse;
    const float cf = coef ? coef[r] : 1.f;
    const int lab = labels ? labels[r] : -1;
    T* drow = dlogits + (int64_t)r * ldd;
    for (int i = tid; i < nv; i += 1024) {
        const float4 x = reinterpret_cast<const float4*>(row)[i];
        float g[4] = {__expf(x.x - lse), __expf(x.y - lse), __expf(x.z - lse), __expf(x.w - lse)};
        const int b0 = i * 4;
        if (lab >= b0 && lab < b0 + 4) g[lab - b0] -= 1.f;
        if constexpr (sizeof(T) == 2) {
            uint2 p;
            p.x = pack_bf16x2(g[0] * cf, g[1] * cf);
            p.y = pack_bf16x2(g[2] * cf, g[3] * cf);
            reinterpret_cast<uint2*>(drow)[i] = p;
        } else {
            reinterpret_cast<float4*>(drow)[i] = make_float4(g[0] * cf, g[1] * cf, g[2] * cf, g[3] * cf);
        }
    }
    for (int i = nv * 4 + tid; i < V; i += 1024) {
        float g = __expf(row[i] - lse);
        if (i == lab) g -= 1.f;
        if constexpr (sizeof(T) == 2) drow[i] = f32_to_bf16(g * cf);
        else drow[i] = g * cf;
    }
}

template <typename T>
static int launch_vocab_rows(const float* logits, int64_t ldl, int R, int V, const int32_t* labels, const float* coef,
                             int32_t* argmax_out, float* nll_out, T* dlogits, int64_t ldd, void* stream) {
    DEVQA_CHECK_ARG(logits, "vocab_rows: null logits");
    DEVQA_CHECK_ARG(argmax_out || nll_out || dlogits, "vocab_rows: nothing to compute");
    DEVQA_CHECK_ARG(!(nll_out || dlogits) || labels, "vocab_rows: labels required for nll/dlogits");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && V > 0 && ldl >= V && ldl % 4 == 0, "vocab_rows: bad shape R=%d V=%d", R, V);
    DEVQA_CHECK_SHAPE(!dlogits || (ldd >= V && ldd % 4 == 0), "vocab_rows: bad dlogits stride");
    hipLaunchKernelGGL(vocab_rows_kernel<T>, dim3(R), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, labels, coef,
                       argmax_out, nll_out, dlogits, ldd);
    DEVQA_LAUNCH_CHECK("vocab_rows");
    return DEVQA_OK;
}
extern "C" int devqa_vocab_rows(const float* logits, int64_t ldl, int R, int V, const int32_t* labels, const float* coef,
                                int32_t* argmax_out, float* nll_out, devqa_bf16* dlogits, int64_t ldd, void* stream) {
    return launch_vocab_rows<bf16_t>(logits, ldl, R, V, labels, coef, argmax_out, nll_out, dlogits, ldd, stream);
}
extern "C" int devqa_vocab_rows_f32(const float* logits, int64_t ldl, int R, int V, const int32_t* labels, const float* coef,
                                    int32_t* argmax_out, float* nll_out, float* dlogits, int64_t ldd, void* stream) {
    return launch_vocab_rows<float>(logits, ldl, R, V, labels, coef, argmax_out, nll_out, dlogits, ldd, stream);
}
