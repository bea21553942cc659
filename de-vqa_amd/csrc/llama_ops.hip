// LLaMA-family row kernels for the LLaVA / Vicuna decoder (BASELINE config #3):
// RMSNorm forward / backward-dx, rotary position embedding, SwiGLU gate.  HBM-bound: 16-byte accesses,
// one wave per row for the reductions (wavefront shuffles, row held in registers).
#include "common.h"
#include <type_traits>

#define RN_MAXV 16

// y = x * rsqrt(mean(x^2) + eps) * w      (HF LlamaRMSNorm: statistics and scaling in fp32)
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                      const float* __restrict__ w, int M, int D, float eps,
                                                      bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* ar = add ? reinterpret_cast<const float4*>(add + (int64_t)row * D) : nullptr;
    float4 v[RN_MAXV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            float4 t = xr[c];
            if (ar) {
                const float4 u = ar[c];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[i] = t;
            q += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float r = rsqrtf(wave_sum(q) / (float)D + eps);
    const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float4 g = w4[c];
            float4 o;
            o.x = v[i].x * r * g.x; o.y = v[i].y * r * g.y; o.z = v[i].z * r * g.z; o.w = v[i].w * r * g.w;
            if (out_f32) reinterpret_cast<float4*>(out_f32 + (int64_t)row * D)[c] = o;
            if (out_bf16) {
                uint2 p;
                p.x = pack_bf16x2(o.x, o.y);
                p.y = pack_bf16x2(o.z, o.w);
                reinterpret_cast<uint2*>(out_bf16 + (int64_t)row * D)[c] = p;
            }
        }
    }
}

extern "C" int devqa_rmsnorm(const float* x, const float* add, const float* w, int M, int D, float eps, devqa_bf16* out_bf16,
                             float* out_f32, void* stream) {
    DEVQA_CHECK_ARG(x && w && (out_bf16 || out_f32), "rmsnorm: null pointer");
    if (M == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * RN_MAXV, "rmsnorm: D=%d unsupported", D);
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add, w, M, D, eps, out_bf16,
                       out_f32);
    DEVQA_LAUNCH_CHECK("rmsnorm");
    return DEVQA_OK;
}

// g = dy*w, r = rsqrt(mean(x^2)+eps):  dx = r*g - x * r^3 * mean(g*x)
__global__ __launch_bounds__(256) void rmsnorm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                             const float* __restrict__ w, const float* __restrict__ dy, int M,
                                                             int D, float eps, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* ar = add ? reinterpret_cast<const float4*>(add + (int64_t)row * D) : nullptr;
    const float4* dr = reinterpret_cast<const float4*>(dy + (int64_t)row * D);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4 v[RN_MAXV], g[RN_MAXV];
    float q = 0.f, s = 0.f;
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            float4 t = xr[c];
            if (ar) {
                const float4 u = ar[c];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            const float4 d = dr[c], ww = w4[c];
            v[i] = t;
            g[i] = make_float4(d.x * ww.x, d.y * ww.y, d.z * ww.z, d.w * ww.w);
            q += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
            s += (g[i].x * t.x + g[i].y * t.y) + (g[i].z * t.z + g[i].w * t.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = v[i];
        }
    }
    const float r = rsqrtf(wave_sum(q) / (float)D + eps);
    const float k = r * r * r * (wave_sum(s) / (float)D);
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            float4 o;
            o.x = r * g[i].x - v[i].x * k; o.y = r * g[i].y - v[i].y * k;
            o.z = r * g[i].z - v[i].z * k; o.w = r * g[i].w - v[i].w * k;
            reinterpret_cast<float4*>(dx + (int64_t)row * D)[c] = o;
        }
    }
}

extern "C" int devqa_rmsnorm_bwd_dx(const float* x, const float* add, const float* w, const float* dy, int M, int D, float eps,
                                    float* dx, void* stream) {
    DEVQA_CHECK_ARG(x && w && dy && dx, "rmsnorm_bwd_dx: null pointer");
    if (M == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * RN_MAXV, "rmsnorm_bwd_dx: D=%d unsupported", D);
    hipLaunchKernelGGL(rmsnorm_bwd_dx_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add, w, dy, M, D, eps, dx);
    DEVQA_LAUNCH_CHECK("rmsnorm_bwd_dx");
    return DEVQA_OK;
}

// ---- rotary embedding, HF "rotate_half" convention, applied in place to n_heads heads of head dim dh that start
// at column 0 of each row (q heads followed by k heads in the fused QKV buffer):
//   for j < dh/2:  (x[j], x[j+dh/2]) <- (x[j]*c - x[j+dh/2]*s,  x[j+dh/2]*c + x[j]*s),  angle = pos * theta^(-2j/dh)
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }

// One wave per row: lane -> (head slot = lane / half, j = lane % half); the angle depends on (pos[row], j) only, so sin / cos
// are computed ONCE per row and lane (full-precision sincosf) and reused for every head the lane visits.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ x, int64_t ld, int R, const int32_t* __restrict__ pos, int n_heads,
                                                   int dh, float log2_theta) {
    const int half = dh >> 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int hp = half >= 64 ? 1 : 64 / half;          // heads per pass (half <= 64: checked on the host)
    const int slot = lane / half, j = lane - slot * half;
    if (slot >= hp) return;
    const float inv_freq = exp2f(-log2_theta * (2.0f * (float)j / (float)dh));
    float sn, cs;
    sincosf((float)pos[row] * inv_freq, &sn, &cs);
    T* base = x + (int64_t)row * ld + j;
    for (int h = slot; h < n_heads; h += hp) {
        T* p = base + (int64_t)h * dh;
        const float a = ldf(p), b = ldf(p + half);
        stf(p, __builtin_fmaf(a, cs, -(b * sn)));
        stf(p + half, __builtin_fmaf(b, cs, a * sn));
    }
}

// The same rotation with 16-byte accesses (bf16; half % 8 == 0, i.e. dh 16 .. 128 in steps of 16): a lane owns 8 consecutive j of a head slot, so
// a pass moves 2 x 16 bytes per lane instead of 2 x 2 (the element-wise form above ran at 2 TB/s on the LLaMA decoders' [R, 64 heads x 128]
// q | k block: 340 us per layer and forward); sin / cos of the lane's 8 angles are computed once per row with the SAME expressions, the
// rotation is the same two fp32 expressions per element: bit-identical.  Two passes are loaded before the first store (a load behind a
// pending store costs the compiler a vmcnt(0) on gfx950).
__global__ __launch_bounds__(256) void rope_bf16x8_kernel(bf16_t* __restrict__ x, int64_t ld, int R, const int32_t* __restrict__ pos, int n_heads,
                                                          int dh, float log2_theta) {
    const int half = dh >> 1, lph = half >> 3;            // lanes per head
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int hp = 64 / lph;                              // heads per pass
    const int slot = lane / lph, j0 = (lane - slot * lph) * 8;
    if (slot >= hp) return;
    float sn[8], cs[8];
    const float p_ = (float)pos[row];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float inv_freq = exp2f(-log2_theta * (2.0f * (float)(j0 + u) / (float)dh));
        sincosf(p_ * inv_freq, &sn[u], &cs[u]);
    }
    bf16_t* base = x + (int64_t)row * ld + j0;
    auto rot = [&](const uint4& a4, const uint4& b4, uint4& oa, uint4& ob) {
        const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
        uint32_t ra[4], rb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a0 = __uint_as_float(aw[k] << 16), a1 = __uint_as_float(aw[k] & 0xffff0000u);
            const float b0 = __uint_as_float(bw[k] << 16), b1 = __uint_as_float(bw[k] & 0xffff0000u);
            // (the contraction the element-wise kernel compiles to: one rounded product, one fma -- spelled out so that both forms round alike)
            ra[k] = (uint32_t)f32_to_bf16(__builtin_fmaf(a0, cs[2 * k], -(b0 * sn[2 * k]))) |
                    ((uint32_t)f32_to_bf16(__builtin_fmaf(a1, cs[2 * k + 1], -(b1 * sn[2 * k + 1]))) << 16);
            rb[k] = (uint32_t)f32_to_bf16(__builtin_fmaf(b0, cs[2 * k], a0 * sn[2 * k])) |
                    ((uint32_t)f32_to_bf16(__builtin_fmaf(b1, cs[2 * k + 1], a1 * sn[2 * k + 1])) << 16);
        }
        oa = make_uint4(ra[0], ra[1], ra[2], ra[3]);
        ob = make_uint4(rb[0], rb[1], rb[2], rb[3]);
    };
    int h = slot;
    for (; h + hp < n_heads; h += 2 * hp) {
        bf16_t* p0 = base + (int64_t)h * dh;
        bf16_t* p1 = base + (int64_t)(h + hp) * dh;
        const uint4 a0 = *reinterpret_cast<const uint4*>(p0), b0 = *reinterpret_cast<const uint4*>(p0 + half);
        const uint4 a1 = *reinterpret_cast<const uint4*>(p1), b1 = *reinterpret_cast<const uint4*>(p1 + half);
        uint4 oa0, ob0, oa1, ob1;
        rot(a0, b0, oa0, ob0);
        rot(a1, b1, oa1, ob1);
        *reinterpret_cast<uint4*>(p0) = oa0;
        *reinterpret_cast<uint4*>(p0 + half) = ob0;
        *reinterpret_cast<uint4*>(p1) = oa1;
        *reinterpret_cast<uint4*>(p1 + half) = ob1;
    }
    if (h < n_heads) {
        bf16_t* p0 = base + (int64_t)h * dh;
        const uint4 a0 = *reinterpret_cast<const uint4*>(p0), b0 = *reinterpret_cast<const uint4*>(p0 + half);
        uint4 oa0, ob0;
        rot(a0, b0, oa0, ob0);
        *reinterpret_cast<uint4*>(p0) = oa0;
        *reinterpret_cast<uint4*>(p0 + half) = ob0;
    }
}

template <typename T>
static int launch_rope(T* x, int64_t ld, int R, const int32_t* pos, int n_heads, int dh, float theta, void* stream) {
    DEVQA_CHECK_ARG(x && pos, "rope: null pointer");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && n_heads > 0 && dh > 0 && dh % 2 == 0 && dh <= 128 && ld >= (int64_t)n_heads * dh && theta > 1.f,
                      "rope: bad dims (dh even, <= 128)");
    if constexpr (std::is_same<T, bf16_t>::value) {
        if ((dh & 31) == 0 && 64 % (dh >> 4) == 0 && (ld & 7) == 0 && (((uintptr_t)x) & 15) == 0) {       // half % 16 == 0: 16-byte pieces stay 16-byte aligned
            hipLaunchKernelGGL(rope_bf16x8_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, R, pos, n_heads, dh, log2f(theta));
            DEVQA_LAUNCH_CHECK("rope");
            return DEVQA_OK;
        }
    }
    hipLaunchKernelGGL(rope_kernel<T>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, R, pos, n_heads, dh, log2f(theta));
    DEVQA_LAUNCH_CHECK("rope");
    return DEVQA_OK;
}
extern "C" int devqa_rope_bf16(devqa_bf16* x, int64_t ld, int R, const int32_t* pos, int n_heads, int dh, float theta, void* stream) {
    return launch_rope<bf16_t>(x, ld, R, pos, n_heads, dh, theta, stream);
}
extern "C" int devqa_rope_f32(float* x, int64_t ld, int R, const int32_t* pos, int n_heads, int dh, float theta, void* stream) {
    return launch_rope<float>(x, ld, R, pos, n_heads, dh, theta, stream);
}

// ---- SwiGLU: out[r, j] = silu(gu[r, j]) * gu[r, F + j]  for the fused [gate | up] GEMM output [R, 2F]
template <typename T>
__global__ void swiglu_kernel(const T* __restrict__ gu, int R, int F, T* __restrict__ out) {
    const int64_t total = (int64_t)R * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % F);
        const int64_t r = i / F;
        const float g = ldf(gu + r * 2 * F + j), u = ldf(gu + r * 2 * F + F + j);
        stf(out + i, g / (1.f + __expf(-g)) * u);
    }
}

// bf16, F % 8 == 0: 16-byte accesses (8 gate + 8 up values in, 8 out per thread and step)
__global__ __launch_bounds__(256) void swiglu_bf16x8_kernel(const bf16_t* __restrict__ gu, int R, int F8, bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)R * F8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % F8);
        const int64_t r = i / F8;
        const uint4 g4 = reinterpret_cast<const uint4*>(gu + r * 16 * F8)[j];
        const uint4 u4 = reinterpret_cast<const uint4*>(gu + r * 16 * F8 + 8 * F8)[j];
        const uint32_t gw[4] = {g4.x, g4.y, g4.z, g4.w}, uw[4] = {u4.x, u4.y, u4.z, u4.w};
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g0 = __uint_as_float(gw[k] << 16), g1 = __uint_as_float(gw[k] & 0xffff0000u);
            const float u0 = __uint_as_float(uw[k] << 16), u1 = __uint_as_float(uw[k] & 0xffff0000u);
            ow[k] = pack_bf16x2(g0 / (1.f + __expf(-g0)) * u0, g1 / (1.f + __expf(-g1)) * u1);
        }
        reinterpret_cast<uint4*>(out + r * 8 * F8)[j] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

template <typename T>
static int launch_swiglu(const T* gu, int R, int F, T* out, void* stream) {
    DEVQA_CHECK_ARG(gu && out, "swiglu: null pointer");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && F > 0, "swiglu: bad dims");
    const int64_t total = (int64_t)R * F;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if constexpr (sizeof(T) == 2) {
        if (F % 8 == 0 && (((uintptr_t)gu | (uintptr_t)out) & 15) == 0) {
            const int64_t tot8 = (int64_t)R * (F / 8);
            const unsigned g8 = (unsigned)((tot8 + 255) / 256 < 65536 ? (tot8 + 255) / 256 : 65536);
            hipLaunchKernelGGL(swiglu_bf16x8_kernel, dim3(g8), dim3(256), 0, (hipStream_t)stream, gu, R, F / 8, out);
            DEVQA_LAUNCH_CHECK("swiglu");
            return DEVQA_OK;
        }
    }
    hipLaunchKernelGGL(swiglu_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gu, R, F, out);
    DEVQA_LAUNCH_CHECK("swiglu");
    return DEVQA_OK;
}
extern "C" int devqa_swiglu_bf16(const devqa_bf16* gu, int R, int F, devqa_bf16* out, void* stream) {
    return launch_swiglu<bf16_t>(gu, R, F, out, stream);
}
extern "C" int devqa_swiglu_f32(const float* gu, int R, int F, float* out, void* stream) {
    return launch_swiglu<float>(gu, R, F, out, stream);
}

// backward of a = silu(g) * u for the explicit backward of the MEND_VL edit path through LLaMA FFNs
// (R/editor/vllm_editors/mend_vl/mend_vl.py:177-186 reaches it through autograd):
//   dg = da * u * sig(g) * (1 + g * (1 - sig(g))),  du = da * silu(g);  gu [R, 2F] (gate | up) in T, da / dgu fp32
template <typename T>
__global__ void swiglu_bwd_kernel(const T* __restrict__ gu, const float* __restrict__ da, int R, int F, float* __restrict__ dgu) {
    const int64_t total = (int64_t)R * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % F);
        const int64_t r = i / F;
        const float g = ldf(gu + r * 2 * F + j), u = ldf(gu + r * 2 * F + F + j);
        const float sg = 1.f / (1.f + expf(-g));
        const float d = da[i];
        dgu[r * 2 * F + j] = d * u * sg * (1.f + g * (1.f - sg));
        dgu[r * 2 * F + F + j] = d * g * sg;
    }
}

template <typename T>
static int launch_swiglu_bwd(const T* gu, const float* da, int R, int F, float* dgu, void* stream) {
    DEVQA_CHECK_ARG(gu && da && dgu, "swiglu_bwd: null pointer");
    DEVQA_CHECK_SHAPE(R > 0 && F > 0, "swiglu_bwd: bad dims");
    const int64_t total = (int64_t)R * F;
    const unsigned grid = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
    hipLaunchKernelGGL(swiglu_bwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gu, da, R, F, dgu);
    DEVQA_LAUNCH_CHECK("swiglu_bwd");
    return DEVQA_OK;
}
extern "C" int devqa_swiglu_bwd_bf16(const devqa_bf16* gu, const float* da, int R, int F, float* dgu, void* stream) {
    return launch_swiglu_bwd<bf16_t>(gu, da, R, F, dgu, stream);
}
extern "C" int devqa_swiglu_bwd_f32(const float* gu, const float* da, int R, int F, float* dgu, void* stream) {
    return launch_swiglu_bwd<float>(gu, da, R, F, dgu, stream);
}
