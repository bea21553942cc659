// Shared device/host helpers for libdevqa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/devqa.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float float4_t;   // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float float16_t; // 32x32 MFMA accumulator

extern thread_local char g_devqa_err[512];
int devqa_fail(int code, const char* fmt, ...);

#define DEVQA_CHECK_ARG(cond, ...)                                   \
    do {                                                             \
        if (!(cond)) return devqa_fail(DEVQA_E_ARG, __VA_ARGS__);    \
    } while (0)
#define DEVQA_CHECK_SHAPE(cond, ...)                                 \
    do {                                                             \
        if (!(cond)) return devqa_fail(DEVQA_E_SHAPE, __VA_ARGS__);  \
    } while (0)
#define DEVQA_LAUNCH_CHECK(name)                                                                 \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return devqa_fail(DEVQA_E_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- measurement hook (profile.hip): HIP events around instrumented launches while devqa_profile(1) is in effect ----
#define DEVQA_PROF_GEMM0 0        /* 0..3: GEMM tile variants 32x128, 64x128, 128x128, 256x256; work = FLOPs */
#define DEVQA_PROF_ATTENTION 4    /* attention_mfma_kernel; work = FLOPs (4 Tq Tk dh per head, as launched) */
#define DEVQA_PROF_FT_ADAMW 5     /* ft_adamw_step_kernel; work = bytes if every edit updates (24 E Dout Din) */
#define DEVQA_PROF_COSINE 6       /* cosine top-k scan; work = corpus bytes */
#define DEVQA_PROF_LAYERNORM 7    /* layernorm_kernel; work = bytes */
#define DEVQA_PROF_SLOTS 8
int devqa_prof_begin(int slot, hipStream_t st);            // -> handle, or -1 when profiling is off / the pool is full
void devqa_prof_end(int handle, double work, hipStream_t st);

// dynamic-LDS opt-in above 48 KiB: once per (kernel instantiation, device), safe against concurrent host threads.
// `done` is a function-local static std::atomic<unsigned> bit mask over device ordinals.
#include <atomic>
template <typename K>
static inline void devqa_set_max_smem(K kern, size_t smem, std::atomic<unsigned>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    done.fetch_or(bit, std::memory_order_release);
}

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; NaN stays NaN (quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fused GEMM epilogue activations (DEVQA_ACT_*)
__device__ __forceinline__ float devqa_act(float v, int act) {
    if (act == DEVQA_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DEVQA_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (act == DEVQA_ACT_QUICK_GELU) return v / (1.f + __expf(-1.702f * v));  // x * sigmoid(1.702 x), CLIP
    return v;
}
__device__ __forceinline__ float4 devqa_act4(float4 v, int act) {
    if (act != DEVQA_ACT_NONE) {
        v.x = devqa_act(v.x, act); v.y = devqa_act(v.y, act); v.z = devqa_act(v.z, act); v.w = devqa_act(v.w, act);
    }
    return v;
}
