// fp32 TN GEMM on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, 1/16 of the
// bf16 rate): the "faithful" compute mode used to pin the HIP path to the reference's fp32 results
// at 1e-3.  Same contract and epilogue as devqa_gemm_bf16.
//
// 64x64 tile per workgroup (4 waves x 16 rows x 64 cols), K streamed through LDS in 32-float chunks;
// LDS row stride 34 floats makes the ds_read_b32 fragment reads (16 rows x 2 k per 32-lane half)
// hit 32 distinct banks.  Global loads are float4; next chunk is prefetched into registers.
#include "common.h"

float* devqa_splitk_workspace(hipStream_t st, size_t bytes);      // gemm_bf16.hip
void devqa_launch_splitk_reduce_epilogue(const float* ws, int splits, int M, int N, const float* bias, float alpha, int act,
                                         const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc, hipStream_t st);

#define F_BK 32
#define F_LD 34

__global__ __launch_bounds__(256) void gemm_f32_tn_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ W, int64_t ldw,
                                                          const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                          int act, const float* residual, float* out_f32, int64_t ldc,
                                                          int tiles_m, int k_per_split) {
    __shared__ float As[64 * F_LD];
    __shared__ float Bs[64 * F_LD];
    const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
    const int m0 = tile_m * 64, n0 = tile_n * 64;
    // split-K (k_per_split > 0): blockIdx.y owns K range [kb, ke) and writes a raw fp32 partial slab [M][ldc]; the reduce pass sums the
    // slabs in slice order (deterministic) and applies the epilogue
    const int kb = k_per_split > 0 ? (int)blockIdx.y * k_per_split : 0;
    const int ke = k_per_split > 0 ? min(K, kb + k_per_split) : K;
    if (k_per_split > 0) out_f32 += (int64_t)blockIdx.y * M * ldc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    float4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // each thread stages 2 float4 per operand per chunk: rows (tid>>3) and (tid>>3)+32, cols (tid&7)*4
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = lr + h * 32;
            const int kk = k0 + lc;
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
            if (kk < ke) {
                if (m0 + r < M) av = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + r) * lda + kk);
                if (n0 + r < N) bv = *reinterpret_cast<const float4*>(W + (int64_t)(n0 + r) * ldw + kk);
            }
            ra[h] = av;
            rb[h] = bv;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* ad = As + (lr + h * 32) * F_LD + lc;
            float* bd = Bs + (lr + h * 32) * F_LD + lc;
            ad[0] = ra[h].x; ad[1] = ra[h].y; ad[2] = ra[h].z; ad[3] = ra[h].w;
            bd[0] = rb[h].x; bd[1] = rb[h].y; bd[2] = rb[h].z; bd[3] = rb[h].w;
        }
    };
    gload(kb);
    for (int k0 = kb; k0 < ke; k0 += F_BK) {
        __syncthreads();  // previous chunk consumed
        lstore();
        __syncthreads();
        if (k0 + F_BK < ke) gload(k0 + F_BK);
#pragma unroll
        for (int ks = 0; ks < F_BK / 4; ++ks) {
            const float af = As[(wave * 16 + fr) * F_LD + ks * 4 + fq];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bf = Bs[(j * 16 + fr) * F_LD + ks * 4 + fq];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0);
            }
        }
    }
    // C/D map: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + fr;
        if (n >= N) continue;
        const float b = (bias && k_per_split == 0) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + fq * 4 + r;
            if (m >= M) continue;
            const int64_t o = (int64_t)m * ldc + n;
            if (k_per_split > 0) {
                out_f32[o] = acc[j][r];
                continue;
            }
            float v = (acc[j][r] + b) * alpha;
            v = devqa_act(v, act);
            if (residual) v += residual[o];
            out_f32[o] = v;
        }
    }
}

extern "C" int devqa_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int M, int N,
                              int K, float alpha, int act, const float* residual, float* out_f32, int64_t ldc,
                              void* stream) {
    DEVQA_CHECK_ARG(A && W && out_f32, "gemm_f32: null pointer");
    DEVQA_CHECK_ARG(act >= 0 && act <= 3, "gemm_f32: bad act %d", act);
    if (M == 0 || N == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "gemm_f32: bad dims %d %d %d", M, N, K);
    DEVQA_CHECK_SHAPE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm_f32: K/lda/ldw must be multiples of 4");
    DEVQA_CHECK_SHAPE(lda >= K && ldw >= K && ldc >= N, "gemm_f32: leading dims too small");
    DEVQA_CHECK_SHAPE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0, "gemm_f32: operands must be 16-byte aligned");
    const int tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    // few rows x a long K (the MEND hyper-network: [n <= 64 x 12800] . [1920 x 12800]^T streams 98 MB of fp32 weights through
    // 30 workgroups): split K over enough workgroups to fill the chip; slabs are summed in slice order, so the result is
    // deterministic (it is no longer ONE k-ordered fmaf chain: partial chains of k_per_split terms, then their ordered sum)
    const int tiles = tiles_m * tiles_n;
    if (M <= 128 && tiles < 128 && K >= 2048 && N % 4 == 0 && ldc % 4 == 0 &&
        ((((uintptr_t)out_f32) | ((uintptr_t)residual) | ((uintptr_t)bias)) & 15) == 0) {
        int splits = 512 / tiles;
        if (splits > K / 512) splits = K / 512;
        if (splits >= 2) {
            int kper = ((K + splits - 1) / splits + F_BK - 1) / F_BK * F_BK;
            const int used = (K + kper - 1) / kper;
            float* ws = devqa_splitk_workspace(st, (size_t)used * M * N * sizeof(float));
            if (ws) {
                hipLaunchKernelGGL(gemm_f32_tn_kernel, dim3(tiles, used), dim3(256), 0, st, A, lda, W, ldw, (const float*)nullptr, M, N, K,
                                   1.0f, 0, (const float*)nullptr, ws, (int64_t)N, tiles_m, kper);
                DEVQA_LAUNCH_CHECK("gemm_f32_splitk");
                devqa_launch_splitk_reduce_epilogue(ws, used, M, N, bias, alpha, act, residual, nullptr, out_f32, ldc, st);
                DEVQA_LAUNCH_CHECK("gemm_f32_splitk_reduce");
                return DEVQA_OK;
            }
        }
    }
    hipLaunchKernelGGL(gemm_f32_tn_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, W, ldw, bias,
                       M, N, K, alpha, act, residual, out_f32, ldc, tiles_m, 0);
    DEVQA_LAUNCH_CHECK("gemm_f32");
    return DEVQA_OK;
}
