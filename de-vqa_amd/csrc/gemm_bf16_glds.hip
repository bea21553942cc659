// bf16 TN GEMM, LDS-DMA staged (global_load_lds_dwordx4): the fast path of devqa_gemm_bf16 for
// K % 64 == 0.  Same math, operand layout, swizzle and epilogue as gemm_bf16.hip; what changes is the
// global->LDS path: tiles are written straight into LDS by LDS-DMA (no VGPR round trip, no ds_write --
// the ds_write_b128 path moves only ~79 B/clk/CU and was the bottleneck of the register-staged kernel).
//
// LDS-DMA writes lane-linear (wave-uniform base + lane*16 B), so the XOR swizzle of the 16-byte chunk
// index is applied on the SOURCE side: the lane that owns LDS slot (row, pc) fetches global chunk
// c = pc ^ ((row>>1)&7) of that row; fragment reads apply the same XOR (conflict-free ds_read_b128).
// Out-of-range rows are clamped to the last valid row (their products land in rows/cols that the
// epilogue never stores), so the K loop has no branches.
//
// Two LDS stages; tile kt+1 is in flight while tile kt is multiplied; one barrier per K step.
#include "common.h"

#define GBK 64

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_glds_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, int M, int N, int K, float alpha, int act, const float* residual,
    bf16_t* out_bf16, float* out_f32, int64_t ldc, int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int SLOTS = (BM + BN) / 8;  // 1-KiB wave-instructions per tile pair (8 rows of 128 B each)
    constexpr int IPW = SLOTS / NW;
    static_assert(SLOTS % NW == 0, "tile rows must split evenly over the waves");
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid % tiles_m;
    const int tile_n = bid / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    // per-lane source pointers of this wave's IPW slots (row clamp + source-side swizzle), advanced by GBK per step
    const bf16_t* src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int s = wave * IPW + i;
        const int row = s * 8 + (lane >> 3);  // row in the concatenated [A rows | W rows] tile
        const int pc = lane & 7;
        if (s * 8 < BM) {
            const int c = pc ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, M - 1);
            src[i] = A + (int64_t)gm * lda + c * 8;
        } else {
            const int rb = row - BM;
            const int c = pc ^ ((rb >> 1) & 7);
            const int gn = min(n0 + rb, N - 1);
            src[i] = W + (int64_t)gn * ldw + c * 8;
        }
    }
    auto issue = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int s = wave * IPW + i;
            __builtin_amdgcn_global_load_lds((gptr_t)src[i], (lptr_t)(smem + buf * STAGE_BYTES + s * 1024), 16, 0, 0);
            src[i] += GBK;
        }
    };

    float4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / GBK;
    issue(0);
    __syncthreads();  // (hipcc drains the LDS-DMA with vmcnt(0) ahead of the barrier)

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1);
        const unsigned char* sa = smem + cur * STAGE_BYTES;
        const unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < GBK / 32; ++ks) {
            short8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 16 + fr;
                const int c = ks * 4 + fq;
                af[i] = *reinterpret_cast<const short8_t*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 16 + fr;
                const int c = ks * 4 + fq;
                bfr[j] = *reinterpret_cast<const short8_t*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D^T: rows = n
        }
        __syncthreads();
    }

    // ---- epilogue.  Operands were fed swapped (W fragment as the MFMA A operand), so a lane's accumulator
    // holds, for output row m = ..+fr, the 4 CONSECUTIVE columns n = ..+4*fq+0..3: 16-byte fp32 / 8-byte bf16
    // accesses per lane instead of four scalar ones.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
            if (n >= N) continue;  // N % 4 == 0: a 4-column group is entirely in or out
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
            v = devqa_act4(v, act);
            const int64_t o = (int64_t)m * ldc + n;
            if (residual) {
                const float4 r = *reinterpret_cast<const float4*>(residual + o);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (out_f32) *reinterpret_cast<float4*>(out_f32 + o) = v;
            if (out_bf16) {
                uint2 p;
                p.x = pack_bf16x2(v.x, v.y);
                p.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2*>(out_bf16 + o) = p;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_glds(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K,
                       float alpha, int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc,
                       hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t smem = 2 * (BM + BN) * 128;
    auto kern = gemm_bf16_glds_kernel<BM, BN, WM, WN>;
    static std::atomic<unsigned> attr_done{0};
    devqa_set_max_smem(kern, smem, attr_done);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), smem, st, A, lda, W, ldw, bias, M, N, K, alpha, act,
                       residual, out_bf16, out_f32, ldc, tiles_m, tiles_n);
    DEVQA_LAUNCH_CHECK("gemm_bf16_glds");
    return DEVQA_OK;
}

// variant ids for the profiling hook (gemm_bf16.hip): 2 = 128x128
int launch_gemm_glds_128x128(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K,
                             float alpha, int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc,
                             hipStream_t st) {
    return launch_glds<128, 128, 2, 2>(A, lda, W, ldw, bias, M, N, K, alpha, act, residual, out_bf16, out_f32, ldc, st);
}
int launch_gemm_glds_64x128(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K,
                            float alpha, int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc,
                            hipStream_t st) {
    return launch_glds<64, 128, 2, 2>(A, lda, W, ldw, bias, M, N, K, alpha, act, residual, out_bf16, out_f32, ldc, st);
}
