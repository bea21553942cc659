// bf16 TN GEMM with an N-stage LDS-DMA ring (generalisation of gemm_bf16_glds.hip):
//   template <BM, BN, WM, WN, BKT, NST>: block tile BM x BN, WM x WN waves, K step BKT (32 or 64),
//   NST LDS stages.  Tile kt+NST-1 is issued at the top of iteration kt into the stage freed by the
//   barrier of iteration kt-1; the end-of-iteration wait is COUNTED (vmcnt((NST-2)*IPW)): tile kt+1 has
//   landed, the NST-2 younger tiles stay in flight across the raw s_barrier.
// LDS image per stage: [BM + BN rows][BKT] bf16, 16-byte chunks XOR-swizzled on the SOURCE address
// (LDS-DMA writes lane-linear): BKT=64: pc = c ^ ((row>>1)&7); BKT=32: pc = c ^ ((-(row>>2))&3).  Both make
// every 16-lane service group of a ds_read_b128 fragment read hit 16 distinct 16-byte slots.
#include "common.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BKT>
__device__ __forceinline__ int swz(int row, int c) {
    if constexpr (BKT == 64) return c ^ ((row >> 1) & 7);
    else return c ^ ((0 - (row >> 2)) & 3);
}

template <int BM, int BN, int WM, int WN, int BKT, int NST, int PRIO = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_pipe_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, int M, int N, int K, float alpha, int act, const float* residual,
    bf16_t* out_bf16, float* out_f32, int64_t ldc, int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int RB = BKT * 2;          // bytes per LDS row
    constexpr int CPR = BKT / 8;         // 16-byte chunks per row
    constexpr int RPI = 1024 / RB;       // rows per 1-KiB LDS-DMA wave-instruction
    constexpr int SLOTS = (BM + BN) / RPI;
    constexpr int IPW = SLOTS / NW;
    constexpr int KS = BKT / 32;
    static_assert(SLOTS % NW == 0 && BM % RPI == 0, "tile rows must split evenly over the waves");
    static_assert(NST >= 2 && NST <= 4, "2..4 stages");
    constexpr int STAGE_BYTES = (BM + BN) * RB;

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

    const bf16_t* src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int s = wave * IPW + i;
        const int row = s * RPI + lane / CPR;  // row in the concatenated [A rows | W rows] tile
        const int pc = lane % CPR;
        if (s * RPI < BM) {
            src[i] = A + (int64_t)min(m0 + row, M - 1) * lda + swz<BKT>(row, pc) * 8;
        } else {
            const int rb = row - BM;
            src[i] = W + (int64_t)min(n0 + rb, N - 1) * ldw + swz<BKT>(rb, pc) * 8;
        }
    }
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int s = wave * IPW + i;
            __builtin_amdgcn_global_load_lds((gptr_t)src[i], (lptr_t)(smem + stage * STAGE_BYTES + s * 1024), 16, 0, 0);
            src[i] += BKT;
        }
    };

    float4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BKT;
    // prologue: tiles 0 .. NST-2 in flight, tile 0 landed
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t);
    {
        const int younger = min(nk - 1, NST - 2);  // tiles issued after tile 0
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        {   // refill the stage that was read in iteration kt-1 (every wave has passed that iteration's barrier)
            const int st_new = stage == 0 ? NST - 1 : stage - 1;  // (stage + NST - 1) % NST
            if (kt + NST - 1 < nk) issue(st_new);
        }
        const unsigned char* sa = smem + stage * STAGE_BYTES;
        const unsigned char* sb = sa + BM * RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            short8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 16 + fr;
                af[i] = *reinterpret_cast<const short8_t*>(sa + row * RB + (swz<BKT>(row, ks * 4 + fq) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const short8_t*>(sb + row * RB + (swz<BKT>(row, ks * 4 + fq) << 4));
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        {   // tile kt+1 must have landed; tiles kt+2 .. kt+NST-1 (those that exist) may stay in flight
            const int younger = min(nk - 1, kt + NST - 1) - (kt + 1);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage = stage == NST - 1 ? 0 : stage + 1;
    }

    // epilogue (operands fed swapped: a lane holds 4 consecutive columns of one output row)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
            if (n >= N) continue;
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

template <int BM, int BN, int WM, int WN, int BKT, int NST, int PRIO = 0>
static int launch_pipe(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K,
                       float alpha, int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc,
                       hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t smem = (size_t)NST * (BM + BN) * BKT * 2;
    auto kern = gemm_bf16_pipe_kernel<BM, BN, WM, WN, BKT, NST, PRIO>;
    static std::atomic<unsigned> attr_done{0};
    devqa_set_max_smem(kern, smem, attr_done);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), smem, st, A, lda, W, ldw, bias, M, N, K, alpha, act,
                       residual, out_bf16, out_f32, ldc, tiles_m, tiles_n);
    DEVQA_LAUNCH_CHECK("gemm_bf16_pipe");
    return DEVQA_OK;
}

// experimental variants selected by devqa_gemm_set_mode(10 + id)
int launch_gemm_pipe(int id, const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K,
                     float alpha, int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc, hipStream_t st) {
#define ARGS A, lda, W, ldw, bias, M, N, K, alpha, act, residual, out_bf16, out_f32, ldc, st
    switch (id) {
        case 0: return launch_pipe<128, 128, 2, 2, 32, 3>(ARGS);   // 48 KiB: 3 workgroups/CU
        case 1: return launch_pipe<128, 128, 2, 2, 32, 4>(ARGS);   // 64 KiB: 2 workgroups/CU
        case 2: return launch_pipe<256, 256, 2, 4, 32, 4>(ARGS);   // 128 KiB: 1 workgroup/CU, 128x64 wave tiles
        case 3: return launch_pipe<256, 128, 4, 2, 32, 4>(ARGS);   // 96 KiB
        case 4: return launch_pipe<256, 128, 4, 2, 64, 3>(ARGS);   // 144 KiB
        case 5: return launch_pipe<128, 128, 2, 2, 64, 2>(ARGS);   // == the 2-stage kernel, raw barrier
        case 6: return launch_pipe<256, 256, 2, 4, 64, 2>(ARGS);   // 128 KiB, simple 2-stage on 256x256
        case 7: return launch_pipe<256, 256, 2, 4, 64, 2, 1>(ARGS);  // as 6 with s_setprio around the MFMA block
        case 8: return launch_pipe<256, 256, 4, 2, 64, 2>(ARGS);     // 64x128 wave tiles
        case 9: return launch_pipe<256, 256, 2, 4, 32, 3>(ARGS);     // 96 KiB ring of 32-deep steps
    }
#undef ARGS
    return devqa_fail(DEVQA_E_ARG, "gemm_pipe: unknown variant %d", id);
}
