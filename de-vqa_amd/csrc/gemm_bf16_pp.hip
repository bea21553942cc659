// bf16 TN GEMM, 256x256x64 tiles, 8 waves (2 x 4, 128x64 per wave), two wave groups running half a phase apart
// ("ping-pong"): while waves 0-3 issue MFMAs, waves 4-7 (their SIMD partners) read LDS fragments and issue the
// LDS-DMA prefetch, and vice versa.  A K-tile is computed in 4 phases (one 64x32 quadrant of the wave tile x K=64,
// 16 MFMA each); every phase prefetches one 128-row half-tile (A0/A1/B0/B1, 16 KiB) of a later K-tile.
//
// LDS (128 KiB): 2 K-tile buffers x [A0 | A1 | B0 | B1], each [128 rows][64] bf16.  A-half h holds rows
// {wr*128 + h*64 + 0..63} of the block (wr = 0,1), B-half h columns {wc*64 + h*32 + 0..31} (wc = 0..3): the rows a
// quadrant needs are one half-tile.  16-byte chunks are XOR-swizzled on the SOURCE address (LDS-DMA writes
// lane-linear): pc = c ^ ((row >> 1) & 7), conflict-free for ds_read_b128.
//
// Schedule (g = 4t + p, tile t, phase p).  Phase p reads: p0 A0(t),B0(t); p1 B1(t); p2 A1(t); p3 nothing, and stages
// slot g: p0 B1(t+1), p1 A1(t+1), p2 A0(t+2), p3 B0(t+2).  After staging, `s_waitcnt vmcnt(8)`: all but the 4
// youngest half-tiles have landed, so a half-tile staged in phase s is readable from phase s+5 (RAW: the wait of
// phase s+4 by EVERY wave precedes a barrier the reader has passed, also for the group running one barrier late),
// and a region is restaged no earlier than 2 phases after its last ds_read (WAR under the stagger).  With two
// buffers this is the deepest prefetch that satisfies both (derivation in DESIGN.md "256x256 ping-pong GEMM").
#include "common.h"

#ifdef PP_TIMING   /* hipcc -DPP_TIMING gemm_bf16_pp.hip -o build/pp_timing: where a tile's time goes (debug builds only) */
#define PP_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(g.ws)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PP_STAMP_RT(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(g.ws)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#define PP_STAMP_RT(i) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_HALF = 16384;         // bytes of one half-tile
constexpr int PP_BUF = 4 * PP_HALF;    // bytes of one K-tile buffer
enum { R_A0 = 0, R_A1 = 1, R_B0 = 2, R_B1 = 3 };

// threadIdx.x behind an asm the optimiser cannot look through: inside the persistent form's tile loop every per-lane quantity (staging offsets,
// fragment addresses, epilogue rows) is then computed per tile instead of being hoisted out of the loop, where it would live across the K loop
// beside 128 accumulator registers (measured: 52-64 VGPRs spilled without this)
__device__ __forceinline__ int pp_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

struct PPState {
    const char* gA;            // uniform base pointers (K-tile 0)
    const char* gW;
    unsigned off[4][2];        // [region][piece] byte offset of this thread's 16 bytes from gA / gW (< 4 GiB, checked on the host)
    unsigned char* smem;
    int dst;                   // wave * 2048: this wave's two 1-KiB pieces inside a half-tile
    int la0, la1, lb0, lb1;    // per-lane fragment byte offsets (ks = 0/1) inside an A / B half-tile
};

template <int REGION>
__device__ __forceinline__ void pp_stage(const PPState& s, int t) {
    unsigned char* d = s.smem + (t & 1) * PP_BUF + REGION * PP_HALF + s.dst;
    const char* g = (REGION < 2 ? s.gA : s.gW) + (int64_t)t * (PP_BK * 2);   // uniform: SGPR base + 32-bit lane offset
    __builtin_amdgcn_global_load_lds((gptr_t)(g + (uint64_t)s.off[REGION][0]), (lptr_t)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(g + (uint64_t)s.off[REGION][1]), (lptr_t)(d + 1024), 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void pp_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void pp_mfma_block(float4_t (&acc)[8][4], const short8_t (&a)[4][2], const short8_t (&b)[2][2],
                                              const int i0, const int j0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[i0 + i][j0 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// TAIL: 0 = tiles t+1 and t+2 exist, 1 = only t+1 exists, 2 = last tile
// BAL (experimental): B0 of tile t+1 is read one phase early (phase 3 of tile t, which otherwise reads nothing) into b0n, so
// the per-phase ds_read counts are 8/4/8/4 instead of 12/4/8/0; B0 is then staged before A0 (slots 4t-6 / 4t-5).
template <int TAIL, bool BAL = false>
__device__ __forceinline__ void pp_tile(const PPState& s, int t, float4_t (&acc)[8][4], short8_t (&a)[4][2], short8_t (&b0)[2][2],
                                        short8_t (&b1)[2][2], short8_t (&b0n)[2][2]) {
    const unsigned char* base = s.smem + (t & 1) * PP_BUF;
    // ---- phase 0: quadrant (a0, b0)
    if constexpr (BAL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { b0[j][0] = b0n[j][0]; b0[j][1] = b0n[j][1]; }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b0[j][0] = *reinterpret_cast<const short8_t*>(base + R_B0 * PP_HALF + j * 2048 + s.lb0);
            b0[j][1] = *reinterpret_cast<const short8_t*>(base + R_B0 * PP_HALF + j * 2048 + s.lb1);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + R_A0 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + R_A0 * PP_HALF + i * 2048 + s.la1);
    }
    if constexpr (TAIL <= 1) { pp_stage<R_B1>(s, t + 1); pp_vmcnt<8>(); } else { pp_vmcnt<2>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 0, 0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: quadrant (a0, b1)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b1[j][0] = *reinterpret_cast<const short8_t*>(base + R_B1 * PP_HALF + j * 2048 + s.lb0);
        b1[j][1] = *reinterpret_cast<const short8_t*>(base + R_B1 * PP_HALF + j * 2048 + s.lb1);
    }
    if constexpr (TAIL <= 1) { pp_stage<R_A1>(s, t + 1); pp_vmcnt<8>(); } else { pp_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b1, 0, 2);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: quadrant (a1, b1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + R_A1 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + R_A1 * PP_HALF + i * 2048 + s.la1);
    }
    if constexpr (TAIL == 0) { pp_stage<BAL ? R_B0 : R_A0>(s, t + 2); pp_vmcnt<8>(); } else if constexpr (TAIL == 1) { pp_vmcnt<6>(); } else { pp_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b1, 4, 2);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: quadrant (a1, b0)
    if constexpr (BAL && TAIL <= 1) {
        const unsigned char* nb = s.smem + ((t + 1) & 1) * PP_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b0n[j][0] = *reinterpret_cast<const short8_t*>(nb + R_B0 * PP_HALF + j * 2048 + s.lb0);
            b0n[j][1] = *reinterpret_cast<const short8_t*>(nb + R_B0 * PP_HALF + j * 2048 + s.lb1);
        }
    }
    if constexpr (TAIL == 0) { pp_stage<BAL ? R_A0 : R_B0>(s, t + 2); pp_vmcnt<8>(); } else if constexpr (TAIL == 1) { pp_vmcnt<4>(); } else { pp_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 4, 0);
    __builtin_amdgcn_s_barrier();
}


// K-tile 0 of a tile whose four half-tiles were staged BEFORE the previous tile's epilogue (persistent form) and are known to have landed: the phases
// stage as always (slots 0-3), but phases 0-2 do not wait -- the epilogue's stores are still in flight, they count in vmcnt like the loads, and a wait
// for "all but the 8 youngest" would wait for them.  Nothing read in this K-tile depends on a wait; phase 3 has the usual one (what K-tile 1's
// phase 0 reads -- A0, B0 of K-tile 1, staged just before this K-tile -- is older than the 8 youngest loads by then).
__device__ __forceinline__ void pp_tile_pre0(const PPState& s, float4_t (&acc)[8][4], short8_t (&a)[4][2], short8_t (&b0)[2][2], short8_t (&b1)[2][2]) {
    const unsigned char* base = s.smem;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b0[j][0] = *reinterpret_cast<const short8_t*>(base + R_B0 * PP_HALF + j * 2048 + s.lb0);
        b0[j][1] = *reinterpret_cast<const short8_t*>(base + R_B0 * PP_HALF + j * 2048 + s.lb1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + R_A0 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + R_A0 * PP_HALF + i * 2048 + s.la1);
    }
    pp_stage<R_B1>(s, 1);
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 0, 0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b1[j][0] = *reinterpret_cast<const short8_t*>(base + R_B1 * PP_HALF + j * 2048 + s.lb0);
        b1[j][1] = *reinterpret_cast<const short8_t*>(base + R_B1 * PP_HALF + j * 2048 + s.lb1);
    }
    pp_stage<R_A1>(s, 1);
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b1, 0, 2);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + R_A1 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + R_A1 * PP_HALF + i * 2048 + s.la1);
    }
    pp_stage<R_A0>(s, 2);
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b1, 4, 2);
    __builtin_amdgcn_s_barrier();
    pp_stage<R_B0>(s, 2);
    pp_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 4, 0);
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// HALF-WIDTH COLUMN TILE (the last column tile when N - n0 <= 128: N = 1408 is 5.5 tiles, 4224 is 16.5).  The same 2 x 4 wave
// layout with 128 x 32 wave tiles: only the (a0, b0) and (a1, b0) quadrants exist, so a K-tile is TWO phases of 16 MFMAs and three
// half-tiles of LDS-DMA ([A0 | A1 | B0], 48 KiB; B0 row r = column n0 + r).  Three K-tile buffers (144 KiB): tile t + 2 is staged
// during tile t -- A0 and B0 in phase 0, A1 in phase 1 (each region was last read two phases earlier by the slowest wave: the WAR
// distance of the full schedule) -- and read four phases after it was staged; the wait of the phase before the read allows the
// younger loads of three phases in flight: phase 0 `vmcnt(10)` (A1(t) has landed behind 4 + 2 + 4 younger loads), phase 1 `vmcnt(8)`
// (A0(t+1), B0(t+1) behind 2 + 4 + 2).  Tails: the tile before the last stages nothing (6 / 2), the last waits for everything.
// Per K-tile the LDS reads are 12 + 8 fragments per wave against 12 + 4 + 8 + 0 of a full tile: 0.56 of its time, not 0.5.
constexpr int PPH_BUF = 3 * PP_HALF;
enum { H_A0 = 0, H_A1 = 1, H_B0 = 2 };

template <int REGION>
__device__ __forceinline__ void pph_stage(const PPState& s, int t, int buf) {
    unsigned char* d = s.smem + buf * PPH_BUF + REGION * PP_HALF + s.dst;
    const char* g = (REGION < 2 ? s.gA : s.gW) + (int64_t)t * (PP_BK * 2);
    __builtin_amdgcn_global_load_lds((gptr_t)(g + (uint64_t)s.off[REGION][0]), (lptr_t)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(g + (uint64_t)s.off[REGION][1]), (lptr_t)(d + 1024), 16, 0, 0);
}

// cur = buffer of tile t, nxt = buffer of tile t + 2
template <int TAIL>
__device__ __forceinline__ void pph_tile(const PPState& s, int t, int cur, int nxt, float4_t (&acc)[8][4], short8_t (&a)[4][2],
                                         short8_t (&b0)[2][2]) {
    const unsigned char* base = s.smem + cur * PPH_BUF;
    // ---- phase 0: quadrant (a0, b0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b0[j][0] = *reinterpret_cast<const short8_t*>(base + H_B0 * PP_HALF + j * 2048 + s.lb0);
        b0[j][1] = *reinterpret_cast<const short8_t*>(base + H_B0 * PP_HALF + j * 2048 + s.lb1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + H_A0 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + H_A0 * PP_HALF + i * 2048 + s.la1);
    }
    if constexpr (TAIL == 0) { pph_stage<H_A0>(s, t + 2, nxt); pph_stage<H_B0>(s, t + 2, nxt); pp_vmcnt<10>(); }
    else if constexpr (TAIL == 1) { pp_vmcnt<6>(); } else { pp_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 0, 0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: quadrant (a1, b0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const short8_t*>(base + H_A1 * PP_HALF + i * 2048 + s.la0);
        a[i][1] = *reinterpret_cast<const short8_t*>(base + H_A1 * PP_HALF + i * 2048 + s.la1);
    }
    if constexpr (TAIL == 0) { pph_stage<H_A1>(s, t + 2, nxt); pp_vmcnt<8>(); }
    else if constexpr (TAIL == 1) { pp_vmcnt<2>(); } else { pp_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    pp_mfma_block(acc, a, b0, 4, 0);
    __builtin_amdgcn_s_barrier();
}

}  // namespace

// Epilogue activation on 4 values; the bf16 kernels use a 1.5e-7-accurate erf (Abramowitz-Stegun 7.1.26) for GELU --
// far below the bf16 rounding of the stored result (the fp32 kernels keep erff).
template <int ACT>
__device__ __forceinline__ float pp_act(float x) {
    if constexpr (ACT == DEVQA_ACT_RELU) return fmaxf(x, 0.f);
    if constexpr (ACT == DEVQA_ACT_GELU) {
        const float z = fabsf(x) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
        float p = fmaf(1.061405429f, t, -1.453152027f);
        p = fmaf(p, t, 1.421413741f);
        p = fmaf(p, t, -0.284496736f);
        p = fmaf(p, t, 0.254829592f);
        const float e = p * t * __expf(-z * z);      // 1 - erf(z)
        return 0.5f * x * (x >= 0.f ? 2.f - e : e);  // 0.5 x (1 + erf(x / sqrt 2))
    }
    if constexpr (ACT == DEVQA_ACT_QUICK_GELU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));
    return x;
}

// OPT-IN (variant 4, DEVQA_GEMM=24; the default keeps the erf form of pp_act): measured -4 % on the ViT fc1 GEMM (531 -> 505 us) = +0.5-0.8 % of
// the bench; 1 % of the stored values move by one bf16 step, and that alone moved the real-dim bf16 loss parity from 8.9e-3 to 1.19e-2 of
// north_star's 1e-2 (tests/test_realdim_batched_gpu.py) -- bf16 noise either way, but the bar is held on the erf form.
// GELU for outputs that are STORED AS bf16 (fc1 of the ViT / Q-Former): x * Phi(x) with
// Phi(x) = 0.5 + x Q(x^2) on |x| <= 4.2, Q a degree-8 minimax polynomial (weighted by x: |Phi error| <= 1.3e-5, |GELU error| <=
// 8.6e-6 |x|: two orders of magnitude below the bf16 rounding of typical outputs, 4e-5 absolute in the negative tail), x clamped
// beyond (Phi(4.2) = 1 - 1.3e-5).  No reciprocal, no exponential, and two values per instruction on v_pk_fma_f32 / v_pk_mul_f32:
// ~6.5 VALU instructions per element instead of ~13 + 2 quarter-rate transcendentals.  Outputs kept in fp32 still take pp_act.
typedef float pp_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pp_f2_t pp_gelu2_bf16(pp_f2_t x) {
    const pp_f2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(x[1], -4.2f, 4.2f)};
    const pp_f2_t s = xc * xc;
    pp_f2_t q = s * 5.997396590e-11f + -5.632817191e-09f;
    q = q * s + 2.343521054e-07f;
    q = q * s + -5.760521982e-06f;
    q = q * s + 9.457214143e-05f;
    q = q * s + -1.114136506e-03f;
    q = q * s + 9.830100464e-03f;
    q = q * s + -6.636033991e-02f;
    q = q * s + 3.989074326e-01f;
    return x * (xc * q + 0.5f);
}

typedef __bf16 pp_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float pp_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pp_pack2(float a, float b) {   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    pp_f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, pp_bf16x2_t));
}

struct PPArgs {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    const float* bias;
    int M, N, K;
    float alpha;
    const float* residual;
    bf16_t* out_bf16;
    float* out_f32;
    int64_t ldc;
    int tiles_m, tiles_n, group_m;
    // stream-K tail: tiles [dp_tiles, tiles_m*tiles_n) are cut into sk_wgs contiguous ranges of `sk_per` K-tile
    // iterations; partial accumulators go through `ws` ([sk tile][max_seg][256x256] fp32), `counters[sk tile]` elects
    // the workgroup that finishes a tile
    int dp_tiles, sk_wgs, sk_per, sk_max_seg;
    float* ws;
    int* counters;
    int bf16_fast;      // bf16-only output that qualifies for pp_epilogue_bf16
    int gelu_poly;      // bf16-stored GELU on the packed polynomial (variant 4; default 0: the erf form)
    int half_n;         // the last column tile runs the half-width schedule when N - n0 <= 128 (DEVQA_GEMM_HALFN=0 turns it off)
    int persist;        // one workgroup per CU walks its XCD's tile list with stride (workgroups per XCD) instead of one workgroup per tile; 2: early staging
    int persist_wgs;
};

// tile id -> (tile_m, tile_n): grouped order (group_m row-tiles of one column-tile, then the next column-tile), so the
// ~32 workgroups an XCD runs together share A row panels and W column panels in its L2
__device__ __forceinline__ void pp_tile_coords(const PPArgs& g, int id, int& tile_m, int& tile_n) {
    const int per_group = g.group_m * g.tiles_n;
    const int grp = id / per_group, in_g = id - grp * per_group;
    const int gm = min(g.group_m, g.tiles_m - grp * g.group_m);
    tile_m = grp * g.group_m + in_g % gm;
    tile_n = in_g / gm;
}

// K-tiles [k0, k0 + nk) of tile (tile_m, tile_n) -> acc.  Every wave executes the same number of barriers.
// drain: the workgroup has stores of a previous tile's epilogue in flight (persistent form) -- they count in vmcnt beside the LDS-DMA loads and the two
// kinds retire out of order, so the prologue waits for everything instead of for all but its 4 youngest half-tiles (the waits of the K loop only get laxer)
template <bool BAL = false>
__device__ __forceinline__ void pp_mainloop(const PPArgs& g, unsigned char* smem, int m0, int n0, int k0, int nk,
                                            float4_t (&acc)[8][4], const bool drain = false) {
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    PPState s;
    s.smem = smem;
    s.gA = reinterpret_cast<const char*>(g.A) + (int64_t)k0 * (PP_BK * 2);
    s.gW = reinterpret_cast<const char*>(g.W) + (int64_t)k0 * (PP_BK * 2);
    s.dst = wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);          // row inside the half-tile
        const int sc = (lane & 7) ^ ((r >> 1) & 7);              // source chunk for LDS chunk (lane & 7)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int grow = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            const int gcol = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), g.N - 1);
            s.off[h][i] = (unsigned)(((int64_t)grow * g.lda + sc * 8) * 2);          // R_A0 + h
            s.off[2 + h][i] = (unsigned)(((int64_t)gcol * g.ldw + sc * 8) * 2);      // R_B0 + h
        }
    }
    {
        const int c0 = (fq ^ ((fr >> 1) & 7)) << 4;
        s.la0 = wr * 8192 + fr * 128 + c0;
        s.la1 = wr * 8192 + fr * 128 + (c0 ^ 64);
        s.lb0 = wc * 4096 + fr * 128 + c0;
        s.lb1 = wc * 4096 + fr * 128 + (c0 ^ 64);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    short8_t a[4][2], b0[2][2], b1[2][2], b0n[2][2];
    // prologue: slots -6..-1 = A0,B0,B1,A1 of tile 0 and the first two half-tiles of tile 1; A0(0), B0(0) must have landed
    pp_stage<R_A0>(s, 0);
    pp_stage<R_B0>(s, 0);
    pp_stage<R_B1>(s, 0);
    pp_stage<R_A1>(s, 0);
    if (nk > 1) {
        pp_stage<BAL ? R_B0 : R_A0>(s, 1);
        pp_stage<BAL ? R_A0 : R_B0>(s, 1);
        if (drain) pp_vmcnt<0>(); else pp_vmcnt<8>();
    } else {
        if (drain) pp_vmcnt<0>(); else pp_vmcnt<4>();
    }
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);
    if constexpr (BAL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b0n[j][0] = *reinterpret_cast<const short8_t*>(smem + R_B0 * PP_HALF + j * 2048 + s.lb0);
            b0n[j][1] = *reinterpret_cast<const short8_t*>(smem + R_B0 * PP_HALF + j * 2048 + s.lb1);
        }
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier interval behind waves 0-3
    int t = 0;
    for (; t + 2 < nk; ++t) pp_tile<0, BAL>(s, t, acc, a, b0, b1, b0n);
    if (t + 1 < nk) {
        pp_tile<1, BAL>(s, t, acc, a, b0, b1, b0n);
        ++t;
    }
    pp_tile<2, BAL>(s, t, acc, a, b0, b1, b0n);
    if (wr == 0) __builtin_amdgcn_s_barrier();   // match the extra barrier of waves 4-7
}

// ---- persistent form with the next tile's K-tile 0 staged under the current tile's epilogue -------------------------------------------------------
// per-lane staging offsets and fragment addresses of tile (m0, n0), K-tiles from 0 (the set-up of pp_mainloop)
__device__ __forceinline__ void pp_make_state(const PPArgs& g, unsigned char* smem, int m0, int n0, PPState& s) {
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    s.smem = smem;
    s.gA = reinterpret_cast<const char*>(g.A);
    s.gW = reinterpret_cast<const char*>(g.W);
    s.dst = wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);
        const int sc = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int grow = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            const int gcol = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), g.N - 1);
            s.off[h][i] = (unsigned)(((int64_t)grow * g.lda + sc * 8) * 2);
            s.off[2 + h][i] = (unsigned)(((int64_t)gcol * g.ldw + sc * 8) * 2);
        }
    }
    const int c0 = (fq ^ ((fr >> 1) & 7)) << 4;
    s.la0 = wr * 8192 + fr * 128 + c0;
    s.la1 = wr * 8192 + fr * 128 + (c0 ^ 64);
    s.lb0 = wc * 4096 + fr * 128 + c0;
    s.lb1 = wc * 4096 + fr * 128 + (c0 ^ 64);
}

// the four half-tiles of K-tile 0 -> buffer 0 ([0, 64 KiB)); the caller's epilogue stages through [64 KiB, 128 KiB)
__device__ __forceinline__ void pp_stage_k0(const PPState& s) {
    pp_stage<R_A0>(s, 0);
    pp_stage<R_B0>(s, 0);
    pp_stage<R_B1>(s, 0);
    pp_stage<R_A1>(s, 0);
}

// K loop of a tile whose K-tile 0 is staged and has landed for THIS wave's pieces (waited for in the epilogue, in front of its first store); nk >= 3
__device__ __forceinline__ void pp_mainloop_pre(const PPState& s, int nk, float4_t (&acc)[8][4]) {
    const int wr = __builtin_amdgcn_readfirstlane(pp_tid() >> 6) >> 2;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    short8_t a[4][2], b0[2][2], b1[2][2], b0n[2][2];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // every wave: its epilogue staging reads are done (buffer 1 is free) and its pieces of K-tile 0 have landed
    pp_stage<R_A0>(s, 1);
    pp_stage<R_B0>(s, 1);
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier interval behind waves 0-3
    pp_tile_pre0(s, acc, a, b0, b1);
    int t = 1;
    for (; t + 2 < nk; ++t) pp_tile<0>(s, t, acc, a, b0, b1, b0n);
    if (t + 1 < nk) {
        pp_tile<1>(s, t, acc, a, b0, b1, b0n);
        ++t;
    }
    pp_tile<2>(s, t, acc, a, b0, b1, b0n);
    if (wr == 0) __builtin_amdgcn_s_barrier();   // match the extra barrier of waves 4-7
}

// Epilogue.  All LDS traffic of the K loop is over (last ds_read in phase 2 of the last tile, all LDS-DMA retired by the
// vmcnt(0) of its phases 1-3, and every wave is past two more barriers).  Each wave transposes its 128x64 accumulator
// tile through a PRIVATE 16-KiB LDS region, 64 rows at a time, so that global accesses are row-contiguous: a lane owns
// 4 consecutive columns of one row, 16 lanes cover 256 B (fp32) / 128 B (bf16) of it.
// MFMA operands were fed swapped: acc[i][j][e] = C[i*16 + fr][j*16 + fq*4 + e].
// HALF (half-width column tile): the wave tile is 128 x 32 (acc[.][0..1]); 8 lanes cover a row's 128 B, 8 rows per step.
template <int ACT, bool HALF = false>
__device__ __forceinline__ void pp_epilogue(const PPArgs& g, unsigned char* smem, int m0, int n0, const float4_t (&acc)[8][4]) {
    constexpr int NJ = HALF ? 2 : 4;        // 16-column blocks of the wave tile
    constexpr int LPR = NJ * 4;             // lanes per row (4 columns each)
    constexpr int RPI = 64 / LPR;           // rows per step
    constexpr int NIT = 64 / RPI;           // steps per 64-row half
    constexpr int RB = NJ * 64;             // staging row bytes
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    unsigned char* stg = smem + wave * 16384;
    const int q = lane % LPR, lrow = lane / LPR;
    const int gn = n0 + wc * (NJ * 16) + q * 4;
    const bool col_ok = gn < g.N;
    const int M = g.M;
    const int64_t ldc = g.ldc;
    const float alpha = g.alpha;
    const float* residual = g.residual;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias != nullptr && col_ok) bv = *reinterpret_cast<const float4*>(g.bias + gn);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int gm0 = m0 + wr * 128 + half * 64 + lrow;   // + it * RPI
        float4 rv[NIT];
        if (residual != nullptr) {   // all loads in flight before the first store (residual may alias out_f32).  Fetching the
                                     // second half's rows under the first half's stores was measured 5-25 % SLOWER (loads
                                     // queue behind the stores, profiles/r01_summary.md section G)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int gm = gm0 + it * RPI;
                rv[it] = (col_ok && gm < M) ? *reinterpret_cast<const float4*>(residual + (int64_t)gm * ldc + gn)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = i * 16 + fr;
                const int sw = HALF ? ((row >> 1) & 7) : (row & 15);
                *reinterpret_cast<float4_t*>(stg + row * RB + (((j * 4 + fq) ^ sw) << 4)) = acc[half * 4 + i][j];
            }
        if (residual != nullptr) {
            // Every residual value is made "used" HERE, behind the LDS writes: on gfx950 stores count in vmcnt like loads, and with loads AND
            // stores pending the compiler cannot count (the two kinds complete out of order) -- it waited vmcnt(0) in front of the add of every
            // later iteration, i.e. for the PREVIOUS iteration's store to be acknowledged: 32 serialised store round trips per tile, the 22-30 k
            // cycles the -DPP_TIMING stamps showed for this epilogue.  One wait for the loads, then the 16 stores of a half pipeline freely.
#pragma unroll
            for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rv[it].x), "+v"(rv[it].y), "+v"(rv[it].z), "+v"(rv[it].w));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + lrow;
            const int sw = HALF ? ((row >> 1) & 7) : (row & 15);
            const float4_t sv = *reinterpret_cast<const float4_t*>(stg + row * RB + ((q ^ sw) << 4));
            const int gm = gm0 + it * RPI;
            float4 v = make_float4((sv[0] + bv.x) * alpha, (sv[1] + bv.y) * alpha, (sv[2] + bv.z) * alpha, (sv[3] + bv.w) * alpha);
            v.x = pp_act<ACT>(v.x); v.y = pp_act<ACT>(v.y); v.z = pp_act<ACT>(v.z); v.w = pp_act<ACT>(v.w);
            if (residual != nullptr) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
            if (col_ok && gm < M) {
                const int64_t o = (int64_t)gm * ldc + gn;
                if (g.out_f32) *reinterpret_cast<float4*>(g.out_f32 + o) = v;
                if (g.out_bf16) {
                    uint2 p;
                    p.x = pp_pack2(v.x, v.y);
                    p.y = pp_pack2(v.z, v.w);
                    *reinterpret_cast<uint2*>(g.out_bf16 + o) = p;
                }
            }
        }
    }
}

// bf16-only outputs (no fp32 output, no residual: QKV, fc1, gate/up projections): bias / alpha / activation are applied in the
// MFMA layout and the tile is transposed through LDS as packed bf16 -- half the LDS bytes of the fp32 transposition above and
// 16-byte global stores (8 lanes cover the wave's 128-byte row segment).  Staging: per wave 2 x [64 rows][128 B], the 16-byte
// chunk index XOR-swizzled with (row >> 1) & 7: the b64 writes of 16 rows x 2 lanes and the b128 reads of 2 rows x 8 lanes are
// both bank-conflict-free.  Needs N % 8 == 0, ldc % 8 == 0 and a 16-byte aligned output (checked on the host: g.bf16_fast).
// PRE (persistent form): the next tile's K-tile 0 is in flight into [0, 64 KiB): one 8-KiB staging area per wave in [64 KiB, 128 KiB) serves both
// 64-row halves (a wave's LDS operations execute in order), and the wave waits for its pieces of that K-tile in front of its first global store --
// from there on its vmcnt counts stores, which the next tile's first phases must not wait for (pp_tile_pre0).
template <int ACT, bool HALF = false, bool PRE = false>
__device__ __forceinline__ void pp_epilogue_bf16(const PPArgs& g, unsigned char* smem, int m0, int n0, const float4_t (&acc)[8][4],
                                                 const PPState* next = nullptr) {
    constexpr int NJ = HALF ? 2 : 4;        // 16-column blocks of the wave tile (HALF: 128 x 32, a row is 64 B = 4 lanes)
    constexpr int RB = NJ * 32;             // staging row bytes
    constexpr int LPR = NJ * 2;             // lanes per row (16 B each)
    constexpr int RPI = 64 / LPR;           // rows per step
    constexpr int NIT = 64 / RPI;           // steps per 64-row half
    constexpr int SWM = LPR - 1;            // chunk swizzle mask
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    unsigned char* stg = PRE ? smem + 65536 + wave * 8192 : smem + wave * 16384;
    const float alpha = g.alpha;
    float4 bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wc * (NJ * 16) + j * 16 + fq * 4;
        bj[j] = (g.bias != nullptr && col < g.N) ? *reinterpret_cast<const float4*>(g.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int rrow = lane / LPR, pc = lane % LPR;
    if constexpr (PRE) {     // the bias values are in registers BEFORE the successor's K-tile 0 is staged: a wait for them behind the LDS-DMA would wait for it too
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(bj[j].x), "+v"(bj[j].y), "+v"(bj[j].z), "+v"(bj[j].w));
        pp_stage_k0(*next);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        unsigned char* area = PRE ? stg : stg + half * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + fr;
            const int sw = (row >> 1) & SWM;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4_t a = acc[half * 4 + i][j];
                float v0, v1, v2, v3;
                if (ACT == DEVQA_ACT_GELU && g.gelu_poly) {
                    const pp_f2_t lo = pp_gelu2_bf16((pp_f2_t){(a[0] + bj[j].x) * alpha, (a[1] + bj[j].y) * alpha});
                    const pp_f2_t hi = pp_gelu2_bf16((pp_f2_t){(a[2] + bj[j].z) * alpha, (a[3] + bj[j].w) * alpha});
                    v0 = lo[0]; v1 = lo[1]; v2 = hi[0]; v3 = hi[1];
                } else {
                    v0 = pp_act<ACT>((a[0] + bj[j].x) * alpha); v1 = pp_act<ACT>((a[1] + bj[j].y) * alpha);
                    v2 = pp_act<ACT>((a[2] + bj[j].z) * alpha); v3 = pp_act<ACT>((a[3] + bj[j].w) * alpha);
                }
                uint2 p;
                p.x = pp_pack2(v0, v1);
                p.y = pp_pack2(v2, v3);
                if constexpr (PRE) {     // an asm write: behind an LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS store it can see
                                         // (it cannot tell that the staging area and the DMA's buffer 0 are disjoint) -- the wait for the K-tile in flight
                    const uint32_t la = (uint32_t)reinterpret_cast<uintptr_t>((lptr_t)(area + row * RB + (((j * 2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8));
                    asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(p) : "memory");
                } else {
                    *reinterpret_cast<uint2*>(area + row * RB + (((j * 2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8) = p;
                }
            }
        }
        if constexpr (PRE) {
            if (half == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of the next tile's K-tile 0 (+ the asm LDS writes)
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rrow;
            uint4 val;
            if constexpr (PRE) {     // (asm for the same reason as the writes: a visible LDS read behind the DMA gets a vmcnt(0), i.e. waits for the previous store)
                const uint32_t la = (uint32_t)reinterpret_cast<uintptr_t>((lptr_t)(area + row * RB + (pc << 4)));
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(val) : "v"(la) : "memory");
            } else {
                val = *reinterpret_cast<const uint4*>(area + row * RB + (pc << 4));
            }
            const int gm = m0 + wr * 128 + half * 64 + row;
            const int gcol = n0 + wc * (NJ * 16) + ((pc ^ ((row >> 1) & SWM)) << 3);
            if (gm < g.M && gcol < g.N) *reinterpret_cast<uint4*>(g.out_bf16 + (int64_t)gm * g.ldc + gcol) = val;
        }
    }
}

// Fused SwiGLU (DEVQA_ACT_SWIGLU_IL16, include/devqa.h): the weight rows are interleaved in blocks of 16 (gate | up), so the wave tile's column
// blocks are (gate, up, gate, up) of 32 consecutive output columns: out = silu(acc[.][2 jj]) * acc[.][2 jj + 1] element by element in the MFMA
// layout, then the half-width bf16 transposition (2 x [64 rows][64 B] per wave, 4 lanes per row) and 16-byte stores into the [M, N / 2] output.
__device__ __forceinline__ void pp_epilogue_swiglu(const PPArgs& g, unsigned char* smem, int m0, int n0, const float4_t (&acc)[8][4]) {
    constexpr int RB = 64, LPR = 4, RPI = 16, NIT = 4, SWM = 3;
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    unsigned char* stg = smem + wave * 16384;
    const int rrow = lane / LPR, pc = lane % LPR;
    const int No = g.N >> 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        unsigned char* area = stg + half * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + fr;
            const int sw = (row >> 1) & SWM;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4_t ga = acc[half * 4 + i][2 * jj], ua = acc[half * 4 + i][2 * jj + 1];
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ga[e] / (1.f + __expf(-ga[e])) * ua[e];      // (the expression of swiglu_bf16x8_kernel)
                uint2 p;
                p.x = pp_pack2(o[0], o[1]);
                p.y = pp_pack2(o[2], o[3]);
                *reinterpret_cast<uint2*>(area + row * RB + (((jj * 2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8) = p;
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rrow;
            const uint4 val = *reinterpret_cast<const uint4*>(area + row * RB + (pc << 4));
            const int gm = m0 + wr * 128 + half * 64 + row;
            const int gcol = (n0 >> 1) + wc * 32 + ((pc ^ ((row >> 1) & SWM)) << 3);
            if (gm < g.M && gcol < No) *reinterpret_cast<uint4*>(g.out_bf16 + (int64_t)gm * g.ldc + gcol) = val;
        }
    }
}

// Half-width column tile: K-tiles [0, nk) of rows [m0, m0 + 256) x columns [n0, n0 + 128) -> acc[.][0..1] (wave (wr, wc): rows
// wr * 128 + 0..127, columns wc * 32 + 0..31).  Same barrier count for every wave.
__device__ __forceinline__ void pp_mainloop_half(const PPArgs& g, unsigned char* smem, int m0, int n0, int nk, float4_t (&acc)[8][4],
                                                 const bool drain = false) {
    const int tid = pp_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    PPState s;
    s.smem = smem;
    s.gA = reinterpret_cast<const char*>(g.A);
    s.gW = reinterpret_cast<const char*>(g.W);
    s.dst = wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);          // row inside the half-tile
        const int sc = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int grow = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            s.off[h][i] = (unsigned)(((int64_t)grow * g.lda + sc * 8) * 2);          // H_A0 + h
        }
        const int gcol = min(n0 + r, g.N - 1);
        s.off[H_B0][i] = (unsigned)(((int64_t)gcol * g.ldw + sc * 8) * 2);
        s.off[3][i] = 0;
    }
    {
        const int c0 = (fq ^ ((fr >> 1) & 7)) << 4;
        s.la0 = wr * 8192 + fr * 128 + c0;
        s.la1 = wr * 8192 + fr * 128 + (c0 ^ 64);
        s.lb0 = wc * 4096 + fr * 128 + c0;
        s.lb1 = wc * 4096 + fr * 128 + (c0 ^ 64);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    short8_t a[4][2], b0[2][2];
    pph_stage<H_A0>(s, 0, 0);
    pph_stage<H_B0>(s, 0, 0);
    pph_stage<H_A1>(s, 0, 0);
    if (nk > 1) {
        pph_stage<H_A0>(s, 1, 1);
        pph_stage<H_B0>(s, 1, 1);
        pph_stage<H_A1>(s, 1, 1);
        if (drain) pp_vmcnt<0>(); else pp_vmcnt<8>();
    } else {
        if (drain) pp_vmcnt<0>(); else pp_vmcnt<2>();
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier interval behind waves 0-3
    int t = 0, cur = 0, nxt = 2;
    for (; t + 2 < nk; ++t) {
        pph_tile<0>(s, t, cur, nxt, acc, a, b0);
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    if (t + 1 < nk) {
        pph_tile<1>(s, t, cur, nxt, acc, a, b0);
        cur = cur == 2 ? 0 : cur + 1;
        ++t;
    }
    pph_tile<2>(s, t, cur, nxt, acc, a, b0);
    if (wr == 0) __builtin_amdgcn_s_barrier();   // match the extra barrier of waves 4-7
}

template <int ACT, bool SK, bool BAL = false, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const PPArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4_t acc[8][4];
    const int nk = g.K / PP_BK;
    const int tid = threadIdx.x;
    PP_STAMP(0);
    PP_STAMP_RT(5);
#ifdef PP_TIMING
    if (threadIdx.x == 0)      // which CU ran this workgroup: XCC_ID (hwreg 20) and the SE / SH / CU fields of HW_ID (hwreg 4)
        reinterpret_cast<unsigned long long*>(g.ws)[blockIdx.x * 8 + 4] =
            ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xff00);
#endif
    if constexpr (!SK && !PERSIST) {   // one whole tile per workgroup: the default
        const int bid = blockIdx.x, nwg = g.dp_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        int tile_m, tile_n;
        pp_tile_coords(g, (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx, tile_m, tile_n);
        if (ACT != DEVQA_ACT_SWIGLU_IL16 && g.half_n && g.N - tile_n * PP_BN <= PP_BN / 2) {      // uniform: the half-width last column tile
            pp_mainloop_half(g, smem, tile_m * PP_BM, tile_n * PP_BN, nk, acc);
            if (g.bf16_fast) pp_epilogue_bf16<ACT, true>(g, smem, tile_m * PP_BM, tile_n * PP_BN, acc);
            else pp_epilogue<ACT, true>(g, smem, tile_m * PP_BM, tile_n * PP_BN, acc);
            return;
        }
        pp_mainloop<BAL>(g, smem, tile_m * PP_BM, tile_n * PP_BN, 0, nk, acc);
        PP_STAMP(2);
        if constexpr (ACT == DEVQA_ACT_SWIGLU_IL16) pp_epilogue_swiglu(g, smem, tile_m * PP_BM, tile_n * PP_BN, acc);
        else if (g.bf16_fast) pp_epilogue_bf16<ACT>(g, smem, tile_m * PP_BM, tile_n * PP_BN, acc);
        else pp_epilogue<ACT>(g, smem, tile_m * PP_BM, tile_n * PP_BN, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_STAMP(3);
        PP_STAMP_RT(6);
        return;
    }
    if constexpr (!SK && PERSIST) {
        // Persistent form (A/B: id 8): one workgroup per CU walks its XCD's contiguous tile list with stride (workgroups per XCD).  When a tile and
        // its successor are both full-width and the output takes the packed-bf16 epilogue (g.persist == 2), the successor's K-tile 0 is staged into
        // buffer 0 BEFORE the epilogue, which then stages through buffer 1 (PRE forms above); otherwise the tiles simply follow each other.
        const int bid = blockIdx.x, nwg = g.dp_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx0 = bid >> 3;
        const int cnt = q + (xcd < r ? 1 : 0), base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int step = (int)(gridDim.x >> 3);
        const bool can_pre = ACT != DEVQA_ACT_SWIGLU_IL16 && g.bf16_fast && g.persist == 2 && nk >= 3;
        PPState sn;
        bool pre = false;                  // this tile's K-tile 0 was staged under the previous tile's epilogue (its state is sn)
        for (int idx = idx0; idx < cnt; idx += step) {
            int tile_m, tile_n;
            pp_tile_coords(g, base + idx, tile_m, tile_n);
            const int m0 = tile_m * PP_BM, n0 = tile_n * PP_BN;
            const bool half = ACT != DEVQA_ACT_SWIGLU_IL16 && g.half_n && g.N - n0 <= PP_BN / 2;
            if (pre) {
                pp_mainloop_pre(sn, nk, acc);
            } else {
                const bool drain = idx != idx0;
                if (drain) {     // every wave is done reading its epilogue staging (the stores that consumed those reads are issued): the prologue's
                                 // LDS-DMA restages all of it.  NOT __syncthreads(): its fence would wait for the stores to be acknowledged
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (half) pp_mainloop_half(g, smem, m0, n0, nk, acc, drain);
                else pp_mainloop<BAL>(g, smem, m0, n0, 0, nk, acc, drain);
            }
            // the successor, if it can be staged early
            pre = false;
            if (can_pre && idx + step < cnt) {
                int nm, nn;
                pp_tile_coords(g, base + idx + step, nm, nn);
                if (!(g.half_n && g.N - nn * PP_BN <= PP_BN / 2)) {
                    pp_make_state(g, smem, nm * PP_BM, nn * PP_BN, sn);      // (staged inside the PRE epilogue, behind its bias loads)
                    pre = true;
                }
            }
            if (half) {
                if (g.bf16_fast) { if (pre) pp_epilogue_bf16<ACT, true, true>(g, smem, m0, n0, acc, &sn); else pp_epilogue_bf16<ACT, true>(g, smem, m0, n0, acc); }
                else pp_epilogue<ACT, true>(g, smem, m0, n0, acc);
            } else if constexpr (ACT == DEVQA_ACT_SWIGLU_IL16) {
                pp_epilogue_swiglu(g, smem, m0, n0, acc);
            } else if (g.bf16_fast) {
                if (pre) pp_epilogue_bf16<ACT, false, true>(g, smem, m0, n0, acc, &sn); else pp_epilogue_bf16<ACT>(g, smem, m0, n0, acc);
            } else {
                pp_epilogue<ACT>(g, smem, m0, n0, acc);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_STAMP(3);
        PP_STAMP_RT(6);
        return;
    }
    // Every workgroup walks a range [it, it1) of the (tile, K-tile) iteration space.  A data-parallel workgroup owns
    // exactly one whole tile; a stream-K workgroup owns `sk_per` consecutive iterations of the tail tiles.
    int it, it1;
    const int w = (int)blockIdx.x - g.dp_tiles;
    if (w < 0) {
        // workgroups that share an XCD (blockIdx % 8) get a contiguous range of tile ids (bijective for any count)
        const int bid = blockIdx.x, nwg = g.dp_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        it = ((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx) * nk;
        it1 = it + nk;
    } else {
        it = g.dp_tiles * nk + w * g.sk_per;
        it1 = min(g.tiles_m * g.tiles_n * nk, it + g.sk_per);
    }
    while (it < it1) {
        const int tile = it / nk;
        const int k0 = it - tile * nk;
        const int kn = min(nk - k0, it1 - it);
        int tile_m, tile_n;
        pp_tile_coords(g, tile, tile_m, tile_n);
        const int m0 = tile_m * PP_BM, n0 = tile_n * PP_BN;
        pp_mainloop(g, smem, m0, n0, k0, kn, acc);
        bool whole = true;
        if (w >= 0) {
            const int ts = tile - g.dp_tiles;             // tail-tile index
            const int first_wg = (ts * nk) / g.sk_per, last_wg = ((ts + 1) * nk - 1) / g.sk_per;
            const int nseg = last_wg - first_wg + 1;
            if (nseg > 1) {
                // partial tile: accumulators -> workspace in fragment order (lane-contiguous 16-B pieces: fully coalesced)
                float4_t* base = reinterpret_cast<float4_t*>(g.ws) + (int64_t)ts * g.sk_max_seg * (32 * 512);
                float4_t* slot = base + (int64_t)(w - first_wg) * (32 * 512);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) slot[(i * 4 + j) * 512 + tid] = acc[i][j];
                // every wave's stores drained, workgroup barrier, then ONE agent-scope acq_rel RMW on the tile's counter
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int* flag = reinterpret_cast<int*>(smem);
                if (tid == 0) {
                    const int old = __hip_atomic_fetch_add(g.counters + ts, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = old == nseg - 1;
                    if (last) __hip_atomic_store(g.counters + ts, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
                    *flag = last;
                }
                __syncthreads();
                whole = *flag != 0;
                __syncthreads();      // the flag word is LDS the epilogue / the next prologue overwrites
                if (whole) {          // this workgroup finished the tile: sum the partials in K order (deterministic)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    for (int sg = 0; sg < nseg; ++sg) {
                        const float4_t* p = base + (int64_t)sg * (32 * 512);
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] += p[(i * 4 + j) * 512 + tid];
                    }
                }
            }
        }
        if (whole) pp_epilogue<ACT>(g, smem, m0, n0, acc);
        it += kn;
        if (it < it1) __syncthreads();   // LDS is restaged by the next segment's prologue
    }
}

template <int ACT, bool SK, bool BAL = false, bool PERSIST = false>
static int launch_pp_k(const PPArgs& g, hipStream_t st) {
    const size_t smem = g.half_n ? 3 * PPH_BUF : 2 * PP_BUF;      // 144 KiB when the launch has a half-width column tile, else 128 KiB
    auto kern = gemm_bf16_pp_kernel<ACT, SK, BAL, PERSIST>;
    static std::atomic<unsigned> attr_done{0};
    devqa_set_max_smem(kern, 3 * PPH_BUF, attr_done);
    hipLaunchKernelGGL(kern, dim3(g.persist ? g.persist_wgs : g.dp_tiles + g.sk_wgs), dim3(512), smem, st, g);
    DEVQA_LAUNCH_CHECK("gemm_bf16_pp");
    return DEVQA_OK;
}

template <int ACT>
static int launch_pp(const PPArgs& g, hipStream_t st) {
    if (g.group_m < 0) {      // variant 5: balanced ds_read schedule (experimental)
        PPArgs h = g;
        h.group_m = -g.group_m;
        return launch_pp_k<ACT, false, true>(h, st);
    }
    if (g.persist) return launch_pp_k<ACT, false, false, true>(g, st);
    return launch_pp_k<ACT, false>(g, st);
}

// id 6: group_m 4 with the fp32 LDS transposition for every output kind (A/B of pp_epilogue_bf16).
// id 7: group_m 4 with a full-width schedule on every column tile (A/B of the half-width last column tile).
// id 8: group_m 4, persistent form (one workgroup per CU walks its XCD's tile list).
// id: 0 group_m 8, 1 group_m 1 (plain column-major tile order), 2 group_m 4 (default), 3 group_m 16,
//     4 = 2 (was: group_m 4 + a stream-K tail.  The partial-tile hand-off -- 256 KiB per segment through HBM/L2 plus an
//       agent-scope acq_rel RMW whose release/acquire writes back / invalidates the XCD's L2 -- cost more than the partial round it
//       removed on every shape of this path: measured 0.50-0.93x, profiles/r01_summary.md; its launcher, the only code that
//       allocated and freed device memory inside a launch path, was removed in round 2.  The SK template parameter of the kernel
//       is kept but no longer instantiated.)
int launch_gemm_pp(int id, const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, int M, int N, int K, float alpha,
                   int act, const float* residual, bf16_t* out_bf16, float* out_f32, int64_t ldc, hipStream_t st) {
    static const int gms[9] = {8, 1, 4, 16, 4, -4, 4, 4, 4};
    if (id < 0 || id > 8) return devqa_fail(DEVQA_E_ARG, "gemm_pp: unknown variant %d", id);
    if (K % PP_BK != 0 || K < PP_BK) return devqa_fail(DEVQA_E_SHAPE, "gemm_pp: K=%d must be a positive multiple of 64", K);
    if ((int64_t)M * lda * 2 >= (1ll << 32) || (int64_t)N * ldw * 2 >= (1ll << 32))
        return devqa_fail(DEVQA_E_SHAPE, "gemm_pp: operands must span < 4 GiB (32-bit lane offsets)");
    PPArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.residual = residual;
    g.out_bf16 = out_bf16; g.out_f32 = out_f32; g.ldc = ldc;
    g.tiles_m = (M + PP_BM - 1) / PP_BM;
    g.tiles_n = (N + PP_BN - 1) / PP_BN;
    g.group_m = gms[id];
    const int T = g.tiles_m * g.tiles_n;
    g.dp_tiles = T; g.sk_wgs = 0; g.sk_per = 1; g.sk_max_seg = 1; g.ws = nullptr; g.counters = nullptr;
    g.bf16_fast = out_bf16 != nullptr && out_f32 == nullptr && residual == nullptr && N % 8 == 0 && ldc % 8 == 0 &&
                  (((uintptr_t)out_bf16) & 15) == 0 && (bias == nullptr || (((uintptr_t)bias) & 15) == 0);
    if (id == 6) g.bf16_fast = 0;       // A/B: the fp32 transposition for every output kind
    g.gelu_poly = id == 4;              // variant 4 (DEVQA_GEMM=24): bf16-stored GELU on the packed polynomial
    static const int halfn_env = getenv("DEVQA_GEMM_HALFN") ? atoi(getenv("DEVQA_GEMM_HALFN")) : 1;
    g.half_n = halfn_env && id != 7 && g.group_m > 0 && N - (g.tiles_n - 1) * PP_BN <= PP_BN / 2;     // id 7: A/B without it
    // persistent form (DEVQA_GEMM_PERSIST=1): as many workgroups as CUs (a workgroup's 128-144 KiB of LDS leave room for one per CU), multiple of 8
    static const int persist_env = getenv("DEVQA_GEMM_PERSIST") ? atoi(getenv("DEVQA_GEMM_PERSIST")) : 0;
    static const int n_cu = [] { int dev = 0, n = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n & ~7; }();
    g.persist = ((persist_env || id == 8) && id != 5 && n_cu >= 8 && T > n_cu) ? (id == 8 ? 2 : persist_env) : 0;      // id 8 (DEVQA_GEMM=28): A/B of the persistent form
    g.persist_wgs = n_cu;
    switch (act) {
        case DEVQA_ACT_NONE: return launch_pp<DEVQA_ACT_NONE>(g, st);
        case DEVQA_ACT_RELU: return launch_pp<DEVQA_ACT_RELU>(g, st);
        case DEVQA_ACT_GELU: return launch_pp<DEVQA_ACT_GELU>(g, st);
        case DEVQA_ACT_QUICK_GELU: return launch_pp<DEVQA_ACT_QUICK_GELU>(g, st);
        case DEVQA_ACT_SWIGLU_IL16:
            if (bias != nullptr || residual != nullptr || out_f32 != nullptr || out_bf16 == nullptr || alpha != 1.f || N % PP_BN != 0 || ldc % 8 != 0 ||
                (((uintptr_t)out_bf16) & 15) != 0)
                return devqa_fail(DEVQA_E_SHAPE, "gemm_pp: fused SwiGLU needs a bf16 output only, no bias / residual / alpha, N %% 256 == 0, ldc %% 8 == 0");
            g.half_n = 0;
            return launch_pp<DEVQA_ACT_SWIGLU_IL16>(g, st);
    }
    return devqa_fail(DEVQA_E_ARG, "gemm_pp: unknown activation %d", act);
}

#ifdef PP_TIMING
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
int devqa_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
int main() {
    struct Sh { const char* name; int M, N, K; int resid; } shapes[] = {{"qkv", 65536, 4224, 1408, 0}, {"proj", 65536, 1408, 1408, 1}, {"fc2", 65536, 1408, 6144, 1},
                                                                       {"opt_fc1", 20480, 10240, 2560, 0}};
    for (auto& sh : shapes) {
        bf16_t *A, *W, *ob; float *of, *bias; unsigned long long* st;
        hipMalloc(&A, (size_t)sh.M * sh.K * 2); hipMalloc(&W, (size_t)sh.N * sh.K * 2); hipMalloc(&ob, (size_t)sh.M * sh.N * 2);
        hipMalloc(&of, (size_t)sh.M * sh.N * 4); hipMalloc(&bias, sh.N * 4);
        const int T = ((sh.M + 255) / 256) * ((sh.N + 255) / 256);
        hipMalloc(&st, (size_t)T * 64);
        std::vector<unsigned short> h((size_t)sh.M * sh.K);
        unsigned x = 1;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 3) & 0x8000)); }
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        h.resize((size_t)sh.N * sh.K);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3800 + ((x >> 9) & 0x3ff) - ((x >> 3) & 0x8000)); }
        hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, sh.N * 4); hipMemset(of, 0, (size_t)sh.M * sh.N * 4);
        PPArgs g;
        g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.bias = bias; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.alpha = 1.f;
        g.residual = sh.resid ? of : nullptr; g.out_bf16 = sh.resid ? nullptr : ob; g.out_f32 = sh.resid ? of : nullptr; g.ldc = sh.N;
        g.tiles_m = (sh.M + 255) / 256; g.tiles_n = (sh.N + 255) / 256; g.group_m = 4; g.dp_tiles = T; g.sk_wgs = 0; g.sk_per = 1; g.sk_max_seg = 1;
        g.ws = (float*)st; g.counters = nullptr; g.bf16_fast = !sh.resid; g.gelu_poly = 0;
        g.half_n = sh.N - (g.tiles_n - 1) * 256 <= 128;
        g.persist = 0; g.persist_wgs = 0;
        for (int rep = 0; rep < 5; ++rep) launch_pp_k<DEVQA_ACT_NONE, false>(g, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned long long> hs((size_t)T * 8);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> pro, loop, epi;
        for (int t = 0; t < T; ++t) {
            if ((t % g.tiles_n) == g.tiles_n - 1 && g.half_n) continue;      // full tiles only
            pro.push_back((double)(hs[t * 8 + 1] - hs[t * 8 + 0]));
            loop.push_back((double)(hs[t * 8 + 2] - hs[t * 8 + 1]));
            epi.push_back((double)(hs[t * 8 + 3] - hs[t * 8 + 2]));
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        // the gap a CU leaves between the end of one workgroup and the start of the next one it runs (same counter: one XCD)
        std::vector<double> gap;
        {
            std::vector<std::pair<unsigned long long, int>> order;
            for (int t = 0; t < T; ++t) order.push_back({(hs[t * 8 + 4] << 44) | (hs[t * 8 + 0] & ((1ull << 44) - 1)), t});
            std::sort(order.begin(), order.end());
            for (size_t i = 1; i < order.size(); ++i) {
                const int a = order[i - 1].second, b = order[i].second;
                if (hs[a * 8 + 4] == hs[b * 8 + 4]) gap.push_back((double)hs[b * 8 + 0] - (double)hs[a * 8 + 3]);
            }
        }
        const double gp = gap.empty() ? 0.0 : med(gap);
        printf("%-8s CU hand-over gap (end of a workgroup -> start of the next on the same CU): median %.0f cycles over %zu pairs\n", sh.name, gp, gap.size());
        const double p = med(pro), l = med(loop), e = med(epi);
        printf("%-8s M=%d N=%d K=%d: median shader cycles per tile: prologue %.0f (%.1f%%)  main loop %.0f (%.1f%%)  epilogue %.0f (%.1f%%)  | MFMA issue floor %d\n",
               sh.name, sh.M, sh.N, sh.K, p, 100 * p / (p + l + e), l, 100 * l / (p + l + e), e, 100 * e / (p + l + e), sh.K / 64 * 2048);
        // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles per 100-MHz real-time tick, per workgroup, and the launch time by
        // HIP events -- the one-tile-per-workgroup form against the persistent form (bf16 outputs: with the successor's K-tile 0 staged under the epilogue)
        for (int form = 0; form < 2; ++form) {
            int ncu = 0;
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
            g.persist = form ? 2 : 0;
            g.persist_wgs = ncu & ~7;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto run = [&] { if (form) launch_pp_k<DEVQA_ACT_NONE, false, false, true>(g, nullptr); else launch_pp_k<DEVQA_ACT_NONE, false>(g, nullptr); };
            for (int rep = 0; rep < 200; ++rep) run();          // ~0.2 s of back-to-back launches before the measured ones
            hipEventRecord(e0, nullptr);
            for (int rep = 0; rep < 20; ++rep) run();
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const int nw = form ? g.persist_wgs : T;
            std::vector<unsigned long long> h2((size_t)nw * 8);
            hipMemcpy(h2.data(), st, h2.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> clk, cyc;
            for (int w = 0; w < nw; ++w) {
                const double c = (double)(h2[w * 8 + 3] - h2[w * 8 + 0]), rt = (double)(h2[w * 8 + 6] - h2[w * 8 + 5]);
                if (rt > 0) { clk.push_back(c / rt * 0.1); cyc.push_back(c); }
            }
            printf("%-8s %s: %.1f us per launch; in-kernel clock %.3f GHz (median over workgroups); shader cycles per workgroup %.0f (%s)\n", sh.name,
                   form ? "persistent form" : "one tile per workgroup", ms * 1e3 / 20, med(clk), med(cyc), form ? "all its tiles" : "one tile");
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        hipFree(A); hipFree(W); hipFree(ob); hipFree(of); hipFree(bias); hipFree(st);
    }
    return 0;
}
#endif
