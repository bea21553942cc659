// What would fewer LDS fragment bytes per MFMA buy the 256 x 256 GEMM tile?  A bare loop per CU -- LDS fragment reads (ds_read_b128 of random bf16 data) +
// v_mfma_f32_16x16x32_bf16, no global traffic, no barriers -- in the two wave layouts:
//   A: 8 waves (two per SIMD), 128 x 64 wave tiles: 24 fragments (24 KB) + 64 MFMAs per wave and K-tile  -> 192 KB of LDS reads per K-tile and CU (the shipped kernel)
//   B: 4 waves (one per SIMD), 128 x 128 wave tiles, accumulators in 256 registers: 32 fragments + 128 MFMAs per wave and K-tile -> 128 KB per K-tile and CU
// Both issue 512 MFMAs per K-tile and CU.  Prints TFLOP/s by HIP events and the in-kernel clock (shader cycles per 100-MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_lds_energy_probe.hip -o build/mfma_lds_probe && ./build/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int NI, int NJ, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const short8_t* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 128 KiB of random bf16 in LDS
    for (int i = tid; i < 131072 / 16; i += WAVES * 64) reinterpret_cast<short8_t*>(smem)[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    float4_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    // per K-tile (K = 64 = two k-steps of 32): NI A fragments + NJ B fragments per k-step, each 1 KiB per wave (lane-linear: conflict-free)
    const unsigned char* base = smem + (wave % 4) * 16384 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + ((it & 1) << 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            short8_t a[NI], b[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const short8_t*>(p + ((ks * (NI + NJ) + i) & 15) * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const short8_t*>(p + ((ks * (NI + NJ) + NI + j) & 15) * 1024);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

// C: the shipped kernel's phase structure without its global traffic: a K-tile is four phases of 16 MFMAs (one 64 x 32 quadrant of the 128 x 64 wave tile x K = 64)
// with the fragment reads of pp_tile (12 / 4 / 8 / 0 per phase) in front, `s_waitcnt lgkmcnt(0)` + s_setprio around the MFMA block and an s_barrier before and after
// it; waves 4-7 run one barrier interval behind waves 0-3 (while one group issues MFMAs the other reads LDS).  SYNC = false: the same code without the barriers.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// DMA: + the shipped kernel's staging: two 1-KiB LDS-DMA pieces per wave and phase (global_load_lds, 16 B per lane; source: a 64-KiB window per workgroup, i.e.
// L2 hits) into the buffer half that is not being read, and `s_waitcnt vmcnt(8)` behind them
template <bool SYNC, bool DMA = false>
__global__ __launch_bounds__(512) void probe_pp(const short8_t* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 131072 / 16; i += 512) reinterpret_cast<short8_t*>(smem)[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    float4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned char* base = smem + (wave % 4) * 16384 + lane * 16;
    short8_t a[4][2], b0[2][2], b1[2][2];
    auto mfma16 = [&](const short8_t (&bb)[2][2], int i0, int j0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j][ks], a[i][ks], acc[i0 + i][j0 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    const char* gsrc = reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x & 255) * 65536 + lane * 16;
    int slot = 0;
    auto stage = [&](int it) {
        if (DMA) {
            unsigned char* d = smem + (((it + 1) & 1) << 16) + (slot & 3) * 16384 + wave * 2048;      // the other buffer half, region `slot`, this wave's two pieces
            const char* g = gsrc + (slot & 3) * 16384 + wave * 2048;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(g + 1024), (lptr_t)(d + 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            ++slot;
        }
    };
    auto bar = [&] { if (SYNC) __builtin_amdgcn_s_barrier(); };
    if (SYNC && wave >= 4) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + ((it & 1) << 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) { b0[j][0] = *reinterpret_cast<const short8_t*>(p + (j * 2) * 1024); b0[j][1] = *reinterpret_cast<const short8_t*>(p + (j * 2 + 1) * 1024); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i][0] = *reinterpret_cast<const short8_t*>(p + (4 + i * 2) * 1024); a[i][1] = *reinterpret_cast<const short8_t*>(p + (5 + i * 2) * 1024); }
        stage(it); bar(); mfma16(b0, 0, 0); bar();
#pragma unroll
        for (int j = 0; j < 2; ++j) { b1[j][0] = *reinterpret_cast<const short8_t*>(p + (12 + j * 2) * 1024); b1[j][1] = *reinterpret_cast<const short8_t*>(p + ((13 + j * 2) & 15) * 1024); }
        stage(it); bar(); mfma16(b1, 0, 2); bar();
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i][0] = *reinterpret_cast<const short8_t*>(p + (i * 2) * 1024 + 8192 * 0); a[i][1] = *reinterpret_cast<const short8_t*>(p + (i * 2 + 1) * 1024); }
        stage(it); bar(); mfma16(b1, 4, 2); bar();
        stage(it); bar(); mfma16(b0, 4, 0); bar();
    }
    if (SYNC && wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool SYNC, bool DMA = false>
static void run_pp(const char* name, const short8_t* src, float* out, unsigned long long* st, int iters) {
    auto k = probe_pp<SYNC, DMA>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, nullptr, src, out, st, iters);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, nullptr, src, out, st, iters);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (int b = 0; b < 256; ++b) if (h[b * 2 + 1]) { clk.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); cyc.push_back((double)h[b * 2] / iters); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 256.0 * 8 * iters * 64.0 * (2.0 * 16 * 16 * 32);
    printf("%-44s %8.2f ms per launch  %7.1f TFLOP/s  clock %.3f GHz  shader cycles per K-tile %.0f (MFMA issue floor 2048)\n", name, ms / 10,
           flop / (ms / 10 * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2]);
}

// F: B with the next k-step's fragments fetched (second register set) while the current k-step's 64 MFMAs issue: one wave per SIMD has no partner to hide its
// LDS latency behind, so the fetch has to sit inside its own MFMA stream.  DMA: + the same 64 KB of LDS-DMA staging per K-tile and CU as E (16 pieces per wave).
template <bool DMA>
__global__ __launch_bounds__(256) void probe_4w(const short8_t* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 131072 / 16; i += 256) reinterpret_cast<short8_t*>(smem)[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    float4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned char* base = smem + wave * 16384 + lane * 16;
    const char* gsrc = reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x & 255) * 65536 + lane * 16;
    short8_t a[2][8], b[2][8];
    auto fetch = [&](int set, const unsigned char* p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[set][i] = *reinterpret_cast<const short8_t*>(p + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[set][j] = *reinterpret_cast<const short8_t*>(p + (8 + j) * 1024);
    };
    fetch(0, base);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cur = ks, nxt = ks ^ 1;
            fetch(nxt, base + (((it + ks + 1) & 1) << 16));
            if (DMA) {     // 8 of the wave's 16 pieces per K-tile in each k-step
                unsigned char* d = smem + (((it + 1) & 1) << 16) + ks * 32768 + wave * 8192;
                const char* g = gsrc + ks * 32768 + wave * 8192;
#pragma unroll
                for (int q = 0; q < 8; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 1024), (lptr_t)(d + q * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0);
            if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <bool DMA>
static void run_4w(const char* name, const short8_t* src, float* out, unsigned long long* st, int iters) {
    auto k = probe_4w<DMA>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, nullptr, src, out, st, iters);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, nullptr, src, out, st, iters);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (int b = 0; b < 256; ++b) if (h[b * 2 + 1]) { clk.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); cyc.push_back((double)h[b * 2] / iters); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 256.0 * 4 * iters * 128.0 * (2.0 * 16 * 16 * 32);
    printf("%-44s %8.2f ms per launch  %7.1f TFLOP/s  clock %.3f GHz  shader cycles per K-tile %.0f (MFMA issue floor 2048)\n", name, ms / 10,
           flop / (ms / 10 * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2]);
}

template <int NI, int NJ, int WAVES>
static void run(const char* name, const short8_t* src, float* out, unsigned long long* st, int iters) {
    auto k = probe<NI, NJ, WAVES>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 131072, nullptr, src, out, st, iters);     // warm: ~0.3 s of load
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 131072, nullptr, src, out, st, iters);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int b = 0; b < 256; ++b) if (h[b * 2 + 1]) clk.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    const double flop = 256.0 * WAVES * iters * 2.0 * NI * NJ * (2.0 * 16 * 16 * 32);
    printf("%-44s %8.2f ms per launch  %7.1f TFLOP/s  clock %.3f GHz  LDS fragment reads per K-tile and CU %d KB\n", name, ms / 10, flop / (ms / 10 * 1e-3) / 1e12,
           clk[clk.size() / 2], WAVES * 2 * (NI + NJ));
}

int main() {
    short8_t* src; float* out; unsigned long long* st;
    hipMalloc(&src, (size_t)(1 << 20) * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 512 * 8);
    std::vector<unsigned short> h((size_t)(1 << 20) * 8);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 3) & 0x8000)); }     // +-[0.5, 2): random mantissas and signs
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 4, 8>("A: 8 waves, 128 x 64 wave tiles", src, out, st, iters);
        run<8, 8, 4>("B: 4 waves, 128 x 128 wave tiles (256 acc regs)", src, out, st, iters / 1);
        run_pp<true>("C: A in the shipped phase structure (barriers)", src, out, st, iters);
        run_pp<true, true>("E: C + the LDS-DMA staging (L2 hits)", src, out, st, iters);
        run_4w<false>("F: B with the next k-step prefetched", src, out, st, iters);
        run_4w<true>("G: F + the LDS-DMA staging (L2 hits)", src, out, st, iters);
    }
    return 0;
}
