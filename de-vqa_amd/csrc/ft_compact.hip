// Column compaction for the FT_VL inner loop.
//
// The gradient of the edited matrix is g = sum_r dy_r (x) a_r with a = relu(fc1(.)) -- constant over
// the loop and SPARSE: a column j whose a_r[j] is 0 for every loss row r has g = 0, hence AdamW moments
// 0 and update 0/(0+eps) = 0 at every step (weight decay 0; the L-inf clamp is a no-op there too).  The
// loop therefore only has to carry the ACTIVE columns J_e = {j : exists r, a[e,r,j] != 0} of each edit:
// state shrinks from 3 x Dout x Din to 3 x Dout x |J_e| floats and the fused AdamW sweep
// (ft_step.hip, unchanged) runs on the compacted matrices.  Exact, not an approximation.
//
//   devqa_active_columns : per edit, ascending list of active columns + count (one WG per edit,
//                          wavefront ballots + LDS prefix)
//   devqa_gather_cols_*  : out[e, row, c] = c < count[e] ? src[e, row, idx[e, c]] : 0
#include "common.h"

__global__ __launch_bounds__(1024) void active_columns_kernel(const float* __restrict__ a, int L, int Din,
                                                              int32_t* __restrict__ idx, int32_t* __restrict__ count) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int e = blockIdx.x;
    const float* ae = a + (int64_t)e * L * Din;
    int32_t* ie = idx + (int64_t)e * Din;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < Din; j0 += 1024) {
        const int j = j0 + tid;
        bool act = false;
        if (j < Din)
            for (int r = 0; r < L; ++r) act = act || (ae[(int64_t)r * Din + j] != 0.f);
        const unsigned long long bal = __ballot(act);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (act) ie[off + within] = j;
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < 16; ++w) s += wave_cnt[w];
            base += s;
        }
        __syncthreads();
    }
    if (tid == 0) count[e] = base;
}

extern "C" int devqa_active_columns(const float* a, int E, int L, int Din, int32_t* idx, int32_t* count, void* stream) {
    DEVQA_CHECK_ARG(a && idx && count, "active_columns: null pointer");
    if (E == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && L > 0 && Din > 0, "active_columns: bad dims");
    hipLaunchKernelGGL(active_columns_kernel, dim3(E), dim3(1024), 0, (hipStream_t)stream, a, L, Din, idx, count);
    DEVQA_LAUNCH_CHECK("active_columns");
    return DEVQA_OK;
}

template <typename T>
__global__ void gather_cols_kernel(const T* __restrict__ src, int64_t src_stride_e, int64_t ld_src, int rows,
                                   const int32_t* __restrict__ idx, int64_t idx_stride_e, const int32_t* __restrict__ count,
                                   int npad, T* __restrict__ out) {
    const int e = blockIdx.y;
    const int n = count[e];
    const int32_t* ie = idx + (int64_t)e * idx_stride_e;
    const T* se = src + (int64_t)e * src_stride_e;
    T* oe = out + (int64_t)e * rows * npad;
    const int64_t total = (int64_t)rows * npad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % npad);
        const int64_t r = i / npad;
        oe[i] = (c < n) ? se[r * ld_src + ie[c]] : (T)0;
    }
}

template <typename T>
static int launch_gather_cols(const T* src, int64_t src_stride_e, int64_t ld_src, int rows, const int32_t* idx,
                              int64_t idx_stride_e, const int32_t* count, int E, int npad, T* out, void* stream) {
    DEVQA_CHECK_ARG(src && idx && count && out, "gather_cols: null pointer");
    if (E == 0 || rows == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && rows > 0 && npad > 0 && ld_src > 0, "gather_cols: bad dims");
    const int64_t total = (int64_t)rows * npad;
    const int gx = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gather_cols_kernel<T>, dim3(gx, E), dim3(256), 0, (hipStream_t)stream, src, src_stride_e, ld_src, rows,
                       idx, idx_stride_e, count, npad, out);
    DEVQA_LAUNCH_CHECK("gather_cols");
    return DEVQA_OK;
}

extern "C" int devqa_gather_cols_f32(const float* src, int64_t src_stride_e, int64_t ld_src, int rows, const int32_t* idx,
                                     int64_t idx_stride_e, const int32_t* count, int E, int npad, float* out, void* stream) {
    return launch_gather_cols<float>(src, src_stride_e, ld_src, rows, idx, idx_stride_e, count, E, npad, out, stream);
}
extern "C" int devqa_gather_cols_bf16(const devqa_bf16* src, int64_t src_stride_e, int64_t ld_src, int rows,
                                      const int32_t* idx, int64_t idx_stride_e, const int32_t* count, int E, int npad,
                                      devqa_bf16* out, void* stream) {
    return launch_gather_cols<bf16_t>(src, src_stride_e, ld_src, rows, idx, idx_stride_e, count, E, npad, out, stream);
}

// dense[e or shared][row][idx[e][c]] (+)= compact[e][row][c]  for c < count[e]   (scatter a compacted delta back)
__global__ void scatter_cols_add_kernel(const float* __restrict__ comp, int rows, const int32_t* __restrict__ idx,
                                        const int32_t* __restrict__ count, int npad, float* __restrict__ dense, int64_t ld_dense) {
    const int n = count[0];
    const int64_t total = (int64_t)rows * npad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % npad);
        const int64_t r = i / npad;
        if (c < n) dense[r * ld_dense + idx[c]] += comp[i];
    }
}
extern "C" int devqa_scatter_cols_add_f32(const float* comp, int rows, const int32_t* idx, const int32_t* count, int npad,
                                          float* dense, int64_t ld_dense, void* stream) {
    DEVQA_CHECK_ARG(comp && idx && count && dense, "scatter_cols_add: null pointer");
    if (rows == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(rows > 0 && npad > 0 && ld_dense > 0, "scatter_cols_add: bad dims");
    const int64_t total = (int64_t)rows * npad;
    const int gx = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(scatter_cols_add_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, comp, rows, idx, count, npad, dense,
                       ld_dense);
    DEVQA_LAUNCH_CHECK("scatter_cols_add");
    return DEVQA_OK;
}
