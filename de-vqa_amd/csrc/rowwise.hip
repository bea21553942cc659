// Row-wise / elementwise kernels: LayerNorm fwd, LayerNorm bwd (dx), patch im2col, ViT embedding
// assembly, token-embedding gather + OPT positions, row gather, casts, delta ops, FT loop control.
// All are HBM-bound: 16-byte (float4 / 8xbf16) accesses, one wave per row where a row reduction
// is needed (wavefront shuffles, no LDS), grids capped and grid-strided.
#include <stdarg.h>
#include "common.h"

thread_local char g_devqa_err[512] = {0};
int devqa_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_devqa_err, sizeof(g_devqa_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* devqa_last_error(void) { return g_devqa_err; }
extern "C" int devqa_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, row held in registers (D <= 64*4*MAXV).
// ------------------------------------------------------------------------------------------
#define LN_MAXV 16
typedef __bf16 ln_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float ln_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ln_pack2(float a, float b) {   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    ln_f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, ln_bf16x2_t));
}

// One wave per row, RPW rows per wave with ALL their loads issued before the first reduction (bytes in flight are what an
// HBM stream needs); NV = float4 pieces per lane (D <= 256 * NV), a template parameter so that no predicated iterations are
// left.  Two-pass statistics from registers (mean, then centred sum of squares), as nn.LayerNorm.
template <int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int M, int D, float eps, bf16_t* __restrict__ out_bf16,
                                                        float* __restrict__ out_f32) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    const int nv = D >> 2;
    float4 v[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, M - 1);     // a clamped duplicate row is computed and not stored
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            v[r][i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (add) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float4* ar = reinterpret_cast<const float4*>(add + (int64_t)min(row0 + r, M - 1) * D);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 64 + lane;
                if (c < nv) {
                    const float4 u = ar[c];
                    v[r][i].x += u.x; v[r][i].y += u.y; v[r][i].z += u.z; v[r][i].w += u.w;
                }
            }
        }
    }
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);   // padding pieces are zero
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i * 64 + lane < nv) {
                const float a = v[r][i].x - mean, b = v[r][i].y - mean, cc = v[r][i].z - mean, d = v[r][i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        const int row = row0 + r;
        if (row >= M) continue;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                const float4 g = g4[c], b = b4[c];
                float4 o;
                o.x = (v[r][i].x - mean) * rstd * g.x + b.x;
                o.y = (v[r][i].y - mean) * rstd * g.y + b.y;
                o.z = (v[r][i].z - mean) * rstd * g.z + b.z;
                o.w = (v[r][i].w - mean) * rstd * g.w + b.w;
                if (out_f32) reinterpret_cast<float4*>(out_f32 + (int64_t)row * D)[c] = o;
                if (out_bf16) {
                    uint2 p;
                    p.x = ln_pack2(o.x, o.y);
                    p.y = ln_pack2(o.z, o.w);
                    reinterpret_cast<uint2*>(out_bf16 + (int64_t)row * D)[c] = p;
                }
            }
        }
    }
}

template <int NV, int RPW>
static void launch_layernorm(const float* x, const float* add, const float* gamma, const float* beta, int M, int D, float eps,
                             bf16_t* out_bf16, float* out_f32, hipStream_t st) {
    hipLaunchKernelGGL((layernorm_kernel<NV, RPW>), dim3((M + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, st, x, add, gamma, beta, M, D,
                       eps, out_bf16, out_f32);
}

extern "C" int devqa_layernorm(const float* x, const float* add, const float* gamma, const float* beta, int M, int D,
                               float eps, devqa_bf16* out_bf16, float* out_f32, void* stream) {
    DEVQA_CHECK_ARG(x && gamma && beta && (out_bf16 || out_f32), "layernorm: null pointer");
    if (M == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "layernorm: D=%d unsupported", D);
    hipStream_t st = (hipStream_t)stream;
#define LN_ARGS x, add, gamma, beta, M, D, eps, out_bf16, out_f32, st
    const int nvl = (D / 4 + 63) / 64;      // float4 pieces per lane
    const int ph = devqa_prof_begin(DEVQA_PROF_LAYERNORM, st);
    if (nvl <= 1) launch_layernorm<1, 4>(LN_ARGS);
    else if (nvl <= 2) launch_layernorm<2, 4>(LN_ARGS);
    else if (nvl <= 3) launch_layernorm<3, 2>(LN_ARGS);
    else if (nvl <= 4) launch_layernorm<4, 2>(LN_ARGS);
    else if (nvl <= 6) launch_layernorm<6, 2>(LN_ARGS);
    else if (nvl <= 8) launch_layernorm<8, 2>(LN_ARGS);
    else if (nvl <= 10) launch_layernorm<10, 2>(LN_ARGS);
    else launch_layernorm<16, 1>(LN_ARGS);
#undef LN_ARGS
    devqa_prof_end(ph, (double)M * D * (4.0 + (add ? 4.0 : 0.0) + (out_bf16 ? 2.0 : 0.0) + (out_f32 ? 4.0 : 0.0)), st);
    DEVQA_LAUNCH_CHECK("layernorm");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. input:  xhat=(x-mean)*rstd, gy=dy*gamma,
//   dx = rstd * (gy - mean(gy) - xhat*mean(gy*xhat))
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ dy, int M, int D, float eps,
                                                               float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* dr = reinterpret_cast<const float4*>(dy + (int64_t)row * D);
    const float4* ar = add ? reinterpret_cast<const float4*>(add + (int64_t)row * D) : nullptr;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 v[LN_MAXV], gy[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            v[i] = xr[c];
            if (ar) {
                const float4 u = ar[c];
                v[i].x += u.x; v[i].y += u.y; v[i].z += u.z; v[i].w += u.w;
            }
            const float4 d = dr[c], g = g4[c];
            gy[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            gy[i] = v[i];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 += (gy[i].x * v[i].x + gy[i].y * v[i].y) + (gy[i].z * v[i].z + gy[i].w * v[i].w);
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            float4 o;
            o.x = rstd * (gy[i].x - m1 - v[i].x * m2);
            o.y = rstd * (gy[i].y - m1 - v[i].y * m2);
            o.z = rstd * (gy[i].z - m1 - v[i].z * m2);
            o.w = rstd * (gy[i].w - m1 - v[i].w * m2);
            reinterpret_cast<float4*>(dx + (int64_t)row * D)[c] = o;
        }
    }
}

extern "C" int devqa_layernorm_bwd_dx(const float* x, const float* add, const float* gamma, const float* dy, int M, int D,
                                      float eps, float* dx, void* stream) {
    DEVQA_CHECK_ARG(x && gamma && dy && dx, "layernorm_bwd_dx: null pointer");
    if (M == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "layernorm_bwd_dx: D=%d unsupported", D);
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add, gamma, dy, M,
                       D, eps, dx);
    DEVQA_LAUNCH_CHECK("layernorm_bwd_dx");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. its parameters (full fine-tuning: LTE_VL training, R/editor/vllm_editors/lte_vl/lte_vl.py:207-233 runs
// autograd through every nn.LayerNorm of the language model):  dgamma[c] (+)= sum_r dy[r,c] * xhat[r,c],  dbeta[c] (+)= sum_r dy[r,c].
// Two launches, both deterministic (no atomics): per-row (mean, rstd) by one wave per row into `stats` [M, 2]; then one thread per
// column and quarter of the rows walks the rows in order (coalesced along the columns), the four partial sums combine in LDS in a
// fixed order.  Also the column sums of a matrix (bias gradients) as the same second kernel without the xhat factor.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, const float* __restrict__ add, int M, int D, float eps,
                                                        int rms, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    const float* ar = add ? add + (int64_t)row * D : nullptr;
    float s = 0.f;
    if (!rms)
        for (int c = lane; c < D; c += 64) s += xr[c] + (ar ? ar[c] : 0.f);
    const float mean = rms ? 0.f : wave_sum(s) / (float)D;      // RMSNorm: xhat = x * rsqrt(mean(x^2) + eps)
    float q = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = xr[c] + (ar ? ar[c] : 0.f) - mean;
        q += v * v;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

template <bool LN>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                         const float* __restrict__ dy, const float* __restrict__ stats, int M, int D,
                                                         int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float g = 0.f, b = 0.f;
    if (c < D) {
        for (int r = ry; r < M; r += 4) {
            const float d = dy[(int64_t)r * D + c];
            b += d;
            if (LN) {
                const float v = x[(int64_t)r * D + c] + (add ? add[(int64_t)r * D + c] : 0.f);
                g += d * (v - stats[2 * r]) * stats[2 * r + 1];
            }
        }
    }
    red[0][ry][cx] = g;
    red[1][ry][cx] = b;
    __syncthreads();
    if (ry == 0 && c < D) {
        const float bs = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + bs;
        if (LN) {
            const float gs = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
            dgamma[c] = (accumulate ? dgamma[c] : 0.f) + gs;
        }
    }
}

extern "C" int devqa_layernorm_bwd_params(const float* x, const float* add, const float* dy, int M, int D, float eps, int rms, int accumulate,
                                          float* dgamma, float* dbeta, float* stats_ws, void* stream) {
    DEVQA_CHECK_ARG(x && dy && dgamma && stats_ws && (dbeta || rms), "layernorm_bwd_params: null pointer");
    DEVQA_CHECK_SHAPE(M >= 0 && D > 0, "layernorm_bwd_params: M=%d D=%d", M, D);
    if (!dbeta) dbeta = stats_ws + 2 * (int64_t)(M > 0 ? M : 1);      // RMSNorm has no shift: the column sums go to scratch
    if (M > 0) {
        hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add, M, D, eps, rms, stats_ws);
        DEVQA_LAUNCH_CHECK("layernorm_bwd_params(stats)");
    }
    hipLaunchKernelGGL(col_reduce_kernel<true>, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, add, dy, stats_ws, M, D, accumulate,
                       dgamma, dbeta);
    DEVQA_LAUNCH_CHECK("layernorm_bwd_params");
    return DEVQA_OK;
}

extern "C" int devqa_colsum_f32(const float* x, int M, int D, int accumulate, float* out, void* stream) {
    DEVQA_CHECK_ARG(x && out, "colsum: null pointer");
    DEVQA_CHECK_SHAPE(M >= 0 && D > 0, "colsum: M=%d D=%d", M, D);
    hipLaunchKernelGGL(col_reduce_kernel<false>, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, x, nullptr, M, D,
                       accumulate, nullptr, out);
    DEVQA_LAUNCH_CHECK("colsum");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
// im2col for the patch-embedding conv: out[b*np + (py*G+px)][(c*P+ky)*P+kx] (bf16), zero pad
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_elem(bf16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void store_elem(float* p, float v) { *p = v; }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    const uint2 e = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_to_f32(e.x & 0xffff), bf16_to_f32(e.x >> 16), bf16_to_f32(e.y & 0xffff), bf16_to_f32(e.y >> 16));
}
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <typename T>
__global__ void im2col_kernel(const float* __restrict__ pix, int B, int S, int P, int Kpad, T* __restrict__ out) {
    const int G = S / P, np = G * G, Kreal = 3 * P * P;
    const int64_t total = (int64_t)B * np * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kpad);
        const int64_t rowi = i / Kpad;
        float v = 0.f;
        if (col < Kreal) {
            const int p = (int)(rowi % np), b = (int)(rowi / np);
            const int c = col / (P * P), rem = col % (P * P), ky = rem / P, kx = rem % P;
            const int y = (p / G) * P + ky, x = (p % G) * P + kx;
            v = pix[(((int64_t)b * 3 + c) * S + y) * S + x];
        }
        store_elem(out + i, v);
    }
}
template <typename T>
static int launch_im2col(const float* pixels, int B, int S, int P, int Kpad, T* out, void* stream) {
    DEVQA_CHECK_ARG(pixels && out, "im2col: null pointer");
    if (B == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(B > 0 && S > 0 && P > 0 && S % P == 0 && Kpad >= 3 * P * P && Kpad % 8 == 0, "im2col: bad shape");
    const int64_t total = (int64_t)B * (S / P) * (S / P) * Kpad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(im2col_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pixels, B, S, P, Kpad, out);
    DEVQA_LAUNCH_CHECK("im2col");
    return DEVQA_OK;
}
extern "C" int devqa_im2col_patches(const float* pixels, int B, int S, int P, int Kpad, devqa_bf16* out, void* stream) {
    return launch_im2col<bf16_t>(pixels, B, S, P, Kpad, out, stream);
}
extern "C" int devqa_im2col_patches_f32(const float* pixels, int B, int S, int P, int Kpad, float* out, void* stream) {
    return launch_im2col<float>(pixels, B, S, P, Kpad, out, stream);
}

__global__ void vit_assemble_kernel(const float* __restrict__ patches, const float* __restrict__ cls,
                                    const float* __restrict__ pos, int B, int np, int D, float* __restrict__ out) {
    const int nv = D >> 2;
    const int64_t total = (int64_t)B * (np + 1) * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv);
        const int64_t r = i / nv;
        const int t = (int)(r % (np + 1)), b = (int)(r / (np + 1));
        float4 v = (t == 0) ? reinterpret_cast<const float4*>(cls)[c]
                            : reinterpret_cast<const float4*>(patches + ((int64_t)b * np + (t - 1)) * D)[c];
        const float4 p = reinterpret_cast<const float4*>(pos + (int64_t)t * D)[c];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
extern "C" int devqa_vit_assemble(const float* patches, const float* cls, const float* pos, int B, int np, int D,
                                  float* out, void* stream) {
    DEVQA_CHECK_ARG(patches && cls && pos && out, "vit_assemble: null pointer");
    if (B == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(B > 0 && np > 0 && D > 0 && D % 4 == 0, "vit_assemble: bad shape");
    const int64_t total = (int64_t)B * (np + 1) * (D / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, patches, cls, pos, B, np, D, out);
    DEVQA_LAUNCH_CHECK("vit_assemble");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
// token embedding / image-token rows + OPT learned positions (offset 2)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_rows_kernel(const int32_t* __restrict__ token, const int32_t* __restrict__ src_row,
                                  const int32_t* __restrict__ pos, const T* __restrict__ embed,
                                  const float* __restrict__ rows_f32, const T* __restrict__ pos_table, int R, int D,
                                  int V, int n_rows_f32, int n_pos, float* __restrict__ out) {
    const int nv = D >> 2;
    const int64_t total = (int64_t)R * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv), r = (int)(i / nv);
        float4 v;
        const int sr = src_row ? src_row[r] : -1;
        if (sr >= 0) {
            v = (sr < n_rows_f32) ? reinterpret_cast<const float4*>(rows_f32 + (int64_t)sr * D)[c]
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            int t = token[r];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            v = load4(embed + (int64_t)t * D + c * 4);
        }
        int p = pos[r] + 2;
        p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
        const float4 e = load4(pos_table + (int64_t)p * D + c * 4);
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
template <typename T>
static int launch_embed_rows(const int32_t* token, const int32_t* src_row, const int32_t* pos, const T* embed,
                             const float* rows_f32, const T* pos_table, int R, int D, int V, int n_rows_f32, int n_pos,
                             float* out, void* stream) {
    DEVQA_CHECK_ARG(token && pos && embed && pos_table && out, "embed_rows: null pointer");
    DEVQA_CHECK_ARG(!src_row || rows_f32 || n_rows_f32 == 0, "embed_rows: src_row given without rows_f32");
    if (R == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(R > 0 && D > 0 && D % 4 == 0 && V > 0 && n_pos > 0, "embed_rows: bad shape");
    const int64_t total = (int64_t)R * (D / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(embed_rows_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, token, src_row, pos, embed,
                       rows_f32, pos_table, R, D, V, n_rows_f32, n_pos, out);
    DEVQA_LAUNCH_CHECK("embed_rows");
    return DEVQA_OK;
}
extern "C" int devqa_embed_rows(const int32_t* token, const int32_t* src_row, const int32_t* pos, const devqa_bf16* embed,
                                const float* rows_f32, const devqa_bf16* pos_table, int R, int D, int V, int n_rows_f32,
                                int n_pos, float* out, void* stream) {
    return launch_embed_rows<bf16_t>(token, src_row, pos, embed, rows_f32, pos_table, R, D, V, n_rows_f32, n_pos, out, stream);
}
extern "C" int devqa_embed_rows_f32(const int32_t* token, const int32_t* src_row, const int32_t* pos, const float* embed,
                                    const float* rows_f32, const float* pos_table, int R, int D, int V, int n_rows_f32,
                                    int n_pos, float* out, void* stream) {
    return launch_embed_rows<float>(token, src_row, pos, embed, rows_f32, pos_table, R, D, V, n_rows_f32, n_pos, out, stream);
}

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ in, const int32_t* __restrict__ idx, int R, int nv,
                                   T* __restrict__ out) {
    const int64_t total = (int64_t)R * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv), r = (int)(i / nv);
        out[i] = in[(int64_t)idx[r] * nv + c];
    }
}
extern "C" int devqa_gather_rows(const void* in, const int32_t* idx, int R, int D, int elem_bytes, void* out, void* stream) {
    DEVQA_CHECK_ARG(in && idx && out, "gather_rows: null pointer");
    DEVQA_CHECK_ARG(elem_bytes == 2 || elem_bytes == 4, "gather_rows: elem_bytes must be 2 or 4");
    if (R == 0) return DEVQA_OK;
    const int row_bytes = D * elem_bytes;
    DEVQA_CHECK_SHAPE(R > 0 && D > 0 && row_bytes % 16 == 0, "gather_rows: row bytes must be a multiple of 16");
    const int nv = row_bytes / 16;
    const int64_t total = (int64_t)R * nv;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(in), idx, R, nv, reinterpret_cast<uint4*>(out));
    DEVQA_LAUNCH_CHECK("gather_rows");
    return DEVQA_OK;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        uint2 p;
        p.x = pack_bf16x2(v.x, v.y);
        p.y = pack_bf16x2(v.z, v.w);
        reinterpret_cast<uint2*>(out)[i] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = f32_to_bf16(in[n4 * 4 + threadIdx.x]);
}
extern "C" int devqa_cast_f32_bf16(const float* in, devqa_bf16* out, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(in && out && n >= 0, "cast: bad args");
    if (n == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "cast: misaligned");
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, (hipStream_t)stream, in, out, n4, n);
    DEVQA_LAUNCH_CHECK("cast_f32_bf16");
    return DEVQA_OK;
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits of x in two bf16 operands (the three-product form of an fp32 GEMM on the
// bf16 MFMA: A.W ~ A_hi.W_hi + A_hi.W_lo + A_lo.W_hi, fp32 accumulation; the dropped A_lo.W_lo term and the tails are <= 2^-16 relative)
__global__ void split_f32_bf16x2_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        const float x[4] = {v.x, v.y, v.z, v.w};
        bf16_t h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = f32_to_bf16(x[k]);
            l[k] = f32_to_bf16(x[k] - bf16_to_f32(h[k]));
        }
        reinterpret_cast<uint2*>(hi)[i] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        reinterpret_cast<uint2*>(lo)[i] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const bf16_t h = f32_to_bf16(in[i]);
        hi[i] = h;
        lo[i] = f32_to_bf16(in[i] - bf16_to_f32(h));
    }
}

extern "C" int devqa_split_f32_bf16x2(const float* in, devqa_bf16* hi, devqa_bf16* lo, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(in && hi && lo, "split_f32_bf16x2: null pointer");
    if (n <= 0) return DEVQA_OK;
    DEVQA_CHECK_ARG(((((uintptr_t)in) & 15) | (((uintptr_t)hi) & 7) | (((uintptr_t)lo) & 7)) == 0, "split_f32_bf16x2: misaligned pointer");
    const int64_t n4 = n >> 2;
    int64_t grid = (n4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(split_f32_bf16x2_kernel, dim3(grid < 1 ? 1 : (unsigned)grid), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)hi, (bf16_t*)lo, n4, n);
    DEVQA_LAUNCH_CHECK("split_f32_bf16x2");
    return DEVQA_OK;
}

// out = act(in) on fp32 pre-activations (act: DEVQA_ACT_NONE / DEVQA_ACT_RELU), written as bf16 and / or fp32.  Used where a low-rank
// term has to enter BEFORE the activation (MEND_VL's edited fc1: relu(h W^T + b + (h x~^T) d~), mend_vl.py:72-79), which the GEMM
// epilogue (residual after the activation) cannot express: the GEMM leaves fp32 pre-activations, this pass finishes them.
__global__ void act_cast_kernel(const float* __restrict__ in, int relu, bf16_t* __restrict__ ob, float* __restrict__ of, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ob) {
            uint2 p;
            p.x = pack_bf16x2(v.x, v.y);
            p.y = pack_bf16x2(v.z, v.w);
            reinterpret_cast<uint2*>(ob)[i] = p;
        }
        if (of) reinterpret_cast<float4*>(of)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        float v = in[n4 * 4 + threadIdx.x];
        if (relu) v = fmaxf(v, 0.f);
        if (ob) ob[n4 * 4 + threadIdx.x] = f32_to_bf16(v);
        if (of) of[n4 * 4 + threadIdx.x] = v;
    }
}
extern "C" int devqa_act_cast(const float* in, int act, devqa_bf16* out_bf16, float* out_f32, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(in && (out_bf16 || out_f32) && n >= 0 && (act == 0 || act == 1), "act_cast: bad args");
    if (n == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0 && (((uintptr_t)out_f32) & 15) == 0, "act_cast: misaligned");
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(act_cast_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, (hipStream_t)stream, in, act, out_bf16, out_f32, n4, n);
    DEVQA_LAUNCH_CHECK("act_cast");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
__global__ void delta_op_kernel(int mode, float* __restrict__ w, const float* __restrict__ w0, float* __restrict__ delta,
                                int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        if (mode == 0) {
            const float4 a = reinterpret_cast<const float4*>(w)[i], b = reinterpret_cast<const float4*>(w0)[i];
            reinterpret_cast<float4*>(delta)[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        } else if (mode == 1) {
            float4 a = reinterpret_cast<float4*>(w)[i];
            const float4 d = reinterpret_cast<const float4*>(delta)[i];
            a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
            reinterpret_cast<float4*>(w)[i] = a;
        } else {
            reinterpret_cast<float4*>(w)[i] = reinterpret_cast<const float4*>(w0)[i];
        }
    }
}
extern "C" int devqa_delta_op(int mode, float* w, const float* w0, float* delta, int64_t n, void* stream) {
    DEVQA_CHECK_ARG(mode >= 0 && mode <= 2 && w, "delta_op: bad args");
    DEVQA_CHECK_ARG((mode == 1) ? (delta != nullptr) : (w0 != nullptr), "delta_op: missing operand");
    DEVQA_CHECK_ARG(mode != 0 || delta, "delta_op: missing delta");
    if (n == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(n > 0 && n % 4 == 0, "delta_op: n must be a multiple of 4");
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(delta_op_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, w, w0, delta, n4);
    DEVQA_LAUNCH_CHECK("delta_op");
    return DEVQA_OK;
}

// ------------------------------------------------------------------------------------------
// FT loop control (one thread per edit)
// ------------------------------------------------------------------------------------------
__global__ void ft_step_control_kernel(const float* __restrict__ nll, const float* __restrict__ mask, int E, int Lmax,
                                       int step, int max_steps, float floor_, int32_t* active, int32_t* do_update,
                                       int32_t* n_steps, int32_t* adam_t, float* losses) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (!active[e]) {
        do_update[e] = 0;
        return;
    }
    float s = 0.f, c = 0.f;
    for (int r = 0; r < Lmax; ++r) {
        const float mk = mask[e * Lmax + r];
        if (mk != 0.f) s += nll[e * Lmax + r] * mk;
        c += mk;
    }
    const float loss = s / c;
    losses[(int64_t)e * max_steps + step] = loss;
    n_steps[e] = step + 1;
    const bool upd = loss >= floor_;  // ft_vl.py:131 (NaN: no update, no break -- same as the reference)
    do_update[e] = upd ? 1 : 0;
    if (upd) adam_t[e] += 1;          // torch.optim.AdamW step counter (1-based t of this update)
    if (loss < floor_) active[e] = 0;  // ft_vl.py:145-146: loop breaks after this step
}
extern "C" int devqa_ft_step_control(const float* nll, const float* mask, int E, int Lmax, int step, int max_steps,
                                     float floor_, int32_t* active, int32_t* do_update, int32_t* n_steps, int32_t* adam_t,
                                     float* losses, void* stream) {
    DEVQA_CHECK_ARG(nll && mask && active && do_update && n_steps && adam_t && losses, "ft_step_control: null pointer");
    if (E == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && Lmax > 0 && step >= 0 && step < max_steps, "ft_step_control: bad shape");
    hipLaunchKernelGGL(ft_step_control_kernel, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, nll, mask, E, Lmax,
                       step, max_steps, floor_, active, do_update, n_steps, adam_t, losses);
    DEVQA_LAUNCH_CHECK("ft_step_control");
    return DEVQA_OK;
}
