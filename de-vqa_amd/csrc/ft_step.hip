// FT_VL inner-step kernels on the edited matrix W [Dout, Din] (fp32 master copies, one per
// in-flight edit).  Both kernels are HBM streams over the matrix with 16-byte accesses:
//
//   ft_adamw_step : rank-L gradient (built on the fly from dy and a; never written to HBM)
//                   -> AdamW (torch.optim.AdamW semantics) -> optional L-inf clamp
//                   -> y = W_new . a for the next step's forward, all in one sweep.
//                   Algorithmic bytes per edit-step: 6 x 4 x Dout x Din (r/w of w, m, v).
//   ..._fm        : the same step WITHOUT a first-moment matrix (FM = factored momentum).  Inside one edit's loop the a-rows are constant
//                   (everything below the edited matrix is frozen), so every gradient is dy_t^T a and the first moment is
//                   m_t = D_t^T a with D_t = lerp(D_{t-1}, dy_t, 1 - beta1): an [L, Dout] state instead of a [Dout, Din] matrix.  m_t is
//                   rebuilt per element from D_t and the a-values the sweep holds anyway (L fmas); only w and v cross HBM:
//                   4 x 4 x Dout x Din bytes per edit-step instead of 6 x 4 (first update: read w0, write w, v).  The second moment
//                   cannot be factored the same way in fp32 (sums of squares of sums cancel), it stays a matrix.
//   rows_matvec   : y = W . a (+bias +resid) for a few cached rows.
//
// Work decomposition: one workgroup (256 threads) owns ROWS consecutive output rows i of one
// edit; threads stride over Din in float4 units, so each wave-instruction touches 1 KiB of
// contiguous HBM.  The a-rows (L x Din fp32, <= 320 KB) are re-read by every workgroup and stay
// L2/MALL resident.  Per-row dot products are reduced with wavefront shuffles + one LDS hop.
#include <stdlib.h>
#include "common.h"

// NARROW (column-compacted matrices, Din of a few hundred): every WAVE owns its own ROWS rows (the workgroup 4 x ROWS) and
// strides Din with its 64 lanes -- with 256 threads on one row group only Din/4 of them would have a float4 to work on.
template <int L, int ROWS, bool NARROW = false, bool FM = false>
__global__ __launch_bounds__(256) void ft_adamw_step_kernel(float* __restrict__ w, float* __restrict__ m,
                                                            float* __restrict__ v, const float* __restrict__ w0,
                                                            const float* __restrict__ a, const float* __restrict__ dy,
                                                            float* __restrict__ y, const int32_t* __restrict__ do_update,
                                                            const int32_t* __restrict__ adam_t, const int32_t* __restrict__ single, int Lmax, int Dout, int Din,
                                                            float lr, float beta1, float beta2, float eps, float wd,
                                                            float clamp_eps, int row_blocks, int64_t w0_stride_e) {
    __shared__ float red[4][ROWS * L];
    const int e = blockIdx.x / row_blocks;
    const int rb = blockIdx.x % row_blocks;
    if (!do_update[e]) return;  // uniform
    w0 += (int64_t)e * w0_stride_e;
    const int t = adam_t[e];
    const bool first = (t <= 1);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    const float decay = 1.f - lr * wd;

    const int i0 = NARROW ? (rb * 4 + (int)(threadIdx.x >> 6)) * ROWS : rb * ROWS;
    const int64_t mat = (int64_t)Dout * Din;
    float* we = w + (int64_t)e * mat;
    float* me = FM ? m + (int64_t)e * (Lmax + 1) * Dout : m + (int64_t)e * mat;       // FM: the [Lmax + 1, Dout] state: D (laid out like dy) | e (SGL)
    float* ve = v + (int64_t)e * mat;
    const float* ae = a + (int64_t)e * Lmax * Din;
    const float* dye = dy + (int64_t)e * Lmax * Dout;
    // SGL (single[e] != 0: only loss row 0 of this edit is ever non-zero): the second moment factors too, v_t[i, j] = e_t[i] a[0, j]^2 with
    // e_t = beta2 e_{t-1} + (1 - beta2) dy_t[0, i]^2 -- one term, nothing cancels -- so only w crosses HBM (8 bytes per element and step)
    const bool sgl = FM && single != nullptr && single[e] != 0;

    float dyv[ROWS][L];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int l = 0; l < L; ++l) dyv[r][l] = (i0 + r < Dout && l < Lmax) ? dye[(int64_t)l * Dout + i0 + r] : 0.f;
    float dn[FM ? ROWS : 1][FM ? L : 1];      // FM: D_t of this workgroup's (wave's) rows
    float en[FM ? ROWS : 1];                  // SGL: e_t of these rows
    if constexpr (FM) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float dold = (!first && i0 + r < Dout && l < Lmax) ? me[(int64_t)l * Dout + i0 + r] : 0.f;
                dn[r][l] = dold + (dyv[r][l] - dold) * (1.f - beta1);
            }
            const float eold = (sgl && !first && i0 + r < Dout) ? me[(int64_t)Lmax * Dout + i0 + r] : 0.f;
            en[r] = eold * beta2 + (1.f - beta2) * dyv[r][0] * dyv[r][0];
        }
    }

    float ysum[ROWS][L];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int l = 0; l < L; ++l) ysum[r][l] = 0.f;

    const int nv = Din >> 2;
    for (int c = NARROW ? (int)(threadIdx.x & 63) : (int)threadIdx.x; c < nv; c += (NARROW ? 64 : 256)) {
        float4 av[L];
#pragma unroll
        for (int l = 0; l < L; ++l)
            av[l] = (l < Lmax) ? reinterpret_cast<const float4*>(ae + (int64_t)l * Din)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = i0 + r;
            if (i >= Dout) continue;
            const int64_t off = (int64_t)i * nv + c;
            float4 wv, mv = make_float4(0.f, 0.f, 0.f, 0.f), vv, w0v;
            if (first) {
                w0v = reinterpret_cast<const float4*>(w0)[off];
                wv = w0v;
                vv = mv;
            } else {
                wv = reinterpret_cast<const float4*>(we)[off];
                if constexpr (!FM) mv = reinterpret_cast<const float4*>(me)[off];
                vv = sgl ? mv : reinterpret_cast<const float4*>(ve)[off];
                if (clamp_eps >= 0.f) w0v = reinterpret_cast<const float4*>(w0)[off];
            }
            float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < L; ++l) {
                g[0] += dyv[r][l] * av[l].x;
                g[1] += dyv[r][l] * av[l].y;
                g[2] += dyv[r][l] * av[l].z;
                g[3] += dyv[r][l] * av[l].w;
            }
            float wq[4] = {wv.x, wv.y, wv.z, wv.w};
            float mq[4] = {mv.x, mv.y, mv.z, mv.w};
            float vq[4] = {vv.x, vv.y, vv.z, vv.w};
            const float w0q[4] = {w0v.x, w0v.y, w0v.z, w0v.w};
            if constexpr (FM) {           // m_t = D_t^T a (the lerp was taken on D)
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    mq[0] += dn[r][l] * av[l].x;
                    mq[1] += dn[r][l] * av[l].y;
                    mq[2] += dn[r][l] * av[l].z;
                    mq[3] += dn[r][l] * av[l].w;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // torch.optim.AdamW (single-tensor path): decay, lerp m, addcmul v, addcdiv
                wq[k] *= decay;
                if constexpr (!FM) mq[k] = mq[k] + (g[k] - mq[k]) * (1.f - beta1);
                const float a0k = k == 0 ? av[0].x : k == 1 ? av[0].y : k == 2 ? av[0].z : av[0].w;
                vq[k] = (FM && sgl) ? en[FM ? r : 0] * (a0k * a0k) : vq[k] * beta2 + (1.f - beta2) * g[k] * g[k];
                const float denom = sqrtf(vq[k]) / bc2_sqrt + eps;
                wq[k] -= step_size * (mq[k] / denom);
                if (clamp_eps >= 0.f) wq[k] = fminf(fmaxf(wq[k], w0q[k] - clamp_eps), w0q[k] + clamp_eps);
            }
            reinterpret_cast<float4*>(we)[off] = make_float4(wq[0], wq[1], wq[2], wq[3]);
            if constexpr (!FM) reinterpret_cast<float4*>(me)[off] = make_float4(mq[0], mq[1], mq[2], mq[3]);
            if (!sgl) reinterpret_cast<float4*>(ve)[off] = make_float4(vq[0], vq[1], vq[2], vq[3]);
#pragma unroll
            for (int l = 0; l < L; ++l)
                ysum[r][l] += (wq[0] * av[l].x + wq[1] * av[l].y) + (wq[2] * av[l].z + wq[3] * av[l].w);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (NARROW) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float s = wave_sum(ysum[r][l]);
                if (lane == 0 && i0 + r < Dout && l < Lmax) {
                    y[((int64_t)e * Lmax + l) * Dout + i0 + r] = s;
                    if constexpr (FM) {
                        me[(int64_t)l * Dout + i0 + r] = dn[r][l];      // (every lane of the wave read the old value at entry)
                        if (sgl && l == 0) me[(int64_t)Lmax * Dout + i0 + r] = en[r];
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float s = wave_sum(ysum[r][l]);
            if (lane == 0) red[wave][r * L + l] = s;
        }
    __syncthreads();
    if (threadIdx.x < ROWS * L) {
        const int r = threadIdx.x / L, l = threadIdx.x % L;
        if (i0 + r < Dout && l < Lmax) {
            const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            y[((int64_t)e * Lmax + l) * Dout + i0 + r] = s;
            if constexpr (FM) {       // behind the barrier: every thread read the old D at entry
                const float dold = first ? 0.f : me[(int64_t)l * Dout + i0 + r];
                const float dyl = dye[(int64_t)l * Dout + i0 + r];
                me[(int64_t)l * Dout + i0 + r] = dold + (dyl - dold) * (1.f - beta1);
                if (sgl && l == 0) {
                    const float eold = first ? 0.f : me[(int64_t)Lmax * Dout + i0 + r];
                    me[(int64_t)Lmax * Dout + i0 + r] = eold * beta2 + (1.f - beta2) * dyl * dyl;
                }
            }
        }
    }
}

// GROUPED (column-compacted matrices whose row is not a multiple of 64 float4: npad = 272 -> 68 float4 left the wave-per-row
// form above a second pass with 4 of 64 lanes active, 53 % of the issued memory instructions doing work -- the 4.4 of 8 TB/s the
// bench line reported): G lanes own a row (G = 16 or 8: 256- or 128-byte contiguous pieces per row and step), a wave 64 / G rows,
// a workgroup 4 x 64 / G; ceil(nv / G) steps cover a row with at most G - 1 idle lane-steps (68 float4: 9 steps of 8, 94 %).
// The per-row dot products for the next forward reduce inside the G-lane group.  Same per-element arithmetic as above.
template <int L, int G, bool FM = false>
__global__ __launch_bounds__(256) void ft_adamw_step_grouped_kernel(float* __restrict__ w, float* __restrict__ m,
                                                                    float* __restrict__ v, const float* __restrict__ w0,
                                                                    const float* __restrict__ a, const float* __restrict__ dy,
                                                                    float* __restrict__ y, const int32_t* __restrict__ do_update,
                                                                    const int32_t* __restrict__ adam_t, const int32_t* __restrict__ single, int Lmax, int Dout, int Din,
                                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                                    float clamp_eps, int row_blocks, int64_t w0_stride_e) {
    constexpr int RW = 64 / G;      // rows per wave
    const int e = blockIdx.x / row_blocks;
    const int rb = blockIdx.x % row_blocks;
    if (!do_update[e]) return;  // uniform
    w0 += (int64_t)e * w0_stride_e;
    const int t = adam_t[e];
    const bool first = (t <= 1);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    const float decay = 1.f - lr * wd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G;
    const int i = (rb * 4 + wave) * RW + lane / G;        // this lane group's row
    const bool row_ok = i < Dout;
    const int ic = row_ok ? i : Dout - 1;                  // idle groups shadow the last row (loads only)
    const int64_t mat = (int64_t)Dout * Din;
    float* we = w + (int64_t)e * mat;
    float* me = FM ? m + (int64_t)e * (Lmax + 1) * Dout : m + (int64_t)e * mat;       // FM: the [Lmax + 1, Dout] state: D (laid out like dy) | e (SGL)
    float* ve = v + (int64_t)e * mat;
    const float* ae = a + (int64_t)e * Lmax * Din;
    const float* dye = dy + (int64_t)e * Lmax * Dout;
    const bool sgl = FM && single != nullptr && single[e] != 0;        // see ft_adamw_step_kernel
    float dyv[L], ysum[L], dn[FM ? L : 1];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        dyv[l] = (l < Lmax) ? dye[(int64_t)l * Dout + ic] : 0.f;
        ysum[l] = 0.f;
        if constexpr (FM) {
            const float dold = (!first && l < Lmax) ? me[(int64_t)l * Dout + ic] : 0.f;
            dn[l] = dold + (dyv[l] - dold) * (1.f - beta1);
        }
    }
    float en = 0.f;
    if (sgl) {
        const float eold = first ? 0.f : me[(int64_t)Lmax * Dout + ic];
        en = eold * beta2 + (1.f - beta2) * dyv[0] * dyv[0];
    }
    const int nv = Din >> 2;
#pragma unroll 2
    for (int c = sub; c < nv; c += G) {
        float4 av[L];
#pragma unroll
        for (int l = 0; l < L; ++l)
            av[l] = (l < Lmax) ? reinterpret_cast<const float4*>(ae + (int64_t)l * Din)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t off = (int64_t)ic * nv + c;
        float4 wv, mv = make_float4(0.f, 0.f, 0.f, 0.f), vv, w0v;
        if (first) {
            w0v = reinterpret_cast<const float4*>(w0)[off];
            wv = w0v;
            vv = mv;
        } else {
            wv = reinterpret_cast<const float4*>(we)[off];
            if constexpr (!FM) mv = reinterpret_cast<const float4*>(me)[off];
            vv = sgl ? mv : reinterpret_cast<const float4*>(ve)[off];
            if (clamp_eps >= 0.f) w0v = reinterpret_cast<const float4*>(w0)[off];
        }
        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < L; ++l) {
            g[0] += dyv[l] * av[l].x;
            g[1] += dyv[l] * av[l].y;
            g[2] += dyv[l] * av[l].z;
            g[3] += dyv[l] * av[l].w;
        }
        float wq[4] = {wv.x, wv.y, wv.z, wv.w};
        float mq[4] = {mv.x, mv.y, mv.z, mv.w};
        float vq[4] = {vv.x, vv.y, vv.z, vv.w};
        const float w0q[4] = {w0v.x, w0v.y, w0v.z, w0v.w};
        if constexpr (FM) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                mq[0] += dn[l] * av[l].x;
                mq[1] += dn[l] * av[l].y;
                mq[2] += dn[l] * av[l].z;
                mq[3] += dn[l] * av[l].w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wq[k] *= decay;
            if constexpr (!FM) mq[k] = mq[k] + (g[k] - mq[k]) * (1.f - beta1);
            const float a0k = k == 0 ? av[0].x : k == 1 ? av[0].y : k == 2 ? av[0].z : av[0].w;
            vq[k] = sgl ? en * (a0k * a0k) : vq[k] * beta2 + (1.f - beta2) * g[k] * g[k];
            const float denom = sqrtf(vq[k]) / bc2_sqrt + eps;
            wq[k] -= step_size * (mq[k] / denom);
            if (clamp_eps >= 0.f) wq[k] = fminf(fmaxf(wq[k], w0q[k] - clamp_eps), w0q[k] + clamp_eps);
        }
        if (row_ok) {
            reinterpret_cast<float4*>(we)[off] = make_float4(wq[0], wq[1], wq[2], wq[3]);
            if constexpr (!FM) reinterpret_cast<float4*>(me)[off] = make_float4(mq[0], mq[1], mq[2], mq[3]);
            if (!sgl) reinterpret_cast<float4*>(ve)[off] = make_float4(vq[0], vq[1], vq[2], vq[3]);
        }
#pragma unroll
        for (int l = 0; l < L; ++l)
            ysum[l] += (wq[0] * av[l].x + wq[1] * av[l].y) + (wq[2] * av[l].z + wq[3] * av[l].w);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float s = ysum[l];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0 && row_ok && l < Lmax) {
            y[((int64_t)e * Lmax + l) * Dout + i] = s;
            if constexpr (FM) {
                me[(int64_t)l * Dout + i] = dn[l];          // (the group's lanes read the old value at entry, same wave)
                if (sgl && l == 0) me[(int64_t)Lmax * Dout + i] = en;
            }
        }
    }
}

// WIDE (17..64 loss rows per edit: long targets, e.g. captions): the G = 16 lane-group form with the a-rows NOT held in registers --
// each float4 column step reads the L a-values twice (gradient, then next-forward dot products) from L1 / L2, where the [L, Din]
// block of an edit (<= 2.6 MB) lives anyway.  Same per-element arithmetic and summation order over l as the kernels above.
template <int L, bool FM = false>
__global__ __launch_bounds__(256) void ft_adamw_step_wide_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                                 const float* __restrict__ w0, const float* __restrict__ a,
                                                                 const float* __restrict__ dy, float* __restrict__ y,
                                                                 const int32_t* __restrict__ do_update, const int32_t* __restrict__ adam_t, const int32_t* __restrict__ single,
                                                                 int Lmax, int Dout, int Din, float lr, float beta1, float beta2, float eps,
                                                                 float wd, float clamp_eps, int row_blocks, int64_t w0_stride_e) {
    constexpr int G = 16, RW = 64 / G;
    const int e = blockIdx.x / row_blocks;
    const int rb = blockIdx.x % row_blocks;
    if (!do_update[e]) return;  // uniform
    w0 += (int64_t)e * w0_stride_e;
    const int t = adam_t[e];
    const bool first = (t <= 1);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    const float decay = 1.f - lr * wd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G;
    const int i = (rb * 4 + wave) * RW + lane / G;
    const bool row_ok = i < Dout;
    const int ic = row_ok ? i : Dout - 1;
    const int64_t mat = (int64_t)Dout * Din;
    float* we = w + (int64_t)e * mat;
    float* me = FM ? m + (int64_t)e * (Lmax + 1) * Dout : m + (int64_t)e * mat;       // FM: the [Lmax + 1, Dout] state: D (laid out like dy) | e (SGL)
    float* ve = v + (int64_t)e * mat;
    const float* ae = a + (int64_t)e * Lmax * Din;
    const float* dye = dy + (int64_t)e * Lmax * Dout;
    const bool sgl = FM && single != nullptr && single[e] != 0;        // see ft_adamw_step_kernel
    float dyv[L], ysum[L], dn[FM ? L : 1];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        dyv[l] = (l < Lmax) ? dye[(int64_t)l * Dout + ic] : 0.f;
        ysum[l] = 0.f;
        if constexpr (FM) {
            const float dold = (!first && l < Lmax) ? me[(int64_t)l * Dout + ic] : 0.f;
            dn[l] = dold + (dyv[l] - dold) * (1.f - beta1);
        }
    }
    float en = 0.f;
    if (sgl) {
        const float eold = first ? 0.f : me[(int64_t)Lmax * Dout + ic];
        en = eold * beta2 + (1.f - beta2) * dyv[0] * dyv[0];
    }
    const int nv = Din >> 2;
    for (int c = sub; c < nv; c += G) {
        const int64_t off = (int64_t)ic * nv + c;
        float4 wv, mv = make_float4(0.f, 0.f, 0.f, 0.f), vv, w0v;
        if (first) {
            w0v = reinterpret_cast<const float4*>(w0)[off];
            wv = w0v;
            vv = mv;
        } else {
            wv = reinterpret_cast<const float4*>(we)[off];
            if constexpr (!FM) mv = reinterpret_cast<const float4*>(me)[off];
            vv = sgl ? mv : reinterpret_cast<const float4*>(ve)[off];
            if (clamp_eps >= 0.f) w0v = reinterpret_cast<const float4*>(w0)[off];
        }
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        float mq[4] = {mv.x, mv.y, mv.z, mv.w};
        const float4 a0v = sgl ? reinterpret_cast<const float4*>(ae)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (l < Lmax) {
                const float4 av = reinterpret_cast<const float4*>(ae + (int64_t)l * Din)[c];
                g[0] += dyv[l] * av.x;
                g[1] += dyv[l] * av.y;
                g[2] += dyv[l] * av.z;
                g[3] += dyv[l] * av.w;
                if constexpr (FM) {
                    mq[0] += dn[l] * av.x;
                    mq[1] += dn[l] * av.y;
                    mq[2] += dn[l] * av.z;
                    mq[3] += dn[l] * av.w;
                }
            }
        }
        float wq[4] = {wv.x, wv.y, wv.z, wv.w};
        float vq[4] = {vv.x, vv.y, vv.z, vv.w};
        const float w0q[4] = {w0v.x, w0v.y, w0v.z, w0v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wq[k] *= decay;
            if constexpr (!FM) mq[k] = mq[k] + (g[k] - mq[k]) * (1.f - beta1);
            const float a0k = k == 0 ? a0v.x : k == 1 ? a0v.y : k == 2 ? a0v.z : a0v.w;
            vq[k] = sgl ? en * (a0k * a0k) : vq[k] * beta2 + (1.f - beta2) * g[k] * g[k];
            const float denom = sqrtf(vq[k]) / bc2_sqrt + eps;
            wq[k] -= step_size * (mq[k] / denom);
            if (clamp_eps >= 0.f) wq[k] = fminf(fmaxf(wq[k], w0q[k] - clamp_eps), w0q[k] + clamp_eps);
        }
        if (row_ok) {
            reinterpret_cast<float4*>(we)[off] = make_float4(wq[0], wq[1], wq[2], wq[3]);
            if constexpr (!FM) reinterpret_cast<float4*>(me)[off] = make_float4(mq[0], mq[1], mq[2], mq[3]);
            if (!sgl) reinterpret_cast<float4*>(ve)[off] = make_float4(vq[0], vq[1], vq[2], vq[3]);
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (l < Lmax) {
                const float4 av = reinterpret_cast<const float4*>(ae + (int64_t)l * Din)[c];
                ysum[l] += (wq[0] * av.x + wq[1] * av.y) + (wq[2] * av.z + wq[3] * av.w);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float s = ysum[l];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0 && row_ok && l < Lmax) {
            y[((int64_t)e * Lmax + l) * Dout + i] = s;
            if constexpr (FM) {
                me[(int64_t)l * Dout + i] = dn[l];
                if (sgl && l == 0) me[(int64_t)Lmax * Dout + i] = en;
            }
        }
    }
}

template <int L, bool FM>
static int launch_adamw_wide(float* w, float* m, float* v, const float* w0, const float* a, const float* dy, float* y,
                             const int32_t* do_update, const int32_t* adam_t, const int32_t* single, int E, int Lmax, int Dout, int Din, float lr,
                             float beta1, float beta2, float eps, float wd, float clamp_eps, int64_t w0_stride_e, hipStream_t st) {
    const int row_blocks = (Dout + 15) / 16;
    const int ph = devqa_prof_begin(DEVQA_PROF_FT_ADAMW, st);
    hipLaunchKernelGGL((ft_adamw_step_wide_kernel<L, FM>), dim3(E * row_blocks), dim3(256), 0, st, w, m, v, w0, a, dy, y, do_update, adam_t, single,
                       Lmax, Dout, Din, lr, beta1, beta2, eps, wd, clamp_eps, row_blocks, w0_stride_e);
    devqa_prof_end(ph, (FM ? 16.0 : 24.0) * E * (double)Dout * Din, st);
    DEVQA_LAUNCH_CHECK("ft_adamw_step(wide)");
    return DEVQA_OK;
}

template <int L, int ROWS, bool FM>
static int launch_adamw(float* w, float* m, float* v, const float* w0, const float* a, const float* dy, float* y,
                        const int32_t* do_update, const int32_t* adam_t, const int32_t* single, int E, int Lmax, int Dout, int Din, float lr,
                        float beta1, float beta2, float eps, float wd, float clamp_eps, int64_t w0_stride_e, hipStream_t st) {
    // column-compacted matrices: lanes per row by issue efficiency nv / (ceil(nv / G) * G); the wave-per-row form when it is >= 0.9 there
    const int nv4 = Din >> 2;
    auto eff = [&](int G) { return (double)nv4 / (double)((nv4 + G - 1) / G * G); };
    static const int grouped_env = getenv("DEVQA_FT_GROUPED") ? atoi(getenv("DEVQA_FT_GROUPED")) : -1;     // 0: off, 8 / 16: forced
    if (Din <= 1024 && grouped_env != 0 && (grouped_env > 0 || eff(64) < 0.9)) {
        const int G = grouped_env > 0 ? grouped_env : (eff(16) >= 0.83 ? 16 : 8);   // measured (tools/debug/ft_sweep_bench.py): 256-byte pieces win unless they idle > 1/6 of the steps
        const int rows_wg = 4 * (64 / G);
        const int row_blocks = (Dout + rows_wg - 1) / rows_wg;
        const int ph = devqa_prof_begin(DEVQA_PROF_FT_ADAMW, st);
        if (G == 16)
            hipLaunchKernelGGL((ft_adamw_step_grouped_kernel<L, 16, FM>), dim3(E * row_blocks), dim3(256), 0, st, w, m, v, w0, a, dy, y, do_update,
                               adam_t, single, Lmax, Dout, Din, lr, beta1, beta2, eps, wd, clamp_eps, row_blocks, w0_stride_e);
        else
            hipLaunchKernelGGL((ft_adamw_step_grouped_kernel<L, 8, FM>), dim3(E * row_blocks), dim3(256), 0, st, w, m, v, w0, a, dy, y, do_update,
                               adam_t, single, Lmax, Dout, Din, lr, beta1, beta2, eps, wd, clamp_eps, row_blocks, w0_stride_e);
        devqa_prof_end(ph, (FM ? 16.0 : 24.0) * E * (double)Dout * Din, st);
        DEVQA_LAUNCH_CHECK("ft_adamw_step(grouped)");
        return DEVQA_OK;
    }
    if (Din <= 1024) {
        const int row_blocks = (Dout + 4 * ROWS - 1) / (4 * ROWS);
        const int ph = devqa_prof_begin(DEVQA_PROF_FT_ADAMW, st);
        hipLaunchKernelGGL((ft_adamw_step_kernel<L, ROWS, true, FM>), dim3(E * row_blocks), dim3(256), 0, st, w, m, v, w0, a, dy, y,
                           do_update, adam_t, single, Lmax, Dout, Din, lr, beta1, beta2, eps, wd, clamp_eps, row_blocks, w0_stride_e);
        devqa_prof_end(ph, (FM ? 16.0 : 24.0) * E * (double)Dout * Din, st);
        DEVQA_LAUNCH_CHECK("ft_adamw_step");
        return DEVQA_OK;
    }
    const int row_blocks = (Dout + ROWS - 1) / ROWS;
    const int ph = devqa_prof_begin(DEVQA_PROF_FT_ADAMW, st);
    hipLaunchKernelGGL((ft_adamw_step_kernel<L, ROWS, false, FM>), dim3(E * row_blocks), dim3(256), 0, st, w, m, v, w0, a, dy, y,
                       do_update, adam_t, single, Lmax, Dout, Din, lr, beta1, beta2, eps, wd, clamp_eps, row_blocks, w0_stride_e);
    devqa_prof_end(ph, (FM ? 16.0 : 24.0) * E * (double)Dout * Din, st);
    DEVQA_LAUNCH_CHECK("ft_adamw_step");
    return DEVQA_OK;
}

template <bool FM>
static int ft_adamw_step_impl(float* w, float* m, float* v, const float* w0, const float* a, const float* dy, float* y,
                              const int32_t* do_update, const int32_t* adam_t, const int32_t* single, int E, int Lmax, int Dout, int Din,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float clamp_eps,
                              int64_t w0_stride_e, void* stream) {
    DEVQA_CHECK_ARG(w && m && v && w0 && a && dy && y && do_update && adam_t, "ft_adamw_step: null pointer");
    if (E == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && Lmax >= 1 && Lmax <= DEVQA_FT_MAX_ROWS, "ft_adamw_step: Lmax=%d unsupported (1..64)", Lmax);
    DEVQA_CHECK_SHAPE(Dout > 0 && Din > 0 && Din % 4 == 0, "ft_adamw_step: bad matrix dims %dx%d", Dout, Din);
    DEVQA_CHECK_SHAPE((long)E * ((Dout + 1) / 2) < 2147483647L, "ft_adamw_step: grid too large");
    hipStream_t st = (hipStream_t)stream;
#define ARGS w, m, v, w0, a, dy, y, do_update, adam_t, single, E, Lmax, Dout, Din, lr, beta1, beta2, eps, weight_decay, clamp_eps, w0_stride_e, st
    if (Lmax <= 1) return launch_adamw<1, 4, FM>(ARGS);
    if (Lmax <= 2) return launch_adamw<2, 4, FM>(ARGS);
    if (Lmax <= 4) return launch_adamw<4, 2, FM>(ARGS);
    if (Lmax <= 8) return launch_adamw<8, 2, FM>(ARGS);
    if (Lmax <= 16) return launch_adamw<16, 1, FM>(ARGS);
    if (Lmax <= 32) return launch_adamw_wide<32, FM>(ARGS);
    return launch_adamw_wide<64, FM>(ARGS);
#undef ARGS
}

extern "C" int devqa_ft_adamw_step(float* w, float* m, float* v, const float* w0, const float* a, const float* dy, float* y,
                                   const int32_t* do_update, const int32_t* adam_t, int E, int Lmax, int Dout, int Din,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, float clamp_eps,
                                   int64_t w0_stride_e, void* stream) {
    return ft_adamw_step_impl<false>(w, m, v, w0, a, dy, y, do_update, adam_t, nullptr, E, Lmax, Dout, Din, lr, beta1, beta2, eps, weight_decay, clamp_eps,
                                     w0_stride_e, stream);
}

extern "C" int devqa_ft_adamw_step_fm(float* w, float* dstate, float* v, const float* w0, const float* a, const float* dy, float* y,
                                      const int32_t* do_update, const int32_t* adam_t, const int32_t* single, int E, int Lmax, int Dout, int Din,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, float clamp_eps,
                                      int64_t w0_stride_e, void* stream) {
    return ft_adamw_step_impl<true>(w, dstate, v, w0, a, dy, y, do_update, adam_t, single, E, Lmax, Dout, Din, lr, beta1, beta2, eps, weight_decay, clamp_eps,
                                    w0_stride_e, stream);
}

// ------------------------------------------------------------------------------------------
template <int L, int ROWS>
__global__ __launch_bounds__(256) void rows_matvec_kernel(const float* __restrict__ w, int64_t w_stride_e,
                                                          const float* __restrict__ a, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, float* __restrict__ y, int Lr,
                                                          int Ls, int Dout, int Din, int row_blocks) {
    __shared__ float red[4][ROWS * L];
    const int e = blockIdx.x / row_blocks;
    const int rb = blockIdx.x % row_blocks;
    const int i0 = rb * ROWS;
    const float* we = w + (int64_t)e * w_stride_e;
    const float* ae = a + (int64_t)e * Ls * Din;       // Ls rows per edit, of which this launch takes Lr (from the caller's offset)
    float ysum[ROWS][L];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int l = 0; l < L; ++l) ysum[r][l] = 0.f;
    const int nv = Din >> 2;
    for (int c = threadIdx.x; c < nv; c += 256) {
        float4 av[L];
#pragma unroll
        for (int l = 0; l < L; ++l)
            av[l] = (l < Lr) ? reinterpret_cast<const float4*>(ae + (int64_t)l * Din)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = i0 + r;
            if (i >= Dout) continue;
            const float4 wv = reinterpret_cast<const float4*>(we)[(int64_t)i * nv + c];
#pragma unroll
            for (int l = 0; l < L; ++l)
                ysum[r][l] += (wv.x * av[l].x + wv.y * av[l].y) + (wv.z * av[l].z + wv.w * av[l].w);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float s = wave_sum(ysum[r][l]);
            if (lane == 0) red[wave][r * L + l] = s;
        }
    __syncthreads();
    if (threadIdx.x < ROWS * L) {
        const int r = threadIdx.x / L, l = threadIdx.x % L;
        if (i0 + r < Dout && l < Lr) {
            float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            const int64_t o = ((int64_t)e * Ls + l) * Dout + i0 + r;
            if (bias) s += bias[i0 + r];
            if (resid) s += resid[o];
            y[o] = s;
        }
    }
}

extern "C" int devqa_rows_matvec_f32(const float* w, int64_t w_stride_e, const float* a, const float* bias,
                                     const float* resid, float* y, int E, int L, int Dout, int Din, void* stream) {
    DEVQA_CHECK_ARG(w && a && y, "rows_matvec: null pointer");
    if (E == 0 || L == 0) return DEVQA_OK;
    DEVQA_CHECK_SHAPE(E > 0 && L >= 1 && L <= DEVQA_FT_MAX_ROWS, "rows_matvec: L=%d unsupported (1..64)", L);
    DEVQA_CHECK_SHAPE(Dout > 0 && Din > 0 && Din % 4 == 0, "rows_matvec: bad matrix dims");
    hipStream_t st = (hipStream_t)stream;
    constexpr int ROWS = 4;
    const int row_blocks = (Dout + ROWS - 1) / ROWS;
    DEVQA_CHECK_SHAPE((long)E * row_blocks < 2147483647L, "rows_matvec: grid too large");
    for (int l0 = 0; l0 < L; l0 += 16) {       // more than 16 rows per edit: 16 at a time (the matrix streams once per chunk)
        const int Lc = L - l0 < 16 ? L - l0 : 16;
        const float* ac = a + (int64_t)l0 * Din;
        const float* rc = resid ? resid + (int64_t)l0 * Dout : nullptr;
        float* yc = y + (int64_t)l0 * Dout;
        if (Lc <= 2)
            hipLaunchKernelGGL((rows_matvec_kernel<2, ROWS>), dim3(E * row_blocks), dim3(256), 0, st, w, w_stride_e, ac, bias, rc, yc, Lc, L,
                               Dout, Din, row_blocks);
        else if (Lc <= 4)
            hipLaunchKernelGGL((rows_matvec_kernel<4, ROWS>), dim3(E * row_blocks), dim3(256), 0, st, w, w_stride_e, ac, bias, rc, yc, Lc, L,
                               Dout, Din, row_blocks);
        else if (Lc <= 8)
            hipLaunchKernelGGL((rows_matvec_kernel<8, ROWS>), dim3(E * row_blocks), dim3(256), 0, st, w, w_stride_e, ac, bias, rc, yc, Lc, L,
                               Dout, Din, row_blocks);
        else
            hipLaunchKernelGGL((rows_matvec_kernel<16, ROWS>), dim3(E * row_blocks), dim3(256), 0, st, w, w_stride_e, ac, bias, rc, yc, Lc, L,
                               Dout, Din, row_blocks);
    }
    DEVQA_LAUNCH_CHECK("rows_matvec");
    return DEVQA_OK;
}
