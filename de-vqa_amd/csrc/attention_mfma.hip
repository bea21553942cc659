// MFMA flash attention for the packed varlen descriptor of include/devqa.h (bf16 in/out, fp32
// softmax statistics and accumulation).  ViT (257x257, dh 88), Q-Former (32x32 / 32x257, dh 64) and
// OPT (causal + visible prefix, dh 80) all go through this kernel.
//
// Workgroup = (sequence, head, 64-query tile); 4 waves x 16 query rows; keys stream through a double-buffered LDS image in
// 64-key chunks (K and V row-major [64][DHP] bf16, rows padded so that both the ds_read_b128 K-fragment reads and the
// ds_read_b64_tr_b16 V reads are conflict-free), one barrier per chunk.
//
// Operand orientation (no LDS round trip for P, no transposed V image):
//   S^T = K . Q^T   -> a lane's 16x16 accumulator holds, for ONE query (lane&15), the keys
//                      16t + 4*(lane>>4) + r of key tile t  -> softmax row statistics need only two
//                      cross-lane steps (xor 16, 32);
//   O^T = V^T . P^T -> the same registers, converted to bf16, ARE the B operand (n = query) of that product if the
//                      k index of the MFMA enumerates keys in the order (tile 2s, r=0..3, tile 2s+1,
//                      r=0..3) for lane group lane>>4; the matching A operand (m = channel) is read from the row-major V
//                      image with ds_read_b64_tr_b16 (4 keys x 16 channels per 16-lane group).  The accumulator then holds,
//                      for the lane's OWN query, 4 consecutive channels per 16-channel tile: the rescale factor and the
//                      normaliser never cross lanes and the output leaves in 8-byte packed stores.
// head dims that are not multiples of 32 (88, 80) are zero-padded in LDS to DHP = 96.
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((address_space(3))) short4_t* lds_s4_ptr;

typedef __bf16 am_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float am_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t am_pack2(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE)
    am_f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, am_bf16x2_t));
}

#define AM_KC 64  // keys per chunk
#define AM_QT 64  // queries per workgroup and query block (QB blocks of 16 queries per wave: AM_QT * QB per workgroup)

// Reductions over the 4 lanes {fr, fr + 16, fr + 32, fr + 48} that hold one query's scores, on the VALU: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two registers, so a butterfly step is one swap + one
// op instead of a ds_bpermute round trip through the LDS crossbar (8 of those sat on every chunk's critical path).
// (Inline asm: clang 22 folds op(r[0], r[1]) of __builtin_amdgcn_permlane{16,32}_swap's two results into r[0] alone.)
__device__ __forceinline__ void am_swap16(float x, float& a, float& b) {
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "=&v"(a), "=&v"(b) : "v"(x));
}
__device__ __forceinline__ void am_swap32(float x, float& a, float& b) {
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "=&v"(a), "=&v"(b) : "v"(x));
}
// max of the 16 scores a lane holds after the S^T MFMAs, as ONE asm block: 7 v_max3_f32 + 1 v_max_f32 without the v_max_f32 x, x
// canonicalisation clang puts in front of every fmaxf operand that comes out of an MFMA (IEEE sNaN quieting; scores are finite or
// -inf here).  The block opens with wait states: on gfx9 a VALU read of an MFMA result is a SOFTWARE-managed hazard, which the
// compiler resolves for its own instructions but not for inline asm -- an unprotected v_max3 here read stale accumulators now and
// then (run-to-run differences at rounding level, since softmax is invariant to its reference point; found by
// tools/debug/att_determinism.py).  12 wait states cover the 4-pass v_mfma_f32_16x16x32_bf16.
__device__ __forceinline__ float am_max16(const float4_t& a, const float4_t& b, const float4_t& c, const float4_t& d) {
    float r;
    asm volatile("s_nop 11\n\t"
                 "v_max3_f32 %0, %1, %2, %3\n\t"
                 "v_max3_f32 %0, %0, %4, %5\n\t"
                 "v_max3_f32 %0, %0, %6, %7\n\t"
                 "v_max3_f32 %0, %0, %8, %9\n\t"
                 "v_max3_f32 %0, %0, %10, %11\n\t"
                 "v_max3_f32 %0, %0, %12, %13\n\t"
                 "v_max3_f32 %0, %0, %14, %15\n\t"
                 "v_max_f32 %0, %0, %16"
                 : "=&v"(r)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]),
                   "v"(c[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
    return r;
}
__device__ __forceinline__ float am_max4(float x) {
    float a, b;
    am_swap16(x, a, b);
    x = fmaxf(a, b);
    am_swap32(x, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float am_sum4(float x) {
    float a, b;
    am_swap16(x, a, b);
    x = a + b;
    am_swap32(x, a, b);
    return a + b;
}

// NW = waves per workgroup (16 * NW * QB queries per tile).  NW = 4 (default): 64-query tiles.  NW = 6 (DHP = 96 only: the chunk image
// must split evenly over the threads): 96-query tiles for sequences whose length leaves a short tail behind the last 64-query
// tile -- ViT-g's 257 tokens are 16 full 16-query blocks + 1 query: 5 tiles x 4 waves stage every K / V chunk five times and park
// 3 idle waves in the fifth workgroup; 3 tiles x 6 waves stage it three times with one idle wave (tools/debug/att_tail_cost.py:
// T = 256 143 us, T = 257 173 us with NW = 4).
// QB = query blocks (16 queries each) per wave.  QB = 1 (default): 64-query tiles, 3 workgroups per CU.  QB = 2 (experimental,
// DEVQA_ATTENTION_QB=2): 128-query tiles -- every K / V fragment read from LDS feeds two MFMAs (half the LDS traffic per query) and
// long sequences need fewer workgroups that each stream the whole K / V (ViT-g, 257 tokens: 3 instead of 5 per image and head);
// measured slower, see launch_attention_mfma.
// blockIdx -> (sequence, head, q tile).  The hardware deals workgroups round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  The q
// tiles of one (sequence, head) GROUP share K / V, and the heads of one sequence share cache lines (adjacent column slices of the same rows),
// so a UNIT of `unit` consecutive groups stays on one XCD and consecutive units go to consecutive XCDs: a unit is n_seq / 64 whole sequences
// (~64 units per launch: enough to balance 8 XCDs, and the sequences of one edit+eval cycle -- an image prefix and the texts that read its K / V
// -- stay together: dealing single sequences round-robin cost the BLIP-2 + MEND_VL config 13 %), one group when there are few sequences.  (Round 2 gave every XCD one contiguous RANGE of groups: in a mixed pack
// -- LLaVA's decoder: 64 prefix sequences of 577 rows first, 192 probe texts of ~20 rows after -- two XCDs then carried all the long
// sequences: 2649 us against 886 us for the same work, tools/debug/att_llava_bench.py.)
// The grid is 8 * ceil(units / 8) * unit * q_tiles workgroups; ids past the last group exit.
__host__ __device__ __forceinline__ int am_unit(int n_seq, int H) { return n_seq >= 32 ? (n_seq >= 128 ? n_seq / 64 : 1) * H : 1; }
__device__ __forceinline__ bool am_remap(int b, int q_tiles, int n_seq, int H, int& bid) {
    if (n_seq < 0) {        // (-n_seq: the round-2 mapping, one contiguous range of ids per XCD; grid = groups * q_tiles exactly)
        const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        return true;
    }
    const int unit = am_unit(n_seq, H);
    const int per = unit * q_tiles;                 // workgroups of a unit
    const int xcd = b & 7, idx = b >> 3;
    const int ul = idx / per, in_u = idx - ul * per;
    const int g = (ul * 8 + xcd) * unit + in_u / q_tiles;
    bid = g * q_tiles + in_u % q_tiles;
    return g < n_seq * H;
}

template <int DHP, int QB, bool DBUF, int NW = 4, int EXP = 0>
__global__ __launch_bounds__(64 * NW, 2) void attention_mfma_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                             const bf16_t* __restrict__ k, int64_t ldk,
                                                             const bf16_t* __restrict__ v, int64_t ldv,
                                                             bf16_t* __restrict__ out, int64_t ldo,
                                                             const int32_t* __restrict__ seq_desc, int H, int dh,
                                                             float scale, int causal, int q_tiles, int n_seq) {
    // bytes per LDS row: the padding makes 8 consecutive rows start in 8 different 4-bank groups (ds_read_b128 K fragments,
    // ds_write_b128 staging) and 4 consecutive rows in 4 different 8-bank groups (ds_read_b64_tr_b16 V fragments); 16 bytes do
    // that for 192- and 128-byte rows (used by the double-buffered instantiation, which needs the capacity: 52 KiB per workgroup
    // at dh 88 / 80 = still 3 workgroups per CU), 32 do it for every row size (the default).
    constexpr int STRIDE = 2 * DHP + ((DBUF && (DHP == 96 || DHP == 64)) ? 16 : 32);
    constexpr int KS = DHP / 32;          // k-steps of the S^T product
    constexpr int DT = DHP / 16;          // 16-channel output tiles
    constexpr int BUF = AM_KC * STRIDE;   // one chunk image
    constexpr int NT = 64 * NW;           // threads per workgroup
    constexpr int QT = 16 * NW * QB;      // queries per workgroup
    // K and V chunks are DOUBLE-buffered: chunk i lives in buffer i & 1, so one barrier per chunk orders both hazards (chunk
    // i + 1 complete before it is read; chunk i - 1 fully consumed before its buffer is refilled) and the LDS writes of the next
    // chunk overlap the MFMAs of this one.  4 x 64 x 208 B = 52 KiB per workgroup at dh 88 / 80: still 3 workgroups per CU.
    // DBUF = false (default): ONE image, two barriers per chunk (all reads done -> refill -> visible).  Measured on ViT-g
    // (tools/attention_bench.py): 184.8 us single-buffered vs 199.3 us double-buffered -- the refill of the next chunk right
    // behind the barrier competes with the K-fragment reads that open the compute segment; kept as an opt-in
    // (DEVQA_ATTENTION_DBUF=1) for shapes where the barrier is the longer wait.
    __shared__ __attribute__((aligned(16))) unsigned char Ks2[(DBUF ? 2 : 1) * BUF];
    __shared__ __attribute__((aligned(16))) unsigned char Vs2[(DBUF ? 2 : 1) * BUF];

    int bid;
    if (!am_remap(blockIdx.x, q_tiles, n_seq, H, bid)) return;
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % H;
    const int s = bid / (q_tiles * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], kp_start = d[2], kp_len = d[3], ko_start = d[4], ko_len = d[5];
    const int q0 = qt * QT;
    if (q0 >= q_len) return;  // uniform per workgroup
    const int nq = min(QT, q_len - q0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int causal_off = ko_len - q_len;  // query i sees own keys 0..i+causal_off
    const int own_hi = causal ? max(0, min(ko_len, q0 + nq + causal_off)) : ko_len;
    const int n_keys = kp_len + own_hi;

    // ---- Q fragments (B operand of S^T = K.Q^T): query row = wave*16 + fr, channels 32ks + 8fq .. +8 ----
    const int qrow0 = q0 + wave * (16 * QB) + fr;  // block b: qrow0 + 16 b (index inside the sequence)
    short8_t qf[QB][KS];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 32 + fq * 8;
            const int qrow = qrow0 + 16 * b;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (qrow < q_len && c < dh) u = *reinterpret_cast<const uint4*>(q + (int64_t)(q_start + qrow) * ldq + h * dh + c);
            qf[b][ks] = *reinterpret_cast<short8_t*>(&u);
        }

    float4_t o[QB][DT];
    float m_run[QB], l_run[QB];  // statistics of query fr of block b (replicated over the 4 lane groups)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int i = 0; i < DT; ++i) o[b][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        m_run[b] = -INFINITY;
        l_run[b] = 0.f;
    }
    const float sc2 = scale * 1.44269504088896340736f;

    constexpr int CH = DHP / 8;  // 16-byte chunks per row
    constexpr int LD = AM_KC * CH / NT;  // 16-byte pieces of K (and of V) per thread and chunk
    static_assert(AM_KC * CH % NT == 0, "chunk must split evenly over the workgroup");
    typedef unsigned am_u32x4_t __attribute__((ext_vector_type(4)));   // (a struct uint4 copied straight from global memory is not
    am_u32x4_t kreg[LD], vreg[LD];                                     //  promoted out of scratch by this compiler)
    // loop-invariant part of this thread's LD staging pieces: LDS offset, channel base pointers, row inside the chunk.  The loads are
    // UNPREDICATED: a row past the last key re-reads the last key (its scores are masked to -inf and p = 0 multiplies a finite V row),
    // a 16-byte piece in the channel padding (dh..DHP) re-reads the last 8 real channels (K: multiplied by Q's zero padding; V: lands
    // in output columns >= dh, which are never stored) -- no exec-masked branches around 2 x LD loads per chunk.
    int st_off[LD], st_row[LD], st_col[LD];
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        const int i = tid + j * NT;
        const int r = i / CH, cv = i - r * CH;
        st_row[j] = r;
        st_off[j] = r * STRIDE + cv * 16;
        st_col[j] = h * dh + min(cv * 8, dh - 8);
    }
#define AM_FETCH(C0)                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < LD; ++j) {                                                                        \
        const int kidx = min((C0) + st_row[j], n_keys - 1);                                                                 \
        const int64_t grow = (kidx < kp_len) ? (int64_t)(kp_start + kidx) : (int64_t)(ko_start + kidx - kp_len);            \
        kreg[j] = *reinterpret_cast<const am_u32x4_t*>(k + grow * ldk + st_col[j]);                                         \
        vreg[j] = *reinterpret_cast<const am_u32x4_t*>(v + grow * ldv + st_col[j]);                                         \
    }
    // Fast form for chunks that lie entirely inside ONE contiguous key range -- the own range of a sequence without a visible
    // prefix, or a visible range alone (every ViT / Q-Former chunk but the last): uniform 64-bit base (SGPRs) + a per-thread 32-bit
    // element offset fixed for the whole loop -> `global_load_dwordx4 v, voff, s[base]` with NO per-chunk vector address arithmetic.
    // (The general form above spends ~50 VALU instructions per chunk on it, 18 of them quarter-rate 32x32 multiplies -- as many
    // cycles as the softmax itself.)
    const bool one_range = kp_len == 0 || own_hi == 0;       // all keys of this tile come from ONE contiguous row range
    const bool fast_rows = one_range && (int64_t)(AM_KC - 1) * max(ldk, ldv) + h * dh + DHP < (1ll << 31);
    uint32_t fk_off[LD], fv_off[LD];
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        fk_off[j] = (uint32_t)(st_row[j] * (int)ldk + st_col[j]);
        fv_off[j] = (uint32_t)(st_row[j] * (int)ldv + st_col[j]);
    }
    const bf16_t* const k_own = k + (int64_t)(kp_len == 0 ? ko_start : kp_start) * ldk;
    const bf16_t* const v_own = v + (int64_t)(kp_len == 0 ? ko_start : kp_start) * ldv;
#define AM_FETCH_ANY(C0)                                                                                                    \
    if (fast_rows && (C0) + AM_KC <= n_keys) {                                                                              \
        const bf16_t* kb = k_own + (int64_t)(C0) * ldk;                                                                     \
        const bf16_t* vb = v_own + (int64_t)(C0) * ldv;                                                                     \
        _Pragma("unroll") for (int j = 0; j < LD; ++j) {                                                                    \
            kreg[j] = *reinterpret_cast<const am_u32x4_t*>(kb + fk_off[j]);                                                 \
            vreg[j] = *reinterpret_cast<const am_u32x4_t*>(vb + fv_off[j]);                                                 \
        }                                                                                                                   \
    } else { AM_FETCH(C0) }
    const bool wave_has_rows = q0 + wave * (16 * QB) < q_len;  // waves without a query still stage K/V and hit the barriers
#define AM_PARK(BUFI)                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < LD; ++j) {                                                                        \
        *reinterpret_cast<am_u32x4_t*>(Ks2 + (BUFI) * BUF + st_off[j]) = kreg[j];                                           \
        *reinterpret_cast<am_u32x4_t*>(Vs2 + (BUFI) * BUF + st_off[j]) = vreg[j];                                           \
    }
    if (n_keys > 0) {
        AM_FETCH_ANY(0)
        AM_PARK(0)
        if (AM_KC < n_keys) { AM_FETCH_ANY(AM_KC) }
    }
    int bi = 0;
    for (int c0 = 0; c0 < n_keys; c0 += AM_KC, bi ^= (DBUF ? 1 : 0)) {
        if (EXP != 3) __syncthreads();  // DBUF: chunk c0 complete in buffer bi, all waves done with buffer bi ^ 1; else: previous chunk consumed
        if (DBUF) {
            if (c0 + AM_KC < n_keys) {  // the next chunk sits in registers: park it in the other buffer, then fetch the one after
                AM_PARK(bi ^ 1)
                if (c0 + 2 * AM_KC < n_keys) { AM_FETCH_ANY(c0 + 2 * AM_KC) }
            }
        } else if (c0 > 0) {            // chunk 0 was parked by the prologue
            if (EXP != 4) { AM_PARK(0) }
            if (EXP != 3) __syncthreads();
            if (EXP != 1 && c0 + AM_KC < n_keys) { AM_FETCH_ANY(c0 + AM_KC) }  // next chunk's loads fly under this chunk's MFMAs
        }
        if (!wave_has_rows) continue;
        const unsigned char* Ks = Ks2 + bi * BUF;
        const unsigned char* Vs = Vs2 + bi * BUF;
        // FULL chunks (64 keys, none hidden by the causal rule) take a branch-free instantiation of the body
        auto body = [&](auto full_tag) {
            constexpr bool FULLC = decltype(full_tag)::value;
        const int nt = FULLC ? 4 : min(4, (n_keys - c0 + 15) >> 4);  // 16-key tiles of this chunk that hold a key (uniform)

            // ---- S^T tiles: keys 16t + (4fq + r), query fr ----
            float4_t st[QB][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int b = 0; b < QB; ++b) st[b][t] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (!FULLC && t >= nt) continue;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const short8_t kf = *reinterpret_cast<const short8_t*>(Ks + (16 * t + fr) * STRIDE + (32 * ks + 8 * fq) * 2);
#pragma unroll
                    for (int b = 0; b < QB; ++b) st[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[b][ks], st[b][t], 0, 0, 0);
                }
            }
            // ---- mask, online softmax for query fr.  Scores stay unscaled: with c = scale * log2(e) > 0,
            // softmax(scale * s) = exp2(c * s - c * max s); masking is needed only in chunks that hold a key past the end
            // or a causally hidden key (uniform per workgroup) ----
            const bool need_mask = !FULLC && ((c0 + AM_KC > n_keys) || (causal && c0 + AM_KC > kp_len));
            short8_t pf[QB][2];
#pragma unroll
            for (int b = 0; b < QB; ++b) {
            float mloc;
            if (FULLC) {
                mloc = am_max16(st[b][0], st[b][1], st[b][2], st[b][3]);
            } else {
                // only the nt (wave-uniform) key tiles that hold a key are touched; the others keep st = 0, which IS their p
                mloc = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= nt) continue;
                    if (need_mask) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kidx = c0 + 16 * t + 4 * fq + r;
                            bool ok = kidx < n_keys;
                            if (causal && kidx >= kp_len) ok = ok && (kidx - kp_len) <= (qrow0 + 16 * b) + causal_off;
                            st[b][t][r] = ok ? st[b][t][r] : -INFINITY;
                        }
                    }
                    mloc = fmaxf(fmaxf(mloc, fmaxf(st[b][t][0], st[b][t][1])), fmaxf(st[b][t][2], st[b][t][3]));   // compiler-scheduled
                }
            }
            mloc = am_max4(mloc);
            float m_new = fmaxf(m_run[b], mloc);                // running max of the UNSCALED scores
            // Lazy rescale: the reference point of the exponentials only has to be CLOSE to the running max.  Unless some query
            // of this wave saw its max grow by more than 2^8 (always true for a query's first visible keys), every lane keeps
            // its previous reference: p <= 256 is harmless in bf16 / fp32, l and O stay consistent, and the alpha broadcast
            // (4 cross-lane gathers) plus the O rescale are skipped for the chunk.
            const bool grow = (m_new - m_run[b]) * sc2 > 8.f;   // m_run[b] = -inf, m_new finite -> true; both -inf -> NaN -> false
            const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;   // wave-uniform
            if (!rescale) m_new = m_run[b];
            float alpha = 1.f, lloc = 0.f;
            if (FULLC || m_new != -INFINITY) {   // a full chunk shows 64 valid, visible keys to every query: m_new is finite
                const float mc = m_new * sc2;
                if (rescale) alpha = __builtin_amdgcn_exp2f(m_run[b] * sc2 - mc);   // m_run[b] = -inf -> 0
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!FULLC && t >= nt) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = EXP == 2 ? fmaf(st[b][t][r], sc2, -mc) : __builtin_amdgcn_exp2f(fmaf(st[b][t][r], sc2, -mc));   // masked (-inf) -> 0
                        st[b][t][r] = p;
                        lloc += p;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) st[b][t] = (float4_t){0.f, 0.f, 0.f, 0.f};
            }
            lloc = am_sum4(lloc);
            l_run[b] = l_run[b] * alpha + lloc;
            m_run[b] = m_new;
            // ---- P fragments (A operand of O = P.V): k order = (tile 2s: r 0..3, tile 2s+1: r 0..3) ----
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                u.x = am_pack2(st[b][2 * s2][0], st[b][2 * s2][1]);
                u.y = am_pack2(st[b][2 * s2][2], st[b][2 * s2][3]);
                u.z = am_pack2(st[b][2 * s2 + 1][0], st[b][2 * s2 + 1][1]);
                u.w = am_pack2(st[b][2 * s2 + 1][2], st[b][2 * s2 + 1][3]);
                pf[b][s2] = *reinterpret_cast<short8_t*>(&u);
            }
            // ---- rescale O^T: its COLUMN is query fr, the query whose alpha this lane already holds ----
            if (rescale) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[b][dt][r] *= alpha;
            }
            }
            // ---- O^T += V^T.P^T: the transposed LDS reads of the row-major V image deliver, for channel fr' and key group fq, the
            // 8 keys of a k-step -- that is the A operand (m = channel) as well as the B operand (n = channel) layout; with P as
            // the B operand (n = query fr, same registers as before) the accumulator tile is O^T: row = channel 16dt + 4fq + r,
            // column = query fr.  Every per-query quantity (alpha, l) then lives in the lane that needs it, and a lane owns 4
            // CONSECUTIVE channels of its query: 8-byte packed stores instead of 2-byte ones. ----
            const int tq = fr >> 2, tp = fr & 3;  // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (!FULLC && 2 * s2 >= nt) continue;   // both key tiles of this k-step are past the last key (P = 0 there)
                    const short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4_ptr)(Vs + (16 * (2 * s2) + 4 * fq + tq) * STRIDE + (16 * dt + 4 * tp) * 2));
                    const short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4_ptr)(Vs + (16 * (2 * s2 + 1) + 4 * fq + tq) * STRIDE + (16 * dt + 4 * tp) * 2));
                    const short8_t vf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                    for (int b = 0; b < QB; ++b) o[b][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[b][s2], o[b][dt], 0, 0, 0);
                }
            }
        };
        // ... also causal chunks that lie entirely at or below the diagonal of this tile's FIRST query (visible to all its rows)
        const bool all_visible = !causal || c0 + AM_KC <= kp_len || (c0 + AM_KC - 1 - kp_len) <= q0 + causal_off;
        if (c0 + AM_KC <= n_keys && all_visible) body(std::true_type{}); else body(std::false_type{});
    }

    // ---- normalise and store: O^T row = channel 16dt + 4fq + r, column = query fr ----
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = q0 + wave * (16 * QB) + 16 * b + fr;
        if (qi >= q_len) continue;
        const float inv = l_run[b] > 0.f ? 1.f / l_run[b] : 0.f;
        bf16_t* orow = out + (int64_t)(q_start + qi) * ldo + h * dh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c = 16 * dt + 4 * fq;
            if (c < dh) {      // dh % 8 == 0: a lane's 4 channels are all inside or all outside
                uint2 u;
                u.x = am_pack2(o[b][dt][0] * inv, o[b][dt][1] * inv);
                u.y = am_pack2(o[b][dt][2] * inv, o[b][dt][3] * inv);
                *reinterpret_cast<uint2*>(orow + c) = u;
            }
        }
    }
}

// ---- LDS-DMA staged variant (the default) ----------------------------------------------------------------------------------------
// Same tiling, operand orientation, softmax and results as attention_mfma_kernel<DHP, 1, false>, different K / V staging: the
// timing-only variants of that kernel (DEVQA_ATTENTION_EXP, tools/debug/att_tail_cost.py) showed that on ViT-g neither the softmax
// VALU work (no exp: +-0 %) nor the barriers (-3 %) bound it but the global -> register -> ds_write_b128 staging of every chunk
// (-24 % without the loads, -39 % without loads and LDS stores).  Here every wave moves its share of the next chunk with
// `global_load_lds_dwordx4` (no staging VGPRs, no LDS store instructions; 64 lanes x 16 B land at consecutive LDS addresses), into the
// other half of a double-buffered image, one barrier per chunk.  The DMA fixes the image layout to plain row-major pieces
// (piece p of 16 bytes at byte 16 p, CH = DHP / 8 pieces per key row, no row padding), so bank conflicts are avoided by a
// source-side permutation instead: LDS position (row R, piece c') holds global piece c' ^ x(R), with x(R) = (R >> 1) & 3 for 4 / 12
// pieces per row, R & 7 for 8, 2R & 15 for 16 -- found by simulating the guide's bank rules for ds_read_b128 (K fragments, 4 x 16
// lane groups) and ds_read_b64_tr_b16 (V fragments, 2 x 32): conflict-free for both images (tools/debug/lds_swizzle_search.py).
typedef __attribute__((address_space(1))) const void* am_gptr_t;
typedef __attribute__((address_space(3))) void* am_lptr_t;
template <int CH>
__device__ __forceinline__ int am_swz(int R) {
    return CH == 8 ? (R & 7) : CH == 16 ? ((2 * R) & 15) : ((R >> 1) & 3);
}

// SB (single-buffered, NW = 2: 32-query tiles) is the form for packs of SHORT sequences (the decoder pack of the edit+eval path: 32 image
// tokens or <= ~25 text tokens per sequence behind a 32-key visible prefix, one or two chunks each): nothing to overlap a second image with,
// so one image (24 KiB at dh 80: 6 workgroups per CU instead of 3) and two waves per (sequence, head) double the pairs in flight per CU.
// Four V^T fragments (ds_read_b64_tr_b16 at byte offsets O0..O3 from one lane address) as ONE asm block that ends with its own lgkmcnt(0).
// Why not the builtin: the compiler cannot tell which LDS-DMA write a transposing read may alias and puts `s_waitcnt vmcnt(0)` in front of
// every one of them -- the P.V half of a round then waits for the chunk just put in flight, i.e. the prefetch overlapped the S = K.Q^T
// half only (seen in the ISA of every instantiation of the kernel below; the plain ds_read_b128 K reads get no such wait).  An asm read is
// invisible to that pass; the DMA / read ordering is the kernel's own (vmcnt + barrier per round).
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void am_tr_read4(uint32_t addr, short4_t& a, short4_t& b, short4_t& c, short4_t& d) {
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\t"
                 "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:%8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(addr), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
                 : "memory");
}
template <int O0, int O1>
__device__ __forceinline__ void am_tr_read2(uint32_t addr, short4_t& a, short4_t& b) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\t"
                 "ds_read_b64_tr_b16 %1, %2 offset:%4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(addr), "n"(O0), "n"(O1)
                 : "memory");
}
// lane id behind an asm the optimiser cannot hoist: per-DMA row / column arithmetic stays in the chunk loop instead of becoming (spilled) lane constants
__device__ __forceinline__ int am_lane_opaque() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// dot product of 8 bf16 pairs (a MFMA operand fragment against 16 bytes of a row) on the VALU, fp32 accumulate
__device__ __forceinline__ float am_dot8(const short8_t& a, const uint4& b, float acc) {
    const uint4 ua = *reinterpret_cast<const uint4*>(&a);
    const uint32_t aw[4] = {ua.x, ua.y, ua.z, ua.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(bw[i] << 16), acc);
        acc = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(bw[i] & 0xffff0000u), acc);
    }
    return acc;
}

template <int DHP, int NW, bool SB = false>
__global__ __launch_bounds__(64 * NW, ((NW == 8 && DHP <= 96) || SB) ? 4 : 2) void attention_mfma_dma_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                 const bf16_t* __restrict__ k, int64_t ldk,
                                                                 const bf16_t* __restrict__ v, int64_t ldv,
                                                                 bf16_t* __restrict__ out, int64_t ldo,
                                                                 const int32_t* __restrict__ seq_desc, int H, int dh,
                                                                 float scale, int causal, int q_tiles, int n_seq) {
    constexpr int ROWB = 2 * DHP;          // bytes per key row in LDS (unpadded)
    constexpr int KS = DHP / 32;
    constexpr int DT = DHP / 16;
    constexpr int CH = DHP / 8;            // 16-byte pieces per row
    constexpr int IMG = AM_KC * ROWB;      // one chunk image
    constexpr int NI = AM_KC * CH / 64;    // DMA instructions (64 lanes x 16 B) per image and chunk
    constexpr int LD = 2 * NI / NW;        // ... per wave and chunk, K and V images together: instruction j * NW + wave of the list K | V
    constexpr int QT = 16 * NW;            // queries per workgroup
    static_assert(2 * NI % NW == 0, "chunk must split evenly over the waves");
    __shared__ __attribute__((aligned(1024))) unsigned char Ks2[(SB ? 1 : 2) * IMG];
    __shared__ __attribute__((aligned(1024))) unsigned char Vs2[(SB ? 1 : 2) * IMG];

    int bid;
    if (!am_remap(blockIdx.x, q_tiles, n_seq, H, bid)) return;
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % H;
    const int s = bid / (q_tiles * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], kp_start = d[2], kp_len = d[3], ko_start = d[4], ko_len = d[5];
    const int q0 = qt * QT;
    if (q0 >= q_len) return;  // uniform per workgroup
    const int nq = min(QT, q_len - q0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int causal_off = ko_len - q_len;
    const int own_hi = causal ? max(0, min(ko_len, q0 + nq + causal_off)) : ko_len;
    const int n_keys = kp_len + own_hi;

    const int qrow = q0 + wave * 16 + fr;
    short8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 32 + fq * 8;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < q_len && c < dh) u = *reinterpret_cast<const uint4*>(q + (int64_t)(q_start + qrow) * ldq + h * dh + c);
        qf[ks] = *reinterpret_cast<short8_t*>(&u);
    }
    float4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = scale * 1.44269504088896340736f;
    // ---- staging: DMA instruction j of this wave fills LDS bytes [(j * 4 + wave) * 1024, + 1024) of an image ----
    // SB: the K and the V piece j, j + LD / 2 of a wave are the same block of their images (LD / 2 * NW == NI), so their row / column
    // constants are kept once and the fast path's element offset is formed at use (two waves per workgroup: registers, not issue slots,
    // bound how many of them a CU holds)
    constexpr int LDH = SB ? LD / 2 : LD;
    static_assert(!SB || (LD % 2 == 0 && (LD / 2 * NW) % NI == 0), "K / V pieces of a wave must pair up");
    int st_row[LDH], st_col[LDH];
    uint32_t f_off[SB ? 1 : LD];
#pragma unroll
    for (int j = 0; j < LDH; ++j) {
        const int blk = (j * NW + wave) % NI;                 // 1-KiB block of its image (wave-uniform, as is the image)
        const int p = blk * 64 + lane;
        const int R = p / CH, cp = p - R * CH;
        const int c = cp ^ am_swz<CH>(R);
        st_row[j] = R;
        st_col[j] = h * dh + min(c * 8, dh - 8);     // channel padding re-reads the last real channels (see the kernel above)
    }
    if constexpr (!SB) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const bool is_v = (j * NW + wave) >= NI;
            f_off[j] = (uint32_t)(st_row[j] * (int)(is_v ? ldv : ldk) + st_col[j]);
        }
    }
    const bool one_range = kp_len == 0 || own_hi == 0;
    const bool fast_rows = one_range && (int64_t)(AM_KC - 1) * max(ldk, ldv) + h * dh + DHP < (1ll << 31);
    const bf16_t* const k_own = k + (int64_t)(kp_len == 0 ? ko_start : kp_start) * ldk;
    const bf16_t* const v_own = v + (int64_t)(kp_len == 0 ? ko_start : kp_start) * ldv;
#define AMD_STAGE(C0, BUFI)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < LD; ++j) {                                                                        \
        const int blk = (j * NW + wave) % NI;                                                                               \
        const bool is_v = (j * NW + wave) >= NI;                                                                            \
        if (blk * 64 / CH >= ((n_keys - (C0) + 31) & ~31)) continue;   /* a last, partial chunk: only the 32-key k-steps of P.V that hold a key (their rows must be finite: p = 0 multiplies them) */ \
        unsigned char* dst = (is_v ? Vs2 : Ks2) + (BUFI) * IMG + blk * 1024;                                                \
        const int64_t ld = is_v ? ldv : ldk;                                                                                \
        const int jj = SB ? j % LDH : j;                                                                                    \
        if (fast_rows && (C0) + AM_KC <= n_keys) {                                                                          \
            const bf16_t* gb = (is_v ? v_own : k_own) + (int64_t)(C0) * ld;                                                 \
            const uint32_t fo = SB ? (uint32_t)(st_row[jj] * (int)ld + st_col[jj]) : f_off[SB ? 0 : j];                     \
            __builtin_amdgcn_global_load_lds((am_gptr_t)(gb + fo), (am_lptr_t)dst, 16, 0, 0);                               \
        } else {                                                                                                            \
            const int kidx = min((C0) + st_row[jj], n_keys - 1);                                                            \
            const int64_t grow = (kidx < kp_len) ? (int64_t)(kp_start + kidx) : (int64_t)(ko_start + kidx - kp_len);        \
            __builtin_amdgcn_global_load_lds((am_gptr_t)((is_v ? v : k) + grow * ld + st_col[jj]), (am_lptr_t)dst, 16, 0, 0); \
        }                                                                                                                   \
    }
    // ---- fragment addresses inside an image (lane constants; tile / k-step offsets are immediates) ----
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = fr * ROWB + (((4 * ks + fq) ^ am_swz<CH>(fr)) * 16);
    const int tq = fr >> 2, tp = fr & 3;
    int voff[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
        voff[dt] = (4 * fq + tq) * ROWB + (((2 * dt + (tp >> 1)) ^ am_swz<CH>(4 * fq + tq)) * 16) + 8 * (tp & 1);

    const bool wave_has_rows = q0 + wave * 16 < q_len;
    if (n_keys > 0) { AMD_STAGE(0, 0) }
    int bi = 0;
    for (int c0 = 0; c0 < n_keys; c0 += AM_KC, bi ^= 1) {
        if constexpr (SB) {
            if (c0 > 0) {
                __syncthreads();                            // every wave is done with chunk c0 - 64: the one image can be refilled
                AMD_STAGE(c0, 0)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c0 have landed
            __syncthreads();                                    // ... everybody's have, and buffer bi ^ 1 (chunk c0 - 64) is consumed
            if (c0 + AM_KC < n_keys) { AMD_STAGE(c0 + AM_KC, bi ^ 1) }
        }
        if (!wave_has_rows) continue;
        const unsigned char* Ks = Ks2 + (SB ? 0 : bi) * IMG;
        const unsigned char* Vs = Vs2 + (SB ? 0 : bi) * IMG;
        auto body = [&](auto full_tag) {
            constexpr bool FULLC = decltype(full_tag)::value;
            const int nt = FULLC ? 4 : min(4, (n_keys - c0 + 15) >> 4);
            float4_t st[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                st[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (!FULLC && t >= nt) continue;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const short8_t kf = *reinterpret_cast<const short8_t*>(Ks + 16 * t * ROWB + koff[ks]);
                    st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[t], 0, 0, 0);
                }
            }
            const bool need_mask = !FULLC && ((c0 + AM_KC > n_keys) || (causal && c0 + AM_KC > kp_len));
            float mloc;
            if (FULLC) {
                mloc = am_max16(st[0], st[1], st[2], st[3]);
            } else {
                mloc = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= nt) continue;
                    if (need_mask) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kidx = c0 + 16 * t + 4 * fq + r;
                            bool ok = kidx < n_keys;
                            if (causal && kidx >= kp_len) ok = ok && (kidx - kp_len) <= qrow + causal_off;
                            st[t][r] = ok ? st[t][r] : -INFINITY;
                        }
                    }
                    mloc = fmaxf(fmaxf(mloc, fmaxf(st[t][0], st[t][1])), fmaxf(st[t][2], st[t][3]));
                }
            }
            mloc = am_max4(mloc);
            float m_new = fmaxf(m_run, mloc);
            const bool grow = (m_new - m_run) * sc2 > 8.f;
            const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;
            if (!rescale) m_new = m_run;
            float alpha = 1.f, lloc = 0.f;
            if (FULLC || m_new != -INFINITY) {
                const float mc = m_new * sc2;
                if (rescale) alpha = __builtin_amdgcn_exp2f(m_run * sc2 - mc);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!FULLC && t >= nt) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], sc2, -mc));
                        st[t][r] = p;
                        lloc += p;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) st[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
            }
            lloc = am_sum4(lloc);
            l_run = l_run * alpha + lloc;
            m_run = m_new;
            short8_t pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                u.x = am_pack2(st[2 * s2][0], st[2 * s2][1]);
                u.y = am_pack2(st[2 * s2][2], st[2 * s2][3]);
                u.z = am_pack2(st[2 * s2 + 1][0], st[2 * s2 + 1][1]);
                u.w = am_pack2(st[2 * s2 + 1][2], st[2 * s2 + 1][3]);
                pf[s2] = *reinterpret_cast<short8_t*>(&u);
            }
            if (rescale) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
            }
            if constexpr (SB) {        // one image, never read while a DMA is in flight: the builtin's vmcnt(0) costs nothing there
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        if (!FULLC && 2 * s2 >= nt) continue;
                        const short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(Vs + 16 * (2 * s2) * ROWB + voff[dt]));
                        const short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(Vs + 16 * (2 * s2 + 1) * ROWB + voff[dt]));
                        const short8_t vf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s2], o[dt], 0, 0, 0);
                    }
                }
            } else {
                const uint32_t vs_lds = (uint32_t)reinterpret_cast<uintptr_t>((am_lptr_t)Vs);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    short4_t b00, b01, b10, b11;
                    if (FULLC || nt > 2) {
                        am_tr_read4<0, 16 * ROWB, 32 * ROWB, 48 * ROWB>(vs_lds + (uint32_t)voff[dt], b00, b01, b10, b11);
                    } else {
                        am_tr_read2<0, 16 * ROWB>(vs_lds + (uint32_t)voff[dt], b00, b01);
                        b10 = b00;
                        b11 = b01;
                    }
                    const short8_t vf0 = {b00[0], b00[1], b00[2], b00[3], b01[0], b01[1], b01[2], b01[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pf[0], o[dt], 0, 0, 0);
                    if (FULLC || nt > 2) {
                        const short8_t vf1 = {b10[0], b10[1], b10[2], b10[3], b11[0], b11[1], b11[2], b11[3]};
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pf[1], o[dt], 0, 0, 0);
                    }
                }
            }
        };
        const bool all_visible = !causal || c0 + AM_KC <= kp_len || (c0 + AM_KC - 1 - kp_len) <= q0 + causal_off;
        if (c0 + AM_KC <= n_keys && all_visible) body(std::true_type{}); else body(std::false_type{});
    }
#undef AMD_STAGE
    const int qi = q0 + wave * 16 + fr;
    if (qi >= q_len) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t* orow = out + (int64_t)(q_start + qi) * ldo + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int c = 16 * dt + 4 * fq;
        if (c < dh) {
            uint2 u;
            u.x = am_pack2(o[dt][0] * inv, o[dt][1] * inv);
            u.y = am_pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2*>(orow + c) = u;
        }
    }
}

// ---- ring variant: NON-CAUSAL attention over long sequences (ViT-g 257 tokens, CLIP-L 577) ------------------------------------------------
// The two-image kernel above on ViT-g (508 images x 16 heads per launch): 643 us in the bench, 2.3 TB/s of HBM, MFMA pipes 17 % busy; two
// workgroups per CU (117 registers), five staging rounds per workgroup and three workgroups per (image, head) -- the third for the 257th query
// alone -- that all stream the whole K / V.  In-kernel stamps (-DAM_TIMING) put a quarter of a workgroup's life into its prologue (descriptor,
// Q rows, first chunk: nothing to issue) and showed the waits for the DMA to be short: what the kernel lacks is other workgroups to run
// meanwhile, not a deeper prefetch.  Here:
//   * one lean body: a partial last chunk stages clamped rows (finite) and masks their scores -- no second (masked) instantiation, no staging
//     constants (row / column of a piece are re-derived at each DMA, ~12 VALU per KiB moved; am_lane_opaque keeps them from being hoisted and
//     spilled), fragment offsets as one / two lane constants + immediates: 72 registers instead of 117 = THREE workgroups per CU with NB = 2
//     (48 KiB each);
//   * NW = 9: 144-query tiles (waves 0..7 stage, all nine compute) -- 257 queries are two tiles, K / V pass L2 -> LDS twice instead of three
//     times and no workgroup exists for one query;
//   * n_keys % 64 == 1 FOLDS the last key: its K / V rows ride into 512 spare bytes of LDS with chunk 0 and it is processed after the last chunk,
//     where the chunk loop would have met it, on the VALU (a dot product over the Q fragments the lane holds, one lazy-rescale step) -- instead of
//     a whole round (barrier + DMA flight) for a chunk of one key: 257 keys = 4 rounds.  Same order of operations as the unfolded form, so results
//     differ from it only where the fp32 score does (VALU fma chain against the MFMA's internal sum), not by a different softmax reference point;
//   * V^T fragments are read by am_tr_read4 (see there), the round's barrier is a bare s_barrier (__syncthreads() carries a fence that lowers
//     to vmcnt(0));
//   * NB = 3 (opt-in, DEVQA_ATTENTION_NBUF=3): a ring of three chunk images, the LDS-DMA two chunks ahead behind counted waits (`vmcnt(LD)`: a
//     wave's pieces complete in issue order, so "all but my newest LD" = the chunk about to be used has landed).  72 KiB = two workgroups per
//     CU: measured SLOWER than NB = 2 (575 vs 493 us, tools/debug/att_vit_bench.py) -- occupancy beats prefetch depth here.
// Descriptor semantics, operand orientation, lazy rescale and output are those of attention_mfma_dma_kernel with causal == 0.
#ifdef AM_TIMING   /* hipcc -DAM_TIMING attention_mfma.hip -o build/am_timing: where a round's time goes (debug builds only) */
__device__ unsigned long long* g_am_stamps;
#define AM_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == 7)) g_am_stamps[((size_t)blockIdx.x * 2 + (wave == 7)) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ int g_am_hs;            // layout experiment of the harness: element offset between heads of q / k / v (0: h * dh, the product layout)
#define AMR_HOFF (g_am_hs ? h * g_am_hs : h * dh)
#else
#define AM_STAMP(i) do { } while (0)
#define AMR_HOFF (h * dh)
#endif
template <int DHP, int NW, int NB = 3>
__global__ __launch_bounds__(64 * NW, DHP > 96 ? 4 : (NW == 9 ? 7 : 6)) void attention_ring_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                                 const bf16_t* __restrict__ k, int64_t ldk,
                                                                                 const bf16_t* __restrict__ v, int64_t ldv,
                                                                                 bf16_t* __restrict__ out, int64_t ldo,
                                                                                 const int32_t* __restrict__ seq_desc, int H, int dh,
                                                                                 float scale, int fold_ok, int q_tiles, int n_seq) {
    constexpr int ROWB = 2 * DHP;
    constexpr int KS = DHP / 32;
    constexpr int DT = DHP / 16;
    constexpr int CH = DHP / 8;
    constexpr int IMG = AM_KC * ROWB;
    constexpr int NI = AM_KC * CH / 64;
    constexpr int NS = 8;                  // staging waves
    constexpr int LD = 2 * NI / NS;        // DMA instructions per staging wave and chunk (K and V images together)
    constexpr int QT = 16 * NW;
    static_assert(NW == 8 || NW == 9, "8 or 9 waves");
    static_assert(2 * NI % NS == 0, "chunk must split evenly over the staging waves");
    static_assert(NB == 2 || NB == 3, "two or three chunk images");
    extern __shared__ __attribute__((aligned(1024))) unsigned char am_ring_smem[];      // 2 * NB * IMG bytes
    unsigned char* const Ks3 = am_ring_smem;
    unsigned char* const Vs3 = am_ring_smem + NB * IMG;

    int bid;
    if (!am_remap(blockIdx.x, q_tiles, n_seq, H, bid)) return;
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % H;
    const int s = bid / (q_tiles * H);
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], kp_start = d[2], kp_len = d[3], ko_start = d[4], ko_len = d[5];
    const int q0 = qt * QT;
    if (q0 >= q_len) return;  // uniform per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // fold_ok bit 1 (value 2): the own key range is causal (query i sees own keys 0 .. i + ko_len - q_len); a tile's chunk loop then ends at its
    // last query's diagonal, and a wave whose 16 queries all lie above a chunk skips it
    const bool causal = (fold_ok & 2) != 0;
    const int causal_off = ko_len - q_len;
    const int own_hi = causal ? max(0, min(ko_len, q0 + min(QT, q_len - q0) + causal_off)) : ko_len;
    const int n_all = kp_len + own_hi;
    const bool fold = (fold_ok & 1) && !causal && (n_all & (AM_KC - 1)) == 1 && n_all > 1;
    const int n_keys = n_all - (fold ? 1 : 0);        // keys that go through the chunk loop (a folded key has index n_keys)
    AM_STAMP(0);

    const int qrow = q0 + wave * 16 + fr;
    // first key this lane's query does NOT see, and the same for the first / last query of the wave (uniform)
    const int lim = causal ? min(n_keys, kp_len + max(0, qrow + causal_off + 1)) : n_keys;
    const int lim_lo = causal ? min(n_keys, kp_len + max(0, q0 + wave * 16 + causal_off + 1)) : n_keys;
    const int lim_hi = causal ? min(n_keys, kp_len + max(0, q0 + wave * 16 + 15 + causal_off + 1)) : n_keys;
    short8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 32 + fq * 8;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < q_len && c < dh) u = *reinterpret_cast<const uint4*>(q + (int64_t)(q_start + qrow) * ldq + AMR_HOFF + c);
        qf[ks] = *reinterpret_cast<short8_t*>(&u);
    }
    // the folded key's rows of K and V go into 512 bytes behind the images (one DMA of wave 0, issued before -- so landed with -- chunk 0): piece i of
    // the K row at byte 16 i, of the V row at 256 + 16 i
    unsigned char* const odd = am_ring_smem + 2 * NB * IMG;
    if (fold && wave == 0 && (lane < CH || (lane >= 16 && lane < 16 + CH))) {
        const int64_t grow = n_keys < kp_len ? (int64_t)(kp_start + n_keys) : (int64_t)(ko_start + n_keys - kp_len);
        const int c = min((lane & 15) * 8, dh - 8);
        const bf16_t* src = (lane < 16 ? k + grow * ldk : v + grow * ldv) + AMR_HOFF + c;
        __builtin_amdgcn_global_load_lds((am_gptr_t)src, (am_lptr_t)odd, 16, 0, 0);
    }
    // ---- staging: DMA instruction j of staging wave w fills LDS bytes [(j * 8 + w) * 1024, + 1024) of the K | V image pair; rows past the last
    // key re-read it (finite data under p = 0), channel padding re-reads the last real channels (multiplied by the zero padding of Q / dropped) ----
#define AMR_STAGE(C0, BUFI)                                                                                                  \
    if (wave < NS) {                                                                                                        \
        const int ln_ = am_lane_opaque();                                                                                   \
        _Pragma("unroll") for (int j = 0; j < LD; ++j) {                                                                    \
            const int blk = (j * NS + wave) % NI;                                                                           \
            const bool is_v = (j * NS + wave) >= NI;                                                                        \
            unsigned char* dst = (is_v ? Vs3 : Ks3) + (BUFI) * IMG + blk * 1024;                                            \
            const int p = blk * 64 + ln_;                                                                                   \
            const int p_row = p / CH;                                                                                       \
            const int p_col = AMR_HOFF + min(((p - p_row * CH) ^ am_swz<CH>(p_row)) * 8, dh - 8);                           \
            const int kidx = min((C0) + p_row, n_keys - 1);                                                                 \
            const int64_t grow = (kidx < kp_len) ? (int64_t)(kp_start + kidx) : (int64_t)(ko_start + kidx - kp_len);        \
            __builtin_amdgcn_global_load_lds((am_gptr_t)((is_v ? v : k) + grow * (is_v ? ldv : ldk) + p_col), (am_lptr_t)dst, 16, 0, 0); \
        }                                                                                                                   \
    }
    // fragment addresses inside an image.  CH == 12 swizzles the two low piece bits only: (4 ks + fq) ^ x = 4 ks + (fq ^ x) and
    // (2 dt + hb) ^ x = 4 (dt >> 1) + ((2 (dt & 1) + hb) ^ x): one (K) / two (V) lane constants + immediates
    constexpr bool AFF = CH == 12;
    const int tq = fr >> 2, tp = fr & 3;
    int koff[AFF ? 1 : KS], voff[AFF ? 2 : DT];
    if constexpr (AFF) {
        koff[0] = fr * ROWB + ((fq ^ am_swz<CH>(fr)) * 16);
        voff[0] = (4 * fq + tq) * ROWB + (((tp >> 1) ^ am_swz<CH>(4 * fq + tq)) * 16) + 8 * (tp & 1);
        voff[1] = (4 * fq + tq) * ROWB + (((2 + (tp >> 1)) ^ am_swz<CH>(4 * fq + tq)) * 16) + 8 * (tp & 1);
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) koff[AFF ? 0 : ks] = fr * ROWB + (((4 * ks + fq) ^ am_swz<CH>(fr)) * 16);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            voff[AFF ? 0 : dt] = (4 * fq + tq) * ROWB + (((2 * dt + (tp >> 1)) ^ am_swz<CH>(4 * fq + tq)) * 16) + 8 * (tp & 1);
    }
#define AMR_KOFF(ks) (AFF ? koff[0] + (ks) * 64 : koff[AFF ? 0 : (ks)])
#define AMR_VOFF(dt) (AFF ? voff[AFF ? (dt) & 1 : 0] + ((dt) >> 1) * 64 : voff[AFF ? 0 : (dt)])

    float4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = scale * 1.44269504088896340736f;
    const bool wave_has_rows = q0 + wave * 16 < q_len;
    if (n_keys > 0) { AMR_STAGE(0, 0) }
    if (NB == 3 && n_keys > AM_KC) { AMR_STAGE(AM_KC, 1) }
    int bi = 0;
    AM_STAMP(1);
    for (int c0 = 0; c0 < n_keys; c0 += AM_KC, bi = (bi == NB - 1 ? 0 : bi + 1)) {
        AM_STAMP(2 + (c0 >> 6) * 6);
        if (NB == 3 && c0 + AM_KC < n_keys) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");   // chunk c0 has landed, chunk c0 + 64 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AM_STAMP(3 + (c0 >> 6) * 6);
        // LDS reads of the previous round are complete (their values fed MFMAs); DMA writes are covered by the counted wait above
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everybody's pieces have landed, image bi + 2 (chunk c0 - 64) is consumed
        AM_STAMP(4 + (c0 >> 6) * 6);
        if (c0 + (NB - 1) * AM_KC < n_keys) { AMR_STAGE(c0 + (NB - 1) * AM_KC, (bi == 0 ? NB - 1 : bi - 1)) }
        AM_STAMP(5 + (c0 >> 6) * 6);
        if (!wave_has_rows || c0 >= lim_hi) continue;          // (causal: every query of this wave lies above the chunk)
        const unsigned char* Ks = Ks3 + bi * IMG;
        const uint32_t vs_lds = (uint32_t)reinterpret_cast<uintptr_t>((am_lptr_t)(Vs3 + bi * IMG));
        float4_t st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const short8_t kf = *reinterpret_cast<const short8_t*>(Ks + 16 * t * ROWB + AMR_KOFF(ks));
                st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[t], 0, 0, 0);
            }
        }
        if (c0 + AM_KC > lim_lo) {            // (uniform per wave) a partial last chunk or the causal diagonal: hidden keys score -inf
            const int rel = lim - c0 - 4 * fq;            // score (t, r) of this lane is visible iff 16 t + r < rel
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[t][r] = (16 * t + r < rel) ? st[t][r] : -INFINITY;
        }
        float mloc = am_max4(am_max16(st[0], st[1], st[2], st[3]));
        float m_new = fmaxf(m_run, mloc);
        const bool grow = (m_new - m_run) * sc2 > 8.f;
        const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;
        if (!rescale) m_new = m_run;
        float alpha = 1.f, lloc = 0.f;
        const float mc = m_new == -INFINITY ? 0.f : m_new * sc2;       // (-inf: a causal query that sees no key yet -- its p are exp2(-inf) = 0)
        if (rescale) alpha = __builtin_amdgcn_exp2f(m_run * sc2 - mc);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], sc2, -mc));
                st[t][r] = p;
                lloc += p;
            }
        lloc = am_sum4(lloc);
        l_run = l_run * alpha + lloc;
        m_run = m_new;
        short8_t pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 u;
            u.x = am_pack2(st[2 * s2][0], st[2 * s2][1]);
            u.y = am_pack2(st[2 * s2][2], st[2 * s2][3]);
            u.z = am_pack2(st[2 * s2 + 1][0], st[2 * s2 + 1][1]);
            u.w = am_pack2(st[2 * s2 + 1][2], st[2 * s2 + 1][3]);
            pf[s2] = *reinterpret_cast<short8_t*>(&u);
        }
        if (rescale) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        }
        AM_STAMP(6 + (c0 >> 6) * 6);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            short4_t b00, b01, b10, b11;
            am_tr_read4<0, 16 * ROWB, 32 * ROWB, 48 * ROWB>(vs_lds + (uint32_t)AMR_VOFF(dt), b00, b01, b10, b11);
            const short8_t vf0 = {b00[0], b00[1], b00[2], b00[3], b01[0], b01[1], b01[2], b01[3]};
            const short8_t vf1 = {b10[0], b10[1], b10[2], b10[3], b11[0], b11[1], b11[2], b11[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pf[0], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pf[1], o[dt], 0, 0, 0);
        }
        AM_STAMP(7 + (c0 >> 6) * 6);
    }
    if (fold && wave_has_rows) {
        // the folded key, after the last chunk (where the chunk loop would have met it), on the VALU: score = dot of the lane's Q fragments with
        // its channels of the K row (summed over the four lanes of the query), then the same lazy-rescale step as a chunk with this one score;
        // P is rounded to bf16 for the product as the MFMA operand would be, the normaliser takes it unrounded (as in the chunk body)
        float sp = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sp = am_dot8(qf[ks], *reinterpret_cast<const uint4*>(odd + 16 * (4 * ks + fq)), sp);
        const float sod = am_sum4(sp);
        float m_new = fmaxf(m_run, sod);
        const bool grow = (m_new - m_run) * sc2 > 8.f;
        const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;
        if (!rescale) m_new = m_run;
        const float mc = m_new * sc2;
        const float alpha = rescale ? __builtin_amdgcn_exp2f(m_run * sc2 - mc) : 1.f;
        const float p = __builtin_amdgcn_exp2f(fmaf(sod, sc2, -mc));
        l_run = l_run * alpha + p;
        m_run = m_new;
        const float pb = __uint_as_float(am_pack2(p, 0.f) << 16);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const uint2 u = *reinterpret_cast<const uint2*>(odd + 256 + 32 * dt + 8 * fq);
            const float vv[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] = fmaf(pb, vv[r], rescale ? o[dt][r] * alpha : o[dt][r]);
        }
    }
#undef AMR_STAGE
#undef AMR_KOFF
#undef AMR_VOFF
    if (!wave_has_rows) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    // ---- output: whole row slices.  A lane holds 4 channels of ONE query per 16-channel tile: stored as they stand, a wave instruction writes 16 rows
    // x 32 bytes -- 16 cache lines for 512 bytes, six times per wave, and the CU's address path (a line per cycle) is what bounds this kernel.  Waves
    // 0..7 pass their 16 x dh block through a quarter of the chunk image that is free (image `bi`: consumed a round ago, nothing in flight into it
    // after the last round's vmcnt(0)) and store 16-byte pieces of consecutive channels: ~5.8 rows of dh x 2 contiguous bytes per instruction.
    // The ninth wave (16 rows of tile 0, one of ViT-g's tile 1) and head dims that are not multiples of 8 keep the direct form. ----
    if (wave < 8 && (dh & 7) == 0) {
        unsigned char* ob = (wave < 4 ? Ks3 : Vs3) + bi * IMG + (wave & 3) * (16 * ROWB);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 u;
            u.x = am_pack2(o[dt][0] * inv, o[dt][1] * inv);
            u.y = am_pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2*>(ob + fr * ROWB + 32 * dt + 8 * fq) = u;
        }
        const int np = dh >> 3;                                  // 16-byte pieces per row
        const int magic = 65536 / np + 1;                        // p / np for p < 256 (np <= 16)
        const int row0 = q0 + wave * 16;
#pragma unroll
        for (int j = 0; j < (16 * CH + 63) / 64; ++j) {
            const int p_ = j * 64 + lane;
            const int r = (p_ * magic) >> 16;
            const int c = p_ - r * np;
            if (r < 16 && row0 + r < q_len) {
                const uint4 u = *reinterpret_cast<const uint4*>(ob + r * ROWB + c * 16);
                *reinterpret_cast<uint4*>(out + (int64_t)(q_start + row0 + r) * ldo + h * dh + c * 8) = u;
            }
        }
        return;
    }
    if (qrow >= q_len) return;
    bf16_t* orow = out + (int64_t)(q_start + qrow) * ldo + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int c = 16 * dt + 4 * fq;
        if (c < dh) {
            uint2 u;
            u.x = am_pack2(o[dt][0] * inv, o[dt][1] * inv);
            u.y = am_pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2*>(orow + c) = u;
        }
    }
}

// ---- pack variant: packs of SHORT sequences (the decoder pack of the edit+eval path: 32 image tokens or <= ~25 text tokens per sequence,
// usually behind a 32-key visible prefix -- one key chunk) -----------------------------------------------------------------------------------
// OPT-IN (DEVQA_ATTENTION_PACK=1), kept as the measured answer to "is the decoder pack bound by its ~3000 wave-instructions per (sequence,
// head)?" (profiles/r02_summary.md): no.  The tiled kernels above give such a (sequence, head) a workgroup of 2-4 waves of which one or two
// hold queries, a staging round through a shared image and two barriers; here a WAVE is the unit -- one (sequence, head, 16-query block) per
// wave, four independent waves per workgroup, no barrier, ~700 instructions per item -- and the launch takes the same 360-387 us (see
// launch_attention_mfma).  What the item does:
//   * K fragments come straight from global memory into the MFMA operand registers (lane (fr, fq) of S^T = K.Q^T holds 8 channels of key
//     16 t + fr: a 16-byte load, one row address per 16-key tile, channel offsets are immediates);
//   * V goes by LDS-DMA into a wave-private image in the same access shape (16 rows x 64 contiguous bytes per instruction, one row address per
//     16 rows): block (g, kb) of 1 KiB holds the 64-byte column group kb of rows 16 g .. 16 g + 15, so a transposing read is ONE lane constant
//     + immediates (4-way bank conflicts on 72 reads per item: noise);
//   * masking is one compare + select per score against a per-lane limit (lim = first hidden key: end of the keys, or the causal diagonal);
//     rows / channels past the end are clamped instead of predicated (no exec-mask branches), Q's padding channels are zeroed by a select;
//   * every load of the item (3 Q + <= 12 K + <= 16 V-DMA) is in flight before the first wait; key tiles past the end are not loaded.
// Sequences with more than 64 keys loop over chunks without overlap (correct, not what this form is for).  Same descriptor semantics, operand
// orientation and arithmetic order as the kernels above (bit-identical to them on one-chunk sequences); a running maximum that grows always
// rescales (one chunk: never).
#ifdef AM_TIMING
#define AMP_STAMP(i) do { if (lane == 0) g_am_stamps[(size_t)item * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AMP_STAMP(i) do { } while (0)
#endif
template <int DHP>
__global__ __launch_bounds__(256) void attention_pack_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k,
                                                             int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                             bf16_t* __restrict__ out, int64_t ldo, const int32_t* __restrict__ seq_desc,
                                                             int H, int dh, float scale, int causal, int qb_shift, int n_items) {
    constexpr int KS = DHP / 32;           // 32-channel k-steps of S = 64-byte column groups of a row
    constexpr int DT = DHP / 16;
    constexpr int IMG = AM_KC * 2 * DHP;   // one wave's V image: 4 row groups x KS blocks of 1 KiB
    extern __shared__ __attribute__((aligned(1024))) unsigned char am_pack_smem[];      // 4 * IMG bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg;
    {       // one contiguous range of workgroups per XCD (blockIdx % 8): the heads and texts of a cycle read their prefix through one L2
        const int b = blockIdx.x, nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int item = wg * 4 + wave;
    if (item >= n_items) return;
    AMP_STAMP(0);
    const int qb = item & ((1 << qb_shift) - 1);       // (items per (sequence, head): a power of two of 16-query blocks; blocks past q_len exit)
    const int sh = item >> qb_shift;
    const int s = sh / H;
    const int h = sh - s * H;
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], kp_start = d[2], kp_len = d[3], ko_start = d[4], ko_len = d[5];
    const int q0 = qb * 16;
    if (q0 >= q_len) return;               // uniform per wave
    AMP_STAMP(1);
    const int ldq32 = (int)ldq, ldk32 = (int)ldk, ldv32 = (int)ldv;       // (row strides fit 31 bits: checked by the launcher) one v_mad_i64_i32 per row address
    const int fr = lane & 15, fq = lane >> 4;
    const int causal_off = ko_len - q_len;
    const int own_hi = causal ? max(0, min(ko_len, q0 + 16 + causal_off)) : ko_len;
    const int n_keys = kp_len + own_hi;
    const int qrow = q0 + fr;
    // first key this lane's query does NOT see: the end of the keys, or the causal diagonal (the visible prefix is never hidden)
    const int lim = causal ? min(n_keys, kp_len + max(0, qrow + causal_off + 1)) : n_keys;
    unsigned char* const Vs = am_pack_smem + wave * IMG;
    const int tq = fr >> 2, tp = fr & 3;
    const uint32_t va = (uint32_t)reinterpret_cast<uintptr_t>((am_lptr_t)Vs) + (uint32_t)((4 * fq + tq) * 64 + (tp >> 1) * 16 + 8 * (tp & 1));
    // channel offsets (elements) of this lane's 16-byte pieces: K / Q fragments (8 fq) and V staging (8 (lane & 3)); only the last 64-byte
    // group can run past dh: it re-reads the last real channels (K, V: multiplied by zeros of Q / dropped with the output padding)
    const int kc_last = min((KS - 1) * 32 + fq * 8, dh - 8), vc_last = min((KS - 1) * 32 + (lane & 3) * 8, dh - 8);
    const bool q_pad = (KS - 1) * 32 + fq * 8 >= dh;

    short8_t qf[KS];
    {
        const bf16_t* qr = q + (int64_t)(q_start + min(qrow, q_len - 1)) * ldq32 + h * dh;       // rows past the end repeat the last one (never stored)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 u = *reinterpret_cast<const uint4*>(qr + (ks < KS - 1 ? ks * 32 + fq * 8 : kc_last));
            if (ks == KS - 1) {
                u.x = q_pad ? 0u : u.x;
                u.y = q_pad ? 0u : u.y;
                u.z = q_pad ? 0u : u.z;
                u.w = q_pad ? 0u : u.w;
            }
            qf[ks] = *reinterpret_cast<short8_t*>(&u);
        }
    }
    float4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = scale * 1.44269504088896340736f;
    for (int c0 = 0; c0 < n_keys; c0 += AM_KC) {
        const int nt = min(4, (n_keys - c0 + 15) >> 4);            // 16-key tiles of this chunk that hold a key (uniform)
        if (c0 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the previous chunk's image is consumed: its reads fed MFMAs)
        // ---- V: row group g = rows 16 g + (lane >> 2); rows past the last key re-read it (finite data under p = 0) ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g >= ((nt + 1) >> 1) * 2) continue;                 // (uniform) only the 32-key k-steps of P.V that hold a key
            const int kidx = min(c0 + 16 * g + (lane >> 2), n_keys - 1);
            const int grow = kidx < kp_len ? kp_start + kidx : ko_start + kidx - kp_len;
            const bf16_t* vr = v + (int64_t)grow * ldv32 + h * dh;
#pragma unroll
            for (int kb = 0; kb < KS; ++kb)
                __builtin_amdgcn_global_load_lds((am_gptr_t)(vr + (kb < KS - 1 ? kb * 32 + (lane & 3) * 8 : vc_last)),
                                                 (am_lptr_t)(Vs + (g * KS + kb) * 1024), 16, 0, 0);
        }
        // ---- S^T = K.Q^T with K from global memory ----
        short8_t kf[4][KS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= nt) continue;
            const int kidx = min(c0 + 16 * t + fr, n_keys - 1);
            const int grow = kidx < kp_len ? kp_start + kidx : ko_start + kidx - kp_len;
            const bf16_t* kr = k + (int64_t)grow * ldk32 + h * dh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 u = *reinterpret_cast<const uint4*>(kr + (ks < KS - 1 ? ks * 32 + fq * 8 : kc_last));
                kf[t][ks] = *reinterpret_cast<const short8_t*>(&u);
            }
        }
        AMP_STAMP(2);
        float4_t st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (t >= nt) continue;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[ks], st[t], 0, 0, 0);
        }
        AMP_STAMP(3);
        const int rel = lim - c0 - 4 * fq;                          // score (t, r) of this lane is visible iff 16 t + r < rel
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[t][r] = (16 * t + r < rel) ? st[t][r] : -INFINITY;      // (tiles past nt: their keys are past lim)
        const float mloc = am_max4(am_max16(st[0], st[1], st[2], st[3]));
        const float m_new = fmaxf(m_run, mloc);
        const float mc = m_new == -INFINITY ? 0.f : m_new * sc2;
        const float alpha = __builtin_amdgcn_exp2f(m_run * sc2 - mc);       // first chunk: exp2(-inf) = 0
        float lloc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], sc2, -mc));
                st[t][r] = p;
                lloc += p;
            }
        lloc = am_sum4(lloc);
        l_run = l_run * alpha + lloc;
        m_run = m_new;
        short8_t pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 u;
            u.x = am_pack2(st[2 * s2][0], st[2 * s2][1]);
            u.y = am_pack2(st[2 * s2][2], st[2 * s2][3]);
            u.z = am_pack2(st[2 * s2 + 1][0], st[2 * s2 + 1][1]);
            u.w = am_pack2(st[2 * s2 + 1][2], st[2 * s2 + 1][3]);
            pf[s2] = *reinterpret_cast<short8_t*>(&u);
        }
        if (c0 > 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        }
        AMP_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's V image has landed (nobody else reads or writes it)
        AMP_STAMP(5);
        if (nt > 2) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                short4_t b00, b01, b10, b11;
                am_tr_read4<0, KS * 1024, 2 * KS * 1024, 3 * KS * 1024>(va + (dt >> 1) * 1024 + (dt & 1) * 32, b00, b01, b10, b11);
                const short8_t vf0 = {b00[0], b00[1], b00[2], b00[3], b01[0], b01[1], b01[2], b01[3]};
                const short8_t vf1 = {b10[0], b10[1], b10[2], b10[3], b11[0], b11[1], b11[2], b11[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pf[0], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pf[1], o[dt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                short4_t b00, b01;
                am_tr_read2<0, KS * 1024>(va + (dt >> 1) * 1024 + (dt & 1) * 32, b00, b01);
                const short8_t vf0 = {b00[0], b00[1], b00[2], b00[3], b01[0], b01[1], b01[2], b01[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pf[0], o[dt], 0, 0, 0);
            }
        }
    }
    AMP_STAMP(6);
    if (qrow >= q_len) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t* orow = out + (int64_t)(q_start + qrow) * ldo + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int c = 16 * dt + 4 * fq;
        if (c < dh) {
            uint2 u;
            u.x = am_pack2(o[dt][0] * inv, o[dt][1] * inv);
            u.y = am_pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2*>(orow + c) = u;
        }
    }
}

// ---- resident variant: non-causal self-attention over short sequences (ViT-g: 257 tokens) ------------------------------------
// One workgroup of 8 waves per (sequence, head).  K and V of the WHOLE sequence are staged into LDS once (rows padded to a multiple
// of 64: 320 x 224 B x 2 = 140 KiB for ViT-g), one barrier, then every wave streams the key chunks out of LDS for two 16-query blocks
// at a time with no further workgroup synchronisation (the chunked kernel above pays two barriers and a staging round per 64 keys
// and streams K / V once per 64-query tile: 5 workgroups per image and head).  Same S^T / P.V orientation, lazy rescale and
// permlane reductions as above.
template <int DHP>
__global__ __launch_bounds__(512, 1) void attention_mfma_resident_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                          const bf16_t* __restrict__ k, int64_t ldk,
                                                                          const bf16_t* __restrict__ v, int64_t ldv,
                                                                          bf16_t* __restrict__ out, int64_t ldo,
                                                                          const int32_t* __restrict__ seq_desc, int H, int dh,
                                                                          float scale, int nkp) {
    constexpr int STRIDE = 2 * DHP + 32;
    constexpr int KS = DHP / 32;
    constexpr int DT = DHP / 16;
    constexpr int CH = DHP / 8;
    constexpr int QB = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char am_smem[];
    unsigned char* Ks = am_smem;
    unsigned char* Vs = am_smem + (size_t)nkp * STRIDE;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int h = bid % H;
    const int s = bid / H;
    const int32_t* d = seq_desc + s * 6;
    const int q_start = d[0], q_len = d[1], ko_start = d[4], n_keys = d[5];
    if (n_keys <= 0 || q_len <= 0) return;   // uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    typedef unsigned am_u32x4_t __attribute__((ext_vector_type(4)));
    // ---- stage K and V: rows >= n_keys re-read the last key (masked below), padding channels re-read the last real ones ----
    for (int i = tid; i < nkp * CH; i += 512) {
        const int r = i / CH, cv = i - r * CH;
        const int64_t grow = ko_start + min(r, n_keys - 1);
        const int col = h * dh + min(cv * 8, dh - 8);
        *reinterpret_cast<am_u32x4_t*>(Ks + r * STRIDE + cv * 16) = *reinterpret_cast<const am_u32x4_t*>(k + grow * ldk + col);
        *reinterpret_cast<am_u32x4_t*>(Vs + r * STRIDE + cv * 16) = *reinterpret_cast<const am_u32x4_t*>(v + grow * ldv + col);
    }
    __syncthreads();
    const float sc2 = scale * 1.44269504088896340736f;
    const int nblk = (q_len + 15) >> 4;
    for (int blk0 = wave; blk0 < nblk; blk0 += 16) {     // this wave: query blocks blk0 and blk0 + 8
        short8_t qf[QB][KS];
        float4_t o[QB][DT];
        float m_run[QB], l_run[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const int qrow = (blk0 + 8 * b) * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + fq * 8;
                uint4 u = make_uint4(0, 0, 0, 0);
                if (qrow < q_len && c < dh) u = *reinterpret_cast<const uint4*>(q + (int64_t)(q_start + qrow) * ldq + h * dh + c);
                qf[b][ks] = *reinterpret_cast<short8_t*>(&u);
            }
#pragma unroll
            for (int i = 0; i < DT; ++i) o[b][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
            m_run[b] = -INFINITY;
            l_run[b] = 0.f;
        }
        for (int c0 = 0; c0 < n_keys; c0 += AM_KC) {
            const unsigned char* Kc = Ks + c0 * STRIDE;
            const unsigned char* Vc = Vs + c0 * STRIDE;
            const bool full = c0 + AM_KC <= n_keys;
            const int nt = full ? 4 : min(4, (n_keys - c0 + 15) >> 4);
            float4_t st[QB][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int b = 0; b < QB; ++b) st[b][t] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (t >= nt) continue;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const short8_t kf = *reinterpret_cast<const short8_t*>(Kc + (16 * t + fr) * STRIDE + (32 * ks + 8 * fq) * 2);
#pragma unroll
                    for (int b = 0; b < QB; ++b) st[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[b][ks], st[b][t], 0, 0, 0);
                }
            }
            short8_t pf[QB][2];
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                if (!full) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[b][t][r] = (c0 + 16 * t + 4 * fq + r < n_keys) ? st[b][t][r] : -INFINITY;
                }
                float mloc = fmaxf(fmaxf(fmaxf(st[b][0][0], st[b][0][1]), fmaxf(st[b][0][2], st[b][0][3])),
                                   fmaxf(fmaxf(st[b][1][0], st[b][1][1]), fmaxf(st[b][1][2], st[b][1][3])));
                mloc = fmaxf(mloc, fmaxf(fmaxf(fmaxf(st[b][2][0], st[b][2][1]), fmaxf(st[b][2][2], st[b][2][3])),
                                         fmaxf(fmaxf(st[b][3][0], st[b][3][1]), fmaxf(st[b][3][2], st[b][3][3]))));
                mloc = am_max4(mloc);
                float m_new = fmaxf(m_run[b], mloc);
                const bool grow = (m_new - m_run[b]) * sc2 > 8.f;
                const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;
                if (!rescale) m_new = m_run[b];
                float alpha = 1.f, lloc = 0.f;
                const float mc = m_new * sc2;      // n_keys > 0 and no causal mask: every query has seen a key after chunk 0
                if (rescale) alpha = __builtin_amdgcn_exp2f(m_run[b] * sc2 - mc);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(st[b][t][r], sc2, -mc));
                        st[b][t][r] = p;
                        lloc += p;
                    }
                lloc = am_sum4(lloc);
                l_run[b] = l_run[b] * alpha + lloc;
                m_run[b] = m_new;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 u;
                    u.x = am_pack2(st[b][2 * s2][0], st[b][2 * s2][1]);
                    u.y = am_pack2(st[b][2 * s2][2], st[b][2 * s2][3]);
                    u.z = am_pack2(st[b][2 * s2 + 1][0], st[b][2 * s2 + 1][1]);
                    u.w = am_pack2(st[b][2 * s2 + 1][2], st[b][2 * s2 + 1][3]);
                    pf[b][s2] = *reinterpret_cast<short8_t*>(&u);
                }
                if (rescale) {
                    float ar[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * fq + r, 64);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[b][dt][r] *= ar[r];
                }
            }
            const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (2 * s2 >= nt) continue;
                    const short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4_ptr)(Vc + (16 * (2 * s2) + 4 * fq + tq) * STRIDE + (16 * dt + 4 * tp) * 2));
                    const short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4_ptr)(Vc + (16 * (2 * s2 + 1) + 4 * fq + tq) * STRIDE + (16 * dt + 4 * tp) * 2));
                    const short8_t vf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                    for (int b = 0; b < QB; ++b) o[b][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][s2], vf, o[b][dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float lr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run[b], 4 * fq + r, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = (blk0 + 8 * b) * 16 + 4 * fq + r;
                if (qi >= q_len) continue;
                const float inv = lr[r] > 0.f ? 1.f / lr[r] : 0.f;
                bf16_t* orow = out + (int64_t)(q_start + qi) * ldo + h * dh;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int c = 16 * dt + fr;
                    if (c < dh) orow[c] = f32_to_bf16(o[b][dt][r] * inv);
                }
            }
        }
    }
}

// Variant switches (A/B measurements and the variant tests): read from the environment ONCE, and again only when the host asks
// (devqa_attention_reload_env; lib.attention re-reads when it sees a DEVQA_ATTENTION_* variable change).  -1 = unset.
namespace {
struct AttnEnv { int resident = -1, qb = -1, dbuf = -1, nw = -1, exp = -1, dma = -1, shrt = -1, xcd = -1, ring = -1, fold = -1, nbuf = -1, pack = -1; };
AttnEnv g_attn_env;
std::atomic<int> g_attn_env_ready{0};
std::mutex g_attn_env_mu;
int env_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; }
void attn_env_load() {
    std::lock_guard<std::mutex> lock(g_attn_env_mu);
    AttnEnv e;
    e.resident = env_int("DEVQA_ATTENTION_RESIDENT");
    e.qb = env_int("DEVQA_ATTENTION_QB");
    e.dbuf = env_int("DEVQA_ATTENTION_DBUF");
    e.nw = env_int("DEVQA_ATTENTION_NW");
    e.dma = env_int("DEVQA_ATTENTION_DMA");
    e.shrt = env_int("DEVQA_ATTENTION_SHORT");
    e.xcd = env_int("DEVQA_ATTENTION_XCD");
    e.ring = env_int("DEVQA_ATTENTION_RING");
    e.fold = env_int("DEVQA_ATTENTION_FOLD");
    e.nbuf = env_int("DEVQA_ATTENTION_NBUF");
    e.pack = env_int("DEVQA_ATTENTION_PACK");
#ifdef DEVQA_EXPERIMENTS
    e.exp = env_int("DEVQA_ATTENTION_EXP");      // timing experiments (WRONG results): compiled in only with -DDEVQA_EXPERIMENTS
#endif
    g_attn_env = e;
    g_attn_env_ready.store(1, std::memory_order_release);
}
}  // namespace

extern "C" int devqa_attention_reload_env() {
    attn_env_load();
    return DEVQA_OK;
}

int launch_attention_mfma(const bf16_t* q, int64_t ldq, const bf16_t* k, int64_t ldk, const bf16_t* v, int64_t ldv, bf16_t* out,
                          int64_t ldo, const int32_t* seq_desc, int n_seq, int max_q_len, int H, int dh, float scale,
                          int causal, void* stream) {
    const int dhp = (dh + 31) / 32 * 32;
    if (dhp != 32 && dhp != 64 && dhp != 96 && dhp != 128) return devqa_fail(DEVQA_E_SHAPE, "attention: dh=%d unsupported", dh);
    if (!g_attn_env_ready.load(std::memory_order_acquire)) attn_env_load();
    const AttnEnv env = g_attn_env;
    hipStream_t st0 = (hipStream_t)stream;
    // causal bit 2 (value 4, include/devqa.h): the caller asserts plain non-causal self-attention (kp_len == 0, ko_len == q_len) for
    // EVERY sequence -> short sequences MAY take the K/V-resident kernel (opt-in: DEVQA_ATTENTION_RESIDENT=1)
    const bool self_full = (causal & 4) != 0;
    causal &= 3;
    if (self_full && causal == 0 && max_q_len <= 320 && (dhp == 96 || dhp == 64) && (long)n_seq * H >= 128) {
        // built, tested and NOT the default: 233 us against 214 us for the chunked kernel on ViT-g (127 x 16 x 257): one 8-wave
        // workgroup per CU (140 KiB of LDS) hides less latency than three 4-wave ones, and the staging is not overlapped
        if (env.resident == 1) {
            const int nkp = (max_q_len + AM_KC - 1) / AM_KC * AM_KC;
            const size_t smem = (size_t)2 * nkp * (2 * dhp + 32);
            const long grid = (long)n_seq * H;
            if (dhp == 96) {
                auto kern = attention_mfma_resident_kernel<96>;
                static std::atomic<unsigned> attr96{0};
                devqa_set_max_smem(kern, 160 * 1024, attr96);
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st0, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, nkp);
            } else {
                auto kern = attention_mfma_resident_kernel<64>;
                static std::atomic<unsigned> attr64{0};
                devqa_set_max_smem(kern, 160 * 1024, attr64);
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st0, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, nkp);
            }
            DEVQA_LAUNCH_CHECK("attention_mfma_resident");
            return DEVQA_OK;
        }
    }
    // QB = 2 (128-query tiles) is built and tested (DEVQA_ATTENTION_QB=2) but NOT the default: on ViT-g (127 images x 16 heads x
    // 257 tokens) it measured 223 us against 207 us for QB = 1 -- halving the LDS fragment traffic does not pay for the occupancy
    // lost to 221 VGPRs; the kernel is bound by its ~19 VALU instructions per MFMA (SQ counters, profiles/r01_summary.md H).
    const int qb = (env.qb == 2 && max_q_len > 2 * AM_QT && dhp <= 96) ? 2 : 1;
    const bool dbuf = env.dbuf == 1 && qb == 1 && dhp <= 96;
    // 96-query tiles (6 waves) when they need fewer wave slots than 64-query tiles for the longest sequence (257 -> 18 vs 20)
    const bool nw_env = env.nw != -1;
    const int slots4 = (max_q_len + 63) / 64 * 4, slots6 = (max_q_len + 95) / 96 * 6;
    int nw = 4;   // 6 waves measured 40 % SLOWER on ViT-g (6 waves over 4 SIMDs: two SIMDs carry twice the work): opt-in only
    (void)slots4; (void)slots6;
    if (nw_env && dhp == 96 && qb == 1 && !dbuf) nw = env.nw == 6 ? 6 : 4;
    // timing experiments (wrong results, tools/debug/att_tail_cost.py): exist only in a -DDEVQA_EXPERIMENTS build
    const int exp_id = (env.exp > 0 && nw == 4 && qb == 1 && !dbuf) ? env.exp : 0;
    const bool dma_env = env.dma != -1;                       // 0: the register-staged kernel (previous default)
    // default for long sequences only: on the decoder pack of the bench (960 sequences of 32-80 keys, visible prefix + causal own
    // range, one or two chunks each) the DMA kernel measured 198 us against 135 us for the register-staged one (rocprofv3 trace of
    // bench.py) -- nothing to overlap the DMA with; DEVQA_ATTENTION_DMA=1 forces it, =0 disables it
    const bool dma = nw == 4 && qb == 1 && !dbuf && exp_id == 0 &&
                     (dma_env ? env.dma != 0 : max_q_len >= 224);
    // 128-query tiles (8 waves) halve the K / V staging per query, which is what bounds the kernel on long sequences (ViT-g, T = 256:
    // 133 -> 116 us); short sequences fill 64-query tiles better (T = 128: 43.6 vs 47.3 us)
    int dma_nw = max_q_len >= 224 ? 8 : 4;
    // dh 128, causal (the LLaMA decoders of LLaVA / MiniGPT-4: 577-row image prefixes): 64-query tiles of 4 waves measured 565 us against 685 us for
    // 128-query tiles of 8 on the prefix pack of a 16-cycle batch (tools/debug/att_llava_bench.py) -- a causal tile's K / V range ends at its last
    // query, so the larger tile stages more keys per query than it saves
    if (dhp == 128 && (causal & 1)) dma_nw = 4;
    if (env.nw == 8 || env.nw == 4) dma_nw = env.nw;
    // non-causal long sequences (ViT-g, CLIP-L): the ring kernel (three chunk images, DMA two chunks ahead; attention_ring_kernel).  144-query
    // tiles of 9 waves when they cover the longest sequence with fewer tiles than 128-query ones (257 tokens: 2 instead of 3).
    // DEVQA_ATTENTION_RING=0 keeps the two-image kernel, DEVQA_ATTENTION_NW=8 / 9 picks the tile, DEVQA_ATTENTION_FOLD=0 turns the folding of
    // key n_keys - 1 (n_keys % 64 == 1) off -- the staging variants are bit-identical to each other only without it.
    // causal long sequences (the LLaMA decoders' 577-row image prefixes): the same kernel with the diagonal masked per lane and chunks above a
    // wave's queries skipped, 128-query tiles.  The LLaVA decoder pack of a 16-cycle batch (64 prefixes of 577 rows + 192 texts behind them, 32
    // heads x 128; tools/debug/att_llava_bench.py): 986 -> 606 us against attention_mfma_dma_kernel<128, 4> (167 registers, two 4-wave
    // workgroups per CU; here 106 registers, two 8-wave workgroups), bit-identical outputs.  DEVQA_ATTENTION_RING=0: the tiled kernel.
    const bool ring_causal = (causal & 1) != 0;
    const bool ring = dma && (causal == 0 ? dhp <= 96 : ring_causal) && max_q_len >= 224 && env.ring != 0 && (env.nw == -1 || env.nw == 8 || env.nw == 9);
    if (ring) dma_nw = env.nw != -1 ? env.nw : ((causal & 1) ? 8 : ((max_q_len + 143) / 144 < (max_q_len + 127) / 128 ? 9 : 8));
    // causal packs of short sequences (decoder probes): 32-query tiles of two waves on ONE LDS-DMA image (attention_mfma_dma_kernel<D, 2, true>);
    // non-causal short-query calls (Q-Former: 32 queries over 257 keys) keep the register-staged kernel
    const bool short_env = env.shrt != -1;
    // measured on a 127-cycle probe pack (tools/debug/att_pack_bench.py, 2159 sequences): OPT heads (dh 80) 368.6 us against 362.6 us for the
    // register-staged kernel -- the pack is bound by the instructions issued per (sequence, head), not by the pairs in flight, and both forms
    // issue about the same; LLaMA heads (dh 128, where the register-staged form holds two workgroups per CU) 57.8 against 72.1 us.  Default:
    // dh 128 only; DEVQA_ATTENTION_SHORT=1 forces it for every head size, =0 turns it off
    const bool short_ok = short_env ? env.shrt != 0 : dhp == 128;
    const bool dma_short = causal && !dma && !dma_env && nw == 4 && qb == 1 && !dbuf && exp_id == 0 && !nw_env && max_q_len <= 64 && short_ok;
    // causal packs of sequences of <= 64 queries, OPT-IN (DEVQA_ATTENTION_PACK=1): one (sequence, head, 16-query block) per WAVE
    // (attention_pack_kernel).  Built, tested (bit-identical to the tiled forms on one-chunk sequences) and NOT the default: on the bench's decoder
    // pack (2159 sequences x 32 heads x 80) it measured 360-387 us in three forms of very different instruction counts against 369 us for the
    // register-staged tiles (tools/debug/att_pack_bench.py) -- the pack is bound by neither issue slots nor HBM (2.4 TB/s) but by the CU's address
    // path: every form fetches 160-byte row slices per head, 16 rows x 64 bytes per wave instruction (profiles/r03_summary.md G).
    const bool pack_ok = (causal & 1) && max_q_len <= 64 && env.pack == 1 && !dma_env && !nw_env && !short_env && qb == 1 &&
                         !dbuf && exp_id == 0 && dh % 8 == 0;
    if (pack_ok) {
        const int qb_shift = max_q_len <= 16 ? 0 : max_q_len <= 32 ? 1 : 2;
        const long n_items = ((long)n_seq * H) << qb_shift;
        DEVQA_CHECK_SHAPE(ldq < (1ll << 31) && ldk < (1ll << 31) && ldv < (1ll << 31), "attention: row stride too large");
        const long wgs = (n_items + 3) / 4;
        DEVQA_CHECK_SHAPE(wgs < 2147483647L && n_items < 2147483647L, "attention: grid too large");
        const int ph = devqa_prof_begin(DEVQA_PROF_ATTENTION, st0);
#define LAUNCH_PACK(D)                                                                                                     \
        hipLaunchKernelGGL((attention_pack_kernel<D>), dim3((unsigned)wgs), dim3(256), 4 * 64 * 2 * D, st0, q, ldq, k, ldk, v, ldv, out, ldo, \
                           seq_desc, H, dh, scale, causal & 1, qb_shift, (int)n_items)
        if (dhp == 32) LAUNCH_PACK(32);
        else if (dhp == 64) LAUNCH_PACK(64);
        else if (dhp == 96) LAUNCH_PACK(96);
        else LAUNCH_PACK(128);
#undef LAUNCH_PACK
        devqa_prof_end(ph, 4.0 * (double)n_seq * H * (double)max_q_len * (double)max_q_len * dh * 0.5, st0);
        DEVQA_LAUNCH_CHECK("attention_pack");
        return DEVQA_OK;
    }
    const int qt = dma_short ? 32 : dma ? 16 * dma_nw : 16 * nw * qb;
    const int q_tiles = (max_q_len + qt - 1) / qt;
    // workgroup -> XCD mapping (am_remap): sequence units for CAUSAL packs with LONG sequences (balance: the LLaMA decoders' mixed packs of
    // 577-row prefixes and 20-row texts), the contiguous ranges of round 2 otherwise -- uniform packs (ViT: every sequence costs the same, ranges
    // keep neighbouring images' lines in one L2) and packs of short sequences (a cycle's texts read their image prefix's K / V through the same
    // L2).  Measured in situ on one box: units everywhere cost the BLIP-2 + FT_VL bench 0.6 % and the MEND_VL config 8 %.
    // DEVQA_ATTENTION_XCD=0 / 1 forces ranges / units.
    const bool units = env.xcd == -1 ? ((causal & 1) && max_q_len >= 224) : env.xcd != 0;
    const long unit = am_unit(n_seq, H);
    const long grid = units ? ((((long)n_seq * H + unit - 1) / unit + 7) / 8) * 8 * unit * q_tiles : (long)n_seq * H * q_tiles;
    const int n_seq_arg = units ? n_seq : -n_seq;
    DEVQA_CHECK_SHAPE(grid < 2147483647L, "attention: grid too large");
    hipStream_t st = (hipStream_t)stream;
#ifdef DEVQA_EXPERIMENTS
#define LAUNCH_EXP()                                                                                                   \
    do {                                                                                                               \
        if (exp_id == 1) hipLaunchKernelGGL((attention_mfma_kernel<96, 1, false, 4, 1>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg); \
        if (exp_id == 2) hipLaunchKernelGGL((attention_mfma_kernel<96, 1, false, 4, 2>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg); \
        if (exp_id == 3) hipLaunchKernelGGL((attention_mfma_kernel<96, 1, false, 4, 3>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg); \
        if (exp_id == 4) hipLaunchKernelGGL((attention_mfma_kernel<96, 1, false, 4, 4>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg); \
    } while (0)
#else
#define LAUNCH_EXP() do { } while (0)     /* exp_id is always 0 in a product build */
#endif
#define LAUNCH(D)                                                                                                      \
    do {                                                                                                               \
        if (qb == 2)                                                                                                   \
            hipLaunchKernelGGL((attention_mfma_kernel<D, 2, false>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, \
                               out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                     \
        else if (dbuf)                                                                                                 \
            hipLaunchKernelGGL((attention_mfma_kernel<(D <= 96 ? D : 96), 1, true>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, \
                               k, ldk, v, ldv, out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                     \
        else if (dma_short)                                                                                            \
            hipLaunchKernelGGL((attention_mfma_dma_kernel<D, 2, true>), dim3((unsigned)grid), dim3(128), 0, st, q, ldq, k, ldk, v, ldv, \
                               out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                     \
        else if (ring) {                                                                                               \
            static std::atomic<unsigned> a82{0}, a92{0}, a83{0}, a93{0};                                               \
            const int fold_ok = (env.fold != 0 ? 1 : 0) | ((causal & 1) ? 2 : 0);                                      \
            const int nb = env.nbuf == 3 ? 3 : 2;      /* default: two images = three workgroups per CU; DEVQA_ATTENTION_NBUF=3: the ring of three */ \
            const size_t smem = (size_t)2 * nb * 64 * 2 * D + 512;                                                     \
            if (nb == 2 && dma_nw == 9) {                                                                              \
                devqa_set_max_smem(attention_ring_kernel<D, 9, 2>, smem, a92);                                         \
                hipLaunchKernelGGL((attention_ring_kernel<D, 9, 2>), dim3((unsigned)grid), dim3(576), smem, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, fold_ok, q_tiles, n_seq_arg);                     \
            } else if (nb == 2) {                                                                                      \
                devqa_set_max_smem(attention_ring_kernel<D, 8, 2>, smem, a82);                                         \
                hipLaunchKernelGGL((attention_ring_kernel<D, 8, 2>), dim3((unsigned)grid), dim3(512), smem, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, fold_ok, q_tiles, n_seq_arg);                     \
            } else if (dma_nw == 9) {                                                                                  \
                devqa_set_max_smem(attention_ring_kernel<D, 9, 3>, smem, a93);                                         \
                hipLaunchKernelGGL((attention_ring_kernel<D, 9, 3>), dim3((unsigned)grid), dim3(576), smem, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, fold_ok, q_tiles, n_seq_arg);                     \
            } else {                                                                                                   \
                devqa_set_max_smem(attention_ring_kernel<D, 8, 3>, smem, a83);                                         \
                hipLaunchKernelGGL((attention_ring_kernel<D, 8, 3>), dim3((unsigned)grid), dim3(512), smem, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, fold_ok, q_tiles, n_seq_arg);                     \
            }                                                                                                          \
        } else if (dma) {                                                                                              \
            if (dma_nw == 8)                                                                                           \
                hipLaunchKernelGGL((attention_mfma_dma_kernel<D, 8>), dim3((unsigned)grid), dim3(512), 0, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                 \
            else                                                                                                       \
                hipLaunchKernelGGL((attention_mfma_dma_kernel<D, 4>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, \
                                   out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                 \
        } else if (exp_id >= 1 && exp_id <= 4 && D == 96) {                                                              \
            LAUNCH_EXP();                                                                                              \
        } else if (nw == 6)                                                                                            \
            hipLaunchKernelGGL((attention_mfma_kernel<96, 1, false, 6>), dim3((unsigned)grid), dim3(384), 0, st, q, ldq, k, ldk, v, ldv, \
                               out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                     \
        else                                                                                                           \
            hipLaunchKernelGGL((attention_mfma_kernel<D, 1, false>), dim3((unsigned)grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, \
                               out, ldo, seq_desc, H, dh, scale, causal, q_tiles, n_seq_arg);                                     \
    } while (0)
    // FLOPs as launched (4 Tq Tk dh per head with Tq = Tk = max_q_len: an upper bound for ragged / causal batches, exact for ViT)
    const int ph = devqa_prof_begin(DEVQA_PROF_ATTENTION, st);
    if (dhp == 32) LAUNCH(32);
    else if (dhp == 64) LAUNCH(64);
    else if (dhp == 96) LAUNCH(96);
    else LAUNCH(128);
#undef LAUNCH
#undef LAUNCH_EXP
    devqa_prof_end(ph, 4.0 * (double)n_seq * H * (double)max_q_len * (double)max_q_len * dh * ((causal & 1) ? 0.5 : 1.0), st);
    DEVQA_LAUNCH_CHECK("attention_mfma");
    return DEVQA_OK;
}

#ifdef AM_TIMING
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
int devqa_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
int devqa_prof_begin(int, hipStream_t) { return -1; }
void devqa_prof_end(int, double, hipStream_t) {}
static int pack_main() {
    // the bench's decoder pack: per cycle 4 image prefixes of 32 rows, 13 texts of ~17 rows, 11 of them behind a prefix; OPT heads 32 x 80
    const int cycles = 127, H = 32, dh = 80;
    std::vector<int32_t> hd;
    int pos = 0;
    unsigned x = 7;
    for (int c = 0; c < cycles; ++c) {
        int pre[4];
        for (int i = 0; i < 4; ++i) { pre[i] = pos; int32_t e[6] = {pos, 32, 0, 0, pos, 32}; hd.insert(hd.end(), e, e + 6); pos += 32; }
        for (int i = 0; i < 13; ++i) {
            x = x * 1664525u + 1013904223u;
            const int n = 12 + (int)((x >> 16) % 11);
            int32_t e[6] = {pos, n, i < 11 ? pre[i % 4] : 0, i < 11 ? 32 : 0, pos, n};
            hd.insert(hd.end(), e, e + 6);
            pos += n;
        }
    }
    const int n_seq = (int)hd.size() / 6;
    const size_t M = pos, ld = 3 * H * dh;
    bf16_t *qkv, *out; int32_t* desc; unsigned long long* st;
    hipMalloc(&qkv, M * ld * 2); hipMalloc(&out, M * H * dh * 2); hipMalloc(&desc, hd.size() * 4);
    std::vector<unsigned short> h(M * ld);
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 3) & 0x8000)); }
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(desc, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    const int q_blocks = 2;
    const long n_items = (long)n_seq * H * q_blocks, wgs = (n_items + 3) / 4;
    hipMalloc(&st, (size_t)n_items * 64);
    hipMemset(st, 0, (size_t)n_items * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_am_stamps), &st, sizeof(st));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        if (rep == 5) hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL((attention_pack_kernel<96>), dim3((unsigned)wgs), dim3(256), 4 * 64 * 192, nullptr, qkv, (int64_t)ld, qkv + H * dh, (int64_t)ld,
                           qkv + 2 * H * dh, (int64_t)ld, out, (int64_t)(H * dh), desc, H, dh, 1.0f, 1, 1, (int)n_items);
        if (rep == 5) hipEventRecord(e1, nullptr);
    }
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs((size_t)n_items * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> ph[6];
    unsigned long long tmin = ~0ull, tmax = 0;
    long live = 0;
    for (long i = 0; i < n_items; ++i) {
        const unsigned long long* s_ = &hs[(size_t)i * 8];
        if (s_[0] == 0 || s_[6] == 0) continue;
        ++live;
        for (int p_ = 0; p_ < 6; ++p_) ph[p_].push_back((double)(s_[p_ + 1] - s_[p_]));
        tmin = std::min(tmin, s_[0]); tmax = std::max(tmax, s_[6]);
    }
    printf("pack: %d sequences, %zu rows, %ld items (%ld with queries): %.1f us (stamped build); kernel span %.0f cycles\n", n_seq, M, n_items, live, ms * 1e3, (double)(tmax - tmin));
    printf("  median cycles: descriptor %.0f | issue Q, V-DMA, K loads %.0f | S MFMAs (waits for Q, K) %.0f | softmax %.0f | wait V %.0f | P.V %.0f\n",
           med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]));
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 0) return pack_main();
    const int n_seq = 508, H = 16, dh = 88, T = argc > 1 ? atoi(argv[1]) : 257;
    const size_t M = (size_t)n_seq * T, ld = 3 * H * dh;
    bf16_t *qkv, *out; int32_t* desc; unsigned long long* st;
    hipMalloc(&qkv, M * ld * 2); hipMalloc(&out, M * H * dh * 2); hipMalloc(&desc, n_seq * 24);
    std::vector<unsigned short> h(M * ld);
    unsigned x = 1;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) - ((x >> 3) & 0x8000)); }
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<int32_t> hd(n_seq * 6);
    for (int i = 0; i < n_seq; ++i) { hd[i * 6] = i * T; hd[i * 6 + 1] = T; hd[i * 6 + 2] = 0; hd[i * 6 + 3] = 0; hd[i * 6 + 4] = i * T; hd[i * 6 + 5] = T; }
    hipMemcpy(desc, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    // layout experiment: the same values head-major and padded, [3][H][M][96] (rows of one head contiguous: whole-line requests)
    bf16_t* hm;
    hipMalloc(&hm, (size_t)3 * H * M * 96 * 2);
    {
        std::vector<unsigned short> g((size_t)3 * H * M * 96, 0);
        for (size_t r = 0; r < M; ++r)
            for (int w = 0; w < 3; ++w)
                for (int hh = 0; hh < H; ++hh)
                    for (int c = 0; c < dh; ++c) g[(((size_t)w * H + hh) * M + r) * 96 + c] = h[r * ld + (size_t)w * H * dh + hh * dh + c];
        hipMemcpy(hm, g.data(), g.size() * 2, hipMemcpyHostToDevice);
    }
    for (int lay = 0; lay < 2; ++lay) {
        const int hs = lay ? (int)(M * 96) : 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_am_hs), &hs, sizeof(hs));
        const bf16_t* qp = lay ? hm : qkv;
        const bf16_t* kp = lay ? hm + (size_t)H * M * 96 : qkv + H * dh;
        const bf16_t* vp = lay ? hm + (size_t)2 * H * M * 96 : qkv + 2 * H * dh;
        const int64_t ldl = lay ? 96 : (int64_t)ld;
        const int q_tiles = (T + 143) / 144;
        const long grid = (long)n_seq * H * q_tiles;
        hipMalloc(&st, (size_t)grid * 2 * 64 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_am_stamps), &st, sizeof(st));
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_ring_kernel<96, 9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 192 + 512);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL((attention_ring_kernel<96, 9, 2>), dim3((unsigned)grid), dim3(576), 4 * 64 * 192 + 512, nullptr, qp, ldl, kp, ldl, vp, ldl, out,
                               (int64_t)(H * dh), desc, H, dh, 0.1066f, 1, q_tiles, -n_seq);
            hipEventRecord(e1, nullptr);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) best = std::min(best, ms);
        }
        printf("layout %s: ring<96, 9, 2> best of 10: %.1f us (stamped build)\n", lay ? "head-major padded [3][H][M][96]" : "row-major fused qkv [M][3 H dh]", best * 1e3);
        hipFree(st);
    }
    { const int hs = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_am_hs), &hs, sizeof(hs)); }
    for (int nw = 8; nw <= 9; ++nw) {
        const int qt = 16 * nw, q_tiles = (T + qt - 1) / qt;
        const long grid = (long)n_seq * H * q_tiles;
        hipMalloc(&st, (size_t)grid * 2 * 64 * 8);
        hipMemset(st, 0, (size_t)grid * 2 * 64 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_am_stamps), &st, sizeof(st));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 6; ++rep) {
            if (rep == 5) hipEventRecord(e0, nullptr);
            if (nw == 9) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention_ring_kernel<96, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 192 + 512);
                hipLaunchKernelGGL((attention_ring_kernel<96, 9>), dim3((unsigned)grid), dim3(576), 6 * 64 * 192 + 512, nullptr, qkv, (int64_t)ld, qkv + H * dh, (int64_t)ld,
                                   qkv + 2 * H * dh, (int64_t)ld, out, (int64_t)(H * dh), desc, H, dh, 0.1066f, 1, q_tiles, -n_seq);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention_ring_kernel<96, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 192 + 512);
                hipLaunchKernelGGL((attention_ring_kernel<96, 8>), dim3((unsigned)grid), dim3(512), 6 * 64 * 192 + 512, nullptr, qkv, (int64_t)ld, qkv + H * dh, (int64_t)ld,
                                   qkv + 2 * H * dh, (int64_t)ld, out, (int64_t)(H * dh), desc, H, dh, 0.1066f, 1, q_tiles, -n_seq);
            }
            if (rep == 5) hipEventRecord(e1, nullptr);
        }
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hs((size_t)grid * 2 * 64);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("T %d, %d waves: %.1f us (stamped build), %ld workgroups\n", T, nw, ms * 1e3, grid);
        for (int tile = 0; tile < q_tiles; ++tile) {
            for (int wsel = 0; wsel < 2; ++wsel) {
                std::vector<double> pro, total;
                std::vector<double> ph[5][5];
                for (long b = tile; b < grid; b += q_tiles) {
                    // the kernel remaps blockIdx -> (seq, head, tile); tile = bid % q_tiles after the remap, not of blockIdx: classify by stamps only
                    const unsigned long long* s_ = &hs[((size_t)b * 2 + wsel) * 64];
                    if (s_[0] == 0 || s_[1] == 0) continue;
                    pro.push_back((double)(s_[1] - s_[0]));
                    int r = 0;
                    for (; r < 5 && s_[2 + r * 6] != 0; ++r) {
                        const unsigned long long* p_ = s_ + 2 + r * 6;
                        ph[r][0].push_back((double)(p_[1] - p_[0]));       // DMA wait
                        ph[r][1].push_back((double)(p_[2] - p_[1]));       // barrier
                        ph[r][2].push_back((double)(p_[3] - p_[2]));       // stage issue
                        if (p_[4] != 0) ph[r][3].push_back((double)(p_[4] - p_[3]));       // S + softmax
                        if (p_[5] != 0) ph[r][4].push_back((double)(p_[5] - p_[4]));       // P.V
                    }
                }
                printf("  blockIdx %% q_tiles == %d, wave %d: prologue %.0f;", tile, wsel ? 7 : 0, med(pro));
                for (int r = 0; r < 5; ++r)
                    if (!ph[r][0].empty())
                        printf("  round %d: wait %.0f bar %.0f stage %.0f S+softmax %.0f PV %.0f |", r, med(ph[r][0]), med(ph[r][1]), med(ph[r][2]), med(ph[r][3]), med(ph[r][4]));
                printf("\n");
            }
        }
        hipFree(st);
    }
    return 0;
}
#endif
