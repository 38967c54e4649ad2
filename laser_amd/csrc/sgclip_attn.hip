// Fused multi-head attention for the CLIP towers (head_dim 64), any sequence
// length: T = 50 (ViT-B/32), 577 (ViT-L/14@336), 75 (text, causal + padding).
//
// One wave owns 32 query rows of one (sequence, head) and walks the keys in
// blocks of 32 with an online softmax.  Scores are computed TRANSPOSED,
// S^T = K_blk * Q^T (v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows), so that
// the query index sits on the lane and the key index in the 16 accumulator
// registers: the row max / row sum are in-lane reductions plus one exchange
// between the two lane halves, and the exponentiated tile, converted to bf16
// pairs of 8 registers, is directly the B operand of O^T += V^T * P^T with no
// lane movement.  Only V needs a transpose; it is staged through a wave-private
// 4 KiB LDS tile and read back with ds_read_b64_tr_b16.
// Q, K fragments are 16-byte loads straight from global memory (each 128-byte
// head row is one cache line, fully consumed by the wave).
#include "common.h"

namespace sgclip {

constexpr int ATT_WAVES = 4;
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
    const bf16_t *qkv;
    bf16_t *out;
    const int32_t *key_mask;  // [n_seq, T] or null
    const int32_t *q_rows;    // null, or [n_seq]: the one query row (global row index) of every sequence
    int n_seq, T, heads, causal;
    int d;        // heads * 64
    int q_tiles;  // ceil(T / 32)
    long long n_items;  // n_seq * heads
};

__global__ __launch_bounds__(ATT_WAVES * 64) void attn_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t vlds[ATT_WAVES][32 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // blockIdx.x = item_group * q_tiles + qt
    const int qt = blockIdx.x % a.q_tiles;
    long long item = (long long)(blockIdx.x / a.q_tiles) * ATT_WAVES + wave;
    const bool live = item < a.n_items;
    if (!live) item = a.n_items - 1;
    const int seq = (int)(item / a.heads), head = (int)(item - (long long)seq * a.heads);
    const int T = a.T;
    const size_t ld = (size_t)3 * a.d;
    const bf16_t *qbase = a.qkv + (size_t)seq * T * ld + head * 64;
    const bf16_t *kbase = qbase + a.d;
    const bf16_t *vbase = qbase + 2 * a.d;
    const int32_t *kmask = a.key_mask ? a.key_mask + (size_t)seq * T : nullptr;

    // pooled mode: a single query per sequence (every lane row carries it, row 0 is stored)
    const bool pooled = a.q_rows != nullptr;
    const int m0 = pooled ? a.q_rows[seq] - seq * T : qt * 32;
    const int m = pooled ? m0 : m0 + r;
    const int mrow = m < T ? m : T - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = *(const bf16x8 *)(qbase + (size_t)mrow * ld + 16 * s + 8 * h);

    f32x16 o0, o1;
#pragma unroll
    for (int g = 0; g < 16; g++) o0[g] = o1[g] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    int n_end = T;
    if (a.causal) n_end = min(T, pooled ? m0 + 1 : m0 + 32);
    const int n_blocks = (n_end + 31) / 32;
    bf16_t *vt = vlds[wave];

    for (int kb = 0; kb < n_blocks; kb++) {
        const int n0 = kb * 32;
        // K fragments (A operand): lane holds K[n0 + r][16 s + 8 h .. +7]
        const int krow = min(n0 + r, T - 1);
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) kf[s] = *(const bf16x8 *)(kbase + (size_t)krow * ld + 16 * s + 8 * h);
        // V block -> LDS, row major [32 keys][64]
        bf16x8 vreg[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int vr = min(n0 + 8 * t + (lane >> 3), T - 1);
            vreg[t] = *(const bf16x8 *)(vbase + (size_t)vr * ld + (lane & 7) * 8);
        }
        // The V tile is private to this wave and LDS operations of one wave execute in order, so
        // wave-level ordering suffices.  The full kernel still uses the block barrier (its four waves
        // have equal trip counts): keeping them in step measured faster; pooled causal rows differ in
        // trip count and must not meet at a block barrier.
        if (pooled) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 4; t++) *(bf16x8 *)(vt + (8 * t + (lane >> 3)) * 64 + (lane & 7) * 8) = vreg[t];

        f32x16 st;
#pragma unroll
        for (int g = 0; g < 16; g++) st[g] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], st, 0, 0, 0);

        // scale + mask; register g holds key n0 + (g&3) + 8*(g>>2) + 4*h for query m
        float mx = NEG_BIG;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int n = n0 + (g & 3) + 8 * (g >> 2) + 4 * h;
            bool ok = n < T;
            if (a.causal) ok = ok && (n <= m);
            if (kmask) ok = ok && (kmask[min(n, T - 1)] != 0);
            st[g] = ok ? st[g] * 0.125f : NEG_BIG;
            mx = fmaxf(mx, st[g]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float p = st[g] > 0.5f * NEG_BIG ? __expf(st[g] - m_new) : 0.f;
            st[g] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            o0[g] *= alpha;
            o1[g] *= alpha;
        }
        bf16x8 pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pb[s2][j] = (bf16_t)st[8 * s2 + j];

        if (pooled) {  // V tile written by all lanes of this wave
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
        // V^T fragments (A operand): element j of k-step s2 is V[16 s2 + 8 (j>>2) + 4 h + (j&3)][32 dt + r]
        const int i16 = lane & 15, g16 = (lane >> 4) & 1;
        const int tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                const bf16_t *p0 = vt + (16 * s2 + 4 * h + tq) * 64 + 32 * dt + 16 * g16 + 4 * tp;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p0);
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p0 + 8 * 64));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                if (dt == 0)
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s2], o0, 0, 0, 0);
                else
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s2], o1, 0, 0, 0);
            }
        }
    }

    if (live && (pooled ? r == 0 : m < T)) {
        const float inv = 1.0f / l_run;
        bf16_t *orow = a.out + (pooled ? (size_t)seq : (size_t)seq * T + m) * a.d + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (bf16_t)((dt == 0 ? o0[4 * gq + e] : o1[4 * gq + e]) * inv);
                *(bf16x4 *)(orow + 32 * dt + 8 * gq + 4 * h) = v;
            }
        }
    }
}

}  // namespace sgclip

using namespace sgclip;

static int attention_launch(const void *qkv_dev, const int32_t *q_rows_dev, void *out_dev, int n_seq, int T, int heads,
                            int causal, const int32_t *key_mask_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seq == 0) return SGCLIP_OK;
    if (!qkv_dev || !out_dev || n_seq < 0 || T <= 0 || heads <= 0) {
        set_error("sgclip_attention: bad arguments n_seq=%d T=%d heads=%d", n_seq, T, heads);
        return SGCLIP_EINVAL;
    }
    AttnArgs a;
    a.qkv = (const bf16_t *)qkv_dev;
    a.out = (bf16_t *)out_dev;
    a.key_mask = key_mask_dev;
    a.q_rows = q_rows_dev;
    a.n_seq = n_seq;
    a.T = T;
    a.heads = heads;
    a.causal = causal;
    a.d = heads * 64;
    a.q_tiles = q_rows_dev ? 1 : ceil_div(T, 32);
    a.n_items = (long long)n_seq * heads;
    const long long groups = (a.n_items + ATT_WAVES - 1) / ATT_WAVES;
    const long long blocks = groups * a.q_tiles;
    if (blocks > 0x7fffffffLL) {
        set_error("sgclip_attention: grid too large");
        return SGCLIP_EINVAL;
    }
    KernelTimer timer(TK_ATTN, 4.0 * n_seq * heads * (double)(q_rows_dev ? 1 : T) * T * 64, stream);
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)blocks), dim3(ATT_WAVES * 64), 0, stream, a);
    return launch_check("sgclip_attention");
}

extern "C" int sgclip_attention(const void *qkv_dev, void *out_dev, int n_seq, int T, int heads, int causal,
                                const int32_t *key_mask_dev, void *stream) {
    return attention_launch(qkv_dev, nullptr, out_dev, n_seq, T, heads, causal, key_mask_dev, stream);
}

extern "C" int sgclip_attention_pooled(const void *qkv_dev, const int32_t *q_rows_dev, void *out_dev, int n_seq, int T,
                                       int heads, int causal, const int32_t *key_mask_dev, void *stream) {
    if (!q_rows_dev) {
        set_error("sgclip_attention_pooled: q_rows_dev is null");
        return SGCLIP_EINVAL;
    }
    return attention_launch(qkv_dev, q_rows_dev, out_dev, n_seq, T, heads, causal, key_mask_dev, stream);
}
