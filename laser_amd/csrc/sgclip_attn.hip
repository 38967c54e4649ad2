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
#include "tile_common.h"

namespace sgclip {

constexpr int ATT_WAVES = 4;

struct AttnArgs {
    const bf16_t *qkv;
    bf16_t *out;
    const int32_t *key_mask;  // [n_seq, T] or null
    const int32_t *q_rows;    // null, or [n_seq]: the one query row (global row index) of every sequence
    int n_seq, T, heads, causal;
    int d;        // heads * 64
    int q_tiles;  // ceil(T / 32)
    long long n_items;  // n_seq * heads
    int remap;    // workgroup order: blocks of one (sequence, head) on one XCD
    int first_key_init;   // attn_shared_kernel: key 0 is the online softmax's initial state, the blocks cover keys 1 .. T - 1
};

__global__ __launch_bounds__(ATT_WAVES * 64, 3) void attn_kernel(AttnArgs a) {
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
    // key 0 as the initial state, the blocks over keys 1 .. T - 1: the same arithmetic, in the same order, as
    // attn_shared_kernel's (the pooled rows of a long unmasked sequence equal the full kernel's rows bit for bit)
    const int first = a.first_key_init;
    const int Tk = T - first;
    if (first) {
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const bf16x8 k0 = *(const bf16x8 *)(kbase + 16 * s + 8 * h);
#pragma unroll
            for (int j = 0; j < 8; j++) dot = __builtin_fmaf((float)qf[s][j], (float)k0[j], dot);
        }
        dot += __shfl_xor(dot, 32, 64);
        m_run = dot * (0.125f * 1.4426950408889634f);
        l_run = 1.0f;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const bf16x4 v0 = *(const bf16x4 *)(vbase + 32 * dt + 8 * gq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; e++) (dt == 0 ? o0 : o1)[4 * gq + e] = (float)v0[e];
            }
    }

    int n_end = Tk;
    if (a.causal) n_end = min(T, pooled ? m0 + 1 : m0 + 32);
    const int n_blocks = (n_end + 31) / 32;
    bf16_t *vt = vlds[wave];

    for (int kb = 0; kb < n_blocks; kb++) {
        const int n0 = kb * 32;
        // K fragments (A operand): lane holds K[n0 + r][16 s + 8 h .. +7]
        const int krow = min(n0 + r, Tk - 1) + first;
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) kf[s] = *(const bf16x8 *)(kbase + (size_t)krow * ld + 16 * s + 8 * h);
        // V block -> LDS, row major [32 keys][64]
        bf16x8 vreg[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int vr = min(n0 + 8 * t + (lane >> 3), Tk - 1) + first;
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

        bf16x8 pb[2];
        float alpha;
        if (n0 + 32 > Tk || kmask || (a.causal && n0 + 31 > m0))
            alpha = online_softmax_step<true>(st, m_run, l_run, pb, n0, m, h, Tk, a.causal, kmask);
        else
            alpha = online_softmax_step<false>(st, m_run, l_run, pb, n0, m, h, T, a.causal, kmask);
        if (__ballot(alpha != 1.0f)) {
            scale_acc(o0, alpha);
            scale_acc(o1, alpha);
        }

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


// ---------------------------------------------------------------------------
// Long sequences (T >= 128, ViT-L/14@336: T = 577): the four waves of a block take four
// consecutive 32-query tiles of ONE (sequence, head) and share every 32-key block of K and V
// through LDS (double buffered, filled one block ahead through registers), so K and V cross
// the L2 -> CU path once per 128 queries instead of once per 32.
// K tile image: [32 keys][128 B], 16-byte chunks XOR-swizzled with (key & 7) so that the
// ds_read_b128 fragment reads (32 keys x one chunk per half wave) are conflict free.
// V tile image: [32 keys][64] with the 16-byte chunk index XORed by 4 on rows 2, 3 mod 4, read transposed with
// ds_read_b64_tr_b16 (conflict free).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(ATT_WAVES * 64, 3) void attn_shared_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][32 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t vts[2][32 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q_groups = (a.q_tiles + ATT_WAVES - 1) / ATT_WAVES;
    // the blocks of one (sequence, head) on one XCD: its K and V rows cross the fabric once, not once per query group
    // (T = 577: 1 320 -> 1 240 us per 454 sequences)
    const int bid = a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qg = bid % q_groups;
    const long long item = bid / q_groups;
    const int seq = (int)(item / a.heads), head = (int)(item - (long long)seq * a.heads);
    const int T = a.T;
    const size_t ld = (size_t)3 * a.d;
    const bf16_t *qbase = a.qkv + (size_t)seq * T * ld + head * 64;
    const bf16_t *kbase = qbase + a.d;
    const bf16_t *vbase = qbase + 2 * a.d;
    const int32_t *kmask = a.key_mask ? a.key_mask + (size_t)seq * T : nullptr;

    const int qt = qg * ATT_WAVES + wave;
    const int m0 = qt * 32;
    const bool active = m0 < T;          // a wave past the end still helps to load K / V
    const int m = m0 + r;
    const int mrow = m < T ? m : T - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = *(const bf16x8 *)(qbase + (size_t)mrow * ld + 16 * s + 8 * h);

    f32x16 o0, o1;
#pragma unroll
    for (int g = 0; g < 16; g++) o0[g] = o1[g] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
    // Sequences of 32 n + 1 tokens (class token + a square of patches: 577, 257) would end in a key block with one key.
    // There key 0 becomes the INITIAL STATE of the online softmax instead -- reference point = its score, weight 1, output =
    // its value row -- and the blocks cover keys 1 .. T - 1, all of them full: one block and every masked step fewer.
    const int first = a.first_key_init;
    const int Tk = T - first;            // keys the blocks cover, starting at row `first`
    if (first) {
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const bf16x8 k0 = *(const bf16x8 *)(kbase + 16 * s + 8 * h);
#pragma unroll
            for (int j = 0; j < 8; j++) dot = __builtin_fmaf((float)qf[s][j], (float)k0[j], dot);
        }
        dot += __shfl_xor(dot, 32, 64);
        m_run = dot * (0.125f * 1.4426950408889634f);
        l_run = 1.0f;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const bf16x4 v0 = *(const bf16x4 *)(vbase + 32 * dt + 8 * gq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; e++) (dt == 0 ? o0 : o1)[4 * gq + e] = (float)v0[e];
            }
    }

    // keys the block needs: all of them, or up to its last query when causal
    const int blk_end = a.causal ? min(T, (qg * ATT_WAVES + ATT_WAVES) * 32) : Tk;
    const int n_blocks = (blk_end + 31) / 32;
    const int my_end = a.causal ? min(T, m0 + 32) : Tk;
    const int my_blocks = active ? (my_end + 31) / 32 : 0;

    // this thread's share of a K / V block: row tid >> 3 (32 rows), 16-byte chunk tid & 7
    const int lrow = threadIdx.x >> 3, lchunk = threadIdx.x & 7;
    auto fetch = [&](int kb, bf16x8 &kreg, bf16x8 &vreg) {
        const int row = min(kb * 32 + lrow, Tk - 1) + first;
        kreg = *(const bf16x8 *)(kbase + (size_t)row * ld + lchunk * 8);
        vreg = *(const bf16x8 *)(vbase + (size_t)row * ld + lchunk * 8);
    };
    auto stash = [&](int buf, const bf16x8 &kreg, const bf16x8 &vreg) {
        *(bf16x8 *)(kt[buf] + lrow * 64 + ((lchunk ^ (lrow & 7)) << 3)) = kreg;
        *(bf16x8 *)(vts[buf] + lrow * 64 + ((lchunk ^ (((lrow >> 1) & 1) << 2)) << 3)) = vreg;
    };
    bf16x8 kreg, vreg;
    fetch(0, kreg, vreg);
    stash(0, kreg, vreg);
    // The query fragments are "used" here so that their loads have landed, as far as the compiler's wait-count
    // bookkeeping knows, before the loop starts: with one of them possibly in flight at the loop head it waits where
    // the loop first reads those registers, in every iteration, and since the load counter retires in order that wait
    // also drains the K / V block just prefetched -- the whole global latency exposed per key block.
    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]) : "memory");
    if (n_blocks > 1) fetch(1, kreg, vreg);
    __syncthreads();

    // Two loops with one barrier per iteration each: the blocks this wave scores, then (causal towers, waves past the
    // end of the sequence) the blocks it only helps to load.  A single loop with the scoring under a condition makes
    // the compiler keep the 32 output accumulators in two register ranges and copy them back and forth every block.
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int tq = i16 >> 2, tp = i16 & 3;
    int kb = 0;
    for (; kb < my_blocks; kb++) {
        const int n0 = kb * 32;
        const bf16_t *kcur = kt[kb & 1], *vcur = vts[kb & 1];
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) kf[s] = *(const bf16x8 *)(kcur + r * 64 + (((2 * s + h) ^ (r & 7)) << 3));
        f32x16 st;
#pragma unroll
        for (int g = 0; g < 16; g++) st[g] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], st, 0, 0, 0);

        bf16x8 pb[2];
#if defined(ATTN_ABLATE) && (ATTN_ABLATE & 1)   // diagnostic: no softmax arithmetic
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pb[s2][j] = (bf16_t)st[8 * s2 + j];
        l_run += 1.0f;
#else
        float alpha;
        if (n0 + 32 > Tk || kmask || (a.causal && n0 + 31 > m0))
            alpha = online_softmax_step<true>(st, m_run, l_run, pb, n0, m, h, Tk, a.causal, kmask);
        else
            alpha = online_softmax_step<false>(st, m_run, l_run, pb, n0, m, h, T, a.causal, kmask);
        if (__ballot(alpha != 1.0f)) {
            scale_acc(o0, alpha);
            scale_acc(o1, alpha);
        }
#endif

#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
#if defined(ATTN_ABLATE) && (ATTN_ABLATE & 2)   // diagnostic: V fragments from the (conflict-free) K image, plain reads
                {
                    const bf16x8 vk = *(const bf16x8 *)(kcur + r * 64 + (((2 * s2 + dt * 4 + h) ^ (r & 7)) << 3));
                    if (dt == 0)
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vk, pb[s2], o0, 0, 0, 0);
                    else
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vk, pb[s2], o1, 0, 0, 0);
                    continue;
                }
#endif
                // 16-byte chunk XORed with 4 on rows 2, 3 mod 4: the four rows x 64 bytes a half wave reads then cover
                // all 64 banks once (rows are 128 bytes = half the bank array, so rows two apart would collide)
                const int vrow = 16 * s2 + 4 * h + tq;
                const bf16_t *p0 = vcur + vrow * 64 + (((4 * dt + 2 * g16 + (tp >> 1)) ^ (((vrow >> 1) & 1) << 2)) << 3) + 4 * (tp & 1);
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
        // the other buffer was last read in the previous iteration, which every wave has left
#if !(defined(ATTN_ABLATE) && (ATTN_ABLATE & 4))   // diagnostic: one K / V block reused, no refill, no barrier
        if (kb + 1 < n_blocks) stash((kb + 1) & 1, kreg, vreg);
        if (kb + 2 < n_blocks) fetch(kb + 2, kreg, vreg);
        __syncthreads();
#endif
    }
    for (; kb < n_blocks; kb++) {
        if (kb + 1 < n_blocks) stash((kb + 1) & 1, kreg, vreg);
        if (kb + 2 < n_blocks) fetch(kb + 2, kreg, vreg);
        __syncthreads();
    }

    if (active && m < T) {
        const float inv = 1.0f / l_run;
        bf16_t *orow = a.out + ((size_t)seq * T + m) * a.d + head * 64;
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
    {
        static int remap = -1;
        if (remap < 0) {
            const char *e = getenv("SGCLIP_ATTN_REMAP");
            remap = e ? atoi(e) : 1;
        }
        a.remap = remap;
        static int first_key = -1;
        if (first_key < 0) {
            const char *e = getenv("SGCLIP_ATTN_FIRST_KEY");
            first_key = e ? atoi(e) : 1;
        }
        a.first_key_init = first_key && T >= 128 && !causal && !key_mask_dev && T % 32 == 1;
    }
    if (!q_rows_dev && T >= 128) {
        const long long q_groups = (a.q_tiles + ATT_WAVES - 1) / ATT_WAVES;
        const long long nblk = a.n_items * q_groups;
        if (nblk > 0x7fffffffLL) {
            set_error("sgclip_attention: grid too large");
            return SGCLIP_EINVAL;
        }
        KernelTimer timer(TK_ATTN, 4.0 * n_seq * heads * (double)T * T * 64, stream);
        hipLaunchKernelGGL(attn_shared_kernel, dim3((unsigned)nblk), dim3(ATT_WAVES * 64), 0, stream, a);
        return launch_check("sgclip_attention");
    }
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
