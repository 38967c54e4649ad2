// Fused QKV projection + self-attention of one encoder layer for short, unmasked
// sequences (the ViT-B/32 image towers: T = 50, 12 heads of 64).
//
// Restates CLIPAttention.forward (transformers models/clip/modeling_clip.py: q/k/v_proj with
// bias, softmax(q k^T / 8) v per head), the code the reference reaches through
// get_image_features (laser/models/llava_clip_model_v3.py:77-83), as ONE kernel: the
// [rows, 3d] QKV activation, which the separate kernels write to and read back from HBM
// (9 KB per token row and layer), never leaves the CU.
//
// Workgroup = (row tile, head).  A row tile is floor(256 / T) whole sequences (250 of 256 rows
// at T = 50); the B tile is the head's 64 q, 64 k and 64 v rows of the fused [3d, d] weight.
// Phase 1, GEMM 256 x 192 x d: 64-deep full-cache-line DMA stages, two slots, fragments double
//   buffered in registers, one barrier per stage (same pipeline as gemm_pipe64_kernel); 8 waves
//   (4 x 2) of 64 rows x 96 columns.  The MFMA operands are swapped (weights as A, activations
//   as B) so that a lane ends up with four consecutive columns of one row.
// Phase 2: q, k, v (+ bias) go to LDS as bf16, three swizzled [256][64] images in the dead ring.
// Phase 3: the waves take (sequence, 32-query tile) tasks; scores are computed transposed
//   (S^T = K Q^T, 32x32x16 MFMA) so the softmax is in-lane, and P^T feeds O^T += V^T P^T
//   directly; V^T fragments come out of the row-major V image with ds_read_b64_tr_b16.
#include "tile_common.h"

#include <stdlib.h>

namespace sgclip {

struct QkvAttnArgs {
    const bf16_t *A;     // [M, d] layer-norm output
    const bf16_t *W;     // [3d, d] fused q, k, v weight
    const float *bias;   // [3d]
    bf16_t *out;         // [M, d] attention output (heads concatenated)
    int lda, ldw;
    int M, n_seq, T, heads, d;
    int ipt;             // sequences per row tile
    int q_tiles;         // ceil(T / 32)
    int debug_mode;      // diagnostics only (SGCLIP_QKVATTN_DEBUG): 1 = stop after the GEMM, 2 = stop after the q/k/v images
    int head_group;      // heads per group of the workgroup order (0: all heads of a row tile side by side)
};

constexpr int QA_A = 256 * 128, QA_W = 192 * 128, QA_SLOT = QA_A + QA_W;  // 56 KiB
constexpr int QA_MAT = 256 * 128;                                         // one q / k / v image

template <bool SYNC, bool DMA, bool READ>
__device__ __forceinline__ void qa_substep(f32x4 (&acc)[4][6], const bf16x8 (&ca)[4], const bf16x8 (&cb)[6],
                                           bf16x8 (&na)[4], bf16x8 (&nb)[6], const unsigned char *rd_a,
                                           const unsigned char *rd_b, unsigned char *wr_slot,
                                           const bf16_t *const (&gsrc)[7], int wave) {
    if (SYNC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (DMA) {
#pragma unroll
        for (int q = 0; q < 7; q++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc[q]), LDS_PTR(wr_slot + (wave + 8 * q) * 1024), 16, 0, 0);
    }
    if (READ) {
#pragma unroll
        for (int i = 0; i < 4; i++) na[i] = *(const bf16x8 *)(rd_a + i * 2048);
#pragma unroll
        for (int j = 0; j < 6; j++) nb[j] = *(const bf16x8 *)(rd_b + j * 2048);
    }
    // weights as the A operand: acc[i][j][e] = C[row i*16 + (lane & 15)][column j*16 + (lane >> 4)*4 + e]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
    if (DMA) {
#pragma unroll
        for (int q = 0; q < 7; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);
        }
    }
    if (READ) {
        // at most one fragment read per two MFMAs where the MFMAs last (one per MFMA saturates the LDS)
#pragma unroll
        for (int q = 0; q < (DMA ? 7 : 10); q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
        if (DMA) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
            }
        }
    }
}

// Phases 2 and 3 (see the file header): the accumulators of the head's q, k, v tile go to LDS as three bf16 images over
// the dead operand ring, then the waves run the attention of the tile's sequences for this head.
__device__ __forceinline__ void qkv_attention_phases(const QkvAttnArgs &g, f32x4 (&acc)[4][6], unsigned char *lds, int wave,
                                                     int lane, int wm, int wn, int head, int seq0, int m0) {
    const int frow = lane & 15, fq = lane >> 4;
    const int d = g.d;
    if (g.debug_mode == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- phase 2: q, k, v (+ bias) as bf16 into three swizzled [256][64] LDS images ----
    __syncthreads();  // every wave has read its last fragments: the ring is free
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int n = wn * 96 + j * 16 + fq * 4;          // first of this lane's four columns, 0..191
        const int mat = n >> 6, c = n & 63;
        const float4 b = *(const float4 *)(g.bias + mat * d + head * 64 + c);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = wm * 64 + i * 16 + frow;
            bf16x4 v;
            v[0] = (bf16_t)(acc[i][j][0] + b.x);
            v[1] = (bf16_t)(acc[i][j][1] + b.y);
            v[2] = (bf16_t)(acc[i][j][2] + b.z);
            v[3] = (bf16_t)(acc[i][j][3] + b.w);
            *(bf16x4 *)(lds + mat * QA_MAT + lds_off(row, c >> 3) + (c & 7) * 2) = v;
        }
    }
    __syncthreads();
    if (g.debug_mode == 2) return;

    // ---- phase 3: attention of this head for the tile's sequences ----
    const int n_here = min(g.ipt, g.n_seq - seq0);
    const int n_tasks = n_here * g.q_tiles;
    const int T = g.T;
    const int r = lane & 31, h = lane >> 5;
    const unsigned char *qimg = lds, *kimg = lds + QA_MAT, *vimg = lds + 2 * QA_MAT;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int tq = i16 >> 2, tp = i16 & 3;
    for (int task = wave; task < n_tasks; task += 8) {
        const int img = task / g.q_tiles, qt = task - img * g.q_tiles;
        const int rb = img * T;            // first tile row of this sequence
        const int m = qt * 32 + r;         // this lane's query
        const int mrow = rb + (m < T ? m : T - 1);
        bf16x8 qf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) qf[s4] = *(const bf16x8 *)(qimg + lds_off(mrow, 2 * s4 + h));
        f32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; e++) o0[e] = o1[e] = 0.f;
        float m_run = NEG_BIG, l_run = 0.f;
        for (int n0 = 0; n0 < T; n0 += 32) {
            const int krow = rb + min(n0 + r, T - 1);
            bf16x8 kf[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) kf[s4] = *(const bf16x8 *)(kimg + lds_off(krow, 2 * s4 + h));
            f32x16 st;
#pragma unroll
            for (int e = 0; e < 16; e++) st[e] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s4], qf[s4], st, 0, 0, 0);
            bf16x8 pb[2];
            float alpha;
            if (n0 + 32 > T)
                alpha = online_softmax_step<true>(st, m_run, l_run, pb, n0, m, h, T, false, nullptr);
            else
                alpha = online_softmax_step<false>(st, m_run, l_run, pb, n0, m, h, T, false, nullptr);
            if (__ballot(alpha != 1.0f)) {
                scale_acc(o0, alpha);
                scale_acc(o1, alpha);
            }
            // V^T fragments: element j of k-step s2 is V[n0 + 16 s2 + 8 (j>>2) + 4 h + (j&3)][32 dt + r]; keys past the
            // sequence carry probability 0, their rows are clamped so that only finite values are multiplied by it
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int k_lo = rb + min(n0 + 16 * s2 + 4 * h + tq, T - 1);
                const int k_hi = rb + min(n0 + 16 * s2 + 4 * h + tq + 8, T - 1);
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const int chunk = 4 * dt + 2 * g16 + (tp >> 1);
                    const unsigned char *p_lo = vimg + lds_off(k_lo, chunk) + (tp & 1) * 8;
                    const unsigned char *p_hi = vimg + lds_off(k_hi, chunk) + (tp & 1) * 8;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p_lo);
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p_hi);
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
        if (m < T) {
            const float inv = 1.0f / l_run;
            bf16_t *orow = g.out + (size_t)(m0 + rb + m) * d + head * 64;
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
}

__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(QkvAttnArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 96 columns
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    // An XCD works through a contiguous run of the (row tile, head) sequence.  Head fastest, the 32 workgroups an XCD runs at
    // a time cover every head of 2-3 row tiles: the x tiles are fetched once, but the whole fused weight (3 d^2 bf16,
    // 3.5 MB at d = 768) streams through the 4 MiB L2 beside them and is fetched again for every row tile.  With the heads
    // in groups the run walks down the row tiles inside one group: the group's weight rows stay in the L2 and each x
    // tile is fetched once per group.
    int tm, head;
    if (g.head_group > 0 && g.head_group < g.heads) {
        const int tiles_m = gridDim.x / g.heads;
        const int per = tiles_m * g.head_group;
        const int grp = swz / per, r = swz - grp * per;
        const int width = min(g.head_group, g.heads - grp * g.head_group);
        tm = r / width;
        head = grp * g.head_group + (r - tm * width);
    } else {
        tm = swz / g.heads;
        head = swz - tm * g.heads;
    }
    const int seq0 = tm * g.ipt;
    const int m0 = seq0 * g.T;
    const int frow = lane & 15, fq = lane >> 4;
    const int d = g.d;

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][6];

    // this wave's seven pieces of a stage: q < 4: A pieces wave + 8 q (32 of them); q >= 4: W pieces wave + 8 (q - 4) (24)
    const bf16_t *gsrc[7];
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const bool isw = q >= 4;
        const int piece = wave + 8 * (isw ? q - 4 : q);
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (isw) {
            const int wrow = (row >> 6) * d + head * 64 + (row & 63);   // q, k or v block of this head
            gsrc[q] = g.W + (size_t)wrow * g.ldw + c * 8;
        } else {
            int grow = m0 + row;
            grow = grow < g.M ? grow : g.M - 1;
            gsrc[q] = g.A + (size_t)grow * g.lda + c * 8;
        }
    }
    const int a0 = lds_off(wm * 64 + frow, fq), b0 = QA_A + lds_off(wn * 96 + frow, fq);
    const int a1 = lds_off(wm * 64 + frow, 4 + fq), b1 = QA_A + lds_off(wn * 96 + frow, 4 + fq);

    const int ns = d / 64;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 7; q++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc[q] + t * 64), LDS_PTR(lds + t * QA_SLOT + (wave + 8 * q) * 1024), 16,
                                             0, 0);
#pragma unroll
    for (int q = 0; q < 7; q++) gsrc[q] += 128;
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; i++) fa[0][i] = *(const bf16x8 *)(lds + a0 + i * 2048);
#pragma unroll
    for (int j = 0; j < 6; j++) fb[0][j] = *(const bf16x8 *)(lds + b0 + j * 2048);

    int s = 0;
    for (; s < ns - 2; s++) {
        unsigned char *cur = lds + (s & 1) * QA_SLOT, *nxt = lds + ((s + 1) & 1) * QA_SLOT;
        qa_substep<false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, gsrc, wave);
        qa_substep<true, true, true>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0, cur, gsrc, wave);
#pragma unroll
        for (int q = 0; q < 7; q++) gsrc[q] += 64;
    }
    {
        unsigned char *cur = lds + (s & 1) * QA_SLOT, *nxt = lds + ((s + 1) & 1) * QA_SLOT;
        qa_substep<false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, gsrc, wave);
        qa_substep<true, false, true>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0, cur, gsrc, wave);
        qa_substep<false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], nxt + a1, nxt + b1, cur, gsrc, wave);
        qa_substep<false, false, false>(acc, fa[1], fb[1], fa[0], fb[0], nxt, nxt, cur, gsrc, wave);
    }

    qkv_attention_phases(g, acc, lds, wave, lane, wm, wn, head, seq0, m0);
}

// ---------------------------------------------------------------------------
// The same kernel with THREE slots for the activation operand (ns >= 4 stages).  The weights of a head are hot in the
// XCD's L2; the activation rows stream from HBM, and fetched one stage ahead the slowest of a stage's cache lines is
// still on its way when the workgroup reaches the stage barrier (the tower GEMMs run 11-19 % faster when their A rows
// are L2 hits).  LDS: A ring 3 x 32 KiB, W ring 2 x 24 KiB; at the barrier in stage s a wave queues the W fetch of stage
// s + 2, then the A fetch of stage s + 3, and the next barrier's counted wait leaves those four A fetches outstanding.
// Fetch sources are a scalar base plus one 32-bit lane offset per operand (p64_dma16).
// ---------------------------------------------------------------------------
constexpr int QD_A = 256 * 128, QD_WS = 192 * 128;     // stage bytes of A and of W
constexpr int QD_W = 3 * QD_A;                         // the W ring lies behind the three A slots
constexpr int QD_LDS = 3 * QD_A + 2 * QD_WS;           // 144 KiB

struct QdSrc {
    const unsigned char *abase, *wbase;   // uniform, advanced by 128 bytes per stage
    unsigned voff_a, voff_w, max_a;
    unsigned stride_a, stride_w;          // uniform: 64 A rows / one of the q, k, v blocks of W, in bytes
};

// NW: fetches the barrier's wait leaves outstanding.  FW / FA: queue this wave's three W / four A fetches behind the barrier.
template <bool SYNC, int NW, bool FW, bool FA, bool READ>
__device__ __forceinline__ void qd_substep(f32x4 (&acc)[4][6], const bf16x8 (&ca)[4], const bf16x8 (&cb)[6], bf16x8 (&na)[4],
                                           bf16x8 (&nb)[6], const unsigned char *rd_a, const unsigned char *rd_b,
                                           unsigned char *wr_w, unsigned char *wr_a, const QdSrc &src, int wave) {
    if (SYNC) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (FW || FA) {
        // eight groups of three MFMAs; the first seven open with one fetch (W, W, W, A, A, A, A), the ten fragment reads
        // are spread 2 1 1 1 2 1 1 1; fenced so that the compiler keeps the interleaving
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (q < 3 && FW)
                p64_dma16(src.wbase, src.voff_w, (unsigned)q * src.stride_w, 0xffffffffu,
                          (unsigned)(uintptr_t)LDS_PTR(wr_w + (wave + 8 * q) * 1024));
            if (q >= 3 && q < 7 && FA)
                p64_dma16<A_STREAMS_NT>(src.abase, src.voff_a, (unsigned)(q - 3) * src.stride_a, src.max_a,
                          (unsigned)(uintptr_t)LDS_PTR(wr_a + (wave + 8 * (q - 3)) * 1024));
            if (READ) {
                const int r0 = q + (q > 0) + (q > 4), r1 = q + 1 + (q + 1 > 0) + (q + 1 > 4);
#pragma unroll
                for (int r = r0; r < r1 && r < 10; r++) {
                    if (r < 4) na[r] = *(const bf16x8 *)(rd_a + r * 2048);
                    else nb[r - 4] = *(const bf16x8 *)(rd_b + (r - 4) * 2048);
                }
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int m = 3 * q + t, i = m / 6, j = m - 6 * i;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if (READ) {
#pragma unroll
        for (int i = 0; i < 4; i++) na[i] = *(const bf16x8 *)(rd_a + i * 2048);
#pragma unroll
        for (int j = 0; j < 6; j++) nb[j] = *(const bf16x8 *)(rd_b + j * 2048);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
    if (READ) {
#pragma unroll
        for (int q = 0; q < 10; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
    }
}

__global__ __launch_bounds__(512, 2) void qkv_attn_deep_kernel(QkvAttnArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 96 columns
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    int tm, head;   // workgroup order: see qkv_attn_kernel
    if (g.head_group > 0 && g.head_group < g.heads) {
        const int tiles_m = gridDim.x / g.heads;
        const int per = tiles_m * g.head_group;
        const int grp = swz / per, r = swz - grp * per;
        const int width = min(g.head_group, g.heads - grp * g.head_group);
        tm = r / width;
        head = grp * g.head_group + (r - tm * width);
    } else {
        tm = swz / g.heads;
        head = swz - tm * g.heads;
    }
    const int seq0 = tm * g.ipt;
    const int m0 = seq0 * g.T;
    const int frow = lane & 15, fq = lane >> 4;
    const int d = g.d;

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][6];

    // piece wave + 8 q holds tile rows 8 (wave + 8 q) .. + 7: A rows m0 + ..; W rows: block q (q, k or v) of this head, rows
    // 8 wave .. + 7 of the block.  The chunk swizzle depends on the piece's parity, the same for every q.
    QdSrc src;
    {
        const int row = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        src.voff_a = (unsigned)(row * g.lda + c * 8) * 2u;
        src.voff_w = (unsigned)(row * g.ldw + c * 8) * 2u;
        src.max_a = (unsigned)(min(g.M - 1 - m0, 255) * g.lda) * 2u + (unsigned)c * 16u;
    }
    src.stride_a = 128u * (unsigned)g.lda;
    src.stride_w = 2u * (unsigned)d * (unsigned)g.ldw;
    src.abase = (const unsigned char *)(g.A + (size_t)m0 * g.lda);
    src.wbase = (const unsigned char *)(g.W + (size_t)head * 64 * g.ldw);
    auto fetch_a = [&](unsigned char *slot) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            p64_dma16<A_STREAMS_NT>(src.abase, src.voff_a, (unsigned)q * src.stride_a, src.max_a, (unsigned)(uintptr_t)LDS_PTR(slot + (wave + 8 * q) * 1024));
        src.abase += 128;
    };
    auto fetch_w = [&](unsigned char *slot) {
#pragma unroll
        for (int q = 0; q < 3; q++)
            p64_dma16(src.wbase, src.voff_w, (unsigned)q * src.stride_w, 0xffffffffu, (unsigned)(uintptr_t)LDS_PTR(slot + (wave + 8 * q) * 1024));
        src.wbase += 128;
    };
    const int a0 = lds_off(wm * 64 + frow, fq), a1 = lds_off(wm * 64 + frow, 4 + fq);
    const int b0 = lds_off(wn * 96 + frow, fq), b1 = lds_off(wn * 96 + frow, 4 + fq);
    const int ns = d / 64;   // >= 4
    fetch_w(lds + QD_W);
    fetch_a(lds);
    fetch_w(lds + QD_W + QD_WS);
    fetch_a(lds + QD_A);
    fetch_a(lds + 2 * QD_A);
    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // stage 0 has landed: W(1), A(1), A(2) may be on their way
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; i++) fa[0][i] = *(const bf16x8 *)(lds + a0 + i * 2048);
#pragma unroll
    for (int j = 0; j < 6; j++) fb[0][j] = *(const bf16x8 *)(lds + QD_W + b0 + j * 2048);

    auto next3 = [](int i) { return i == 2 ? 0 : i + 1; };
    int ai = 0, wi = 0, s = 0;
    for (; s < ns - 3; s++) {   // fetches W(s + 2) and A(s + 3)
        unsigned char *ca = lds + ai * QD_A, *cw = lds + QD_W + wi * QD_WS;
        const int an = next3(ai);
        unsigned char *na = lds + an * QD_A, *nw = lds + QD_W + (wi ^ 1) * QD_WS;
        qd_substep<false, 0, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], ca + a1, cw + b1, cw, ca, src, wave);
        qd_substep<true, 4, true, true, true>(acc, fa[1], fb[1], fa[0], fb[0], na + a0, nw + b0, cw, ca, src, wave);
        src.abase += 128;
        src.wbase += 128;
        ai = an;
        wi ^= 1;
    }
    {   // stage ns - 3: the last W fetch, no A stage left
        unsigned char *ca = lds + ai * QD_A, *cw = lds + QD_W + wi * QD_WS;
        const int an = next3(ai);
        unsigned char *na = lds + an * QD_A, *nw = lds + QD_W + (wi ^ 1) * QD_WS;
        qd_substep<false, 0, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], ca + a1, cw + b1, cw, ca, src, wave);
        qd_substep<true, 4, true, false, true>(acc, fa[1], fb[1], fa[0], fb[0], na + a0, nw + b0, cw, ca, src, wave);
        ai = an;
        wi ^= 1;
    }
    {   // stages ns - 2 and ns - 1: nothing left to fetch, then nothing left to read
        unsigned char *ca = lds + ai * QD_A, *cw = lds + QD_W + wi * QD_WS;
        const int an = next3(ai);
        unsigned char *na = lds + an * QD_A, *nw = lds + QD_W + (wi ^ 1) * QD_WS;
        qd_substep<false, 0, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], ca + a1, cw + b1, cw, ca, src, wave);
        qd_substep<true, 0, false, false, true>(acc, fa[1], fb[1], fa[0], fb[0], na + a0, nw + b0, cw, ca, src, wave);
        qd_substep<false, 0, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], na + a1, nw + b1, cw, ca, src, wave);
        qd_substep<false, 0, false, false, false>(acc, fa[1], fb[1], fa[0], fb[0], na, nw, cw, ca, src, wave);
    }
    qkv_attention_phases(g, acc, lds, wave, lane, wm, wn, head, seq0, m0);
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_qkv_attention_supported(int T, int heads) { return T >= 1 && T <= 128 && heads >= 2 ? 1 : 0; }

extern "C" int sgclip_qkv_attention(const void *x_dev, int ldx, const void *w_qkv_dev, int ldw, const float *bias_dev,
                                    void *out_dev, int n_seq, int T, int heads, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seq == 0) return SGCLIP_OK;
    const int d = heads * 64;
    if (!x_dev || !w_qkv_dev || !bias_dev || !out_dev || n_seq < 0 || !sgclip_qkv_attention_supported(T, heads) || ldx < d ||
        ldw < d) {
        set_error("sgclip_qkv_attention: bad arguments n_seq=%d T=%d heads=%d ldx=%d ldw=%d (T <= 128, heads >= 2)", n_seq, T,
                  heads, ldx, ldw);
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)x_dev & 15) || ((uintptr_t)w_qkv_dev & 15) || ((uintptr_t)bias_dev & 15) || ((uintptr_t)out_dev & 7) ||
        (ldx & 7) || (ldw & 7)) {
        set_error("sgclip_qkv_attention: operands must be 16-byte aligned with leading dimensions a multiple of 8");
        return SGCLIP_EINVAL;
    }
    QkvAttnArgs g;
    g.A = (const bf16_t *)x_dev;
    g.W = (const bf16_t *)w_qkv_dev;
    g.bias = bias_dev;
    g.out = (bf16_t *)out_dev;
    g.lda = ldx;
    g.ldw = ldw;
    g.n_seq = n_seq;
    g.T = T;
    g.heads = heads;
    g.d = d;
    g.M = n_seq * T;
    g.ipt = 256 / T;
    g.q_tiles = ceil_div(T, 32);
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char *e = getenv("SGCLIP_QKVATTN_DEBUG");
            dbg = e ? atoi(e) : 0;
        }
        g.debug_mode = dbg;
        static int hg = -1;
        if (hg < 0) {
            const char *e = getenv("SGCLIP_QKV_HEAD_GROUP");
            hg = e ? atoi(e) : -2;
        }
        if (hg >= 0) {
            g.head_group = hg;
        } else {
            // Heads per group such that a group's q, k, v weight rows (3 x 64 x d bf16 per head) stay in an XCD's 4 MiB L2
            // beside the streams (2 MB allowed), in groups of equal width.  ViT-B: two groups of 6 (x is fetched twice, the
            // weights once: 27.7 GB per headline step at the L2's fabric side against 41.2 GB with all 12 heads side by
            // side, where every round of 32 workgroups re-fetches the 3.5 MB weight; the launch takes the same time either way).
            const int fit = (int)(2.0e6 / (3.0 * 64 * d * 2));
            const int groups = fit >= heads ? 1 : ceil_div(heads, fit < 1 ? 1 : fit);
            g.head_group = groups == 1 ? 0 : ceil_div(heads, groups);
        }
    }
    const long long tiles_m = ceil_div(n_seq, g.ipt);
    const long long grid = tiles_m * heads;
    if (grid > 0x7fffffffLL || (long long)n_seq * T > 0x7fffffffLL) {
        set_error("sgclip_qkv_attention: problem too large");
        return SGCLIP_EINVAL;
    }
    static int deep = -1;
    if (deep < 0) {
        const char *e = getenv("SGCLIP_QKV_DEEP_A");
        deep = e ? atoi(e) : 1;
    }
    KernelTimer timer(TK_GEMM0 + SGCLIP_EPI_BIAS_BF16, 2.0 * g.M * 3.0 * d * d + 4.0 * n_seq * heads * (double)T * T * 64, stream);
    if (deep && d >= 256 && (size_t)3 * d * ldw * 2 < (1ull << 32) && (size_t)256 * ldx * 2 < (1ull << 32)) {
        if (!lds_attr((const void *)qkv_attn_deep_kernel, QD_LDS)) return SGCLIP_ELAUNCH;
        hipLaunchKernelGGL(qkv_attn_deep_kernel, dim3((unsigned)grid), dim3(512), QD_LDS, stream, g);
        return launch_check("sgclip_qkv_attention");
    }
    if (!lds_attr((const void *)qkv_attn_kernel, 2 * QA_SLOT)) return SGCLIP_ELAUNCH;
    hipLaunchKernelGGL(qkv_attn_kernel, dim3((unsigned)grid), dim3(512), 2 * QA_SLOT, stream, g);
    return launch_check("sgclip_qkv_attention");
}
