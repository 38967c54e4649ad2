// Helpers shared by the tiled MFMA kernels (GEMM, attention, fused QKV + attention).
#pragma once
#include "common.h"

namespace sgclip {

// sched_group_barrier instruction classes
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_VMEM_RD 0x020
#define SG_DS_RD 0x100

// byte offset of 16-byte chunk `c` (0..7) of tile row `row` in a swizzled LDS image of 128-byte rows:
// ds_read_b128 of 16 consecutive rows x one chunk index touches every bank group once
__device__ __forceinline__ int lds_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// XCD-aware remap: blocks that share an XCD (same blockIdx % 8) walk neighbouring
// tiles of one A row panel, so the panel stays in that XCD's L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// LDS-DMA of 16 bytes per lane from base + min(voff + step, limit) to LDS at lds_addr + 16 * lane: `base` is uniform
// (SGPR pair), the lane offsets are 32 bits.  The offset arithmetic sits inside the assembly so that it is redone at every
// fetch in one transient register: left to the compiler the offsets of a stage's fetches are hoisted out of the stage
// loop into as many registers, which then spill (and a reload, a vector memory load, waits for the fetches before it).
// NT: the fetch carries the non-temporal hint.  Measured on the activation stream of the tower GEMMs and the fused q/k/v
// kernel (-DSGCLIP_A_NT=1): 3 % slower per headline step (the workgroups that share a row tile lose their L2 hits), so
// it is off; non-temporal loads / stores of the fp32 residual in the epilogue change nothing either way.
template <bool NT = false>
__device__ __forceinline__ void p64_dma16(const unsigned char *base, unsigned voff, unsigned step, unsigned limit,
                                          unsigned lds_addr) {
    unsigned off;
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %5\n\tv_add_u32 %0, %2, %1\n\tv_min_u32 %0, %0, %3\n\tglobal_load_lds_dwordx4 %0, %4 nt"
                     : "=&v"(off)
                     : "v"(voff), "s"(step), "v"(limit), "s"(base), "s"(lds_addr));
    else
        asm volatile("s_mov_b32 m0, %5\n\tv_add_u32 %0, %2, %1\n\tv_min_u32 %0, %0, %3\n\tglobal_load_lds_dwordx4 %0, %4"
                     : "=&v"(off)
                     : "v"(voff), "s"(step), "v"(limit), "s"(base), "s"(lds_addr));
}

#ifndef SGCLIP_A_NT
#define SGCLIP_A_NT 0
#endif
constexpr bool A_STREAMS_NT = SGCLIP_A_NT != 0;

constexpr float NEG_BIG = -1.0e30f;
// The running maximum of the online softmax is only a reference point: any value works as long as the exponentials stay
// in range.  It moves only when a block's maximum exceeds it by more than this many binary orders of magnitude, so the
// weights reach at most 2^8 (exact in bf16, harmless in the fp32 sums) and the output accumulators are rescaled once or
// twice per row instead of in almost every key block.
constexpr float SOFTMAX_SLACK = 8.0f;

// One key block of the online softmax, in the base-2 domain (scores * 0.125 * log2 e).
// st: S^T accumulators of this lane's query (register g = key n0 + (g&3) + 8*(g>>2) + 4*h); on
// return pb holds exp2(score - running max) as the bf16 B operand of O^T += V^T * P^T.
// MASKED = false is the interior-block path: every key valid, no comparisons.
// Returns the factor the caller scales its output accumulators by (1 where the running maximum did not move).  The
// accumulators are deliberately not touched in here: updating them under this function's branches makes the compiler
// keep them in two register ranges and copy all 32 registers back and forth on every key block.
template <bool MASKED>
__device__ __forceinline__ float online_softmax_step(f32x16 &st, float &m_run, float &l_run, bf16x8 (&pb)[2], int n0, int m,
                                                     int h, int T, bool causal, const int32_t *kmask) {
    constexpr float C = 0.125f * 1.4426950408889634f;
    float mx = NEG_BIG;
    float m_new;
    float psum = 0.f;
    if (MASKED) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            float t = st[g] * C;
            const int n = n0 + (g & 3) + 8 * (g >> 2) + 4 * h;
            bool ok = n < T;
            if (causal) ok = ok && (n <= m);
            if (kmask) ok = ok && (kmask[min(n, T - 1)] != 0);
            t = ok ? t : NEG_BIG;
            st[g] = t;
            mx = fmaxf(mx, t);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_new = mx > m_run + SOFTMAX_SLACK ? mx : m_run;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            float p = __builtin_amdgcn_exp2f(st[g] - m_new);
            p = st[g] > 0.5f * NEG_BIG ? p : 0.f;   // a row with no valid key yet has m_new == NEG_BIG
            st[g] = p;
            psum += p;
        }
    } else {
        // interior block: the maximum is taken on the raw scores (the scale is positive) and the scale rides in the
        // exponent's fused multiply-add, one vector operation per score fewer
#pragma unroll
        for (int g = 0; g < 16; g++) mx = fmaxf(mx, st[g]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_new = mx * C > m_run + SOFTMAX_SLACK ? mx * C : m_run;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[g], C, -m_new));
            st[g] = p;
            psum += p;
        }
    }
    psum += __shfl_xor(psum, 32, 64);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) pb[s2][j] = (bf16_t)st[8 * s2 + j];
    return alpha;
}

// o *= alpha on the packed fp32 multiplier (two registers per instruction)
__device__ __forceinline__ void scale_acc(f32x16 &o, float alpha) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a2 = {alpha, alpha};
#pragma unroll
    for (int g = 0; g < 16; g += 2) {
        f32x2 v = {o[g], o[g + 1]};
        v = v * a2;
        o[g] = v[0];
        o[g + 1] = v[1];
    }
}

}  // namespace sgclip
