// bf16 MFMA GEMM for the CLIP towers:  C = epilogue(A[M,K] * W[N,K]^T).
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights
// [out,in]), so A and W tiles are staged identically: rows x 64 k of bf16
// (128 B per row), written to LDS by global_load_lds_dwordx4 (1 KiB per wave
// instruction = 8 rows x 128 B).  Bank conflicts of the ds_read_b128 fragment
// reads are removed by an XOR swizzle of the 16-byte chunk index with bits 1..3
// of the row, applied on the per-lane global SOURCE address (the LDS image of a
// DMA is lane-linear) and again on the read address.
//
// Main kernel (gemm_ring_kernel): workgroup = 8 waves (4 x 2), tile 256 x 128,
// each wave 64 x 64 as 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators.  The K loop
// runs over a 3-slot LDS ring (3 x 48 KiB): the DMA of tile t+2 is issued right
// after the barrier that retires tile t-1's slot, and the wait before the barrier
// is a counted s_waitcnt vmcnt(6) that leaves tile t+1's six DMA instructions in
// flight across it (raw s_barrier, never __syncthreads(), which would drain them).
// Small problems (M <= 128) use the 128 x 128, 4-wave, 2-slot kernel.
// Epilogues fuse bias, quick-GELU, the fp32 residual add and the patch-embedding
// row remap + position embedding.
#include "tile_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace sgclip {

constexpr int BN = 128, BK = 64;

struct GemmArgs {
    const bf16_t *A;
    const bf16_t *W;
    void *C;
    const float *bias;
    const float *aux;
    int lda, ldw, ldc;
    int M, N, K;        // K: depth one workgroup walks (the whole K unless the launch splits it over blockIdx.y)
    int k_total;
    int aux_rows;
    int tiles_m, tiles_n;
    int wide_epilogue;  // N % 8 == 0, ldc % 8 == 0 and C, bias, aux 16-byte aligned
    int debug_mode;     // diagnostics only: 1 = skip epilogue stores, 2 = single K step, 3 / 4 = main-loop studies of
                        // gemm_pipe64_kernel, 5 = per-workgroup time trace, 6 = no epilogue (persistent kernels),
                        // 9 = every tile stores to the first 256 rows
    float *colsum;      // EPI_MULGRAD through epilogue_passes: += column sums of C (null: none)
    int col_group;      // gemm_pipe64_kernel: tiles walk down the rows inside groups of this many tile columns (0: row major)
    int tn_gm, tn_gn, tn_upx;   // gemm_tn_kernel: tile group shape and (group, slice) units per XCD (tn_gm = 0: plain grid)
};

// one 1 KiB piece (8 rows x 128 B) of an operand tile: lane -> (row, swizzled chunk)
__device__ __forceinline__ void stage_piece(const bf16_t *__restrict__ G, int ld, int row0, int max_row, int k0,
                                            unsigned char *lds_tile, int piece, int lane) {
    const int row = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < max_row ? grow : max_row - 1;
    const bf16_t *src = G + (size_t)grow * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
}

// x * sigmoid(1.702 x) with the hardware exp2 / rcp (relative error ~1e-6, far below bf16 output rounding)
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x));
}

// d/dx of x * sigmoid(1.702 x)
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x));
    return s + 1.702f * x * s * (1.0f - s);
}

constexpr int EPI_GELU_BOTH = SGCLIP_EPI_BIAS_GELU_BOTH_BF16, EPI_MULGRAD = SGCLIP_EPI_BIAS_MULGELUGRAD_BF16;
template <int EPI>
constexpr bool epi_has_bias() {
    return EPI == SGCLIP_EPI_BIAS_BF16 || EPI == SGCLIP_EPI_BIAS_GELU_BF16 || EPI == SGCLIP_EPI_BIAS_RESID_F32 || EPI == EPI_GELU_BOTH ||
           EPI == EPI_MULGRAD;
}
template <int EPI>
constexpr bool epi_bf16_out() {
    return EPI == SGCLIP_EPI_BIAS_BF16 || EPI == SGCLIP_EPI_BIAS_GELU_BF16 || EPI == EPI_GELU_BOTH || EPI == EPI_MULGRAD;
}

// split-K launches: workgroup row blockIdx.y owns K range [y * g.K, min((y + 1) * g.K, k_total))
__device__ __forceinline__ void split_k_setup(GemmArgs &g) {
    const int k0 = blockIdx.y * g.K;
    g.A += k0;
    g.W += k0;
    g.K = min(g.K, g.k_total - k0);
}

__device__ __forceinline__ void atomic_add4(float *p, float a, float b, float c, float d) {
    unsafeAtomicAdd(p, a);
    unsafeAtomicAdd(p + 1, b);
    unsafeAtomicAdd(p + 2, c);
    unsafeAtomicAdd(p + 3, d);
}

// one 64-deep K step of a wave's 64 x 64 sub-tile
__device__ __forceinline__ void mma_step(const unsigned char *la, const unsigned char *lb, int arow0, int brow0, int frow,
                                         int fq, f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *(const bf16x8 *)(la + lds_off(arow0 + i * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; j++) bfr[j] = *(const bf16x8 *)(lb + lds_off(brow0 + j * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
}

// acc[i][j][e] is C[mw + i*16 + fq*4 + e][nw + j*16 + frow]
template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs &g, const f32x4 (&acc)[4][4], int mw, int nw, int frow, int fq) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = nw + j * 16 + frow;
        if (n >= g.N) continue;
        float bias = 0.f;
        if (epi_has_bias<EPI>()) bias = g.bias[n];
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = mw + i * 16 + fq * 4 + e;
                if (m >= g.M) continue;
                const float v = acc[i][j][e] + bias;
                if (EPI == SGCLIP_EPI_BIAS_BF16) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] = (bf16_t)v;
                } else if (EPI == SGCLIP_EPI_BIAS_GELU_BF16) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] = (bf16_t)quick_gelu(v);
                } else if (EPI == EPI_GELU_BOTH) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] = (bf16_t)v;
                    ((bf16_t *)g.aux)[(size_t)m * g.ldc + n] = (bf16_t)quick_gelu((float)(bf16_t)v);
                } else if (EPI == EPI_MULGRAD) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] =
                        (bf16_t)(v * quick_gelu_grad((float)((const bf16_t *)g.aux)[(size_t)m * g.ldc + n]));
                } else if (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                    const size_t o = (size_t)m * g.ldc + n;
                    ((float *)g.C)[o] = (g.aux ? g.aux[o] : ((float *)g.C)[o]) + v;
                } else if (EPI == SGCLIP_EPI_PATCH_F32) {
                    const int P = g.aux_rows;
                    const int img = m / P, pp = m - img * P;
                    const size_t row = (size_t)img * (P + 1) + 1 + pp;
                    ((float *)g.C)[row * g.ldc + n] = v + g.aux[(size_t)(1 + pp) * g.ldc + n];
                } else if (EPI == SGCLIP_EPI_F32_ATOMIC) {
                    unsafeAtomicAdd((float *)g.C + (size_t)m * g.ldc + n, v);
                } else {
                    ((float *)g.C)[(size_t)m * g.ldc + n] = v;
                }
            }
        }
    }
}

// Coalesced epilogue: the wave parks its 64 x 64 fp32 sub-tile in a private LDS
// region (row stride 68 floats: the two row groups of a 32-lane half land 16 banks
// apart) and reads it back row-contiguous, so every global access is a float4 (or
// 4 x bf16) per lane and 256 (128) contiguous bytes per row.
constexpr int EPI_LD = 68;
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD * 4;  // 17 408

template <int EPI>
__device__ __forceinline__ void epilogue_lds(const GemmArgs &g, const f32x4 (&acc)[4][4], float *region, int mw, int nw,
                                             int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) region[(i * 16 + fq * 4 + e) * EPI_LD + j * 16 + frow] = acc[i][j][e];
    __syncthreads();
    if constexpr (EPI == SGCLIP_EPI_F32_ATOMIC) {
        // one dword per lane, 256 contiguous bytes of one row per instruction: the shape atomics run at full rate in
#pragma unroll 4
        for (int row = 0; row < 64; row++) {
            const int m = mw + row, n = nw + lane;
            if (m < g.M && n < g.N) unsafeAtomicAdd((float *)g.C + (size_t)m * g.ldc + n, region[row * EPI_LD + lane]);
        }
        return;
    }
    const int c4 = (lane & 15) * 4, r0 = lane >> 4;
    const int n = nw + c4;
    if (n >= g.N) return;  // N is a multiple of 4 wherever this epilogue is used
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (epi_has_bias<EPI>()) bias = *(const float4 *)(g.bias + n);
#pragma unroll 4
    for (int rr = 0; rr < 16; rr++) {
        const int row = rr * 4 + r0;
        const int m = mw + row;
        if (m >= g.M) continue;
        float4 v = *(const float4 *)(region + row * EPI_LD + c4);
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (epi_bf16_out<EPI>()) {
            if (EPI == SGCLIP_EPI_BIAS_GELU_BF16) {
                v.x = quick_gelu(v.x); v.y = quick_gelu(v.y); v.z = quick_gelu(v.z); v.w = quick_gelu(v.w);
            }
            if (EPI == EPI_MULGRAD) {
                const bf16x4 u = *(const bf16x4 *)((const bf16_t *)g.aux + (size_t)m * g.ldc + n);
                v.x *= quick_gelu_grad((float)u[0]); v.y *= quick_gelu_grad((float)u[1]);
                v.z *= quick_gelu_grad((float)u[2]); v.w *= quick_gelu_grad((float)u[3]);
            }
            bf16x4 o;
            o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
            *(bf16x4 *)((bf16_t *)g.C + (size_t)m * g.ldc + n) = o;
            if (EPI == EPI_GELU_BOTH) {
                bf16x4 a;
#pragma unroll
                for (int q = 0; q < 4; q++) a[q] = (bf16_t)quick_gelu((float)o[q]);
                *(bf16x4 *)((bf16_t *)g.aux + (size_t)m * g.ldc + n) = a;
            }
        } else if (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
            float4 *p = (float4 *)((float *)g.C + (size_t)m * g.ldc + n);
            const float4 c = g.aux ? *(const float4 *)(g.aux + (size_t)m * g.ldc + n) : *p;
            *p = make_float4(c.x + v.x, c.y + v.y, c.z + v.z, c.w + v.w);
        } else if (EPI == SGCLIP_EPI_PATCH_F32) {
            const int P = g.aux_rows;
            const int img = m / P, pp = m - img * P;
            const size_t orow = (size_t)img * (P + 1) + 1 + pp;
            const float4 a = *(const float4 *)(g.aux + (size_t)(1 + pp) * g.ldc + n);
            *(float4 *)((float *)g.C + orow * g.ldc + n) = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
        } else if (EPI == SGCLIP_EPI_F32_ATOMIC) {
            atomic_add4((float *)g.C + (size_t)m * g.ldc + n, v.x, v.y, v.z, v.w);
        } else {
            *(float4 *)((float *)g.C + (size_t)m * g.ldc + n) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// 128 x 128 tile, 4 waves, 2 LDS slots (small problems)
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int TILE = 128 * 128;  // bytes per operand tile
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * TILE];  // A0 B0 A1 B1
    split_k_setup(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = swz / g.tiles_n, tn = swz - tm * g.tiles_n;
    const int m0 = tm * 128, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kt, unsigned char *slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) stage_piece(g.A, g.lda, m0, g.M, kt * BK, slot, wave * 4 + i, lane);
#pragma unroll
        for (int i = 0; i < 4; i++) stage_piece(g.W, g.ldw, n0, g.N, kt * BK, slot + TILE, wave * 4 + i, lane);
    };
    const int nk = g.K / BK;
    stage(0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        unsigned char *la = lds + (kt & 1) * 2 * TILE;
        if (kt + 1 < nk) stage(kt + 1, lds + ((kt + 1) & 1) * 2 * TILE);
        mma_step(la, la + TILE, wm * 64, wn * 64, frow, fq, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    epilogue<EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, frow, fq);
}

// ---------------------------------------------------------------------------
// 256 x 128 tile, 8 waves (4 x 2), 3-slot LDS ring, counted vmcnt across a raw barrier
// ---------------------------------------------------------------------------
constexpr int RING_BM = 256;
constexpr int RING_A = RING_BM * 128, RING_B = BN * 128, RING_SLOT = RING_A + RING_B;  // 48 KiB
constexpr int RING_SLOTS = 3;
constexpr int RING_PW = (RING_SLOT / 1024) / 8;  // DMA instructions per wave per tile = 6

template <int EPI>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    split_k_setup(g);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = swz / g.tiles_n, tn = swz - tm * g.tiles_n;
    const int m0 = tm * RING_BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 48 pieces per tile: piece = wave + 8 i; i < 4 -> A (32 pieces), i = 4, 5 -> W (16 pieces)
    auto stage = [&](int kt, unsigned char *slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) stage_piece(g.A, g.lda, m0, g.M, kt * BK, slot, wave + 8 * i, lane);
#pragma unroll
        for (int i = 0; i < 2; i++) stage_piece(g.W, g.ldw, n0, g.N, kt * BK, slot + RING_A, wave + 8 * i, lane);
    };
    const int nk = g.K / BK;
    stage(0, lds);
    if (nk > 1) stage(1, lds + RING_SLOT);
    const int frow = lane & 15, fq = lane >> 4;
    int slot = 0;
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed once at most the DMAs of tile kt+1 are outstanding
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING_PW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; slot of tile kt-1 is free
        asm volatile("" ::: "memory");  // keep this iteration's LDS reads and DMA issue below the barrier
        if (kt + 2 < nk) {
            int ns = slot + 2;
            ns = ns >= RING_SLOTS ? ns - RING_SLOTS : ns;
            stage(kt + 2, lds + ns * RING_SLOT);
        }
        const unsigned char *la = lds + slot * RING_SLOT;
        __builtin_amdgcn_s_setprio(1);
        mma_step(la, la + RING_A, wm * 64, wn * 64, frow, fq, acc);
        __builtin_amdgcn_s_setprio(0);
        slot = slot + 1 == RING_SLOTS ? 0 : slot + 1;
    }
    if (g.wide_epilogue) {
        __syncthreads();  // every wave has finished reading the last tile: the ring becomes epilogue space
        epilogue_lds<EPI>(g, acc, (float *)(lds + wave * EPI_WAVE_BYTES), m0 + wm * 64, n0 + wn * 64, lane);
    } else {
        epilogue<EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, frow, fq);
    }
}

// ---------------------------------------------------------------------------
// 256 x 128 tile, 8 waves, K step 32, 3-slot ring of 24 KiB: two workgroups per CU, so
// one workgroup's epilogue (LDS round trip + global stores) runs under the other's MFMAs.
//
// LDS image of a 32-deep operand tile: 128-byte LDS row i holds tile row i (64 B) followed
// by tile row i + RH (RH = half the tile's rows); the 16-byte chunk positions of an LDS row
// are XOR-swizzled with bits 1..3 of i, which keeps the ds_read_b128 fragment reads
// conflict free exactly as in the 64-deep image.
// ---------------------------------------------------------------------------
constexpr int R32_A = 256 * 64, R32_B = BN * 64, R32_SLOT = R32_A + R32_B;  // 24 KiB
constexpr int R32_SLOTS = 3;
constexpr int R32_PW = (R32_SLOT / 1024) / 8;  // 3 DMA instructions per wave per tile

template <int RH>
__device__ __forceinline__ void stage_piece32(const bf16_t *__restrict__ G, int ld, int row0, int max_row, int k0,
                                              unsigned char *lds_tile, int piece, int lane) {
    const int lrow = piece * 8 + (lane >> 3);
    const int pos = (lane & 7) ^ ((lrow >> 1) & 7);  // logical position held by this lane's LDS chunk
    int grow = row0 + lrow + (pos >> 2) * RH;
    grow = grow < max_row ? grow : max_row - 1;
    const bf16_t *src = G + (size_t)grow * ld + k0 + (pos & 3) * 8;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
}

template <int RH>
__device__ __forceinline__ int lds_off32(int row, int c) {
    const int lrow = row & (RH - 1);
    const int pos = c + 4 * (row / RH);
    return lrow * 128 + ((pos ^ ((lrow >> 1) & 7)) << 4);
}

__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// 16-row passes of the coalesced epilogue: LDS bytes per wave for NJ column blocks
constexpr int epi_pass_bytes(int nj) { return 16 * (nj * 16 + 4) * 4; }

// NJ = 16-column blocks per wave: the wave's rows are NJ*16 contiguous columns, staged 16 rows at a
// time (row stride NJ*16 + 4 floats keeps the two row groups of a 32-lane half 16 banks apart).
template <int EPI, int NI, int NJ>
__device__ __forceinline__ void epilogue_passes(const GemmArgs &g, const f32x4 (&acc)[NI][NJ], float *region, int mw, int nw,
                                                int lane, unsigned long long *trace = nullptr) {
    constexpr int LDW = NJ * 16 + 4;
    if constexpr (EPI == SGCLIP_EPI_F32_ATOMIC) {
        // Atomic adds run at full rate only when a wave instruction covers contiguous bytes (one dword per lane,
        // 256 bytes of one row); with four columns per lane every 64-byte request would carry 16 useful bytes.
        const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int i = 0; i < NI; i++) {
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) region[(fq * 4 + e) * LDW + j * 16 + frow] = acc[i][j][e];
            wave_lds_sync();
#pragma unroll 4
            for (int row = 0; row < 16; row++) {
                const int m = mw + i * 16 + row;
#pragma unroll
                for (int cb = 0; cb < NJ * 16; cb += 64) {
                    const int n = nw + cb + lane;
                    const float v = region[row * LDW + cb + lane];
                    if (m < g.M && n < g.N) unsafeAtomicAdd((float *)g.C + (size_t)m * g.ldc + n, v);
                }
            }
            wave_lds_sync();
        }
        return;
    }
    constexpr bool BF = epi_bf16_out<EPI>();
    constexpr int CPL = BF ? 8 : 4;                // columns per lane
    constexpr int LPR = NJ * 16 / CPL;             // lanes per row
    constexpr int RPI = 64 / LPR;                  // rows per store instruction
    const int frow = lane & 15, fq = lane >> 4;
    const int cl = (lane % LPR) * CPL, rl = lane / LPR;
    const int n = nw + cl;
    float bias[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) bias[q] = 0.f;
    if (epi_has_bias<EPI>() && n < g.N) {   // the wide epilogue needs N % 8 == 0
#pragma unroll
        for (int q = 0; q < CPL; q += 4) {
            const float4 b = *(const float4 *)(g.bias + n + q);
            bias[q] = b.x; bias[q + 1] = b.y; bias[q + 2] = b.z; bias[q + 3] = b.w;
        }
    }
    float csum[EPI == EPI_MULGRAD ? CPL : 1];   // this lane's columns summed over the rows it stores
#pragma unroll
    for (int q = 0; q < (EPI == EPI_MULGRAD ? CPL : 1); q++) csum[q] = 0.f;
    // Epilogues that read a second input (the residual stream, the position rows, the pre-activation) fetch a whole
    // pass of it, 16 bytes per lane and row, one pass ahead of its use: issued next to its store, every fetch would
    // wait out the memory latency on its own, 16 / RPI round trips per pass with one KiB in flight per wave.
    constexpr bool SIDE = EPI == SGCLIP_EPI_BIAS_RESID_F32 || EPI == SGCLIP_EPI_PATCH_F32 || EPI == EPI_MULGRAD;
    constexpr int NR = 16 / RPI;                   // store instructions per pass
    float4 side[SIDE ? 2 : 1][SIDE ? NR : 1];
    const int n_in = n < g.N ? n : 0;
    auto side_fetch = [&](int i, float4 (&dst)[SIDE ? NR : 1]) {
        if constexpr (SIDE) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                int m = mw + i * 16 + rr * RPI + rl;
                m = m < g.M ? m : g.M - 1;
                const void *src;
                if constexpr (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                    src = (g.aux ? g.aux : (const float *)g.C) + (size_t)m * g.ldc + n_in;
                } else if constexpr (EPI == SGCLIP_EPI_PATCH_F32) {
                    src = g.aux + (size_t)(1 + m % g.aux_rows) * g.ldc + n_in;
                } else {
                    src = (const bf16_t *)g.aux + (size_t)m * g.ldc + n_in;
                }
                dst[rr] = *(const float4 *)src;
            }
        }
    };
    side_fetch(0, side[0]);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        if (i + 1 < NI) side_fetch(i + 1, side[SIDE ? (i + 1) & 1 : 0]);
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) region[(fq * 4 + e) * LDW + j * 16 + frow] = acc[i][j][e];
        // the region is private to this wave: its LDS writes only have to precede the reads of its
        // other lanes (a memory fence here would also wait for the previous pass's global stores)
        wave_lds_sync();
        if (i == 0) {
            // The bias must have arrived before the first bounds branch: a fetch still pending at a branch stays
            // pending on the path around it, and from there on the compiler waits for vmcnt(0) in front of every
            // use of it -- that is, for the previous global store's acknowledgement, 16 / RPI round trips per pass.
#pragma unroll
            for (int q = 0; q < CPL; q++) asm volatile("" : "+v"(bias[q]));
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int row = rr * RPI + rl;
            const int m = mw + i * 16 + row;
            float v[CPL];
#pragma unroll
            for (int q = 0; q < CPL; q += 4) {
                const float4 t = *(const float4 *)(region + row * LDW + cl + q);
                v[q] = t.x + bias[q]; v[q + 1] = t.y + bias[q + 1]; v[q + 2] = t.z + bias[q + 2]; v[q + 3] = t.w + bias[q + 3];
            }
            [[maybe_unused]] const float4 sd = side[SIDE ? i & 1 : 0][SIDE ? rr : 0];
            if (m >= g.M || n >= g.N) continue;
            if constexpr (BF) {
                bf16x8 o;
                if constexpr (EPI == EPI_MULGRAD) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sd);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        o[q] = (bf16_t)(v[q] * quick_gelu_grad((float)u[q]));
                        csum[q] += (float)o[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) o[q] = (bf16_t)(EPI == SGCLIP_EPI_BIAS_GELU_BF16 ? quick_gelu(v[q]) : v[q]);
                }
                // streamed past the L2's replacement order: the result is not read again by this launch, and written the
                // ordinary way it pushes the operand panels out (fc1 at 262 144 rows: 1 314 -> 1 217 us)
                __builtin_nontemporal_store(o, (bf16x8 *)((bf16_t *)g.C + (size_t)m * g.ldc + n));
                if constexpr (EPI == EPI_GELU_BOTH) {
                    bf16x8 a;
#pragma unroll
                    for (int q = 0; q < 8; q++) a[q] = (bf16_t)quick_gelu((float)o[q]);
                    *(bf16x8 *)((bf16_t *)g.aux + (size_t)m * g.ldc + n) = a;
                }
            } else if constexpr (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                *(float4 *)((float *)g.C + (size_t)m * g.ldc + n) = make_float4(sd.x + v[0], sd.y + v[1], sd.z + v[2], sd.w + v[3]);
            } else if constexpr (EPI == SGCLIP_EPI_PATCH_F32) {
                const int P = g.aux_rows;
                const int img = m / P, pp = m - img * P;
                const size_t orow = (size_t)img * (P + 1) + 1 + pp;
                *(float4 *)((float *)g.C + orow * g.ldc + n) = make_float4(v[0] + sd.x, v[1] + sd.y, v[2] + sd.z, v[3] + sd.w);
            } else if constexpr (EPI == SGCLIP_EPI_F32_ATOMIC) {
                atomic_add4((float *)g.C + (size_t)m * g.ldc + n, v[0], v[1], v[2], v[3]);
            } else {
                *(float4 *)((float *)g.C + (size_t)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        wave_lds_sync();
        if (trace) trace[i] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (EPI == EPI_MULGRAD) {
        if (g.colsum) {
            // the lanes of one column group are the RPI row lanes, LPR apart; the wave's NJ * 16 column sums are left at
            // the start of its region for the kernel to add up over the waves of a column strip
#pragma unroll
            for (int q = 0; q < CPL; q++) {
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) csum[q] += __shfl_xor(csum[q], o, 64);
            }
            if (rl == 0) {
#pragma unroll
                for (int q = 0; q < CPL; q++) region[cl + q] = csum[q];
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_ring32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    split_k_setup(g);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = swz / g.tiles_n, tn = swz - tm * g.tiles_n;
    const int m0 = tm * RING_BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 24 pieces per tile: piece = wave + 8 i; i = 0, 1 -> A (16 pieces), i = 2 -> W (8 pieces)
    auto stage = [&](int kt, unsigned char *slot) {
        stage_piece32<128>(g.A, g.lda, m0, g.M, kt * 32, slot, wave, lane);
        stage_piece32<128>(g.A, g.lda, m0, g.M, kt * 32, slot, wave + 8, lane);
        stage_piece32<64>(g.W, g.ldw, n0, g.N, kt * 32, slot + R32_A, wave, lane);
    };
    const int nk = g.K / 32;
    stage(0, lds);
    if (nk > 1) stage(1, lds + R32_SLOT);
    const int frow = lane & 15, fq = lane >> 4;
    int slot = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R32_PW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) {
            int ns = slot + 2;
            ns = ns >= R32_SLOTS ? ns - R32_SLOTS : ns;
            stage(kt + 2, lds + ns * R32_SLOT);
        }
        const unsigned char *la = lds + slot * R32_SLOT;
        const unsigned char *lb = la + R32_A;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *(const bf16x8 *)(la + lds_off32<128>(wm * 64 + i * 16 + frow, fq));
#pragma unroll
        for (int j = 0; j < 4; j++) bfr[j] = *(const bf16x8 *)(lb + lds_off32<64>(wn * 64 + j * 16 + frow, fq));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        slot = slot + 1 == R32_SLOTS ? 0 : slot + 1;
    }
    if (g.wide_epilogue) {
        __syncthreads();
        epilogue_passes<EPI, 4, 4>(g, acc, (float *)(lds + wave * epi_pass_bytes(4)), m0 + wm * 64, n0 + wn * 64, lane);
    } else {
        epilogue<EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, frow, fq);
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (4 x 2) of 64 rows x 128 columns, K step 32, 4-slot ring of 32 KiB.
// The larger wave tile needs 12 fragment reads per 32 MFMAs (the 64 x 64 tile: 8 per 16) and
// the larger block tile halves the DMA bytes per MFMA, which takes the LDS off the critical
// path; used when the grid still fills the chip.
// ---------------------------------------------------------------------------
constexpr int BIG_A = 256 * 64, BIG_SLOT = 2 * BIG_A;  // 32 KiB
constexpr int BIG_SLOTS = 4;
constexpr int BIG_PW = (BIG_SLOT / 1024) / 8;  // 4 DMA instructions per wave per tile

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    split_k_setup(g);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 128 columns
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = swz / g.tiles_n, tn = swz - tm * g.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 32 pieces per tile: piece = wave + 8 i; i = 0, 1 -> A, i = 2, 3 -> W
    auto stage = [&](int kt, unsigned char *slot) {
        stage_piece32<128>(g.A, g.lda, m0, g.M, kt * 32, slot, wave, lane);
        stage_piece32<128>(g.A, g.lda, m0, g.M, kt * 32, slot, wave + 8, lane);
        stage_piece32<128>(g.W, g.ldw, n0, g.N, kt * 32, slot + BIG_A, wave, lane);
        stage_piece32<128>(g.W, g.ldw, n0, g.N, kt * 32, slot + BIG_A, wave + 8, lane);
    };
    const int nk = g.debug_mode == 2 ? 1 : g.K / 32;
    stage(0, lds);
    if (nk > 1) stage(1, lds + BIG_SLOT);
    if (nk > 2) stage(2, lds + 2 * BIG_SLOT);
    const int frow = lane & 15, fq = lane >> 4;
    int slot = 0;
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed once at most the DMAs of tiles kt+1 and kt+2 are outstanding
        if (kt + 2 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BIG_PW) : "memory");
        else if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BIG_PW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 3 < nk) stage(kt + 3, lds + ((slot + 3) & 3) * BIG_SLOT);
        const unsigned char *la = lds + slot * BIG_SLOT;
        const unsigned char *lb = la + BIG_A;
        bf16x8 af[4], bfr[8];
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *(const bf16x8 *)(la + lds_off32<128>(wm * 64 + i * 16 + frow, fq));
#pragma unroll
        for (int j = 0; j < 8; j++) bfr[j] = *(const bf16x8 *)(lb + lds_off32<128>(wn * 128 + j * 16 + frow, fq));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        slot = (slot + 1) & 3;
    }
    if (g.debug_mode == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (g.wide_epilogue) {
        __syncthreads();
        epilogue_passes<EPI, 4, 8>(g, acc, (float *)(lds + wave * epi_pass_bytes(8)), m0 + wm * 64, n0 + wn * 128, lane);
    } else {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 part[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) part[i][j] = acc[i][half * 4 + j];
            epilogue<EPI>(g, part, m0 + wm * 64, n0 + wn * 128 + half * 64, frow, fq);
        }
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 tile, 64-deep stages: every DMA piece is 8 rows x 128 contiguous bytes, a whole
// cache line per row.  The L2 serves a CU a fixed number of requests per unit time whether a
// request is 64 or 128 bytes (measured: 64 GB/s per CU in 64-byte row pieces, 128 GB/s in
// 128-byte ones), so the 32-deep stages above fetch operands at half the rate.
// Two 64 KiB slots; a stage is two 32-deep MFMA sub-steps with register double-buffered
// fragments, and ONE barrier per stage: at the start of sub-step 1 of stage s every wave has
// read all of stage s (its slot takes the DMA of stage s+2) and stage s+1 has landed.
// Needs K % 64 == 0 and K >= 128.
// ---------------------------------------------------------------------------
constexpr int P64_OP = 256 * 128, P64_SLOT = 2 * P64_OP;  // 64 KiB

template <bool SYNC, bool DMA, bool READ, int DBG>
__device__ __forceinline__ void pipe64_substep(f32x4 (&acc)[4][8], const bf16x8 (&ca)[4], const bf16x8 (&cb)[8],
                                               bf16x8 (&na)[4], bf16x8 (&nb)[8], const unsigned char *rd_a,
                                               const unsigned char *rd_b, unsigned char *wr_slot,
                                               const bf16_t *const (&gsrc)[8], int wave) {
    if (SYNC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (DMA && DBG != 3) {
#pragma unroll
        for (int q = 0; q < 8; q++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc[q]), LDS_PTR(wr_slot + (wave + 8 * q) * 1024), 16, 0, 0);
    }
    if (READ) {
#pragma unroll
        for (int i = 0; i < 4; i++) na[i] = *(const bf16x8 *)(rd_a + i * 2048);
#pragma unroll
        for (int j = 0; j < 8; j++) nb[j] = *(const bf16x8 *)(rd_b + j * 2048);
    }
    if (DBG == 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" ::"v"(ca[i]));
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" ::"v"(cb[j]));
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
    if (DMA && DBG != 3) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);
        }
    }
    if (READ) {
#pragma unroll
        for (int q = 0; q < 12; q++) {
            // one fragment read per two MFMAs: at one per MFMA the two waves of a SIMD together ask the LDS for
            // 1 KiB per 16-cycle MFMA slot on every SIMD, which is the LDS's whole bandwidth (measured: 24 % slower)
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
    }
}

// debug_mode 5: every workgroup of gemm_pipe64_kernel leaves 16 records: {start, main loop end, end, hardware id, after the
// epilogue's barrier, after each of wave 0's four epilogue passes} in 10 ns ticks
constexpr int TRACE_WGS = 16384;
static __device__ unsigned long long g_pipe64_trace[16 * TRACE_WGS];

template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void gemm_pipe64_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const bool tracing = g.debug_mode == 5 && blockIdx.x < TRACE_WGS && threadIdx.x == 0;
    if (tracing) {
        g_pipe64_trace[16 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        g_pipe64_trace[16 * blockIdx.x + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                             (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    split_k_setup(g);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 128 columns
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    // An XCD works through a contiguous run of the tile sequence, 32 tiles at a time.  Row major, a wide N puts all of W
    // under every round; when W is larger than the XCD's L2 it is then fetched again every round.  Inside a group of
    // col_group tile columns the run walks down the rows instead, and the group's W panels stay in the L2.
    int tm, tn;
    if (g.col_group) {
        const int per = g.tiles_m * g.col_group;
        const int cg = swz / per, r = swz - cg * per;
        const int width = min(g.col_group, g.tiles_n - cg * g.col_group);
        tm = r / width;
        tn = cg * g.col_group + (r - tm * width);
    } else {
        tm = swz / g.tiles_n;
        tn = swz - tm * g.tiles_n;
    }
    const int m0 = tm * 256, n0 = tn * 256;
    const int frow = lane & 15, fq = lane >> 4;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][8];

    // this wave's eight pieces of a stage: pieces wave + 8 q, q < 4 of A (rows 8 piece ..), q >= 4 of W
    const bf16_t *gsrc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int piece = wave + 8 * (q & 3);
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const bool isw = q >= 4;
        int grow = (isw ? n0 : m0) + row;
        const int lim = isw ? g.N : g.M;
        grow = grow < lim ? grow : lim - 1;
        gsrc[q] = (isw ? g.W + (size_t)grow * g.ldw : g.A + (size_t)grow * g.lda) + c * 8;
    }
    // LDS piece order inside a slot: A pieces 0..31, then W pieces 0..31 -> piece (wave + 8 q) for q < 4, +32 for q >= 4
    // (wr_slot + (wave + 8 q) * 1024 with q = 4..7 lands in the W half because 8 * 4 = 32 pieces = P64_OP bytes)
    const int a0 = lds_off(wm * 64 + frow, fq), b0 = P64_OP + lds_off(wn * 128 + frow, fq);
    const int a1 = lds_off(wm * 64 + frow, 4 + fq), b1 = P64_OP + lds_off(wn * 128 + frow, 4 + fq);

    const int ns = g.K / 64;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 8; q++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc[q] + t * 64), LDS_PTR(lds + t * P64_SLOT + (wave + 8 * q) * 1024),
                                             16, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) gsrc[q] += 128;  // next stage to fetch: 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; i++) fa[0][i] = *(const bf16x8 *)(lds + a0 + i * 2048);
#pragma unroll
    for (int j = 0; j < 8; j++) fb[0][j] = *(const bf16x8 *)(lds + b0 + j * 2048);

    int s = 0;
    for (; s < ns - 2; s++) {
        unsigned char *cur = lds + (s & 1) * P64_SLOT, *nxt = lds + ((s + 1) & 1) * P64_SLOT;
        pipe64_substep<false, false, true, DBG>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, gsrc, wave);
        pipe64_substep<true, true, true, DBG>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0, cur, gsrc, wave);
#pragma unroll
        for (int q = 0; q < 8; q++) gsrc[q] += 64;
    }
    {   // stage ns - 2: nothing left to fetch; stage ns - 1: nothing left to read
        unsigned char *cur = lds + (s & 1) * P64_SLOT, *nxt = lds + ((s + 1) & 1) * P64_SLOT;
        pipe64_substep<false, false, true, DBG>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, gsrc, wave);
        pipe64_substep<true, false, true, DBG>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0, cur, gsrc, wave);
        pipe64_substep<false, false, true, DBG>(acc, fa[0], fb[0], fa[1], fb[1], nxt + a1, nxt + b1, cur, gsrc, wave);
        pipe64_substep<false, false, false, DBG>(acc, fa[1], fb[1], fa[0], fb[0], nxt, nxt, cur, gsrc, wave);
    }
    if (DBG != 0) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (tracing) g_pipe64_trace[16 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    if (g.wide_epilogue) {
        __syncthreads();
        // debug_mode 9 (diagnostic, wrong results): every tile stores to rows 0..255, which stay in L2 -- the epilogue
        // without its HBM traffic
        if (tracing) g_pipe64_trace[16 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
        epilogue_passes<EPI, 4, 8>(g, acc, (float *)(lds + wave * epi_pass_bytes(8)), (g.debug_mode == 9 ? 0 : m0) + wm * 64,
                                   n0 + wn * 128, lane, tracing ? g_pipe64_trace + 16 * blockIdx.x + 5 : nullptr);
        if constexpr (EPI == EPI_MULGRAD) {
            if (g.colsum) {   // uniform over the grid
                // one atomic per column of the tile, a wave's 64 on neighbouring addresses (atomics that share a cache
                // line within one instruction cost one; scattered ones serialise)
                __syncthreads();
                if (tid < 256 && n0 + tid < g.N) {
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++) t += ((const float *)(lds + (r * 2 + (tid >> 7)) * epi_pass_bytes(8)))[tid & 127];
                    unsafeAtomicAdd(g.colsum + n0 + tid, t);
                }
            }
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 part[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) part[i][j] = acc[i][half * 4 + j];
            epilogue<EPI>(g, part, m0 + wm * 64, n0 + wn * 128 + half * 64, frow, fq);
        }
    }
    if (tracing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores acknowledged
        g_pipe64_trace[16 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------
// The same pipeline as a persistent kernel: one workgroup per CU walks through its share of the tiles, and the
// operand stream never drains.  The last two stages of a tile already fetch the first two stages of the next one, so a
// tile pays neither the workgroup turnaround (1-2 us) nor the latency of its first fetch (1.5-2 us), and the epilogue
// needs no workgroup barrier: it is staged in 4 KiB per wave beyond the ring (8 rows x 128 columns of fp32, 16-column
// groups XOR-swizzled by the lane quarter instead of padded rows), so no wave waits for the others to leave the ring.
// ---------------------------------------------------------------------------
constexpr int P64P_STAGE = 4096;                                  // epilogue staging bytes per wave
constexpr int P64P_LDS = 2 * P64_SLOT + 8 * P64P_STAGE;           // 160 KiB

// Eight-row epilogue passes of one wave's 64 x 128 block: pass (i, h) takes the rows 16 i + 4 fq + 2 h + {0, 1}.
// SPLIT: the staging rows lie in four separate 1 KiB pieces, 8 KiB apart (two rows each): the wave's own four DMA
// pieces of an operand slot (gemm_pipe64q_kernel), so that the fetch it queues into that slot right after its epilogue can
// only overwrite its own, finished staging.
template <int EPI, bool SPLIT = false>
__device__ __forceinline__ void epilogue_rows8(const GemmArgs &g, const f32x4 (&acc)[4][8], float *region, int mw, int nw,
                                               int lane) {
#define ROWF(lr) (SPLIT ? ((lr) >> 1) * 2048 + ((lr) & 1) * 128 : (lr) * 128)   /* float offset of staging row lr */
    constexpr bool BF = epi_bf16_out<EPI>();
    constexpr int CPL = BF ? 8 : 4;                // columns per lane
    constexpr int LPR = 128 / CPL;                 // lanes per row
    constexpr int RPI = 64 / LPR;                  // rows per store instruction
    constexpr int NR = 8 / RPI;                    // store instructions per pass
    constexpr int NP = 8;                          // passes
    constexpr bool SIDE = EPI == SGCLIP_EPI_BIAS_RESID_F32 || EPI == SGCLIP_EPI_PATCH_F32 || EPI == EPI_MULGRAD;
    constexpr int DEPTH = SIDE ? (BF ? 4 : 4) : 1; // passes of the second input in flight
    const int frow = lane & 15, fq = lane >> 4;
    const int cl = (lane % LPR) * CPL, rl = lane / LPR;
    const int n = nw + cl;
    const int n_in = n < g.N ? n : 0;
    float bias[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) bias[q] = 0.f;
    if (epi_has_bias<EPI>() && n < g.N) {
#pragma unroll
        for (int q = 0; q < CPL; q += 4) {
            const float4 b = *(const float4 *)(g.bias + n + q);
            bias[q] = b.x; bias[q + 1] = b.y; bias[q + 2] = b.z; bias[q + 3] = b.w;
        }
    }
    float csum[EPI == EPI_MULGRAD ? CPL : 1];
#pragma unroll
    for (int q = 0; q < (EPI == EPI_MULGRAD ? CPL : 1); q++) csum[q] = 0.f;
    // local row lr of pass (i, h) is tile row 16 i + 4 (lr >> 1) + 2 h + (lr & 1)
    auto tile_row = [&](int p, int lr) { return (p >> 1) * 16 + (lr >> 1) * 4 + (p & 1) * 2 + (lr & 1); };
    float4 side[DEPTH][SIDE ? NR : 1];
    auto side_fetch = [&](int p, float4 (&dst)[SIDE ? NR : 1]) {
        if constexpr (SIDE) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                int m = mw + tile_row(p, rr * RPI + rl);
                m = m < g.M ? m : g.M - 1;
                const void *src;
                if constexpr (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                    src = (g.aux ? g.aux : (const float *)g.C) + (size_t)m * g.ldc + n_in;
                } else if constexpr (EPI == SGCLIP_EPI_PATCH_F32) {
                    src = g.aux + (size_t)(1 + m % g.aux_rows) * g.ldc + n_in;
                } else {
                    src = (const bf16_t *)g.aux + (size_t)m * g.ldc + n_in;
                }
                dst[rr] = *(const float4 *)src;
            }
        }
    };
    if constexpr (SIDE) {
#pragma unroll
        for (int p = 0; p < DEPTH; p++) side_fetch(p, side[p]);
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int i = p >> 1, h = p & 1;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 2; e++) region[ROWF(fq * 2 + e) + ((j ^ fq) << 4) + frow] = acc[i][j][2 * h + e];
        wave_lds_sync();
        if (p == 0) {
            // the bias must have arrived before the first bounds branch (a fetch pending at a branch stays pending on the
            // path around it, and every later use then waits for vmcnt(0), i.e. for the previous store's acknowledgement)
#pragma unroll
            for (int q = 0; q < CPL; q++) asm volatile("" : "+v"(bias[q]));
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int lr = rr * RPI + rl;
            const int m = mw + tile_row(p, lr);
            float v[CPL];
#pragma unroll
            for (int q = 0; q < CPL; q += 4) {
                const float4 t = *(const float4 *)(region + ROWF(lr) + ((cl + q) ^ ((lr >> 1) << 4)));
                v[q] = t.x + bias[q]; v[q + 1] = t.y + bias[q + 1]; v[q + 2] = t.z + bias[q + 2]; v[q + 3] = t.w + bias[q + 3];
            }
            [[maybe_unused]] const float4 sd = side[SIDE ? p % DEPTH : 0][SIDE ? rr : 0];
            if (m >= g.M || n >= g.N) continue;
            if constexpr (BF) {
                bf16x8 o;
                if constexpr (EPI == EPI_MULGRAD) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sd);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        o[q] = (bf16_t)(v[q] * quick_gelu_grad((float)u[q]));
                        csum[q] += (float)o[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) o[q] = (bf16_t)(EPI == SGCLIP_EPI_BIAS_GELU_BF16 ? quick_gelu(v[q]) : v[q]);
                }
                __builtin_nontemporal_store(o, (bf16x8 *)((bf16_t *)g.C + (size_t)m * g.ldc + n));
                if constexpr (EPI == EPI_GELU_BOTH) {
                    bf16x8 a;
#pragma unroll
                    for (int q = 0; q < 8; q++) a[q] = (bf16_t)quick_gelu((float)o[q]);
                    *(bf16x8 *)((bf16_t *)g.aux + (size_t)m * g.ldc + n) = a;
                }
            } else if constexpr (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                *(float4 *)((float *)g.C + (size_t)m * g.ldc + n) = make_float4(sd.x + v[0], sd.y + v[1], sd.z + v[2], sd.w + v[3]);
            } else if constexpr (EPI == SGCLIP_EPI_PATCH_F32) {
                const int P = g.aux_rows;
                const int img = m / P, pp = m - img * P;
                const size_t orow = (size_t)img * (P + 1) + 1 + pp;
                *(float4 *)((float *)g.C + orow * g.ldc + n) = make_float4(v[0] + sd.x, v[1] + sd.y, v[2] + sd.z, v[3] + sd.w);
            } else {
                *(float4 *)((float *)g.C + (size_t)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (p + DEPTH < NP) side_fetch(p + DEPTH, side[SIDE ? p % DEPTH : 0]);
        wave_lds_sync();
    }
    if constexpr (EPI == EPI_MULGRAD) {
        if (g.colsum) {
#pragma unroll
            for (int q = 0; q < CPL; q++) {
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) csum[q] += __shfl_xor(csum[q], o, 64);
            }
            if (rl == 0) {
#pragma unroll
                for (int q = 0; q < CPL; q++) region[cl + q] = csum[q];
            }
        }
    }
}

// DMA sources of the persistent kernel (see there)
struct P64Src {
    const unsigned char *abase, *wbase;   // uniform
    unsigned voff_a, voff_b, max_a, max_b, chunk;
    unsigned stride_a, stride_b;          // uniform: 64 rows in bytes
};

__device__ __forceinline__ void pipe64p_fetch(const P64Src &t, unsigned char *wr_slot, int wave) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const bool isw = q >= 4;
        p64_dma16(isw ? t.wbase : t.abase, isw ? t.voff_b : t.voff_a, (unsigned)(q & 3) * (isw ? t.stride_b : t.stride_a),
                  isw ? t.max_b : t.max_a, (unsigned)(uintptr_t)LDS_PTR(wr_slot + (wave + 8 * q) * 1024));
    }
}

// pipe64_substep with the sources above.  STORES > 0 with `newest_are_stores` set (uniform): the wave's STORES most recent
// vector memory instructions are result stores of the previous tile, issued after every fetch this barrier is about; the
// counter retires in order, so leaving that many outstanding still covers the fetches, and the stage does not wait for
// the stores' acknowledgement.
template <bool SYNC, bool DMA, bool READ, int DBG, int STORES = 0>
__device__ __forceinline__ void pipe64p_substep(f32x4 (&acc)[4][8], const bf16x8 (&ca)[4], const bf16x8 (&cb)[8],
                                                bf16x8 (&na)[4], bf16x8 (&nb)[8], const unsigned char *rd_a,
                                                const unsigned char *rd_b, unsigned char *wr_slot, const P64Src &src, int wave,
                                                bool newest_are_stores = false) {
    if (SYNC) {
        if (STORES > 0 && newest_are_stores)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (DMA) {
        // The eight fetches of a stage go out one per four MFMAs (issued back to back behind the barrier, the 64 fetches of
        // the eight waves reach the CU's one address unit together).  The groups are fenced so that the compiler keeps the
        // interleaving: the fetches are opaque assembly to it.  Worth 2 % on the q/k/v shape, nothing on the others.
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool isw = q >= 4;
            p64_dma16(isw ? src.wbase : src.abase, isw ? src.voff_b : src.voff_a, (unsigned)(q & 3) * (isw ? src.stride_b : src.stride_a),
                      isw ? src.max_b : src.max_a, (unsigned)(uintptr_t)LDS_PTR(wr_slot + (wave + 8 * q) * 1024));
            if (READ) {   // twelve fragment reads over the eight groups: 2 1 2 1 2 1 2 1
                const int r0 = (3 * q + 1) / 2, r1 = (3 * (q + 1) + 1) / 2;
#pragma unroll
                for (int r = r0; r < r1; r++) {
                    if (r < 4) na[r] = *(const bf16x8 *)(rd_a + r * 2048);
                    else nb[r - 4] = *(const bf16x8 *)(rd_b + (r - 4) * 2048);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int i = q >> 1, j = 4 * (q & 1) + t;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if (READ) {
#pragma unroll
        for (int i = 0; i < 4; i++) na[i] = *(const bf16x8 *)(rd_a + i * 2048);
#pragma unroll
        for (int j = 0; j < 8; j++) nb[j] = *(const bf16x8 *)(rd_b + j * 2048);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
    if (READ) {
#pragma unroll
        for (int q = 0; q < 12; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
    }
}

// vector memory stores one wave issues per full tile in epilogue_rows8, all after its last fetch
template <int EPI>
constexpr int epilogue_rows8_stores() {
    return epi_bf16_out<EPI>() ? (EPI == EPI_GELU_BOTH ? 32 : 16) : 32;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pipe64p_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 128 columns
    const int frow = lane & 15, fq = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    // virtual block index: the place this tile would have in a grid of one workgroup per tile (same XCD-aware order)
    int vb = blockIdx.x;
    int m0, n0;
    auto tile_origin = [&](int b, int &mo, int &no) {
        const int swz = xcd_remap(b, ntiles);
        int tm, tn;
        if (g.col_group) {
            const int per = g.tiles_m * g.col_group;
            const int cg = swz / per, r = swz - cg * per;
            const int width = min(g.col_group, g.tiles_n - cg * g.col_group);
            tm = r / width;
            tn = cg * g.col_group + (r - tm * width);
        } else {
            tm = swz / g.tiles_n;
            tn = swz - tm * g.tiles_n;
        }
        mo = tm * 256;
        no = tn * 256;
    };
    // This wave's eight pieces of a stage: pieces wave + 8 q, q < 4 of A (rows 8 piece ..), q >= 4 of W.  The source of a
    // piece is a uniform pointer (the tile's first row at the stage's depth, in SGPRs, advanced by scalar adds) plus a
    // 32-bit lane offset: piece 0's offset + 64 rows per further piece, limited to the operand's last row.  Four VGPRs
    // for all sixteen sources; per-lane 64-bit pointers (sixteen) do not fit beside the accumulators and the two
    // fragment sets, and a reload inside the loop waits for the DMA issued before it.
    P64Src src;
    {
        const int row = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);   // the same for every piece: 64 rows further leaves (row >> 1) & 7
        src.voff_a = (unsigned)(row * g.lda + c * 8) * 2u;
        src.voff_b = (unsigned)(row * g.ldw + c * 8) * 2u;
        src.chunk = (unsigned)c * 16u;
    }
    src.stride_a = 128u * (unsigned)g.lda;
    src.stride_b = 128u * (unsigned)g.ldw;
    auto set_sources = [&](int mo, int no) {
        src.abase = (const unsigned char *)(g.A + (size_t)mo * g.lda);
        src.wbase = (const unsigned char *)(g.W + (size_t)no * g.ldw);
        src.max_a = (unsigned)(min(g.M - 1 - mo, 255) * g.lda) * 2u + src.chunk;
        src.max_b = (unsigned)(min(g.N - 1 - no, 255) * g.ldw) * 2u + src.chunk;
    };
    tile_origin(vb, m0, n0);
    set_sources(m0, n0);
    const int a0 = lds_off(wm * 64 + frow, fq), b0 = P64_OP + lds_off(wn * 128 + frow, fq);
    const int a1 = lds_off(wm * 64 + frow, 4 + fq), b1 = P64_OP + lds_off(wn * 128 + frow, 4 + fq);
    const int ns = g.K / 64;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        pipe64p_fetch(src, lds + t * P64_SLOT, wave);
        src.abase += 128;   // next stage to fetch
        src.wbase += 128;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int par = 0;   // ring slot of the current tile's stage 0
    bool after_full_tile = false;   // the previous tile of this workgroup issued every one of its result stores
    f32x4 acc[4][8];
    bf16x8 fa[2][4], fb[2][8];
    const bool tracing = (g.debug_mode == 5 || g.debug_mode == 6) && threadIdx.x == 0;
    if (tracing && blockIdx.x < TRACE_WGS)
        g_pipe64_trace[16 * blockIdx.x + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    while (true) {
        if (tracing && vb < TRACE_WGS) {
            g_pipe64_trace[16 * vb] = __builtin_amdgcn_s_memrealtime();
            g_pipe64_trace[16 * vb + 3] = g_pipe64_trace[16 * blockIdx.x + 3];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const unsigned char *first = lds + par * P64_SLOT;
#pragma unroll
            for (int i = 0; i < 4; i++) fa[0][i] = *(const bf16x8 *)(first + a0 + i * 2048);
#pragma unroll
            for (int j = 0; j < 8; j++) fb[0][j] = *(const bf16x8 *)(first + b0 + j * 2048);
        }
        int s = 0;
        for (; s < ns - 2; s++) {
            unsigned char *cur = lds + ((par + s) & 1) * P64_SLOT, *nxt = lds + ((par + s + 1) & 1) * P64_SLOT;
            pipe64p_substep<false, false, true, 0>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, src, wave);
            pipe64p_substep<true, true, true, 0, epilogue_rows8_stores<EPI>()>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0,
                                                                              cur, src, wave, s == 0 && after_full_tile);
            src.abase += 128;
            src.wbase += 128;
            if (tracing && s < 4 && vb < TRACE_WGS) g_pipe64_trace[16 * vb + 9 + s] = __builtin_amdgcn_s_memrealtime();
        }
        // Stages ns - 2 and ns - 1 fetch stages 0 and 1 of this workgroup's next tile.  After the last tile they fetch its
        // own first stages once more instead of branching around the fetch: with two versions of these sub-steps the
        // compiler no longer accumulates in place and spills accumulators.
        const int vb_next = vb + gridDim.x;
        const bool more = vb_next < ntiles;
        int m1, n1;
        tile_origin(more ? vb_next : vb, m1, n1);
        unsigned char *cur = lds + ((par + s) & 1) * P64_SLOT, *nxt = lds + ((par + s + 1) & 1) * P64_SLOT;
        pipe64p_substep<false, false, true, 0>(acc, fa[0], fb[0], fa[1], fb[1], cur + a1, cur + b1, cur, src, wave);
        set_sources(m1, n1);
        pipe64p_substep<true, true, true, 0>(acc, fa[1], fb[1], fa[0], fb[0], nxt + a0, nxt + b0, cur, src, wave);
        src.abase += 128;
        src.wbase += 128;
        pipe64p_substep<false, false, true, 0>(acc, fa[0], fb[0], fa[1], fb[1], nxt + a1, nxt + b1, cur, src, wave);
        pipe64p_substep<true, true, false, 0>(acc, fa[1], fb[1], fa[0], fb[0], nxt, nxt, nxt, src, wave);
        src.abase += 128;
        src.wbase += 128;
        if (tracing && vb < TRACE_WGS) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            g_pipe64_trace[16 * vb + 1] = t;
            g_pipe64_trace[16 * vb + 4] = t;
        }
        {
            // lane index through an opaque move: everything the epilogue derives from it is recomputed per tile; hoisted
            // out of the tile loop it would stay live across the main loop, which has no register to spare
            int el = lane;
            asm volatile("" : "+v"(el));
            if (g.debug_mode != 6)
                epilogue_rows8<EPI>(g, acc, (float *)(lds + 2 * P64_SLOT + wave * P64P_STAGE), m0 + wm * 64, n0 + wn * 128, el);
            else {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("" ::"v"(acc[i][j]));
            }
        }
        if constexpr (EPI == EPI_MULGRAD) {
            if (g.colsum) {   // uniform over the grid
                // one atomic per column of the tile, a wave's 64 on neighbouring addresses
                __syncthreads();
                if (tid < 256 && n0 + tid < g.N) {
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        t += ((const float *)(lds + 2 * P64_SLOT + (r * 2 + (tid >> 7)) * P64P_STAGE))[tid & 127];
                    unsafeAtomicAdd(g.colsum + n0 + tid, t);
                }
                __syncthreads();
            }
        }
        if (tracing && vb < TRACE_WGS) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            for (int q = 5; q < 9; q++) g_pipe64_trace[16 * vb + q] = t;
            g_pipe64_trace[16 * vb + 2] = t;
        }
        if (!more) break;   // (the wave's end waits for the fetches still in flight)
        after_full_tile = m0 + 256 <= g.M && n0 + 256 <= g.N && (g.debug_mode == 0 || g.debug_mode == 5) && !(EPI == EPI_MULGRAD && g.colsum);
        vb = vb_next;
        m0 = m1;
        n0 = n1;
        par = (par + ns) & 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// The persistent kernel with the A operand fetched TWO stages ahead.
//
// Measured on the tower shapes (262 144 rows): when every tile reads the first tile's A rows (all L2 hits) the launches
// take 11-19 % less time, with the first tile's W rows nothing changes: the weights are hot in every XCD's L2, the
// activation stream comes from HBM, and a fetch that is issued one stage (about 1.5 us) before its barrier has the
// slowest of its 512 cache lines still on the way when the workgroup gets there.  The 160 KiB of LDS hold five operand
// stages of 32 KiB: two for W, as before, and THREE for A.  At the barrier in the middle of stage s a wave queues the W
// fetch of stage s + 2 and the A fetch of stage s + 3 (W first: the next barrier's counted wait leaves the four A
// fetches outstanding), so an A stage has two and a half stages to land.
// The epilogue has no room of its own: it is staged in the A slot that the tile's last stage has just left, each wave
// inside its own four DMA pieces of that slot (epilogue_rows8<.., SPLIT>), and the fetch of the next tile's stage 2, which
// goes into the same slot, is queued by each wave when its own epilogue is through (stages 0 and 1 of the next tile
// were fetched under the last stages, as in gemm_pipe64p_kernel).  Needs K >= 192.
// ---------------------------------------------------------------------------
constexpr int Q_OP = 256 * 128;        // one operand stage
constexpr int Q_W = 3 * Q_OP;          // the W ring lies behind the three A slots
constexpr int Q_LDS = 5 * Q_OP;        // 160 KiB

// FW / FA: queue this wave's four W / four A fetches behind the barrier; waits as described above.
template <bool SYNC, bool FW, bool FA, bool READ, int STORES = 0>
__device__ __forceinline__ void pipe64q_substep(f32x4 (&acc)[4][8], const bf16x8 (&ca)[4], const bf16x8 (&cb)[8],
                                                bf16x8 (&na)[4], bf16x8 (&nb)[8], const unsigned char *rd_a,
                                                const unsigned char *rd_b, unsigned char *wr_w, unsigned char *wr_a,
                                                const P64Src &src, int wave, bool newest_are_stores = false) {
    if (SYNC) {
        if (STORES > 0 && newest_are_stores)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + STORES) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (FW || FA) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool isa = q >= 4;
            if (isa && FA)
                p64_dma16<A_STREAMS_NT>(src.abase, src.voff_a, (unsigned)(q & 3) * src.stride_a, src.max_a,
                                        (unsigned)(uintptr_t)LDS_PTR(wr_a + (wave + 8 * (q & 3)) * 1024));
            if (!isa && FW)
                p64_dma16(src.wbase, src.voff_b, (unsigned)(q & 3) * src.stride_b, src.max_b,
                          (unsigned)(uintptr_t)LDS_PTR(wr_w + (wave + 8 * (q & 3)) * 1024));
            if (READ) {   // twelve fragment reads over the eight groups: 2 1 2 1 2 1 2 1
                const int r0 = (3 * q + 1) / 2, r1 = (3 * (q + 1) + 1) / 2;
#pragma unroll
                for (int r = r0; r < r1; r++) {
                    if (r < 4) na[r] = *(const bf16x8 *)(rd_a + r * 2048);
                    else nb[r - 4] = *(const bf16x8 *)(rd_b + (r - 4) * 2048);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int i = q >> 1, j = 4 * (q & 1) + t;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if (READ) {
#pragma unroll
        for (int i = 0; i < 4; i++) na[i] = *(const bf16x8 *)(rd_a + i * 2048);
#pragma unroll
        for (int j = 0; j < 8; j++) nb[j] = *(const bf16x8 *)(rd_b + j * 2048);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
    if (READ) {
#pragma unroll
        for (int q = 0; q < 12; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pipe64q_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 128 columns
    const int frow = lane & 15, fq = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    int vb = blockIdx.x;
    int m0, n0;
    auto tile_origin = [&](int b, int &mo, int &no) {
        const int swz = xcd_remap(b, ntiles);
        int tm, tn;
        if (g.col_group) {
            const int per = g.tiles_m * g.col_group;
            const int cg = swz / per, r = swz - cg * per;
            const int width = min(g.col_group, g.tiles_n - cg * g.col_group);
            tm = r / width;
            tn = cg * g.col_group + (r - tm * width);
        } else {
            tm = swz / g.tiles_n;
            tn = swz - tm * g.tiles_n;
        }
        mo = tm * 256;
        no = tn * 256;
    };
    P64Src src;
    {
        const int row = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        src.voff_a = (unsigned)(row * g.lda + c * 8) * 2u;
        src.voff_b = (unsigned)(row * g.ldw + c * 8) * 2u;
        src.chunk = (unsigned)c * 16u;
    }
    src.stride_a = 128u * (unsigned)g.lda;
    src.stride_b = 128u * (unsigned)g.ldw;
    // the lane's 16-byte chunk inside a row, recomputed where a tile's sources are set (once per tile) from the lane index
    // passed through an opaque move: kept in a register across the stage loop it would be spilled, and its reload, a vector
    // memory load, would wait for the fetches queued before it
    auto lane_chunk = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int row = wave * 8 + (l >> 3);
        return (unsigned)(((l & 7) ^ ((row >> 1) & 7)) * 16);
    };
    auto set_a = [&](int mo) {
        src.abase = (const unsigned char *)(g.A + (size_t)mo * g.lda);
        src.max_a = (unsigned)(min(g.M - 1 - mo, 255) * g.lda) * 2u + lane_chunk();
    };
    auto set_w = [&](int no) {
        src.wbase = (const unsigned char *)(g.W + (size_t)no * g.ldw);
        src.max_b = (unsigned)(min(g.N - 1 - no, 255) * g.ldw) * 2u + lane_chunk();
    };
    // fetch of one operand stage outside a sub-step (prologue, a tile's stage 2)
    auto fetch_a = [&](unsigned char *slot) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            p64_dma16<A_STREAMS_NT>(src.abase, src.voff_a, (unsigned)q * src.stride_a, src.max_a, (unsigned)(uintptr_t)LDS_PTR(slot + (wave + 8 * q) * 1024));
        src.abase += 128;
    };
    auto fetch_w = [&](unsigned char *slot) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            p64_dma16(src.wbase, src.voff_b, (unsigned)q * src.stride_b, src.max_b, (unsigned)(uintptr_t)LDS_PTR(slot + (wave + 8 * q) * 1024));
        src.wbase += 128;
    };
    tile_origin(vb, m0, n0);
    set_a(m0);
    set_w(n0);
    const int fa0 = lds_off(wm * 64 + frow, fq), fa1 = lds_off(wm * 64 + frow, 4 + fq);      // A fragments inside a slot: k 0..31, 32..63
    const int fb0 = lds_off(wn * 128 + frow, fq), fb1 = lds_off(wn * 128 + frow, 4 + fq);
    const int ns = g.K / 64;   // >= 3
    fetch_w(lds + Q_W);
    fetch_a(lds);
    fetch_w(lds + Q_W + Q_OP);
    fetch_a(lds + Q_OP);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int ap = 0, wp = 0;             // ring slots of the current tile's stage 0
    bool after_full_tile = false;   // the previous tile of this workgroup issued every one of its result stores
    f32x4 acc[4][8];
    bf16x8 fa[2][4], fb[2][8];
    auto next3 = [](int i) { return i == 2 ? 0 : i + 1; };
    while (true) {
        // this tile's stage 2 goes into the slot the previous epilogue was staged in
        int ai = ap, wi = wp;       // slots of the stage being computed
        fetch_a(lds + next3(next3(ap)) * Q_OP);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const unsigned char *pa = lds + ai * Q_OP, *pw = lds + Q_W + wi * Q_OP;
#pragma unroll
            for (int i = 0; i < 4; i++) fa[0][i] = *(const bf16x8 *)(pa + fa0 + i * 2048);
#pragma unroll
            for (int j = 0; j < 8; j++) fb[0][j] = *(const bf16x8 *)(pw + fb0 + j * 2048);
        }
        // Stages ns - 3 and ns - 2 switch the A and then the W source to the workgroup's next tile (after its last tile, to
        // that tile's own first stages once more): scalar work inside ONE version of the stage body; with a second
        // version the compiler no longer accumulates in place and runs out of registers.
        const int vb_next = vb + gridDim.x;
        const bool more = vb_next < ntiles;
        int m1, n1;
        tile_origin(more ? vb_next : vb, m1, n1);
        for (int s = 0; s < ns - 1; s++) {
            unsigned char *ca = lds + ai * Q_OP, *cw = lds + Q_W + wi * Q_OP;
            const int an = next3(ai);
            unsigned char *na = lds + an * Q_OP, *nw = lds + Q_W + (wi ^ 1) * Q_OP;
            pipe64q_substep<false, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], ca + fa1, cw + fb1, cw, ca, src, wave);
            if (s == ns - 3) set_a(m1);
            if (s == ns - 2) set_w(n1);
            pipe64q_substep<true, true, true, true, epilogue_rows8_stores<EPI>()>(acc, fa[1], fb[1], fa[0], fb[0], na + fa0, nw + fb0,
                                                                                cw, ca, src, wave, s == 0 && after_full_tile);
            src.abase += 128;
            src.wbase += 128;
            ai = an;
            wi ^= 1;
        }
        {   // stage ns - 1: W of the next tile's stage 1; its A slot becomes the epilogue's staging area
            unsigned char *ca = lds + ai * Q_OP, *cw = lds + Q_W + wi * Q_OP;
            pipe64q_substep<false, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], ca + fa1, cw + fb1, cw, ca, src, wave);
            pipe64q_substep<true, true, false, false>(acc, fa[1], fb[1], fa[0], fb[0], ca, cw, cw, ca, src, wave);
            src.wbase += 128;
            {
                // lane index through an opaque move (see gemm_pipe64p_kernel)
                int el = lane;
                asm volatile("" : "+v"(el));
                if (g.debug_mode != 6)
                    epilogue_rows8<EPI, true>(g, acc, (float *)(ca + wave * 1024), m0 + wm * 64, n0 + wn * 128, el);
                else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 8; j++) asm volatile("" ::"v"(acc[i][j]));
                }
            }
            ai = next3(ai);
            wi ^= 1;
        }
        if (!more) break;
        after_full_tile = m0 + 256 <= g.M && n0 + 256 <= g.N && g.debug_mode != 6;
        vb = vb_next;
        m0 = m1;
        n0 = n1;
        ap = ai;
        wp = wi;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// C_f32[M, N] += A^T B for A bf16 [Kd, M] and B bf16 [Kd, N], both ROW major (the reduction index is
// the row): the weight gradient dW = dY^T X straight from the activations as they lie in memory, no
// transposed copies.  256 x 256 tile, 64 reduction rows per stage, the pipeline of gemm_pipe64_kernel.
// An operand stage is [64 rows][256 columns] (512 B per row) in LDS; the MFMA fragments (8 reduction
// steps of one column per lane) are read TRANSPOSED with ds_read_b64_tr_b16: the 16 lanes of a
// fragment quarter fetch a 4 row x 16 column block, 8 bytes each, and every lane receives one column
// of it.  16-byte chunk c of row r sits at chunk c ^ swz(r), swz(r) = 2 * ((r & 3) | ((r >> 3) & 1) << 2):
// the 8 rows x 2 chunks a 32-lane read group touches then fall on 16 different bank groups.
// The reduction is split over blockIdx.y; every workgroup adds its tile with fp32 atomics.
// Rows past Kd are fetched from a zero block, columns past M / N are clamped (computed, not stored).
// ---------------------------------------------------------------------------
static __device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int tn_swz(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }

// The whole-stage kernel's DMA source of piece q = a uniform pointer (src[q]: the piece's first row, in SGPRs) + a 32-bit
// per-lane offset that is the same for every piece of an operand: two VGPRs and no vector arithmetic.  (Per-lane 64-bit
// pointers and their row-limit selects cost about 30 VGPRs around the fetch; that kernel spills, and the vmcnt(0) wait of
// every reload in the loop also waits for the DMA just issued, so the fetch latency is no longer hidden.)
struct TnSrc {
    const unsigned char *src[8];   // q < 4: A pieces wave + 8 q, q >= 4: B pieces
    unsigned voff_a, voff_b;
};

// 16 bytes per lane from base + voff to LDS at lds_addr + 16 * lane.  Assembly for the scalar-base form of the
// instruction; M0 (the LDS address of the transfer) is used by nothing else in that kernel.
__device__ __forceinline__ void tn_dma16(const unsigned char *base, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr));
}

// FAST: every stage lies inside the reduction range (t is used; goff .. row_lim are not).  Otherwise rows at or past
// row_lim read zeros (t is not used).
template <bool FAST, bool SYNC, bool DMA, bool READ>
__device__ __forceinline__ void tn_substep(f32x4 (&acc)[4][8], const bf16x8 (&ca)[4], const bf16x8 (&cb)[8],
                                           bf16x8 (&na)[4], bf16x8 (&nb)[8], const unsigned char *rd_row, int a_x, int b_x,
                                           unsigned char *wr_slot, const unsigned (&goff)[8], const unsigned char *abase,
                                           const unsigned char *bbase, const unsigned char *zero, unsigned row_lim, int wave,
                                           const TnSrc &t) {
    if (SYNC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (DMA && FAST) {
#pragma unroll
        for (int q = 0; q < 8; q++)
            tn_dma16(t.src[q], q < 4 ? t.voff_a : t.voff_b, (unsigned)(uintptr_t)LDS_PTR(wr_slot + (wave + 8 * q) * 1024));
    }
    if (DMA && !FAST) {
        // rows at or past row_lim (the reduction rows left in this workgroup's range) read zeros
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned row = 2u * (unsigned)(wave + 8 * (q & 3)) + ((unsigned)threadIdx.x >> 5 & 1u);
            const unsigned char *src = (q < 4 ? abase : bbase) + goff[q];
            src = row < row_lim ? src : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(wr_slot + (wave + 8 * q) * 1024), 16, 0, 0);
        }
    }
    if (READ) {
        // rd_row: this lane's row 8 fq + tq of the sub-step (+ its 8-byte half); 4 rows further = + 2048 bytes, same swizzle.
        // Fragment i / j starts 16 columns = two 16-byte chunks further: chunk bits 1.. of the swizzled offset flip.
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned char *p = rd_row + (a_x ^ (i << 5));
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p + 2048));
            na[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned char *p = rd_row + 32768 + (b_x ^ (j << 5));
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p + 2048));
            nb[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
    if (DMA && !FAST) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);
        }
    }
    if (READ) {
#pragma unroll
        for (int q = 0; q < 24; q++) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
        }
    }
}

// FAST: the host promises that every workgroup's reduction range is a multiple of 64 rows, at least 128
template <bool FAST>
__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 rows x 128 columns of C
    // Workgroups are dealt to the 8 XCDs round robin and every XCD has its own L2.  A unit = a group of tn_gm x tn_gn
    // tiles on one reduction slice: its workgroups walk the same rows of tn_gm + tn_gn operand panels side by side, so
    // with a unit's workgroups on ONE XCD (tn_upx units each) its L2 fetches every panel row once.
    int tm, tn, split;
    if (g.tn_gm) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int gsz = g.tn_gm * g.tn_gn;
        const int ul = slot / gsz, tig = slot - ul * gsz;
        const int unit = xcd * g.tn_upx + ul;
        const int groups_n = g.tiles_n / g.tn_gn, groups = (g.tiles_m / g.tn_gm) * groups_n;
        split = unit / groups;
        const int grp = unit - split * groups;
        const int gmi = grp / groups_n, gni = grp - gmi * groups_n;
        const int tim = tig / g.tn_gn;
        tm = gmi * g.tn_gm + tim;
        tn = gni * g.tn_gn + (tig - tim * g.tn_gn);
    } else {
        tm = blockIdx.x / g.tiles_n;
        tn = blockIdx.x - tm * g.tiles_n;
        split = blockIdx.y;
    }
    const int m0 = tm * 256, n0 = tn * 256;
    // reduction range of this workgroup: g.K rows from split * g.K (a multiple of 64), cut at k_total
    const int k_begin = split * g.K;
    const int k_len = min(g.K, g.k_total - k_begin);
    const int ns = max(2, (k_len + 63) / 64);   // a second stage of zeros when the range is that short

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][8];

    // DMA piece p (32 per operand and stage) = stage rows 2p, 2p + 1; lane -> row 2p + (lane >> 5), LDS chunk position
    // lane & 31, which holds source chunk (lane & 31) ^ swz(row).  q < 4: A pieces wave + 8 q; q >= 4: B pieces.
    unsigned goff[8];
    TnSrc t;
    if (FAST) {
        // swz(row) only looks at row bits 0, 1 and 3, which the 16 q rows between a wave's pieces leave alone: the
        // per-lane part of the offset is that of piece q = 0
        const int c = (lane & 31) ^ tn_swz(2 * wave + (lane >> 5));
        const int ca = m0 + c * 8, cb = n0 + c * 8;          // past the matrix: any valid chunk (never stored)
        t.voff_a = ((unsigned)(lane >> 5) * (unsigned)g.lda + (unsigned)(ca < g.M ? ca : g.M - 8)) * 2u;
        t.voff_b = ((unsigned)(lane >> 5) * (unsigned)g.ldw + (unsigned)(cb < g.N ? cb : g.N - 8)) * 2u;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const size_t row = (size_t)k_begin + 2 * wave + 16 * (q & 3);
            t.src[q] = (const unsigned char *)(q < 4 ? g.A + row * g.lda : g.W + row * g.ldw);
        }
    }
#pragma unroll
    for (int q = 0; q < 8 && !FAST; q++) {
        const bool isb = q >= 4;
        const int row = 2 * (wave + 8 * (q & 3)) + (lane >> 5);
        const int c = (lane & 31) ^ tn_swz(row);
        const int col0 = (isb ? n0 : m0) + c * 8;
        const int lim = isb ? g.N : g.M;                     // multiples of 8
        const int col = col0 < lim ? col0 : lim - 8;         // past the matrix: any valid chunk (never stored)
        goff[q] = ((unsigned)row * (unsigned)(isb ? g.ldw : g.lda) + (unsigned)col) * 2u;
    }
    const unsigned char *abase = (const unsigned char *)(g.A + (size_t)k_begin * g.lda);
    const unsigned char *bbase = (const unsigned char *)(g.W + (size_t)k_begin * g.ldw);
    const unsigned char *zero = (const unsigned char *)g_zero16;
    const unsigned astep = 64u * (unsigned)g.lda * 2u, bstep = 64u * (unsigned)g.ldw * 2u;

    // fragment reads: lane = (fq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3) fetches row 8 fq + tq, columns 4 tp .. + 3
    const int fq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int frow = 8 * fq + tq;
    const int row_off = frow * 512 + (tp & 1) * 8;
    const int a_x = (((wm * 8 + (tp >> 1)) ^ tn_swz(frow)) << 4);
    const int b_x = (((wn * 16 + (tp >> 1)) ^ tn_swz(frow)) << 4);

    auto advance = [&]() {
        if (FAST) {
#pragma unroll
            for (int q = 0; q < 8; q++) t.src[q] += q < 4 ? astep : bstep;
        } else {
            abase += astep;
            bbase += bstep;
        }
    };
    // stages 0 and 1
#pragma unroll
    for (int st = 0; st < 2; st++) {
        const unsigned lim = (unsigned)max(0, min(64, k_len - 64 * st));
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (FAST) {
                tn_dma16(t.src[q], q < 4 ? t.voff_a : t.voff_b,
                         (unsigned)(uintptr_t)LDS_PTR(lds + st * P64_SLOT + (wave + 8 * q) * 1024));
            } else {
                const unsigned row = 2u * (unsigned)(wave + 8 * (q & 3)) + ((unsigned)lane >> 5);
                const unsigned char *src = (q < 4 ? abase : bbase) + goff[q];
                src = row < lim ? src : zero;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds + st * P64_SLOT + (wave + 8 * q) * 1024), 16, 0, 0);
            }
        }
        advance();
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // fragments of stage 0, sub-step 0
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned char *p = lds + row_off + (a_x ^ (i << 5));
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p + 2048));
        fa[0][i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned char *p = lds + row_off + 32768 + (b_x ^ (j << 5));
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p + 2048));
        fb[0][j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }

    // sub-step 1 of a stage reads rows 32.. of the same slot: + 32 * 512 bytes
    int s = 0;
    for (; s < ns - 2; s++) {
        unsigned char *cur = lds + (s & 1) * P64_SLOT, *nxt = lds + ((s + 1) & 1) * P64_SLOT;
        const unsigned lim = (unsigned)max(0, min(64, k_len - 64 * (s + 2)));
        tn_substep<FAST, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], cur + 16384 + row_off, a_x, b_x, cur, goff, abase, bbase,
                                       zero, lim, wave, t);
        tn_substep<FAST, true, true, true>(acc, fa[1], fb[1], fa[0], fb[0], nxt + row_off, a_x, b_x, cur, goff, abase, bbase, zero, lim,
                                     wave, t);
        advance();
    }
    {
        unsigned char *cur = lds + (s & 1) * P64_SLOT, *nxt = lds + ((s + 1) & 1) * P64_SLOT;
        tn_substep<FAST, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], cur + 16384 + row_off, a_x, b_x, cur, goff, abase, bbase,
                                       zero, 0u, wave, t);
        tn_substep<FAST, true, false, true>(acc, fa[1], fb[1], fa[0], fb[0], nxt + row_off, a_x, b_x, cur, goff, abase, bbase, zero, 0u,
                                      wave, t);
        tn_substep<FAST, false, false, true>(acc, fa[0], fb[0], fa[1], fb[1], nxt + 16384 + row_off, a_x, b_x, cur, goff, abase, bbase,
                                       zero, 0u, wave, t);
        tn_substep<FAST, false, false, false>(acc, fa[1], fb[1], fa[0], fb[0], nxt, a_x, b_x, cur, goff, abase, bbase, zero, 0u, wave, t);
    }
    __syncthreads();
    epilogue_passes<SGCLIP_EPI_F32_ATOMIC, 4, 8>(g, acc, (float *)(lds + wave * epi_pass_bytes(8)), m0 + wm * 64, n0 + wn * 128,
                                                  lane);
}

// dynamic LDS limit of `fn` on the current device (see lds_attr); a refusal surfaces through launch_check as the
// launch failure that follows it
static void lds_attr_once(const void *fn, int bytes) { (void)lds_attr(fn, bytes); }

static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SGCLIP_GEMM_VARIANT");
        v = e ? atoi(e) : 7;
    }
    return v;
}

static int pipe64_min_tiles() {   // 256 x 256 tiles from this many on (smaller grids: 256 x 128 tiles, two workgroups per CU)
    static int min_tiles = -1;
    if (min_tiles < 0) {
        const char *e = getenv("SGCLIP_PIPE64_MIN_TILES");
        min_tiles = e ? atoi(e) : 256;
    }
    return min_tiles;
}

// whether launch() runs the shape on gemm_pipe64_kernel (k_shortest: the shortest K slice of a split launch)
static bool runs_on_pipe64(int M, int N, int K, int k_shortest, int splits) {
    const long big_tiles = (long)ceil_div(M, 256) * ceil_div(N, 256) * splits;
    return gemm_variant() >= 7 && big_tiles >= pipe64_min_tiles() && M > 128 && K % 64 == 0 && k_shortest >= 128 && N >= 256;
}

// Tile columns per group for gemm_pipe64_kernel's tile order (0: row major), from the bytes the 8 L2s would fetch:
// row major reads A once and, when W does not fit an L2 beside the streams, W's panels once per round of 256 tiles;
// grouped reads A once per group and each group's panels (which fit) once per XCD.
static int pipe64_col_group(int tiles_m, int tiles_n, int K) {
    static int env = -2;
    if (env == -2) {
        const char *e = getenv("SGCLIP_COL_GROUP");
        env = e ? atoi(e) : -1;
    }
    if (env >= 0) return env >= tiles_n ? 0 : env;
    const double panel = 256.0 * K * 2, l2_room = 2.5e6;
    const double a_bytes = panel * tiles_m, rounds = (double)tiles_m * tiles_n / 256.0;
    if (panel * tiles_n <= l2_room || rounds < 2) return 0;
    const double row_major = a_bytes + rounds * 8 * std::min(tiles_n, 32) * panel;
    const int gn = (int)(l2_room / panel);
    if (gn < 1) return 0;
    const double grouped = a_bytes * ceil_div(tiles_n, gn) + 8.0 * panel * tiles_n;
    return grouped < 0.8 * row_major ? gn : 0;
}

static thread_local float *g_mulgrad_colsum = nullptr;   // set around its call by sgclip_gemm_mulgelugrad_bf16

template <int EPI>
static void launch(const GemmArgs &g0, hipStream_t stream, int splits = 1) {
    GemmArgs g = g0;
    const unsigned gy = (unsigned)splits;
    // measured on MI355X (M = 65536, random data): the 32-deep two-workgroups-per-CU kernel wins
    // for short K (epilogue-heavy), the 64-deep ring for long K
    const int variant = gemm_variant();
    const bool use32 = variant == 2 || (variant >= 3 && g.K <= 1024);
    const long big_tiles = (long)ceil_div(g.M, 256) * ceil_div(g.N, 256) * splits;
    // the last K slice of a split launch is the shortest (split_k_setup): the 64-deep pipeline needs two stages of it
    const int k_shortest = g.k_total - (splits - 1) * g.K;
    if (runs_on_pipe64(g.M, g.N, g.K, k_shortest, splits)) {
        g.tiles_m = ceil_div(g.M, 256);
        g.tiles_n = ceil_div(g.N, 256);
        g.col_group = splits == 1 ? pipe64_col_group(g.tiles_m, g.tiles_n, g.K) : 0;
        if constexpr (EPI != SGCLIP_EPI_F32_ATOMIC) {
            static int persist = -1;
            if (persist < 0) {
                const char *e = getenv("SGCLIP_GEMM_PERSISTENT");
                persist = e ? atoi(e) : 1;
            }
            const int cus = device_cus();
            // Mode 2: one workgroup per tile for deep reductions with many tile columns (the ViT-L q/k/v, fc1 and fc2
            // shapes), where workgroups that run in step beat the two-slot persistent kernel's drifted ones by 2-3 %.
            // With the A operand fetched two stages ahead (gemm_pipe64q_kernel) the persistent form wins there too
            // (ViT-L fc1 2 050 -> 1 910 us, q/k/v 1 650 -> 1 605 us per 262 144 rows), so mode 1 is the default.
            const bool deep_wide = g.K >= 1024 && g.tiles_n >= 8;
            if (persist && !(persist == 2 && deep_wide) && splits == 1 && g.wide_epilogue &&
                (g.debug_mode == 0 || g.debug_mode == 5 || g.debug_mode == 6)) {
                const long tiles = (long)g.tiles_m * g.tiles_n;
                static int deep_a = -1;
                if (deep_a < 0) {
                    const char *e = getenv("SGCLIP_GEMM_DEEP_A");
                    deep_a = e ? atoi(e) : 1;
                }
                if (deep_a && g.K >= 192 && EPI != EPI_MULGRAD && g.debug_mode != 5) {
                    lds_attr_once((const void *)gemm_pipe64q_kernel<EPI>, Q_LDS);
                    hipLaunchKernelGGL(gemm_pipe64q_kernel<EPI>, dim3((unsigned)std::min<long>(tiles, cus)), dim3(512), Q_LDS,
                                       stream, g);
                    return;
                }
                lds_attr_once((const void *)gemm_pipe64p_kernel<EPI>, P64P_LDS);
                hipLaunchKernelGGL(gemm_pipe64p_kernel<EPI>, dim3((unsigned)std::min<long>(tiles, cus)), dim3(512), P64P_LDS,
                                   stream, g);
                return;
            }
        }
        auto go = [&](auto kernel) {
            lds_attr_once((const void *)kernel, 2 * P64_SLOT);
            hipLaunchKernelGGL(kernel, dim3((unsigned)g.tiles_m * g.tiles_n, gy), dim3(512), 2 * P64_SLOT, stream, g);
        };
        if (EPI == SGCLIP_EPI_BIAS_BF16 && g.debug_mode == 1) go(gemm_pipe64_kernel<SGCLIP_EPI_BIAS_BF16, 1>);
        else if (EPI == SGCLIP_EPI_BIAS_BF16 && g.debug_mode == 3) go(gemm_pipe64_kernel<SGCLIP_EPI_BIAS_BF16, 3>);
        else if (EPI == SGCLIP_EPI_BIAS_BF16 && g.debug_mode == 4) go(gemm_pipe64_kernel<SGCLIP_EPI_BIAS_BF16, 4>);
        else go(gemm_pipe64_kernel<EPI, 0>);
    } else if ((variant == 4 || (variant >= 5 && big_tiles >= 512)) && g.M > 128 && g.K % 32 == 0 && g.N >= 256) {
        g.tiles_m = ceil_div(g.M, 256);
        g.tiles_n = ceil_div(g.N, 256);
        lds_attr_once((const void *)gemm_big_kernel<EPI>, BIG_SLOTS * BIG_SLOT);
        hipLaunchKernelGGL(gemm_big_kernel<EPI>, dim3((unsigned)g.tiles_m * g.tiles_n, gy), dim3(512), BIG_SLOTS * BIG_SLOT,
                           stream, g);
    } else if (use32 && g.M > 128 && g.K % 32 == 0) {
        g.tiles_m = ceil_div(g.M, RING_BM);
        lds_attr_once((const void *)gemm_ring32_kernel<EPI>, R32_SLOTS * R32_SLOT);
        hipLaunchKernelGGL(gemm_ring32_kernel<EPI>, dim3((unsigned)g.tiles_m * g.tiles_n, gy), dim3(512), R32_SLOTS * R32_SLOT,
                           stream, g);
    } else if (gemm_variant() >= 1 && g.M > 128) {
        g.tiles_m = ceil_div(g.M, RING_BM);
        lds_attr_once((const void *)gemm_ring_kernel<EPI>, RING_SLOTS * RING_SLOT);
        hipLaunchKernelGGL(gemm_ring_kernel<EPI>, dim3((unsigned)g.tiles_m * g.tiles_n, gy), dim3(512), RING_SLOTS * RING_SLOT,
                           stream, g);
    } else {
        g.tiles_m = ceil_div(g.M, 128);
        hipLaunchKernelGGL(gemm_kernel<EPI>, dim3((unsigned)g.tiles_m * g.tiles_n, gy), dim3(256), 0, stream, g);
    }
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_gemm_bf16(const void *A_dev, int lda, const void *W_dev, int ldw, void *C_dev, int ldc, int M,
                                int N, int K, int epilogue, const float *bias_dev, const void *aux_void, int aux_rows,
                                void *stream_) {
    const float *aux_dev = (const float *)aux_void;
    hipStream_t stream = (hipStream_t)stream_;
    if (M == 0 || N == 0) return SGCLIP_OK;
    if (!A_dev || !W_dev || !C_dev || M < 0 || N < 0 || K <= 0 || K % BK != 0 || lda < K || ldw < K || ldc < N) {
        set_error("sgclip_gemm_bf16: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d (K must be a multiple of %d)", M, N,
                  K, lda, ldw, ldc, BK);
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)W_dev & 15) || (lda & 7) || (ldw & 7)) {
        set_error("sgclip_gemm_bf16: A/W must be 16-byte aligned with leading dimensions a multiple of 8");
        return SGCLIP_EINVAL;
    }
    const bool needs_bias = epilogue == SGCLIP_EPI_BIAS_BF16 || epilogue == SGCLIP_EPI_BIAS_GELU_BF16 ||
                            epilogue == SGCLIP_EPI_BIAS_RESID_F32 || epilogue == EPI_GELU_BOTH || epilogue == EPI_MULGRAD;
    const bool needs_aux = epilogue == EPI_GELU_BOTH || epilogue == EPI_MULGRAD;
    if ((needs_bias && !bias_dev) || (epilogue == SGCLIP_EPI_PATCH_F32 && (!aux_dev || aux_rows <= 0)) || (needs_aux && !aux_dev)) {
        set_error("sgclip_gemm_bf16: epilogue %d needs bias/aux", epilogue);
        return SGCLIP_EINVAL;
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)A_dev;
    g.W = (const bf16_t *)W_dev;
    g.C = C_dev;
    g.bias = bias_dev;
    g.aux = aux_dev;
    g.lda = lda;
    g.ldw = ldw;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    g.K = K;
    g.k_total = K;
    g.aux_rows = aux_rows;
    g.colsum = epilogue == EPI_MULGRAD ? g_mulgrad_colsum : nullptr;
    g.tiles_m = 0;
    g.tiles_n = ceil_div(N, BN);
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char *e = getenv("SGCLIP_GEMM_DEBUG");
            dbg = e ? atoi(e) : 0;
            if (dbg)
                fprintf(stderr, "sgclip: SGCLIP_GEMM_DEBUG=%d is a profiling mode: sgclip_gemm_bf16 skips part of its work and "
                                "its RESULTS ARE WRONG\n", dbg);
        }
        g.debug_mode = dbg;
    }
    g.wide_epilogue = (N % 8 == 0) && (ldc % 8 == 0) && !((uintptr_t)C_dev & 15) && !((uintptr_t)bias_dev & 15) &&
                      !((uintptr_t)aux_dev & 15);
    KernelTimer timer(TK_GEMM0 + (epilogue == EPI_GELU_BOTH ? SGCLIP_EPI_BIAS_GELU_BF16 : epilogue == EPI_MULGRAD ? SGCLIP_EPI_BIAS_BF16 : epilogue),
                      2.0 * M * N * K, stream);
    switch (epilogue) {
        case SGCLIP_EPI_BIAS_BF16: launch<SGCLIP_EPI_BIAS_BF16>(g, stream); break;
        case SGCLIP_EPI_BIAS_GELU_BF16: launch<SGCLIP_EPI_BIAS_GELU_BF16>(g, stream); break;
        case SGCLIP_EPI_BIAS_RESID_F32: launch<SGCLIP_EPI_BIAS_RESID_F32>(g, stream); break;
        case EPI_GELU_BOTH: launch<EPI_GELU_BOTH>(g, stream); break;
        case EPI_MULGRAD: launch<EPI_MULGRAD>(g, stream); break;
        case SGCLIP_EPI_PATCH_F32: launch<SGCLIP_EPI_PATCH_F32>(g, stream); break;
        case SGCLIP_EPI_F32: launch<SGCLIP_EPI_F32>(g, stream); break;
        case SGCLIP_EPI_F32_ATOMIC: {
            // K slices: about `target` workgroups in flight, each at least 512 deep.  More slices fill more CUs but every
            // slice adds a full tile of fp32 atomics; 2-3 slices of the towers' weight gradients measured fastest
            static int target = -1;
            if (target < 0) {
                const char *e = getenv("SGCLIP_SPLITK_WGS");
                target = e ? atoi(e) : 96;
            }
            const long tiles = (long)ceil_div(M, 256) * ceil_div(N, N >= 256 ? 256 : BN);
            int splits = (int)std::max(1L, std::min((long)ceil_div(K, 512), target / tiles));
            const int kc = ceil_div(ceil_div(K, splits), BK) * BK;
            splits = ceil_div(K, kc);
            g.K = kc;
            launch<SGCLIP_EPI_F32_ATOMIC>(g, stream, splits);
            break;
        }
        default: set_error("sgclip_gemm_bf16: unknown epilogue %d", epilogue); return SGCLIP_EINVAL;
    }
    return launch_check("sgclip_gemm_bf16");
}

// diagnostics (not part of the declared interface): the last traced launch's per-workgroup records, 16 x u64 each
extern "C" int sgclip_debug_gemm_trace(unsigned long long *host_out, int wgs) {
    if (wgs > TRACE_WGS) wgs = TRACE_WGS;
    if (hipDeviceSynchronize() != hipSuccess) return SGCLIP_ELAUNCH;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pipe64_trace), (size_t)wgs * 128) != hipSuccess) return SGCLIP_ELAUNCH;
    return wgs;
}

extern "C" int sgclip_gemm_mulgelugrad_bf16(const void *A_dev, int lda, const void *W_dev, int ldw, void *C_dev, int ldc, int M,
                                            int N, int K, const float *bias_dev, const void *u_dev, float *colsum_dev,
                                            void *stream) {
    const bool wide = (N % 8 == 0) && (ldc % 8 == 0) && !((uintptr_t)C_dev & 15) && !((uintptr_t)bias_dev & 15) &&
                      !((uintptr_t)u_dev & 15);
    if (colsum_dev && M > 0 && N > 0 && !(wide && runs_on_pipe64(M, N, K, K, 1))) {
        // shapes that the 256 x 256 kernel does not take: the product, then a pass over C
        const int rc = sgclip_gemm_bf16(A_dev, lda, W_dev, ldw, C_dev, ldc, M, N, K, EPI_MULGRAD, bias_dev, u_dev, 0, stream);
        return rc != SGCLIP_OK ? rc : sgclip_colsum(C_dev, SGCLIP_DTYPE_BF16, ldc, M, N, colsum_dev, stream);
    }
    g_mulgrad_colsum = colsum_dev;
    const int rc = sgclip_gemm_bf16(A_dev, lda, W_dev, ldw, C_dev, ldc, M, N, K, EPI_MULGRAD, bias_dev, u_dev, 0, stream);
    g_mulgrad_colsum = nullptr;
    return rc;
}

namespace sgclip {

// Launch plan of the whole-stage weight-gradient kernel.  The plain grid (tile fastest, slices on blockIdx.y) fills the
// CUs best but scatters the tiles that share an operand panel over the XCDs: every XCD then pulls most panels through
// its own L2 and the launch runs at the rate of the fabric behind the L2s.  The grouped grid keeps gm x gn tiles of one
// slice on one XCD (at most 32 workgroups there, one per CU).  Both are priced with the rates measured on MI355X
// (1.24 PFLOP/s per fully used chip, 5.9 TB/s into the L2s) and the cheaper one is launched.
struct TnPlan {
    int gm = 0, gn = 0, upx = 0, splits = 1, kc = 0;
    double cost = 1e30;
};

static int tn_slice_rows(int kd, int &splits) {   // slices of whole stages, the last at least two stages
    int kc = ceil_div(ceil_div(kd, splits), 64) * 64;
    while (kd % kc == 64) kc += 64;
    splits = ceil_div(kd, kc);
    return kc;
}

static TnPlan tn_plan(int tiles_m, int tiles_n, int kd, int target_wgs) {
    const double flops = 2.0 * 256 * 256 * (double)kd * tiles_m * tiles_n, panel_bytes_per_row = 512.0;
    auto price = [&](double util, double panels_fetched_rows) {
        const double fetch = panels_fetched_rows * panel_bytes_per_row / 5.9e12;
        return std::max(flops / (1.24e15 * util), fetch) + 0.1 * fetch;   // of two compute-bound plans, the one that fetches less
    };
    TnPlan best;
    {   // plain grid: per XCD and slice, the distinct panels of the tiles t with (t + tiles * slice) % 8 == xcd
        const int tiles = tiles_m * tiles_n;
        int splits = (int)std::max(1L, std::min((long)ceil_div(kd, 512), (long)target_wgs / tiles));
        const int kc = tn_slice_rows(kd, splits);
        double panel_rows = 0;
        for (int sl = 0; sl < splits; sl++) {
            const int rows = std::min(kc, kd - sl * kc);
            for (int x = 0; x < 8; x++) {
                unsigned long long ms = 0, ns = 0;   // tiles_m, tiles_n <= 64 here, else no set is kept (all distinct)
                int cnt = 0;
                for (int t = 0; t < tiles; t++)
                    if ((t + tiles * sl) % 8 == x) {
                        ms |= 1ull << ((t / tiles_n) & 63);
                        ns |= 1ull << ((t % tiles_n) & 63);
                        cnt++;
                    }
                const int distinct = tiles_m <= 64 && tiles_n <= 64 ? __builtin_popcountll(ms) + __builtin_popcountll(ns) : 2 * cnt;
                panel_rows += (double)distinct * rows;
            }
        }
        const long wgs = (long)tiles * splits;
        const double util = (double)wgs / (ceil_div(wgs, 256) * 256.0);
        best.splits = splits, best.kc = kc, best.cost = price(util, panel_rows);
    }
    for (int gm = 1; gm <= tiles_m; gm++) {
        if (tiles_m % gm) continue;
        for (int gn = 1; gn <= tiles_n; gn++) {
            if (tiles_n % gn || gm * gn > 32) continue;
            const int groups = (tiles_m / gm) * (tiles_n / gn);
            for (int upx = 32 / (gm * gn); upx >= 1; upx--) {
                if ((8 * upx) % groups) continue;
                int splits = 8 * upx / groups;
                if (splits > ceil_div(kd, 512) && splits > 1) continue;
                const int want = splits;
                const int kc = tn_slice_rows(kd, splits);
                if (splits != want) continue;   // every unit needs its slice
                const double cost = price(gm * gn * upx / 32.0, (double)(gm + gn) * groups * kd);
                if (cost < best.cost * 0.97) best = TnPlan{gm, gn, upx, splits, kc, cost};   // a tie goes to the plain grid
                break;
            }
        }
    }
    return best;
}

}  // namespace sgclip

extern "C" int sgclip_gemm_tn_bf16(const void *A_dev, int lda, const void *B_dev, int ldb, float *C_dev, int ldc, int M, int N,
                                   int Kd, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M == 0 || N == 0 || Kd == 0) return SGCLIP_OK;
    if (!A_dev || !B_dev || !C_dev || M < 0 || N < 0 || Kd < 0 || (M & 7) || (N & 7) || lda < M || ldb < N || ldc < N) {
        set_error("sgclip_gemm_tn_bf16: bad shape M=%d N=%d Kd=%d lda=%d ldb=%d ldc=%d (M, N multiples of 8)", M, N, Kd, lda,
                  ldb, ldc);
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)B_dev & 15) || ((uintptr_t)C_dev & 3) || (lda & 7) || (ldb & 7) ||
        (size_t)64 * lda * 2 + (size_t)M * 2 >= (1ull << 32) || (size_t)64 * ldb * 2 + (size_t)N * 2 >= (1ull << 32)) {
        set_error("sgclip_gemm_tn_bf16: A/B must be 16-byte aligned with leading dimensions a multiple of 8");
        return SGCLIP_EINVAL;
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)A_dev;
    g.W = (const bf16_t *)B_dev;
    g.C = C_dev;
    g.lda = lda;
    g.ldw = ldb;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    g.k_total = Kd;
    g.tiles_m = ceil_div(M, 256);
    g.tiles_n = ceil_div(N, 256);
    // reduction slices: about one workgroup per CU, each at least 512 rows deep
    static int target = -1;
    if (target < 0) {
        const char *e = getenv("SGCLIP_TN_WGS");
        target = e ? atoi(e) : 256;
    }
    if (!lds_attr((const void *)gemm_tn_kernel<true>, 2 * P64_SLOT) || !lds_attr((const void *)gemm_tn_kernel<false>, 2 * P64_SLOT))
        return SGCLIP_ELAUNCH;
    const long tiles = (long)g.tiles_m * g.tiles_n;
    KernelTimer timer(TK_GEMM0 + SGCLIP_EPI_F32_ATOMIC, 2.0 * M * N * Kd, stream);
    // Whole 64-row stages go to the kernel that assumes them (every slice a multiple of 64 rows, at least two stages);
    // the last Kd % 64 rows, or a reduction shorter than two stages, to the one that checks every row.
    const int kd_main = Kd / 64 * 64 >= 128 ? Kd / 64 * 64 : 0;
    if (kd_main) {
        static int grouped = -1;
        if (grouped < 0) {
            const char *e = getenv("SGCLIP_TN_GROUPED");
            grouped = e ? atoi(e) : 1;
        }
        TnPlan plan = tn_plan(g.tiles_m, g.tiles_n, kd_main, target);
        if (!grouped && plan.gm) {
            plan = TnPlan();
            plan.splits = (int)std::max(1L, std::min((long)ceil_div(kd_main, 512), target / tiles));
            plan.kc = tn_slice_rows(kd_main, plan.splits);
        }
        g.k_total = kd_main;
        g.K = plan.kc;
        g.tn_gm = plan.gm, g.tn_gn = plan.gn, g.tn_upx = plan.upx;
        const dim3 grid = plan.gm ? dim3((unsigned)(8 * plan.upx * plan.gm * plan.gn), 1) : dim3((unsigned)tiles, (unsigned)plan.splits);
        hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(512), 2 * P64_SLOT, stream, g);
        g.tn_gm = 0;
    }
    if (Kd > kd_main) {
        g.A = (const bf16_t *)A_dev + (size_t)kd_main * lda;
        g.W = (const bf16_t *)B_dev + (size_t)kd_main * ldb;
        g.k_total = Kd - kd_main;   // at most 127 rows
        g.K = 128;
        hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3((unsigned)tiles, 1), dim3(512), 2 * P64_SLOT, stream, g);
    }
    return launch_check("sgclip_gemm_tn_bf16");
}
