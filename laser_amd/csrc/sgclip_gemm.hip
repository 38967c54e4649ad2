// bf16 MFMA GEMM for the CLIP towers:  C = epilogue(A[M,K] * W[N,K]^T).
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights
// [out,in]), so A and W tiles are staged identically: 128 rows x 64 k of bf16
// per tile, written to LDS by global_load_lds_dwordx4 (1 KiB per wave
// instruction = 8 rows x 128 B).  Bank conflicts of the ds_read_b128 fragment
// reads are removed by an XOR swizzle of the 16-byte chunk index with bits 1..3
// of the row, applied on the per-lane global SOURCE address (the LDS image of a
// DMA is lane-linear) and again on the read address.
//
// Workgroup = 4 waves (2 x 2), tile 128 x 128, each wave 64 x 64 as 4 x 4
// v_mfma_f32_16x16x32_bf16 accumulators; double-buffered LDS, one barrier per
// 64-deep K step with the next tile's DMA issued before the current tile's MFMAs.
// Epilogues fuse bias, quick-GELU, the fp32 residual add and the patch-embedding
// row remap + position embedding.
#include "common.h"

namespace sgclip {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const bf16_t *A;
    const bf16_t *W;
    void *C;
    const float *bias;
    const float *aux;
    int lda, ldw, ldc;
    int M, N, K;
    int aux_rows;
    int tiles_m, tiles_n;
};

// byte offset of 16-byte chunk `c` (0..7) of tile row `row` in the swizzled LDS image
__device__ __forceinline__ int lds_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// one operand tile: 16 pieces of 1 KiB, 4 per wave
__device__ __forceinline__ void stage_tile(const bf16_t *__restrict__ G, int ld, int row0, int max_row, int k0,
                                           unsigned char *lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int piece = wave * 4 + i;
        const int row = piece * 8 + (lane >> 3);
        const int pos = lane & 7;
        const int c = pos ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row - 1;
        const bf16_t *src = G + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * TILE_BYTES];  // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap: blocks that share an XCD (same blockIdx % 8) walk
    // neighbouring tiles of one A row panel, so the panel stays in that XCD's L2.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = swz / g.tiles_n, tn = swz - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage_tile(g.A, g.lda, m0, g.M, 0, lds, wave, lane);
    stage_tile(g.W, g.ldw, n0, g.N, 0, lds + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        unsigned char *la = lds + cur * 2 * TILE_BYTES;
        unsigned char *lb = la + TILE_BYTES;
        if (kt + 1 < nk) {
            unsigned char *na = lds + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(g.A, g.lda, m0, g.M, (kt + 1) * BK, na, wave, lane);
            stage_tile(g.W, g.ldw, n0, g.N, (kt + 1) * BK, na + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = wm * 64 + i * 16 + frow;
                af[i] = *(const bf16x8 *)(la + lds_off(row, kk * 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = wn * 64 + j * 16 + frow;
                bfr[j] = *(const bf16x8 *)(lb + lds_off(row, kk * 4 + fq));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: acc[i][j][e] is C[m0 + wm*64 + i*16 + fq*4 + e][n0 + wn*64 + j*16 + frow]
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n0 + wn * 64 + j * 16 + frow;
        if (n >= g.N) continue;
        float bias = 0.f;
        if (EPI == SGCLIP_EPI_BIAS_BF16 || EPI == SGCLIP_EPI_BIAS_GELU_BF16 || EPI == SGCLIP_EPI_BIAS_RESID_F32)
            bias = g.bias[n];
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
                if (m >= g.M) continue;
                float v = acc[i][j][e] + bias;
                if (EPI == SGCLIP_EPI_BIAS_BF16) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] = (bf16_t)v;
                } else if (EPI == SGCLIP_EPI_BIAS_GELU_BF16) {
                    ((bf16_t *)g.C)[(size_t)m * g.ldc + n] = (bf16_t)quick_gelu(v);
                } else if (EPI == SGCLIP_EPI_BIAS_RESID_F32) {
                    float *p = (float *)g.C + (size_t)m * g.ldc + n;
                    *p = *p + v;
                } else if (EPI == SGCLIP_EPI_PATCH_F32) {
                    const int P = g.aux_rows;
                    const int img = m / P, pp = m - img * P;
                    const size_t row = (size_t)img * (P + 1) + 1 + pp;
                    ((float *)g.C)[row * g.ldc + n] = v + g.aux[(size_t)(1 + pp) * g.ldc + n];
                } else {
                    ((float *)g.C)[(size_t)m * g.ldc + n] = v;
                }
            }
        }
    }
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_gemm_bf16(const void *A_dev, int lda, const void *W_dev, int ldw, void *C_dev, int ldc, int M,
                                int N, int K, int epilogue, const float *bias_dev, const float *aux_dev, int aux_rows,
                                void *stream_) {
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
                            epilogue == SGCLIP_EPI_BIAS_RESID_F32;
    if ((needs_bias && !bias_dev) || (epilogue == SGCLIP_EPI_PATCH_F32 && (!aux_dev || aux_rows <= 0))) {
        set_error("sgclip_gemm_bf16: epilogue %d needs bias/aux", epilogue);
        return SGCLIP_EINVAL;
    }
    GemmArgs g;
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
    g.aux_rows = aux_rows;
    g.tiles_m = ceil_div(M, BM);
    g.tiles_n = ceil_div(N, BN);
    const dim3 grid((unsigned)g.tiles_m * g.tiles_n), block(GEMM_THREADS);
    KernelTimer timer(TK_GEMM0 + epilogue, 2.0 * M * N * K, stream);
    switch (epilogue) {
        case SGCLIP_EPI_BIAS_BF16: hipLaunchKernelGGL(gemm_kernel<SGCLIP_EPI_BIAS_BF16>, grid, block, 0, stream, g); break;
        case SGCLIP_EPI_BIAS_GELU_BF16: hipLaunchKernelGGL(gemm_kernel<SGCLIP_EPI_BIAS_GELU_BF16>, grid, block, 0, stream, g); break;
        case SGCLIP_EPI_BIAS_RESID_F32: hipLaunchKernelGGL(gemm_kernel<SGCLIP_EPI_BIAS_RESID_F32>, grid, block, 0, stream, g); break;
        case SGCLIP_EPI_PATCH_F32: hipLaunchKernelGGL(gemm_kernel<SGCLIP_EPI_PATCH_F32>, grid, block, 0, stream, g); break;
        case SGCLIP_EPI_F32: hipLaunchKernelGGL(gemm_kernel<SGCLIP_EPI_F32>, grid, block, 0, stream, g); break;
        default: set_error("sgclip_gemm_bf16: unknown epilogue %d", epilogue); return SGCLIP_EINVAL;
    }
    return launch_check("sgclip_gemm_bf16");
}
