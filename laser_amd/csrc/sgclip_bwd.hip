// Backward building blocks of the train step (scripts/train_sgclip.py:682-694, 877-890 run the
// same forward() under autograd; the ops below are the adjoints of the kernels of this library).
// They are correctness-first: plain HBM-bound / VALU kernels around the bf16 MFMA GEMM, which
// does the heavy data- and weight-gradient products (sgclip_gemm_bf16 on transposed operands).
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace sgclip {

// ---------------------------------------------------------------------------
// out[c][r] = in[r][c], bf16, 64 x 64 tiles through LDS; columns rows..ld_out-1 of out are zeroed
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t *__restrict__ in, long long ld_in, int rows,
                                                             int cols, bf16_t *__restrict__ out, long long ld_out) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (bf16_t)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < ld_out) out[(size_t)c * ld_out + r] = tile[tx][i];
    }
}

// The same transpose with 16-byte global accesses: cols, ld_in, ld_out multiples of 8 and both
// pointers 16-byte aligned.  LDS row stride 66 elements (33 words): the 8 x 8 (row chunk, column)
// pattern of the read-back falls on 32 distinct banks.
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const bf16_t *__restrict__ in, long long ld_in, int rows,
                                                                 int cols, bf16_t *__restrict__ out, long long ld_out) {
    __shared__ unsigned int tile[64 * 33];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = threadIdx.x + 256 * i;  // 64 rows x 8 chunks of 8 columns
        const int r = idx >> 3, ch = idx & 7;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *(const uint4 *)(in + (size_t)(r0 + r) * ld_in + c0 + ch * 8);
        unsigned int *dst = tile + r * 33 + ch * 4;
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    }
    __syncthreads();
    const unsigned short *t16 = (const unsigned short *)tile;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = threadIdx.x + 256 * i;  // 64 columns x 8 chunks of 8 rows
        const int c = idx >> 3, rc = idx & 7;
        if (c0 + c >= cols || r0 + rc * 8 >= ld_out) continue;
        unsigned int w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned int lo = t16[(rc * 8 + 2 * j) * 66 + c], hi = t16[(rc * 8 + 2 * j + 1) * 66 + c];
            w[j] = lo | (hi << 16);
        }
        *(uint4 *)(out + (size_t)(c0 + c) * ld_out + r0 + rc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// out[n] += sum over a slice of rows of in[m][n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ in, long long ld, int M, int N,
                                                     float *__restrict__ out, int rows_per_block) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int m = m0 + part; m < m1; m += 4) s += (float)in[(size_t)m * ld + n];
    __shared__ float red[4][64];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && n < N) atomicAdd(out + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__device__ __forceinline__ float sigmoid1702(float x) { return 1.0f / (1.0f + __expf(-1.702f * x)); }

// eight elements per thread (16-byte accesses; torch allocations are 16-byte aligned), scalar tail
__global__ void quick_gelu_kernel(const bf16_t *__restrict__ u, bf16_t *__restrict__ g, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const bf16x8 x = *(const bf16x8 *)(u + i);
        bf16x8 y;
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = (bf16_t)((float)x[q] * sigmoid1702((float)x[q]));
        *(bf16x8 *)(g + i) = y;
    } else {
        for (long long j = i; j < n; j++) g[j] = (bf16_t)((float)u[j] * sigmoid1702((float)u[j]));
    }
}

__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = sigmoid1702(x);
    return s + 1.702f * x * s * (1.0f - s);
}

__global__ void quick_gelu_bwd_kernel(const bf16_t *__restrict__ u, const bf16_t *__restrict__ dg, bf16_t *__restrict__ du,
                                      long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const bf16x8 x = *(const bf16x8 *)(u + i), d = *(const bf16x8 *)(dg + i);
        bf16x8 y;
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = (bf16_t)((float)d[q] * quick_gelu_grad((float)x[q]));
        *(bf16x8 *)(du + i) = y;
    } else {
        for (long long j = i; j < n; j++) du[j] = (bf16_t)((float)dg[j] * quick_gelu_grad((float)u[j]));
    }
}

// out = bf16(in) and colsum[n] += sum_m in[m][n] in one pass over in f32 [M, N] (contiguous rows, N % 8 == 0,
// N / 8 <= 256): the cast of a gradient for the bf16 GEMMs and the bias gradient it also is.  Thread = 8 columns,
// a block's blockDim.x / (N / 8) row lanes walk its slice of rows.
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float *__restrict__ in, bf16_t *__restrict__ out, int M, int N,
                                                          float *__restrict__ colsum, int rows_per_block) {
    __shared__ float red[256][9];
    const int ct = N >> 3, rt = blockDim.x / ct;
    const int tx = threadIdx.x % ct, ty = threadIdx.x / ct;
    const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ty < rt) {
        auto one = [&](int m, const float4 &a, const float4 &b) {
            bf16x8 y;
            y[0] = (bf16_t)a.x; y[1] = (bf16_t)a.y; y[2] = (bf16_t)a.z; y[3] = (bf16_t)a.w;
            y[4] = (bf16_t)b.x; y[5] = (bf16_t)b.y; y[6] = (bf16_t)b.z; y[7] = (bf16_t)b.w;
            *(bf16x8 *)(out + (size_t)m * N + tx * 8) = y;
            s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
            s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
        };
        int m = m0 + ty;
        for (; m + 3 * rt < m1; m += 4 * rt) {   // four rows in flight per thread
            float4 a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float *p = in + (size_t)(m + r * rt) * N + tx * 8;
                a[r] = *(const float4 *)p;
                b[r] = *(const float4 *)(p + 4);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) one(m + r * rt, a[r], b[r]);
        }
        for (; m < m1; m += rt) {
            const float *p = in + (size_t)m * N + tx * 8;
            one(m, *(const float4 *)p, *(const float4 *)(p + 4));
        }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) red[threadIdx.x][q] = s[q];
    __syncthreads();
    // one dword per lane, neighbouring lanes on neighbouring columns: a wave's atomic instruction covers whole cache lines
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        float t = 0.f;
        for (int y = 0; y < rt; y++) t += red[y * ct + (c >> 3)][c & 7];
        atomicAdd(colsum + c, t);
    }
}

// out += <a, b>
__global__ __launch_bounds__(256) void dot_f32_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n,
                                                      float *__restrict__ out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// eight elements per thread (two 16-byte loads, one 16-byte store), scalar tail
__global__ void cast_f32_bf16_kernel(const float *__restrict__ in, bf16_t *__restrict__ out, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *(const float4 *)(in + i), b = *(const float4 *)(in + i + 4);
        bf16x8 y;
        y[0] = (bf16_t)a.x; y[1] = (bf16_t)a.y; y[2] = (bf16_t)a.z; y[3] = (bf16_t)a.w;
        y[4] = (bf16_t)b.x; y[5] = (bf16_t)b.y; y[6] = (bf16_t)b.z; y[7] = (bf16_t)b.w;
        *(bf16x8 *)(out + i) = y;
    } else {
        for (long long j = i; j < n; j++) out[j] = (bf16_t)in[j];
    }
}

// ---------------------------------------------------------------------------
// LayerNorm backward: one wave per row, the row held in registers (NV values per lane, d <= 64 NV);
// each wave walks rpw rows and keeps its dw / db partial sums in registers, the block's four
// waves are summed through LDS and added to dw / db with one atomic per column and block.
//   xh = (x - mean) rstd;  g = dy w;  dx = rstd (g - mean(g) - xh mean(g xh))
// ---------------------------------------------------------------------------
constexpr int LN_RPW_MAX = 16;  // rows per wave: fewer when the launch would not fill the chip

// NV4 = float4 groups per lane: lane l holds columns 4 (l + 64 j) .. + 3, j < NV4 (d <= 256 NV4, d % 4 == 0)
template <int NV4, bool DY_BF16>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, long long x_stride, int rows, int d,
                                                            const float *__restrict__ w, float eps, const void *dy_,
                                                            float *dx, int accumulate, float *__restrict__ dw,
                                                            float *__restrict__ db, int rpw) {
    __shared__ float red[4][2][NV4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = d >> 2;
    float4 wv[NV4], dwacc[NV4], dbacc[NV4];
#pragma unroll
    for (int j = 0; j < NV4; j++) {
        const int v = lane + 64 * j;
        wv[j] = v < nvec ? ((const float4 *)w)[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        dwacc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        dbacc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row0 = (blockIdx.x * 4 + wave) * rpw;
    const float inv_d = 1.0f / d;
    for (int r = 0; r < rpw; r++) {
        const int row = row0 + r;
        if (row >= rows) break;
        const float4 *xr = (const float4 *)(x + (size_t)row * x_stride);
        float4 xv[NV4], dyv[NV4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            const int v = lane + 64 * j;
            xv[j] = dyv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < nvec) {
                xv[j] = xr[v];
                if (DY_BF16) {
                    const bf16x4 t = ((const bf16x4 *)((const bf16_t *)dy_ + (size_t)row * d))[v];
                    dyv[j] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
                } else {
                    dyv[j] = ((const float4 *)((const float *)dy_ + (size_t)row * d))[v];
                }
            }
            s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            if (lane + 64 * j < nvec) {
                const float a0 = xv[j].x - mean, a1 = xv[j].y - mean, a2 = xv[j].z - mean, a3 = xv[j].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            const bool ok = lane + 64 * j < nvec;
            float4 xh;
            xh.x = ok ? (xv[j].x - mean) * rstd : 0.f;
            xh.y = ok ? (xv[j].y - mean) * rstd : 0.f;
            xh.z = ok ? (xv[j].z - mean) * rstd : 0.f;
            xh.w = ok ? (xv[j].w - mean) * rstd : 0.f;
            xv[j] = xh;
            const float g0 = dyv[j].x * wv[j].x, g1 = dyv[j].y * wv[j].y, g2 = dyv[j].z * wv[j].z, g3 = dyv[j].w * wv[j].w;
            sg += (g0 + g1) + (g2 + g3);
            sgx += (g0 * xh.x + g1 * xh.y) + (g2 * xh.z + g3 * xh.w);
            dwacc[j].x += dyv[j].x * xh.x; dwacc[j].y += dyv[j].y * xh.y; dwacc[j].z += dyv[j].z * xh.z; dwacc[j].w += dyv[j].w * xh.w;
            dbacc[j].x += dyv[j].x; dbacc[j].y += dyv[j].y; dbacc[j].z += dyv[j].z; dbacc[j].w += dyv[j].w;
        }
        const float mg = wave_sum(sg) * inv_d, mgx = wave_sum(sgx) * inv_d;
        float4 *dxr = (float4 *)(dx + (size_t)row * d);
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            const int v = lane + 64 * j;
            if (v < nvec) {
                float4 o;
                o.x = rstd * (dyv[j].x * wv[j].x - mg - xv[j].x * mgx);
                o.y = rstd * (dyv[j].y * wv[j].y - mg - xv[j].y * mgx);
                o.z = rstd * (dyv[j].z * wv[j].z - mg - xv[j].z * mgx);
                o.w = rstd * (dyv[j].w * wv[j].w - mg - xv[j].w * mgx);
                if (accumulate) {
                    const float4 c = dxr[v];
                    o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
                }
                dxr[v] = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV4; j++) {
        ((float4 *)red[wave][0])[lane + 64 * j] = dwacc[j];
        ((float4 *)red[wave][1])[lane + 64 * j] = dbacc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        atomicAdd(dw + c, red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
        atomicAdd(db + c, red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
    }
}

// ---------------------------------------------------------------------------
// Attention backward for short sequences (T <= 128, head_dim 64): one workgroup per
// (sequence, head); Q, K, V, dO as bf16 and the T x T probabilities as f32 in LDS.
//   P = softmax(Q K^T / 8 + mask);  dV = P^T dO;  dP = dO V^T;  D_i = sum_k P_ik dP_ik
//   dS = P (dP - D);  dQ = dS K / 8;  dK = dS^T Q / 8
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_bwd_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dout,
                                                            bf16_t *__restrict__ dqkv, int T, int heads, int causal,
                                                            const int32_t *__restrict__ key_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = heads * 64;
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const size_t ld = (size_t)3 * d;
    bf16_t *Q = (bf16_t *)smem, *K = Q + T * 64, *V = K + T * 64, *dO = V + T * 64;
    float *P = (float *)(dO + T * 64);  // [T][T], later dS in place
    const bf16_t *base = qkv + (size_t)seq * T * ld + head * 64;
    const int32_t *km = key_mask ? key_mask + (size_t)seq * T : nullptr;
    for (int i = threadIdx.x; i < T * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        Q[i] = base[(size_t)t * ld + c];
        K[i] = base[(size_t)t * ld + d + c];
        V[i] = base[(size_t)t * ld + 2 * d + c];
        dO[i] = dout[((size_t)seq * T + t) * d + head * 64 + c];
    }
    __syncthreads();
    // scores
    for (int e = threadIdx.x; e < T * T; e += 256) {
        const int i = e / T, k = e - i * T;
        bool ok = true;
        if (causal) ok = k <= i;
        if (km) ok = ok && km[k] != 0;
        float s = 0.f;
        if (ok) {
            for (int c = 0; c < 64; c++) s += (float)Q[i * 64 + c] * (float)K[k * 64 + c];
            s *= 0.125f;
        }
        P[e] = ok ? s : -1.0e30f;
    }
    __syncthreads();
    // row softmax: wave per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < T; i += 4) {
        float mx = -1.0e30f;
        for (int k = lane; k < T; k += 64) mx = fmaxf(mx, P[i * T + k]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int k = lane; k < T; k += 64) {
            const float p = P[i * T + k] > -0.5e30f ? __expf(P[i * T + k] - mx) : 0.f;
            P[i * T + k] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        for (int k = lane; k < T; k += 64) P[i * T + k] *= inv;
    }
    __syncthreads();
    bf16_t *obase = dqkv + (size_t)seq * T * ld + head * 64;
    // dV[k][c] = sum_i P[i][k] dO[i][c]
    for (int e = threadIdx.x; e < T * 64; e += 256) {
        const int k = e >> 6, c = e & 63;
        float s = 0.f;
        for (int i = 0; i < T; i++) s += P[i * T + k] * (float)dO[i * 64 + c];
        obase[(size_t)k * ld + 2 * d + c] = (bf16_t)s;
    }
    __syncthreads();
    // dS in place: wave per row (needs the whole row of P and dP)
    for (int i = wave; i < T; i += 4) {
        float dpv[2];  // T <= 128: at most two keys per lane
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = lane + 64 * j;
            dpv[j] = 0.f;
            if (k < T) {
                float s = 0.f;
                for (int c = 0; c < 64; c++) s += (float)dO[i * 64 + c] * (float)V[k * 64 + c];
                dpv[j] = s;
                dsum += P[i * T + k] * s;
            }
        }
        dsum = wave_sum(dsum);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = lane + 64 * j;
            if (k < T) P[i * T + k] = P[i * T + k] * (dpv[j] - dsum) * 0.125f;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * 64; e += 256) {
        const int t = e >> 6, c = e & 63;
        float sq = 0.f, sk = 0.f;
        for (int j = 0; j < T; j++) {
            sq += P[t * T + j] * (float)K[j * 64 + c];   // dQ[t] = sum_k dS[t][k] K[k]
            sk += P[j * T + t] * (float)Q[j * 64 + c];   // dK[t] = sum_i dS[i][t] Q[i]
        }
        obase[(size_t)t * ld + c] = (bf16_t)sq;
        obase[(size_t)t * ld + d + c] = (bf16_t)sk;
    }
}

// ---------------------------------------------------------------------------
// The same adjoint on the matrix cores (v_mfma_f32_16x16x32_bf16), T padded to TP = 32 / 64 / 96 / 128.
// One workgroup (4 waves) per (sequence, head).
//   load:    Q, K, V, dO row-major [TP][72] (stride 72 keeps 16-byte fragment reads conflict free) and
//            Q^T, K^T, dO^T [64][TP + 8], zero padded
//   phase A: wave w owns query tiles w, w + 4: S = Q K^T and dP = dO V^T for whole rows, row softmax and
//            D = sum_k P dP by in-register + 16-lane reductions, dS = P (dP - D) / 8; P and dS kept as packed bf16
//   write:   P^T, dS, dS^T [TP][TP + 8] over the (now dead) row-major operands
//   phase B: dV = P^T dO, dK = dS^T Q, dQ = dS K, each wave two 16-row tiles x 64 columns
// ---------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int TP>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dout,
                                                                 bf16_t *__restrict__ dqkv, int T, int heads, int causal,
                                                                 const int32_t *__restrict__ key_mask,
                                                                 float *__restrict__ dbias, int colsum_off) {
    constexpr int LDR = 72, LDT = TP + 8, NT = TP / 16, RT = (NT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Qt = (bf16_t *)smem, *Kt = Qt + 64 * LDT, *dOt = Kt + 64 * LDT;
    bf16_t *rm = dOt + 64 * LDT;
    bf16_t *Qr = rm, *Kr = Qr + TP * LDR, *Vr = Kr + TP * LDR, *dOr = Vr + TP * LDR;
    bf16_t *Pt = rm, *dS = Pt + TP * LDT, *dSt = dS + TP * LDT;   // phase B, over the row-major operands

    const int d = heads * 64;
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const size_t ld = (size_t)3 * d;
    const bf16_t *base = qkv + (size_t)seq * T * ld + head * 64;
    const bf16_t *dobase = dout + (size_t)seq * T * d + head * 64;
    const int32_t *km = key_mask ? key_mask + (size_t)seq * T : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- load: consecutive lanes take consecutive rows, so the transposed 2-byte stores are conflict free
    for (int idx = threadIdx.x; idx < TP * 8; idx += 256) {
        const int t = idx % TP, ch = idx / TP;
        bf16x8 q = {}, k = {}, v = {}, o = {};
        if (t < T) {
            const bf16_t *r = base + (size_t)t * ld + ch * 8;
            q = *(const bf16x8 *)r;
            k = *(const bf16x8 *)(r + d);
            v = *(const bf16x8 *)(r + 2 * d);
            o = *(const bf16x8 *)(dobase + (size_t)t * d + ch * 8);
        }
        *(bf16x8 *)(Qr + t * LDR + ch * 8) = q;
        *(bf16x8 *)(Kr + t * LDR + ch * 8) = k;
        *(bf16x8 *)(Vr + t * LDR + ch * 8) = v;
        *(bf16x8 *)(dOr + t * LDR + ch * 8) = o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            Qt[(ch * 8 + j) * LDT + t] = q[j];
            Kt[(ch * 8 + j) * LDT + t] = k[j];
            dOt[(ch * 8 + j) * LDT + t] = o[j];
        }
    }
    __syncthreads();

    // ---- phase A
    bf16x4 pP[RT][NT], pS[RT][NT];
#pragma unroll
    for (int r2 = 0; r2 < RT; r2++) {
        const int rt = wave + 4 * r2;
        if (rt >= NT) break;
        f32x4_t S[NT], dP[NT];
        bf16x8 aq[2], ao[2];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            aq[kk] = *(const bf16x8 *)(Qr + (rt * 16 + frow) * LDR + kk * 32 + fq * 8);
            ao[kk] = *(const bf16x8 *)(dOr + (rt * 16 + frow) * LDR + kk * 32 + fq * 8);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
            S[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            dP[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const bf16x8 bk = *(const bf16x8 *)(Kr + (ct * 16 + frow) * LDR + kk * 32 + fq * 8);
                const bf16x8 bv = *(const bf16x8 *)(Vr + (ct * 16 + frow) * LDR + kk * 32 + fq * 8);
                S[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[kk], bk, S[ct], 0, 0, 0);
                dP[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[kk], bv, dP[ct], 0, 0, 0);
            }
        }
        // element e of tile ct: query i = rt*16 + fq*4 + e, key k = ct*16 + frow
        float mx[4] = {-1.0e30f, -1.0e30f, -1.0e30f, -1.0e30f};
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
            const int k = ct * 16 + frow;
            const bool kok = k < T && (!km || km[k < T ? k : 0] != 0);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = rt * 16 + fq * 4 + e;
                const bool ok = kok && (!causal || k <= i);
                S[ct][e] = ok ? S[ct][e] * 0.125f : -1.0e30f;
                mx[e] = fmaxf(mx[e], S[ct][e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], o));
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float p = S[ct][e] > -0.5e30f ? __expf(S[ct][e] - mx[e]) : 0.f;
                S[ct][e] = p;
                sum[e] += p;
            }
        float dsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum[e] += __shfl_xor(sum[e], o);
            sum[e] = sum[e] > 0.f ? 1.0f / sum[e] : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                S[ct][e] *= sum[e];
                dsum[e] += S[ct][e] * dP[ct][e];
            }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) dsum[e] += __shfl_xor(dsum[e], o);
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                pP[r2][ct][e] = (bf16_t)S[ct][e];
                pS[r2][ct][e] = (bf16_t)(S[ct][e] * (dP[ct][e] - dsum[e]) * 0.125f);
            }
    }
    __syncthreads();   // every wave has finished reading the row-major operands

    // ---- write P^T, dS^T (4 consecutive queries per store) and dS
#pragma unroll
    for (int r2 = 0; r2 < RT; r2++) {
        const int rt = wave + 4 * r2;
        if (rt >= NT) break;
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
            const int k = ct * 16 + frow, i0 = rt * 16 + fq * 4;
            *(bf16x4 *)(Pt + k * LDT + i0) = pP[r2][ct];
            *(bf16x4 *)(dSt + k * LDT + i0) = pS[r2][ct];
#pragma unroll
            for (int e = 0; e < 4; e++) dS[(i0 + e) * LDT + k] = pS[r2][ct][e];
        }
    }
    __syncthreads();

    // ---- phase B
    bf16_t *obase = dqkv + (size_t)seq * T * ld + head * 64;
    // dbias: the column sums of this workgroup's dqkv block (the gradient of the q, k, v biases) ride along: per lane the
    // rounded values of its column over its rows, then the lane quarters, the four waves through LDS, and one
    // 64-float contiguous atomic per matrix
    float cs[3][4];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 4; c++) cs[q][c] = 0.f;
#pragma unroll
    for (int r2 = 0; r2 < RT; r2++) {
        const int rt = wave + 4 * r2;
        if (rt >= NT) break;
        f32x4_t aV[4], aK[4], aQ[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            aV[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            aK[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            aQ[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s0 = 0; s0 < TP; s0 += 32) {
            const bf16x8 fp = *(const bf16x8 *)(Pt + (rt * 16 + frow) * LDT + s0 + fq * 8);
            const bf16x8 fst = *(const bf16x8 *)(dSt + (rt * 16 + frow) * LDT + s0 + fq * 8);
            const bf16x8 fs = *(const bf16x8 *)(dS + (rt * 16 + frow) * LDT + s0 + fq * 8);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bf16x8 bo = *(const bf16x8 *)(dOt + (c * 16 + frow) * LDT + s0 + fq * 8);
                const bf16x8 bq = *(const bf16x8 *)(Qt + (c * 16 + frow) * LDT + s0 + fq * 8);
                const bf16x8 bk = *(const bf16x8 *)(Kt + (c * 16 + frow) * LDT + s0 + fq * 8);
                aV[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp, bo, aV[c], 0, 0, 0);
                aK[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fst, bq, aK[c], 0, 0, 0);
                aQ[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fs, bk, aQ[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int t = rt * 16 + fq * 4 + e;
            if (t >= T) continue;
            bf16_t *o = obase + (size_t)t * ld + frow;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bf16_t gq = (bf16_t)aQ[c][e], gk = (bf16_t)aK[c][e], gv = (bf16_t)aV[c][e];
                o[c * 16] = gq;
                o[d + c * 16] = gk;
                o[2 * d + c * 16] = gv;
                cs[0][c] += (float)gq;
                cs[1][c] += (float)gk;
                cs[2][c] += (float)gv;
            }
        }
    }
    if (dbias) {
        float *part = (float *)(smem + colsum_off);   // [4 waves][3][64]
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float v = cs[q][c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (fq == 0) part[(wave * 3 + q) * 64 + c * 16 + frow] = v;
            }
        __syncthreads();
        if (threadIdx.x < 192) {
            const int q = threadIdx.x >> 6, col = threadIdx.x & 63;
            const float v = part[q * 64 + col] + part[(3 + q) * 64 + col] + part[(6 + q) * 64 + col] + part[(9 + q) * 64 + col];
            atomicAdd(dbias + (size_t)q * d + head * 64 + col, v);
        }
    }
}

template <int TP>
static void attention_bwd_mfma_launch(const bf16_t *qkv, const bf16_t *dout, bf16_t *dqkv, int n_seq, int T, int heads,
                                      int causal, const int32_t *km, float *dbias, hipStream_t stream) {
    constexpr int LDT = TP + 8;
    constexpr size_t phase_a = (size_t)4 * TP * 72 * 2, phase_b = (size_t)3 * TP * LDT * 2;
    constexpr size_t body = (size_t)3 * 64 * LDT * 2 + (phase_a > phase_b ? phase_a : phase_b);
    constexpr size_t lds = body + 4 * 3 * 64 * sizeof(float);   // + the waves' column sums
    static_assert(lds <= 160 * 1024, "attention backward tile exceeds the LDS");
    static_assert(body % 16 == 0, "column sums start aligned");
    if (!lds_attr((const void *)attention_bwd_mfma_kernel<TP>, (int)lds)) return;   // the error text is set; launch_check reports it
    hipLaunchKernelGGL(attention_bwd_mfma_kernel<TP>, dim3((unsigned)(n_seq * heads)), dim3(256), lds, stream, qkv, dout, dqkv,
                       T, heads, causal, km, dbias, (int)body);
}

// ---------------------------------------------------------------------------
// The same adjoint for sequences of any length (ViT-L/14 at 336 px: T = 577), in blocks of 64 queries x 64 keys.
// The forward pass saves nothing, so one workgroup (4 waves) per (sequence, head) makes three sweeps over the
// block pairs, recomputing S = Q K^T / 8 and dP = dO V^T on the matrix cores each time:
//   sweep 0  per query row: running maximum m, normaliser l and sum_k exp(S - m) dP (online, rescaled when m moves)
//            -> m, 1 / l and D = sum_k P dP in LDS
//   sweep 1  key block outer, query blocks inner: P = exp(S - m) / l, dS = P (dP - D) / 8;
//            dV += P^T dO, dK += dS^T Q accumulate in registers (wave = 16 keys x 64 channels)
//   sweep 2  query block outer, key blocks inner: dQ += dS K (wave = 16 queries x 64 channels)
// Operand blocks live in LDS row-major [64][72] and transposed [64 channels][72]; P^T, dS^T / dS pass through
// LDS as bf16 to become MFMA operands, as in the short-sequence kernel.  Causal sweeps skip the block pairs
// above the diagonal.
// ---------------------------------------------------------------------------
constexpr int LB = 64, LLD = 72, LBLK = LB * LLD;   // block rows, LDS row stride and block size (bf16 elements)

struct LongBlocks {
    bf16_t *Qr, *dOr, *Kr, *Vr, *Qt, *dOt, *Kt, *X0, *X1;
    float *m, *il, *D;
};

// rows [r0, r0 + 64) of a [T][ld] bf16 matrix (64 channels from column 0) -> row-major and/or transposed LDS blocks
__device__ __forceinline__ void long_load(const bf16_t *__restrict__ src, size_t ld, int r0, int T, bf16_t *rowmaj,
                                          bf16_t *transposed) {
    for (int idx = threadIdx.x; idx < LB * 8; idx += 256) {
        const int r = idx % LB, ch = idx / LB;
        bf16x8 v = {};
        if (r0 + r < T) v = *(const bf16x8 *)(src + (size_t)(r0 + r) * ld + ch * 8);
        if (rowmaj) *(bf16x8 *)(rowmaj + r * LLD + ch * 8) = v;
        if (transposed) {
#pragma unroll
            for (int j = 0; j < 8; j++) transposed[(ch * 8 + j) * LLD + r] = v[j];
        }
    }
}

// S (scaled, masked with -1e30) and dP of this wave's 16 queries against the 64 keys of the block in LDS:
// element e of tile ct is query i0 + fq*4 + e (i0 = first query of the wave's tile), key k0 + ct*16 + frow
__device__ __forceinline__ void long_scores(const LongBlocks &L, int wave, int frow, int fq, int i0, int k0, int T, int causal,
                                            const int32_t *km, f32x4_t (&S)[4], f32x4_t (&dP)[4]) {
    bf16x8 aq[2], ao[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        aq[kk] = *(const bf16x8 *)(L.Qr + (wave * 16 + frow) * LLD + kk * 32 + fq * 8);
        ao[kk] = *(const bf16x8 *)(L.dOr + (wave * 16 + frow) * LLD + kk * 32 + fq * 8);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        S[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        dP[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const bf16x8 bk = *(const bf16x8 *)(L.Kr + (ct * 16 + frow) * LLD + kk * 32 + fq * 8);
            const bf16x8 bv = *(const bf16x8 *)(L.Vr + (ct * 16 + frow) * LLD + kk * 32 + fq * 8);
            S[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[kk], bk, S[ct], 0, 0, 0);
            dP[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[kk], bv, dP[ct], 0, 0, 0);
        }
        const int k = k0 + ct * 16 + frow;
        const bool kok = k < T && (!km || km[k < T ? k : 0] != 0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = i0 + fq * 4 + e;
            const bool ok = kok && (!causal || k <= i);
            S[ct][e] = ok ? S[ct][e] * 0.125f : -1.0e30f;
        }
    }
}

__global__ __launch_bounds__(256) void attention_bwd_long_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dout,
                                                                 bf16_t *__restrict__ dqkv, int T, int heads, int causal,
                                                                 const int32_t *__restrict__ key_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = (T + LB - 1) / LB, TP = nb * LB;
    LongBlocks L;
    L.Qr = (bf16_t *)smem; L.dOr = L.Qr + LBLK; L.Kr = L.dOr + LBLK; L.Vr = L.Kr + LBLK;
    L.Qt = L.Vr + LBLK; L.dOt = L.Qt + LBLK; L.Kt = L.dOt + LBLK; L.X0 = L.Kt + LBLK; L.X1 = L.X0 + LBLK;
    L.m = (float *)(L.X1 + LBLK); L.il = L.m + TP; L.D = L.il + TP;

    const int d = heads * 64;
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const size_t ld = (size_t)3 * d;
    const bf16_t *qbase = qkv + (size_t)seq * T * ld + head * 64;
    const bf16_t *dobase = dout + (size_t)seq * T * d + head * 64;
    bf16_t *obase = dqkv + (size_t)seq * T * ld + head * 64;
    const int32_t *km = key_mask ? key_mask + (size_t)seq * T : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- sweep 0: row statistics
    for (int qb = 0; qb < nb; qb++) {
        __syncthreads();
        long_load(qbase, ld, qb * LB, T, L.Qr, nullptr);
        long_load(dobase, (size_t)d, qb * LB, T, L.dOr, nullptr);
        float m[4], l[4], aD[4];
#pragma unroll
        for (int e = 0; e < 4; e++) m[e] = -1.0e30f, l[e] = 0.f, aD[e] = 0.f;
        const int kb_end = causal ? min(nb, qb + 1) : nb;
        for (int kb = 0; kb < kb_end; kb++) {
            __syncthreads();
            long_load(qbase + d, ld, kb * LB, T, L.Kr, nullptr);
            long_load(qbase + 2 * d, ld, kb * LB, T, L.Vr, nullptr);
            __syncthreads();
            f32x4_t S[4], dP[4];
            long_scores(L, wave, frow, fq, qb * LB + wave * 16, kb * LB, T, causal, km, S, dP);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float bm = fmaxf(fmaxf(S[0][e], S[1][e]), fmaxf(S[2][e], S[3][e]));
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
                const float mn = fmaxf(m[e], bm);
                const float sc = __expf(m[e] - mn);
                float ls = 0.f, ds = 0.f;
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {
                    const float p = S[ct][e] > -0.5e30f ? __expf(S[ct][e] - mn) : 0.f;
                    ls += p;
                    ds += p * dP[ct][e];
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    ls += __shfl_xor(ls, o);
                    ds += __shfl_xor(ds, o);
                }
                l[e] = l[e] * sc + ls;
                aD[e] = aD[e] * sc + ds;
                m[e] = mn;
            }
        }
        if (frow == 0) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = qb * LB + wave * 16 + fq * 4 + e;
                const float inv = l[e] > 0.f ? 1.0f / l[e] : 0.f;
                L.m[i] = m[e];
                L.il[i] = inv;
                L.D[i] = aD[e] * inv;
            }
        }
    }

    // ---- sweep 1: dK, dV
    for (int kb = 0; kb < nb; kb++) {
        __syncthreads();
        long_load(qbase + d, ld, kb * LB, T, L.Kr, nullptr);
        long_load(qbase + 2 * d, ld, kb * LB, T, L.Vr, nullptr);
        f32x4_t aV[4], aK[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            aV[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            aK[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        for (int qb = causal ? kb : 0; qb < nb; qb++) {
            __syncthreads();   // the previous pair's products have been read
            long_load(qbase, ld, qb * LB, T, L.Qr, L.Qt);
            long_load(dobase, (size_t)d, qb * LB, T, L.dOr, L.dOt);
            __syncthreads();
            f32x4_t S[4], dP[4];
            const int i0 = qb * LB + wave * 16;
            long_scores(L, wave, frow, fq, i0, kb * LB, T, causal, km, S, dP);
            float mi[4], ili[4], Di[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                mi[e] = L.m[i0 + fq * 4 + e];
                ili[e] = L.il[i0 + fq * 4 + e];
                Di[e] = L.D[i0 + fq * 4 + e];
            }
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                bf16x4 pp, ps;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float p = S[ct][e] > -0.5e30f ? __expf(S[ct][e] - mi[e]) * ili[e] : 0.f;
                    pp[e] = (bf16_t)p;
                    ps[e] = (bf16_t)(p * (dP[ct][e] - Di[e]) * 0.125f);
                }
                *(bf16x4 *)(L.X0 + (ct * 16 + frow) * LLD + wave * 16 + fq * 4) = pp;   // P^T  [key][query]
                *(bf16x4 *)(L.X1 + (ct * 16 + frow) * LLD + wave * 16 + fq * 4) = ps;   // dS^T [key][query]
            }
            __syncthreads();
#pragma unroll
            for (int s0 = 0; s0 < LB; s0 += 32) {
                const bf16x8 fp = *(const bf16x8 *)(L.X0 + (wave * 16 + frow) * LLD + s0 + fq * 8);
                const bf16x8 fst = *(const bf16x8 *)(L.X1 + (wave * 16 + frow) * LLD + s0 + fq * 8);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bf16x8 bo = *(const bf16x8 *)(L.dOt + (c * 16 + frow) * LLD + s0 + fq * 8);
                    const bf16x8 bq = *(const bf16x8 *)(L.Qt + (c * 16 + frow) * LLD + s0 + fq * 8);
                    aV[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp, bo, aV[c], 0, 0, 0);
                    aK[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fst, bq, aK[c], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int t = kb * LB + wave * 16 + fq * 4 + e;
            if (t >= T) continue;
            bf16_t *o = obase + (size_t)t * ld + frow;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                o[d + c * 16] = (bf16_t)aK[c][e];
                o[2 * d + c * 16] = (bf16_t)aV[c][e];
            }
        }
    }

    // ---- sweep 2: dQ
    for (int qb = 0; qb < nb; qb++) {
        __syncthreads();
        long_load(qbase, ld, qb * LB, T, L.Qr, nullptr);
        long_load(dobase, (size_t)d, qb * LB, T, L.dOr, nullptr);
        f32x4_t aQ[4];
#pragma unroll
        for (int c = 0; c < 4; c++) aQ[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int i0 = qb * LB + wave * 16;
        const int kb_end = causal ? min(nb, qb + 1) : nb;
        for (int kb = 0; kb < kb_end; kb++) {
            __syncthreads();
            long_load(qbase + d, ld, kb * LB, T, L.Kr, L.Kt);
            long_load(qbase + 2 * d, ld, kb * LB, T, L.Vr, nullptr);
            __syncthreads();
            f32x4_t S[4], dP[4];
            long_scores(L, wave, frow, fq, i0, kb * LB, T, causal, km, S, dP);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float mi = L.m[i0 + fq * 4 + e], ili = L.il[i0 + fq * 4 + e], Di = L.D[i0 + fq * 4 + e];
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {
                    const float p = S[ct][e] > -0.5e30f ? __expf(S[ct][e] - mi) * ili : 0.f;
                    L.X0[(wave * 16 + fq * 4 + e) * LLD + ct * 16 + frow] = (bf16_t)(p * (dP[ct][e] - Di) * 0.125f);   // dS [query][key]
                }
            }
            __syncthreads();
#pragma unroll
            for (int s0 = 0; s0 < LB; s0 += 32) {
                const bf16x8 fs = *(const bf16x8 *)(L.X0 + (wave * 16 + frow) * LLD + s0 + fq * 8);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bf16x8 bk = *(const bf16x8 *)(L.Kt + (c * 16 + frow) * LLD + s0 + fq * 8);
                    aQ[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fs, bk, aQ[c], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int t = i0 + fq * 4 + e;
            if (t >= T) continue;
            bf16_t *o = obase + (size_t)t * ld + frow;
#pragma unroll
            for (int c = 0; c < 4; c++) o[c * 16] = (bf16_t)aQ[c][e];
        }
    }
}

// ---------------------------------------------------------------------------
// C[M][N] (+)= alpha * op(A) op(B), fp32, 64 x 64 tiles (the small products of the fp32 tail:
// projection, similarity).  op(A) is [M][K]: A[m][k] or, transposed, A[k][m]; likewise op(B) [K][N].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void matmul_f32_kernel(const float *__restrict__ A, long long lda, int transA,
                                                         const float *__restrict__ B, long long ldb, int transB,
                                                         float *__restrict__ C, long long ldc, int M, int N, int K,
                                                         float alpha, int accumulate) {
    __shared__ float As[16][65], Bs[16][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int kk = i >> 6, mm = i & 63;
            const int k = k0 + kk;
            float a = 0.f, b = 0.f;
            if (k < K && m0 + mm < M) a = transA ? A[(size_t)k * lda + m0 + mm] : A[(size_t)(m0 + mm) * lda + k];
            if (k < K && n0 + mm < N) b = transB ? B[(size_t)(n0 + mm) * ldb + k] : B[(size_t)k * ldb + n0 + mm];
            As[kk][mm] = a;
            Bs[kk][mm] = b;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = As[kk][ty * 4 + i], b[i] = Bs[kk][tx * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) {
                float *c = C + (size_t)m * ldc + n;
                *c = accumulate ? *c + alpha * acc[i][j] : alpha * acc[i][j];
            }
        }
}

// dlogits from dprobs: softmax over the rows of every column, sigmoid, or pass-through
__global__ void prob_bwd_kernel(const float *__restrict__ p, const float *__restrict__ dp, int K, int N, int mode,
                                float *__restrict__ dl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (mode == SGCLIP_SIM_SOFTMAX) {
        float dot = 0.f;
        for (int k = 0; k < K; k++) dot += p[(size_t)k * N + n] * dp[(size_t)k * N + n];
        for (int k = 0; k < K; k++) dl[(size_t)k * N + n] = p[(size_t)k * N + n] * (dp[(size_t)k * N + n] - dot);
    } else if (mode == SGCLIP_SIM_SIGMOID) {
        for (int k = 0; k < K; k++) {
            const float v = p[(size_t)k * N + n];
            dl[(size_t)k * N + n] = v * (1.0f - v) * dp[(size_t)k * N + n];
        }
    } else {
        for (int k = 0; k < K; k++) dl[(size_t)k * N + n] = dp[(size_t)k * N + n];
    }
}

// xh = x / |x| (rows):  dx = (dxh - xh <xh, dxh>) / |x|;  wave per row
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dxh,
                                                            int rows, int E, float *__restrict__ dx) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * E, *gr = dxh + (size_t)row * E;
    float nn = 0.f, dot = 0.f;
    for (int i = lane; i < E; i += 64) {
        nn += xr[i] * xr[i];
        dot += xr[i] * gr[i];
    }
    nn = wave_sum(nn);
    dot = wave_sum(dot);
    const float inv = rsqrtf(nn);
    for (int i = lane; i < E; i += 64) dx[(size_t)row * E + i] = (gr[i] - xr[i] * dot / nn) * inv;
}

// xh = x / |x| per row; wave per row
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float *__restrict__ x, int rows, int E, float *__restrict__ xh) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * E;
    float nn = 0.f;
    for (int i = lane; i < E; i += 64) nn += xr[i] * xr[i];
    const float inv = 1.0f / sqrtf(wave_sum(nn));
    for (int i = lane; i < E; i += 64) xh[(size_t)row * E + i] = xr[i] * inv;
}

// adjoint of the per-group column mean: dprobs[k][n] = dmean[k][seg[n]] * inv_count[seg[n]]
__global__ void segment_mean_bwd_kernel(const float *__restrict__ dmean, const int32_t *__restrict__ seg,
                                        const float *__restrict__ inv_count, int K, int N, int G, float *__restrict__ dprobs) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int g = seg[n];
    const float w = inv_count[g];
    for (int k = 0; k < K; k++) dprobs[(size_t)k * N + n] = dmean[(size_t)k * G + g] * w;
}

// x[s*T + t][:] = table[ids[s*T + t]][:] + pos[t][:]   (text embeddings, modeling_clip.py:190-207)
__global__ void embed_text_kernel(const float *__restrict__ table, const float *__restrict__ pos, const int32_t *__restrict__ ids,
                                  int T, int d, float *__restrict__ x) {
    const int row = blockIdx.x, t = row % T;
    const float *src = table + (size_t)ids[row] * d, *p = pos + (size_t)t * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(size_t)row * d + i] = src[i] + p[i];
}

// x[s*T][:] = cls[:] + pos[0][:]   (class token row of every image, modeling_clip.py:160-171)
__global__ void embed_cls_kernel(const float *__restrict__ cls, const float *__restrict__ pos, int T, int d, float *__restrict__ x) {
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(size_t)blockIdx.x * T * d + i] = cls[i] + pos[i];
}

// dst[idx[r]][:] += src[r][:]
__global__ void rows_scatter_add_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int d,
                                        float *__restrict__ dst) {
    const float *s = src + (size_t)blockIdx.x * d;
    float *o = dst + (size_t)idx[blockIdx.x] * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) atomicAdd(o + i, s[i]);
}

// AdamW on fp32 master weights (decoupled weight decay)
__global__ void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float c1, float c2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] * (1.0f - lr * wd) - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
}

}  // namespace sgclip

using namespace sgclip;

#define BWD_ARGS_OK(cond, name)                         \
    if (!(cond)) {                                      \
        set_error(name ": bad arguments");              \
        return SGCLIP_EINVAL;                           \
    }

extern "C" int sgclip_transpose_bf16(const void *in_dev, int64_t ld_in, int rows, int cols, void *out_dev, int64_t ld_out,
                                     void *stream) {
    if (rows == 0 || cols == 0) return SGCLIP_OK;
    BWD_ARGS_OK(in_dev && out_dev && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, "sgclip_transpose_bf16");
    const dim3 grid(ceil_div((int)ld_out, 64), ceil_div(cols, 64));
    if (cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && !((uintptr_t)in_dev & 15) && !((uintptr_t)out_dev & 15)) {
        hipLaunchKernelGGL(transpose_bf16_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)in_dev,
                           (long long)ld_in, rows, cols, (bf16_t *)out_dev, (long long)ld_out);
        return launch_check("sgclip_transpose_bf16");
    }
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)in_dev,
                       (long long)ld_in, rows, cols, (bf16_t *)out_dev, (long long)ld_out);
    return launch_check("sgclip_transpose_bf16");
}

extern "C" int sgclip_colsum(const void *in_dev, int in_dtype, int64_t ld, int M, int N, float *out_dev, void *stream) {
    if (M == 0 || N == 0) return SGCLIP_OK;
    BWD_ARGS_OK(in_dev && out_dev && M > 0 && N > 0 && ld >= N, "sgclip_colsum");
    const int rpb = std::max(64, ceil_div(M, 512));
    const dim3 grid(ceil_div(N, 64), ceil_div(M, rpb));
    if (in_dtype == SGCLIP_DTYPE_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)in_dev,
                           (long long)ld, M, N, out_dev, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)in_dev,
                           (long long)ld, M, N, out_dev, rpb);
    return launch_check("sgclip_colsum");
}

extern "C" int sgclip_quick_gelu(const void *u_dev, void *g_dev, int64_t n, void *stream) {
    if (n == 0) return SGCLIP_OK;
    BWD_ARGS_OK(u_dev && g_dev && n > 0, "sgclip_quick_gelu");
    BWD_ARGS_OK(!((uintptr_t)u_dev & 15) && !((uintptr_t)g_dev & 15), "sgclip_quick_gelu (16-byte alignment)");
    hipLaunchKernelGGL(quick_gelu_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)u_dev, (bf16_t *)g_dev, (long long)n);
    return launch_check("sgclip_quick_gelu");
}

extern "C" int sgclip_quick_gelu_bwd(const void *u_dev, const void *dg_dev, void *du_dev, int64_t n, void *stream) {
    if (n == 0) return SGCLIP_OK;
    BWD_ARGS_OK(u_dev && dg_dev && du_dev && n > 0, "sgclip_quick_gelu_bwd");
    BWD_ARGS_OK(!((uintptr_t)u_dev & 15) && !((uintptr_t)dg_dev & 15) && !((uintptr_t)du_dev & 15),
                "sgclip_quick_gelu_bwd (16-byte alignment)");
    hipLaunchKernelGGL(quick_gelu_bwd_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)u_dev, (const bf16_t *)dg_dev, (bf16_t *)du_dev, (long long)n);
    return launch_check("sgclip_quick_gelu_bwd");
}

extern "C" int sgclip_cast_f32_bf16(const float *in_dev, void *out_dev, int64_t n, void *stream) {
    if (n == 0) return SGCLIP_OK;
    BWD_ARGS_OK(in_dev && out_dev && n > 0 && !((uintptr_t)in_dev & 15) && !((uintptr_t)out_dev & 15),
                "sgclip_cast_f32_bf16 (16-byte alignment)");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, in_dev,
                       (bf16_t *)out_dev, (long long)n);
    return launch_check("sgclip_cast_f32_bf16");
}

extern "C" int sgclip_cast_colsum_f32_bf16(const float *in_dev, void *out_dev, int M, int N, float *colsum_dev, void *stream) {
    if (M == 0 || N == 0) return SGCLIP_OK;
    BWD_ARGS_OK(in_dev && out_dev && colsum_dev && M > 0 && N > 0 && N % 8 == 0 && N / 8 <= 256 && !((uintptr_t)in_dev & 15) &&
                    !((uintptr_t)out_dev & 15),
                "sgclip_cast_colsum_f32_bf16 (N a multiple of 8, at most 2048; 16-byte alignment)");
    const int rpb = std::max(16, ceil_div(M, 1024));   // every block ends in N atomics on the same addresses
    hipLaunchKernelGGL(cast_colsum_kernel, dim3(ceil_div(M, rpb)), dim3(256), 0, (hipStream_t)stream, in_dev, (bf16_t *)out_dev,
                       M, N, colsum_dev, rpb);
    return launch_check("sgclip_cast_colsum_f32_bf16");
}

extern "C" int sgclip_layernorm_bwd(const float *x_dev, int64_t x_stride, int rows, int d, const float *w_dev, float eps,
                                    const void *dy_dev, int dy_dtype, float *dx_dev, int accumulate_dx, float *dw_dev,
                                    float *db_dev, void *stream) {
    if (rows == 0) return SGCLIP_OK;
    BWD_ARGS_OK(x_dev && w_dev && dy_dev && dx_dev && dw_dev && db_dev && rows > 0 && d > 0 && x_stride >= d,
                "sgclip_layernorm_bwd");
    if (d > 1024 || (d & 3) || (x_stride & 3) || ((uintptr_t)x_dev & 15) || ((uintptr_t)dx_dev & 15) || ((uintptr_t)w_dev & 15) ||
        ((uintptr_t)dy_dev & 15)) {
        set_error("sgclip_layernorm_bwd: d=%d must be a multiple of 4, <= 1024, rows 16-byte aligned", d);
        return SGCLIP_EINVAL;
    }
    const int rpw = std::max(1, std::min(LN_RPW_MAX, rows / (4 * 2048)));
    const dim3 grid(ceil_div(rows, 4 * rpw));
    const bool bf = dy_dtype == SGCLIP_DTYPE_BF16;
#define LN_BWD_LAUNCH(NV)                                                                                                  \
    do {                                                                                                                   \
        if (bf)                                                                                                            \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, true>), grid, dim3(256), 0, (hipStream_t)stream, x_dev,          \
                               (long long)x_stride, rows, d, w_dev, eps, dy_dev, dx_dev, accumulate_dx, dw_dev, db_dev,   \
                               rpw);                                                                                       \
        else                                                                                                               \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, false>), grid, dim3(256), 0, (hipStream_t)stream, x_dev,         \
                               (long long)x_stride, rows, d, w_dev, eps, dy_dev, dx_dev, accumulate_dx, dw_dev, db_dev,   \
                               rpw);                                                                                       \
    } while (0)
    if (d <= 256)
        LN_BWD_LAUNCH(1);
    else if (d <= 512)
        LN_BWD_LAUNCH(2);
    else if (d <= 768)
        LN_BWD_LAUNCH(3);
    else
        LN_BWD_LAUNCH(4);
#undef LN_BWD_LAUNCH
    return launch_check("sgclip_layernorm_bwd");
}

static int attention_bwd_impl(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, int n_seq, int T, int heads,
                              int causal, const int32_t *key_mask_dev, float *dbias_dev, void *stream);

extern "C" int sgclip_attention_bwd(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, int n_seq, int T, int heads,
                                    int causal, const int32_t *key_mask_dev, void *stream) {
    return attention_bwd_impl(qkv_dev, dout_dev, dqkv_dev, n_seq, T, heads, causal, key_mask_dev, nullptr, stream);
}

extern "C" int sgclip_attention_bwd_bias(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, float *dbias_dev, int n_seq,
                                         int T, int heads, int causal, const int32_t *key_mask_dev, void *stream) {
    if (!dbias_dev) {
        set_error("sgclip_attention_bwd_bias: dbias_dev is null");
        return SGCLIP_EINVAL;
    }
    return attention_bwd_impl(qkv_dev, dout_dev, dqkv_dev, n_seq, T, heads, causal, key_mask_dev, dbias_dev, stream);
}

// dbias_dev (or null): f32 [3d] += column sums of dqkv.  The matrix-core kernel of short sequences adds them itself; behind
// the other kernels the column-sum kernel runs over their result.
static int attention_bwd_impl(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, int n_seq, int T, int heads,
                              int causal, const int32_t *key_mask_dev, float *dbias_dev, void *stream) {
    if (n_seq == 0) return SGCLIP_OK;
    BWD_ARGS_OK(qkv_dev && dout_dev && dqkv_dev && n_seq > 0 && T > 0 && heads > 0, "sgclip_attention_bwd");
    auto then_colsum = [&](int rc) {
        if (rc != SGCLIP_OK || !dbias_dev) return rc;
        return sgclip_colsum(dqkv_dev, SGCLIP_DTYPE_BF16, (int64_t)3 * heads * 64, n_seq * T, 3 * heads * 64, dbias_dev, stream);
    };
    if (T > 128) {
        // long sequences: blocked kernel with recomputed statistics
        const int TP = (T + LB - 1) / LB * LB;
        const size_t lds = (size_t)9 * LBLK * sizeof(bf16_t) + (size_t)3 * TP * sizeof(float);
        if (lds > 160 * 1024 || ((uintptr_t)qkv_dev & 15) || ((uintptr_t)dout_dev & 15)) {
            set_error("sgclip_attention_bwd: T=%d too long for the per-row statistics to fit the LDS, or operands not "
                      "16-byte aligned", T);
            return SGCLIP_EINVAL;
        }
        if (!lds_attr((const void *)attention_bwd_long_kernel, 160 * 1024)) return SGCLIP_ELAUNCH;
        hipLaunchKernelGGL(attention_bwd_long_kernel, dim3((unsigned)(n_seq * heads)), dim3(256), lds, (hipStream_t)stream,
                           (const bf16_t *)qkv_dev, (const bf16_t *)dout_dev, (bf16_t *)dqkv_dev, T, heads, causal, key_mask_dev);
        return then_colsum(launch_check("sgclip_attention_bwd"));
    }
    static int use_valu = -1;   // SGCLIP_ATTN_BWD_VALU=1: the plain VALU kernel (cross-check)
    if (use_valu < 0) {
        const char *e = getenv("SGCLIP_ATTN_BWD_VALU");
        use_valu = e ? atoi(e) : 0;
    }
    if (!use_valu && !((uintptr_t)qkv_dev & 15) && !((uintptr_t)dout_dev & 15)) {
        const bf16_t *q = (const bf16_t *)qkv_dev, *o = (const bf16_t *)dout_dev;
        bf16_t *g = (bf16_t *)dqkv_dev;
        hipStream_t hs = (hipStream_t)stream;
        if (T <= 32)
            attention_bwd_mfma_launch<32>(q, o, g, n_seq, T, heads, causal, key_mask_dev, dbias_dev, hs);
        else if (T <= 64)
            attention_bwd_mfma_launch<64>(q, o, g, n_seq, T, heads, causal, key_mask_dev, dbias_dev, hs);
        else if (T <= 96)
            attention_bwd_mfma_launch<96>(q, o, g, n_seq, T, heads, causal, key_mask_dev, dbias_dev, hs);
        else
            attention_bwd_mfma_launch<128>(q, o, g, n_seq, T, heads, causal, key_mask_dev, dbias_dev, hs);
        return launch_check("sgclip_attention_bwd");
    }
    const size_t lds = (size_t)4 * T * 64 * sizeof(bf16_t) + (size_t)T * T * sizeof(float);
    if (!lds_attr((const void *)attention_bwd_kernel, 160 * 1024)) return SGCLIP_ELAUNCH;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3((unsigned)(n_seq * heads)), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t *)qkv_dev, (const bf16_t *)dout_dev, (bf16_t *)dqkv_dev, T, heads, causal, key_mask_dev);
    return then_colsum(launch_check("sgclip_attention_bwd"));
}

extern "C" int sgclip_matmul_f32(const float *A_dev, int64_t lda, int transA, const float *B_dev, int64_t ldb, int transB,
                                 float *C_dev, int64_t ldc, int M, int N, int K, float alpha, int accumulate, void *stream) {
    if (M == 0 || N == 0) return SGCLIP_OK;
    BWD_ARGS_OK(A_dev && B_dev && C_dev && M > 0 && N > 0 && K >= 0 && ldc >= N, "sgclip_matmul_f32");
    hipLaunchKernelGGL(matmul_f32_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, (hipStream_t)stream, A_dev,
                       (long long)lda, transA, B_dev, (long long)ldb, transB, C_dev, (long long)ldc, M, N, K, alpha, accumulate);
    return launch_check("sgclip_matmul_f32");
}

extern "C" int sgclip_dot_f32(const float *a_dev, const float *b_dev, int64_t n, float *out_dev, void *stream) {
    if (n == 0) return SGCLIP_OK;
    BWD_ARGS_OK(a_dev && b_dev && out_dev && n > 0, "sgclip_dot_f32");
    const unsigned blocks = (unsigned)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(dot_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a_dev, b_dev, (long long)n, out_dev);
    return launch_check("sgclip_dot_f32");
}

extern "C" int sgclip_prob_bwd(const float *probs_dev, const float *dprobs_dev, int K, int N, int mode, float *dlogits_dev,
                               void *stream) {
    if (K == 0 || N == 0) return SGCLIP_OK;
    BWD_ARGS_OK(probs_dev && dprobs_dev && dlogits_dev && K > 0 && N > 0, "sgclip_prob_bwd");
    hipLaunchKernelGGL(prob_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, probs_dev, dprobs_dev, K, N,
                       mode, dlogits_dev);
    return launch_check("sgclip_prob_bwd");
}

extern "C" int sgclip_normalize_bwd(const float *x_dev, const float *dxhat_dev, int rows, int E, float *dx_dev, void *stream) {
    if (rows == 0) return SGCLIP_OK;
    BWD_ARGS_OK(x_dev && dxhat_dev && dx_dev && rows > 0 && E > 0, "sgclip_normalize_bwd");
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x_dev, dxhat_dev, rows,
                       E, dx_dev);
    return launch_check("sgclip_normalize_bwd");
}

extern "C" int sgclip_normalize_rows(const float *x_dev, int rows, int E, float *xhat_dev, void *stream) {
    if (rows == 0) return SGCLIP_OK;
    BWD_ARGS_OK(x_dev && xhat_dev && rows > 0 && E > 0, "sgclip_normalize_rows");
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x_dev, rows, E, xhat_dev);
    return launch_check("sgclip_normalize_rows");
}

extern "C" int sgclip_segment_mean_bwd(const float *dmean_dev, const int32_t *seg_ids_dev, const float *inv_count_dev, int K,
                                       int N, int G, float *dprobs_dev, void *stream) {
    if (K == 0 || N == 0) return SGCLIP_OK;
    BWD_ARGS_OK(dmean_dev && seg_ids_dev && inv_count_dev && dprobs_dev && K > 0 && N > 0 && G > 0, "sgclip_segment_mean_bwd");
    hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, dmean_dev, seg_ids_dev,
                       inv_count_dev, K, N, G, dprobs_dev);
    return launch_check("sgclip_segment_mean_bwd");
}

extern "C" int sgclip_embed_text(const float *table_dev, const float *pos_dev, const int32_t *ids_dev, int n_texts, int T, int d,
                                 float *x_dev, void *stream) {
    if (n_texts == 0) return SGCLIP_OK;
    BWD_ARGS_OK(table_dev && pos_dev && ids_dev && x_dev && n_texts > 0 && T > 0 && d > 0, "sgclip_embed_text");
    hipLaunchKernelGGL(embed_text_kernel, dim3((unsigned)(n_texts * T)), dim3(128), 0, (hipStream_t)stream, table_dev, pos_dev,
                       ids_dev, T, d, x_dev);
    return launch_check("sgclip_embed_text");
}

extern "C" int sgclip_embed_cls(const float *cls_dev, const float *pos_dev, int n_images, int T, int d, float *x_dev,
                                void *stream) {
    if (n_images == 0) return SGCLIP_OK;
    BWD_ARGS_OK(cls_dev && pos_dev && x_dev && n_images > 0 && T > 0 && d > 0, "sgclip_embed_cls");
    hipLaunchKernelGGL(embed_cls_kernel, dim3((unsigned)n_images), dim3(128), 0, (hipStream_t)stream, cls_dev, pos_dev, T, d, x_dev);
    return launch_check("sgclip_embed_cls");
}

extern "C" int sgclip_rows_scatter_add(const float *src_dev, const int32_t *idx_dev, int rows, int d, float *dst_dev,
                                       void *stream) {
    if (rows == 0) return SGCLIP_OK;
    BWD_ARGS_OK(src_dev && idx_dev && dst_dev && rows > 0 && d > 0, "sgclip_rows_scatter_add");
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((unsigned)rows), dim3(128), 0, (hipStream_t)stream, src_dev, idx_dev, d,
                       dst_dev);
    return launch_check("sgclip_rows_scatter_add");
}

extern "C" int sgclip_adamw(float *param_dev, const float *grad_dev, float *m_dev, float *v_dev, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, void *stream) {
    if (n == 0) return SGCLIP_OK;
    BWD_ARGS_OK(param_dev && grad_dev && m_dev && v_dev && n > 0 && step >= 1, "sgclip_adamw");
    const float c1 = 1.0f - powf(beta1, (float)step), c2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param_dev, grad_dev,
                       m_dev, v_dev, (long long)n, lr, beta1, beta2, eps, weight_decay, c1, c2);
    return launch_check("sgclip_adamw");
}
