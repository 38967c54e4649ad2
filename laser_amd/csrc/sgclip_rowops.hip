// Row-wise kernels of the SGCLIP path: LayerNorm, row norms,
// the text x image similarity contraction (bf16 MFMA with a hi/lo operand split
// for fp32-grade accuracy), softmax / sigmoid over the predicate vocabulary with
// wavefront reductions, and the per-object mean of category probabilities.
#include "common.h"

#include <mutex>

namespace sgclip {

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row, fp32 statistics (mean, then centred variance).
// ---------------------------------------------------------------------------
template <int NV, bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, long long x_stride,
                                                        const int32_t *__restrict__ row_idx, int rows, int d,
                                                        const float *__restrict__ w, const float *__restrict__ b,
                                                        float eps, void *y) {  // y may alias x (in-place rows)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long src_row = row_idx ? row_idx[row] : row;
    const float4 *xr = (const float4 *)(x + src_row * x_stride);
    const int nvec = d >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = lane + 64 * i;
        v[i] = idx < nvec ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            const float4 ww = ((const float4 *)w)[idx], bb = ((const float4 *)b)[idx];
            const float o0 = (v[i].x - mean) * rstd * ww.x + bb.x;
            const float o1 = (v[i].y - mean) * rstd * ww.y + bb.y;
            const float o2 = (v[i].z - mean) * rstd * ww.z + bb.z;
            const float o3 = (v[i].w - mean) * rstd * ww.w + bb.w;
            if (OUT_BF16) {
                bf16x4 o;
                o[0] = (bf16_t)o0; o[1] = (bf16_t)o1; o[2] = (bf16_t)o2; o[3] = (bf16_t)o3;
                ((bf16x4 *)((bf16_t *)y + (size_t)row * d))[idx] = o;
            } else {
                ((float4 *)((float *)y + (size_t)row * d))[idx] = make_float4(o0, o1, o2, o3);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// similarity: inverse L2 norms, then logits[k][n] = scale * inv_k * inv_n * <t_k, i_n>
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inv_norm_kernel(const float *__restrict__ x, int rows, int E, float *__restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * E;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += xr[i] * xr[i];
    s = wave_sum(s);
    if (lane == 0) inv[row] = 1.0f / sqrtf(s);
}

__device__ __forceinline__ void split_bf16(const float *p, bf16x8 &hi, bf16x8 &lo) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const bf16_t h = (bf16_t)v[j];
        hi[j] = h;
        lo[j] = (bf16_t)(v[j] - (float)h);
    }
}

// out[a][b] = scale * sa[a] * sb[b] * <A_a, B_b> for fp32 rows A [Ma, E], B [Nb, E]:
// one wave per 32 x 32 tile, every fp32 operand split into bf16 hi + lo and the
// product taken as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (error ~2^-16
// relative, i.e. fp32-grade for cosine logits).  sa / sb may be null (= 1).
__global__ __launch_bounds__(64) void nt_split_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                      const float *__restrict__ sa, const float *__restrict__ sb, int Ma,
                                                      int Nb, int E, float scale, float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const float *ar = A + (size_t)min(a0 + r, Ma - 1) * E + 8 * h;
    const float *br = B + (size_t)min(b0 + r, Nb - 1) * E + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; g++) acc[g] = 0.f;
    for (int e = 0; e < E; e += 16) {
        bf16x8 ah, al, bh, bl;
        split_bf16(ar + e, ah, al);
        split_bf16(br + e, bh, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    const int b = b0 + r;
    if (b >= Nb) return;
    const float sn = (sb ? sb[b] : 1.0f) * scale;
#pragma unroll
    for (int g = 0; g < 16; g++) {
        const int a = a0 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (a < Ma) out[(size_t)a * ldo + b] = acc[g] * (sa ? sa[a] : 1.0f) * sn;
    }
}

// softmax over k (rows) for every column n.  A wave owns 16 columns: lane = (kq, c),
// kq = lane>>4 strides the rows, c = lane&15 the column; the 4 row phases are combined
// with two cross-lane exchanges (xor 16, xor 32).
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ x, int K, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int n = (blockIdx.x * 4 + wave) * 16 + c;
    const bool ok = n < N;
    float mx = -INFINITY;
    for (int k = kq; k < K; k += 4)
        if (ok) mx = fmaxf(mx, x[(size_t)k * N + n]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float s = 0.f;
    for (int k = kq; k < K; k += 4)
        if (ok) s += expf(x[(size_t)k * N + n] - mx);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.0f / s;
    for (int k = kq; k < K; k += 4)
        if (ok) x[(size_t)k * N + n] = expf(x[(size_t)k * N + n] - mx) * inv;
}

__global__ void sigmoid_kernel(float *x, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = 1.0f / (1.0f + expf(-x[i]));
}

// out[k][g] = mean over columns n with seg[n] == g, accumulated in column order
__global__ void segment_mean_kernel(const float *__restrict__ p, const int32_t *__restrict__ seg, int K, int N, int G,
                                    float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * G) return;
    const int k = i / G, g = i - k * G;
    float s = 0.f;
    int cnt = 0;
    for (int n = 0; n < N; n++)
        if (seg[n] == g) { s += p[(size_t)k * N + n]; cnt++; }
    out[i] = cnt ? s / cnt : 0.f;
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_layernorm(const float *x_dev, int64_t x_stride, const int32_t *row_idx_dev, int rows, int d,
                                const float *w_dev, const float *b_dev, float eps, int out_dtype, void *y_dev,
                                void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (rows == 0) return SGCLIP_OK;
    if (!x_dev || !w_dev || !b_dev || !y_dev || rows < 0 || d <= 0 || d % 4 != 0 || d > 1024 || (x_stride & 3)) {
        set_error("sgclip_layernorm: bad arguments rows=%d d=%d (d must be a multiple of 4, <= 1024)", rows, d);
        return SGCLIP_EINVAL;
    }
    const int nv = ceil_div(d >> 2, 64);
    const dim3 grid(ceil_div(rows, 4)), block(256);
    const bool bf = out_dtype == SGCLIP_DTYPE_BF16;
#define LN_LAUNCH(NVV)                                                                                              \
    if (bf)                                                                                                         \
        hipLaunchKernelGGL((layernorm_kernel<NVV, true>), grid, block, 0, stream, x_dev, (long long)x_stride,       \
                           row_idx_dev, rows, d, w_dev, b_dev, eps, y_dev);                                         \
    else                                                                                                            \
        hipLaunchKernelGGL((layernorm_kernel<NVV, false>), grid, block, 0, stream, x_dev, (long long)x_stride,      \
                           row_idx_dev, rows, d, w_dev, b_dev, eps, y_dev);
    switch (nv) {
        case 1: LN_LAUNCH(1) break;
        case 2: LN_LAUNCH(2) break;
        case 3: LN_LAUNCH(3) break;
        default: LN_LAUNCH(4) break;
    }
#undef LN_LAUNCH
    return launch_check("sgclip_layernorm");
}

extern "C" size_t sgclip_similarity_workspace_bytes(int K, int N) { return align_up((size_t)(K + N) * 4, 256); }

extern "C" int sgclip_similarity(const float *text_dev, const float *img_dev, int K, int N, int E,
                                 float exp_logit_scale, int mode, float *out_dev, void *workspace_dev,
                                 size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K == 0 || N == 0) return SGCLIP_OK;
    if (!text_dev || !img_dev || !out_dev || !workspace_dev || K < 0 || N < 0 || E <= 0 || E % 16 != 0) {
        set_error("sgclip_similarity: bad arguments K=%d N=%d E=%d (E must be a multiple of 16)", K, N, E);
        return SGCLIP_EINVAL;
    }
    if (workspace_bytes < sgclip_similarity_workspace_bytes(K, N)) {
        set_error("sgclip_similarity: workspace too small");
        return SGCLIP_ENOSPC;
    }
    float *inv_t = (float *)workspace_dev, *inv_i = inv_t + K;
    hipLaunchKernelGGL(inv_norm_kernel, dim3(ceil_div(K, 4)), dim3(256), 0, stream, text_dev, K, E, inv_t);
    hipLaunchKernelGGL(inv_norm_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, stream, img_dev, N, E, inv_i);
    hipLaunchKernelGGL(nt_split_kernel, dim3(ceil_div(N, 32), ceil_div(K, 32)), dim3(64), 0, stream, text_dev, img_dev,
                       inv_t, inv_i, K, N, E, exp_logit_scale, out_dev, N);
    if (mode == SGCLIP_SIM_SOFTMAX) {
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, stream, out_dev, K, N);
    } else if (mode == SGCLIP_SIM_SIGMOID) {
        const long long tot = (long long)K * N;
        hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, out_dev, tot);
    } else if (mode != SGCLIP_SIM_LOGIT) {
        set_error("sgclip_similarity: unknown mode %d", mode);
        return SGCLIP_EINVAL;
    }
    return launch_check("sgclip_similarity");
}

// ---------------------------------------------------------------------------
// sgclip_similarity without a caller workspace and without any allocation: the inverse row norms
// (the partial results sgclip_similarity keeps in its workspace) live in a file-scope device array.
// The array's name is a host-side handle, not a device address: its address on each GPU comes from
// hipGetSymbolAddress, asked once per device with that device current, and only that pointer is
// handed to kernels.
// ---------------------------------------------------------------------------
namespace sgclip {
constexpr int SIM_STATIC_MAX = 65536;   // K + N floats: 256 KiB on every GPU the library runs on
static __device__ __align__(256) float scratch[SIM_STATIC_MAX];
constexpr int MAX_DEVICES = 64;
static float *g_scratch_ptr[MAX_DEVICES];   // per device: each GPU holds its own copy of the array
static std::mutex g_scratch_mu;

// device address of `scratch` on the current device (nullptr + error text on failure)
static float *static_scratch() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        set_error("sgclip_similarity_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("sgclip_similarity_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = (float *)p;
    }
    return g_scratch_ptr[dev];
}
}  // namespace sgclip

extern "C" int sgclip_similarity_static_max(void) { return SIM_STATIC_MAX; }

extern "C" int sgclip_similarity_static(const float *text_dev, const float *img_dev, int K, int N, int E,
                                        float exp_logit_scale, int mode, float *out_dev, void *stream_) {
    if (K == 0 || N == 0) return SGCLIP_OK;
    if (K < 0 || N < 0 || (long long)K + N > SIM_STATIC_MAX) {
        set_error("sgclip_similarity_static: K + N = %lld exceeds the static scratch of %d floats; use "
                  "sgclip_similarity with a workspace", (long long)K + N, SIM_STATIC_MAX);
        return SGCLIP_ENOSPC;
    }
    float *ws = static_scratch();
    if (!ws) return SGCLIP_ELAUNCH;
    // every one of the K + N entries read is written first by this call: nothing to clear
    return sgclip_similarity(text_dev, img_dev, K, N, E, exp_logit_scale, mode, out_dev, ws,
                             sgclip_similarity_workspace_bytes(K, N), stream_);
}

extern "C" int sgclip_linear_f32(const float *x_dev, const float *w_dev, int rows, int out_features, int in_features,
                                 float *y_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (rows == 0) return SGCLIP_OK;
    if (!x_dev || !w_dev || !y_dev || rows < 0 || out_features <= 0 || in_features <= 0 || in_features % 16 != 0) {
        set_error("sgclip_linear_f32: bad arguments rows=%d out=%d in=%d (in must be a multiple of 16)", rows,
                  out_features, in_features);
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(nt_split_kernel, dim3(ceil_div(out_features, 32), ceil_div(rows, 32)), dim3(64), 0, stream, x_dev,
                       w_dev, (const float *)nullptr, (const float *)nullptr, rows, out_features, in_features, 1.0f, y_dev,
                       out_features);
    return launch_check("sgclip_linear_f32");
}

extern "C" int sgclip_segment_mean(const float *probs_dev, const int32_t *seg_ids_dev, int K, int N, int G,
                                   float *out_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K == 0 || G == 0) return SGCLIP_OK;
    if (!probs_dev || !seg_ids_dev || !out_dev || K < 0 || N < 0 || G < 0) {
        set_error("sgclip_segment_mean: bad arguments");
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(segment_mean_kernel, dim3(ceil_div(K * G, 128)), dim3(128), 0, stream, probs_dev, seg_ids_dev, K,
                       N, G, out_dev);
    return launch_check("sgclip_segment_mean");
}
