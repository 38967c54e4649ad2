// fp32 forms of the encoder-layer building blocks: a linear layer on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) and an fp32 attention.
//
// The training-mode pair selection of the reference ranks objects by their per-object mean
// category probability and keeps the best `topk_cate` per name
// (laser/models/llava_clip_model_v3.py:303-314, 347-386); which pairs are scored hangs on
// that ranking.  Where the bf16 towers leave the boundary of the ranking closer than their
// own error, the few objects concerned are scored again through these kernels, in the
// reference's fp32 arithmetic (transformers models/clip/modeling_clip.py:269-277, 298-350).
// Rows are few (tens of crops), so the kernels are simple: no LDS staging of the operands.
#include "common.h"

namespace sgclip {

// y = epi(x W^T + bias): one wave per 32 x 32 tile.  Lane (r, h) = (lane & 31, lane >> 5) feeds
// x[m0 + r][k] and W[n0 + r][k] for the k of its half: within every group of 8 reduction indices
// half h takes 4 h .. 4 h + 3 (any assignment of the indices to the two halves gives the same sum
// up to the order of the fp32 additions).
__global__ __launch_bounds__(64) void linear_exact_kernel(const float *__restrict__ X, long long ldx,
                                                          const float *__restrict__ W, long long ldw,
                                                          const float *__restrict__ bias, int rows, int out_f,
                                                          int in_f, int mode, const float *aux, int aux_rows,
                                                          int out_tokens, float *Y, long long ldy) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float *xr = X + (size_t)min(m0 + r, rows - 1) * ldx + 4 * h;
    const float *wr = W + (size_t)min(n0 + r, out_f - 1) * ldw + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; g++) acc[g] = 0.f;
    for (int k = 0; k < in_f; k += 8) {
        const float4 a = *(const float4 *)(xr + k), b = *(const float4 *)(wr + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    const int n = n0 + r;
    if (n >= out_f) return;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < 16; g++) {
        const int m = m0 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (m >= rows) continue;
        float v = acc[g] + bn;
        size_t orow = (size_t)m;
        if (mode == SGCLIP_F32_GELU) {
            v = v * (1.0f / (1.0f + expf(-1.702f * v)));
        } else if (mode == SGCLIP_F32_RESID) {
            v = aux[(size_t)m * ldy + n] + v;
        } else if (mode == SGCLIP_F32_PATCH) {
            const int img = m / aux_rows, pp = m - img * aux_rows;
            orow = (size_t)img * out_tokens + 1 + pp;
            v = v + aux[(size_t)(1 + pp) * ldy + n];
        }
        Y[orow * ldy + n] = v;
    }
}

// softmax(q k^T / 8 + mask) v per (sequence, head), fp32 throughout.  A thread owns one query row
// (its 64 query values and 64 output sums in registers); keys and values pass through LDS 64 rows at
// a time and are read by all threads at the same address (a broadcast).  Running maximum and sum
// per query (the result equals the two-pass softmax up to fp32 rounding).
constexpr int AF_KEYS = 64;

__global__ __launch_bounds__(128) void attention_f32_kernel(const float *__restrict__ qkv, float *__restrict__ out,
                                                            int T, int heads, int causal,
                                                            const int32_t *__restrict__ key_mask) {
    __shared__ float ks[AF_KEYS][64], vs[AF_KEYS][64];
    __shared__ int ok[AF_KEYS];
    const int seq = blockIdx.z, head = blockIdx.y;
    const int d = heads * 64;
    const size_t ld = (size_t)3 * d;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < T;
    const float *base = qkv + (size_t)seq * T * ld + (size_t)head * 64;
    float q[64], o[64];
#pragma unroll
    for (int c = 0; c < 64; c++) {
        q[c] = live ? base[(size_t)t * ld + c] * 0.125f : 0.f;
        o[c] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f;
    const int t_hi = min(T, (int)(blockIdx.x + 1) * (int)blockDim.x);   // causal: keys beyond the block's last query are dead
    const int k_end = causal ? t_hi : T;
    for (int k0 = 0; k0 < k_end; k0 += AF_KEYS) {
        __syncthreads();
        for (int i = threadIdx.x; i < AF_KEYS * 64; i += blockDim.x) {
            const int kr = i >> 6, c = i & 63;
            const int kk = k0 + kr;
            ks[kr][c] = kk < T ? base[(size_t)kk * ld + d + c] : 0.f;
            vs[kr][c] = kk < T ? base[(size_t)kk * ld + 2 * d + c] : 0.f;
        }
        for (int i = threadIdx.x; i < AF_KEYS; i += blockDim.x) {
            const int kk = k0 + i;
            ok[i] = kk < T && (!key_mask || key_mask[(size_t)seq * T + kk] != 0);
        }
        __syncthreads();
        const int nk = min(AF_KEYS, k_end - k0);
        for (int kr = 0; kr < nk; kr++) {
            if (!ok[kr] || (causal && k0 + kr > t)) continue;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 64; c++) s += q[c] * ks[kr][c];
            const float nm = fmaxf(mx, s);
            const float scale = expf(mx - nm), p = expf(s - nm);
            sum = sum * scale + p;
#pragma unroll
            for (int c = 0; c < 64; c++) o[c] = o[c] * scale + p * vs[kr][c];
            mx = nm;
        }
    }
    if (!live) return;
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float *orow = out + ((size_t)seq * T + t) * d + (size_t)head * 64;
#pragma unroll
    for (int c = 0; c < 64; c++) orow[c] = o[c] * inv;
}

// x[k][n] <- softmax over k of scale * x[k][n], one thread per column (K is a vocabulary: tens to hundreds)
__global__ void scale_softmax_columns_kernel(float *__restrict__ x, int K, int N, float scale, int do_softmax) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float mx = -INFINITY;
    for (int k = 0; k < K; k++) {
        const float v = x[(size_t)k * N + n] * scale;
        x[(size_t)k * N + n] = v;
        mx = fmaxf(mx, v);
    }
    if (!do_softmax) return;
    float s = 0.f;
    for (int k = 0; k < K; k++) s += expf(x[(size_t)k * N + n] - mx);
    for (int k = 0; k < K; k++) x[(size_t)k * N + n] = expf(x[(size_t)k * N + n] - mx) / s;
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_scale_softmax_columns(float *x_dev, int K, int N, float scale, int do_softmax, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K == 0 || N == 0) return SGCLIP_OK;
    if (!x_dev || K < 0 || N < 0) {
        set_error("sgclip_scale_softmax_columns: bad arguments K=%d N=%d", K, N);
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(scale_softmax_columns_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, stream, x_dev, K, N, scale,
                       do_softmax);
    return launch_check("sgclip_scale_softmax_columns");
}

extern "C" int sgclip_linear_exact_f32(const float *x_dev, int64_t ldx, const float *w_dev, int64_t ldw,
                                       const float *bias_dev, int rows, int out_features, int in_features, int mode,
                                       const float *aux_dev, int aux_rows, int out_tokens, float *y_dev, int64_t ldy,
                                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (rows == 0) return SGCLIP_OK;
    const bool need_aux = mode == SGCLIP_F32_RESID || mode == SGCLIP_F32_PATCH;
    if (!x_dev || !w_dev || !y_dev || rows < 0 || out_features <= 0 || in_features <= 0 || in_features % 8 != 0 ||
        (ldx & 3) || (ldw & 3) || ((uintptr_t)x_dev & 15) || ((uintptr_t)w_dev & 15) || mode < 0 ||
        mode > SGCLIP_F32_PATCH || (need_aux && !aux_dev) ||
        (mode == SGCLIP_F32_PATCH && (aux_rows <= 0 || out_tokens < aux_rows + 1))) {
        set_error("sgclip_linear_exact_f32: bad arguments rows=%d out=%d in=%d mode=%d (in must be a multiple of 8, "
                  "operand rows 16-byte aligned)", rows, out_features, in_features, mode);
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(linear_exact_kernel, dim3(ceil_div(out_features, 32), ceil_div(rows, 32)), dim3(64), 0, stream,
                       x_dev, (long long)ldx, w_dev, (long long)ldw, bias_dev, rows, out_features, in_features, mode,
                       aux_dev, aux_rows, out_tokens, y_dev, (long long)ldy);
    return launch_check("sgclip_linear_exact_f32");
}

extern "C" int sgclip_attention_f32(const float *qkv_dev, float *out_dev, int n_seq, int T, int heads, int causal,
                                    const int32_t *key_mask_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seq == 0) return SGCLIP_OK;
    if (!qkv_dev || !out_dev || n_seq < 0 || T <= 0 || heads <= 0 || heads > 65535 || n_seq > 65535) {
        set_error("sgclip_attention_f32: bad arguments n_seq=%d T=%d heads=%d", n_seq, T, heads);
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(attention_f32_kernel, dim3(ceil_div(T, 128), heads, n_seq), dim3(128), 0, stream, qkv_dev, out_dev,
                       T, heads, causal, key_mask_dev);
    return launch_check("sgclip_attention_f32");
}
