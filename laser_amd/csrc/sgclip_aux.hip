// Data formats either side of the scoring path.
//
//  * topk_columns_kernel: the k best keywords of every crop straight from the dense
//    [keywords, crops] probabilities, in the order Python's sorted((p, name), reverse=True)
//    gives (laser scripts/eval_sgclip.py:132-133, 170-171), so the evaluation metrics never
//    materialise per-entry dictionaries.
//  * mask_unpack_kernel: bit-packed segmentation masks -> the byte-per-pixel layout the crop
//    kernel reads, plus every mask's tight box and pixel count in the same pass
//    (laser/loading.py:84-104 bitmasks2bboxes; laser/training/llava_dataset.py:127,135
//    empty / minimum-size filters).
#include "common.h"

#include <limits.h>

#include <algorithm>

namespace sgclip {

constexpr int TOPK_MAX = 16;

// thread = column; probs is row-major, so a wave reads 64 consecutive floats per row
__global__ __launch_bounds__(256) void topk_columns_kernel(const float *__restrict__ probs, int64_t ld, int K, int N,
                                                           const int32_t *__restrict__ tie_rank, int k,
                                                           int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float bv[TOPK_MAX];
    int bi[TOPK_MAX], br[TOPK_MAX];
#pragma unroll
    for (int j = 0; j < TOPK_MAX; j++) {
        bv[j] = -INFINITY;
        bi[j] = -1;
        br[j] = INT_MIN;
    }
    for (int r = 0; r < K; r++) {
        float v = probs[(int64_t)r * ld + n];
        if (v != v) continue;  // NaN never ranks
        int idx = r, rk = tie_rank ? tie_rank[r] : -r;
        // insertion into the descending list: (value, tie rank) compared lexicographically
#pragma unroll
        for (int j = 0; j < TOPK_MAX; j++) {
            if (j < k) {
                const bool better = v > bv[j] || (v == bv[j] && rk > br[j]);
                if (better) {
                    const float tv = bv[j];
                    const int ti = bi[j], tr = br[j];
                    bv[j] = v, bi[j] = idx, br[j] = rk;
                    v = tv, idx = ti, rk = tr;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TOPK_MAX; j++) {
        if (j < k) {
            out_idx[(int64_t)j * N + n] = bi[j];
            out_val[(int64_t)j * N + n] = bi[j] < 0 ? 0.f : bv[j];
        }
    }
}

struct MaskRec {  // mirrors sgclip_mask_t
    int64_t packed_off, out_off;
    int32_t H, W;
};

__global__ void mask_box_init_kernel(int32_t *boxes, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        boxes[i * 5 + 0] = INT_MAX;
        boxes[i * 5 + 1] = INT_MAX;
        boxes[i * 5 + 2] = -1;
        boxes[i * 5 + 3] = -1;
        boxes[i * 5 + 4] = 0;
    }
}

constexpr int MASK_ROWS = 16;  // rows per workgroup

// grid = (row blocks, masks); thread = one packed byte of a row = 8 pixels, most significant bit first
__global__ __launch_bounds__(256) void mask_unpack_kernel(const uint8_t *__restrict__ packed,
                                                          const MaskRec *__restrict__ recs, uint8_t *__restrict__ out,
                                                          int32_t *__restrict__ boxes) {
    const MaskRec rec = recs[blockIdx.y];
    const int H = rec.H, W = rec.W, rb = (W + 7) >> 3;
    const int r0 = blockIdx.x * MASK_ROWS;
    if (r0 >= H) return;
    const int r1 = min(H, r0 + MASK_ROWS);
    const uint8_t *src = packed + rec.packed_off;
    uint8_t *dst = out + rec.out_off;
    const bool wide = ((W & 7) == 0) && (((uintptr_t)dst & 7) == 0);
    int xmin = INT_MAX, ymin = INT_MAX, xmax = -1, ymax = -1, cnt = 0;
    for (int y = r0; y < r1; y++) {
        for (int t = threadIdx.x; t < rb; t += blockDim.x) {
            unsigned b = src[(size_t)y * rb + t];
            const int x0 = t * 8, valid = min(8, W - x0);
            b &= (0xFFu << (8 - valid)) & 0xFFu;  // bits past the row end are padding
            if (wide) {
                // byte j of the 8-byte store is pixel x0 + j = bit 7 - j
                uint2 o;
                o.x = ((b >> 7) & 1u) | (((b >> 6) & 1u) << 8) | (((b >> 5) & 1u) << 16) | (((b >> 4) & 1u) << 24);
                o.y = ((b >> 3) & 1u) | (((b >> 2) & 1u) << 8) | (((b >> 1) & 1u) << 16) | ((b & 1u) << 24);
                *(uint2 *)(dst + (size_t)y * W + x0) = o;
            } else {
                for (int j = 0; j < valid; j++) dst[(size_t)y * W + x0 + j] = (b >> (7 - j)) & 1u;
            }
            if (b) {
                xmin = min(xmin, x0 + (__clz((int)b) - 24));
                xmax = max(xmax, x0 + 7 - (__ffs((int)b) - 1));
                ymin = min(ymin, y);
                ymax = max(ymax, y);
                cnt += __popc(b);
            }
        }
    }
    // wave reduction, then one set of atomics per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = min(xmin, __shfl_xor(xmin, o));
        ymin = min(ymin, __shfl_xor(ymin, o));
        xmax = max(xmax, __shfl_xor(xmax, o));
        ymax = max(ymax, __shfl_xor(ymax, o));
        cnt += __shfl_xor(cnt, o);
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0) {
        int32_t *bx = boxes + (size_t)blockIdx.y * 5;
        atomicMin(bx + 0, xmin);
        atomicMin(bx + 1, ymin);
        atomicMax(bx + 2, xmax);
        atomicMax(bx + 3, ymax);
        atomicAdd(bx + 4, cnt);
    }
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_topk_columns(const float *probs_dev, int64_t ld, int K, int N, const int32_t *tie_rank_dev, int k,
                                   int32_t *out_idx_dev, float *out_val_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0 || k == 0) return SGCLIP_OK;
    if (!probs_dev || !out_idx_dev || !out_val_dev || K < 0 || N < 0 || ld < N) {
        set_error("sgclip_topk_columns: bad arguments");
        return SGCLIP_EINVAL;
    }
    if (k < 0 || k > TOPK_MAX) {
        set_error("sgclip_topk_columns: k=%d outside 0..%d", k, TOPK_MAX);
        return SGCLIP_EINVAL;
    }
    hipLaunchKernelGGL(topk_columns_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, probs_dev, ld, K, N,
                       tie_rank_dev, k, out_idx_dev, out_val_dev);
    return launch_check("sgclip_topk_columns");
}

extern "C" int sgclip_mask_unpack(const uint8_t *packed_dev, size_t packed_bytes, const sgclip_mask_t *masks,
                                  sgclip_mask_t *masks_dev, int n_masks, uint8_t *out_dev, size_t out_bytes,
                                  int32_t *boxes_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    static_assert(sizeof(MaskRec) == sizeof(sgclip_mask_t), "mask record layout");
    if (n_masks == 0) return SGCLIP_OK;
    if (!packed_dev || !masks || !masks_dev || !out_dev || !boxes_dev || n_masks < 0) {
        set_error("sgclip_mask_unpack: null argument");
        return SGCLIP_EINVAL;
    }
    int max_h = 0;
    for (int i = 0; i < n_masks; i++) {
        const sgclip_mask_t &m = masks[i];
        if (m.H <= 0 || m.W <= 0 || m.packed_off < 0 || m.out_off < 0 ||
            (size_t)m.packed_off + (size_t)m.H * ((m.W + 7) / 8) > packed_bytes ||
            (size_t)m.out_off + (size_t)m.H * m.W > out_bytes) {
            set_error("sgclip_mask_unpack: mask %d (%dx%d) outside its buffer", i, m.H, m.W);
            return SGCLIP_EINVAL;
        }
        max_h = std::max(max_h, m.H);
    }
    hipError_t e = hipMemcpyAsync(masks_dev, masks, (size_t)n_masks * sizeof(sgclip_mask_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
        set_error("sgclip_mask_unpack: table upload failed: %s", hipGetErrorString(e));
        return SGCLIP_ELAUNCH;
    }
    hipLaunchKernelGGL(mask_box_init_kernel, dim3(ceil_div(n_masks, 256)), dim3(256), 0, stream, boxes_dev, n_masks);
    for (int first = 0; first < n_masks; first += 32768) {  // grid.y limit
        const int n = std::min(32768, n_masks - first);
        hipLaunchKernelGGL(mask_unpack_kernel, dim3(ceil_div(max_h, MASK_ROWS), n), dim3(256), 0, stream, packed_dev,
                           (const MaskRec *)masks_dev + first, out_dev, boxes_dev + (size_t)first * 5);
    }
    return launch_check("sgclip_mask_unpack");
}
