// Mask-guided crop + PIL-compatible bicubic resize + center crop + normalize.
//
// One workgroup produces BR output rows of one crop.  It builds the fixed-point
// resampling coefficients of both axes in LDS (double precision, the exact
// operation order of Pillow's precompute_coeffs / normalize_coeffs_8bpc), then
// walks the source rows the band needs in chunks of CH rows:
//   A  coalesced dword copy of the frame bytes and mask bytes of the chunk into LDS
//   B  per pixel: mask test + level-table blend -> one packed RGBX dword in LDS
//   C  horizontal pass (thread = output column) -> u8 RGBX in LDS
//   D  vertical pass accumulates into per-thread int32 registers
// and finally rounds, maps levels through the normalisation table and stores
// bf16/f32 in CHW or ViT-patch (im2col) layout.  HBM traffic is the crop's own
// box of frame + mask bytes (re-read only where neighbouring bands overlap by
// the filter support) plus the output.
//
// Replaces laser/models/model_utils.py:29-56,87-140 and the HF CLIPImageProcessor
// call at laser/models/llava_clip_model_v3.py:244,443 (PIL Resample.c semantics).
#include "common.h"

#include <math.h>
#include <string.h>

#include <vector>

namespace sgclip {

constexpr int BR = 16;        // output rows per workgroup
constexpr int CH = 8;         // source rows per chunk
constexpr int NT = 256;       // threads per workgroup
constexpr int PREC = 22;      // Pillow PRECISION_BITS for 8-bit images
constexpr int LDS_LIMIT = 160 * 1024;

struct CropLds {
    int lut8, norm, xmin, xcnt, kh, ymin, ycnt, kv, raw, ma, mb, pix, hbuf, total;
    int raw_stride, m_stride;
};

__host__ __device__ inline int al16(int v) { return (v + 15) & ~15; }

__host__ __device__ inline CropLds crop_lds(int S, int ksx, int ksy, int wn) {
    CropLds L;
    int o = 0;
    L.lut8 = o; o += 768;
    L.norm = o; o += 768 * 4;
    L.xmin = o; o += al16(S * 4);
    L.xcnt = o; o += al16(S * 4);
    L.kh = o; o += al16(ksx * S * 4);
    L.ymin = o; o += al16(BR * 4);
    L.ycnt = o; o += al16(BR * 4);
    L.kv = o; o += al16(BR * ksy * 4);
    L.raw_stride = al16(wn * 3 + 8);
    L.m_stride = al16(wn + 8);
    L.raw = o; o += CH * L.raw_stride;
    L.ma = o; o += CH * L.m_stride;
    L.mb = o; o += CH * L.m_stride;
    L.pix = o; o += al16(CH * wn * 4);
    L.hbuf = o; o += al16(CH * S * 4);
    L.total = o;
    return L;
}

// shortest edge -> S, long edge -> floor(S*long/short); then the center S x S window
__host__ __device__ inline void crop_geom(int w, int h, int S, int &new_w, int &new_h, int &top, int &left) {
    if (w <= h) {
        new_w = S;
        new_h = (int)(((long long)S * h) / w);
    } else {
        new_h = S;
        new_w = (int)(((long long)S * w) / h);
    }
    top = (new_h - S) / 2;
    left = (new_w - S) / 2;
}

__host__ __device__ inline int axis_ksize(int in, int out) {
    if (in == out) return 1;
    double scale = (double)in / out;
    double fs = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * fs) * 2 + 1;
}

__host__ __device__ inline double bicubic(double x) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((1.5 * x) - 2.5) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
    return 0.0;
}

// first source index and tap count of output index i (Pillow precompute_coeffs)
__host__ __device__ inline void axis_bounds(int in, int out, int i, int &xmin, int &cnt) {
#pragma clang fp contract(off)
    if (in == out) {
        xmin = i;
        cnt = 1;
        return;
    }
    double scale = (double)in / out;
    double fs = scale < 1.0 ? 1.0 : scale;
    double support = 2.0 * fs;
    double center = (i + 0.5) * scale;
    int xm = (int)(center - support + 0.5);
    if (xm < 0) xm = 0;
    int xM = (int)(center + support + 0.5);
    if (xM > in) xM = in;
    xmin = xm;
    cnt = xM - xm;
}

__device__ inline void axis_coeffs(int in, int out, int i, int &xmin, int &cnt, int *k, int kstride) {
#pragma clang fp contract(off)
    if (in == out) {
        xmin = i;
        cnt = 1;
        k[0] = 1 << PREC;
        return;
    }
    double scale = (double)in / out;
    double fs = scale < 1.0 ? 1.0 : scale;
    double support = 2.0 * fs;
    double ss = 1.0 / fs;
    double center = (i + 0.5) * scale;
    int xm = (int)(center - support + 0.5);
    if (xm < 0) xm = 0;
    int xM = (int)(center + support + 0.5);
    if (xM > in) xM = in;
    int n = xM - xm;
    double ww = 0.0;
    for (int x = 0; x < n; x++) ww += bicubic((x + xm - center + 0.5) * ss);
    for (int x = 0; x < n; x++) {
        double w = bicubic((x + xm - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        k[x * kstride] = w < 0 ? (int)(-0.5 + w * (double)(1 << PREC)) : (int)(0.5 + w * (double)(1 << PREC));
    }
    xmin = xm;
    cnt = n;
}

struct CropArgs {
    const uint8_t *frames;
    const uint8_t *masks;
    const sgclip_crop_t *crops;
    const uint8_t *lut8;
    const float *normlut;
    void *out;
    int S, nbands, layout, patch, kpad, out_dtype;
    int ksx_cap, ksy_cap, wn_cap;
    int need_lo, need_hi;  // this launch handles the crops whose own LDS need is in (need_lo, need_hi]
};

// LDS bytes one crop needs when staged with exactly its own tap counts and source span
__host__ __device__ inline int crop_need(int w, int h, int S) {
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);
    int x0, n0, x1, n1;
    axis_bounds(w, new_w, left, x0, n0);
    axis_bounds(w, new_w, left + S - 1, x1, n1);
    return crop_lds(S, axis_ksize(w, new_w), axis_ksize(h, new_h), x1 + n1 - x0).total;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <int CPT>
__global__ __launch_bounds__(NT) void crop_kernel(CropArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int crop_i = blockIdx.x / a.nbands;
    const int band = blockIdx.x - crop_i * a.nbands;
    const sgclip_crop_t c = a.crops[crop_i];
    const int S = a.S;
    const int w = c.x2 - c.x1, h = c.y2 - c.y1;
    {   // crops are binned by LDS need so that small crops run at high occupancy
        const int need = crop_need(w, h, S);
        if (need <= a.need_lo || need > a.need_hi) return;
    }
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);

    const CropLds L = crop_lds(S, a.ksx_cap, a.ksy_cap, a.wn_cap);
    uint8_t *lut8 = smem + L.lut8;
    float *norm = (float *)(smem + L.norm);
    int *xmin = (int *)(smem + L.xmin);
    int *xcnt = (int *)(smem + L.xcnt);
    int *kh = (int *)(smem + L.kh);
    int *ymin = (int *)(smem + L.ymin);
    int *ycnt = (int *)(smem + L.ycnt);
    int *kv = (int *)(smem + L.kv);
    uint8_t *raw = smem + L.raw;
    uint8_t *ma = smem + L.ma;
    uint8_t *mb = smem + L.mb;
    unsigned *pix = (unsigned *)(smem + L.pix);
    unsigned *hbuf = (unsigned *)(smem + L.hbuf);

    for (int i = tid; i < 768; i += NT) {
        lut8[i] = a.lut8[i];
        norm[i] = a.normlut[i];
    }
    for (int col = tid; col < S; col += NT) {
        int xm, n;
        axis_coeffs(w, new_w, left + col, xm, n, kh + col, S);
        xmin[col] = xm;
        xcnt[col] = n;
    }
    if (tid < BR) {
        int ym, n;
        axis_coeffs(h, new_h, top + band * BR + tid, ym, n, kv + tid * a.ksy_cap, 1);
        ymin[tid] = ym;
        ycnt[tid] = n;
    }
    __syncthreads();

    const int cx0 = xmin[0];
    const int wn = xmin[S - 1] + xcnt[S - 1] - cx0;
    const int ry0 = ymin[0];
    const int ry1 = ymin[BR - 1] + ycnt[BR - 1];
    const int wn_cap = a.wn_cap;
    if (wn > wn_cap) return;  // host sizing guarantees this never triggers

    int vacc[CPT][BR][3];
#pragma unroll
    for (int cc = 0; cc < CPT; cc++)
#pragma unroll
        for (int rr = 0; rr < BR; rr++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) vacc[cc][rr][ch] = 1 << (PREC - 1);

    const int mode = c.mode;
    const uint8_t *fbase = a.frames + c.frame_off;
    const uint8_t *mabase = a.masks + c.mask_a_off;
    const uint8_t *mbbase = a.masks + c.mask_b_off;

    for (int cy = ry0; cy < ry1; cy += CH) {
        const int nrows = min(CH, ry1 - cy);
        // ---- A: frame / mask bytes of the chunk -> LDS (wave per row, dword per lane)
        for (int r = wave; r < nrows; r += NT / 64) {
            const size_t pix0 = (size_t)(c.y1 + cy + r) * c.frame_w + (c.x1 + cx0);
            const uint8_t *src = fbase + pix0 * 3;
            const int lead = (int)((uintptr_t)src & 3);
            const unsigned *s4 = (const unsigned *)(src - lead);
            const int ndw = (lead + wn * 3 + 3) >> 2;
            unsigned *dst = (unsigned *)(raw + r * L.raw_stride);
            for (int i = lane; i < ndw; i += 64) dst[i] = s4[i];
            if (mode != SGCLIP_CROP_PLAIN) {
                const uint8_t *msrc = mabase + pix0;
                const int ml = (int)((uintptr_t)msrc & 3);
                const unsigned *m4 = (const unsigned *)(msrc - ml);
                const int mdw = (ml + wn + 3) >> 2;
                unsigned *md = (unsigned *)(ma + r * L.m_stride);
                for (int i = lane; i < mdw; i += 64) md[i] = m4[i];
                if (mode == SGCLIP_CROP_PAIR) {
                    const uint8_t *bsrc = mbbase + pix0;
                    const int bl = (int)((uintptr_t)bsrc & 3);
                    const unsigned *b4 = (const unsigned *)(bsrc - bl);
                    const int bdw = (bl + wn + 3) >> 2;
                    unsigned *bd = (unsigned *)(mb + r * L.m_stride);
                    for (int i = lane; i < bdw; i += 64) bd[i] = b4[i];
                }
            }
        }
        __syncthreads();
        // ---- B: blend -> packed pixels
        for (int r = wave; r < nrows; r += NT / 64) {
            const size_t pix0 = (size_t)(c.y1 + cy + r) * c.frame_w + (c.x1 + cx0);
            const int lead = (int)((uintptr_t)(fbase + pix0 * 3) & 3);
            const int ml = (int)((uintptr_t)(mabase + pix0) & 3);
            const int bl = (int)((uintptr_t)(mbbase + pix0) & 3);
            const uint8_t *rr = raw + r * L.raw_stride + lead;
            const uint8_t *mr = ma + r * L.m_stride + ml;
            const uint8_t *br = mb + r * L.m_stride + bl;
            unsigned *pr = pix + r * wn_cap;
            for (int p = lane; p < wn; p += 64) {
                unsigned b0 = rr[3 * p], b1 = rr[3 * p + 1], b2 = rr[3 * p + 2];
                if (mode == SGCLIP_CROP_OBJECT) {
                    const int base = mr[p] ? 256 : 0;
                    b0 = lut8[base + b0];
                    b1 = lut8[base + b1];
                    b2 = lut8[base + b2];
                } else if (mode == SGCLIP_CROP_PAIR) {
                    const bool red = mr[p] != 0, blue = br[p] != 0;
                    if (red | blue) {
                        if (blue) b0 = lut8[512 + b0];
                        if (red) b2 = lut8[512 + b2];
                    } else {
                        b0 = lut8[b0];
                        b1 = lut8[b1];
                        b2 = lut8[b2];
                    }
                }
                pr[p] = b0 | (b1 << 8) | (b2 << 16);
            }
        }
        __syncthreads();
        // ---- C: horizontal pass, thread = output column
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            const int col = tid + cc * NT;
            if (col < S) {
                const int xm = xmin[col] - cx0, n = xcnt[col];
                int acc[CH][3];
#pragma unroll
                for (int r = 0; r < CH; r++) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (PREC - 1);
                for (int t = 0; t < n; t++) {
                    const int k = kh[t * S + col];
#pragma unroll
                    for (int r = 0; r < CH; r++) {
                        const unsigned px = pix[r * wn_cap + xm + t];
                        acc[r][0] += __mul24(k, (int)(px & 255u));
                        acc[r][1] += __mul24(k, (int)((px >> 8) & 255u));
                        acc[r][2] += __mul24(k, (int)((px >> 16) & 255u));
                    }
                }
#pragma unroll
                for (int r = 0; r < CH; r++)
                    hbuf[r * S + col] = (unsigned)clip8(acc[r][0]) | ((unsigned)clip8(acc[r][1]) << 8) |
                                        ((unsigned)clip8(acc[r][2]) << 16);
            }
        }
        __syncthreads();
        // ---- D: vertical pass, accumulate this chunk's rows
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            const int col = tid + cc * NT;
            if (col < S) {
#pragma unroll
                for (int rr = 0; rr < BR; rr++) {
                    const int ym = __builtin_amdgcn_readfirstlane(ymin[rr]);
                    const int n = __builtin_amdgcn_readfirstlane(ycnt[rr]);
                    const int t0 = max(0, cy - ym), t1 = min(n, cy + nrows - ym);
                    for (int t = t0; t < t1; t++) {
                        const int k = kv[rr * a.ksy_cap + t];
                        const unsigned px = hbuf[(ym + t - cy) * S + col];
                        vacc[cc][rr][0] += __mul24(k, (int)(px & 255u));
                        vacc[cc][rr][1] += __mul24(k, (int)((px >> 8) & 255u));
                        vacc[cc][rr][2] += __mul24(k, (int)((px >> 16) & 255u));
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- output
    const int patch = a.patch;
    const int G = (a.layout == SGCLIP_LAYOUT_PATCHES) ? S / patch : 1;
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        const int col = tid + cc * NT;
        if (col < S) {
#pragma unroll
            for (int rr = 0; rr < BR; rr++) {
                const int oy = band * BR + rr;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float v = norm[ch * 256 + clip8(vacc[cc][rr][ch])];
                    size_t o;
                    if (a.layout == SGCLIP_LAYOUT_PATCHES) {
                        const size_t prow = (size_t)crop_i * G * G + (size_t)(oy / patch) * G + col / patch;
                        o = prow * a.kpad + (size_t)ch * patch * patch + (oy % patch) * patch + col % patch;
                    } else {
                        o = (((size_t)crop_i * 3 + ch) * S + oy) * S + col;
                    }
                    if (a.out_dtype == SGCLIP_DTYPE_BF16)
                        ((bf16_t *)a.out)[o] = (bf16_t)v;
                    else
                        ((float *)a.out)[o] = v;
                }
            }
        }
    }
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_crop_preprocess(const uint8_t *frames_dev, size_t frames_bytes, const uint8_t *masks_dev,
                                      size_t masks_bytes, const sgclip_crop_t *crops, sgclip_crop_t *crops_dev,
                                      int crops_dev_ready, int n_crops, const uint8_t *luts, const float *norm_lut,
                                      void *lut_dev, int S, int layout, int patch, int kpad, int out_dtype,
                                      void *out_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crops == 0) return SGCLIP_OK;
    if (!frames_dev || !crops || !crops_dev || !luts || !norm_lut || !lut_dev || !out_dev || n_crops < 0) {
        set_error("sgclip_crop_preprocess: null argument");
        return SGCLIP_EINVAL;
    }
    if (S <= 0 || S % BR != 0 || S > 2 * NT) {
        set_error("sgclip_crop_preprocess: S=%d must be a multiple of %d and <= %d", S, BR, 2 * NT);
        return SGCLIP_EINVAL;
    }
    if (layout == SGCLIP_LAYOUT_PATCHES) {
        if (patch <= 0 || S % patch != 0 || kpad < 3 * patch * patch) {
            set_error("sgclip_crop_preprocess: bad patch layout (S=%d patch=%d kpad=%d)", S, patch, kpad);
            return SGCLIP_EINVAL;
        }
    } else if (layout != SGCLIP_LAYOUT_CHW) {
        set_error("sgclip_crop_preprocess: unknown layout %d", layout);
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)frames_dev & 3) || ((uintptr_t)masks_dev & 3)) {
        set_error("sgclip_crop_preprocess: frames_dev / masks_dev must be 4-byte aligned");
        return SGCLIP_EINVAL;
    }
    // occupancy classes by LDS need: 4 / 3 / 2 / 1 workgroups per CU
    static const int class_hi[4] = {40 * 1024, 53 * 1024, 80 * 1024, LDS_LIMIT};
    int ksx[4] = {1, 1, 1, 1}, ksy[4] = {1, 1, 1, 1}, wnc[4] = {1, 1, 1, 1}, members[4] = {0, 0, 0, 0};
    double algo_bytes = 0;  // per crop: 3hw frame + m*hw mask + out
    for (int i = 0; i < n_crops; i++) {
        const sgclip_crop_t &c = crops[i];
        const int w = c.x2 - c.x1, h = c.y2 - c.y1;
        if (w <= 0 || h <= 0) {
            set_error("image bbox issue: crop %d has an empty box [%d:%d, %d:%d]", i, c.y1, c.y2, c.x1, c.x2);
            return SGCLIP_EBBOX;
        }
        if (c.x1 < 0 || c.y1 < 0 || c.x2 > c.frame_w || c.y2 > c.frame_h || c.frame_off < 0 ||
            (size_t)c.frame_off + (size_t)c.frame_w * c.frame_h * 3 > frames_bytes) {
            set_error("sgclip_crop_preprocess: crop %d box or frame outside the frame buffer", i);
            return SGCLIP_EINVAL;
        }
        const size_t mbytes = (size_t)c.frame_w * c.frame_h;
        if (c.mode == SGCLIP_CROP_OBJECT || c.mode == SGCLIP_CROP_PAIR) {
            if (!masks_dev || c.mask_a_off < 0 || (size_t)c.mask_a_off + mbytes > masks_bytes ||
                (c.mode == SGCLIP_CROP_PAIR && (c.mask_b_off < 0 || (size_t)c.mask_b_off + mbytes > masks_bytes))) {
                set_error("sgclip_crop_preprocess: crop %d mask outside the mask buffer", i);
                return SGCLIP_EINVAL;
            }
        } else if (c.mode != SGCLIP_CROP_PLAIN) {
            set_error("sgclip_crop_preprocess: crop %d has unknown mode %d", i, c.mode);
            return SGCLIP_EINVAL;
        }
        algo_bytes += (3.0 + (c.mode == SGCLIP_CROP_PAIR ? 2 : (c.mode == SGCLIP_CROP_OBJECT ? 1 : 0))) * w * h +
                      3.0 * S * S * (out_dtype == SGCLIP_DTYPE_BF16 ? 2 : 4);
        int new_w, new_h, top, left;
        crop_geom(w, h, S, new_w, new_h, top, left);
        int x0, n0, x1, n1;
        axis_bounds(w, new_w, left, x0, n0);
        axis_bounds(w, new_w, left + S - 1, x1, n1);
        const int need = crop_need(w, h, S);
        if (need > LDS_LIMIT) {
            set_error("sgclip_crop_preprocess: crop %d needs %d bytes of LDS staging (limit %d): source span %d px", i, need,
                      LDS_LIMIT, x1 + n1 - x0);
            return SGCLIP_EINVAL;
        }
        int k = 0;
        while (need > class_hi[k]) k++;
        ksx[k] = std::max(ksx[k], axis_ksize(w, new_w));
        ksy[k] = std::max(ksy[k], axis_ksize(h, new_h));
        wnc[k] = std::max(wnc[k], x1 + n1 - x0);
        members[k]++;
    }
    uint8_t *lut8_dev = (uint8_t *)lut_dev;
    float *norm_dev = (float *)(lut8_dev + 768);
    hipError_t e = hipMemcpyAsync(lut8_dev, luts, 768, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(norm_dev, norm_lut, 768 * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && !crops_dev_ready)
        e = hipMemcpyAsync(crops_dev, crops, (size_t)n_crops * sizeof(sgclip_crop_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
        set_error("sgclip_crop_preprocess: table upload failed: %s", hipGetErrorString(e));
        return SGCLIP_ELAUNCH;
    }
    CropArgs a;
    a.frames = frames_dev;
    a.masks = masks_dev;
    a.crops = crops_dev;
    a.lut8 = lut8_dev;
    a.normlut = norm_dev;
    a.out = out_dev;
    a.S = S;
    a.nbands = S / BR;
    a.layout = layout;
    a.patch = patch;
    a.kpad = kpad;
    a.out_dtype = out_dtype;
    const dim3 grid((unsigned)n_crops * a.nbands), block(NT);
    KernelTimer timer(TK_CROP, algo_bytes, stream);
    for (int k = 0; k < 4; k++) {
        if (!members[k]) continue;
        a.ksx_cap = ksx[k];
        a.ksy_cap = ksy[k];
        a.wn_cap = wnc[k];
        a.need_lo = k ? class_hi[k - 1] : 0;
        a.need_hi = class_hi[k];
        // a class's shared capacities can exceed its bound (taps from one crop, span from another)
        const int lds_bytes = crop_lds(S, ksx[k], ksy[k], wnc[k]).total;
        if (lds_bytes > LDS_LIMIT) {
            set_error("sgclip_crop_preprocess: LDS class %d needs %d bytes", k, lds_bytes);
            return SGCLIP_EINVAL;
        }
        if (S <= NT) {
            (void)hipFuncSetAttribute((const void *)crop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL(crop_kernel<1>, grid, block, lds_bytes, stream, a);
        } else {
            (void)hipFuncSetAttribute((const void *)crop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL(crop_kernel<2>, grid, block, lds_bytes, stream, a);
        }
    }
    return launch_check("sgclip_crop_preprocess");
}
