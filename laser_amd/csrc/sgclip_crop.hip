// Mask-guided crop + PIL-compatible bicubic resize + center crop + normalize.
//
// One workgroup produces one or more bands of BR output rows of one crop (the
// host picks bands per workgroup so that the grid still fills the chip).  It
// builds the fixed-point resampling coefficients of both axes in LDS once
// (double precision, the exact operation order of Pillow's precompute_coeffs /
// normalize_coeffs_8bpc), then for every band walks the source rows it needs in
// chunks of CH rows:
//   A  asynchronous 16-byte-per-lane copy of the chunk's frame and mask bytes into LDS,
//      issued one chunk ahead so that it runs underneath C and D
//   B  per source column, all rows of the chunk: mask test + level-table blend -> packed
//      RGBX dwords in LDS, a column's CH rows side by side
//   C  horizontal pass (thread = output column) -> u8 RGBX in LDS
//   D  vertical pass accumulates into per-thread int32 registers
// and finally rounds, maps levels through the normalisation table and stores
// bf16/f32 in CHW or ViT-patch (im2col) layout (bf16: transposed through LDS and
// stored 16 bytes per lane).  HBM traffic is the crop's own box of frame + mask
// bytes (re-read only where neighbouring bands overlap by the filter support)
// plus the output.
// A crop whose source span is too wide for that staging to fit the LDS (frames beyond ~1 700 px on the
// short side, e.g. a whole 4K frame) runs the same passes over SEGMENTS of output columns: passes A-C
// once per segment and chunk on the segment's own source span, pass D after the last segment.
//
// Replaces laser/models/model_utils.py:29-56,87-140 and the HF CLIPImageProcessor
// call at laser/models/llava_clip_model_v3.py:244,443 (PIL Resample.c semantics).
#include "common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

namespace sgclip {

constexpr int BR = 16;        // output rows per workgroup
constexpr int CH = 8;         // source rows per chunk
#ifndef CROP_NG
#define CROP_NG 2
#endif
constexpr int NG = CROP_NG;   // thread groups: each takes CH/NG chunk rows in pass C and BR/NG output rows in pass D
constexpr int COLS = 256;     // output columns per group pass
constexpr int NT = COLS * NG; // threads per workgroup
constexpr int CR = CH / NG, DR = BR / NG;
constexpr int PREC = 22;      // Pillow PRECISION_BITS for 8-bit images
constexpr int LDS_LIMIT = 160 * 1024;

struct CropLds {
    int lut8, norm, xmin, xcnt, kh, ymin, ycnt, kv, raw, ma, mb, pix, hbuf, total;
    int raw_stride, m_stride;
};

__host__ __device__ inline int al16(int v) { return (v + 15) & ~15; }

// vrows = output rows one workgroup produces (bands per workgroup * BR)
__host__ __device__ inline CropLds crop_lds(int S, int ksx, int ksy, int wn, int vrows) {
    CropLds L;
    int o = 0;
    L.lut8 = o; o += 768;
    L.norm = o; o += 768 * 4;
    L.xmin = o; o += al16(S * 4);
    L.xcnt = o; o += al16(S * 4);
    L.kh = o; o += al16(ksx * S * 4);
    L.ymin = o; o += al16(vrows * 4);
    L.ycnt = o; o += al16(vrows * 4);
    L.kv = o; o += al16(vrows * ksy * 4);
    // a staged row holds the source bytes from the 4-byte boundary below its first
    // pixel, copied in whole 16-byte pieces
    L.raw_stride = al16(wn * 3 + 3);
    L.m_stride = al16(wn + 3);
    L.raw = o; o += CH * L.raw_stride;
    L.ma = o; o += CH * L.m_stride;
    L.mb = o; o += CH * L.m_stride;
    L.pix = o; o += CH * wn * 4;  // [source column][chunk row]
    L.hbuf = o; o += al16(CH * S * 4);
    // a band's bf16 output is transposed through pix/hbuf once they are idle
    const int out_stage = BR * 3 * S * 2;
    if (o - L.pix < out_stage) o = L.pix + out_stage;
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
    int bands_per_wg;      // divides nbands; a workgroup walks this many consecutive bands
    int ksx_cap, ksy_cap, wn_cap;
    int need_lo, need_hi;  // this launch handles the crops whose own LDS need is in (need_lo, need_hi]
    int nseg;              // column segments (segmented launch only; 1 otherwise)
    int mfma_split;        // 1: crops that crop_mfma_ok() accepts belong to crop_mfma_kernel, the others to crop_kernel
};

// output columns of segment s: [seg_begin(s), seg_begin(s + 1)); column c lies in segment c * nseg / S
__host__ __device__ inline int seg_begin(int s, int S, int nseg) { return (s * S + nseg - 1) / nseg; }

// LDS bytes one crop needs when staged with exactly its own tap counts and source span
__host__ __device__ inline int crop_need(int w, int h, int S, int vrows) {
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);
    int x0, n0, x1, n1;
    axis_bounds(w, new_w, left, x0, n0);
    axis_bounds(w, new_w, left + S - 1, x1, n1);
    return crop_lds(S, axis_ksize(w, new_w), axis_ksize(h, new_h), x1 + n1 - x0, vrows).total;
}

// ---- the matrix-core variant (crop_mfma_kernel below): LDS layout, eligibility -------------------------------------
constexpr int MR = 16;        // source rows per chunk
constexpr int NTM = 512;      // threads per workgroup: 8 waves, each owns up to 3 blocks of 16 output columns
constexpr int MKB = 2;        // 64-pixel source windows per block of 16 output columns

struct CropLdsM {
    int lut8, norm, xmin, xcnt, ymin, ycnt, kv, raw, ma, mb, planes, kh, total;
    int raw_stride, m_stride, pst;
};

__host__ __device__ inline CropLdsM crop_lds_m(int S, int ksx, int ksy, int wn, int vrows) {
    CropLdsM L;
    int o = 0;
    L.lut8 = o; o += 1024;   // the three level tables and the identity
    L.norm = o; o += 768 * 4;
    L.xmin = o; o += al16(S * 4);
    L.xcnt = o; o += al16(S * 4);
    L.ymin = o; o += al16(vrows * 4);
    L.ycnt = o; o += al16(vrows * 4);
    L.kv = o; o += al16(vrows * ksy * 4);
    // a blend item is four pixels read as whole dwords: the last one of a row may read up to three pixels and one
    // dword past the span
    L.raw_stride = al16(wn * 3 + 3 + 9 + 4);
    L.m_stride = al16(wn + 3 + 3 + 4);
    L.raw = o; o += MR * L.raw_stride;
    L.ma = o; o += MR * L.m_stride;
    L.mb = o; o += MR * L.m_stride;
    // channel planes [3][MR][pst] of blended bytes; the horizontal coefficient table is only read while the per-wave
    // operands are built and the band's output staging only after its last chunk, so both share the planes' bytes.
    // A block's 128-byte window may start anywhere in a row: the last row needs that much room behind it.
    L.pst = al16(wn + 3) + 16;
    int body = 3 * MR * L.pst + 64 * MKB;
    if (body < al16(ksx * S * 4)) body = al16(ksx * S * 4);
    if (body < BR * 3 * S * 2) body = BR * 3 * S * 2;
    L.planes = o; L.kh = o; o += body;
    L.total = o;
    return L;
}

__host__ __device__ inline int crop_need_m(int w, int h, int S, int vrows) {
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);
    int x0, n0, x1, n1;
    axis_bounds(w, new_w, left, x0, n0);
    axis_bounds(w, new_w, left + S - 1, x1, n1);
    return crop_lds_m(S, axis_ksize(w, new_w), axis_ksize(h, new_h), x1 + n1 - x0, vrows).total;
}

// Whether crop_mfma_kernel takes the crop: at most 24 column blocks, and the source pixels under any 16 neighbouring
// output columns (15 column steps of `scale` pixels, the taps of the last one, up to 15 pixels of alignment and one of
// rounding on either side) fit MKB windows of 64 pixels.  The kernel re-checks the exact spans and leaves the output
// untouched if this bound were ever wrong, so a miss shows in the parity tests.
__host__ __device__ inline bool crop_mfma_ok(int w, int h, int S, int vrows) {
#pragma clang fp contract(off)
    if (S > 16 * 24) return false;
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);
    const double scale = (double)w / new_w;
    const double span = 15.0 * scale + axis_ksize(w, new_w) + 17.0;
    if (span > 64.0 * MKB) return false;
    return crop_need_m(w, h, S, vrows) <= LDS_LIMIT;
}

// 64-pixel windows the crop's column blocks need by the same bound: 1 up to a scale of about 2.4 (three quarters of
// the pair crops of a 720p clip), else 2.  The one-window kernel holds half the coefficient operands (no spills with two
// column blocks per wave) and issues half the row-pass MFMAs.
__host__ __device__ inline int crop_mfma_windows(int w, int h, int S) {
#pragma clang fp contract(off)
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);
    const double scale = (double)w / new_w;
    return 15.0 * scale + axis_ksize(w, new_w) + 17.0 <= 64.0 ? 1 : 2;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Hides a thread index from loop-invariant code motion: address arithmetic derived from
// the result is redone where it is used instead of living in registers across the kernel.
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int CPT, bool SEG>
__global__ __launch_bounds__(NT, 4) void crop_kernel(CropArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = tid / COLS, gcol = tid % COLS;  // grp is wave-uniform
    const int bpw = a.bands_per_wg;
    const int wg_per_crop = a.nbands / bpw;
    const int slot = blockIdx.x / wg_per_crop;
    const int band0 = (blockIdx.x - slot * wg_per_crop) * bpw;
    const int ord = a.crops[slot].order;          // 1 + the crop this group of workgroups takes, or 0: its own slot
    const int crop_i = ord > 0 ? ord - 1 : slot;
    const int vrows = bpw * BR;
    const sgclip_crop_t c = a.crops[crop_i];
    const int S = a.S;
    const int w = c.x2 - c.x1, h = c.y2 - c.y1;
    if (a.mfma_split && crop_mfma_ok(w, h, S, vrows)) return;
    {   // crops are binned by LDS need so that small crops run at high occupancy
        const int need = crop_need(w, h, S, vrows);
        if (need <= a.need_lo || need > a.need_hi) return;
    }
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);

    const CropLds L = crop_lds(S, a.ksx_cap, a.ksy_cap, a.wn_cap, vrows);
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

    // ---- setup: level tables and both axes' coefficients (once per workgroup)
    if (tid < 192) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a.lut8 + tid * 4), LDS_PTR(lut8 + wave * 256), 4, 0, 0);
    if (tid < 256) {
#pragma unroll
        for (int i = 0; i < 3; i++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a.normlut + i * 256 + tid), LDS_PTR(norm + i * 256 + wave * 64), 4, 0, 0);
    }
    for (int col = tid; col < S; col += NT) {
        int xm, n;
        axis_coeffs(w, new_w, left + col, xm, n, kh + col, S);
        xmin[col] = xm;
        xcnt[col] = n;
    }
    // vertical axis from the top thread down: those threads have no column
    for (int vr = NT - 1 - tid; vr < vrows; vr += NT) {
        int ym, n;
        axis_coeffs(h, new_h, top + band0 * BR + vr, ym, n, kv + vr, vrows);
        ymin[vr] = ym;
        ycnt[vr] = n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nseg = SEG ? a.nseg : 1;
    // source span staged at a time: the whole visible span, or one segment's (set per segment below)
    int cx0 = __builtin_amdgcn_readfirstlane(xmin[0]);
    int wn = __builtin_amdgcn_readfirstlane(xmin[S - 1] + xcnt[S - 1]) - cx0;
    auto seg_span = [&](int sg) {
        const int c0 = seg_begin(sg, S, nseg), c1 = seg_begin(sg + 1, S, nseg) - 1;
        cx0 = __builtin_amdgcn_readfirstlane(xmin[c0]);
        wn = __builtin_amdgcn_readfirstlane(xmin[c1] + xcnt[c1]) - cx0;
    };
    if (SEG) {
        for (int sg = 0; sg < nseg; sg++) {
            seg_span(sg);
            if (wn > a.wn_cap) return;  // host sizing guarantees this never triggers
        }
        seg_span(0);
    } else if (wn > a.wn_cap) {
        return;  // host sizing guarantees this never triggers
    }

    int hx[CPT], hn[CPT];  // this thread's columns: first source pixel (relative to its segment's span) and tap count
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        const int col = min(gcol + cc * COLS, S - 1);
        hx[cc] = xmin[col] - (SEG ? xmin[seg_begin(col * nseg / S, S, nseg)] : cx0);
        hn[cc] = xcnt[col];
    }

    const int mode = c.mode;
    const uint8_t *fbase = a.frames + c.frame_off;
    const uint8_t *mabase = a.masks + c.mask_a_off;
    const uint8_t *mbbase = a.masks + c.mask_b_off;

    // A: wave per source row; the load unit writes LDS itself (no registers), so the
    // copy is asynchronous until the next s_waitcnt vmcnt(0)
    auto copy_row = [&](const uint8_t *src, int nbytes, uint8_t *dst) {
        const int lead = (int)((uintptr_t)src & 3);
        const uint8_t *s0 = src - lead;
        const int npiece = (lead + nbytes + 15) >> 4;
        for (int i = 0; i < npiece; i += 64)
            if (i + lane < npiece)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(s0 + (size_t)(i + lane) * 16), LDS_PTR(dst + i * 16), 16, 0, 0);
    };
    auto issue_chunk = [&](int cy, int cy_end) {
        const int nrows = min(CH, cy_end - cy);
        for (int r = wave; r < nrows; r += NT / 64) {
            const size_t pix0 = (size_t)(c.y1 + cy + r) * c.frame_w + (c.x1 + cx0);
            copy_row(fbase + pix0 * 3, wn * 3, raw + r * L.raw_stride);
            if (mode != SGCLIP_CROP_PLAIN) {
                copy_row(mabase + pix0, wn, ma + r * L.m_stride);
                if (mode == SGCLIP_CROP_PAIR) copy_row(mbbase + pix0, wn, mb + r * L.m_stride);
            }
        }
    };
    auto band_rows = [&](int bl, int &r0, int &r1) {
        r0 = __builtin_amdgcn_readfirstlane(ymin[bl * BR]);
        r1 = __builtin_amdgcn_readfirstlane(ymin[bl * BR + BR - 1] + ycnt[bl * BR + BR - 1]);
    };

    const int patch = a.patch;
    const bool patches = a.layout == SGCLIP_LAYOUT_PATCHES;
    const int G = patches ? S / patch : 1;
    const bool wide = a.out_dtype == SGCLIP_DTYPE_BF16 && ((uintptr_t)a.out & 15) == 0 &&
                      (!patches || ((patch & 7) == 0 && (a.kpad & 7) == 0));

    int ry0, ry1;
    band_rows(0, ry0, ry1);
    issue_chunk(ry0, ry1);
    for (int bl = 0; bl < bpw; bl++) {
        const int band = band0 + bl;
        int ym[DR], yn[DR];  // wave-uniform: first source row and tap count of this group's output rows
#pragma unroll
        for (int rr = 0; rr < DR; rr++) {
            ym[rr] = __builtin_amdgcn_readfirstlane(ymin[bl * BR + grp * DR + rr]);
            yn[rr] = __builtin_amdgcn_readfirstlane(ycnt[bl * BR + grp * DR + rr]);
        }
        int vacc[CPT][DR][3];
#pragma unroll
        for (int cc = 0; cc < CPT; cc++)
#pragma unroll
            for (int rr = 0; rr < DR; rr++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) vacc[cc][rr][ch] = 1 << (PREC - 1);
        int nry0 = 0, nry1 = 0;
        if (bl + 1 < bpw) band_rows(bl + 1, nry0, nry1);

        for (int cy = ry0; cy < ry1; cy += CH) {
            const int nrows = min(CH, ry1 - cy);
          for (int sg = 0; sg < nseg; sg++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- B: blend -> packed pixels; thread = source column, all rows of the chunk, so
            // that a column's CH pixels sit side by side for pass C (rows past nrows hold
            // stale bytes whose results are never used)
            {
                // offset of a row's first pixel from the 4-byte boundary its copy started at
                const size_t pix0 = (size_t)(c.y1 + cy) * c.frame_w + (c.x1 + cx0);
                const int lead0 = (int)((uintptr_t)(fbase + pix0 * 3) & 3), lead_step = (c.frame_w * 3) & 3;
                const int ml0 = (int)((uintptr_t)(mabase + pix0) & 3), bo0 = (int)((uintptr_t)(mbbase + pix0) & 3);
                const int m_step = c.frame_w & 3;
                // RG rows per thread: narrow crops split the chunk's rows over the threads
                auto blend = [&](auto rg_c) {
                    constexpr int RG = decltype(rg_c)::value;  // rows per thread
                    constexpr int RQ = RG < 4 ? RG : 4;        // rows per store
                    constexpr int XT = NT / (CH / RG);         // threads along x
                    const int bt = fresh(tid);
                    const int rbase = (bt / XT) * RG;
                    for (int x = bt % XT; x < wn; x += XT) {
#pragma nounroll
                        for (int r0 = rbase; r0 < rbase + RG; r0 += RQ) {
                            unsigned o[RQ];
#pragma unroll
                            for (int i = 0; i < RQ; i++) {
                                const int r = r0 + i;
                                o[i] = 0;
                                if (r >= nrows) continue;
                                const int ml = (ml0 + r * m_step) & 3, bo = (bo0 + r * m_step) & 3;
                                const uint8_t *rb = raw + r * L.raw_stride + ((lead0 + r * lead_step) & 3) + 3 * x;
                                unsigned b0 = rb[0], b1 = rb[1], b2 = rb[2];
                                if (mode == SGCLIP_CROP_OBJECT) {
                                    const int base = ma[r * L.m_stride + ml + x] ? 256 : 0;
                                    b0 = lut8[base + b0];
                                    b1 = lut8[base + b1];
                                    b2 = lut8[base + b2];
                                } else if (mode == SGCLIP_CROP_PAIR) {
                                    const bool red = ma[r * L.m_stride + ml + x] != 0;
                                    const bool blue = mb[r * L.m_stride + bo + x] != 0;
                                    if (red | blue) {
                                        if (blue) b0 = lut8[512 + b0];
                                        if (red) b2 = lut8[512 + b2];
                                    } else {
                                        b0 = lut8[b0];
                                        b1 = lut8[b1];
                                        b2 = lut8[b2];
                                    }
                                }
                                o[i] = b0 | (b1 << 8) | (b2 << 16);
                            }
                            if constexpr (RQ == 1)
                                pix[x * CH + r0] = o[0];
                            else if constexpr (RQ == 2)
                                *(uint2 *)(pix + x * CH + r0) = make_uint2(o[0], o[1]);
                            else
                                *(uint4 *)(pix + x * CH + r0) = make_uint4(o[0], o[1], o[2], o[3]);
                        }
                    }
                };
                if (wn > NT / 2)
                    blend(std::integral_constant<int, 8>{});
                else if (wn > NT / 4)
                    blend(std::integral_constant<int, 4>{});
                else if (wn > NT / 8)
                    blend(std::integral_constant<int, 2>{});
                else
                    blend(std::integral_constant<int, 1>{});
            }
            __syncthreads();
            // the raw bytes are consumed: refill them underneath C and D (next segment of this chunk, else
            // the first segment of the next chunk / band)
            if (SEG && sg + 1 < nseg) {
                seg_span(sg + 1);
                issue_chunk(cy, ry1);
            } else {
                if (SEG) seg_span(0);
                if (cy + CH < ry1)
                    issue_chunk(cy + CH, ry1);
                else if (bl + 1 < bpw)
                    issue_chunk(nry0, nry1);
            }
            // ---- C: horizontal pass, thread = output column x CR chunk rows
            if (grp * CR < nrows) {
                const int ccol = fresh(gcol);
#pragma unroll
                for (int cc = 0; cc < CPT; cc++) {
                    const int col = ccol + cc * COLS;
                    if (col < S && (!SEG || col * nseg / S == sg)) {
                        const int xm = hx[cc], n = hn[cc];
                        int acc[CR][3];
#pragma unroll
                        for (int r = 0; r < CR; r++) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (PREC - 1);
                        for (int t = 0; t < n; t++) {
                            const int k = kh[t * S + col];
                            unsigned pxv[CR];
                            const unsigned *src = pix + (xm + t) * CH + grp * CR;
                            if constexpr (CR == 8) {
                                const uint4 q0 = *(const uint4 *)src, q1 = *(const uint4 *)(src + 4);
                                pxv[0] = q0.x, pxv[1] = q0.y, pxv[2] = q0.z, pxv[3] = q0.w;
                                pxv[4] = q1.x, pxv[5] = q1.y, pxv[6] = q1.z, pxv[7] = q1.w;
                            } else if constexpr (CR == 4) {
                                const uint4 q = *(const uint4 *)src;
                                pxv[0] = q.x, pxv[1] = q.y, pxv[2] = q.z, pxv[3] = q.w;
                            } else {
                                const uint2 q = *(const uint2 *)src;
                                pxv[0] = q.x, pxv[1] = q.y;
                            }
#pragma unroll
                            for (int r = 0; r < CR; r++) {
                                const unsigned px = pxv[r];
                                acc[r][0] += __mul24(k, (int)(px & 255u));
                                acc[r][1] += __mul24(k, (int)((px >> 8) & 255u));
                                acc[r][2] += __mul24(k, (int)((px >> 16) & 255u));
                            }
                        }
#pragma unroll
                        for (int r = 0; r < CR; r++)
                            hbuf[(grp * CR + r) * S + col] = (unsigned)clip8(acc[r][0]) | ((unsigned)clip8(acc[r][1]) << 8) |
                                                             ((unsigned)clip8(acc[r][2]) << 16);
                    }
                }
            }
          }
            __syncthreads();
            // ---- D: vertical pass, thread = output column x DR output rows; accumulate this chunk's rows
            const int dcol = fresh(gcol);
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) {
                const int col = dcol + cc * COLS;
                if (col < S) {
#pragma unroll
                    for (int rr = 0; rr < DR; rr++) {
                        const int t0 = max(0, cy - ym[rr]), t1 = min(yn[rr], cy + nrows - ym[rr]);
                        for (int t = t0; t < t1; t++) {
                            const int k = kv[t * vrows + bl * BR + grp * DR + rr];
                            const unsigned px = hbuf[(ym[rr] + t - cy) * S + col];
                            vacc[cc][rr][0] += __mul24(k, (int)(px & 255u));
                            vacc[cc][rr][1] += __mul24(k, (int)((px >> 8) & 255u));
                            vacc[cc][rr][2] += __mul24(k, (int)((px >> 16) & 255u));
                        }
                    }
                }
            }
        }
        ry0 = nry0;
        ry1 = nry1;

        // ---- output of this band
        if (wide) {
            // transpose through LDS so that every lane stores 8 consecutive columns (16 bytes)
            bf16_t *stage = (bf16_t *)(smem + L.pix);
            __syncthreads();
            const int ocol = fresh(gcol);
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) {
                const int col = ocol + cc * COLS;
                if (col < S) {
#pragma unroll
                    for (int rr = 0; rr < DR; rr++)
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            stage[((grp * DR + rr) * 3 + ch) * S + col] = (bf16_t)norm[ch * 256 + clip8(vacc[cc][rr][ch])];
                }
            }
            __syncthreads();
            const int per_row = S >> 3;
            for (int v = fresh(tid); v < BR * 3 * per_row; v += NT) {
                const int line = v / per_row, col = (v - line * per_row) << 3;
                const int rr = line / 3, ch = line - rr * 3;
                const int oy = band * BR + rr;
                size_t o;
                if (patches) {
                    const size_t prow = (size_t)crop_i * G * G + (size_t)(oy / patch) * G + col / patch;
                    o = prow * a.kpad + (size_t)ch * patch * patch + (oy % patch) * patch + col % patch;
                } else {
                    o = (((size_t)crop_i * 3 + ch) * S + oy) * S + col;
                }
                *(uint4 *)((bf16_t *)a.out + o) = *(const uint4 *)(stage + line * S + col);
            }
            continue;
        }
        const int scol = fresh(gcol);
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            const int col = scol + cc * COLS;
            if (col < S) {
#pragma unroll
                for (int rr = 0; rr < DR; rr++) {
                    const int oy = band * BR + grp * DR + rr;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float v = norm[ch * 256 + clip8(vacc[cc][rr][ch])];
                        size_t o;
                        if (patches) {
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
}


// ---------------------------------------------------------------------------
// The same crop on the matrix cores.  Pillow's two resampling passes are integer matrix products with banded
// coefficient matrices: row pass  H[y][c] = clip8((2^21 + sum_x P[y][x] K[x][c]) >> 22), column pass likewise over y.
// v_mfma_i32_16x16x64_i8 multiplies signed bytes exactly, so
//   * pixels are held as p - 128 (the constant 128 * sum of a column's coefficients goes into the accumulator start),
//   * every 22-bit coefficient is three signed byte digits d0 + 2^8 d1 + 2^16 d2: three products, recombined with shifts.
// A wave owns NBW blocks of 16 output columns.  Its B operands of the row pass (the digits of the coefficients of its
// columns over a window of MKB * 64 source pixels that starts at a 16-byte boundary) are built once per workgroup and
// stay in registers.  Per chunk of MR = 16 source rows:
//   A  DMA of the chunk's frame and mask rows into LDS (issued one chunk ahead)
//   B  blend: work item = 4 neighbouring pixels of a row -> one dword per channel plane [channel][row][x]
//   C  per channel and block: A operand = 16 rows x 64 pixels straight from the plane (ds_read_b128), 3 digit products;
//      the result lane holds 4 neighbouring rows of one column: clipped and packed, that dword IS the B operand of
//   D  the column pass (K slot (g, j < 4) <-> chunk row 4g + j, the other slots zero), A operand = the digits of this
//      band's vertical coefficients for the chunk's rows; accumulated per lane for 4 output rows of its column.
// No pass-C output goes through LDS and there is no barrier between C and D; two barriers per chunk.
// ---------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));

// c = d0 + 256 d1 + 65536 d2 with d0, d1 in [-128, 127] (|c| < 2^23: d2 fits a signed byte as well)
__device__ __forceinline__ void byte_digits(int c, int &d0, int &d1, int &d2) {
    d0 = (int)(signed char)(c & 255);
    const int c1 = (c - d0) >> 8;
    d1 = (int)(signed char)(c1 & 255);
    d2 = (c1 - d1) >> 8;
}

// bytes 0..3 = clip8(v0) .. clip8(v3).  v_ashr_pk_u8_i32 shifts two values, saturates them to bytes and writes HALF of
// its destination (op_sel picks the high half); the other half keeps the register's old bits (tools/micro/ashr_pk.hip).
// The compiler forms the low-half instruction by itself from shift + clamp + pack code and then treats the stale upper
// half as zero, so such code must not be left to it where the upper bytes matter.
__device__ __forceinline__ unsigned pack_levels(int v0, int v1, int v2, int v3) {
    unsigned d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "=v"(d) : "v"(v0), "v"(v1));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22 op_sel:[0,0,0,1]" : "+v"(d) : "v"(v2), "v"(v3));
    return d;
}
static_assert(PREC == 22, "pack_levels shifts by the literal 22");

// (two-window crops have spans beyond ~540 pixels: their staging leaves room for one workgroup per CU anyway, so that
// kernel may use the registers of two: with 128 it spills coefficient operands, and every reload in the chunk loop
// waits for the row prefetch as well)
template <int NBW, int KB>
__global__ __launch_bounds__(NTM, (NBW <= 2 && KB == 1) ? 4 : 2) void crop_mfma_kernel(CropArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpw = a.bands_per_wg;
    const int wg_per_crop = a.nbands / bpw;
    const int slot = blockIdx.x / wg_per_crop;
    const int band0 = (blockIdx.x - slot * wg_per_crop) * bpw;
    const int ord = a.crops[slot].order;          // 1 + the crop this group of workgroups takes, or 0: its own slot
    const int crop_i = ord > 0 ? ord - 1 : slot;
    const int vrows = bpw * BR;
    const sgclip_crop_t c = a.crops[crop_i];
    const int S = a.S;
    const int w = c.x2 - c.x1, h = c.y2 - c.y1;
    if (!crop_mfma_ok(w, h, S, vrows) || crop_mfma_windows(w, h, S) != KB) return;
    {
        const int need = crop_need_m(w, h, S, vrows);
        if (need <= a.need_lo || need > a.need_hi) return;
    }
    int new_w, new_h, top, left;
    crop_geom(w, h, S, new_w, new_h, top, left);

    const CropLdsM L = crop_lds_m(S, a.ksx_cap, a.ksy_cap, a.wn_cap, vrows);
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
    uint8_t *planes = smem + L.planes;
    const int PST = L.pst;

    // ---- setup: level tables and both axes' coefficients (once per workgroup)
    if (tid < 192) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a.lut8 + tid * 4), LDS_PTR(lut8 + wave * 256), 4, 0, 0);
    if (tid < 256) {
#pragma unroll
        for (int i = 0; i < 3; i++)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a.normlut + i * 256 + tid), LDS_PTR(norm + i * 256 + wave * 64), 4, 0, 0);
    }
    if (tid < 64) ((unsigned *)(lut8 + 768))[tid] = 0x03020100u + 0x04040404u * (unsigned)tid;   // identity table
    for (int col = tid; col < S; col += NTM) {
        int xm, n;
        axis_coeffs(w, new_w, left + col, xm, n, kh + col, S);
        xmin[col] = xm;
        xcnt[col] = n;
    }
    for (int vr = NTM - 1 - tid; vr < vrows; vr += NTM) {
        int ym, n;
        axis_coeffs(h, new_h, top + band0 * BR + vr, ym, n, kv + vr, vrows);
        ymin[vr] = ym;
        ycnt[vr] = n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int cx0 = __builtin_amdgcn_readfirstlane(xmin[0]);
    const int wn = __builtin_amdgcn_readfirstlane(xmin[S - 1] + xcnt[S - 1]) - cx0;
    const int nblocks = (S + 15) >> 4;
    bool fits = wn <= a.wn_cap;
    for (int nb = 0; nb < nblocks; nb++) {   // uniform: every wave reaches the same verdict
        const int cl = min(16 * nb + 15, S - 1);
        const int lo = (__builtin_amdgcn_readfirstlane(xmin[16 * nb]) - cx0) & ~15;
        const int hi = __builtin_amdgcn_readfirstlane(xmin[cl] + xcnt[cl]) - cx0;
        fits = fits && hi - lo <= 64 * KB;
    }
    if (!fits) return;   // host sizing and crop_mfma_ok guarantee this never triggers

    // ---- this wave's column blocks: window start, coefficient digits (B operands), accumulator start
    const int fr = lane & 15, kg = lane >> 4;
    const int nb_mine = max(0, min(NBW, nblocks - wave * NBW));
    int x0[NBW], hstart[NBW];
    v4i bco[NBW][KB][3];
#pragma unroll
    for (int nbi = 0; nbi < NBW; nbi++) {
        const int nb = min(wave * NBW + nbi, nblocks - 1);
        const int colr = 16 * nb + fr;
        const bool valid = nbi < nb_mine && colr < S;
        const int col = min(colr, S - 1);
        x0[nbi] = (__builtin_amdgcn_readfirstlane(xmin[16 * nb]) - cx0) & ~15;
        const int xm = xmin[col] - cx0, n = valid ? xcnt[col] : 0;
        int ksum = 0;
        for (int t = 0; t < n; t++) ksum += kh[t * S + col];
        hstart[nbi] = (1 << (PREC - 1)) + 128 * ksum;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            unsigned pk[3][4];
#pragma unroll
            for (int q = 0; q < 4; q++) pk[0][q] = pk[1][q] = pk[2][q] = 0u;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int t = x0[nbi] + 64 * kb + 16 * kg + j - xm;
                const int cf = (t >= 0 && t < n) ? kh[t * S + col] : 0;
                int d0, d1, d2;
                byte_digits(cf, d0, d1, d2);
                pk[0][j >> 2] |= (unsigned)(d0 & 255) << (8 * (j & 3));
                pk[1][j >> 2] |= (unsigned)(d1 & 255) << (8 * (j & 3));
                pk[2][j >> 2] |= (unsigned)(d2 & 255) << (8 * (j & 3));
            }
#pragma unroll
            for (int d = 0; d < 3; d++) bco[nbi][kb][d] = (v4i){(int)pk[d][0], (int)pk[d][1], (int)pk[d][2], (int)pk[d][3]};
        }
    }
    // the blend pass wants level - 128: fold that into the tables (the DMA that brought them in was waited for above)
    if (tid < 256) ((unsigned *)lut8)[tid] ^= 0x80808080u;
    __syncthreads();   // the coefficient table is dead from here on: its bytes become the planes

    const int mode = c.mode;
    const uint8_t *fbase = a.frames + c.frame_off;
    const uint8_t *mabase = a.masks + c.mask_a_off;
    const uint8_t *mbbase = a.masks + c.mask_b_off;

    auto copy_row = [&](const uint8_t *src, int nbytes, uint8_t *dst) {
        const int lead = (int)((uintptr_t)src & 3);
        const uint8_t *s0 = src - lead;
        const int npiece = (lead + nbytes + 15) >> 4;
        for (int i = 0; i < npiece; i += 64)
            if (i + lane < npiece)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(s0 + (size_t)(i + lane) * 16), LDS_PTR(dst + i * 16), 16, 0, 0);
    };
    auto issue_chunk = [&](int cy, int cy_end) {
        const int nrows = min(MR, cy_end - cy);
        for (int r = wave; r < nrows; r += NTM / 64) {
            const size_t pix0 = (size_t)(c.y1 + cy + r) * c.frame_w + (c.x1 + cx0);
            copy_row(fbase + pix0 * 3, wn * 3, raw + r * L.raw_stride);
            if (mode != SGCLIP_CROP_PLAIN) {
                copy_row(mabase + pix0, wn, ma + r * L.m_stride);
                if (mode == SGCLIP_CROP_PAIR) copy_row(mbbase + pix0, wn, mb + r * L.m_stride);
            }
        }
    };
    auto band_rows = [&](int bl, int &r0, int &r1) {
        r0 = __builtin_amdgcn_readfirstlane(ymin[bl * BR]);
        r1 = __builtin_amdgcn_readfirstlane(ymin[bl * BR + BR - 1] + ycnt[bl * BR + BR - 1]);
    };

    const int patch = a.patch;
    const bool patches = a.layout == SGCLIP_LAYOUT_PATCHES;
    const int G = patches ? S / patch : 1;
    const bool wide = a.out_dtype == SGCLIP_DTYPE_BF16 && ((uintptr_t)a.out & 15) == 0 &&
                      (!patches || ((patch & 7) == 0 && (a.kpad & 7) == 0));
    const v4i zero4 = {0, 0, 0, 0};

    int ry0, ry1;
    band_rows(0, ry0, ry1);
    issue_chunk(ry0, ry1);
    for (int bl = 0; bl < bpw; bl++) {
        const int band = band0 + bl;
        // this lane's output rows in the column pass: 4 kg + e; its A-operand row there: fr
        v4i vsum[3][NBW];
        {
            const int kg = fresh(lane) >> 4;
            int vstart[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int orow = bl * BR + 4 * kg + e;
                const int n = ycnt[orow];
                int ks = 0;
                for (int t = 0; t < n; t++) ks += kv[t * vrows + orow];
                vstart[e] = (1 << (PREC - 1)) + 128 * ks;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int nbi = 0; nbi < NBW; nbi++) vsum[ch][nbi] = (v4i){vstart[0], vstart[1], vstart[2], vstart[3]};
        }
        const int a_ym = ymin[bl * BR + fr], a_yn = ycnt[bl * BR + fr];
        int nry0 = 0, nry1 = 0;
        if (bl + 1 < bpw) band_rows(bl + 1, nry0, nry1);

        for (int cy = ry0; cy < ry1; cy += MR) {
            const int nrows = min(MR, ry1 - cy);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- B: blend; 32 threads per row, an item = 4 pixels (rows past nrows keep stale bytes: the column
            // pass has no coefficient for them, and integers have no NaN).  The LDS is this pass's bottleneck, so the
            // 12 frame bytes and 4 + 4 mask bytes of an item come in as whole dwords (shifted into place by the row's
            // byte offset), every channel goes through exactly one table read -- the identity table where the
            // reference leaves the level alone -- and nothing branches per pixel.
            {
                // (thread indices through fresh(): addresses derived from them are recomputed here instead of living in
                // registers across the kernel, where they push the coefficient operands into scratch)
                const int bt = fresh(tid);
                const int r = bt >> 5;
                if (r < nrows) {
                    const size_t pix0 = (size_t)(c.y1 + cy + r) * c.frame_w + (c.x1 + cx0);
                    const unsigned lead = (unsigned)((uintptr_t)(fbase + pix0 * 3) & 3);
                    const unsigned ml = (unsigned)((uintptr_t)(mabase + pix0) & 3), bo = (unsigned)((uintptr_t)(mbbase + pix0) & 3);
                    const unsigned *raw32 = (const unsigned *)(raw + r * L.raw_stride);
                    const unsigned *ma32 = (const unsigned *)(ma + r * L.m_stride);
                    const unsigned *mb32 = (const unsigned *)(mb + r * L.m_stride);
                    unsigned *p0 = (unsigned *)(planes + (0 * MR + r) * PST);
                    unsigned *p1 = (unsigned *)(planes + (1 * MR + r) * PST);
                    unsigned *p2 = (unsigned *)(planes + (2 * MR + r) * PST);
                    const int nq = (wn + 3) >> 2;
                    auto blend = [&](auto mode_c) {
                        constexpr int MODE = decltype(mode_c)::value;
                        for (int xq = bt & 31; xq < nq; xq += 32) {
                            const unsigned w0 = raw32[3 * xq], w1 = raw32[3 * xq + 1], w2 = raw32[3 * xq + 2], w3 = raw32[3 * xq + 3];
                            // bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                            const unsigned q0 = __builtin_amdgcn_alignbyte(w1, w0, lead), q1 = __builtin_amdgcn_alignbyte(w2, w1, lead),
                                           q2 = __builtin_amdgcn_alignbyte(w3, w2, lead);
                            // v_perm_b32 picks result bytes from {hi, lo}: selector byte 0-3 = lo's bytes, 4-7 = hi's, 0x0c = zero
                            if (MODE == SGCLIP_CROP_PLAIN) {
                                const unsigned tr = __builtin_amdgcn_perm(q1, q0, 0x0c060300u), tg = __builtin_amdgcn_perm(q1, q0, 0x0c070401u),
                                               tb = __builtin_amdgcn_perm(q1, q0, 0x0c0c0502u);
                                p0[xq] = __builtin_amdgcn_perm(q2, tr, 0x05020100u) ^ 0x80808080u;
                                p1[xq] = __builtin_amdgcn_perm(q2, tg, 0x06020100u) ^ 0x80808080u;
                                p2[xq] = __builtin_amdgcn_perm(q2, tb, 0x07040100u) ^ 0x80808080u;
                                continue;
                            }
                            // Table number of every pixel and channel as packed bytes (0, 1: the object's two tables; pair:
                            // 0 outside both masks, 2 for the channel its mask colours, 3 = identity for the others), then
                            // one v_perm per lookup builds the address (table << 8) | level.  The tables hold level - 128
                            // already (see the setup), so the packed results are the plane bytes.
                            auto nonzero_bytes = [](unsigned x) { return ((((x | 0x80808080u) - 0x01010101u) | x) >> 7) & 0x01010101u; };
                            const unsigned a1 = nonzero_bytes(__builtin_amdgcn_alignbyte(ma32[xq + 1], ma32[xq], ml));
                            unsigned t4[3] = {a1, a1, a1};
                            if (MODE == SGCLIP_CROP_PAIR) {
                                const unsigned b1 = nonzero_bytes(__builtin_amdgcn_alignbyte(mb32[xq + 1], mb32[xq], bo));
                                const unsigned any3 = (a1 | b1) * 3u;
                                t4[0] = any3 - b1;
                                t4[1] = any3;
                                t4[2] = any3 - a1;
                            }
                            const unsigned qw[3] = {q0, q1, q2};
                            unsigned o[3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                unsigned lv[4];
#pragma unroll
                                for (int i = 0; i < 4; i++) {
                                    constexpr unsigned ZZ = 0x0c0c0000u;
                                    const int lin = 3 * i + ch;   // byte of the 12-byte item
                                    lv[i] = lut8[__builtin_amdgcn_perm(t4[ch], qw[lin >> 2], ZZ | ((4u + i) << 8) | (unsigned)(lin & 3))];
                                }
                                o[ch] = lv[0] | (lv[1] << 8) | (lv[2] << 16) | (lv[3] << 24);
                            }
                            p0[xq] = o[0];
                            p1[xq] = o[1];
                            p2[xq] = o[2];
                        }
                    };
                    if (mode == SGCLIP_CROP_PAIR)
                        blend(std::integral_constant<int, SGCLIP_CROP_PAIR>{});
                    else if (mode == SGCLIP_CROP_OBJECT)
                        blend(std::integral_constant<int, SGCLIP_CROP_OBJECT>{});
                    else
                        blend(std::integral_constant<int, SGCLIP_CROP_PLAIN>{});
                }
            }
            __syncthreads();
            // the raw bytes are consumed: refill them underneath C and D
            if (cy + MR < ry1)
                issue_chunk(cy + MR, ry1);
            else if (bl + 1 < bpw)
                issue_chunk(nry0, nry1);
            // ---- C + D on this wave's column blocks
            if (nb_mine > 0) {
                const int cl_lane = fresh(lane);
                const int fr = cl_lane & 15, kg = cl_lane >> 4;
                // column-pass A operand: digits of the coefficients of output row fr for chunk rows 4 kg .. 4 kg + 3
                unsigned av[3] = {0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int t = cy + 4 * kg + e - a_ym;
                    const int cf = (t >= 0 && t < a_yn) ? kv[t * vrows + bl * BR + fr] : 0;
                    int d0, d1, d2;
                    byte_digits(cf, d0, d1, d2);
                    av[0] |= (unsigned)(d0 & 255) << (8 * e);
                    av[1] |= (unsigned)(d1 & 255) << (8 * e);
                    av[2] |= (unsigned)(d2 & 255) << (8 * e);
                }
                const v4i av0 = {(int)av[0], 0, 0, 0}, av1 = {(int)av[1], 0, 0, 0}, av2 = {(int)av[2], 0, 0, 0};
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
#pragma unroll
                    for (int nbi = 0; nbi < NBW; nbi++) {
                        if (nbi >= nb_mine) continue;
                        const uint8_t *prow = planes + (ch * MR + fr) * PST + x0[nbi] + 16 * kg;
                        v4i s0 = {hstart[nbi], hstart[nbi], hstart[nbi], hstart[nbi]}, s1 = zero4, s2 = zero4;
#pragma unroll
                        for (int kb = 0; kb < KB; kb++) {
                            const v4i px = *(const v4i *)(prow + 64 * kb);
                            s0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(px, bco[nbi][kb][0], s0, 0, 0, 0);
                            s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(px, bco[nbi][kb][1], s1, 0, 0, 0);
                            s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(px, bco[nbi][kb][2], s2, 0, 0, 0);
                        }
                        // digits recombined Horner fashion: two shift-adds per value
                        const unsigned hp = pack_levels((((s2[0] << 8) + s1[0]) << 8) + s0[0], (((s2[1] << 8) + s1[1]) << 8) + s0[1],
                                                        (((s2[2] << 8) + s1[2]) << 8) + s0[2], (((s2[3] << 8) + s1[3]) << 8) + s0[3]);
                        const v4i hb = {(int)(hp ^ 0x80808080u), 0, 0, 0};
                        const v4i t0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av0, hb, zero4, 0, 0, 0);
                        const v4i t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av1, hb, zero4, 0, 0, 0);
                        const v4i t2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av2, hb, zero4, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; e++) vsum[ch][nbi][e] += (((t2[e] << 8) + t1[e]) << 8) + t0[e];
                    }
                }
            }
        }
        ry0 = nry0;
        ry1 = nry1;

        // ---- output of this band: lane = column 16 nb + fr, rows 4 kg + e
        const int ol = fresh(lane);
        const int fr = ol & 15, kg = ol >> 4;
        if (wide) {
            bf16_t *stage = (bf16_t *)(smem + L.planes);
            __syncthreads();   // every wave is done with the planes
#pragma unroll
            for (int nbi = 0; nbi < NBW; nbi++) {
                const int col = 16 * (wave * NBW + nbi) + fr;
                if (nbi < nb_mine && col < S) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            stage[((4 * kg + e) * 3 + ch) * S + col] = (bf16_t)norm[ch * 256 + clip8(vsum[ch][nbi][e])];
                }
            }
            __syncthreads();
            const int per_row = S >> 3;
            for (int v = fresh(tid); v < BR * 3 * per_row; v += NTM) {
                const int line = v / per_row, col = (v - line * per_row) << 3;
                const int rr = line / 3, ch = line - rr * 3;
                const int oy = band * BR + rr;
                size_t o;
                if (patches) {
                    const size_t prow = (size_t)crop_i * G * G + (size_t)(oy / patch) * G + col / patch;
                    o = prow * a.kpad + (size_t)ch * patch * patch + (oy % patch) * patch + col % patch;
                } else {
                    o = (((size_t)crop_i * 3 + ch) * S + oy) * S + col;
                }
                *(uint4 *)((bf16_t *)a.out + o) = *(const uint4 *)(stage + line * S + col);
            }
            continue;
        }
#pragma unroll
        for (int nbi = 0; nbi < NBW; nbi++) {
            const int col = 16 * (wave * NBW + nbi) + fr;
            if (nbi < nb_mine && col < S) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int oy = band * BR + 4 * kg + e;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float v = norm[ch * 256 + clip8(vsum[ch][nbi][e])];
                        size_t o;
                        if (patches) {
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
}

}  // namespace sgclip

using namespace sgclip;

// Bands per workgroup: as many as keep roughly four workgroups per CU slot in flight.
static int pick_bands_per_wg(int n_crops, int nbands) {
    const long target = 4L * 768;
    int best = 1;
    for (int b = 1; b <= nbands; b++)
        if (nbands % b == 0 && (long)n_crops * (nbands / b) >= target) best = b;
    return best;
}

extern "C" int sgclip_crop_preprocess(const uint8_t *frames_dev, size_t frames_bytes, const uint8_t *masks_dev,
                                      size_t masks_bytes, const sgclip_crop_t *crops, sgclip_crop_t *crops_dev,
                                      int crops_dev_ready, int n_crops, const uint8_t *luts, const float *norm_lut,
                                      void *lut_dev, int S, int layout, int patch, int kpad, int out_dtype,
                                      void *out_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crops == 0) return SGCLIP_OK;
    if (!frames_dev || !crops || !crops_dev || (!luts != !norm_lut) || !lut_dev || !out_dev || n_crops < 0) {
        set_error("sgclip_crop_preprocess: null argument");
        return SGCLIP_EINVAL;
    }
    if (S <= 0 || S % BR != 0 || S > 2 * COLS) {
        set_error("sgclip_crop_preprocess: S=%d must be a multiple of %d and <= %d", S, BR, 2 * COLS);
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
    if (((uintptr_t)frames_dev & 3) || ((uintptr_t)masks_dev & 3) || ((uintptr_t)lut_dev & 3)) {
        set_error("sgclip_crop_preprocess: frames_dev / masks_dev / lut_dev must be 4-byte aligned");
        return SGCLIP_EINVAL;
    }
    const int nbands = S / BR;
    int bpw = pick_bands_per_wg(n_crops, nbands);
    if (const char *e = getenv("SGCLIP_CROP_BANDS")) {
        const int v = atoi(e);
        if (v > 0 && nbands % v == 0) bpw = v;
    }
    // occupancy classes by LDS need: 4 / 3 / 2 / 1 workgroups per CU; class 4 = crops whose whole visible source span
    // does not fit even with one band per workgroup: they run over column segments
    static const int class_hi[4] = {40 * 1024, 53 * 1024, 80 * 1024, LDS_LIMIT};
    // the matrix-core kernel's classes: two workgroups per CU (its registers allow no more), or one
    static const int mclass_hi[2] = {80 * 1024, LDS_LIMIT};
    static int use_mfma_env = -1;
    if (use_mfma_env < 0) {
        const char *e = getenv("SGCLIP_CROP_MFMA");
        use_mfma_env = e ? atoi(e) : 1;
    }
    bool use_mfma = use_mfma_env != 0;
    int ksx[5], ksy[5], wnc[5], members[5];
    int mksx[2], mksy[2], mwnc[2], mmembers[2], mwin[2][2];   // mwin[class][windows - 1]: members by window count
    double algo_bytes = 0;  // per crop: 3hw frame + m*hw mask + out
    {   // the order hints: all zero, or a permutation of 1 .. n
        int nonzero = 0;
        for (int i = 0; i < n_crops; i++) nonzero += crops[i].order != 0;
        if (nonzero) {
            std::vector<unsigned char> seen((size_t)n_crops, 0);
            bool ok = nonzero == n_crops;
            for (int i = 0; ok && i < n_crops; i++) {
                const int o = crops[i].order;
                ok = o >= 1 && o <= n_crops && !seen[o - 1];
                if (ok) seen[o - 1] = 1;
            }
            if (!ok) {
                set_error("sgclip_crop_preprocess: the order fields must be all zero or a permutation of 1 .. %d", n_crops);
                return SGCLIP_EINVAL;
            }
        }
    }
    for (;;) {
        bool too_big = false;
        for (int k = 0; k < 5; k++) ksx[k] = ksy[k] = wnc[k] = 1, members[k] = 0;
        for (int k = 0; k < 2; k++) mksx[k] = mksy[k] = mwnc[k] = 1, mmembers[k] = mwin[k][0] = mwin[k][1] = 0;
        algo_bytes = 0;
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
            if (use_mfma && crop_mfma_ok(w, h, S, bpw * BR)) {
                const int km = crop_need_m(w, h, S, bpw * BR) > mclass_hi[0] ? 1 : 0;
                mksx[km] = std::max(mksx[km], axis_ksize(w, new_w));
                mksy[km] = std::max(mksy[km], axis_ksize(h, new_h));
                mwnc[km] = std::max(mwnc[km], x1 + n1 - x0);
                mmembers[km]++;
                mwin[km][crop_mfma_windows(w, h, S) - 1]++;
                continue;
            }
            const int need = crop_need(w, h, S, bpw * BR);
            int k = 0;
            if (need > LDS_LIMIT) {
                if (bpw > 1) {
                    too_big = true;
                    break;
                }
                k = 4;
            } else {
                while (need > class_hi[k]) k++;
            }
            ksx[k] = std::max(ksx[k], axis_ksize(w, new_w));
            ksy[k] = std::max(ksy[k], axis_ksize(h, new_h));
            wnc[k] = std::max(wnc[k], x1 + n1 - x0);
            members[k]++;
        }
        if (!too_big) {
            // a class's shared capacities can exceed its bound (taps from one crop, span from another)
            for (int k = 0; k < 4; k++)
                if (members[k] && crop_lds(S, ksx[k], ksy[k], wnc[k], bpw * BR).total > LDS_LIMIT) too_big = true;
        }
        bool m_too_big = false;
        for (int k = 0; k < 2; k++)
            if (mmembers[k] && crop_lds_m(S, mksx[k], mksy[k], mwnc[k], bpw * BR).total > LDS_LIMIT) m_too_big = true;
        if (!too_big && !m_too_big) break;
        if (bpw > 1) {
            bpw = 1;  // per-band coefficient tables are the smallest layout
            continue;
        }
        if (m_too_big && use_mfma) {
            use_mfma = false;   // shared class capacities beyond the LDS: everything on the vector kernel
            continue;
        }
        set_error("sgclip_crop_preprocess: an LDS class exceeds %d bytes", LDS_LIMIT);
        return SGCLIP_EINVAL;
    }
    // segmented class: the fewest column segments whose widest source span fits the LDS
    int nseg = 1;
    if (members[4]) {
        for (nseg = 2; nseg <= S; nseg++) {
            int span = 1;
            for (int i = 0; i < n_crops; i++) {
                const sgclip_crop_t &c = crops[i];
                const int w = c.x2 - c.x1, h = c.y2 - c.y1;
                if ((use_mfma && crop_mfma_ok(w, h, S, BR)) || crop_need(w, h, S, BR) <= LDS_LIMIT) continue;
                int new_w, new_h, top, left;
                crop_geom(w, h, S, new_w, new_h, top, left);
                for (int sg = 0; sg < nseg; sg++) {
                    int x0, n0, x1, n1;
                    axis_bounds(w, new_w, left + seg_begin(sg, S, nseg), x0, n0);
                    axis_bounds(w, new_w, left + seg_begin(sg + 1, S, nseg) - 1, x1, n1);
                    span = std::max(span, x1 + n1 - x0);
                }
            }
            wnc[4] = span;
            if (crop_lds(S, ksx[4], ksy[4], span, BR).total <= LDS_LIMIT) break;
        }
        if (nseg > S) {
            set_error("sgclip_crop_preprocess: a crop's resampling tables alone (%d x %d taps) exceed the %d bytes of LDS",
                      ksx[4], ksy[4], LDS_LIMIT);
            return SGCLIP_EINVAL;
        }
    }
    uint8_t *lut8_dev = (uint8_t *)lut_dev;
    float *norm_dev = (float *)(lut8_dev + 768);
    hipError_t e = hipSuccess;
    if (luts) {   // null tables: lut_dev already holds them
        e = hipMemcpyAsync(lut8_dev, luts, 768, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(norm_dev, norm_lut, 768 * 4, hipMemcpyHostToDevice, stream);
    }
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
    a.nbands = nbands;
    a.bands_per_wg = bpw;
    a.layout = layout;
    a.patch = patch;
    a.kpad = kpad;
    a.out_dtype = out_dtype;
    const dim3 grid((unsigned)n_crops * (nbands / bpw)), block(NT);
    KernelTimer timer(TK_CROP, algo_bytes, stream);
    a.nseg = 1;
    a.mfma_split = use_mfma ? 1 : 0;
    for (int k = 0; k < 2 && use_mfma; k++) {
        if (!mmembers[k]) continue;
        a.ksx_cap = mksx[k];
        a.ksy_cap = mksy[k];
        a.wn_cap = mwnc[k];
        a.need_lo = k ? mclass_hi[k - 1] : 0;
        a.need_hi = mclass_hi[k];
        const int lds_bytes = crop_lds_m(S, mksx[k], mksy[k], mwnc[k], bpw * BR).total;
        auto go = [&](auto kernel) {
            (void)lds_attr((const void *)kernel, LDS_LIMIT);
            hipLaunchKernelGGL(kernel, grid, dim3(NTM), lds_bytes, stream, a);
        };
        const int blocks = (S + 15) / 16;
        if (mwin[k][0]) {
            if (blocks <= 8) go(crop_mfma_kernel<1, 1>); else if (blocks <= 16) go(crop_mfma_kernel<2, 1>); else go(crop_mfma_kernel<3, 1>);
        }
        if (mwin[k][1]) {
            if (blocks <= 8) go(crop_mfma_kernel<1, 2>); else if (blocks <= 16) go(crop_mfma_kernel<2, 2>); else go(crop_mfma_kernel<3, 2>);
        }
    }
    for (int k = 0; k < 5; k++) {
        if (!members[k]) continue;
        a.ksx_cap = ksx[k];
        a.ksy_cap = ksy[k];
        a.wn_cap = wnc[k];
        a.need_lo = k ? class_hi[k - 1] : 0;
        a.need_hi = k < 4 ? class_hi[k] : INT32_MAX;
        const int lds_bytes = crop_lds(S, ksx[k], ksy[k], wnc[k], bpw * BR).total;
        auto go = [&](auto kernel) {
            (void)lds_attr((const void *)kernel, LDS_LIMIT);
            hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, a);
        };
        if (k == 4) {
            a.nseg = nseg;
            if (S <= COLS) go(crop_kernel<1, true>); else go(crop_kernel<2, true>);
        } else {
            if (S <= COLS) go(crop_kernel<1, false>); else go(crop_kernel<2, false>);
        }
    }
    return launch_check("sgclip_crop_preprocess");
}
