// CLIP tower orchestration on the SGCLIP kernels: weight blob layout, workspace
// sizing, the vision encode (patch GEMM -> pre-LN -> L x [LN, QKV, attention,
// out-proj(+residual), LN, fc1(+quick-GELU), fc2(+residual)] -> post-LN on the
// class token -> projection) and the text encode (token+position embedding ->
// causal/padding-masked layers -> final LN at the eos position -> projection).
// Also the library's error plumbing.
//
// Arithmetic restated from transformers models/clip/modeling_clip.py
// (CLIPVisionModel.forward, CLIPTextModel.forward, CLIPEncoderLayer.forward),
// the code the reference reaches through get_image_features / get_text_features
// (laser/models/llava_clip_model_v3.py:63-82).
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace sgclip {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SGCLIP_ELAUNCH;
    }
    return SGCLIP_OK;
}

bool lds_attr(const void *fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({fn, dev});
    if (it != done.end() && it->second >= bytes) return true;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        // HIP's last-error state keeps the failure, so a launch_check() behind the skipped launch reports it too
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev, hipGetErrorString(e));
        return false;
    }
    done[{fn, dev}] = bytes;
    return true;
}

int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
    return n;
}

// ---- opt-in kernel timing ---------------------------------------------------
struct TimingRec { hipEvent_t a, b; int kind; double work; };
static bool g_timing = false;
static std::vector<TimingRec> g_recs;
static size_t g_used = 0;

KernelTimer::KernelTimer(int kind, double work, hipStream_t s) : slot(-1), stream(s) {
    if (!g_timing || g_used >= g_recs.size()) return;
    slot = (int)g_used++;
    g_recs[slot].kind = kind;
    g_recs[slot].work = work;
    (void)hipEventRecord(g_recs[slot].a, stream);
}
KernelTimer::~KernelTimer() {
    if (slot >= 0) (void)hipEventRecord(g_recs[slot].b, stream);
}

// x[img*T + 0][:] = class_embedding + position_embedding[0]
__global__ void cls_rows_kernel(float *x, const float *cls, const float *pos, int n_images, int T, int d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_images * d) return;
    const int img = (int)(i / d), j = (int)(i - (long long)img * d);
    x[(size_t)img * T * d + j] = cls[j] + pos[j];
}

// x[k*T + t][:] = token_embedding[ids[k,t]] + position_embedding[t]
__global__ void text_embed_kernel(float *x, const float *tok, const float *pos, const int32_t *ids, int T, int d,
                                  int vocab) {
    const int row = blockIdx.x;  // k*T + t
    const int t = row % T;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4 *tr = (const float4 *)(tok + (size_t)id * d);
    const float4 *pr = (const float4 *)(pos + (size_t)t * d);
    float4 *xr = (float4 *)(x + (size_t)row * d);
    for (int i = threadIdx.x; i < (d >> 2); i += blockDim.x) {
        const float4 a = tr[i], b = pr[i];
        xr[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// pooled row per text: first eos position, or argmax of the ids when eos_id == 2
// (transformers CLIPTextModel.forward, the two pooling branches)
// rows[k] = k * T: the class token of every image
__global__ void first_rows_kernel(int K, int T, int32_t *rows) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) rows[k] = k * T;
}

// dst[s][:] = src[rows[s]][:]
__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ rows, float *__restrict__ dst,
                                   int d4) {
    const float4 *s4 = (const float4 *)src + (size_t)rows[blockIdx.x] * d4;
    float4 *o4 = (float4 *)dst + (size_t)blockIdx.x * d4;
    for (int i = threadIdx.x; i < d4; i += blockDim.x) o4[i] = s4[i];
}

__global__ void eos_rows_kernel(const int32_t *ids, int K, int T, int eos_id, int32_t *rows) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int32_t *r = ids + (size_t)k * T;
    int pos = 0;
    if (eos_id == 2) {
        int best = r[0];
        for (int t = 1; t < T; t++)
            if (r[t] > best) { best = r[t]; pos = t; }
    } else {
        for (int t = 0; t < T; t++)
            if (r[t] == eos_id) { pos = t; break; }
    }
    rows[k] = k * T + pos;
}

struct Layout {
    int64_t off[SGCLIP_T_COUNT];     // offset of layer 0 (or the only instance)
    int64_t cnt[SGCLIP_T_COUNT];
    bool bf[SGCLIP_T_COUNT];
    bool per_layer[SGCLIP_T_COUNT];
    int64_t layer_stride_bf, layer_stride_f32;
    int64_t total_bf, total_f32;
};

static int vit_chunk() {
    static int v = 0;
    if (!v) {
        const char *e = getenv("SGCLIP_VIT_CHUNK_ROWS");
        v = e ? atoi(e) : 0;
        if (v <= 0) v = 262144;
    }
    return v;
}

static bool fused_qkv_attention() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SGCLIP_FUSED_QKV_ATTN");   // 0 = separate kernels (A/B measurements)
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

static bool build_layout(const sgclip_tower_desc_t *t, Layout &L) {
    if (!t || t->hidden <= 0 || t->hidden % 64 != 0 || t->heads * 64 != t->hidden || t->ffn % 64 != 0 ||
        t->layers <= 0 || t->tokens <= 0 || t->proj <= 0 || t->hidden > 1024) {
        set_error("tower descriptor: need hidden %% 64 == 0, hidden <= 1024, heads*64 == hidden, ffn %% 64 == 0");
        return false;
    }
    const int64_t d = t->hidden, f = t->ffn, E = t->proj, T = t->tokens;
    for (int i = 0; i < SGCLIP_T_COUNT; i++) { L.off[i] = -1; L.cnt[i] = 0; L.bf[i] = false; L.per_layer[i] = false; }
    int64_t ob = 0, of = 0;
    auto addb = [&](int id, int64_t n) { L.off[id] = ob; L.cnt[id] = n; L.bf[id] = true; ob += n; };
    auto addf = [&](int id, int64_t n) { L.off[id] = of; L.cnt[id] = n; L.bf[id] = false; of += n; };
    if (t->kind == 0) {
        if (t->patch <= 0 || t->image % t->patch != 0 || (t->image / t->patch) * (t->image / t->patch) + 1 != T) {
            set_error("vision descriptor: tokens must equal (image/patch)^2 + 1");
            return false;
        }
        addb(SGCLIP_T_PATCH_W, d * sgclip_patch_kpad(t->patch));
        addf(SGCLIP_T_CLS, d);
        addf(SGCLIP_T_POS, T * d);
        addf(SGCLIP_T_PRE_LN_W, d);
        addf(SGCLIP_T_PRE_LN_B, d);
    } else {
        if (t->vocab <= 0) { set_error("text descriptor: vocab must be positive"); return false; }
        addf(SGCLIP_T_TOK_EMB, (int64_t)t->vocab * d);
        addf(SGCLIP_T_POS, T * d);
    }
    const int64_t lb0 = ob, lf0 = of;
    addb(SGCLIP_T_QKV_W, 3 * d * d);
    addb(SGCLIP_T_O_W, d * d);
    addb(SGCLIP_T_FC1_W, f * d);
    addb(SGCLIP_T_FC2_W, d * f);
    addf(SGCLIP_T_LN1_W, d);
    addf(SGCLIP_T_LN1_B, d);
    addf(SGCLIP_T_QKV_B, 3 * d);
    addf(SGCLIP_T_O_B, d);
    addf(SGCLIP_T_LN2_W, d);
    addf(SGCLIP_T_LN2_B, d);
    addf(SGCLIP_T_FC1_B, f);
    addf(SGCLIP_T_FC2_B, d);
    L.layer_stride_bf = ob - lb0;
    L.layer_stride_f32 = of - lf0;
    const int per[] = {SGCLIP_T_QKV_W, SGCLIP_T_O_W, SGCLIP_T_FC1_W, SGCLIP_T_FC2_W, SGCLIP_T_LN1_W, SGCLIP_T_LN1_B,
                       SGCLIP_T_QKV_B, SGCLIP_T_O_B, SGCLIP_T_LN2_W, SGCLIP_T_LN2_B, SGCLIP_T_FC1_B, SGCLIP_T_FC2_B};
    for (int id : per) L.per_layer[id] = true;
    ob = lb0 + L.layer_stride_bf * t->layers;
    of = lf0 + L.layer_stride_f32 * t->layers;
    addf(SGCLIP_T_POST_LN_W, d);
    addf(SGCLIP_T_POST_LN_B, d);
    addf(SGCLIP_T_PROJ_W, E * d);
    L.total_bf = ob;
    L.total_f32 = of;
    return true;
}

struct TowerPtrs {
    const Layout &L;
    const bf16_t *wb;
    const float *wf;
    const bf16_t *B(int id, int layer = 0) const { return wb + L.off[id] + (L.per_layer[id] ? layer * L.layer_stride_bf : 0); }
    const float *F(int id, int layer = 0) const { return wf + L.off[id] + (L.per_layer[id] ? layer * L.layer_stride_f32 : 0); }
};

struct Work {
    float *x;       // [rows, d] residual stream
    bf16_t *h;      // [rows, d] LN output / attention output
    bf16_t *big;    // [rows, max(3d, f)] qkv / mlp hidden
    float *pooled;  // [seqs, d] final LayerNorm output, fp32
    int32_t *rows;  // [seqs] row index of the pooled token (class token / end-of-text token)
    float *xp;      // [seqs, d] residual stream of the pooled rows through the last layer
    bf16_t *hp;     // [seqs, d] their attention output / LN output
    size_t total;
};

static Work carve(const sgclip_tower_desc_t *t, void *base, int64_t seqs, int64_t T) {
    const int64_t d = t->hidden, wide = std::max<int64_t>(3 * d, t->ffn), rows = seqs * T;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t p = o; o += align_up(bytes, 256); return p; };
    Work w;
    char *b = (char *)base;
    w.x = (float *)(b + take((size_t)rows * d * 4));
    w.h = (bf16_t *)(b + take((size_t)rows * d * 2));
    w.big = (bf16_t *)(b + take((size_t)rows * wide * 2));
    w.pooled = (float *)(b + take((size_t)seqs * d * 4));
    w.rows = (int32_t *)(b + take((size_t)seqs * 4));
    w.xp = (float *)(b + take((size_t)seqs * d * 4));
    w.hp = (bf16_t *)(b + take((size_t)seqs * d * 2));
    w.total = o;
    return w;
}

#define CHECK(call)                   \
    do {                              \
        int _rc = (call);             \
        if (_rc != SGCLIP_OK) return _rc; \
    } while (0)

// Only the pooled token of every sequence (w.rows) leaves the tower, so the last layer computes
// keys and values for all rows but attention output, out-projection and the MLP for those rows
// alone; the result is the pooled rows' residual stream in w.xp.  pooled_first: the pooled row is row 0 of every
// sequence (image towers), which lets the last layer's query projection run on those rows alone.
static int run_layers(const sgclip_tower_desc_t *t, const TowerPtrs &P, const Work &w, int seqs, int T, int causal,
                      const int32_t *key_mask, bool pooled_first, void *stream) {
    const int d = t->hidden, f = t->ffn, M = seqs * T;
    const int last = t->layers - 1;
    // short unmasked sequences in bulk (the image towers at T = 50): q/k/v projection and attention in one kernel,
    // attention output into the front of `big` (the layer-norm output in `h` is its input)
    const bool fused = fused_qkv_attention() && !causal && !key_mask && sgclip_qkv_attention_supported(T, t->heads) &&
                       (long long)ceil_div(seqs, 256 / T) * t->heads >= 512;
    for (int l = 0; l < last; l++) {
        CHECK(sgclip_layernorm(w.x, d, nullptr, M, d, P.F(SGCLIP_T_LN1_W, l), P.F(SGCLIP_T_LN1_B, l), t->ln_eps,
                               SGCLIP_DTYPE_BF16, w.h, stream));
        const bf16_t *attn_out = w.h;
        if (fused) {
            CHECK(sgclip_qkv_attention(w.h, d, P.B(SGCLIP_T_QKV_W, l), d, P.F(SGCLIP_T_QKV_B, l), w.big, seqs, T, t->heads,
                                       stream));
            attn_out = w.big;
        } else {
            CHECK(sgclip_gemm_bf16(w.h, d, P.B(SGCLIP_T_QKV_W, l), d, w.big, 3 * d, M, 3 * d, d, SGCLIP_EPI_BIAS_BF16,
                                   P.F(SGCLIP_T_QKV_B, l), nullptr, 0, stream));
            CHECK(sgclip_attention(w.big, w.h, seqs, T, t->heads, causal, key_mask, stream));
        }
        CHECK(sgclip_gemm_bf16(attn_out, d, P.B(SGCLIP_T_O_W, l), d, w.x, d, M, d, d, SGCLIP_EPI_BIAS_RESID_F32,
                               P.F(SGCLIP_T_O_B, l), nullptr, 0, stream));
        CHECK(sgclip_layernorm(w.x, d, nullptr, M, d, P.F(SGCLIP_T_LN2_W, l), P.F(SGCLIP_T_LN2_B, l), t->ln_eps,
                               SGCLIP_DTYPE_BF16, w.h, stream));
        CHECK(sgclip_gemm_bf16(w.h, d, P.B(SGCLIP_T_FC1_W, l), d, w.big, f, M, f, d, SGCLIP_EPI_BIAS_GELU_BF16,
                               P.F(SGCLIP_T_FC1_B, l), nullptr, 0, stream));
        CHECK(sgclip_gemm_bf16(w.big, f, P.B(SGCLIP_T_FC2_W, l), f, w.x, d, M, d, f, SGCLIP_EPI_BIAS_RESID_F32,
                               P.F(SGCLIP_T_FC2_B, l), nullptr, 0, stream));
    }
    {
        const int l = last;
        hipStream_t hs = (hipStream_t)stream;
        CHECK(sgclip_layernorm(w.x, d, nullptr, M, d, P.F(SGCLIP_T_LN1_W, l), P.F(SGCLIP_T_LN1_B, l), t->ln_eps,
                               SGCLIP_DTYPE_BF16, w.h, stream));
        if (pooled_first) {
            // keys and values of every row, queries of the pooled rows only (row 0 of every sequence: strided operands)
            CHECK(sgclip_gemm_bf16(w.h, d, P.B(SGCLIP_T_QKV_W, l) + (size_t)d * d, d, w.big + d, 3 * d, M, 2 * d, d,
                                   SGCLIP_EPI_BIAS_BF16, P.F(SGCLIP_T_QKV_B, l) + d, nullptr, 0, stream));
            CHECK(sgclip_gemm_bf16(w.h, T * d, P.B(SGCLIP_T_QKV_W, l), d, w.big, T * 3 * d, seqs, d, d, SGCLIP_EPI_BIAS_BF16,
                                   P.F(SGCLIP_T_QKV_B, l), nullptr, 0, stream));
        } else {
            CHECK(sgclip_gemm_bf16(w.h, d, P.B(SGCLIP_T_QKV_W, l), d, w.big, 3 * d, M, 3 * d, d, SGCLIP_EPI_BIAS_BF16,
                                   P.F(SGCLIP_T_QKV_B, l), nullptr, 0, stream));
        }
        CHECK(sgclip_attention_pooled(w.big, w.rows, w.hp, seqs, T, t->heads, causal, key_mask, stream));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)seqs), dim3(128), 0, hs, w.x, w.rows, w.xp, d / 4);
        CHECK(launch_check("gather_rows_kernel"));
        CHECK(sgclip_gemm_bf16(w.hp, d, P.B(SGCLIP_T_O_W, l), d, w.xp, d, seqs, d, d, SGCLIP_EPI_BIAS_RESID_F32,
                               P.F(SGCLIP_T_O_B, l), nullptr, 0, stream));
        CHECK(sgclip_layernorm(w.xp, d, nullptr, seqs, d, P.F(SGCLIP_T_LN2_W, l), P.F(SGCLIP_T_LN2_B, l), t->ln_eps,
                               SGCLIP_DTYPE_BF16, w.hp, stream));
        CHECK(sgclip_gemm_bf16(w.hp, d, P.B(SGCLIP_T_FC1_W, l), d, w.big, f, seqs, f, d, SGCLIP_EPI_BIAS_GELU_BF16,
                               P.F(SGCLIP_T_FC1_B, l), nullptr, 0, stream));
        CHECK(sgclip_gemm_bf16(w.big, f, P.B(SGCLIP_T_FC2_W, l), f, w.xp, d, seqs, d, f, SGCLIP_EPI_BIAS_RESID_F32,
                               P.F(SGCLIP_T_FC2_B, l), nullptr, 0, stream));
    }
    return SGCLIP_OK;
}

}  // namespace sgclip

using namespace sgclip;

extern "C" int sgclip_timing_enable(int on, int max_records) {
    if (on && g_recs.empty()) {
        g_recs.resize(max_records > 0 ? max_records : 65536);
        for (auto &r : g_recs) {
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
                set_error("sgclip_timing_enable: hipEventCreate failed");
                return SGCLIP_ELAUNCH;
            }
        }
    }
    g_timing = on != 0;
    if (on) g_used = 0;
    return SGCLIP_OK;
}

extern "C" int sgclip_timing_read(int kind, double *total_ms, int64_t *launches, double *work) {
    double ms = 0, w = 0;
    int64_t n = 0;
    for (size_t i = 0; i < g_used; i++) {
        if (g_recs[i].kind != kind) continue;
        if (hipEventSynchronize(g_recs[i].b) != hipSuccess) {
            set_error("sgclip_timing_read: event synchronize failed");
            return SGCLIP_ELAUNCH;
        }
        float t = 0;
        (void)hipEventElapsedTime(&t, g_recs[i].a, g_recs[i].b);
        ms += t;
        w += g_recs[i].work;
        n++;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (work) *work = w;
    return SGCLIP_OK;
}

extern "C" int sgclip_abi_version(void) { return 1; }
extern "C" const char *sgclip_last_error(void) { return g_err; }
extern "C" int sgclip_patch_kpad(int patch) { return (3 * patch * patch + 63) / 64 * 64; }

extern "C" int sgclip_tower_layout(const sgclip_tower_desc_t *desc, int id, int layer, int64_t *offset, int64_t *count,
                                   int *is_bf16) {
    Layout L;
    if (!build_layout(desc, L)) return SGCLIP_EINVAL;
    if (id == -1) {
        if (offset) *offset = L.total_bf;
        if (count) *count = L.total_f32;
        if (is_bf16) *is_bf16 = 0;
        return SGCLIP_OK;
    }
    if (id < 0 || id >= SGCLIP_T_COUNT || L.off[id] < 0 || layer < 0 || (L.per_layer[id] && layer >= desc->layers) ||
        (!L.per_layer[id] && layer != 0)) {
        set_error("sgclip_tower_layout: tensor %d / layer %d not part of this tower", id, layer);
        return SGCLIP_EINVAL;
    }
    const int64_t stride = L.bf[id] ? L.layer_stride_bf : L.layer_stride_f32;
    if (offset) *offset = L.off[id] + (L.per_layer[id] ? layer * stride : 0);
    if (count) *count = L.cnt[id];
    if (is_bf16) *is_bf16 = L.bf[id] ? 1 : 0;
    return SGCLIP_OK;
}

static int vit_chunk_images(const sgclip_tower_desc_t *desc, int n_images) {
    int c = std::max(1, vit_chunk() / desc->tokens);
    return std::min(c, std::max(n_images, 1));
}

extern "C" size_t sgclip_vit_workspace_bytes(const sgclip_tower_desc_t *desc, int n_images) {
    Layout L;
    if (!build_layout(desc, L) || desc->kind != 0 || n_images <= 0) return 0;
    return carve(desc, nullptr, vit_chunk_images(desc, n_images), desc->tokens).total;
}

extern "C" int sgclip_vit_encode(const sgclip_tower_desc_t *desc, const void *w_bf16_dev, const float *w_f32_dev,
                                 const void *patches_dev, int n_images, void *workspace_dev, size_t workspace_bytes,
                                 float *out_dev, void *stream) {
    Layout L;
    if (!build_layout(desc, L)) return SGCLIP_EINVAL;
    if (desc->kind != 0) { set_error("sgclip_vit_encode: descriptor is not a vision tower"); return SGCLIP_EINVAL; }
    if (n_images == 0) return SGCLIP_OK;
    if (!w_bf16_dev || !w_f32_dev || !patches_dev || !workspace_dev || !out_dev || n_images < 0) {
        set_error("sgclip_vit_encode: null argument");
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)workspace_dev & 255) || ((uintptr_t)w_bf16_dev & 15) || ((uintptr_t)w_f32_dev & 15)) {
        set_error("sgclip_vit_encode: workspace must be 256-byte aligned, weight blobs 16-byte aligned");
        return SGCLIP_EINVAL;
    }
    const int chunk = vit_chunk_images(desc, n_images);
    const int T = desc->tokens, P = T - 1, d = desc->hidden, E = desc->proj;
    const int kpad = sgclip_patch_kpad(desc->patch);
    const Work w = carve(desc, workspace_dev, chunk, T);
    if (workspace_bytes < w.total) {
        set_error("sgclip_vit_encode: workspace %zu < required %zu bytes", workspace_bytes, w.total);
        return SGCLIP_ENOSPC;
    }
    const TowerPtrs Pp{L, (const bf16_t *)w_bf16_dev, w_f32_dev};
    hipStream_t hs = (hipStream_t)stream;
    for (int i0 = 0; i0 < n_images; i0 += chunk) {
        const int nc = std::min(chunk, n_images - i0);
        const bf16_t *patches = (const bf16_t *)patches_dev + (size_t)i0 * P * kpad;
        CHECK(sgclip_gemm_bf16(patches, kpad, Pp.B(SGCLIP_T_PATCH_W), kpad, w.x, d, nc * P, d, kpad, SGCLIP_EPI_PATCH_F32,
                               nullptr, Pp.F(SGCLIP_T_POS), P, stream));
        hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)(((long long)nc * d + 255) / 256)), dim3(256), 0, hs, w.x,
                           Pp.F(SGCLIP_T_CLS), Pp.F(SGCLIP_T_POS), nc, T, d);
        CHECK(launch_check("cls_rows_kernel"));
        CHECK(sgclip_layernorm(w.x, d, nullptr, nc * T, d, Pp.F(SGCLIP_T_PRE_LN_W), Pp.F(SGCLIP_T_PRE_LN_B), desc->ln_eps,
                               SGCLIP_DTYPE_F32, w.x, stream));
        hipLaunchKernelGGL(first_rows_kernel, dim3(ceil_div(nc, 256)), dim3(256), 0, hs, nc, T, w.rows);
        CHECK(launch_check("first_rows_kernel"));
        CHECK(run_layers(desc, Pp, w, nc, T, 0, nullptr, true, stream));
        CHECK(sgclip_layernorm(w.xp, d, nullptr, nc, d, Pp.F(SGCLIP_T_POST_LN_W), Pp.F(SGCLIP_T_POST_LN_B),
                               desc->ln_eps, SGCLIP_DTYPE_F32, w.pooled, stream));
        CHECK(sgclip_linear_f32(w.pooled, Pp.F(SGCLIP_T_PROJ_W), nc, E, d, out_dev + (size_t)i0 * E, stream));
    }
    return SGCLIP_OK;
}

extern "C" size_t sgclip_text_workspace_bytes(const sgclip_tower_desc_t *desc, int n_texts, int seq_len) {
    Layout L;
    if (!build_layout(desc, L) || desc->kind != 1 || n_texts <= 0 || seq_len <= 0) return 0;
    return carve(desc, nullptr, n_texts, seq_len).total;
}

extern "C" int sgclip_text_encode(const sgclip_tower_desc_t *desc, const void *w_bf16_dev, const float *w_f32_dev,
                                  const int32_t *ids_dev, const int32_t *attn_mask_dev, int n_texts, int seq_len,
                                  void *workspace_dev, size_t workspace_bytes, float *out_dev, void *stream) {
    Layout L;
    if (!build_layout(desc, L)) return SGCLIP_EINVAL;
    if (desc->kind != 1) { set_error("sgclip_text_encode: descriptor is not a text tower"); return SGCLIP_EINVAL; }
    if (n_texts == 0) return SGCLIP_OK;
    if (!w_bf16_dev || !w_f32_dev || !ids_dev || !workspace_dev || !out_dev || n_texts < 0 || seq_len <= 0 ||
        seq_len > desc->tokens) {
        set_error("sgclip_text_encode: bad arguments (seq_len %d, max positions %d)", seq_len, desc->tokens);
        return SGCLIP_EINVAL;
    }
    if (((uintptr_t)workspace_dev & 255) || ((uintptr_t)w_bf16_dev & 15) || ((uintptr_t)w_f32_dev & 15)) {
        set_error("sgclip_text_encode: workspace must be 256-byte aligned, weight blobs 16-byte aligned");
        return SGCLIP_EINVAL;
    }
    const int d = desc->hidden, E = desc->proj, T = seq_len;
    const Work w = carve(desc, workspace_dev, n_texts, T);
    if (workspace_bytes < w.total) {
        set_error("sgclip_text_encode: workspace %zu < required %zu bytes", workspace_bytes, w.total);
        return SGCLIP_ENOSPC;
    }
    const TowerPtrs Pp{L, (const bf16_t *)w_bf16_dev, w_f32_dev};
    hipStream_t hs = (hipStream_t)stream;
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)(n_texts * T)), dim3(128), 0, hs, w.x, Pp.F(SGCLIP_T_TOK_EMB),
                       Pp.F(SGCLIP_T_POS), ids_dev, T, d, desc->vocab);
    hipLaunchKernelGGL(eos_rows_kernel, dim3(ceil_div(n_texts, 64)), dim3(64), 0, hs, ids_dev, n_texts, T, desc->eos_id,
                       w.rows);
    CHECK(launch_check("text_embed_kernel"));
    CHECK(run_layers(desc, Pp, w, n_texts, T, 1, attn_mask_dev, false, stream));
    CHECK(sgclip_layernorm(w.xp, d, nullptr, n_texts, d, Pp.F(SGCLIP_T_POST_LN_W), Pp.F(SGCLIP_T_POST_LN_B), desc->ln_eps,
                           SGCLIP_DTYPE_F32, w.pooled, stream));
    CHECK(sgclip_linear_f32(w.pooled, Pp.F(SGCLIP_T_PROJ_W), n_texts, E, d, out_dev, stream));
    return SGCLIP_OK;
}
