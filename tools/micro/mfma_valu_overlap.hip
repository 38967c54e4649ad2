// Microbenchmark: does vector work run beside the matrix pipe?  One loop iteration = NM independent 32x32x16 bf16 MFMAs
// and NV independent v_fma_f32 (or v_exp_f32), interleaved in one instruction stream, at 1, 2 and 4 waves per SIMD.
// Reported: ns per iteration per SIMD next to the two parts run alone (sum = no overlap, max = full overlap).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int TRANS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int g = 0; g < 16; g++) acc[i][g] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed * j); }
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); m++) {
            if (NM > 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / (NM > 0 ? NM : 1); q++) {
                const int i = (m * (NV / (NM > 0 ? NM : 1)) + q) & 15;
                if (TRANS) v[i] = __builtin_amdgcn_exp2f(v[i]);
                else v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += acc[i][0];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NM, int NV, int TRANS>
static double run(int wps) {
    static float *out = nullptr;
    if (!out) (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<NM, NV, TRANS>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV, TRANS>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.001f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters / wps;   // ns per wave-iteration per SIMD
}

template <int NM, int NV, int TRANS>
static void report(const char *name) {
    for (int wps : {1, 2, 4}) {
        const double both = run<NM, NV, TRANS>(wps), m = run<NM, 0, TRANS>(wps), v = run<0, NV, TRANS>(wps);
        printf("%-34s waves/SIMD %d: both %7.1f ns  mfma alone %7.1f  valu alone %7.1f  (sum %7.1f, max %7.1f)\n", name, wps, both, m,
               v, m + v, m > v ? m : v);
    }
}

int main() {
    report<8, 64, 0>("8 MFMA + 64 v_fma_f32");
    report<8, 128, 0>("8 MFMA + 128 v_fma_f32");
    report<8, 32, 1>("8 MFMA + 32 v_exp_f32");
    report<8, 64, 1>("8 MFMA + 64 v_exp_f32");
    return 0;
}
