// Microbenchmark: issue cost of the vector instructions the softmax / GELU code paths lean on (cycles per wave64
// instruction on one SIMD), with 1, 2 and 4 waves per SIMD.  Every workgroup is 256 threads (one wave per SIMD);
// grid = 256 CUs x waves-per-SIMD.  Independent chains of 8 registers per lane, so latency is not what is measured.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
                if (OP == 1) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                if (OP == 2) v[i] = __builtin_amdgcn_rcpf(v[i]);
                if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
                if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 6) v[i] = __builtin_amdgcn_exp2f(v[i]) + v[(i + 3) & 7] * 0.25f;   // exp + fma mix
                if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
                if (OP == 8) asm volatile("v_rcp_f16 %0, %0" : "+v"(v[i]));
                if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(f32x2 *)&v[i & 6]) : "v"(*(f32x2 *)&v[(i + 2) & 6]));
                if (OP == 10) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 14) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (OP == 15) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            }
            if (OP == 4) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f32x2 t = {v[i], v[i + 1]};
                    t = t * (f32x2){1.0001f, 1.0001f};
                    v[i] = t[0];
                    v[i + 1] = t[1];
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, int per_iter) {
    float *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.001f);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.001f);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double instr_per_simd = (double)iters * per_iter * wps;
        printf("%-28s waves/SIMD %d: %8.3f ms  %.2f ns per wave instruction per SIMD (x clock GHz = cycles)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd);
    }
    (void)hipFree(out);
}

int main() {
    run<1>("v_fma_f32", 64);
    run<0>("v_exp_f32", 64);
    run<2>("v_rcp_f32", 64);
    run<3>("v_max3_f32", 64);
    run<4>("v_pk_mul_f32 (2 regs)", 32);
    run<5>("v_cvt_pk_bf16_f32", 64);
    run<6>("v_exp_f32 + v_fma_f32 pair", 64);
    run<7>("v_exp_f16", 64);
    run<8>("v_rcp_f16", 64);
    run<9>("v_pk_fma_f32", 64);
    run<10>("v_pk_add_f16", 64);
    run<11>("v_add_f32", 64);
    run<12>("v_max_f32", 64);
    run<13>("v_mul_f32", 64);
    run<14>("v_cvt_pk_f16_f32", 64);
    run<15>("v_pk_mul_f16", 64);
    return 0;
}
