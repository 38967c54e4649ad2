// Microbenchmark: per-CU rate of L2-resident operand fetches, LDS-DMA against register loads.
// Every workgroup sweeps the same 2 MiB buffer (stays in each XCD's L2) in the piece shapes a GEMM stage uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// mode 0: LDS-DMA 16 B/lane, each 8 lanes cover 128 contiguous bytes of a row (row stride ROWB)
// mode 1: same addresses, global_load_dwordx4 to registers (summed so the loads are kept)
// seg = contiguous bytes per row piece (64 or 128): lanes per row = seg / 16
template <int MODE>
__global__ __launch_bounds__(512) void sweep(const unsigned char *buf, size_t bytes, int rowb, int seg, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = seg / 16;                 // lanes per row
    const int rows_per_inst = 64 / lpr;
    float4 accv = make_float4(0, 0, 0, 0);
    unsigned nrows = 1;
    while ((size_t)nrows * 2 * rowb <= bytes) nrows *= 2;
    const unsigned mask = nrows - 1;
    unsigned row = (blockIdx.x * 977u + wave * rows_per_inst + lane / lpr) & mask;
    const int lcol = (lane % lpr) * 16;
    int koff = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned char *src = buf + (size_t)row * rowb + koff + lcol;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds + (wave * 4 + q) * 1024), 16, 0, 0);
            } else {
                const float4 v = *(const float4 *)src;
                accv.x += v.x; accv.y += v.y; accv.z += v.z; accv.w += v.w;
            }
            row = (row + 8 * rows_per_inst) & mask;
        }
        if ((it & 7) == 7) { koff += seg; koff = koff >= rowb ? 0 : koff; }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accv.x + accv.y + accv.z + accv.w == 12345.f) sink[0] = 1.f;
}

int main() {
    const size_t bytes = 2u << 20;
    unsigned char *buf; float *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, grid = 256;
    for (int seg : {64, 128}) for (int rowb : {1536, 6144}) for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(grid), dim3(512), 32768, 0, buf, bytes, rowb, seg, iters, sink);
            else hipLaunchKernelGGL(sweep<1>, dim3(grid), dim3(512), 32768, 0, buf, bytes, rowb, seg, iters, sink);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double total = (double)grid * 8 * 4 * 1024.0 * iters;
        printf("seg %3d rowb %4d mode %s: %.3f ms  %.1f GB/s per CU  %.1f TB/s chip\n", seg, rowb, mode ? "regs" : "dma ", ms,
               total / grid / ms / 1e6, total / ms / 1e9);
    }
    return 0;
}
