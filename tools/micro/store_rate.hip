// Microbenchmark: per-CU store rate, 16 B per lane, rows of 512 contiguous bytes (a 256-column bf16 tile row),
// streaming over a large buffer (HBM) against rewriting a small per-workgroup region (stays in L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void store_sweep(uint4 *out, size_t region_u4, size_t stride_u4, int iters, int row_u4, int ld_u4) {
    // region: this workgroup's window (in uint4 units); each iteration writes 256 rows x row_u4 uint4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    uint4 *base = out + (size_t)blockIdx.x * stride_u4;
    size_t off = 0;
    const int lanes_per_row = row_u4;            // e.g. 32 lanes cover a 512-byte row
    const int rows_per_inst = 64 / lanes_per_row;
    for (int it = 0; it < iters; it++) {
        // the wave owns 32 rows of the 256-row tile
#pragma unroll 4
        for (int r = 0; r < 32; r += rows_per_inst) {
            const int row = wave * 32 + r + lane / lanes_per_row;
            base[off + (size_t)row * ld_u4 + lane % lanes_per_row] = v;
        }
        off += (size_t)256 * ld_u4;
        if (off + (size_t)256 * ld_u4 > region_u4) off = 0;
    }
}
int main() {
    const size_t total = 3ull << 30;
    uint4 *buf; (void)hipMalloc(&buf, total);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 72;
    for (int grid : {256}) {
    printf("grid %d\n", grid);
    struct Case { const char *name; size_t region_bytes; int row_bytes; int ld_bytes; } cases[] = {
        {"stream, 512 B rows (ld 4608)", total / 256, 512, 4608},
        {"stream, 512 B rows (dense)", total / 256, 512, 512},
        {"rewrite 128 KiB per WG (L2)", 256 * 512, 512, 512},
        {"rewrite 256 KiB per WG (64 MiB at 256 WGs)", 512 * 512, 512, 512},
        {"rewrite 512 KiB per WG (128 MiB at 256 WGs)", 1024 * 512, 512, 512},
        {"rewrite 768 KiB per WG (192 MiB at 256 WGs)", 1536 * 512, 512, 512},
        {"stream, 1024 B rows (fp32 tile, ld 3072)", total / 256, 1024, 3072},
    };
    for (auto &c : cases) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(store_sweep, dim3(grid), dim3(512), 0, 0, buf, c.region_bytes / 16, (total / 256) / 16, iters,
                               c.row_bytes / 16, c.ld_bytes / 16);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        double bytes = (double)grid * iters * 256.0 * c.row_bytes;
        printf("%-44s %.3f ms  %.1f GB/s per CU  %.2f TB/s chip\n", c.name, ms, bytes / grid / ms / 1e6, bytes / ms / 1e9);
    }
    }
    return 0;
}
