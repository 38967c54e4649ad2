// Which K index does byte j of lane l's 16-byte operand of v_mfma_i32_16x16x64_i8 hold?  Feeds one-hot operands and
// prints the map, then checks a random product against the host under that map.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_i8_layout mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void mm(const int8_t *a /*[64 lanes][16]*/, const int8_t *b, int *c /*[64][4]*/) {
    const int l = threadIdx.x;
    v4i av = *(const v4i *)(a + l * 16), bv = *(const v4i *)(b + l * 16), cv = {0, 0, 0, 0};
    cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, cv, 0, 0, 0);
    *(v4i *)(c + l * 4) = cv;
}

int main() {
    int8_t *a, *b;
    int *c;
    (void)hipHostMalloc((void **)&a, 1024);
    (void)hipHostMalloc((void **)&b, 1024);
    (void)hipHostMalloc((void **)&c, 1024);
    // K index of (lane, byte) of A: set A one-hot at (lane la, byte ja) = 1, B all ones in a pattern that encodes k:
    // B lane lb byte jb = value unique per (lb >> 4, jb) -> C tells which B k-slot met it.  Simpler: brute force both maps.
    int kmapA[64][16], kmapB[64][16];
    // first find rows/cols: A one-hot with all B = 1 -> which C entries light up gives A's row; (and K irrelevant)
    for (int which = 0; which < 2; which++) {
        for (int l0 = 0; l0 < 64; l0++)
            for (int j0 = 0; j0 < 16; j0++) {
                // operand under test one-hot at (l0, j0); the other operand: byte value = 1 + (k-slot id guess-free code)
                for (int i = 0; i < 1024; i++) a[i] = b[i] = 0;
                int8_t *t = which == 0 ? a : b, *o = which == 0 ? b : a;
                t[l0 * 16 + j0] = 1;
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 16; j++) o[l * 16 + j] = (int8_t)(1 + (l >> 4) * 16 + j);   // 1..64: code of (lane group, byte)
                mm<<<1, 64>>>(a, b, c);
                (void)hipDeviceSynchronize();
                // every nonzero C entry carries the code of the other operand's slot that shares this K
                int code = 0;
                for (int i = 0; i < 256; i++)
                    if (c[i]) code = c[i];
                (which == 0 ? kmapA : kmapB)[l0][j0] = code - 1;   // = (group, byte) of the partner holding the same k
            }
    }
    printf("A (lane group g, byte j) pairs with B slot (g', j'):\n");
    int same = 1;
    for (int g = 0; g < 4; g++) {
        for (int j = 0; j < 16; j++) {
            int code = kmapA[g * 16][j];
            printf(" %d:%d", code >> 4, code & 15);
            if (code != g * 16 + j) same = 0;
        }
        printf("\n");
    }
    printf("A and B use the same (group, byte) -> k map: %s\n", same ? "yes" : "NO");
    // rows / columns and output layout under the natural assumption
    srand(1);
    int8_t A[16][64], B[64][16];
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 64; k++) A[i][k] = (int8_t)(rand() % 256 - 128), B[k][i] = (int8_t)(rand() % 256 - 128);
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
            a[l * 16 + j] = A[l & 15][(l >> 4) * 16 + j];
            b[l * 16 + j] = B[(l >> 4) * 16 + j][l & 15];
        }
    mm<<<1, 64>>>(a, b, c);
    (void)hipDeviceSynchronize();
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int e = 0; e < 4; e++) {
            int want = 0;
            const int row = (l >> 4) * 4 + e, col = l & 15;
            for (int k = 0; k < 64; k++) want += (int)A[row][k] * (int)B[k][col];
            if (want != c[l * 4 + e]) bad++;
        }
    printf("A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], C[4(l>>4)+e][l&15]: %d mismatches of 256\n", bad);
    return 0;
}
