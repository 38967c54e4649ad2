// What v_ashr_pk_u8_i32 writes: two 32-bit values shifted right, saturated to bytes and packed into HALF of the
// destination (the low half, or the high half with op_sel:[0,0,0,1]); the other half keeps what the register held.
//   hipcc --offload-arch=gfx950 -O2 -o ashr_pk ashr_pk.hip && ./ashr_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k(const int *a, unsigned *o, unsigned *o2) {
    const int t = threadIdx.x;
    int a0 = a[t * 4], a1 = a[t * 4 + 1], a2 = a[t * 4 + 2], a3 = a[t * 4 + 3];
    unsigned d = 0;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(d) : "v"(a0), "v"(a1));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22 op_sel:[0,0,0,1]" : "+v"(d) : "v"(a2), "v"(a3));
    o[t] = d;
    unsigned e = 0xdeadbeefu;   // does the low-half form keep the upper half?
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(e) : "v"(a0), "v"(a1));
    o2[t] = e;
}

static unsigned clip8(int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); }

int main() {
    int *a;
    unsigned *o, *o2;
    (void)hipHostMalloc((void **)&a, 64 * 16);
    (void)hipHostMalloc((void **)&o, 256);
    (void)hipHostMalloc((void **)&o2, 256);
    srand(2);
    for (int i = 0; i < 256; i++) a[i] = (int)((unsigned)rand() * 2654435761u);   // all 32 bits vary: negatives, > 255 << 22
    for (int i = 0; i < 16; i++) a[i] = (i * 37 - 40) << 22 | (i * 999);
    k<<<1, 64>>>(a, o, o2);
    (void)hipDeviceSynchronize();
    int bad = 0, keep = 0;
    for (int t = 0; t < 64; t++) {
        const unsigned want = clip8(a[t * 4]) | clip8(a[t * 4 + 1]) << 8 | clip8(a[t * 4 + 2]) << 16 | clip8(a[t * 4 + 3]) << 24;
        if (o[t] != want) { if (bad < 4) printf("lane %d: got %08x want %08x\n", t, o[t], want); bad++; }
        if ((o2[t] >> 16) == 0xdead) keep++;
    }
    printf("two-instruction pack of four levels: %d mismatches of 64; low-half form kept the upper half in %d of 64 lanes\n", bad, keep);
    return 0;
}
