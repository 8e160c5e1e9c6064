// tools/tr_probe.hip -- empirical semantics of ds_read_b64_tr_b16 on gfx950.
// LDS holds u16 value = element index.  For several per-lane address patterns, print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(int pattern, int pitch, unsigned short *out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr = 0;  // byte address relative to lds
    if (pattern == 0) addr = 0;
    if (pattern == 1) addr = l * 8;
    if (pattern == 2) addr = ((l & 15) * pitch + (l >> 4) * 4) * 2;         // lane -> row (l&15), 4 cols at (l>>4)*4
    if (pattern == 3) addr = (((l >> 4) * 4 + (l & 3)) * pitch + (l & 12)) * 2; // guess: rows by l&3 within 4-row group
    if (pattern == 4) addr = ((l & 15) * 4 + (l >> 4) * pitch * 4) * 2;
    unsigned base = (unsigned)(uintptr_t)lds;
    unsigned long long v;
    unsigned a = base + addr;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}

int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    for (int pat = 0; pat < 5; ++pat)
        for (int pitch : {16, 64}) {
            if (pat < 2 && pitch != 16) continue;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pat, pitch, d);
            hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
            printf("pattern %d pitch %d\n", pat, pitch);
            for (int l = 0; l < 64; ++l) {
                printf("  l%02d:[%4d %4d %4d %4d]", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
                if (l % 4 == 3) printf("\n");
            }
        }
    return 0;
}
