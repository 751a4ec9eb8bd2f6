// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS element e (bf16 slot) holds the value e.
// Every lane supplies the byte address `addr_in[lane]`; the 4 returned 16-bit values per lane are dumped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const int* addr_in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(size_t)(&lds[0]) + (unsigned)addr_in[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15;
            if (mode == 0) h_addr[l] = (g * 64 + i * 4) * 2;                       // contiguous [4][16] tile per group
            if (mode == 1) h_addr[l] = ((g * 4 + (i >> 2)) * 64 + (i & 3) * 4) * 2;  // rows of 64 elements (128-B stride)
            if (mode == 2) h_addr[l] = (((i & 3)) * 64 + (i >> 2) * 4 + g * 1024) * 2; // alt hypothesis: lane i -> row i&3
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr_elem %5d -> %5d %5d %5d %5d\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
        }
    }
    return 0;
}
