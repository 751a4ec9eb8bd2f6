// Cycles per v_mfma_f32_32x32x16_bf16 issued by ONE wave per SIMD as a function of the number of independent accumulator chains
// (1, 2, 3, 4, 8) and of where the B operand comes from (fixed registers / rotating registers).  gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_dep.hip -o tools/debug/_mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC, int NB>
__global__ __launch_bounds__(256, 1) void chains(const uint4* __restrict__ src, int iters, float* out, unsigned long long* clk) {
    const int tid = threadIdx.x;
    bf16x8 a[NB], b[NB];
    for (int i = 0; i < NB; ++i) {
        uint4 ua = src[tid + 256 * i], ub = src[tid + 256 * i + 2048];
        a[i] = *reinterpret_cast<bf16x8*>(&ua);
        b[i] = *reinterpret_cast<bf16x8*>(&ub);
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % NB], b[(i / 2) % NB], acc[i % NACC], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int NACC, int NB>
static void run(const uint4* src, float* out, unsigned long long* clk, const char* what) {
    const int iters = 2000;
    chains<NACC, NB><<<256, 256>>>(src, iters, out, clk);
    chains<NACC, NB><<<256, 256>>>(src, iters, out, clk);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s chains %d  operand sets %d : %.1f cycles per MFMA\n", what, NACC, NB, (double)c / (iters * 16.0));
}

int main() {
    uint4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 4096 * 16 * 2); hipMalloc(&out, 64); hipMalloc(&clk, 64);
    uint4* h = (uint4*)malloc(4096 * 16 * 2);
    uint32_t st = 12345;
    for (int i = 0; i < 4096 * 2; ++i) {
        uint32_t w[4];
        for (int j = 0; j < 4; ++j) { st = st * 1664525u + 1013904223u; w[j] = (st & 0x7fff7fffu) | 0x3c003c00u & 0x3fff3fffu; }
        h[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    hipMemcpy(src, h, 4096 * 16 * 2, hipMemcpyHostToDevice);
    hipMemset(out, 0, 64);
    run<1, 1>(src, out, clk, "one wave / SIMD");
    run<2, 1>(src, out, clk, "one wave / SIMD");
    run<3, 1>(src, out, clk, "one wave / SIMD");
    run<4, 1>(src, out, clk, "one wave / SIMD");
    run<8, 1>(src, out, clk, "one wave / SIMD");
    run<2, 8>(src, out, clk, "one wave / SIMD, rotating A / B registers");
    run<4, 8>(src, out, clk, "one wave / SIMD, rotating A / B registers");
    run<8, 8>(src, out, clk, "one wave / SIMD, rotating A / B registers");
    return 0;
}
