// Sustained MFMA ceiling probe (gfx950): register-only v_mfma_f32_32x32x16_bf16 loops, operands = zeros / constant / random bf16.
// Reports TFLOP/s over a ~launch of a few ms and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
// hipcc --offload-arch=gfx950 -O3 tools/debug/mfma_peak.hip -o tools/debug/_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, int iters, float* out, unsigned long long* clk) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    uint4 ua = src[tid % 4096], ub = src[(tid * 7 + 13) % 4096];
    bf16x8 a = *reinterpret_cast<bf16x8*>(&ua), b = *reinterpret_cast<bf16x8*>(&ub);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if (tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    const int n = 4096;
    uint4* h = new uint4[n];
    uint4* d; float* out; unsigned long long* clk;
    hipMalloc(&d, n * sizeof(uint4)); hipMalloc(&out, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zeros", "const 1.0", "random"};
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
        for (int mode = 0; mode < 3; ++mode) {
            uint32_t st = 12345u;
            for (int i = 0; i < n; ++i) {
                uint32_t w[4];
                for (int k = 0; k < 4; ++k) {
                    if (mode == 0) w[k] = 0;
                    else if (mode == 1) w[k] = 0x3f803f80u;
                    else {   // random bf16 in (-2, 2): random sign / mantissa, exponent 0x7c .. 0x7f
                        uint32_t v = 0;
                        for (int hlf = 0; hlf < 2; ++hlf) {
                            st = st * 1664525u + 1013904223u;
                            const uint32_t r = st >> 8;
                            const uint32_t bf = ((r & 1) << 15) | ((0x7c + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f);
                            v |= bf << (16 * hlf);
                        }
                        w[k] = v;
                    }
                }
                h[i] = uint4{w[0], w[1], w[2], w[3]};
            }
            hipMemcpy(d, h, n * sizeof(uint4), hipMemcpyHostToDevice);
            const int blocks = 256 * waves_per_simd, iters = 40000;
            mfma_loop<4><<<blocks, 256>>>(d, 2000, out, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<4><<<blocks, 256>>>(d, iters, out, clk);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
            printf("%d wave(s)/SIMD %-10s %7.3f ms  %7.0f TFLOP/s   memtime/realtime = %.3f (x100 MHz => %.0f MHz if memtime counts shader clocks)\n",
                   waves_per_simd, names[mode], ms, flop / ms / 1e9, (double)hc[0] / (double)hc[1], 100.0 * hc[0] / hc[1]);
        }
    return 0;
}
