// Follow-up to valu_under_mfma.hip (gfx950): (A) VALU instructions of the SAME wave between its MFMAs -- cycles per MFMA with k
// independent v_fma_f32 behind each;  (B) a second wave's VALU stream beside a busy MFMA wave with s_setprio 3 on the VALU wave, and
// with the MFMA wave at s_setprio 0 / the VALU wave issuing v_pk_fma_f32 / v_dot2 / integer ops instead of v_fma_f32.
// hipcc --offload-arch=gfx950 -O3 tools/debug/valu_under_mfma2.hip -o tools/debug/_valu_under_mfma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256) void same_wave(const uint4* __restrict__ src, int iters, float* out, unsigned long long* clk) {
    const int tid = threadIdx.x;
    uint4 ua = src[tid], ub = src[tid + 512];
    bf16x8 a = *reinterpret_cast<bf16x8*>(&ua), b = *reinterpret_cast<bf16x8*>(&ub);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(tid + i);
    const float m = out[1], c = out[2];
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) v[(i + k) & 7] = fmaf(v[(i + k) & 7], m, c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        s += v[i];
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    if (s == 123.456f) out[0] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

__global__ __launch_bounds__(512) void cross(const uint4* __restrict__ src, int kind, int prio, int iters, float* out, unsigned long long* clk) {
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave < 4) {
        uint4 ua = src[tid], ub = src[tid + 512];
        bf16x8 a = *reinterpret_cast<bf16x8*>(&ua), b = *reinterpret_cast<bf16x8*>(&ub);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 123.456f) out[0] = s;
        if (tid == 0 && blockIdx.x == 0) clk[1] = 1;
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v[8];
    unsigned w[8];
    for (int i = 0; i < 8; ++i) { v[i] = (float)(tid + i); w[i] = tid * 17 + i; }
    const float m = out[1], c = out[2];
    __builtin_amdgcn_s_sleep(20);
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (kind == 0) v[i] = fmaf(v[i], m, c);
                else if (kind == 1) w[i] = (w[i] << 3) ^ (w[i] >> 5);          // integer: shift + xor (v_lshlrev, v_lshrrev, v_xor or fused)
                else if (kind == 2) v[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(2))) __bf16, w[i]),
                                                                          __builtin_bit_cast(__attribute__((ext_vector_type(2))) __bf16, w[(i + 1) & 7]), v[i], false);
                else w[i] = __builtin_amdgcn_perm(w[i], w[(i + 3) & 7], 0x07060302u);
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + (float)w[i];
    if (s == 123.456f) out[0] = s;
    if (tid == 256 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[2] = clk[1]; }
}

int main() {
    uint4* d; float* out; unsigned long long* clk;
    hipMalloc(&d, 1024 * sizeof(uint4)); hipMalloc(&out, 16); hipMalloc(&clk, 32);
    uint4 h[1024];
    uint32_t st = 1u;
    for (int i = 0; i < 1024; ++i) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; w[k] = (st & 0x807f807fu) | 0x3e003e00u; }
        h[i] = uint4{w[0], w[1], w[2], w[3]};
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    float ho[4] = {0.f, 0.999f, 0.001f, 0.f};
    hipMemcpy(out, ho, 16, hipMemcpyHostToDevice);
    unsigned long long c;
    const int iters = 4000;
#define RUN_SAME(K)                                                                                           \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(same_wave<K>, dim3(256), dim3(256), 0, 0, d, iters, out, clk); \
    hipDeviceSynchronize();                                                                                   \
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);                                                             \
    printf("same wave, 1 wave/SIMD: MFMA + %d v_fma_f32 each: %6.2f cycles per MFMA\n", K, (double)c / iters / 8);
    RUN_SAME(0) RUN_SAME(1) RUN_SAME(2) RUN_SAME(4) RUN_SAME(6) RUN_SAME(8)
    const char* kinds[4] = {"v_fma_f32", "shift+xor", "v_dot2_f32_bf16", "v_perm_b32"};
    for (int kind = 0; kind < 4; ++kind)
        for (int prio = 0; prio < 2; ++prio) {
            unsigned long long z[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                hipMemcpy(clk, z, 32, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(cross, dim3(256), dim3(512), 0, 0, d, kind, prio, 1000, out, clk);
                hipDeviceSynchronize();
            }
            unsigned long long r[4];
            hipMemcpy(r, clk, 32, hipMemcpyDeviceToHost);
            printf("other wave beside a busy MFMA wave: %-16s prio %d: %7.2f cycles per op group (MFMA wave %s)\n", kinds[kind], prio * 3,
                   (double)r[0] / 1000 / 32, r[2] ? "had finished!" : "still running");
        }
    return 0;
}
