// Shared device/host helpers for libdvq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dvq_hip.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdvq_hip targets gfx950 (MI355X) only"

#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // raw bf16 bits

#define DVQ_WAVE 64

// Barrier that PUBLISHES LDS-DMA data (buffer_load / global_load ... lds issued by this wave).  __syncthreads() alone is a
// workgroup-scope fence + s_barrier, and that fence waits for LDS traffic (lgkmcnt) only on gfx950: whether the compiler also waits
// for the DMA pieces depends on its alias guess for the LDS accesses behind the barrier.  A wave that passes with its pieces in flight
// lets the other waves' ds_reads overtake them (seen once a second process shared the GPU; tools/lint_dma_barriers.py checks the
// generated code of every kernel).  vmcnt(0) also drains any ordinary global load of the wave: use it where none is meant to stay in
// flight across the barrier.
__device__ __forceinline__ void dvq_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void dvq_set_error(const char* fmt, ...);

#define DVQ_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            dvq_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define DVQ_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            dvq_set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));   \
            return DVQ_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Probe modes -- timing experiments that REMOVE pieces of a kernel (no epilogue, no MFMA loop, DMA out of range ...) and therefore
// produce wrong results -- exist only in builds made with -DDVQ_PROBES (`python -m dynamicvectorquantization_amd.build --probes`
// -> libdvq_hip_probes.so, loaded when DVQ_USE_PROBES_LIB=1; tools/debug/ uses it).  In the product library their environment
// variables (DVQ_HALO_DBG, DVQ_HALO_LDS_PAD, DVQ_WGRAD_DBG, DVQ_ATTN_DBG, DVQ_VQ_DBG, DVQ_HALO2*) are not read at all: a stray
// variable cannot corrupt a training run.
#ifdef DVQ_PROBES
static inline int dvq_probe_env(const char* name) {
    const char* e = getenv(name);
    return e != nullptr ? atoi(e) : 0;
}
#else
static inline int dvq_probe_env(const char*) { return 0; }
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) costs ~0.3 ms per call: do it once per kernel symbol.
void dvq_ensure_dynamic_lds(const void* kernel, int bytes);
// caller-registered scratch buffer (dvq_set_workspace); null if none
void* dvq_workspace(int64_t* bytes);
// the calling stream's slot of that buffer (1/4 of it): concurrent streams never share scratch
void* dvq_workspace_stream(hipStream_t stream, int64_t* bytes);

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32 (device)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one instruction per pair instead of ~6 integer ops per element)
typedef __attribute__((ext_vector_type(4))) unsigned dvq_u32x4;      // payload type of the raw buffer load / store builtins
typedef __attribute__((ext_vector_type(2))) float dvq_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 dvq_bf16x2;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const dvq_f32x2 v = {lo, hi};
    const dvq_bf16x2 r = __builtin_convertvector(v, dvq_bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T>
struct ElemIO;
template <>
struct ElemIO<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct ElemIO<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); pointers must be 16-B aligned
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = pack_bf16x2(v[0], v[1]); a.y = pack_bf16x2(v[2], v[3]);
    a.z = pack_bf16x2(v[4], v[5]); a.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = a;
}

// streaming variants for passes over tensors far larger than the caches (GroupNorm at 256 x 256: 1 GB per operand): nontemporal
// loads / stores -- the lines are not kept in L2 / Infinity Cache, where they would only evict each other.  -DDVQ_STREAM_NT=0: plain
#ifndef DVQ_STREAM_NT
#define DVQ_STREAM_NT 1
#endif
__device__ __forceinline__ void load8_nt(const float* p, float (&v)[8]) {
#if DVQ_STREAM_NT
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 1);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
#else
    load8(p, v);
#endif
}
__device__ __forceinline__ void load8_nt(const bf16_t* p, float (&v)[8]) {
#if DVQ_STREAM_NT
    const dvq_u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const dvq_u32x4*>(p));
    v[0] = __uint_as_float(a[0] << 16); v[1] = __uint_as_float(a[0] & 0xffff0000u);
    v[2] = __uint_as_float(a[1] << 16); v[3] = __uint_as_float(a[1] & 0xffff0000u);
    v[4] = __uint_as_float(a[2] << 16); v[5] = __uint_as_float(a[2] & 0xffff0000u);
    v[6] = __uint_as_float(a[3] << 16); v[7] = __uint_as_float(a[3] & 0xffff0000u);
#else
    load8(p, v);
#endif
}
__device__ __forceinline__ void store8_nt(float* p, const float (&v)[8]) {
#if DVQ_STREAM_NT
    const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p) + 1);
#else
    store8(p, v);
#endif
}
__device__ __forceinline__ void store8_nt(bf16_t* p, const float (&v)[8]) {
#if DVQ_STREAM_NT
    const dvq_u32x4 a = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    __builtin_nontemporal_store(a, reinterpret_cast<dvq_u32x4*>(p));
#else
    store8(p, v);
#endif
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// dropout decisions: keep element `idx` iff dvq_hash32(lo32(idx) * rm + ra + hi32(idx) * 0x9E3779B1) >= p * 2^32, with the
// odd multiplier rm and the offset ra derived from the 64-bit seed (different seeds are not shifted copies of each other).
// Shared by dvq_dropout and the fused attention kernels (same seed + same element index = same mask).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned dvq_hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
static inline void dvq_dropout_seed(uint64_t seed, unsigned* rm, unsigned* ra) {
    *rm = dvq_hash32((unsigned)seed ^ 0x9E3779B9u) | 1u;
    *ra = dvq_hash32((unsigned)(seed >> 32) + 0x85ebca6bu) ^ dvq_hash32((unsigned)seed + 0xc2b2ae35u);
}

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division costs ~10 more VALU instructions per element, which
// dominated the GroupNorm+swish prologue of the fused convolutions
__device__ __forceinline__ float sigmoidf_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
}
__device__ __forceinline__ float swishf(float z) { return z * sigmoidf_fast(z); }
// d/dz [z * sigmoid(z)] = s * (1 + z * (1 - s))
__device__ __forceinline__ float swish_grad(float z) {
    float s = sigmoidf_fast(z);
    return s * (1.0f + z * (1.0f - s));
}

#define DVQ_DISPATCH_DTYPE(dtype, T, ...)          \
    do {                                           \
        if ((dtype) == DVQ_F32) {                  \
            using T = float;                       \
            __VA_ARGS__                            \
        } else {                                   \
            using T = bf16_t;                      \
            __VA_ARGS__                            \
        }                                          \
    } while (0)
