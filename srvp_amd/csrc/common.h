// Common device helpers for the SRVP gfx950 (MI355X / CDNA4) kernels.
// Everything here is written for wave64 + MFMA; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define SRVP_OK 0
#define SRVP_ERR_ARG 1
#define SRVP_ERR_LAUNCH 2

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even (same as torch's float->bfloat16): the gfx950 hardware conversion v_cvt_pk_bf16_f32 -- one
// instruction per PAIR of values (the bit-twiddling version was ~10 VALU instructions per value, which made the
// conversion-heavy epilogues and the BatchNorm passes VALU-bound)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ void unpack8(const u32x4_t& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    u32x4_t v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- element type of the activation / gradient tensors: bf16_t (production) or float (precision = 'fp32' parity mode).
// The HBM-streaming kernels are templated on it and move eight consecutive channels per access either way.
template <class E> struct El;
template <> struct El<bf16_t> {
    static constexpr bool is_f32 = false;
    static __device__ __forceinline__ void ld8(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const u32x4_t*>(p), f); }
    static __device__ __forceinline__ void ld8_nt(const bf16_t* p, float* f) { unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)), f); }
    static __device__ __forceinline__ void st8(bf16_t* p, const float* f) { *reinterpret_cast<u32x4_t*>(p) = pack8(f); }
    static __device__ __forceinline__ void st8_nt(bf16_t* p, const float* f) { __builtin_nontemporal_store(pack8(f), reinterpret_cast<u32x4_t*>(p)); }
    static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }        // the value as the tensor stores it
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct El<float> {
    static constexpr bool is_f32 = true;
    static __device__ __forceinline__ void ld8(const float* p, float* f) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    }
    static __device__ __forceinline__ void ld8_nt(const float* p, float* f) { ld8(p, f); }
    static __device__ __forceinline__ void st8(float* p, const float* f) {
        *reinterpret_cast<f32x4_t*>(p) = f32x4_t{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4_t*>(p + 4) = f32x4_t{f[4], f[5], f[6], f[7]};
    }
    static __device__ __forceinline__ void st8_nt(float* p, const float* f) { st8(p, f); }
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// activation ids shared with the host side (srvp_hip.h)
#define ACT_NONE 0
#define ACT_LRELU 1   // LeakyReLU(0.2)  (reference utils.py:41)
#define ACT_TANH 2
#define ACT_RELU 3
#define ACT_SIGMOID 4
#define LRELU_SLOPE 0.2f

__device__ __forceinline__ float act_fwd(float v, int act) {
    switch (act) {
        case ACT_LRELU: return v > 0.f ? v : LRELU_SLOPE * v;
        case ACT_TANH: return tanhf(v);
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}
// derivative given the pre-activation value v (and for tanh/sigmoid recomputed output)
__device__ __forceinline__ float act_bwd(float v, int act) {
    switch (act) {
        case ACT_LRELU: return v > 0.f ? 1.f : LRELU_SLOPE;
        case ACT_TANH: { float t = tanhf(v); return 1.f - t * t; }
        case ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case ACT_SIGMOID: { float s = 1.f / (1.f + __expf(-v)); return s * (1.f - s); }
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// XCD-aware workgroup remap (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroup b is dispatched to XCD b % 8).
// Returns a logical index such that every XCD owns one CONTIGUOUS range of logical indices, so workgroups that share
// operand panels (neighbouring logical indices) hit the same L2.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- deterministic mode (srvp_set_deterministic / SRVP_DETERMINISTIC=1; precision = 'fp32' only): every cross-workgroup sum that the
// default build forms with atomics in arrival order is formed in a FIXED order instead -- workgroups write their partial sums into
// slab[workgroup][n] of a caller-provided workspace and a second tiny launch adds them up in workgroup order -- or by a single
// workgroup / a single split (one atomic per destination element onto a value that only stream order precedes).
extern int g_srvp_det;                 // util.hip
extern void* g_srvp_det_ws;            // caller-owned device workspace
extern long long g_srvp_det_ws_bytes;
template <class T>
__global__ void det_sum_kernel(const T* __restrict__ slab, int nwg, int n, T* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = 0;
    for (int w = 0; w < nwg; ++w) acc += slab[(size_t)w * n + i];
    dst[i] += acc;
}

// ---- error reporting across the C ABI (no exceptions cross it) ----
void srvp_set_error(const char* fmt, ...);
#define SRVP_CHECK_LAUNCH(name)                                                         \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            srvp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return SRVP_ERR_LAUNCH;                                                     \
        }                                                                               \
    } while (0)
#define SRVP_REQUIRE(cond, ...)                                                         \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            srvp_set_error(__VA_ARGS__);                                                \
            return SRVP_ERR_ARG;                                                        \
        }                                                                               \
    } while (0)
