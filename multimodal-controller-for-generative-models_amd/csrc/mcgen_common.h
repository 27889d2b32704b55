// Shared device/host helpers for libmcgen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mcgen_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MCGEN_CK 32          // input channels per K chunk (one bf16 16x16x32 MFMA step)

int mcgen_fail(const char* fmt, ...);   // records the message, returns 1

#define MCGEN_CHECK(cond, ...) do { if (!(cond)) return mcgen_fail(__VA_ARGS__); } while (0)
#define MCGEN_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return mcgen_fail("%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// ---- element traits -------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int BYTES = 4;
    typedef f32x8 vec8;                                   // 8 consecutive channels
    static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ vec8 load8v(const float* p) { return *reinterpret_cast<const vec8*>(p); }
    static __device__ __forceinline__ void unpack8(const vec8& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r[i];
    }
    static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
    }
    static __device__ __forceinline__ float to_f(float x) { return x; }
    static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <> struct Elem<bf16_t> {
    static constexpr int BYTES = 2;
    typedef bf16x8 vec8;
    static __device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(r[i] << 16);
            v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ vec8 load8v(const bf16_t* p) { return *reinterpret_cast<const vec8*>(p); }
    static __device__ __forceinline__ void unpack8(const vec8& r8, float (&v)[8]) {
        union { vec8 h; u32x4 w; } u; u.h = r8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(u.w[i] << 16);
            v[2 * i + 1] = __uint_as_float(u.w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8*>(p) = o;
    }
    static __device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
    static __device__ __forceinline__ bf16_t from_f(float x) { return (bf16_t)x; }
};

static __device__ __forceinline__ void load8f(const float* p, float (&v)[8]) { Elem<float>::load8(p, v); }

static inline int ilog2_exact(int x) {          // -1 when x is not a power of two
    if (x <= 0 || (x & (x - 1))) return -1;
    int l = 0; while ((1 << l) < x) ++l; return l;
}
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
