/* Host stand-in for <cuda_fp16.h>: TEST INFRASTRUCTURE (oracle/_ref build only; see oracle/build_ref.py).
 * Lets g++ compile the reference's own device code (the bodies between `__global__` / `__device__`) for the CPU, so that the
 * restated oracle (oracle/tcnn_oracle.c) can be pinned against the reference's arithmetic bit for bit.
 * __half is IEEE binary16 with round-to-nearest-even conversions, as on the device; each arithmetic operator rounds its result to
 * binary16 once (a product or sum of two binary16 values is exact in binary64, so computing there and rounding once IS the correctly
 * rounded binary16 result the device's hadd / hmul / hfma deliver). */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define __shared__ /* `extern __shared__ __half shmem[]` names one host array (ref_driver.cpp); blocks run one after the other */

struct __half {
	_Float16 v;
	__half() = default;
	constexpr __half(_Float16 x) : v(x) {}
	__half(float x) : v((_Float16)x) {}
	__half(double x) : v((_Float16)x) {}
	__half(int x) : v((_Float16)x) {}
	__half(unsigned x) : v((_Float16)x) {}
	operator float() const { return (float)v; }
};
typedef __half half;
static inline __half ref_round(double x) { return __half((_Float16)x); }
static inline __half operator+(__half a, __half b) { return ref_round((double)a.v + (double)b.v); }
static inline __half operator-(__half a, __half b) { return ref_round((double)a.v - (double)b.v); }
static inline __half operator*(__half a, __half b) { return ref_round((double)a.v * (double)b.v); }
static inline __half operator/(__half a, __half b) { return ref_round((double)a.v / (double)b.v); }
static inline __half operator-(__half a) { return __half((_Float16)-a.v); }
static inline __half& operator+=(__half& a, __half b) { a = a + b; return a; }
static inline __half& operator-=(__half& a, __half b) { a = a - b; return a; }
static inline __half& operator*=(__half& a, __half b) { a = a * b; return a; }
static inline __half& operator/=(__half& a, __half b) { a = a / b; return a; }
static inline bool operator<(__half a, __half b) { return a.v < b.v; }
static inline bool operator>(__half a, __half b) { return a.v > b.v; }
static inline bool operator<=(__half a, __half b) { return a.v <= b.v; }
static inline bool operator>=(__half a, __half b) { return a.v >= b.v; }
static inline bool operator==(__half a, __half b) { return a.v == b.v; }
static inline bool operator!=(__half a, __half b) { return a.v != b.v; }
static inline __half __float2half(float x) { return __half(x); }
static inline __half __float2half_rn(float x) { return __half(x); }
static inline float __half2float(__half x) { return (float)x.v; }
static inline __half __hfma(__half a, __half b, __half c) { return ref_round((double)a.v * (double)b.v + (double)c.v); }  /* a*b exact in binary64; one rounding of the sum */
static inline __half __hmul(__half a, __half b) { return a * b; }
static inline __half __hadd(__half a, __half b) { return a + b; }
static inline __half __hsub(__half a, __half b) { return a - b; }
static inline __half __hmax(__half a, __half b) { return a.v > b.v ? a : b; }
static inline __half __hmin(__half a, __half b) { return a.v < b.v ? a : b; }
static inline __half hsqrt(__half a) { return __half((_Float16)std::sqrt((float)a.v)); }
static inline __half hexp(__half a) { return __half((_Float16)std::exp((float)a.v)); }

struct __align__(4) __half2 {
	__half x, y;
	__half2() = default;
	__half2(__half a, __half b) : x(a), y(b) {}
};
typedef __half2 half2;
static inline __half2 __hsub2(__half2 a, __half2 b) { return {a.x - b.x, a.y - b.y}; }
static inline __half2 __h2div(__half2 a, __half2 b) { return {a.x / b.x, a.y / b.y}; }
static inline __half2 __hadd2(__half2 a, __half2 b) { return {a.x + b.x, a.y + b.y}; }
static inline __half2 __hmul2(__half2 a, __half2 b) { return {a.x * b.x, a.y * b.y}; }
static inline __half2 __hfma2(__half2 a, __half2 b, __half2 c) { return {__hfma(a.x, b.x, c.x), __hfma(a.y, b.y, c.y)}; }
static inline __half2 __floats2half2_rn(float a, float b) { return {__half(a), __half(b)}; }
static inline __half2 __half2half2(__half a) { return {a, a}; }

/* one host thread runs the "kernel" for one (block, thread) at a time: atomics are plain read-modify-writes */
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline __half atomicAdd(__half* p, __half v) { __half o = *p; *p = o + v; return o; }
static inline __half2 atomicAdd(__half2* p, __half2 v) { __half2 o = *p; *p = __hadd2(o, v); return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

struct ref_dim3 { unsigned x = 1, y = 1, z = 1; };
static thread_local ref_dim3 threadIdx, blockIdx, blockDim, gridDim;
/* kernels without barriers run thread after thread (ref_sync_hook == nullptr); the fused network kernels run a block's threads as
 * fibers and the hook switches to the next one (oracle/ref_driver.cpp, BlockFibers) */
static thread_local void (*ref_sync_hook)() = nullptr;
static inline void __syncthreads() { if (ref_sync_hook) ref_sync_hook(); }
static inline void __syncwarp() { if (ref_sync_hook) ref_sync_hook(); }
/* the fast-math intrinsics (__expf, __sinf, ...) are glibc-internal names on the host: the device versions are approximations anyway,
 * nothing on the pinned path (index / interpolation / loss / optimizer arithmetic) uses them */
#define __expf(x) std::exp((float)(x))
#define __logf(x) std::log((float)(x))
#define __powf(a, b) std::pow((float)(a), (float)(b))
#define __sinf(x) std::sin((float)(x))
#define __cosf(x) std::cos((float)(x))
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float normcdff(float x) { return 0.5f * std::erfc(-x * 0.70710678118654752440f); }
using std::fmaf;
using std::isfinite;  /* vec.h calls ::isfinite in device code */
static inline float fmaf(__half a, __half b, __half c) { return std::fmaf((float)a, (float)b, (float)c); }
/* CUDA's built-in vector types as far as vec.h converts from / to them */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
