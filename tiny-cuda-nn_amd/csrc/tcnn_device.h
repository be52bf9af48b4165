// tcnn_device.h -- device-side vocabulary shared by all gfx950 kernels of the tiny-cuda-nn_amd hot path.
//
// The kernels are written once.  The product build compiles them with hipcc for gfx950.  The test
// suite can additionally compile the same sources for the host with TCNN_HOST_EMU defined, where
// tests/emu/hip_emu.h supplies a lock-step SIMT emulator (threads, LDS, barriers, MFMA lane maps,
// atomics).  The emulator is test infrastructure: nothing under tiny-cuda-nn_amd/ builds it in.
#pragma once

#include <stdint.h>

#if defined(TCNN_HOST_EMU)
#include "hip_emu.h"
#define TCNN_DYN_LDS(name) unsigned char* name = (unsigned char*)::emu::dyn_lds()
#define TCNN_SET_MAX_DYN_LDS(kernel, bytes) (void)0
#else
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TCNN_DEVICE __device__ __forceinline__
#define TCNN_HOST_DEVICE __host__ __device__ __forceinline__
namespace tcnn_hip {
// bit 0: synchronise the stream after every launch and check for errors (TCNN_DEBUG_SYNC=1 / tcnn_set_debug_launches);
// bit 1: print the kernel's name to stderr before launching it (TCNN_DEBUG_TRACE=1): the last line names a faulting kernel
inline int& debug_launch_flags() {
	static int flags = (getenv("TCNN_DEBUG_SYNC") && atoi(getenv("TCNN_DEBUG_SYNC")) ? 1 : 0) | (getenv("TCNN_DEBUG_TRACE") && atoi(getenv("TCNN_DEBUG_TRACE")) ? 2 : 0);
	return flags;
}
inline void launch_trace(const char* kernel, dim3 grid, dim3 block, size_t shmem) {
	if (debug_launch_flags() & 2) {
		fprintf(stderr, "[tcnn launch] %s grid %u block %u lds %zu\n", kernel, grid.x, block.x, shmem);
		fflush(stderr);
	}
}
// every launch is checked (the reference wraps its launches the same way, common_host.h:97-102): a launch the runtime refuses
// (kernel arguments, LDS size, launch bounds) throws instead of leaving the step to run on garbage
inline void launch_check(const char* kernel, hipStream_t stream) {
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && (debug_launch_flags() & 1)) e = hipStreamSynchronize(stream);
	if (e != hipSuccess) throw std::runtime_error(std::string("tiny-cuda-nn_amd: launch of ") + kernel + " failed: " + hipGetErrorString(e));
}
}  // namespace tcnn_hip
// (the error state is cleared BEFORE the launch: hipGetLastError returns and clears whatever non-sticky error another library -- torch,
// RCCL -- left on this thread, and it would otherwise be reported as this kernel's launch failure)
#define TCNN_LAUNCH(kernel, grid, block, shmem, stream, ...)                  \
	do {                                                                      \
		::tcnn_hip::launch_trace(#kernel, grid, block, shmem);                \
		(void)hipGetLastError();                                              \
		hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);  \
		::tcnn_hip::launch_check(#kernel, stream);                            \
	} while (0)
// dynamic LDS: one 16-byte aligned carve base per kernel (cdna_hip_programming.md G17)
#define TCNN_DYN_LDS(name)                                                        \
	extern __shared__ __attribute__((aligned(16))) unsigned char tcnn_dyn_lds_[]; \
	unsigned char* name = tcnn_dyn_lds_
// kernels may use up to the full 160 KiB of a CU's LDS; HIP needs to be told above 64 KiB
#define TCNN_SET_MAX_DYN_LDS(kernel, bytes)                                                                                     \
	do {                                                                                                                        \
		if (hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) \
			throw std::runtime_error("tiny-cuda-nn_amd: could not reserve dynamic LDS for a kernel");                            \
	} while (0)
#endif

namespace tcnn_hip {

// The parameter / activation / gradient type is a compile-time choice, as in the reference (TCNN_HALF_PRECISION ->
// network_precision_t, common.h:66-70): the sources build once with IEEE fp16 (libtcnn_hip.so) and once, with -DTCNN_BF16,
// with bfloat16 (libtcnn_hip_bf16.so: v_mfma_f32_16x16x32_bf16, fp32 range for activations and gradients -- the stress
// shape of BASELINE configs[4]).  `half_t` is that type; everything below is written against it.
#if defined(TCNN_BF16)
typedef __bf16 half_t;
typedef __bf16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 h8 __attribute__((ext_vector_type(8)));
constexpr bool HALF_IS_BF16 = true;
#else
typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr bool HALF_IS_BF16 = false;
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

constexpr uint32_t WAVE = 64;                      // CDNA wavefront
constexpr uint32_t BATCH_SIZE_GRANULARITY = 256;   // reference common.h:246
constexpr uint32_t MAX_N_LEVELS = 128;             // reference multi_level_interface.h:84-88
constexpr float LOSS_SCALE_FP16 = 128.0f;          // reference common.h:243

// ---------------------------------------------------------------------------------------------
// MFMA wrappers.  Fragment maps (cdna_hip_programming.md section 3):
//   16x16x32 f16:  A[i = lane&15][k = 8*(lane>>4) + j], B[k = 8*(lane>>4) + j][n = lane&15], j<8
//   16x16x16 f16:  A[i = lane&15][k = 4*(lane>>4) + j], B[k = 4*(lane>>4) + j][n = lane&15], j<4
//   C/D (both):    row = 4*(lane>>4) + r, col = lane&15, r<4
// ---------------------------------------------------------------------------------------------
#if !defined(TCNN_HOST_EMU)
#if defined(TCNN_BF16)
typedef short bf16_bits2 __attribute__((ext_vector_type(2)));
typedef short bf16_bits4 __attribute__((ext_vector_type(4)));
TCNN_DEVICE f4 mfma_16x16x32(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
TCNN_DEVICE f4 mfma_16x16x16(h4 a, h4 b, f4 c) {
	return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16_bits4, a), __builtin_bit_cast(bf16_bits4, b), c, 0, 0, 0);
}
// Packed global atomic add (global_atomic_pk_add_bf16), no return value.
TCNN_DEVICE void atomic_add_h2(half_t* addr, h2 v) {
	__builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16_bits2*)addr, __builtin_bit_cast(bf16_bits2, v));
}
#else
TCNN_DEVICE f4 mfma_16x16x32(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
TCNN_DEVICE f4 mfma_16x16x16(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// Packed-half global atomic add (global_atomic_pk_add_f16), no return value.
TCNN_DEVICE void atomic_add_h2(half_t* addr, h2 v) {
	__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)addr, v);
}
#endif
// ds_read_b64_tr_b16 (gfx950): every lane passes the address of 4 consecutive 16-bit elements in LDS (8-byte aligned); inside
// each group of 16 lanes the 16 x 4 elements are exchanged so that lane c, element j receives element (c & 3) of the word that
// lane 4j + (c >> 2) addressed (probed: scripts/probe_tr16.hip).  With lane i of the group addressing
// M[row0 + (i >> 2)][col0 + 4 (i & 3)] of a row-major image, lane c gets M[row0 + j][col0 + c], j < 4: a column of 4 -- the
// "consecutive k for one n" fragment of an MFMA operand, out of an image whose rows run along n.  All 64 lanes must execute it.
typedef short lds_tr_bits4 __attribute__((ext_vector_type(4)));
TCNN_DEVICE h4 lds_read_tr4(const half_t* word) {
	return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lds_tr_bits4*)word));
}
TCNN_DEVICE void atomic_add_f32(float* addr, float v) { unsafeAtomicAdd(addr, v); }
TCNN_DEVICE void lds_atomic_add_f32(float* addr, float v) { atomicAdd(addr, v); }  // ds_add_f32
TCNN_DEVICE void lds_atomic_add_u64(unsigned long long* addr, unsigned long long v) { atomicAdd(addr, v); }  // ds_add_u64
TCNN_DEVICE uint32_t atomic_add_u32(uint32_t* addr, uint32_t v) { return atomicAdd(addr, v); }  // returns the old value (LDS or global)
#if defined(TCNN_BF16)
TCNN_DEVICE void lds_atomic_add_h2(h2* addr, h2 v) {  // ds_pk_add_bf16
	__builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) bf16_bits2*)addr, __builtin_bit_cast(bf16_bits2, v));
}
// gfx950 has no bf16 fma: the interpolation chain of grid.h:144-163 (`fma` in the parameter type) is an fp32 fma rounded
// to bf16 -- the oracle's bf16 mode restates exactly this
TCNN_DEVICE half_t fma_h(half_t a, half_t b, half_t c) { return (half_t)__builtin_fmaf((float)a, (float)b, (float)c); }
TCNN_DEVICE h2 fma_h2(h2 a, h2 b, h2 c) { return h2{fma_h(a[0], b[0], c[0]), fma_h(a[1], b[1], c[1])}; }
#else
TCNN_DEVICE void lds_atomic_add_h2(h2* addr, h2 v) {  // ds_pk_add_f16
	__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)addr, v);
}
TCNN_DEVICE h2 fma_h2(h2 a, h2 b, h2 c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f16
TCNN_DEVICE half_t fma_h(half_t a, half_t b, half_t c) { return __builtin_fmaf16(a, b, c); }
#endif
TCNN_DEVICE uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)); }  // HW_REG_XCC_ID
#endif

TCNN_DEVICE h8 pack8(h4 a, h4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
TCNN_DEVICE h4 pack4(h2 a, h2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
TCNN_DEVICE f4 zero4() { return f4{0.0f, 0.0f, 0.0f, 0.0f}; }

TCNN_DEVICE uint32_t lane_id() { return threadIdx.x & 63u; }

// Orders the LDS accesses of ONE wavefront (all 64 lanes must call it): what a block barrier does for a workgroup,
// for code that only one wave executes.  The hardware runs a wave in lock step; this pins the compiler and drains
// the LDS queue.
#if defined(TCNN_HOST_EMU)
TCNN_DEVICE void wave_lds_sync() { ::emu::wave_barrier(); }
#else
TCNN_DEVICE void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// Sum of `v` over the 64 lanes of a wavefront, in every lane (all lanes must call it; xor butterfly, fixed order).
TCNN_DEVICE float wave_sum_f32(float v) {
#if defined(TCNN_HOST_EMU)
	return ::emu::wave_sum_f32(v);
#endif
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += __builtin_bit_cast(float, __shfl_xor(__builtin_bit_cast(int, v), d, 64));
	return v;
}

// 4 x 4 transpose of 16-bit elements across the four 16-lane rows of a wavefront (all 64 lanes must call it): lane (g, c),
// g = lane >> 4, passes v[0..3] and receives t[r] = element g of what lane (r, c) passed.  No LDS: gfx950's row-swapping moves --
// v_permlane32_swap (rows 2, 3 of the first operand <-> rows 0, 1 of the second) exchanges the 2 x 2 blocks, v_permlane16_swap
// (odd rows of the first <-> even rows of the second) brings the row partner's pair next to the lane's own, v_perm_b32 picks
// the halves.  Probed on the chip: scripts/probe_permlane_swap.hip.
TCNN_DEVICE h4 wave_rows_transpose4(h4 v) {
#if defined(TCNN_HOST_EMU)
	return ::emu::wave_rows_transpose4(v);
#else
	const uint32_t a = __builtin_bit_cast(uint32_t, __builtin_shufflevector(v, v, 0, 1)), b = __builtin_bit_cast(uint32_t, __builtin_shufflevector(v, v, 2, 3));
	const auto blocks = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // [0]: rows 0, 1 keep a, rows 2, 3 get b of row - 2; [1]: rows 0, 1 get a of row + 2
	const uint32_t sel = (threadIdx.x & 16u) ? 0x07060302u : 0x05040100u;     // odd rows keep the high halves, even rows the low ones
	const auto lo = __builtin_amdgcn_permlane16_swap(blocks[0], blocks[0], false, false);  // even rows: {own, row + 1's}; odd rows: {row - 1's, own}
	const auto hi = __builtin_amdgcn_permlane16_swap(blocks[1], blocks[1], false, false);
	const uint32_t t01 = __builtin_amdgcn_perm(lo[1], lo[0], sel), t23 = __builtin_amdgcn_perm(hi[1], hi[0], sel);
	const h2 p = __builtin_bit_cast(h2, t01), q = __builtin_bit_cast(h2, t23);
	return h4{p[0], p[1], q[0], q[1]};
#endif
}

// Scheduling fence: the compiler may not move instructions across it.  Bounds the live ranges of a long unrolled body
// (the scheduler otherwise hoists every LDS/global read to the top of the block and runs out of registers).
#if defined(TCNN_HOST_EMU)
TCNN_DEVICE void sched_fence() {}
#else
#if defined(TCNN_NO_SCHED_FENCE)
TCNN_DEVICE void sched_fence() {}
#else
TCNN_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif
#endif

// fp32 -> fp16 with exactly ONE extra rounding (RNE) of an already rounded fp32 value.
// Without the barrier hipcc folds `(half)(a * b)` / `(half)(a + b)` into v_fma_mixlo_f16, which rounds
// the exact product/sum once -- a different (double- vs single-rounding) result in rare tie cases, and
// therefore not bit-identical to the reference's "compute in fp32, then convert" (grid.h:162, 254).
#if defined(TCNN_HOST_EMU)
TCNN_DEVICE half_t to_half_rn(float x) { return (half_t)x; }
#else
TCNN_DEVICE half_t to_half_rn(float x) {
	asm volatile("" : "+v"(x));
	return (half_t)x;
}
#endif

template <typename T>
TCNN_HOST_DEVICE T div_round_up(T a, T b) { return (a + b - 1) / b; }
template <typename T>
TCNN_HOST_DEVICE T next_multiple(T a, T b) { return div_round_up(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// Grid indexing -- reference common_device.h:767-895, 1000-1043.  Integer work: bit-exact.
// ---------------------------------------------------------------------------------------------
enum class GridType : int { Hash = 0, Dense = 1, Tiled = 2 };
enum class InterpolationType : int { Nearest = 0, Linear = 1, Smoothstep = 2 };

struct GridLevelTable {
	// One entry per level, computed ONCE on the host in fp32 (canonical scale table) and
	// passed by value; the reference recomputes exp2f per thread (grid.h:97), which is not portable
	// bit-for-bit across devices.
	uint32_t offset[MAX_N_LEVELS + 1];  // in grid entries
};

template <uint32_t D>
TCNN_HOST_DEVICE uint32_t coherent_prime_hash(const uint32_t (&p)[D]) {
	constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
	uint32_t r = 0;
#pragma unroll
	for (uint32_t i = 0; i < D; ++i) r ^= p[i] * primes[i];
	return r;
}

template <uint32_t D>
TCNN_HOST_DEVICE uint32_t grid_index(bool is_hash, uint32_t hashmap_size, uint32_t resolution, const uint32_t (&p)[D]) {
	constexpr uint32_t MAX_BASES[11] = {0x0, 0xFFFFFFFF, 0xFFFF, 0x659, 0xFF, 0x54, 0x28, 0x17, 0xF, 0xB, 0x9};
	uint32_t stride = 1;
	uint32_t index = 0;
	if (resolution <= MAX_BASES[D]) {
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) {
			index += p[d] * stride;
			stride *= resolution;
		}
	} else {
		stride = 0xFFFFFFFFu;
	}
	if (is_hash && hashmap_size < stride) index = coherent_prime_hash<D>(p);
	// hashed levels have power-of-two tables (min(dense, 2^log2_hashmap_size)): a mask instead of the
	// ~15-instruction u32 remainder; the condition is wave-uniform (one level per workgroup)
	const uint32_t mask = hashmap_size - 1u;
	if ((hashmap_size & mask) == 0u) return index & mask;
	// densely indexed level (table = resolution^D entries, rounded up to 8): a position inside the unit cube gives
	// index <= res + res^2 + ... + res^D < 2 * table, so the remainder is at most one subtraction; the ~25-instruction u32
	// division only runs for lanes whose position lies outside (same result either way: x % m == (x - m) % m)
	if (index < hashmap_size) return index;
	index -= hashmap_size;
	if (index < hashmap_size) return index;
	return index % hashmap_size;
}

}  // namespace tcnn_hip
