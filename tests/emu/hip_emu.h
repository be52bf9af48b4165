// hip_emu.h -- a minimal lock-step SIMT emulator so that the gfx950 kernel SOURCES of
// tiny-cuda-nn_amd/csrc can also be compiled for the host and exercised by `pytest -m "not gpu"`.
//
// TEST INFRASTRUCTURE ONLY (lives under tests/).  It is not a fallback: the product library never
// includes this header, and nothing here is reachable from tiny-cuda-nn_amd/.  Purpose: catch
// indexing / fragment-layout / barrier bugs in the kernels on a machine without a GPU.
//
// Model: one workgroup at a time; every thread of the workgroup is a ucontext fiber on ONE OS
// thread (so "atomics" are trivially atomic and runs are deterministic).  __syncthreads() and the
// wave-collective MFMA are rendezvous points at which fibers yield.  MFMA lane maps follow
// cdna_hip_programming.md section 3 (the same assumption the device code makes).
#pragma once

#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) {
	memset(p, value, bytes);
	return hipSuccess;
}

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline
#define TCNN_DEVICE inline
#define TCNN_HOST_DEVICE inline

namespace emu {

typedef _Float16 eh8 __attribute__((ext_vector_type(8)));
typedef _Float16 eh4 __attribute__((ext_vector_type(4)));
typedef _Float16 eh2 __attribute__((ext_vector_type(2)));
typedef float ef4 __attribute__((ext_vector_type(4)));

struct Fiber {
	ucontext_t ctx;
	dim3 tidx;
	bool done = false;
	char* stack = nullptr;
};

struct Wave {
	unsigned gen = 0, arrived = 0, size = 64;
	eh8 a[64], b[64];
	unsigned long long vote[2] = {0, 0};  // ballots of alternating rendezvous
};

struct State {
	dim3 grid, block, bidx;
	Fiber* cur = nullptr;
	std::vector<Fiber> fibers;
	std::vector<Wave> waves;
	ucontext_t sched;
	unsigned block_gen = 0, block_arrived = 0, n_threads = 0;
	unsigned long events = 0;
	std::vector<unsigned char> dyn;
	std::function<void()> fn;
};

inline State g;
constexpr size_t STACK_BYTES = 256 * 1024;

inline unsigned char* dyn_lds() { return g.dyn.data(); }
inline void yield() { swapcontext(&g.cur->ctx, &g.sched); }

inline void block_barrier() {
	const unsigned gen = g.block_gen;
	if (++g.block_arrived == g.n_threads) {
		g.block_arrived = 0;
		g.block_gen++;
		g.events++;
	} else {
		while (g.block_gen == gen) yield();
	}
}

inline void wave_barrier() {
	Wave& w = g.waves[g.cur->tidx.x / 64];
	const unsigned gen = w.gen;
	if (++w.arrived == w.size) {
		w.arrived = 0;
		w.gen++;
		g.events++;
	} else {
		while (w.gen == gen) yield();
	}
}

// __ballot for fully converged waves (every lane of the wave must execute it)
inline unsigned long long wave_ballot(bool pred) {
	Wave& w = g.waves[g.cur->tidx.x / 64];
	const unsigned slot = w.gen & 1u;
	if (w.arrived == 0) w.vote[slot] = 0;
	if (pred) w.vote[slot] |= 1ull << (g.cur->tidx.x & 63u);
	wave_barrier();
	return w.vote[slot];
}

// __shfl_xor for fully converged waves of 64
inline int wave_shfl_xor(int v, int mask) {
	Wave& w = g.waves[g.cur->tidx.x / 64];
	const unsigned lane = g.cur->tidx.x & 63u;
	w.a[lane][0] = 0;
	memcpy(&w.b[lane], &v, sizeof(int));
	wave_barrier();
	int out;
	memcpy(&out, &w.b[(lane ^ (unsigned)mask) & 63u], sizeof(int));
	wave_barrier();
	return out;
}

// wave_rows_transpose4 of the device code (tcnn_device.h): lane (g, c) receives t[r] = element g of what lane (r, c) passed
inline eh4 wave_rows_transpose4(eh4 v) {
	Wave& w = g.waves[g.cur->tidx.x / 64];
	const unsigned lane = g.cur->tidx.x & 63u, row = lane >> 4, col = lane & 15u;
	memcpy(&w.b[lane], &v, sizeof(eh4));
	wave_barrier();
	eh4 out;
	for (unsigned r = 0; r < 4; ++r) {
		eh4 src;
		memcpy(&src, &w.b[r * 16u + col], sizeof(eh4));
		out[r] = src[row];
	}
	wave_barrier();
	return out;
}

// sum over the wave in the xor-butterfly order of the device code (one rendezvous instead of six shuffles)
inline float wave_sum_f32(float v) {
	Wave& w = g.waves[g.cur->tidx.x / 64];
	const unsigned lane = g.cur->tidx.x & 63u;
	memcpy(&w.b[lane], &v, sizeof(float));
	wave_barrier();
	float vals[64], next[64];
	for (unsigned l = 0; l < 64; ++l) {
		if (l < w.size) memcpy(&vals[l], &w.b[l], sizeof(float));
		else vals[l] = 0.0f;
	}
	for (unsigned d = 32; d > 0; d >>= 1) {
		for (unsigned l = 0; l < 64; ++l) next[l] = vals[l] + vals[l ^ d];
		memcpy(vals, next, sizeof(vals));
	}
	wave_barrier();
	return vals[lane];
}

inline void trampoline() {
	g.fn();
	g.cur->done = true;
	g.events++;
	swapcontext(&g.cur->ctx, &g.sched);
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> fn) {
	g.grid = grid;
	g.block = block;
	g.n_threads = block.x * block.y * block.z;
	g.fn = std::move(fn);
	g.dyn.assign(shmem + 64, 0xCD);  // poison: kernels must not rely on zeroed LDS
	if (g.fibers.size() < g.n_threads) {
		const size_t old = g.fibers.size();
		g.fibers.resize(g.n_threads);
		for (size_t i = old; i < g.n_threads; ++i) g.fibers[i].stack = (char*)malloc(STACK_BYTES);
	}
	g.waves.assign((g.n_threads + 63) / 64, Wave());
	for (unsigned linear = 0; linear < grid.x * grid.y; ++linear) {  // (two-dimensional grids: x fastest, as the hardware dispatches them)
		const unsigned bx = linear % grid.x;
		g.bidx = dim3(bx, linear / grid.x, 0);
		g.block_gen = 0;
		g.block_arrived = 0;
		for (unsigned w = 0; w < g.waves.size(); ++w) {
			g.waves[w].gen = 0;
			g.waves[w].arrived = 0;
			g.waves[w].size = std::min(64u, g.n_threads - 64 * w);
		}
		for (unsigned t = 0; t < g.n_threads; ++t) {
			Fiber& f = g.fibers[t];
			f.tidx = dim3(t, 0, 0);
			f.done = false;
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = f.stack;
			f.ctx.uc_stack.ss_size = STACK_BYTES;
			f.ctx.uc_link = &g.sched;
			makecontext(&f.ctx, (void (*)())trampoline, 0);
		}
		unsigned n_done = 0;
		while (n_done < g.n_threads) {
			const unsigned long before = g.events;
			n_done = 0;
			for (unsigned t = 0; t < g.n_threads; ++t) {
				Fiber& f = g.fibers[t];
				if (f.done) {
					n_done++;
					continue;
				}
				g.cur = &f;
				swapcontext(&g.sched, &f.ctx);
				if (f.done) n_done++;
			}
			if (n_done < g.n_threads && g.events == before) {
				fprintf(stderr, "hip_emu: deadlock in block %u (a barrier was not reached by every thread)\n", bx);
				abort();
			}
		}
	}
}

}  // namespace emu

#define threadIdx (::emu::g.cur->tidx)
#define blockIdx (::emu::g.bidx)
#define blockDim (::emu::g.block)
#define gridDim (::emu::g.grid)
#define __syncthreads() ::emu::block_barrier()
#define __ballot(pred) ::emu::wave_ballot(pred)

#define TCNN_LAUNCH(kernel, grid, block, shmem, stream, ...) \
	::emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); })

#define __shfl_xor(v, mask, width) ::emu::wave_shfl_xor(v, mask)

template <typename T>
inline T min(T a, T b) { return a < b ? a : b; }
template <typename T>
inline T max(T a, T b) { return a < b ? b : a; }
inline uint32_t atomicMax(uint32_t* addr, uint32_t v) {
	const uint32_t old = *addr;
	if (v > old) *addr = v;
	return old;
}

namespace tcnn_hip {

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_f16:  A[i][k]: lane = i + 16*(k/8), element k%8;  B[k][n]: lane = n + 16*(k/8);
// D[row][col]: lane = col + 16*(row/4), element row%4.
inline f4 mfma_16x16x32(h8 a, h8 b, f4 c) {
	const unsigned lane = ::emu::g.cur->tidx.x & 63u;
	::emu::Wave& w = ::emu::g.waves[::emu::g.cur->tidx.x / 64];
	w.a[lane] = a;
	w.b[lane] = b;
	::emu::wave_barrier();
	f4 d = c;
	const unsigned col = lane & 15u, g = lane >> 4;
	for (unsigned r = 0; r < 4; ++r) {
		const unsigned row = 4 * g + r;
		float s = c[r];
		for (unsigned k = 0; k < 32; ++k) s += (float)w.a[row + 16 * (k / 8)][k % 8] * (float)w.b[col + 16 * (k / 8)][k % 8];
		d[r] = s;
	}
	::emu::wave_barrier();
	return d;
}

// v_mfma_f32_16x16x16_f16: 4 halves per lane, k = 4*(lane>>4) + j
inline f4 mfma_16x16x16(h4 a, h4 b, f4 c) {
	const unsigned lane = ::emu::g.cur->tidx.x & 63u;
	::emu::Wave& w = ::emu::g.waves[::emu::g.cur->tidx.x / 64];
	h8 a8 = {}, b8 = {};
	for (unsigned j = 0; j < 4; ++j) {
		a8[j] = a[j];
		b8[j] = b[j];
	}
	w.a[lane] = a8;
	w.b[lane] = b8;
	::emu::wave_barrier();
	f4 d = c;
	const unsigned col = lane & 15u, g = lane >> 4;
	for (unsigned r = 0; r < 4; ++r) {
		const unsigned row = 4 * g + r;
		float s = c[r];
		for (unsigned k = 0; k < 16; ++k) s += (float)w.a[row + 16 * (k / 4)][k % 4] * (float)w.b[col + 16 * (k / 4)][k % 4];
		d[r] = s;
	}
	::emu::wave_barrier();
	return d;
}

// ds_read_b64_tr_b16: lane c of a 16-lane group, element j <- element (c & 3) of the word lane 4j + (c >> 2) addressed
inline h4 lds_read_tr4(const _Float16* word) {
	const unsigned lane = ::emu::g.cur->tidx.x & 63u;
	::emu::Wave& w = ::emu::g.waves[::emu::g.cur->tidx.x / 64];
	h8 mine = {};
	for (unsigned j = 0; j < 4; ++j) mine[j] = word[j];
	w.a[lane] = mine;
	::emu::wave_barrier();
	const unsigned grp = lane & ~15u, c = lane & 15u;
	h4 out;
	for (unsigned j = 0; j < 4; ++j) out[j] = w.a[grp + 4 * j + (c >> 2)][c & 3u];
	::emu::wave_barrier();
	return out;
}

inline _Float16 emu_round_h(double v) { return (_Float16)v; }  // double -> half is a single RNE rounding

inline void atomic_add_h2(half_t* addr, h2 v) {
	addr[0] = emu_round_h((double)addr[0] + (double)v[0]);
	addr[1] = emu_round_h((double)addr[1] + (double)v[1]);
}
inline void atomic_add_f32(float* addr, float v) { *addr = *addr + v; }
inline void lds_atomic_add_f32(float* addr, float v) { *addr = *addr + v; }
inline void lds_atomic_add_u64(unsigned long long* addr, unsigned long long v) { *addr = *addr + v; }
inline void lds_atomic_add_h2(h2* addr, h2 v) { atomic_add_h2((half_t*)addr, v); }
inline uint32_t atomic_add_u32(uint32_t* addr, uint32_t v) { const uint32_t old = *addr; *addr = old + v; return old; }
inline h2 fma_h2(h2 a, h2 b, h2 c) {
	return h2{emu_round_h((double)a[0] * (double)b[0] + (double)c[0]), emu_round_h((double)a[1] * (double)b[1] + (double)c[1])};
}
inline half_t fma_h(half_t a, half_t b, half_t c) { return emu_round_h((double)a * (double)b + (double)c); }
inline uint32_t xcc_id() { return ::emu::g.bidx.x & 7u; }

}  // namespace tcnn_hip
