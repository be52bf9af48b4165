// direct_exchange.hip -- see direct_exchange.h for the scheme and its status.
#include "direct_exchange.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace tcnn_hip {

static_assert(sizeof(hipIpcMemHandle_t) == DIRECT_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");

struct PeerBuffers {
	half_t* p[DIRECT_MAX_RANKS];
};
struct PeerSignals {
	uint32_t* p[DIRECT_MAX_RANKS];
};
constexpr uint32_t DX_THREADS = 256;

// Accesses to a PEER's memory carry system scope (sc0 sc1 on gfx950): a load is served by the memory it names, not by a line this GPU's
// L2s kept from the step before, and a store is written through to the peer instead of resting in an L2 here.  The trainer buffers are
// ordinary (coarse-grained) hipMalloc memory, for which the caches are only reconciled at kernel boundaries -- and which of the eight
// XCD-private L2s a boundary's fences reach is the runtime's business; with the scope on the instruction itself the data path does not
// depend on it (the signal words' release / acquire order the accesses, the self-test checks the whole arrangement on the node).
// 8-byte relaxed atomics: what the scope builtins offer; a lane's 16 bytes are two of them.
TCNN_DEVICE h8 peer_load16(const half_t* p) {
#if defined(TCNN_HOST_EMU)
	return *(const h8*)p;
#else
	const unsigned long long* q = (const unsigned long long*)p;
	const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	return __builtin_bit_cast(h8, (ull2){lo, hi});
#endif
}
TCNN_DEVICE void peer_store16(half_t* p, h8 v) {
#if defined(TCNN_HOST_EMU)
	*(h8*)p = v;
#else
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	const ull2 w = __builtin_bit_cast(ull2, v);
	unsigned long long* q = (unsigned long long*)p;
	__hip_atomic_store(q, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	__hip_atomic_store(q + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
TCNN_DEVICE half_t peer_load2(const half_t* p) {
#if defined(TCNN_HOST_EMU)
	return *p;
#else
	const unsigned short w = __hip_atomic_load((const unsigned short*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	return __builtin_bit_cast(half_t, w);
#endif
}
TCNN_DEVICE void peer_store2(half_t* p, half_t v) {
#if defined(TCNN_HOST_EMU)
	*p = v;
#else
	__hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// "step s of rank `me`, phase `row`, is done": one 4-byte store into every rank's signal block (the own one included), released at
// system scope -- everything this rank's earlier kernels on the stream wrote is visible to whoever then reads the counter.
__global__ void k_direct_signal(const PeerSignals peers, const int n_ranks, const int me, const int row, const uint32_t step) {
	if ((int)threadIdx.x < n_ranks) {
		__hip_atomic_store(&peers.p[threadIdx.x][row * DIRECT_MAX_RANKS + me], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
// lane r waits until rank r has signalled `step` in `row` of THIS rank's block; gives up after `timeout_ticks` of the 100 MHz wall
// clock and records which phase starved (the queue moves on: wrong results, reported by direct_exchange_status, instead of a hung GPU)
__global__ void k_direct_wait(const uint32_t* own, const int n_ranks, const int row, const uint32_t step, const uint64_t timeout_ticks, uint32_t* error_flag) {
	if ((int)threadIdx.x < n_ranks) {
		const uint64_t t0 = wall_clock64();
		while ((int32_t)(__hip_atomic_load(&own[row * DIRECT_MAX_RANKS + threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
			if (wall_clock64() - t0 > timeout_ticks) {
				atomicExch(error_flag, (uint32_t)(1 + row));
				break;
			}
			__builtin_amdgcn_s_sleep(64);
		}
	}
}
// this rank's shard [begin, begin + count) of the summed gradient: fp32 sum over the ranks IN RANK ORDER, one rounding.  8 parameters
// (16 bytes) per lane and iteration, one load per rank in flight; P - 1 of them cross a link each.
__global__ void __launch_bounds__(DX_THREADS) k_direct_reduce(const PeerBuffers grads, const int n_ranks, const int me, const size_t begin, const size_t count) {
	const size_t n8 = count / 8;
	for (size_t i = (size_t)blockIdx.x * DX_THREADS + threadIdx.x; i < n8; i += (size_t)gridDim.x * DX_THREADS) {
		float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		h8 v[DIRECT_MAX_RANKS];  // (fully unrolled: registers; every rank's load is issued before the first sum)
#pragma unroll
		for (int r = 0; r < DIRECT_MAX_RANKS; ++r) {
			if (r < n_ranks) v[r] = r == me ? *(const h8*)(grads.p[r] + begin + 8 * i) : peer_load16(grads.p[r] + begin + 8 * i);
		}
#pragma unroll
		for (int r = 0; r < DIRECT_MAX_RANKS; ++r) {
			if (r < n_ranks) {
#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) acc[k] = acc[k] + (float)v[r][k];
			}
		}
		h8 out;
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) out[k] = to_half_rn(acc[k]);
		*(h8*)(grads.p[me] + begin + 8 * i) = out;
	}
	// what does not fill a 16-byte group (only the LAST rank's shard can end off a multiple of 8: it carries the remainder of the
	// buffer): element by element, by the first workgroup.  Nobody else writes -- or steps -- these elements.
	if (blockIdx.x == 0) {
		for (size_t e = n8 * 8 + threadIdx.x; e < count; e += DX_THREADS) {
			float acc = 0.0f;
			for (int r = 0; r < n_ranks; ++r) acc = acc + (float)(r == me ? grads.p[r][begin + e] : peer_load2(grads.p[r] + begin + e));
			grads.p[me][begin + e] = to_half_rn(acc);
		}
	}
}
// the own stepped parameter shard into every peer's parameter buffer
__global__ void __launch_bounds__(DX_THREADS) k_direct_push(const PeerBuffers params, const int n_ranks, const int me, const size_t begin, const size_t count) {
	const size_t n8 = count / 8;  // shards begin on multiples of 8; only the last rank's may end off one
	for (size_t i = (size_t)blockIdx.x * DX_THREADS + threadIdx.x; i < n8; i += (size_t)gridDim.x * DX_THREADS) {
		const h8 v = *(const h8*)(params.p[me] + begin + 8 * i);
		for (int r = 0; r < n_ranks; ++r) {
			if (r != me) peer_store16(params.p[r] + begin + 8 * i, v);
		}
	}
	if (blockIdx.x == 0) {
		for (size_t e = n8 * 8 + threadIdx.x; e < count; e += DX_THREADS) {
			const half_t v = params.p[me][begin + e];
			for (int r = 0; r < n_ranks; ++r) {
				if (r != me) peer_store2(params.p[r] + begin + e, v);
			}
		}
	}
}

// ---- link check (direct_exchange_selftest): a pattern only (rank, element, round, seed) determine; multiples of 1/16 below 1/2, so that the
// sum over up to 16 ranks (< 8: seven bits) is exact in fp16 AND bfloat16 whatever the order
TCNN_HOST_DEVICE float selftest_pattern(uint32_t rank, uint64_t i, uint32_t round, uint32_t seed) {
	return (float)((uint32_t)((i * 7ull + rank * 13ull + round * 29ull + seed * 5ull) % 8ull)) * 0.0625f;
}
__global__ void __launch_bounds__(DX_THREADS) k_direct_selftest_fill(half_t* grads, const size_t n, const uint32_t rank, const uint32_t round, const uint32_t seed) {
	for (size_t i = (size_t)blockIdx.x * DX_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DX_THREADS) grads[i] = (half_t)selftest_pattern(rank, i, round, seed);
}
// after reduce + push every element of the own buffer must hold the sum of all ranks' patterns of THIS round
__global__ void __launch_bounds__(DX_THREADS) k_direct_selftest_check(const half_t* grads, const size_t n, const uint32_t n_ranks, const uint32_t round, const uint32_t seed,
                                                                      unsigned long long* mismatches) {
	unsigned long long bad = 0;
	for (size_t i = (size_t)blockIdx.x * DX_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DX_THREADS) {
		float expected = 0.0f;
		for (uint32_t r = 0; r < n_ranks; ++r) expected += selftest_pattern(r, i, round, seed);
		if ((float)grads[i] != expected) ++bad;
	}
	if (bad) atomicAdd(mismatches, bad);
}

static void hip_ok(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string("direct exchange: ") + what + ": " + hipGetErrorString(e));
}

void direct_exchange_export(DirectExchange& dx, void* buffer, const half_t* params, const half_t* grads, uint64_t n_params, DirectExport& out) {
	if (!dx.own_signals) {
		const size_t bytes = (2 * DIRECT_MAX_RANKS + 16) * sizeof(uint32_t);
		// signal words are polled while peers write them: uncached device memory where the runtime offers it
		void* p = nullptr;
		if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
			(void)hipGetLastError();
			hip_ok(hipMalloc(&p, bytes), "hipMalloc(signals)");
		}
		hip_ok(hipMemset(p, 0, bytes), "hipMemset(signals)");
		hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
		dx.own_signals = (uint32_t*)p;
		dx.error_flag = dx.own_signals + 2 * DIRECT_MAX_RANKS;
		void* h = nullptr;
		hip_ok(hipHostMalloc(&h, sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc(error word)");
		dx.host_error = (volatile uint32_t*)h;
	}
	std::memset(&out, 0, sizeof(out));
	hipIpcMemHandle_t h;
	hip_ok(hipIpcGetMemHandle(&h, buffer), "hipIpcGetMemHandle(trainer buffer) -- direct exchange needs the plain allocator (no TCNN_DEBUG_ALLOC)");
	std::memcpy(out.buffer_handle, &h, DIRECT_HANDLE_BYTES);
	hip_ok(hipIpcGetMemHandle(&h, dx.own_signals), "hipIpcGetMemHandle(signals)");
	std::memcpy(out.signal_handle, &h, DIRECT_HANDLE_BYTES);
	out.params_offset = (uint64_t)((const char*)params - (const char*)buffer);
	out.grads_offset = (uint64_t)((const char*)grads - (const char*)buffer);
	out.n_params = n_params;
}

void direct_exchange_open(DirectExchange& dx, int rank, int n_ranks, const DirectExport* exports, half_t* own_params, half_t* own_grads) {
	if (n_ranks < 1 || n_ranks > DIRECT_MAX_RANKS || rank < 0 || rank >= n_ranks) throw std::runtime_error("direct exchange: rank / n_ranks out of range (at most 16 ranks)");
	if (!dx.own_signals) throw std::runtime_error("direct exchange: export before open");
	direct_exchange_close(dx);
	const uint64_t n = exports[rank].n_params;
	for (int r = 0; r < n_ranks; ++r) {
		if (exports[r].n_params != n) throw std::runtime_error("direct exchange: the ranks' models differ in their parameter count");
	}
	for (int r = 0; r < n_ranks; ++r) {
		if (r == rank) {
			dx.grads[r] = own_grads;
			dx.params[r] = own_params;
			dx.signals[r] = dx.own_signals;
			continue;
		}
		hipIpcMemHandle_t h;
		std::memcpy(&h, exports[r].buffer_handle, DIRECT_HANDLE_BYTES);
		hip_ok(hipIpcOpenMemHandle(&dx.mapped_buffers[r], h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(peer trainer buffer)");
		std::memcpy(&h, exports[r].signal_handle, DIRECT_HANDLE_BYTES);
		hip_ok(hipIpcOpenMemHandle(&dx.mapped_signals[r], h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(peer signals)");
		dx.grads[r] = (half_t*)((char*)dx.mapped_buffers[r] + exports[r].grads_offset);
		dx.params[r] = (half_t*)((char*)dx.mapped_buffers[r] + exports[r].params_offset);
		dx.signals[r] = (uint32_t*)dx.mapped_signals[r];
	}
	dx.rank = rank;
	dx.n_ranks = n_ranks;
	dx.n_params = n;
	// rank r owns [r * shard, (r + 1) * shard); the LAST rank's shard runs to the end of the buffer (the < 8 P parameters that do not divide
	// are its own too).  Every parameter has exactly one owner: one rank reads the peers' gradients of it, writes its sum, steps it and
	// pushes it -- nothing is reduced in place by several ranks at once (round 4 had everyone reduce a replicated tail: a cross-rank race).
	dx.shard = (size_t)(n / (8ull * n_ranks)) * 8;
	dx.own_begin = (size_t)rank * dx.shard;
	dx.own_end = rank == n_ranks - 1 ? (size_t)n : (size_t)(rank + 1) * dx.shard;
	dx.step = 0;
	const char* e = getenv("TCNN_DIRECT_TIMEOUT_MS");
	dx.timeout_ms = e ? (uint32_t)atoi(e) : 2000u;
	hip_ok(hipMemset(dx.own_signals, 0, (2 * DIRECT_MAX_RANKS + 16) * sizeof(uint32_t)), "hipMemset(signals)");
	hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
	*dx.host_error = 0;
	dx.step_recorded = false;
}

void direct_exchange_close(DirectExchange& dx) {
	if (dx.step_done) {
		(void)hipEventSynchronize((hipEvent_t)dx.step_done);
		(void)hipEventDestroy((hipEvent_t)dx.step_done);
		dx.step_done = nullptr;
		dx.step_recorded = false;
	}
	for (int r = 0; r < DIRECT_MAX_RANKS; ++r) {
		if (dx.mapped_buffers[r]) (void)hipIpcCloseMemHandle(dx.mapped_buffers[r]);
		if (dx.mapped_signals[r]) (void)hipIpcCloseMemHandle(dx.mapped_signals[r]);
		dx.mapped_buffers[r] = dx.mapped_signals[r] = nullptr;
		dx.grads[r] = dx.params[r] = nullptr;
		dx.signals[r] = nullptr;
	}
	dx.n_ranks = 0;
	dx.rank = -1;
}

static void signal_and_wait(hipStream_t stream, DirectExchange& dx, int row) {
	PeerSignals ps;
	for (int r = 0; r < DIRECT_MAX_RANKS; ++r) ps.p[r] = dx.signals[r];
	const uint64_t ticks = (uint64_t)dx.timeout_ms * 100000ull;  // wall_clock64: 100 MHz
	TCNN_LAUNCH(k_direct_signal, dim3(1), dim3(64), 0, stream, ps, dx.n_ranks, dx.rank, row, dx.step);
	TCNN_LAUNCH(k_direct_wait, dim3(1), dim3(64), 0, stream, (const uint32_t*)dx.own_signals, dx.n_ranks, row, dx.step, ticks, dx.error_flag);
}

// ---- a step's exchange, phase by phase (the trainer brackets each with profiler events so that a node run explains itself) ------------
void direct_exchange_begin_step(DirectExchange& dx) {
	if (!dx.active()) throw std::runtime_error("direct exchange: not open");
	// The error word as of the step before: a wait that gave up means this replica stepped on unreduced gradients or stale parameters -- fail
	// here instead of training on.  The pinned copy is refreshed by a copy queued behind every step, and the host runs ahead of the GPU: the
	// event behind that copy is waited for here, so that the word read IS the previous step's (every step is serialised against the peers
	// anyway; the host keeps one step of lead, not several steps on gradients nobody reduced).
	if (dx.step_done && dx.step_recorded) hip_ok(hipEventSynchronize((hipEvent_t)dx.step_done), "hipEventSynchronize(previous step of the direct exchange)");
	if (dx.host_error && *dx.host_error) {
		throw std::runtime_error("direct exchange: a wait for the peers' " + std::string(*dx.host_error == 1 ? "gradients" : "parameters") +
		                         " timed out in an earlier step (TCNN_DIRECT_TIMEOUT_MS): the replicas are no longer in lock-step");
	}
	++dx.step;
}
void direct_exchange_signal_wait(hipStream_t stream, DirectExchange& dx, int row) { signal_and_wait(stream, dx, row); }
void direct_exchange_reduce_own(hipStream_t stream, DirectExchange& dx) {
	PeerBuffers g;
	for (int r = 0; r < DIRECT_MAX_RANKS; ++r) g.p[r] = dx.grads[r];
	if (dx.own_count()) TCNN_LAUNCH(k_direct_reduce, dim3(dx.blocks()), dim3(DX_THREADS), 0, stream, g, dx.n_ranks, dx.rank, dx.own_begin, dx.own_count());
}
void direct_exchange_push_own(hipStream_t stream, DirectExchange& dx) {
	PeerBuffers p;
	for (int r = 0; r < DIRECT_MAX_RANKS; ++r) p.p[r] = dx.params[r];
	if (dx.own_count() && dx.n_ranks > 1) TCNN_LAUNCH(k_direct_push, dim3(dx.blocks()), dim3(DX_THREADS), 0, stream, p, dx.n_ranks, dx.rank, dx.own_begin, dx.own_count());
}
void direct_exchange_finish_step(hipStream_t stream, DirectExchange& dx) {
	if (dx.host_error) hip_ok(hipMemcpyAsync((void*)dx.host_error, dx.error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(error word)");
	if (!dx.step_done) {
		hipEvent_t e;
		hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
		dx.step_done = (void*)e;
	}
	hip_ok(hipEventRecord((hipEvent_t)dx.step_done, stream), "hipEventRecord");
	dx.step_recorded = true;
}

void direct_exchange_reduce(hipStream_t stream, DirectExchange& dx) {
	direct_exchange_begin_step(dx);
	signal_and_wait(stream, dx, 0);  // everybody's gradients of this step are final
	direct_exchange_reduce_own(stream, dx);
}

void direct_exchange_push(hipStream_t stream, DirectExchange& dx) {
	if (!dx.active()) throw std::runtime_error("direct exchange: not open");
	direct_exchange_push_own(stream, dx);
	signal_and_wait(stream, dx, 1);  // everybody's parameters have arrived here; nobody reads this rank's gradients any more
	direct_exchange_finish_step(stream, dx);
}

// Link check before the exchange is trusted with a training run: `rounds` times every rank fills its GRADIENT buffer with a pattern of the
// round, the ranks reduce their shards exactly as a step does (signal, wait, read the peers), push the reduced shard into every peer's
// gradient buffer exactly as a step pushes parameters (peer writes, signal, wait), and every rank compares its whole buffer with the sum it
// must hold.  A stale line anywhere (a poll that never sees the peer's store, a remote read served from a cache that kept last round's
// data, a push that had not landed when the signal did) shows up as mismatches or as a timed-out wait.  Collective: every rank must call it
// with the same rounds and seed, between steps; it clobbers the gradient buffer and nothing else.
void direct_exchange_selftest(hipStream_t stream, DirectExchange& dx, uint32_t rounds, uint32_t seed, uint64_t* mismatches, int* status) {
	if (!dx.active()) throw std::runtime_error("direct exchange: not open");
	unsigned long long* counter = nullptr;
	hip_ok(hipMalloc((void**)&counter, sizeof(*counter)), "hipMalloc(self-test counter)");
	hip_ok(hipMemsetAsync(counter, 0, sizeof(*counter), stream), "hipMemsetAsync");
	PeerBuffers g;
	for (int r = 0; r < DIRECT_MAX_RANKS; ++r) g.p[r] = dx.grads[r];
	half_t* own = dx.grads[dx.rank];
	const size_t n = (size_t)dx.n_params;
	const uint32_t blocks = (uint32_t)std::min<size_t>(div_round_up<size_t>(n, DX_THREADS), 4096);
	for (uint32_t round = 0; round < rounds; ++round) {
		TCNN_LAUNCH(k_direct_selftest_fill, dim3(blocks), dim3(DX_THREADS), 0, stream, own, n, (uint32_t)dx.rank, round, seed);
		direct_exchange_reduce(stream, dx);
		if (dx.own_count() && dx.n_ranks > 1) TCNN_LAUNCH(k_direct_push, dim3(dx.blocks()), dim3(DX_THREADS), 0, stream, g, dx.n_ranks, dx.rank, dx.own_begin, dx.own_count());
		signal_and_wait(stream, dx, 1);
		TCNN_LAUNCH(k_direct_selftest_check, dim3(blocks), dim3(DX_THREADS), 0, stream, (const half_t*)own, n, (uint32_t)dx.n_ranks, round, seed, counter);
	}
	unsigned long long bad = 0;
	hip_ok(hipMemcpyAsync(&bad, counter, sizeof(bad), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(self-test counter)");
	hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
	(void)hipFree(counter);
	if (mismatches) *mismatches = (uint64_t)bad;
	const int st = direct_exchange_status(stream, dx);
	if (dx.host_error) *dx.host_error = (uint32_t)st;  // (the stream is idle: what the first step's begin_step reads is the self-test's outcome)
	if (status) *status = st;
}

int direct_exchange_status(hipStream_t stream, DirectExchange& dx) {
	if (!dx.error_flag) return 0;
	uint32_t v = 0;
	hip_ok(hipMemcpyAsync(&v, dx.error_flag, sizeof(v), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(error flag)");
	hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
	return (int)v;
}

}  // namespace tcnn_hip
