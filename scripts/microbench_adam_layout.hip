// Which layout of Adam's state moves fewer bytes per second of wall clock when only a FRACTION of the table entries is stepped?
// (configs[4] -- T = 2^22, 89 M parameters, ~40 % of the entries touched per batch -- spends 0.50 of its 1.18 ms
// in k_adam_step streaming the dense 3.2 GB of SoA state, because nearly every 128-byte line of `master`, `m`, `v` holds a touched
// neighbour.  Gate: keep grid-parameter state entry-major inside the trainer only if this shows >= 1.3x.)
//
//   SoA    what the trainer keeps: master[n], m1[n], m2[n] (fp32), grads[n] + params[n] (fp16); a lane owns 4 consecutive parameters,
//          the 8 lanes of a 128-byte line vote (`dense_store`): a line is read and written whole or not at all
//   AoS32  entry-major records {master x2, m1 x2, m2 x2, pad x2} = 32 B per table entry (F = 2 parameters); grads / params stay fp16 SoA
//          (the gather reads params, the backward writes grads); a lane owns ONE entry, reads its 4-byte gradient pair, and touches
//          the record only if either gradient is non-zero
//   AoS24  the same without padding (24 B records: 3 x 8-byte accesses, records straddle 32-byte sectors)
// The arithmetic is Adam's (adam.h:48-127: two moment updates, bias correction, sqrt, divide) so that the kernels are not pure copies.
// Touched entries: Bernoulli(density) per ENTRY, independent (what random sample positions do to a hashed level).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_adam_layout.hip -o scripts/microbench_adam_layout.bin
// Output: one line per (layout, density): microseconds per pass, bytes the layout must move at its own granularity, GB/s.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
	do {                                                                                      \
		hipError_t e_ = (x);                                                                  \
		if (e_ != hipSuccess) {                                                               \
			fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(1);                                                                          \
		}                                                                                     \
	} while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr uint32_t THREADS = 256;

__device__ __forceinline__ void adam_one(float g, float& w, float& m1, float& m2) {
	const float b1 = 0.9f, b2 = 0.99f, lr = 1e-2f, eps = 1e-15f;
	g *= (1.0f / 128.0f);
	m1 = b1 * m1 + (1.0f - b1) * g;
	m2 = b2 * m2 + (1.0f - b2) * g * g;
	w -= lr * (m1 * 10.0f) / (sqrtf(m2 * 100.0f) + eps);
}

// ---- SoA, line-voting (the product's form, reduced to its memory behaviour) ----
__global__ void __launch_bounds__(THREADS) k_soa(uint32_t n, float* __restrict__ master, float* __restrict__ m1, float* __restrict__ m2, const _Float16* __restrict__ grads,
                                                  _Float16* __restrict__ params) {
	const uint32_t i0 = (blockIdx.x * THREADS + threadIdx.x) * 4u;
	if (i0 >= n) return;
	const h4 g = *(const h4*)(grads + i0);
	const bool any = g[0] != (_Float16)0 || g[1] != (_Float16)0 || g[2] != (_Float16)0 || g[3] != (_Float16)0;
	// the 8 lanes that share a 128-byte line of fp32 state vote
	const unsigned long long b = __ballot(any);
	const uint32_t lane = threadIdx.x & 63u;
	const bool line = ((b >> (lane & ~7u)) & 0xFFull) != 0ull;
	if (!line) return;
	f4 w = __builtin_nontemporal_load((const f4*)(master + i0)), a = __builtin_nontemporal_load((const f4*)(m1 + i0)), c = __builtin_nontemporal_load((const f4*)(m2 + i0));
	h4 wh;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		float wj = w[j], aj = a[j], cj = c[j];
		if (g[j] != (_Float16)0) adam_one((float)g[j], wj, aj, cj);
		w[j] = wj;
		a[j] = aj;
		c[j] = cj;
		wh[j] = (_Float16)wj;
	}
	__builtin_nontemporal_store(w, (f4*)(master + i0));
	__builtin_nontemporal_store(a, (f4*)(m1 + i0));
	__builtin_nontemporal_store(c, (f4*)(m2 + i0));
	*(h4*)(params + i0) = wh;
}

// ---- entry-major records ----
struct alignas(32) Rec32 {
	f2 w, a, c, pad;
};
struct Rec24 {
	f2 w, a, c;
};
template <typename REC>
__global__ void __launch_bounds__(THREADS) k_aos(uint32_t n_entries, REC* __restrict__ state, const _Float16* __restrict__ grads, _Float16* __restrict__ params) {
	const uint32_t e = blockIdx.x * THREADS + threadIdx.x;
	if (e >= n_entries) return;
	const h2 g = *(const h2*)(grads + 2u * e);
	if (g[0] == (_Float16)0 && g[1] == (_Float16)0) return;
	REC* r = state + e;
	const f2 w = r->w, a = r->a, c = r->c;
	float w0 = w[0], w1 = w[1], a0 = a[0], a1 = a[1], c0 = c[0], c1 = c[1];
	adam_one((float)g[0], w0, a0, c0);
	adam_one((float)g[1], w1, a1, c1);
	r->w = f2{w0, w1};
	r->a = f2{a0, a1};
	r->c = f2{c0, c1};
	*(h2*)(params + 2u * e) = h2{(_Float16)w0, (_Float16)w1};
}
// two entries per lane: 64-byte granules per lane, 16-byte gradient reads per lane pair
template <typename REC>
__global__ void __launch_bounds__(THREADS) k_aos2(uint32_t n_entries, REC* __restrict__ state, const _Float16* __restrict__ grads, _Float16* __restrict__ params) {
	const uint32_t e0 = (blockIdx.x * THREADS + threadIdx.x) * 2u;
	if (e0 >= n_entries) return;
	const h4 g = *(const h4*)(grads + 2u * e0);
	h4 wh = *(const h4*)(params + 2u * e0);
	bool any = false;
#pragma unroll
	for (uint32_t k = 0; k < 2; ++k) {
		if (g[2 * k] == (_Float16)0 && g[2 * k + 1] == (_Float16)0) continue;
		REC* r = state + e0 + k;
		const f2 w = r->w, a = r->a, c = r->c;
		float w0 = w[0], w1 = w[1], a0 = a[0], a1 = a[1], c0 = c[0], c1 = c[1];
		adam_one((float)g[2 * k], w0, a0, c0);
		adam_one((float)g[2 * k + 1], w1, a1, c1);
		r->w = f2{w0, w1};
		r->a = f2{a0, a1};
		r->c = f2{c0, c1};
		wh[2 * k] = (_Float16)w0;
		wh[2 * k + 1] = (_Float16)w1;
		any = true;
	}
	if (any) *(h4*)(params + 2u * e0) = wh;
}

__global__ void k_fill_grads(uint32_t n_entries, float density, uint32_t seed, _Float16* __restrict__ grads) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_entries) return;
	uint32_t h = e * 2654435761u ^ seed;  // a hash per entry: independent Bernoulli(density)
	h ^= h >> 16;
	h *= 0x7feb352du;
	h ^= h >> 15;
	h *= 0x846ca68bu;
	h ^= h >> 16;
	const bool touched = (float)(h >> 8) * (1.0f / 16777216.0f) < density;
	grads[2 * e] = touched ? (_Float16)0.25f : (_Float16)0;
	grads[2 * e + 1] = touched ? (_Float16)-0.5f : (_Float16)0;
}

template <typename F>
static float time_us(F&& launch, int reps = 20) {
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	for (int i = 0; i < 3; ++i) launch();
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; ++i) launch();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms * 1000.0f / reps;
}

int main(int argc, char** argv) {
	const uint32_t n_entries = argc > 1 ? (uint32_t)atoll(argv[1]) : 44540024u;  // configs[4]: 89 080 048 parameters / 2
	const uint32_t n = n_entries * 2;
	printf("entries %u (parameters %u)\n", n_entries, n);
	float *master, *m1, *m2;
	_Float16 *grads, *params;
	Rec32* s32;
	Rec24* s24;
	CHECK(hipMalloc(&master, (size_t)n * 4));
	CHECK(hipMalloc(&m1, (size_t)n * 4));
	CHECK(hipMalloc(&m2, (size_t)n * 4));
	CHECK(hipMalloc(&grads, (size_t)n * 2));
	CHECK(hipMalloc(&params, (size_t)n * 2));
	CHECK(hipMalloc(&s32, (size_t)n_entries * sizeof(Rec32)));
	CHECK(hipMalloc(&s24, (size_t)n_entries * sizeof(Rec24)));
	CHECK(hipMemset(master, 0, (size_t)n * 4));
	CHECK(hipMemset(m1, 0, (size_t)n * 4));
	CHECK(hipMemset(m2, 0, (size_t)n * 4));
	CHECK(hipMemset(params, 0, (size_t)n * 2));
	CHECK(hipMemset(s32, 0, (size_t)n_entries * sizeof(Rec32)));
	CHECK(hipMemset(s24, 0, (size_t)n_entries * sizeof(Rec24)));
	const float densities[] = {1.0f, 0.8f, 0.6f, 0.4f, 0.3f, 0.2f, 0.1f, 0.05f};
	for (float d : densities) {
		k_fill_grads<<<(n_entries + 255) / 256, 256>>>(n_entries, d, 12345u, grads);
		CHECK(hipDeviceSynchronize());
		const double touched = (double)n_entries * d;
		// bytes each layout has to move at ITS granularity: SoA -- 128-byte lines of fp32 state (32 parameters = 16 entries) that hold a
		// touched entry, three arrays read + written, + all gradients + the 16-bit weights of those lines; AoS -- the touched records read +
		// written, + all gradients + the touched 16-bit pairs
		const double p_line = 1.0 - pow(1.0 - d, 16.0);
		const double soa_bytes = (double)n * 2 + p_line * (double)n * (24.0 + 2.0);
		const double aos32_bytes = (double)n * 2 + touched * (2.0 * 24.0 + 4.0);
		const float t_soa = time_us([&] { k_soa<<<(n / 4 + THREADS - 1) / THREADS, THREADS>>>(n, master, m1, m2, grads, params); });
		const float t_32 = time_us([&] { k_aos<Rec32><<<(n_entries + THREADS - 1) / THREADS, THREADS>>>(n_entries, s32, grads, params); });
		const float t_24 = time_us([&] { k_aos<Rec24><<<(n_entries + THREADS - 1) / THREADS, THREADS>>>(n_entries, s24, grads, params); });
		const float t_322 = time_us([&] { k_aos2<Rec32><<<(n_entries / 2 + THREADS - 1) / THREADS, THREADS>>>(n_entries, s32, grads, params); });
		printf("density %.2f | SoA line-voting %8.1f us (%6.0f MB at its granularity, %5.0f GB/s) | AoS32 %8.1f us (%6.0f MB, %5.0f GB/s) x%.2f | AoS32 2/lane %8.1f us x%.2f | AoS24 %8.1f us x%.2f\n",
		       d, t_soa, soa_bytes / 1e6, soa_bytes / t_soa / 1e3, t_32, aos32_bytes / 1e6, aos32_bytes / t_32 / 1e3, t_soa / t_32, t_322, t_soa / t_322, t_24, t_soa / t_24);
	}
	return 0;
}
