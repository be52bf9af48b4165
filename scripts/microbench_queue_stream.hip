// What rate can the bucket queues of the grid backward be streamed at on one MI355X, and in which form?
//   read side  (pass B, k_grid_bucket_owner): 1024 queues x 16384 pair records, one 512-thread workgroup per queue, 64 KiB of LDS each
//   write side (pass A, k_grid_bucket_scatter): 256-thread workgroups append runs of 32 records to 64 queues per 512-sample tile
// Build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_queue_stream.hip -o scripts/microbench_queue_stream.bin
// Output: one line per variant -- microseconds per pass and GB/s over the bytes the variant moves.
#include <hip/hip_runtime.h>
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

constexpr uint32_t N_QUEUES = 1024, COUNT = 16384, CAP = 16384 + 512;
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
extern __shared__ unsigned char lds_raw[];

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) {
	if constexpr (NT) return __builtin_nontemporal_load(p);
	else return *p;
}

// ---- read side -------------------------------------------------------------------------------------------------
// MODE 0: the product's form -- U 12-byte records per lane and round (lane stride 12 B), the next round after the previous one is consumed
// MODE 1: 16-byte loads over the same bytes (a lane's loads no longer start on record boundaries)
// MODE 2: rolling: two half-rounds in flight, the next half-round issued before the previous one is consumed
// MODE 3: 10-byte records as two arrays: 8-byte payloads + 2-byte headers
template <uint32_t THREADS, int MODE, bool NT, uint32_t U, bool EPILOGUE>
__global__ void __launch_bounds__(THREADS) k_read(const uint32_t* __restrict__ queues, const uint32_t* __restrict__ counts, uint32_t* __restrict__ out,
                                                   uint32_t* __restrict__ slices, uint32_t n_queues) {
	uint32_t acc = 0;
	for (uint32_t queue = blockIdx.x; queue < n_queues; queue += gridDim.x) {
		const uint32_t count = counts[queue];
		const uint32_t* q = queues + (size_t)queue * CAP * 3;
		if (EPILOGUE) {  // table clear
			for (uint32_t e = threadIdx.x; e < 4096; e += THREADS) ((u4*)lds_raw)[e] = u4{0, 0, 0, 0};
			__syncthreads();
		}
		if constexpr (MODE == 0) {
			for (uint32_t base = threadIdx.x; base < count; base += THREADS * U) {
				u3 rec[U];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) rec[u] = ld<NT>((const u3*)(q + (size_t)min(base + u * THREADS, count - 1) * 3));
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) acc ^= rec[u][0] ^ rec[u][1] ^ rec[u][2];
			}
		} else if constexpr (MODE == 1) {
			const uint32_t n16 = count * 3 / 4;
			for (uint32_t base = threadIdx.x; base < n16; base += THREADS * U) {
				u4 rec[U];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) rec[u] = ld<NT>((const u4*)q + min(base + u * THREADS, n16 - 1));
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) acc ^= rec[u][0] ^ rec[u][1] ^ rec[u][2] ^ rec[u][3];
			}
		} else if constexpr (MODE == 2) {
			constexpr uint32_t H = U / 2;
			u3 a[H], b[H];
			uint32_t base = threadIdx.x;
#pragma unroll
			for (uint32_t u = 0; u < H; ++u) a[u] = ld<NT>((const u3*)(q + (size_t)min(base + u * THREADS, count - 1) * 3));
			base += THREADS * H;
			for (; base < count + THREADS * H; base += THREADS * 2 * H) {
#pragma unroll
				for (uint32_t u = 0; u < H; ++u) b[u] = ld<NT>((const u3*)(q + (size_t)min(base + u * THREADS, count - 1) * 3));
#pragma unroll
				for (uint32_t u = 0; u < H; ++u) acc ^= a[u][0] ^ a[u][1] ^ a[u][2];
#pragma unroll
				for (uint32_t u = 0; u < H; ++u) a[u] = ld<NT>((const u3*)(q + (size_t)min(base + (H + u) * THREADS, count - 1) * 3));
#pragma unroll
				for (uint32_t u = 0; u < H; ++u) acc ^= b[u][0] ^ b[u][1] ^ b[u][2];
			}
		} else {
			const u2* pay = (const u2*)q;                                   // [CAP] 8-byte payloads
			const unsigned short* hdr = (const unsigned short*)(pay + CAP);  // [CAP] 2-byte headers
			for (uint32_t base = threadIdx.x; base < count; base += THREADS * U) {
				u2 p[U];
				unsigned short h[U];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					const uint32_t t = min(base + u * THREADS, count - 1);
					p[u] = ld<NT>(pay + t);
					h[u] = ld<NT>(hdr + t);
				}
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) acc ^= p[u][0] ^ p[u][1] ^ h[u];
			}
		}
		if (EPILOGUE) {  // conversion + store of the 32 KiB slice
			__syncthreads();
			for (uint32_t e = threadIdx.x; e < 2048; e += THREADS) {
				u4 v = ((const u4*)lds_raw)[e];
				v[0] ^= acc;
				((u4*)slices)[(size_t)queue * 2048 + e] = v;
			}
		}
	}
	out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

// LDS-DMA ring: every wave streams its share of the queue (1 KiB pieces, lane-linear) through DEPTH private 1 KiB slots next to the
// 64 KiB table, keeps DEPTH pieces in flight, and reads a landed piece back as 12-byte records (here: 16-byte words, same bytes).
template <uint32_t THREADS, uint32_t DEPTH, bool EPILOGUE>
__global__ void __launch_bounds__(THREADS) k_read_dma(const uint32_t* __restrict__ queues, const uint32_t* __restrict__ counts, uint32_t* __restrict__ out,
                                                       uint32_t* __restrict__ slices, uint32_t n_queues, uint32_t table_bytes) {
	constexpr uint32_t N_WAVES = THREADS / 64;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63u;
	unsigned char* ring = lds_raw + table_bytes + wave * DEPTH * 1024u;
	uint32_t acc = 0;
	for (uint32_t queue = blockIdx.x; queue < n_queues; queue += gridDim.x) {
		const uint32_t count = counts[queue];
		const unsigned char* q = (const unsigned char*)(queues + (size_t)queue * CAP * 3);
		const uint32_t n_pieces = (count * 12u + 1023u) / 1024u;  // the whole queue in 1 KiB pieces; wave w takes pieces w, w + N_WAVES, ...
		if (EPILOGUE) {
			for (uint32_t e = threadIdx.x; e < 4096; e += THREADS) ((u4*)lds_raw)[e] = u4{0, 0, 0, 0};
			__syncthreads();
		}
		const uint32_t mine = n_pieces > wave ? (n_pieces - wave + N_WAVES - 1) / N_WAVES : 0u;
		auto issue = [&](uint32_t k) {
			const unsigned char* src = q + (size_t)(wave + k * N_WAVES) * 1024u + lane * 16u;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
			                                 (__attribute__((address_space(3))) void*)(ring + (k % DEPTH) * 1024u), 16, 0, 0);
		};
#pragma unroll
		for (uint32_t k = 0; k < DEPTH; ++k) {
			if (k < mine) issue(k);
		}
		for (uint32_t k = 0; k < mine; ++k) {
			// the oldest piece has landed when at most (pieces issued after it) remain outstanding
			const uint32_t after = min(mine - 1 - k, DEPTH - 1);
			if (after >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
			else if (after == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
			else if (after == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
			else if (after == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
			else if (after == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
			else if (after == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
			else if (after == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const u4 v = *(const u4*)(ring + (k % DEPTH) * 1024u + lane * 16u);
			acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is read before it is refilled
			if (k + DEPTH < mine) issue(k + DEPTH);
		}
		if (EPILOGUE) {
			__syncthreads();
			for (uint32_t e = threadIdx.x; e < 2048; e += THREADS) {
				u4 v = ((const u4*)lds_raw)[e];
				v[0] ^= acc;
				((u4*)slices)[(size_t)queue * 2048 + e] = v;
			}
		}
	}
	out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

// ---- write side ------------------------------------------------------------------------------------------------
// A 256-thread workgroup appends, per tile, 2048 records as 64 runs of 32 to the 64 queues of its level (16 levels x 64 queues); run
// positions are a function of the tile (no atomics here).  MODE 0: 12-byte records, nt; 1: plain; 2: write-through (sc1); 3: 8 + 2 bytes nt
template <int MODE>
__global__ void __launch_bounds__(256) k_write(uint32_t* __restrict__ queues, uint32_t tiles_per_level, uint32_t wgs_per_level) {
	const uint32_t level = blockIdx.x / wgs_per_level, first = blockIdx.x % wgs_per_level;
	for (uint32_t tile = first; tile < tiles_per_level; tile += wgs_per_level) {
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			const uint32_t r = threadIdx.x + k * 256u, b = r >> 5, pos = tile * 32u + (r & 31u);
			const uint32_t queue = level * 64u + b;
			uint32_t* dst = queues + ((size_t)queue * CAP + pos) * 3;
			const u3 rec = u3{r, tile, k};
			if constexpr (MODE == 0) {
				__builtin_nontemporal_store(rec, (u3*)dst);
			} else if constexpr (MODE == 1) {
				*(u3*)dst = rec;
			} else if constexpr (MODE == 2) {
				asm volatile("global_store_dwordx3 %0, %1, off sc1" ::"v"(dst), "v"(rec) : "memory");
			} else {
				u2* pay = (u2*)(queues + (size_t)queue * CAP * 3);
				unsigned short* hdr = (unsigned short*)(pay + CAP);
				__builtin_nontemporal_store(u2{r, tile}, pay + pos);
				__builtin_nontemporal_store((unsigned short)k, hdr + pos);
			}
		}
	}
}

template <typename F>
static float time_it(const char* name, double bytes, F&& launch, int reps = 20) {
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	for (int i = 0; i < 3; ++i) launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; ++i) launch();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	CHECK(hipGetLastError());
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	const double us = ms * 1e3 / reps;
	printf("%-64s %8.1f us  %7.2f TB/s\n", name, us, bytes / us / 1e6);
	fflush(stdout);
	return (float)us;
}

int main() {
	uint32_t *qbuf[3], *counts, *out, *slices;
	const size_t queue_words = (size_t)N_QUEUES * CAP * 3;
	for (int i = 0; i < 3; ++i) {  // three sets of queues used in turn: a pass never finds its 201 MB in the 256 MiB Infinity Cache
		CHECK(hipMalloc(&qbuf[i], queue_words * 4));
		CHECK(hipMemset(qbuf[i], 1, queue_words * 4));
	}
	int turn = 0;
#define queues (qbuf[turn = (turn + 1) % 3])
	CHECK(hipMalloc(&counts, N_QUEUES * 4));
	CHECK(hipMalloc(&out, 2048 * 1024 * 4));
	CHECK(hipMalloc(&slices, (size_t)N_QUEUES * 32768));
	std::vector<uint32_t> c(N_QUEUES, COUNT);
	CHECK(hipMemcpy(counts, c.data(), N_QUEUES * 4, hipMemcpyHostToDevice));
	const double rd = (double)N_QUEUES * COUNT * 12.0, rd10 = (double)N_QUEUES * COUNT * 10.0, st = (double)N_QUEUES * 32768.0;
	constexpr uint32_t LDS64 = 64 * 1024;
#define READ(NAME, KERNEL, GRID, THREADS_, LDS, BYTES)                                                                   \
	CHECK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));                    \
	time_it(NAME, BYTES, [&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS_), LDS, 0, queues, counts, out, slices, N_QUEUES); });
	printf("== read side: 1024 queues x 16384 records (201 MB as 12-byte records) ==\n");
	READ("12 B nt, U=8, 512 thr, 1 WG per queue (product form), loads only", (k_read<512, 0, true, 8, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B plain, U=8, 512 thr, 1 WG per queue", (k_read<512, 0, false, 8, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B nt, U=8, + clear and 32 KiB slice store", (k_read<512, 0, true, 8, true>), N_QUEUES, 512, LDS64, rd + st)
	READ("12 B nt, U=4", (k_read<512, 0, true, 4, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B nt, U=16", (k_read<512, 0, true, 16, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B nt, U=8, 1024 thr (1 WG per CU at 128 KiB)", (k_read<1024, 0, true, 8, false>), N_QUEUES, 1024, 2 * LDS64, rd)
	READ("12 B nt, U=8, 256 thr, 32 KiB (4 WG per CU)", (k_read<256, 0, true, 8, false>), N_QUEUES, 256, LDS64 / 2, rd)
	READ("16 B nt, U=6", (k_read<512, 1, true, 6, false>), N_QUEUES, 512, LDS64, rd)
	READ("16 B plain, U=6", (k_read<512, 1, false, 6, false>), N_QUEUES, 512, LDS64, rd)
	READ("16 B nt, U=8", (k_read<512, 1, true, 8, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B nt rolling 4+4", (k_read<512, 2, true, 8, false>), N_QUEUES, 512, LDS64, rd)
	READ("12 B nt rolling 8+8", (k_read<512, 2, true, 16, false>), N_QUEUES, 512, LDS64, rd)
	READ("8+2 B nt (10-byte records), U=8", (k_read<512, 3, true, 8, false>), N_QUEUES, 512, LDS64, rd10)
	READ("8+2 B nt, U=8, + clear and slice store", (k_read<512, 3, true, 8, true>), N_QUEUES, 512, LDS64, rd10 + st)
	READ("12 B nt, U=8, persistent 512 WGs", (k_read<512, 0, true, 8, false>), 512, 512, LDS64, rd)
	READ("12 B nt, U=8, persistent 512 WGs + clear and slice store", (k_read<512, 0, true, 8, true>), 512, 512, LDS64, rd + st)
	READ("16 B nt, U=6, persistent 512 WGs", (k_read<512, 1, true, 6, false>), 512, 512, LDS64, rd)
#define READ_DMA(NAME, THREADS_, DEPTH, EPI, GRID, TABLE)                                                                                      \
	{                                                                                                                                          \
		const uint32_t lds = TABLE + (THREADS_ / 64) * DEPTH * 1024;                                                                           \
		CHECK(hipFuncSetAttribute((const void*)k_read_dma<THREADS_, DEPTH, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));             \
		time_it(NAME, EPI ? rd + st : rd, [&] {                                                                                                \
			hipLaunchKernelGGL((k_read_dma<THREADS_, DEPTH, EPI>), dim3(GRID), dim3(THREADS_), lds, 0, queues, counts, out, slices, N_QUEUES, (uint32_t)TABLE); \
		});                                                                                                                                    \
	}
	READ_DMA("LDS-DMA 16 B, depth 2, 512 thr, 64 KiB table (2 WG per CU)", 512, 2, false, N_QUEUES, LDS64)
	READ_DMA("LDS-DMA depth 4, 512 thr, 48 KiB table (2 WG per CU)", 512, 4, false, N_QUEUES, 48 * 1024)
	READ_DMA("LDS-DMA depth 8, 512 thr, 64 KiB table (1 WG per CU)", 512, 8, false, N_QUEUES, LDS64)
	READ_DMA("LDS-DMA depth 6, 1024 thr, 64 KiB table (1 WG per CU)", 1024, 6, false, N_QUEUES, LDS64)
	READ_DMA("LDS-DMA depth 4, 1024 thr, 64 KiB table (1 WG per CU)", 1024, 4, false, N_QUEUES, LDS64)
	READ_DMA("LDS-DMA depth 2, 512 thr, persistent 512 WGs", 512, 2, false, 512, LDS64)
	READ_DMA("LDS-DMA depth 2, 512 thr, + clear and slice store", 512, 2, true, N_QUEUES, LDS64)
	READ_DMA("LDS-DMA depth 6, 1024 thr, persistent 256 WGs + clear and slice store", 1024, 6, true, 256, LDS64)

	printf("== write side: 16 levels x 512 tiles x 2048 records in runs of 32 (201 MB as 12-byte records) ==\n");
	const uint32_t tiles = 512, wgs = 128;
#define WRITE(NAME, MODE, BYTES) time_it(NAME, BYTES, [&] { hipLaunchKernelGGL((k_write<MODE>), dim3(16 * wgs), dim3(256), 0, 0, queues, tiles, wgs); });
	WRITE("12 B nt stores", 0, rd)
	WRITE("12 B plain stores", 1, rd)
	WRITE("12 B write-through (sc1) stores", 2, rd)
	WRITE("8+2 B nt stores", 3, rd10)

	// Does the read side pay for the write side's deferred write-backs?  Alternating launches on one buffer (the scatter -> owner sequence of
	// the product; the write kernel's stores are line-aligned runs here), the read kernel timed on its own with events around each launch.
	printf("== read right after the write kernel filled the same queues (the product's sequence) ==\n");
	{
		hipEvent_t a, b;
		CHECK(hipEventCreate(&a));
		CHECK(hipEventCreate(&b));
		CHECK(hipFuncSetAttribute((const void*)k_read<512, 0, true, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64));
		for (int mode = 0; mode < 3; ++mode) {
			double read_ms = 0, write_ms = 0;
			const int reps = 20;
			for (int i = 0; i < reps + 3; ++i) {
				uint32_t* q = queues;
				float ms = 0;
				CHECK(hipEventRecord(a));
				if (mode == 0) hipLaunchKernelGGL((k_write<1>), dim3(16 * wgs), dim3(256), 0, 0, q, tiles, wgs);
				else if (mode == 1) hipLaunchKernelGGL((k_write<0>), dim3(16 * wgs), dim3(256), 0, 0, q, tiles, wgs);
				else hipLaunchKernelGGL((k_write<2>), dim3(16 * wgs), dim3(256), 0, 0, q, tiles, wgs);
				CHECK(hipEventRecord(b));
				CHECK(hipEventSynchronize(b));
				CHECK(hipEventElapsedTime(&ms, a, b));
				if (i >= 3) write_ms += ms;
				CHECK(hipEventRecord(a));
				hipLaunchKernelGGL((k_read<512, 0, true, 8, false>), dim3(N_QUEUES), dim3(512), LDS64, 0, q, counts, out, slices, N_QUEUES);
				CHECK(hipEventRecord(b));
				CHECK(hipEventSynchronize(b));
				CHECK(hipEventElapsedTime(&ms, a, b));
				if (i >= 3) read_ms += ms;
			}
			printf("%-40s write %7.1f us   read of what was just written %7.1f us\n", mode == 0 ? "plain stores" : (mode == 1 ? "nt stores" : "write-through (sc1) stores"), write_ms * 1e3 / reps,
			       read_ms * 1e3 / reps);
		}
		// the same back to back without a host synchronisation in between (one pair of events around both kernels)
		for (int mode = 0; mode < 2; ++mode) {
			float ms = 0;
			const int reps = 20;
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(a));
			for (int i = 0; i < reps; ++i) {
				uint32_t* q = queues;
				if (mode == 0) hipLaunchKernelGGL((k_write<1>), dim3(16 * wgs), dim3(256), 0, 0, q, tiles, wgs);
				else hipLaunchKernelGGL((k_write<2>), dim3(16 * wgs), dim3(256), 0, 0, q, tiles, wgs);
				hipLaunchKernelGGL((k_read<512, 0, true, 8, false>), dim3(N_QUEUES), dim3(512), LDS64, 0, q, counts, out, slices, N_QUEUES);
			}
			CHECK(hipEventRecord(b));
			CHECK(hipEventSynchronize(b));
			CHECK(hipEventElapsedTime(&ms, a, b));
			printf("%-40s write + read back to back %7.1f us per pair\n", mode == 0 ? "plain stores" : "write-through (sc1) stores", ms * 1e3 / reps);
		}
	}
	return 0;
}
