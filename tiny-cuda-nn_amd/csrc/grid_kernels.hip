// grid_kernels.hip -- see grid_kernels.h for the design notes and the reference lines restated.
// Build with -ffp-contract=off: the fp32 position/weight arithmetic must round exactly as written so
// that indices AND interpolated features are bit-identical to the CPU oracle.
#include "grid_kernels.h"
#include "elementwise_kernels.h"  // Pcg32 (stochastic interpolation)
#include "exp_diag.h"             // experiment switches: compile-time zeros in the product build

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace tcnn_hip {

constexpr uint32_t GRID_THREADS = 256;
constexpr uint32_t GRID_SPT = 4;  // samples per thread (independent gathers in flight per lane)
constexpr uint32_t GRID_TILE = GRID_THREADS * GRID_SPT;

// Work distribution: block b -> (level, tile) with level % 8 == b % 8.  Blocks are dispatched
// round-robin over the 8 XCDs (observed, not guaranteed): every XCD then gathers from ceil(L/8)
// level tables only, in level order, so one table at a time is hot in its private L2
// (measured: 237 G gathers/s with XCD-local tables vs 67 G/s mixed, profiles/r01_microbench_atomics.txt).
TCNN_DEVICE bool grid_work_item(uint32_t n_levels, uint32_t tiles, uint32_t& level, uint32_t& tile) {
	const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
	if (xcd >= n_levels) return false;
	const uint32_t levels_here = (n_levels - xcd + 7u) / 8u;
	if (slot >= levels_here * tiles) return false;
	level = xcd + 8u * (slot / tiles);
	tile = slot % tiles;
	return true;
}

static uint32_t grid_n_blocks(uint32_t n_levels, uint32_t n) {
	return 8u * div_round_up(n_levels, 8u) * div_round_up(n, GRID_TILE);
}

// ---------------------------------------------------------------------------------------------
// per-level constants and per-sample cell data shared by every kernel below
// ---------------------------------------------------------------------------------------------
template <uint32_t D>
struct Level {
	uint32_t hashmap_size, resolution, mask;
	float scale;
	bool is_hash, smooth, nearest;
	bool fast;  // hashed level with a power-of-two table: index = coherent_prime_hash & mask
};

template <uint32_t D>
TCNN_DEVICE Level<D> make_level(const GridMeta& meta, uint32_t level) {
	constexpr uint32_t MAX_BASES[11] = {0x0, 0xFFFFFFFF, 0xFFFF, 0x659, 0xFF, 0x54, 0x28, 0x17, 0xF, 0xB, 0x9};
	Level<D> lv;
	lv.hashmap_size = meta.offset[level + 1] - meta.offset[level];
	lv.resolution = meta.resolution[level];
	lv.mask = lv.hashmap_size - 1u;
	lv.scale = meta.scale[level];
	lv.is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	lv.smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	lv.nearest = meta.interp == (uint32_t)InterpolationType::Nearest;
	// same decision as grid_index (common_device.h:868-881): hashed iff hashmap_size < resolution^D
	uint32_t stride = 0xFFFFFFFFu;
	if (lv.resolution <= MAX_BASES[D]) {
		stride = 1;
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) stride *= lv.resolution;
	}
	lv.fast = lv.is_hash && lv.hashmap_size < stride && (lv.hashmap_size & lv.mask) == 0u;
	return lv;
}

// make_level's `fast` on the host (the owner kernel's item descriptors carry it)
static inline bool level_is_fast(const GridMeta& meta, uint32_t level) {
	constexpr uint32_t MAX_BASES[11] = {0x0, 0xFFFFFFFF, 0xFFFF, 0x659, 0xFF, 0x54, 0x28, 0x17, 0xF, 0xB, 0x9};
	const uint32_t hashmap_size = meta.offset[level + 1] - meta.offset[level], resolution = meta.resolution[level];
	uint32_t stride = 0xFFFFFFFFu;
	if (meta.n_dims < 11 && resolution <= MAX_BASES[meta.n_dims]) {
		stride = 1;
		for (uint32_t d = 0; d < meta.n_dims; ++d) stride *= resolution;
	}
	return meta.grid_type == (uint32_t)GridType::Hash && hashmap_size < stride && (hashmap_size & (hashmap_size - 1u)) == 0u;
}

TCNN_DEVICE float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
TCNN_DEVICE float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

template <uint32_t D>
struct Cell {
	uint32_t grid[D];         // integer cell coordinate (may wrap, common_device.h:1002-1007)
	uint32_t hlo[D], hhi[D];  // grid[d] * prime[d] and (grid[d] + 1) * prime[d]  (mod 2^32)
	float w[D][2];            // [d][0] = 1 - frac, [d][1] = frac  (after the interpolation function)
	float derivative[D];
};

// reference common_device.h:1016-1043 (pos_fract) for every dimension of one sample (x = its position)
template <uint32_t D, bool FAST>
TCNN_DEVICE Cell<D> make_cell(const Level<D>& lv, const float (&x)[D]) {
	constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
	Cell<D> c;
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) {
		float p = __builtin_fmaf(lv.scale, x[d], 0.5f);
		const float tmp = __builtin_floorf(p);
		c.grid[d] = (uint32_t)(int)tmp;
		p -= tmp;
		c.derivative[d] = lv.smooth ? smoothstep_derivative(p) : 1.0f;
		if (lv.smooth) p = smoothstep(p);
		c.w[d][0] = 1 - p;
		c.w[d][1] = p;
		if constexpr (FAST) {
			c.hlo[d] = c.grid[d] * primes[d];
			c.hhi[d] = c.hlo[d] + primes[d];
		}
	}
	return c;
}

template <uint32_t D, bool TRY_PACKED = false>
TCNN_DEVICE void load_position(const GridIO& io, uint32_t i, float (&x)[D]) {
#if !defined(TCNN_HOST_EMU)
	// (forward kernels only: in the record scatter the same load measured 8 us SLOWER than three strided dword loads)
	if (TRY_PACKED && io.pos_stride_d == 1u && io.pos_stride_i == D) {
		// sample-major contiguous positions (what every caller of the hot path passes): ONE D-dword load per lane instead of D
		// strided ones (a 12-byte lane stride costs an instruction ~20 clk whatever its width; wave-uniform branch).  A buffer
		// load, because the 4-byte-aligned 12-byte access is split into two by the compiler in its global form.
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)io.positions, 0, (int)(io.n * D * 4u), 0x00020000);
		// (the result is cast as a whole: indexing the builtin's vector_size type directly is miscompiled by ROCm 7.2's clang into
		// one dword splat over all elements; the 16-byte form is narrowed to the 12 bytes that are used)
		typedef float f2 __attribute__((ext_vector_type(2)));
		if constexpr (D == 2) {
			const f2 p = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(i * D * 4u), 0, 0));
			x[0] = p[0];
			x[1] = p[1];
		} else {
			const f4 p = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(i * D * 4u), 0, 0));
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) x[d] = p[d];
		}
		return;
	}
#endif
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) x[d] = io.positions[(size_t)i * io.pos_stride_i + (size_t)d * io.pos_stride_d];
}

template <uint32_t D, bool FAST>
TCNN_DEVICE Cell<D> make_cell(const Level<D>& lv, const GridIO& io, uint32_t i) {
	float x[D];
	load_position<D>(io, i, x);
	return make_cell<D, FAST>(lv, x);
}

// entry index of corner `idx` (bit d of idx selects +1 in dimension d, grid.h:147-160)
template <uint32_t D, bool FAST>
TCNN_DEVICE uint32_t corner_index(const Level<D>& lv, const Cell<D>& c, uint32_t idx) {
	if constexpr (FAST) {
		uint32_t h = 0;
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) h ^= ((idx >> d) & 1u) ? c.hhi[d] : c.hlo[d];
		return h & lv.mask;
	} else {
		uint32_t local[D];
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) local[d] = c.grid[d] + ((idx >> d) & 1u);
		return grid_index<D>(lv.is_hash, lv.hashmap_size, lv.resolution, local);
	}
}

// interpolation weight of corner `idx`: ((1 * w0) * w1) * w2 ... in the reference's order (grid.h:148-160)
template <uint32_t D>
TCNN_DEVICE float corner_weight(const Cell<D>& c, uint32_t idx) {
	float weight = ((idx & 1u) ? c.w[0][1] : c.w[0][0]);
#pragma unroll
	for (uint32_t d = 1; d < D; ++d) weight *= ((idx >> d) & 1u) ? c.w[d][1] : c.w[d][0];
	return weight;
}

// Corner weight of the SECOND-ORDER scatter (kernel_grid_backward_input_backward_grid, grid.h:427-455, summed over the
// gradient dimensions): scale * sum_d ddx[d] * pos'(d) * (+1 right / -1 left along d) * prod_{e != d} w_e(corner).
template <uint32_t D>
TCNN_DEVICE float corner_weight_second_order(const Level<D>& lv, const Cell<D>& c, uint32_t idx, const float (&ddx)[D]) {
	float total = 0.0f;
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) {
		float weight = lv.scale * ddx[d] * c.derivative[d];
#pragma unroll
		for (uint32_t e = 0; e < D; ++e) {
			if (e != d) weight *= ((idx >> e) & 1u) ? c.w[e][1] : c.w[e][0];
		}
		total += ((idx >> d) & 1u) ? weight : -weight;
	}
	return total;
}
template <uint32_t D>
TCNN_DEVICE void load_ddx(const GridIO& io, uint32_t i, float (&v)[D]) {
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) v[d] = io.ddx[(size_t)i * io.ddx_stride_i + (size_t)d * io.ddx_stride_d];
}

// F halves at `p` -> NP packed pairs (F == 1: {x, 0})
template <uint32_t F>
TCNN_DEVICE void load_features(const half_t* p, h2 (&v)[(F + 1) / 2]) {
	if constexpr (F == 1) {
		v[0] = h2{p[0], (half_t)0.0f};
	} else if constexpr (F == 2) {
		v[0] = *(const h2*)p;
	} else if constexpr (F == 4) {
		const h4 t = *(const h4*)p;
		v[0] = h2{t[0], t[1]};
		v[1] = h2{t[2], t[3]};
	} else {
		static_assert(F == 8, "n_features_per_level must be 1, 2, 4 or 8 (grid.h:1811-1821)");
		const h8 t = *(const h8*)p;
		v[0] = h2{t[0], t[1]};
		v[1] = h2{t[2], t[3]};
		v[2] = h2{t[4], t[5]};
		v[3] = h2{t[6], t[7]};
	}
}

// =============================================================================================
// forward (grid.h:49-212)
// =============================================================================================
template <uint32_t D, uint32_t F, bool DYDX, bool FAST>
TCNN_DEVICE void grid_forward_sample(const Level<D>& lv, const GridIO& io, const half_t* __restrict__ grid, uint32_t level, uint32_t i,
                                     bool level_off, half_t* __restrict__ out, float* __restrict__ dy_dx) {
	constexpr uint32_t NP = (F + 1) / 2;
	h2 result[NP];
#pragma unroll
	for (uint32_t p = 0; p < NP; ++p) result[p] = h2{(half_t)0.0f, (half_t)0.0f};
	float grads[DYDX ? F : 1][D];
#pragma unroll
	for (uint32_t f = 0; f < (DYDX ? F : 1); ++f)
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) grads[f][d] = 0.0f;

	if (!level_off) {
		const Cell<D> c = make_cell<D, FAST>(lv, io, i);
		if (lv.nearest) {
			load_features<F>(grid + (size_t)corner_index<D, FAST>(lv, c, 0) * F, result);
		} else {
			// gather all corners first (independent loads in flight) ...
			h2 val[1u << D][NP];
#pragma unroll
			for (uint32_t idx = 0; idx < (1u << D); ++idx) load_features<F>(grid + (size_t)corner_index<D, FAST>(lv, c, idx) * F, val[idx]);
			// ... then the N-linear interpolation, corner order and fp16 fma chain of grid.h:144-163
#pragma unroll
			for (uint32_t idx = 0; idx < (1u << D); ++idx) {
				const half_t wh = to_half_rn(corner_weight<D>(c, idx));
				const h2 w2 = h2{wh, wh};
#pragma unroll
				for (uint32_t p = 0; p < NP; ++p) result[p] = fma_h2(w2, val[idx][p], result[p]);
			}
			if constexpr (DYDX) {  // grid.h:172-211
#pragma unroll
				for (uint32_t gd = 0; gd < D; ++gd) {
#pragma unroll
					for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
						float weight = lv.scale;
						uint32_t corner = 0;  // corner with the gradient dimension at its low side
#pragma unroll
						for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
							const uint32_t dim = ngd >= gd ? (ngd + 1) : ngd;
							const uint32_t bit = (idx >> ngd) & 1u;
							weight *= bit ? c.w[dim][1] : c.w[dim][0];
							corner |= bit << dim;
						}
						const h2(&vl)[NP] = val[corner];
						const h2(&vr)[NP] = val[corner | (1u << gd)];
#pragma unroll
						for (uint32_t f = 0; f < F; ++f) {
							const float diff = (float)vr[f / 2][f % 2] - (float)vl[f / 2][f % 2];
							float t = weight * diff;
							t = t * c.derivative[gd];
							grads[f][gd] = grads[f][gd] + t;
						}
					}
				}
			}
		}
	}
#pragma unroll
	for (uint32_t f = 0; f < F; ++f) {
		const uint32_t k = level * F + f;
		if (out) out[(size_t)k * io.stride_k + (size_t)i * io.stride_i] = result[f / 2][f % 2];
		if constexpr (DYDX) {
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)k * io.n + i) * D + d] = grads[f][d];
		}
	}
}

template <uint32_t D, uint32_t F, bool DYDX>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_forward(const GridMeta meta, const GridIO io, const half_t* __restrict__ params,
                                                                half_t* __restrict__ out, float* __restrict__ dy_dx) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;
	const Level<D> lv = make_level<D>(meta, level);
	const half_t* __restrict__ grid = params + (size_t)meta.offset[level] * F;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;  // grid.h:72
	const bool level_off = (float)level >= max_level + 1e-3f;               // grid.h:75
	if (lv.fast) {  // wave-uniform: one lean code path per level kind
#pragma unroll
		for (uint32_t s = 0; s < GRID_SPT; ++s) {
			const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
			if (i < io.n) grid_forward_sample<D, F, DYDX, true>(lv, io, grid, level, i, level_off, out, dy_dx);
		}
	} else {
		for (uint32_t s = 0; s < GRID_SPT; ++s) {
			const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
			if (i < io.n) grid_forward_sample<D, F, DYDX, false>(lv, io, grid, level, i, level_off, out, dy_dx);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// forward, the form training and inference run (no dy_dx).
//
// What bounds it (scripts/microbench_l1.hip, profiles/r02_microbench_l1.txt): a gather instruction whose 64 lanes miss
// the CU's L1 costs ~150 clk per CU however wide the access is and whatever cache-policy bits it carries -- the XCD's
// L2 hands out one 128-byte line per channel and clock (~263 G lines/s chip-wide), and each x-neighbour corner pair of
// a sample is one line with 8 useful bytes in it.  The same instruction costs ~37 clk when it hits L1, ~20 clk for the
// 12-byte-strided position loads, ~16 clk from LDS.  So the levers are (1) as few instructions per (sample, level) as
// possible besides the 2^(D-1) line fetches, (2) an even load per XCD:
//   * a thread owns SPT samples of ONE level and issues all their 2^D * SPT gathers before the first use (walking
//     several levels per block with the positions loaded once measured slower: two tables then compete for the L2);
//   * the (level, tile) items are laid end to end, each with a cost weight (tables that fit the CU's L1 are cheap),
//     and cut into 8 runs of equal cost, one per XCD (block b runs on XCD b % 8 -- observed, only speed depends on
//     it): every XCD sees its 1-3 tables, 2 MiB each at the headline size, hot in its private 4 MiB L2.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t FWD_MAX_SEGMENTS = 20;  // per XCD: ceil(MAX_N_LEVELS / 8) + the two cut levels at the ends of a run
struct ForwardPlan {
	uint32_t tiles;  // sample tiles per level
	uint32_t n_segments[8];
	// a run of one level's tiles with what the workgroup needs of that level (make_level's inputs): the workgroup's whole "what am I?" is ONE
	// round of scalar loads -- the XCD's first four segments at once, searched in registers -- instead of a loop of dependent loads over the
	// segments followed by a round for the level's table geometry (a workgroup lives a few microseconds: every round trip ahead of its first
	// gather is occupancy the L2's line rate does not get)
	struct Segment {
		uint32_t level, tile_begin, tile_end, hashmap_size, resolution, scale_bits, offset, fast;
	} segments[8][FWD_MAX_SEGMENTS];
};
constexpr uint32_t FWD_SEGMENTS_AT_ONCE = 4;
template <uint32_t D>
TCNN_DEVICE Level<D> level_of_segment(const GridMeta& meta, const ForwardPlan::Segment& seg) {
	Level<D> lv;
	lv.hashmap_size = seg.hashmap_size;
	lv.resolution = seg.resolution;
	lv.mask = seg.hashmap_size - 1u;
	lv.scale = __builtin_bit_cast(float, seg.scale_bits);
	lv.is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	lv.smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	lv.nearest = meta.interp == (uint32_t)InterpolationType::Nearest;
	lv.fast = (seg.fast & 1u) != 0u;  // (an experiment build keeps a region of the table in the upper bits: exp_diag.h, EXP_FWD_REGION_LOG2)
	return lv;
}

template <uint32_t D, uint32_t F, uint32_t SPT, bool FAST>
TCNN_DEVICE void grid_forward_tile(const Level<D>& lv, const GridIO& io, const half_t* __restrict__ grid, uint32_t level, uint32_t first,
                                   const float (&x)[SPT][D], half_t* __restrict__ out, uint32_t region_begin = 0u, uint32_t region_end = 0xffffffffu) {
	constexpr uint32_t NP = (F + 1) / 2, NC = 1u << D;
	Cell<D> c[SPT];
	h2 val[SPT][NC][NP];
#pragma unroll
	for (uint32_t s = 0; s < SPT; ++s) {
		c[s] = make_cell<D, FAST>(lv, x[s]);
#pragma unroll
		for (uint32_t idx = 0; idx < NC; ++idx) {
			const uint32_t index = corner_index<D, FAST>(lv, c[s], idx);
			if constexpr (EXP_FWD_REGION_LOG2 != 0u) {  // region-pass timing build: only the corners inside this pass's slice of the table are fetched
#pragma unroll
				for (uint32_t p = 0; p < NP; ++p) val[s][idx][p] = h2{(half_t)0.0f, (half_t)0.0f};
				if (index >= region_begin && index < region_end) load_features<F>(grid + (size_t)index * F, val[s][idx]);
			} else {
				load_features<F>(grid + (size_t)index * F, val[s][idx]);
			}
		}
	}
#pragma unroll
	for (uint32_t s = 0; s < SPT; ++s) {
		h2 result[NP];
#pragma unroll
		for (uint32_t p = 0; p < NP; ++p) result[p] = h2{(half_t)0.0f, (half_t)0.0f};
#pragma unroll
		for (uint32_t idx = 0; idx < NC; ++idx) {  // corner order and fp16 fma chain of grid.h:144-163
			const half_t wh = to_half_rn(corner_weight<D>(c[s], idx));
			const h2 w2 = h2{wh, wh};
#pragma unroll
			for (uint32_t p = 0; p < NP; ++p) result[p] = fma_h2(w2, val[s][idx][p], result[p]);
		}
		const uint32_t i = first + s * GRID_THREADS + threadIdx.x;
		if (i < io.n) {
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) out[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i] = result[f / 2][f % 2];
		}
	}
}

template <uint32_t D, uint32_t F, uint32_t SPT>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_forward_tiles(const GridMeta meta, const GridIO io, const ForwardPlan plan,
                                                                      const half_t* __restrict__ params, half_t* __restrict__ out) {
	constexpr uint32_t TILE = GRID_THREADS * SPT;
	// block -> (segment of its XCD's run, tile): level-major, so an XCD walks one table at a time
	const uint32_t xcd = blockIdx.x & 7u;
	uint32_t slot = blockIdx.x >> 3, tile = 0;
	bool found = false;
	ForwardPlan::Segment mine = {};
	{
		ForwardPlan::Segment head[FWD_SEGMENTS_AT_ONCE];  // (rows are zero-padded: an unused segment holds no tiles and never matches)
#pragma unroll
		for (uint32_t k = 0; k < FWD_SEGMENTS_AT_ONCE; ++k) head[k] = plan.segments[xcd][k];
#if !defined(TCNN_HOST_EMU)
		{  // (all of it now, in ONE round -- the four segments and what the position loads need of the other kernel arguments)
			uint64_t positions = (uint64_t)(uintptr_t)io.positions;
			uint32_t a = io.pos_stride_i, b = io.pos_stride_d, c = io.n, d = meta.grid_type, e = meta.interp;
			asm volatile("" : "+s"(positions), "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(head[0].level), "+s"(head[0].tile_begin), "+s"(head[0].tile_end),
			             "+s"(head[0].hashmap_size), "+s"(head[0].resolution), "+s"(head[0].scale_bits), "+s"(head[0].offset), "+s"(head[0].fast));
#pragma unroll
			for (uint32_t k = 1; k < FWD_SEGMENTS_AT_ONCE; ++k) {
				asm volatile("" : "+s"(head[k].level), "+s"(head[k].tile_begin), "+s"(head[k].tile_end), "+s"(head[k].hashmap_size), "+s"(head[k].resolution),
				             "+s"(head[k].scale_bits), "+s"(head[k].offset), "+s"(head[k].fast));
			}
		}
#endif
		// branch-free (selects): written with branches the compiler sinks each segment's loads into "the segments before it did not match"
		// and the one round of loads becomes up to four
#pragma unroll
		for (uint32_t k = 0; k < FWD_SEGMENTS_AT_ONCE; ++k) {
			const uint32_t n = head[k].tile_end - head[k].tile_begin;
			const bool here = !found && slot < n;
			mine.level = here ? head[k].level : mine.level;
			mine.hashmap_size = here ? head[k].hashmap_size : mine.hashmap_size;
			mine.resolution = here ? head[k].resolution : mine.resolution;
			mine.scale_bits = here ? head[k].scale_bits : mine.scale_bits;
			mine.offset = here ? head[k].offset : mine.offset;
			mine.fast = here ? head[k].fast : mine.fast;
			tile = here ? head[k].tile_begin + slot : tile;
			slot -= (found || here) ? 0u : n;
			found = found || here;
		}
	}
	if (!found) {  // (more than 32 levels: the rest of the run, one segment at a time)
		for (uint32_t k = FWD_SEGMENTS_AT_ONCE; k < plan.n_segments[xcd]; ++k) {
			const ForwardPlan::Segment seg = plan.segments[xcd][k];
			const uint32_t n = seg.tile_end - seg.tile_begin;
			if (slot < n) {
				mine = seg;
				tile = seg.tile_begin + slot;
				found = true;
				break;
			}
			slot -= n;
		}
	}
	const uint32_t level = mine.level;
	if (!found) return;
	const uint32_t first = tile * TILE;
	float x[SPT][D];
#pragma unroll
	for (uint32_t s = 0; s < SPT; ++s) load_position<D, true>(io, min(first + s * GRID_THREADS + threadIdx.x, io.n - 1u), x[s]);
	const Level<D> lv = level_of_segment<D>(meta, mine);
	const half_t* __restrict__ grid = params + (size_t)mine.offset * F;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;  // grid.h:72
	const bool level_off = (float)level >= max_level + 1e-3f;               // grid.h:75
	if (level_off || lv.nearest) {  // rare forms: one sample at a time
		for (uint32_t s = 0; s < SPT; ++s) {
			const uint32_t i = first + s * GRID_THREADS + threadIdx.x;
			if (i < io.n) grid_forward_sample<D, F, false, false>(lv, io, grid, level, i, level_off, out, nullptr);
		}
	} else if (lv.fast) {  // wave-uniform: one lean code path per level kind
		if constexpr (EXP_FWD_REGION_LOG2 != 0u) {
			const uint32_t n_regions = mine.fast >> 16, region = (mine.fast >> 8) & 0xffu;
			if (n_regions > 1u) grid_forward_tile<D, F, SPT, true>(lv, io, grid, level, first, x, out, region << EXP_FWD_REGION_LOG2, (region + 1u) << EXP_FWD_REGION_LOG2);
			else grid_forward_tile<D, F, SPT, true>(lv, io, grid, level, first, x, out);
		} else {
			grid_forward_tile<D, F, SPT, true>(lv, io, grid, level, first, x, out);
		}
	} else {
		grid_forward_tile<D, F, SPT, false>(lv, io, grid, level, first, x, out);
	}
}

// =============================================================================================
// backward, the reference's formulation (grid.h:215-320): one packed-half global atomic per corner.
// Kept for A/B measurements only (F >= 2): scattered global atomics top out at ~21 G updates/s on
// MI355X whatever their flavour (profiles/r01_microbench_atomics.txt) -- 1.6 ms for one headline step.
// =============================================================================================
template <uint32_t D, uint32_t F>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_backward_atomic(const GridMeta meta, const GridIO io, const half_t* __restrict__ dL_dy,
                                                                        half_t* __restrict__ grid_gradient) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	if ((float)level > max_level + 1e-3f) return;  // grid.h:242 (sic: '>' here, '>=' in forward)
	const Level<D> lv = make_level<D>(meta, level);
	half_t* __restrict__ grad = grid_gradient + (size_t)meta.offset[level] * F;
	const bool second_order = io.ddx != nullptr;  // kernel_grid_backward_input_backward_grid (grid.h:427-455): another corner weight

	for (uint32_t s = 0; s < GRID_SPT; ++s) {
		const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
		if (i >= io.n) continue;
		Cell<D> c = make_cell<D, false>(lv, io, i);
		half_t g[F];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) g[f] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
		float dd[D];
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) dd[d] = 0.0f;
		if (second_order) load_ddx<D>(io, i, dd);
		const bool one_corner = !second_order && (lv.nearest || meta.stochastic != 0u);
		if (!second_order && meta.stochastic != 0u && !lv.nearest) {  // grid.h:284-299, random_val(1337, i + level * n): common_device.h:469-473
			Pcg32 rng(1337u);
			rng.advance((int64_t)(uint32_t)(i + level * io.n));
			const float sample = rng.next_float();
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) {
				if (!(sample >= c.w[d][1])) c.grid[d] += 1u;
			}
		}
		const uint32_t n_corners = one_corner ? 1u : (1u << D);
		for (uint32_t idx = 0; idx < n_corners; ++idx) {
			const float weight = second_order ? corner_weight_second_order<D>(lv, c, idx, dd) : corner_weight<D>(c, idx);
			const uint32_t index = corner_index<D, false>(lv, c, idx);
			if constexpr (F == 1) {
				// fp32 product rounded once, as the bucketed form does for F == 1; a packed atomic on the aligned pair, the partner gets +0
				const half_t v = one_corner ? g[0] : to_half_rn(weight * (float)g[0]);
				atomic_add_h2(grad + (index & ~1u), (index & 1u) ? h2{(half_t)0.0f, v} : h2{v, (half_t)0.0f});
			} else {
				const half_t wh = one_corner ? (half_t)1.0f : to_half_rn(weight);
				const h2 w2 = h2{wh, wh};
#pragma unroll
				for (uint32_t p = 0; p < F / 2; ++p) atomic_add_h2(grad + (size_t)index * F + 2 * p, w2 * h2{g[2 * p], g[2 * p + 1]});  // (GRAD_T)weight * grad, grid.h:254
			}
		}
	}
}

// =============================================================================================
// fp32 encodings: GridEncodingTemplated<float> (what cpp_api.cu:165-168 instantiates for create_encoding(..., Precision::Fp32),
// tcnn.Encoding(dtype=torch.float32)).  Parameters, encoded features and gradients are fp32; the interpolation is the reference's
// kernel_grid<float> -- fp32 weights, result = fma(weight, value, result) in fp32 (grid.h:144-163) --, the backward pass its
// kernel_grid_backward<float, float>: one fp32 global atomic per corner and feature (grid.h:252-255; gradients of any magnitude survive,
// nothing is scaled).  Not a hot path of the step (the trainer's encoding is 16-bit): one thread per (sample, level), the reference's
// formulation; same index / weight code as the 16-bit kernels above.
// =============================================================================================
template <uint32_t D, uint32_t F, bool DYDX>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_forward_f32(const GridMeta meta, const GridIO io, const float* __restrict__ params, float* __restrict__ out,
                                                                    float* __restrict__ dy_dx) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;
	const Level<D> lv = make_level<D>(meta, level);
	const float* __restrict__ grid = params + (size_t)meta.offset[level] * F;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;  // grid.h:72
	const bool level_off = (float)level >= max_level + 1e-3f;               // grid.h:75
	for (uint32_t s = 0; s < GRID_SPT; ++s) {
		const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
		if (i >= io.n) continue;
		float result[F], grads[DYDX ? F : 1][D];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) result[f] = 0.0f;
#pragma unroll
		for (uint32_t f = 0; f < (DYDX ? F : 1); ++f)
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) grads[f][d] = 0.0f;
		if (!level_off) {
			const Cell<D> c = make_cell<D, false>(lv, io, i);
			if (lv.nearest) {
				const float* v = grid + (size_t)corner_index<D, false>(lv, c, 0) * F;
#pragma unroll
				for (uint32_t f = 0; f < F; ++f) result[f] = v[f];
			} else {
				float val[1u << D][F];
#pragma unroll
				for (uint32_t idx = 0; idx < (1u << D); ++idx) {
					const float* v = grid + (size_t)corner_index<D, false>(lv, c, idx) * F;
#pragma unroll
					for (uint32_t f = 0; f < F; ++f) val[idx][f] = v[f];
				}
#pragma unroll
				for (uint32_t idx = 0; idx < (1u << D); ++idx) {
					const float weight = corner_weight<D>(c, idx);
#pragma unroll
					for (uint32_t f = 0; f < F; ++f) result[f] = __builtin_fmaf(weight, val[idx][f], result[f]);
				}
				if constexpr (DYDX) {  // grid.h:172-211
#pragma unroll
					for (uint32_t gd = 0; gd < D; ++gd) {
#pragma unroll
						for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
							float weight = lv.scale;
							uint32_t corner = 0;
#pragma unroll
							for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
								const uint32_t dim = ngd >= gd ? (ngd + 1) : ngd;
								const uint32_t bit = (idx >> ngd) & 1u;
								weight *= bit ? c.w[dim][1] : c.w[dim][0];
								corner |= bit << dim;
							}
#pragma unroll
							for (uint32_t f = 0; f < F; ++f) {
								const float diff = val[corner | (1u << gd)][f] - val[corner][f];
								float t = weight * diff;
								t = t * c.derivative[gd];
								grads[f][gd] = grads[f][gd] + t;
							}
						}
					}
				}
			}
		}
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) {
			const uint32_t k = level * F + f;
			if (out) out[(size_t)k * io.stride_k + (size_t)i * io.stride_i] = result[f];
			if constexpr (DYDX) {
#pragma unroll
				for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)k * io.n + i) * D + d] = grads[f][d];
			}
		}
	}
}

template <uint32_t D, uint32_t F>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_backward_atomic_f32(const GridMeta meta, const GridIO io, const float* __restrict__ dL_dy,
                                                                            float* __restrict__ grid_gradient) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	if ((float)level > max_level + 1e-3f) return;  // grid.h:242
	const Level<D> lv = make_level<D>(meta, level);
	float* __restrict__ grad = grid_gradient + (size_t)meta.offset[level] * F;
	for (uint32_t s = 0; s < GRID_SPT; ++s) {
		const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
		if (i >= io.n) continue;
		Cell<D> c = make_cell<D, false>(lv, io, i);
		float g[F];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) g[f] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
		const bool one_corner = lv.nearest || meta.stochastic != 0u;
		if (meta.stochastic != 0u && !lv.nearest) {  // grid.h:284-299
			Pcg32 rng(1337u);
			rng.advance((int64_t)(uint32_t)(i + level * io.n));
			const float sample = rng.next_float();
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) {
				if (!(sample >= c.w[d][1])) c.grid[d] += 1u;
			}
		}
		const uint32_t n_corners = one_corner ? 1u : (1u << D);
		for (uint32_t idx = 0; idx < n_corners; ++idx) {
			const float weight = one_corner ? 1.0f : corner_weight<D>(c, idx);
			const uint32_t index = corner_index<D, false>(lv, c, idx);
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) atomic_add_f32(grad + (size_t)index * F + f, weight * g[f]);  // (T)weight * grad, T = float (grid.h:254)
		}
	}
}

// =============================================================================================
// backward, owner-computes form (the default).  No global atomics on the hot path: a workgroup OWNS a
// contiguous slice of one level's table, keeps it in LDS, walks the samples, recomputes the corner
// indices (integer ALU is cheap) and accumulates only the corners that fall into its slice; the slice
// is then written back with plain coalesced stores -- which also makes the reference's per-step
// gradient memset (grid.h:865-867) unnecessary.
//
// Measured LDS atomic rates that shape this (profiles/r01_microbench_lds_atomics.txt): a dense
// ds_add_f32 / ds_pk_add_f16 wave instruction costs ~170 clk (floating-point LDS atomics are serialised
// per lane, ~2.6 clk each), a dense ds_add_u32 / ds_add_u64 7 / 11 clk; with <= 2-3 active lanes all of
// them cost ~7 clk.  Hence two accumulator kinds, chosen per level on the host:
//   * small tables (coarse levels, nearly every corner of every sample hits the slice -> dense
//     atomics): 64-bit fixed point (2^-24 resolution, exact and order-independent, cannot overflow for
//     any fp16 input), the SAMPLES are additionally split over several workgroups, each flushing its
//     partial table with a few packed-half global atomics;
//   * large tables (fine / hashed levels, a slice sees ~1/16 of the corners -> sparse atomics):
//     packed fp16 (the reference's own accumulation type, vec.h:328-351) or fp32 slices.
// =============================================================================================
#ifndef TCNN_SLICED_THREADS
#define TCNN_SLICED_THREADS 1024
#endif
constexpr uint32_t SLICED_THREADS = TCNN_SLICED_THREADS;
constexpr uint32_t SLICED_LDS_BYTES = 128 * 1024;      // default slice size
constexpr uint32_t SLICED_LDS_MAX_BYTES = 160 * 1024;  // one CU's LDS
constexpr double FIXED_SCALE = 16777216.0;             // 2^24: below the smallest fp16 subnormal

enum SliceKind : uint32_t { SLICE_FIXED64 = 0, SLICE_FLOAT = 1, SLICE_GLOBAL_ATOMIC = 2, SLICE_BUCKET = 3 };

// Work list of one launch, in dispatch order (long slice passes first, short work fills the tail).
struct SlicePlan {
	uint32_t n_items;
	uint32_t blocks_per_item;                // launch stride (max workgroups of any item) of a near-uniform plan, else 0
	uint32_t block_begin[MAX_N_LEVELS + 1];  // first workgroup of item p
	uint32_t n_slices[MAX_N_LEVELS];         // slices (FIXED64 / FLOAT) or sample tiles (GLOBAL_ATOMIC) of item p
	uint8_t level[MAX_N_LEVELS];             // grid level of item p
	uint8_t kind[MAX_N_LEVELS];              // SliceKind of item p
	uint8_t slot[MAX_N_LEVELS];              // SLICE_BUCKET: slot of the level in the BucketPlan
};

enum class Acc { F32, PK16, FIX64 };

// round(v * 2^24) as a 64-bit integer using fp32 / int32 ops only (no fp64 conversions in the hot loop).
// v is a product of two halves: |v| <= 2^32 and at most 22 significant bits, so v * 2^8 splits exactly into an
// integer part (|hi| <= 2^40 would overflow -> clamp to the fp16 range first: |v| <= 65504 < 2^16 -> |hi| < 2^24)
// and a fraction |r| < 1 that is rounded to 16 bits.
TCNN_DEVICE long long to_fixed(float v) {
	v = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
	const float s = v * 256.0f;
	const float hi = __builtin_truncf(s);
	const int lo = (int)__builtin_rintf((s - hi) * 65536.0f);
	return (long long)(int)hi * 65536ll + (long long)lo;
}

template <uint32_t D, uint32_t F, Acc ACC, bool FAST>
TCNN_DEVICE void sliced_accumulate(const Level<D>& lv, const GridIO& io, const half_t* __restrict__ dL_dy, uint32_t level, uint32_t begin,
                                   uint32_t end, uint32_t slice_begin, uint32_t slice_count, unsigned char* lds_raw) {
	constexpr uint32_t N_CORNERS = 1u << D;
	float* tab_f = (float*)lds_raw;                            // [entries][F]
	h2* tab_h = (h2*)lds_raw;                                  // [entries][F/2]
	unsigned long long* tab_q = (unsigned long long*)lds_raw;  // [entries][F]
	// U samples per lane and iteration: all their position / gradient loads are issued before the first
	// use (each workgroup streams the whole batch; with one sample in flight the loop is latency-bound).
	constexpr uint32_t U = 4;
	for (uint32_t base = begin + threadIdx.x; base < end; base += SLICED_THREADS * U) {
		float x[U][D];
		half_t g[U][F];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) {
			const uint32_t i = min(base + u * SLICED_THREADS, end - 1);  // clamped: out-of-range lanes are masked below
			load_position<D>(io, i, x[u]);
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) g[u][f] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
		}
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) {
			const Cell<D> c = make_cell<D, FAST>(lv, x[u]);
			// which of this sample's corners live in my slice?  (branch-free bit mask)
			uint32_t match = 0;
#pragma unroll
			for (uint32_t idx = 0; idx < N_CORNERS; ++idx) {
				const uint32_t rel = corner_index<D, FAST>(lv, c, idx) - slice_begin;
				match |= (rel < slice_count ? 1u : 0u) << idx;
			}
			if (lv.nearest) match &= 1u;
			if (base + u * SLICED_THREADS >= end) match = 0;

			while (match) {
				const uint32_t idx = (uint32_t)__builtin_ctz(match);
				match &= match - 1;
				const uint32_t rel = corner_index<D, FAST>(lv, c, idx) - slice_begin;
				const float weight = lv.nearest ? 1.0f : corner_weight<D>(c, idx);
				const half_t wh = to_half_rn(weight);  // (GRAD_T)weight, grid.h:254
				if constexpr (ACC == Acc::PK16) {
					const h2 w2 = h2{wh, wh};
#pragma unroll
					for (uint32_t p = 0; p < F / 2; ++p) lds_atomic_add_h2(&tab_h[rel * (F / 2) + p], w2 * h2{g[u][2 * p], g[u][2 * p + 1]});
				} else {
					const float wq = F == 1 ? weight : (float)wh;  // F == 1: grad_t is float in the reference (grid.h:665)
#pragma unroll
					for (uint32_t f = 0; f < F; ++f) {
						const float prod = wq * (float)g[u][f];
						if constexpr (ACC == Acc::FIX64) {
							lds_atomic_add_u64(&tab_q[rel * F + f], (unsigned long long)to_fixed(prod));
						} else {
							lds_atomic_add_f32(&tab_f[rel * F + f], prod);
						}
					}
				}
			}
		}
	}
}

template <uint32_t D, uint32_t F, Acc ACC>
TCNN_DEVICE void sliced_level(const GridMeta& meta, const GridIO& io, const Level<D>& lv, uint32_t level, uint32_t slice, uint32_t chunk,
                              uint32_t n_chunks, uint32_t entries_per_slice, const half_t* __restrict__ dL_dy, half_t* __restrict__ grid_gradient,
                              bool accumulate, bool level_off, unsigned char* lds_raw) {
	const uint32_t slice_begin = slice * entries_per_slice;
	const uint32_t slice_count = slice_begin < lv.hashmap_size ? min(entries_per_slice, lv.hashmap_size - slice_begin) : 0u;
	half_t* __restrict__ grad = grid_gradient + ((size_t)meta.offset[level] + slice_begin) * F;

	constexpr uint32_t WORDS_PER_VALUE_X2 = ACC == Acc::FIX64 ? 4 : (ACC == Acc::F32 ? 2 : 1);  // 32-bit words per value, times two
	const uint32_t lds_words = slice_count * F * WORDS_PER_VALUE_X2 / 2;
	for (uint32_t e = threadIdx.x; e < lds_words; e += SLICED_THREADS) ((uint32_t*)lds_raw)[e] = 0u;  // +0.0f / (0, 0) / 0
	__syncthreads();

	if (!level_off) {
		const uint32_t per_chunk = div_round_up(io.n, n_chunks);
		const uint32_t begin = chunk * per_chunk;
		const uint32_t end = min(begin + per_chunk, io.n);
		if (lv.fast) {
			sliced_accumulate<D, F, ACC, true>(lv, io, dL_dy, level, begin, end, slice_begin, slice_count, lds_raw);
		} else {
			sliced_accumulate<D, F, ACC, false>(lv, io, dL_dy, level, begin, end, slice_begin, slice_count, lds_raw);
		}
	}
	__syncthreads();

	// ---- write the slice back: this workgroup is its only writer when n_chunks == 1
	const uint32_t n_halves = slice_count * F;  // even: level sizes are multiples of 8
	for (uint32_t e2 = threadIdx.x; e2 < n_halves / 2; e2 += SLICED_THREADS) {
		h2 v;
		if constexpr (ACC == Acc::PK16) {
			v = ((const h2*)lds_raw)[e2];
		} else if constexpr (ACC == Acc::F32) {
			v = h2{(half_t)((const float*)lds_raw)[2 * e2], (half_t)((const float*)lds_raw)[2 * e2 + 1]};
		} else {
			const long long q0 = ((const long long*)lds_raw)[2 * e2], q1 = ((const long long*)lds_raw)[2 * e2 + 1];
			v = h2{(half_t)(float)((double)q0 * (1.0 / FIXED_SCALE)), (half_t)(float)((double)q1 * (1.0 / FIXED_SCALE))};
		}
		if (n_chunks == 1) {
			if (accumulate) v += *(const h2*)(grad + 2 * e2);
			*(h2*)(grad + 2 * e2) = v;
		} else if (v[0] != (half_t)0.0f || v[1] != (half_t)0.0f) {
			atomic_add_h2(grad + 2 * e2, v);  // (table size) x (chunks) per level, small tables only
		}
	}
}

// =============================================================================================
// backward, bucket-once form for the large levels.  The slice passes above re-derive every corner of every
// sample once PER SLICE (16 x 13 passes over the batch at the headline config: VALU-bound).  Here each corner
// is derived ONCE:
//   pass A (k_grid_bucket_scatter): a workgroup takes one (level, sample tile), computes the corner records
//     {entry index, (GRAD_T)weight * grad} (grid.h:254), ranks them by table slice ("bucket") with integer LDS
//     atomics, reorders them by bucket in LDS and appends each bucket's run to that bucket's queue in HBM with
//     coalesced stores (one global integer atomic per (workgroup, bucket) reserves the run);
//   pass B (kind SLICE_BUCKET of k_grid_backward_sliced): the workgroup that owns a slice streams its queue and
//     accumulates in 64-bit fixed point in LDS (dense ds_add_u64: 11 clk per wave instruction vs ~170 for the
//     floating-point LDS atomics), then stores the slice -- exact, order-independent, no memset, no float atomics;
//   overflow: records that did not fit their queue (capacity = 2x the uniform expectation) or whose x-neighbour lives in
//     another bucket travel through one list; each owner picks its slice's records out of it before it stores (exact);
//     beyond OVERFLOW_INLINE_MAX records (strongly clustered inputs) the last owner to finish applies the list with the
//     reference's global atomics instead.
// HBM traffic: 2 x 8 B per corner (F = 2) -- 0.44 GB per headline step, a fraction of the chip's bandwidth.
// =============================================================================================
#ifndef TCNN_BUCKET_THREADS
#define TCNN_BUCKET_THREADS 256
#endif
constexpr uint32_t BUCKET_THREADS = TCNN_BUCKET_THREADS;
constexpr uint32_t MAX_BUCKET_LEVELS = 32;
#ifndef TCNN_BUCKET_RESIDENT_WGS
#define TCNN_BUCKET_RESIDENT_WGS 2048  // measured: 2 tiles in flight per resident slot beat 1, 1.5, 4 and 8 (profiles/r01_exp_scatter_wgs.txt)
#endif
constexpr uint32_t BUCKET_RESIDENT_WGS = TCNN_BUCKET_RESIDENT_WGS;  // persistent scatter workgroups over all levels
constexpr uint32_t MAX_BUCKETS_PER_LEVEL = 4096;
// overflow records up to which every bucket owner scans the list for its own (4 MiB of L2 reads per owner at the bound)
constexpr uint32_t OVERFLOW_INLINE_MAX = 1u << 18;
#ifndef TCNN_BUCKET_STAGE_BYTES
#define TCNN_BUCKET_STAGE_BYTES (32 * 1024)  // measured: 32 KiB (4 workgroups per CU) beats 64 and 16 KiB
#endif
constexpr uint32_t BUCKET_STAGE_BYTES = TCNN_BUCKET_STAGE_BYTES;  // LDS staging area of pass A

struct BucketPlan {
	uint32_t n_levels;  // bucketed levels
	uint32_t shift;     // log2(entries per bucket)
	uint32_t tiles;     // sample tiles per level in pass A
	uint32_t wgs_per_level;      // persistent pass-A workgroups per level (each walks tiles wg, wg + wgs_per_level, ...)
	uint32_t scatter_blocks;     // n_levels * wgs_per_level: pass-A blocks beyond these zero the gradients of chunked levels
	uint32_t overflow_counter;   // index of the overflow counter (== total number of queues); the one after it counts finished pass-C blocks
	TCNN_HOST_DEVICE uint32_t sum_slot(uint32_t j) const { return j; }
	TCNN_HOST_DEVICE uint32_t table_offset(const GridMeta& meta, uint32_t level) const { return meta.offset[level]; }
	uint32_t level_sum_base;     // (even) index of slot 0's 64-bit sums (LEVEL_SUM_PARTS per level) of |dL/dy| over the batch, 2^-32 units (OwnerScale; bfloat16 build only)
	uint32_t overflow_capacity;  // records
	uint32_t n_owner_blocks;     // workgroups of pass B that own a bucket (the last one to finish resets the bookkeeping counters)
	uint32_t packed_owner;       // pass B's bucket items run in k_grid_bucket_owner (packed accumulators), not in k_grid_backward_sliced
	uint8_t level[MAX_BUCKET_LEVELS];             // grid level of slot j
	uint32_t n_buckets[MAX_BUCKET_LEVELS];        // table slices
	uint32_t n_chunks[MAX_BUCKET_LEVELS];         // sample chunks: a queue belongs to one (chunk, bucket); > 1 only for small tables
	uint32_t tiles_per_chunk[MAX_BUCKET_LEVELS];
	uint32_t capacity[MAX_BUCKET_LEVELS];         // PAIRS of records per queue
	uint32_t counter_base[MAX_BUCKET_LEVELS];     // first counter of slot j; queue (chunk, bucket) uses counter chunk * n_buckets + bucket
	uint32_t zero_block_begin[MAX_BUCKET_LEVELS + 1];  // pass-A zeroing blocks of slot j (4 KiB each; none unless chunked && !accumulate)
	uint64_t queue_base[MAX_BUCKET_LEVELS];       // first pair of slot j's queues
};
constexpr uint32_t ZERO_BLOCK_HALVES = 2048;  // 4 KiB per zeroing block

// record = {entry index within the level, payload}: payload = F halves packed in pairs (F == 1: one fp32, the
// reference's grad_t for a single feature is float, grid.h:665)
//
// Queue unit: a PAIR of records -- the two corners that differ in dimension 0 only -- in 1 + 2 * PAYLOAD_WORDS words:
//   word 0 = index of the first entry (25 bits) | t << 25 | has_second << 30, then the two payloads.
// The second entry is DERIVED: dense-indexed levels: index + 1 (wrapping at the table size); hashed levels (prime[0] == 1,
// power-of-two table): index ^ (2^(t+1) - 1), t = number of trailing one bits of the cell's x coordinate.  12 bytes per
// pair for F == 2 instead of 16: the queues are the backward pass's HBM traffic.
template <uint32_t F>
struct BucketRecord {
	static constexpr uint32_t PAYLOAD_WORDS = (F + 1) / 2;
	static constexpr uint32_t WORDS = 1 + PAYLOAD_WORDS;           // overflow-list record: one entry
	static constexpr uint32_t PAIR_WORDS = 1 + 2 * PAYLOAD_WORDS;  // queue record: two entries
};
constexpr uint32_t PAIR_INDEX_BITS = 25, PAIR_INDEX_MASK = (1u << PAIR_INDEX_BITS) - 1u, PAIR_HAS_SECOND = 1u << 30;
template <uint32_t D>
TCNN_DEVICE uint32_t pair_second_index(const Level<D>& lv, uint32_t word0) {
	const uint32_t i0 = word0 & PAIR_INDEX_MASK;
	if (lv.fast) return (i0 ^ ((2u << ((word0 >> PAIR_INDEX_BITS) & 31u)) - 1u)) & lv.mask;
	const uint32_t i1 = i0 + 1u;
	return i1 == lv.hashmap_size ? 0u : i1;
}
// samples per thread of pass A: as many as fit the staging area, at least one
TCNN_HOST_DEVICE constexpr uint32_t bucket_spt(uint32_t D, uint32_t F) {
	const uint32_t per_sample_bytes = ((1u << D) / 2u) * (1u + 2u * ((F + 1) / 2)) * 4u;
	const uint32_t spt = BUCKET_STAGE_BYTES / (per_sample_bytes * BUCKET_THREADS);
	return spt < 1u ? 1u : (spt > 8u ? 8u : spt);
}

// The queues are written once and read once: stream them past the caches (non-temporal) so that they do not evict the
// optimizer state the step's last kernel re-reads.  TCNN_QUEUE_TEMPORAL=1 builds the plain variant for A/B runs.
#if defined(TCNN_HOST_EMU) || defined(TCNN_QUEUE_TEMPORAL)
TCNN_DEVICE void queue_store(uint32_t* p, uint32_t v) { *p = v; }
TCNN_DEVICE uint32_t queue_load(const uint32_t* p) { return *p; }
#else
TCNN_DEVICE void queue_store(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
TCNN_DEVICE uint32_t queue_load(const uint32_t* p) { return __builtin_nontemporal_load(p); }
#endif
constexpr uint32_t BUCKET_INVALID_INDEX = 0xFFFFFFFFu;  // second record of a pair that has none
TCNN_DEVICE uint32_t h2_bits(h2 v) { return __builtin_bit_cast(uint32_t, v); }
TCNN_DEVICE h2 bits_h2(uint32_t v) { return __builtin_bit_cast(h2, v); }

// Fixed-point exponent of a bucket owner's accumulators (pass B): a record v is accumulated as the integer round(v * 2^k).
//   IEEE half: k = 24 for every slice -- a half times 2^24 is an integer already (11 significant bits, exponent >= -24): exact sums.
//   bfloat16 (-DTCNN_BF16): the type reaches down to 2^-133, and at a fixed 2^-24 records below 2^-25 vanished and small ones lost most of
//   their eight bits (round 5's stress-shape test had to tolerate entries that the oracle touched and the GPU left at zero).  k is chosen per
//   slice from what pass A measured: the level's sum of |dL/dy| over the batch (a 64-bit integer sum, so the same k every run) divided by
//   the level's slices, with a factor 8 of headroom over that uniform share -- k = 30 - ceil(log2(8 * share)), 20 <= k <= 40.  At the
//   stress shape: k = 31 - 34 for the hashed levels, resolution 2^-31 and finer against records of 1e-7 and up.  A slice whose records
//   exceed the headroom (clustered samples) fails the int32 bound test as before and is redone with 64 bits per value at the same k.
struct OwnerScale {
	int k;
	TCNN_DEVICE float up(float v) const { return HALF_IS_BF16 ? __builtin_ldexpf(v, k) : v * 16777216.0f; }
	TCNN_DEVICE float down(float v) const { return HALF_IS_BF16 ? __builtin_ldexpf(v, -k) : v * (1.0f / 16777216.0f); }
	TCNN_DEVICE float safe_abs_sum() const { return HALF_IS_BF16 ? __builtin_ldexpf(0.9375f, 31 - k) : 120.0f; }  // < 2^31 / 2^k, with room for the bound's own rounding
	TCNN_DEVICE double up64() const { return HALF_IS_BF16 ? __builtin_ldexp(1.0, k) : 16777216.0; }
	TCNN_DEVICE double down64() const { return HALF_IS_BF16 ? __builtin_ldexp(1.0, -k) : 1.0 / 16777216.0; }
};
constexpr uint32_t LEVEL_SUM_PARTS = 8;  // words a level's sum is spread over (the scatter's workgroups add into word blockIdx % 8)
constexpr float LEVEL_SUM_CLAMP = 4096.0f;  // per sample: 2^18 .. 2^20 samples of it stay inside 64 bits at 2^-32 units
template <typename PLAN>
TCNN_DEVICE OwnerScale owner_scale(const PLAN& plan, const uint32_t* counters, uint32_t j) {
	if constexpr (!HALF_IS_BF16) return OwnerScale{24};
	unsigned long long sum = 0;
#pragma unroll
	for (uint32_t p = 0; p < LEVEL_SUM_PARTS; ++p) sum += *(const unsigned long long*)(counters + plan.level_sum_base + 2u * (plan.sum_slot(j) * LEVEL_SUM_PARTS + p));
	if (sum == 0ull) return OwnerScale{40};
	const float share = (float)sum * (8.0f / 4294967296.0f) / (float)(plan.n_buckets[j] * plan.n_chunks[j]);
	int e;
	(void)__builtin_frexpf(share, &e);  // share < 2^e
	const int k = 30 - e;
	return OwnerScale{k < 20 ? 20 : (k > 40 ? 40 : k)};
}

// the 64-bit-per-value forms: IEEE half through to_fixed() (fp32 / int32 operations only); bfloat16 at the slice's exponent
TCNN_DEVICE long long to_fixed64(float v, const OwnerScale& sc) {
	if constexpr (!HALF_IS_BF16) return to_fixed(v);
	const double s = (double)v * sc.up64();
	if (!(__builtin_fabs(s) < 9.0e18)) return 0;  // beyond 64 bits, infinite or NaN: gradients no sum can represent (the reference's atomics would carry NaN / Inf on)
	return (long long)__builtin_rint(s);
}
TCNN_DEVICE half_t from_fixed64(long long q, const OwnerScale& sc) { return (half_t)(float)((double)q * sc.down64()); }

// SECOND_ORDER: scatter d(dL_dx)/d(grid) instead of dy/d(grid) (backward_backward_input's parameter part) -- a compile-time switch: as a
// run-time select the corner weight of the second-order form (three products per dimension and corner) sits next to the first-order one in
// every training step's instruction stream and register budget
template <uint32_t D, uint32_t F, bool SECOND_ORDER>
__global__ void __launch_bounds__(BUCKET_THREADS) k_grid_bucket_scatter(const GridMeta meta, const GridIO io, const BucketPlan plan,
                                                                         const half_t* __restrict__ dL_dy, uint32_t* __restrict__ counters,
                                                                         uint32_t* __restrict__ queues, uint32_t* __restrict__ overflow,
                                                                         half_t* __restrict__ grid_gradient) {
	constexpr uint32_t N_CORNERS = 1u << D, PW = BucketRecord<F>::PAYLOAD_WORDS, W = BucketRecord<F>::WORDS, PWP = BucketRecord<F>::PAIR_WORDS;
	constexpr uint32_t N_PAIRS_PER_SAMPLE = N_CORNERS / 2;
	constexpr uint32_t SPT = bucket_spt(D, F), TILE = SPT * BUCKET_THREADS, N_PAIR = TILE * N_PAIRS_PER_SAMPLE;
	constexpr uint32_t INVALID = BUCKET_INVALID_INDEX;
	TCNN_DYN_LDS(lds_raw);
	constexpr uint32_t diag_scatter = EXP_DIAG_SCATTER;  // 0 in the product build (exp_diag.h)
	if (blockIdx.x >= plan.scatter_blocks) {
		// gradients of chunked levels are accumulated with atomics by several owners in pass B: zero them here
		const uint32_t z = blockIdx.x - plan.scatter_blocks;
		uint32_t zj = 0;
		while (zj + 1 < plan.n_levels && z >= plan.zero_block_begin[zj + 1]) ++zj;
		const uint32_t zl = plan.level[zj];
		const uint32_t n_halves = (meta.offset[zl + 1] - meta.offset[zl]) * F;  // a multiple of 8
		const uint32_t h = (z - plan.zero_block_begin[zj]) * ZERO_BLOCK_HALVES + threadIdx.x * 8u;
		if (h < n_halves) *(u4*)(grid_gradient + (size_t)meta.offset[zl] * F + h) = u4{0u, 0u, 0u, 0u};
		return;
	}
	// persistent workgroup: `wgs_per_level` of them share the sample tiles of one level
	const uint32_t j = blockIdx.x / plan.wgs_per_level, first_tile = blockIdx.x % plan.wgs_per_level;
	const uint32_t level = plan.level[j], nb = plan.n_buckets[j], shift = plan.shift;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	if ((float)level > max_level + 1e-3f) return;  // grid.h:242: no records, the owners store zeros
	const Level<D> lv = make_level<D>(meta, level);

	// Queue unit: a PAIR of records -- the two corners that differ in dimension 0 only.  Their table entries are
	// neighbours (dense index +1; hashed: prime[0] == 1, so the indices differ in the low bits only) and therefore
	// share a bucket except once in ~2^shift pairs: the second record of such a pair is routed through the overflow
	// list instead.  Halves the ranking / reordering work per corner; a pair is 16 bytes for F == 2.
	uint32_t* stage = (uint32_t*)lds_raw;   // [N_PAIR][PWP]
	uint32_t* cnt = stage + N_PAIR * PWP;   // [nb] pairs of this tile per bucket
	uint32_t* delta = cnt + nb;             // [nb] exclusive prefix of cnt, later (queue position - staging position)
	uint32_t* part = delta + nb;            // [64] scan scratch of wave 0
	uint32_t* total_p = part + 64;          // [1]
	for (uint32_t b = threadIdx.x; b < nb; b += BUCKET_THREADS) cnt[b] = 0u;

	auto push_overflow = [&](uint32_t index, const uint32_t* payload) {
		const uint32_t o = atomic_add_u32(&counters[plan.overflow_counter], 1u);
		if (o < plan.overflow_capacity) {
			uint32_t* dst = overflow + (size_t)o * (W + 1);
			dst[0] = level;
			dst[1] = index;
#pragma unroll
			for (uint32_t p = 0; p < PW; ++p) dst[2 + p] = payload[p];
		}
	};
	auto load_tile = [&](uint32_t tile, float (&x)[SPT][D], half_t (&g)[SPT][F]) {
#pragma unroll
		for (uint32_t s = 0; s < SPT; ++s) {
			const uint32_t i = min(tile * TILE + s * BUCKET_THREADS + threadIdx.x, io.n - 1u);
			load_position<D>(io, i, x[s]);
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) g[s][f] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
		}
	};

	float x[SPT][D], x_next[SPT][D];
	half_t g[SPT][F], g_next[SPT][F];
	constexpr bool second_order = SECOND_ORDER;
	if (first_tile < plan.tiles) load_tile(first_tile, x, g);
#if !defined(TCNN_HOST_EMU)
	// The first tile's inputs are waited for HERE, not at the loop's top.  gfx9 counts loads and stores in one counter (vmcnt); the waits
	// the compiler places in the loop header serve the first iteration (inputs still on their way) and every later one (inputs long there:
	// the wait for the reservation atomics covered them) alike, and on the later ones "s_waitcnt vmcnt(0)" sits out the round trip of the
	// queue stores the previous tile's append loop has just issued -- once per tile and workgroup.
#pragma unroll
	for (uint32_t s = 0; s < SPT; ++s) {
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) asm volatile("" : "+v"(x[s][d]));
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) asm volatile("" : "+v"(g[s][f]));
	}
#endif
	__syncthreads();

	float level_abs_sum = 0.0f;  // (bfloat16 build)
	for (uint32_t tile = first_tile; tile < plan.tiles; tile += plan.wgs_per_level) {
		const uint32_t chunk = tile / plan.tiles_per_chunk[j];
		uint32_t* __restrict__ my_counters = counters + plan.counter_base[j] + chunk * nb;

#if defined(TCNN_BF16)  // the level's sum of |dL/dy| (OwnerScale), gathered per thread over the workgroup's tiles
#pragma unroll
		for (uint32_t s = 0; s < SPT; ++s) {
			float m = 0.0f;
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) m = __builtin_fmaxf(m, __builtin_fabsf((float)g[s][f]));
			if (tile * TILE + s * BUCKET_THREADS + threadIdx.x < io.n) level_abs_sum += __builtin_fminf(m, LEVEL_SUM_CLAMP);  // (NaN -> the other operand: the bound test sees it)
		}
#endif
		// ---- derive the records of my samples; rank each pair within its bucket
		uint32_t ridx[SPT][N_CORNERS], rank[SPT][N_PAIRS_PER_SAMPLE], pay[SPT][N_CORNERS][PW];
		auto derive = [&](auto fast_tag) {
			constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
			for (uint32_t s = 0; s < SPT; ++s) {
				const bool valid = tile * TILE + s * BUCKET_THREADS + threadIdx.x < io.n;
				const Cell<D> c = make_cell<D, FAST>(lv, x[s]);
				float dd[D];
#pragma unroll
				for (uint32_t d = 0; d < D; ++d) dd[d] = 0.0f;
				if constexpr (second_order) load_ddx<D>(io, min(tile * TILE + s * BUCKET_THREADS + threadIdx.x, io.n - 1u), dd);
#pragma unroll
				for (uint32_t idx = 0; idx < N_CORNERS; ++idx) {
					float weight;
					if constexpr (second_order) weight = corner_weight_second_order<D>(lv, c, idx, dd);
					else weight = lv.nearest ? 1.0f : corner_weight<D>(c, idx);
					if constexpr (F == 1) {
						pay[s][idx][0] = __builtin_bit_cast(uint32_t, weight * (float)g[s][0]);
					} else {
						const half_t wh = to_half_rn(weight);  // (GRAD_T)weight, grid.h:254
						const h2 w2 = h2{wh, wh};
#pragma unroll
						for (uint32_t p = 0; p < PW; ++p) pay[s][idx][p] = h2_bits(w2 * h2{g[s][2 * p], g[s][2 * p + 1]});
					}
					ridx[s][idx] = corner_index<D, FAST>(lv, c, idx);
				}
				// Hashed levels (prime[0] == 1): the two entries of EVERY pair of a sample differ by the same low bits, x ^ (x + 1) under the table's
				// mask -- whether a pair's second record can ride with the first (same bucket; derivable from word 0 by construction) and the
				// flip count t are properties of the sample, not of the pair
				uint32_t fast_tag = 0;
				bool fast_together = true;
				if constexpr (FAST) {
					const uint32_t flips = c.hlo[0] ^ c.hhi[0];
					fast_together = ((flips & lv.mask) >> shift) == 0u;
					fast_tag = fast_together ? ((((uint32_t)__builtin_popcount(flips) - 1u) << PAIR_INDEX_BITS) | PAIR_HAS_SECOND) : 0u;
				}
#pragma unroll
				for (uint32_t pr = 0; pr < N_PAIRS_PER_SAMPLE; ++pr) {
					const bool live = valid && (pr == 0u || !lv.nearest);
					const uint32_t bucket = ridx[s][2 * pr] >> shift;
					if (!live) {
						ridx[s][2 * pr] = INVALID;
						ridx[s][2 * pr + 1] = INVALID;
					} else if (lv.nearest) {
						ridx[s][2 * pr + 1] = INVALID;
					} else if constexpr (FAST) {
						if (!fast_together) {
							push_overflow(ridx[s][2 * pr + 1], pay[s][2 * pr + 1]);
							ridx[s][2 * pr + 1] = INVALID;
						}
						ridx[s][2 * pr] |= fast_tag;
					} else {
						// word 0 of the pair; the second entry must be derivable from it AND live in the same bucket,
						// otherwise (about one pair in 2^shift) it travels through the overflow list
						uint32_t t = 0;
						if constexpr (FAST) t = (uint32_t)__builtin_popcount(c.hlo[0] ^ c.hhi[0]) - 1u;
						const uint32_t word0 = ridx[s][2 * pr] | (t << PAIR_INDEX_BITS) | PAIR_HAS_SECOND;
						const uint32_t i1 = ridx[s][2 * pr + 1];
						if ((i1 >> shift) != bucket || pair_second_index<D>(lv, word0) != i1) {
							push_overflow(i1, pay[s][2 * pr + 1]);
							ridx[s][2 * pr + 1] = INVALID;
						} else {
							ridx[s][2 * pr] = word0;
						}
					}
					rank[s][pr] = live && !(diag_scatter & 8u) ? atomic_add_u32(&cnt[bucket], 1u) : 0u;
				}
			}
		};
		if (lv.fast) derive(std::true_type{}); else derive(std::false_type{});
		// the next tile's inputs travel while this one is ranked, reordered and written
		const uint32_t next_tile = tile + plan.wgs_per_level;
		if (next_tile < plan.tiles) load_tile(next_tile, x_next, g_next);
		__syncthreads();

		// ---- reserve this tile's run in every bucket queue (one returning global atomic per non-empty bucket; the
		// common case has one bucket per thread and hides the round trip behind the scan and the reordering) ...
		uint32_t reserved = 0;
		if (nb <= BUCKET_THREADS && threadIdx.x < nb) {
			const uint32_t c = cnt[threadIdx.x];
			if (c && !(diag_scatter & 2u)) reserved = atomic_add_u32(&my_counters[threadIdx.x], c);
		}
		// ... while wave 0 turns the counts into staging offsets (exclusive scan, wave-synchronous)
		if (threadIdx.x < WAVE) {
			const uint32_t per_lane = div_round_up(nb, WAVE);
			const uint32_t b_begin = min(threadIdx.x * per_lane, nb), b_end = min(b_begin + per_lane, nb);
			uint32_t sum = 0;
			for (uint32_t b = b_begin; b < b_end; ++b) sum += cnt[b];
			// exclusive prefix over the wave's lanes in registers: sibling blocks of 1, 2, 4, ... lanes merge, a lane in the upper sibling adds
			// the lower sibling's total (six cross-lane moves; the LDS ladder this replaces cost eighteen LDS round trips while three waves wait)
			uint32_t block_total = sum, running = 0;
#pragma unroll
			for (uint32_t d = 1; d < WAVE; d <<= 1) {
				const uint32_t sibling = (uint32_t)__shfl_xor((int)block_total, (int)d, 64);
				if (threadIdx.x & d) running += sibling;
				block_total += sibling;
			}
			for (uint32_t b = b_begin; b < b_end; ++b) {
				delta[b] = running;
				running += cnt[b];
			}
			if (threadIdx.x == WAVE - 1) total_p[0] = block_total;
		}
		__syncthreads();
		const uint32_t total = total_p[0];

		// ---- reorder by bucket in LDS
#pragma unroll
		for (uint32_t s = 0; s < SPT; ++s) {
#pragma unroll
			for (uint32_t pr = 0; pr < N_PAIRS_PER_SAMPLE; ++pr) {
				const uint32_t word0 = ridx[s][2 * pr];  // index | t | has_second (INVALID: no pair)
				if (word0 == INVALID) continue;
				const uint32_t pos = delta[(word0 & PAIR_INDEX_MASK) >> shift] + rank[s][pr];
				if (diag_scatter & 4u) continue;
				stage[pos * PWP] = word0;
#pragma unroll
				for (uint32_t p = 0; p < PW; ++p) {
					stage[pos * PWP + 1 + p] = pay[s][2 * pr][p];
					stage[pos * PWP + 1 + PW + p] = pay[s][2 * pr + 1][p];
				}
			}
		}
		__syncthreads();
#if !defined(TCNN_HOST_EMU)
		// (the next tile's inputs, requested before the ranking barrier, are waited for here -- ahead of this tile's queue stores, see the
		// note at the first tile's loads: nothing of this lane's is in flight when the stores go out, and nothing waits behind them)
#pragma unroll
		for (uint32_t s = 0; s < SPT; ++s) {
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) asm volatile("" : "+v"(x_next[s][d]));
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) asm volatile("" : "+v"(g_next[s][f]));
		}
#endif
		// delta[b] := (position of the run in bucket b's queue) - (position of the run in the staging area);
		// the counts are dead from here on: clear them for the next tile
		if (nb <= BUCKET_THREADS) {
			if (threadIdx.x < nb) {
				delta[threadIdx.x] = reserved - delta[threadIdx.x];
				cnt[threadIdx.x] = 0u;
			}
		} else {
			for (uint32_t b = threadIdx.x; b < nb; b += BUCKET_THREADS) {
				const uint32_t c = cnt[b];
				delta[b] = (c ? atomic_add_u32(&my_counters[b], c) : 0u) - delta[b];
				cnt[b] = 0u;
			}
		}
		__syncthreads();

		// ---- append the runs to the bucket queues: consecutive threads -> consecutive pairs
		const uint32_t cap = plan.capacity[j];
		uint32_t* __restrict__ q = queues + (plan.queue_base[j] + (size_t)chunk * nb * cap) * PWP;
		for (uint32_t t = threadIdx.x; t < total; t += BUCKET_THREADS) {
			uint32_t rec[PWP];
#pragma unroll
			for (uint32_t w = 0; w < PWP; ++w) rec[w] = stage[t * PWP + w];
			const uint32_t b = (rec[0] & PAIR_INDEX_MASK) >> shift;
			const uint32_t pos = t + delta[b];  // wraps like the subtraction above
			if (pos < cap) {
				uint32_t* dst = q + ((size_t)b * cap + pos) * PWP;
				if (diag_scatter & 1u) continue;
#pragma unroll
				for (uint32_t w = 0; w < PWP; ++w) queue_store(dst + w, rec[w]);
			} else {
				push_overflow(rec[0] & PAIR_INDEX_MASK, &rec[1]);
				if (rec[0] & PAIR_HAS_SECOND) push_overflow(pair_second_index<D>(lv, rec[0]), &rec[1 + PW]);
			}
		}
		__syncthreads();  // the staging area and the offsets are reused by the next tile
#pragma unroll
		for (uint32_t s = 0; s < SPT; ++s) {
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) x[s][d] = x_next[s][d];
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) g[s][f] = g_next[s][f];
		}
	}
#if defined(TCNN_BF16)
	{  // ONE 64-bit integer atomic per workgroup, into one of the level's LEVEL_SUM_PARTS words (per wave and tile -- 2048 same-address atomics per
	   // level -- the atomics serialised in their L2 channel and the pass took five times as long)
		const float wave_total = wave_sum_f32(level_abs_sum);
		if (lane_id() == 0) part[threadIdx.x / WAVE] = __builtin_bit_cast(uint32_t, wave_total);
		__syncthreads();
		if (threadIdx.x == 0) {
			float total = 0.0f;
			for (uint32_t w = 0; w < BUCKET_THREADS / WAVE; ++w) total += __builtin_bit_cast(float, part[w]);
			if (total > 0.0f) {
				atomicAdd((unsigned long long*)(counters + plan.level_sum_base + 2u * (j * LEVEL_SUM_PARTS + (blockIdx.x % LEVEL_SUM_PARTS))),
				          (unsigned long long)((double)total * 4294967296.0));
			}
		}
	}
#else
	(void)level_abs_sum;
#endif
}

// What every owner of a (bucket, chunk) does last.  Every thread read the counters before the barriers of the caller: they end
// the call zeroed.  The last owner to get here (all owners have read the overflow count by then) resets the two bookkeeping
// counters -- after draining a long overflow list with the reference's global atomics.
template <uint32_t F, uint32_t THREADS, typename PLAN>
TCNN_DEVICE void bucket_owner_epilogue(const GridMeta& meta, const PLAN& plan, uint32_t j, uint32_t queue, bool inline_overflow, uint32_t n_over,
                                       uint32_t* __restrict__ counters, const uint32_t* __restrict__ overflow, half_t* __restrict__ grid_gradient) {
	constexpr uint32_t PW = BucketRecord<F>::PAYLOAD_WORDS, OW = BucketRecord<F>::WORDS + 1;
	__shared__ uint32_t last_owner;
	__syncthreads();  // this slice's stores are issued
	if (threadIdx.x == 0) {
		counters[plan.counter_base[j] + queue] = 0u;
		if (!inline_overflow) {  // the drain's atomics execute memory-side: the slices must be there first (release, agent scope)
#if !defined(TCNN_HOST_EMU)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
		}
		last_owner = atomic_add_u32(&counters[plan.overflow_counter + 1], 1u) == plan.n_owner_blocks - 1u ? 1u : 0u;
	}
	__syncthreads();
	if (last_owner) {
		if (!inline_overflow) {
#if !defined(TCNN_HOST_EMU)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
			for (uint32_t t = threadIdx.x; t < n_over; t += THREADS) {
				const uint32_t* rec = overflow + (size_t)t * OW;
				half_t* __restrict__ g = grid_gradient + (size_t)meta.offset[rec[0]] * F;
				const uint32_t index = rec[1];
				if constexpr (F == 1) {
					const half_t v = (half_t)__builtin_bit_cast(float, rec[2]);
					atomic_add_h2(g + (index & ~1u), (index & 1u) ? h2{(half_t)0.0f, v} : h2{v, (half_t)0.0f});
				} else {
#pragma unroll
					for (uint32_t p = 0; p < PW; ++p) atomic_add_h2(g + (size_t)index * F + 2 * p, bits_h2(rec[2 + p]));
				}
			}
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			counters[plan.overflow_counter] = 0u;
			counters[plan.overflow_counter + 1] = 0u;
		}
		if constexpr (HALF_IS_BF16) {  // (every owner read its level's sum before it signed off)
			for (uint32_t t = threadIdx.x; t < 2u * LEVEL_SUM_PARTS * plan.n_levels; t += THREADS) counters[plan.level_sum_base + t] = 0u;
		}
	}
}

// pass B: the owner of bucket `bucket` of slot `j` streams its queue into a 64-bit fixed-point LDS table
template <uint32_t D, uint32_t F>
TCNN_DEVICE void bucket_level(const GridMeta& meta, const Level<D>& lv, uint32_t level, uint32_t j, uint32_t bucket, uint32_t chunk,
                              const BucketPlan& plan, uint32_t* __restrict__ counters, const uint32_t* __restrict__ queues,
                              const uint32_t* __restrict__ overflow, half_t* __restrict__ grid_gradient, bool accumulate, unsigned char* lds_raw) {
	constexpr uint32_t PW = BucketRecord<F>::PAYLOAD_WORDS, PWP = BucketRecord<F>::PAIR_WORDS, OW = BucketRecord<F>::WORDS + 1;
	// records that did not fit their queue (or whose x-neighbour lives in another bucket: about one pair in 2^shift).  Up to
	// OVERFLOW_INLINE_MAX of them every owner picks its own out of the list -- exact, no atomics, no extra launch; beyond
	// that (strongly clustered inputs) the last owner to finish sends the list through the reference's global atomics.
	const uint32_t n_over = min(counters[plan.overflow_counter], plan.overflow_capacity);
	const bool inline_overflow = n_over <= OVERFLOW_INLINE_MAX;
	const uint32_t entries_per_bucket = 1u << plan.shift;
	const uint32_t slice_begin = bucket * entries_per_bucket;
	const uint32_t slice_count = slice_begin < lv.hashmap_size ? min(entries_per_bucket, lv.hashmap_size - slice_begin) : 0u;
	unsigned long long* tab = (unsigned long long*)lds_raw;  // [entries][F]
	const OwnerScale sc = owner_scale(plan, counters, j);
	const uint32_t cap = plan.capacity[j], n_chunks = plan.n_chunks[j];
	const uint32_t queue = chunk * plan.n_buckets[j] + bucket;
	const uint32_t count = min(counters[plan.counter_base[j] + queue], cap);  // in flight while the table is cleared
	const uint32_t* __restrict__ q = queues + (plan.queue_base[j] + (size_t)queue * cap) * PWP;  // `count` PAIRS of records
	for (uint32_t e = threadIdx.x; e < slice_count * F / 2; e += SLICED_THREADS) ((u4*)lds_raw)[e] = u4{0u, 0u, 0u, 0u};  // slice_count * F is even
	__syncthreads();

	auto add_record = [&](uint32_t index, const uint32_t* payload) {
		const uint32_t rel = index & (entries_per_bucket - 1u);
		if constexpr (F == 1) {
			lds_atomic_add_u64(&tab[rel], (unsigned long long)to_fixed64(__builtin_bit_cast(float, payload[0]), sc));
		} else {
#pragma unroll
			for (uint32_t p = 0; p < PW; ++p) {
				const h2 v = bits_h2(payload[p]);
				lds_atomic_add_u64(&tab[rel * F + 2 * p], (unsigned long long)to_fixed64((float)v[0], sc));
				lds_atomic_add_u64(&tab[rel * F + 2 * p + 1], (unsigned long long)to_fixed64((float)v[1], sc));
			}
		}
	};
	{
		// U pair records (12 bytes each for F == 2) in flight per lane: the queue is streamed at memory speed, not at one
		// round trip per record
		constexpr uint32_t U = PWP <= 3 ? 8 : (PWP <= 5 ? 4 : 2);
		for (uint32_t base = threadIdx.x; base < count; base += SLICED_THREADS * U) {
			uint32_t rec[U][PWP];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) {
				const uint32_t t = min(base + u * SLICED_THREADS, count - 1u);
#pragma unroll
				for (uint32_t w = 0; w < PWP; ++w) rec[u][w] = queue_load(q + (size_t)t * PWP + w);
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) {
				if (base + u * SLICED_THREADS >= count) continue;
				add_record(rec[u][0] & PAIR_INDEX_MASK, &rec[u][1]);
				if (rec[u][0] & PAIR_HAS_SECOND) add_record(pair_second_index<D>(lv, rec[u][0]), &rec[u][1 + PW]);
			}
		}
	}
	if (inline_overflow && chunk == 0u) {  // overflow records of this slice (level, index): the first chunk's owner takes them
		for (uint32_t t = threadIdx.x; t < n_over; t += SLICED_THREADS) {
			const uint32_t* rec = overflow + (size_t)t * OW;
			if (rec[0] == level && (rec[1] >> plan.shift) == bucket) add_record(rec[1], rec + 2);
		}
	}
	__syncthreads();

	half_t* __restrict__ grad = grid_gradient + ((size_t)meta.offset[level] + slice_begin) * F;
	const uint32_t n_halves = slice_count * F;  // a multiple of 8: level sizes are multiples of 8
	for (uint32_t e2 = threadIdx.x; e2 < n_halves / 2; e2 += SLICED_THREADS) {
		const long long q0 = ((const long long*)lds_raw)[2 * e2], q1 = ((const long long*)lds_raw)[2 * e2 + 1];
		h2 v = h2{from_fixed64(q0, sc), from_fixed64(q1, sc)};
		if (n_chunks == 1) {  // sole owner of the slice: plain stores
			if (accumulate) v += *(const h2*)(grad + 2 * e2);
			*(h2*)(grad + 2 * e2) = v;
		} else if (v[0] != (half_t)0.0f || v[1] != (half_t)0.0f) {
			atomic_add_h2(grad + 2 * e2, v);  // small tables only: (table size) x (chunks) updates per level
		}
	}
	bucket_owner_epilogue<F, SLICED_THREADS>(meta, plan, j, queue, inline_overflow, n_over, counters, overflow, grid_gradient);
}

// ---------------------------------------------------------------------------------------------
// pass B, packed form (even F; what the bucketed backward runs unless grid_owner_mode() == 1).
//
// The form above spends one ds_add_u64 per VALUE (two per table entry at F = 2) and 16 bytes of LDS per entry, so a
// 8192-entry slice takes 128 KiB: one workgroup per CU, whose clear / stream / convert phases cannot overlap anything.
// Here the two features of a payload word share ONE 64-bit LDS word: the addend is the two's-complement number
// V1 * 2^32 + V0 (V = value * 2^24, an exact int32 while |value| < 128), so a single ds_add_u64 accumulates both, and the
// sums come apart again as S0 = sign-extended low word, S1 = (X - S0) >> 32 -- provided both lie inside int32.  They do
// whenever the absolute values of all addends of the slice sum to less than 2^31 per feature (= 128.0 in gradient
// units; the sum over a whole LEVEL is sum_i |dL/dy_i| -- about loss_scale * mean error -- and a slice sees 1/64 of
// it): every lane keeps that running bound in fp32, the workgroup adds the lanes up once, and a slice that fails the
// test (huge or non-finite gradients) is simply redone with the 64-bit-per-value table above, in sub-slices that fit the
// same LDS.  Same exact sums, same single rounding, bit-identical output either way; half the LDS atomics, half the LDS
// (64 KiB: two workgroups share a CU, one streaming while the other clears or converts), no fp64 arithmetic.
// ---------------------------------------------------------------------------------------------
#ifndef TCNN_OWNER_THREADS
#define TCNN_OWNER_THREADS 512
#endif
constexpr uint32_t OWNER_THREADS = TCNN_OWNER_THREADS;

#if defined(TCNN_HOST_EMU)
inline unsigned long owner_slice_stats[2] = {0, 0};  // emulator only: slices finished from the packed table / redone wide
#else
// slices the packed owner kernel had to redo with 64 bits per value since the process started (grid_owner_wide_slices()): the redo
// costs that slice twice the time, so a workload whose gradients keep failing the int32 bound should be visible
__device__ unsigned long long g_owner_wide_slices = 0ull;
#endif

// round(v * 2^k) (OwnerScale; IEEE half: k = 24, |v| < 128); saturates beyond (such a slice fails the bound and is redone in 64 bits).  A
// 16-bit float times 2^24 is an integer already when the type is IEEE half (11 significant bits, exponent >= -24): the conversion
// instruction alone (v_cvt_i32_f32 saturates and maps NaN to 0 -- written as asm because the C++ conversion is undefined out of range).
TCNN_DEVICE int to_fixed32(float v, const OwnerScale& sc) {
#if defined(TCNN_HOST_EMU)
	v = __builtin_fminf(__builtin_fmaxf(v, -127.0f), 127.0f);
	return (int)__builtin_rintf(v * 16777216.0f);
#else
	float s = sc.up(v);
	if constexpr (HALF_IS_BF16) s = __builtin_rintf(s);  // bfloat16 records reach below 2^-k
	int r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(s));
	return r;
#endif
}
template <uint32_t D, uint32_t F, uint32_t THREADS, typename PLAN>
TCNN_DEVICE void bucket_level_packed(const GridMeta& meta, const Level<D>& lv, uint32_t level, uint32_t j, uint32_t bucket, uint32_t chunk,
                                     const PLAN& plan, uint32_t* __restrict__ counters, const uint32_t* queues,
                                     const uint32_t* __restrict__ overflow, half_t* __restrict__ grid_gradient, bool accumulate, unsigned char* lds_raw,
                                     uint32_t lds_bytes, bool force_wide) {
	static_assert(F % 2 == 0, "the packed owner pairs the features of a payload word");
	constexpr uint32_t PW = BucketRecord<F>::PAYLOAD_WORDS, PWP = BucketRecord<F>::PAIR_WORDS, OW = BucketRecord<F>::WORDS + 1;
	constexpr uint32_t N_WAVES = THREADS / WAVE;
	__shared__ float bound_parts[N_WAVES][F];
	const uint32_t entries_per_bucket = 1u << plan.shift;
	const uint32_t slice_begin = bucket * entries_per_bucket;
	const uint32_t slice_count = slice_begin < lv.hashmap_size ? min(entries_per_bucket, lv.hashmap_size - slice_begin) : 0u;
	const uint32_t cap = plan.capacity[j], n_chunks = plan.n_chunks[j];
	const uint32_t queue = chunk * plan.n_buckets[j] + bucket;
	// (not __restrict__, and neither is `queues`: loads the compiler may treat as invariant are moved wherever it likes -- it sank
	// the first round below the barrier, next to its use)
	const uint32_t* q = queues + (plan.queue_base[j] + (size_t)queue * cap) * PWP;  // `count` PAIRS of records
	half_t* __restrict__ grad = grid_gradient + ((size_t)plan.table_offset(meta, level) + slice_begin) * F;

	// U pair records (12 bytes each for F == 2) in flight per lane.  The FIRST round is requested right here, before the queue's
	// length is known (a queue holds `cap` records of memory whatever its count; what lies beyond the count is never used): it
	// travels together with the counters and while the table is cleared, instead of one more memory round trip after them -- a
	// 196 KiB queue is only four rounds per lane, and a workgroup with nothing in flight is a workgroup not streaming
	// (profiles/r03_exp_notes.txt: the pass moved its 201 MB at 4.3 TB/s with everything but the loads compiled out).
	constexpr uint32_t STREAM_U = PWP <= 3 ? 8 : (PWP <= 5 ? 4 : 2);
	auto load_round = [&](uint32_t base, uint32_t last, uint32_t (&rec)[STREAM_U][PWP]) {
#pragma unroll
		for (uint32_t u = 0; u < STREAM_U; ++u) {
			const uint32_t t = min(base + u * THREADS, last);
#pragma unroll
			for (uint32_t w = 0; w < PWP; ++w) rec[u][w] = queue_load(q + (size_t)t * PWP + w);
		}
	};
	const uint32_t n_over = min(counters[plan.overflow_counter], plan.overflow_capacity);
	const bool inline_overflow = n_over <= OVERFLOW_INLINE_MAX;
	const uint32_t count = min(counters[plan.counter_base[j] + queue], cap);  // in flight while the table is cleared
	constexpr uint32_t diag_owner = EXP_DIAG_OWNER;  // 0 in the product build (exp_diag.h)
	const OwnerScale sc = owner_scale(plan, counters, j);
	bool safe = !force_wide;
	// the packed table is cleared first (LDS only), the first round requested behind it: nothing then stands between the loads and
	// their use but the barrier (cleared after the loads, the compiler parks part of a record in other registers and waits for it)
	// (both happen at the top of the `if (safe)` block below -- ONE block from the issue of the hand-made loads to their last use, so that no
	// control-flow path of the compiled code leads from an issued load to anything but its wait: scripts/check_asm_load_hazard.py checks that)
	// Three-word records (F <= 2) on the GPU: the stream is PIPELINED BY HAND, two half-rounds of STREAM_U / 2 records per lane that are
	// consumed and re-requested in turn -- while the records of one half go through the conversions and LDS atomics (45 VALU
	// instructions each; the four waves of a SIMD all want the ALU when their loads arrive), the other half's loads are on their
	// way.  Written in C++ the compiler rotates the record registers (copies at the loop's back edge) and waits for the loads it
	// has just issued before it copies them (tried twice, profiles/r03_exp_notes.txt 10b, r04_exp_notes.txt): hence loads the
	// compiler does not see (asm), into registers that keep their identity, with counted waits.  vmcnt counts in issue order, so
	// "at most 4 outstanding" means the OLDER half has landed whatever the compiler's own loads do around it.
#if !defined(TCNN_HOST_EMU) && !defined(TCNN_OWNER_PLAIN_STREAM)
	constexpr bool PIPELINED = PWP == 3 && STREAM_U == 8;
#else
	constexpr bool PIPELINED = false;
#endif
	typedef uint32_t rec3_t __attribute__((ext_vector_type(3)));
#ifndef TCNN_OWNER_GROUPS
#define TCNN_OWNER_GROUPS 2
#endif
	constexpr uint32_t NG = TCNN_OWNER_GROUPS;  // groups of 4 records in flight per lane (3 / 4 / 6 / 8 groups measured 0.0524 / 0.0538 / 0.0574 / 0.0710 ms against 0.0510: profiles/r04_exp_notes.txt 14d)
	rec3_t grp[NG][4];
	uint32_t first_round[PIPELINED ? 1 : STREAM_U][PWP];
#if !defined(TCNN_HOST_EMU)
	const uint64_t q_address = (uint64_t)(uintptr_t)q;  // wave-uniform: into a scalar register pair, the loads' base
	// (readfirstlane returns a signed int: without the casts the low word is sign-extended over the high one)
	const uint64_t q_scalar = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(q_address >> 32)) << 32) |
	                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)q_address);
	auto issue_half = [&](rec3_t (&h)[4], uint32_t first, uint32_t last) {
#pragma unroll
		for (uint32_t u = 0; u < 4; ++u) {
			const uint32_t byte_offset = __umul24(min(first + u * THREADS, last), 12u);  // record index < 2^24: a queue holds < 2^31 records of < 2^32 bytes, see the check below
			asm volatile("global_load_dwordx3 %0, %1, %2 nt" : "=v"(h[u]) : "v"(byte_offset), "s"(q_scalar) : "memory");
		}
	};
	// (all NG groups outstanding: "at most 4 (NG - 1) loads outstanding" means the oldest group has landed)
	auto await_oldest_group = [&](rec3_t (&h)[4]) {
		asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]) : [n] "n"(4 * (NG - 1)) : "memory");
	};
#endif
	// streams the queue (and this slice's share of the overflow list) through `add(index, payload)`; `first`: the lane's first
	// round if it is in registers already (the first pass over the queue), null to load it here (the 64-bit redo)
	auto stream = [&](const uint32_t (*first)[PWP], auto&& add) {
		constexpr uint32_t U = STREAM_U;
		auto add_round = [&](uint32_t base, const uint32_t (&rec)[U][PWP]) {
			if (base + (U - 1) * THREADS < count) {  // all U records exist (every round but a lane's last): no per-record test
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					add(rec[u][0] & PAIR_INDEX_MASK, &rec[u][1]);
					if (rec[u][0] & PAIR_HAS_SECOND) add(pair_second_index<D>(lv, rec[u][0]), &rec[u][1 + PW]);
				}
			} else {
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					if (base + u * THREADS >= count) continue;
					add(rec[u][0] & PAIR_INDEX_MASK, &rec[u][1]);
					if (rec[u][0] & PAIR_HAS_SECOND) add(pair_second_index<D>(lv, rec[u][0]), &rec[u][1 + PW]);
				}
			}
		};
		uint32_t base = threadIdx.x;
		if (first) {
			if (base < count) add_round(base, *(const uint32_t (*)[U][PWP])first);
			base += THREADS * U;
		}
		for (; base < count; base += THREADS * U) {
			uint32_t rec[U][PWP];
			load_round(base, count - 1u, rec);
			add_round(base, rec);
		}
		if (inline_overflow && chunk == 0u) {  // overflow records of this slice (level, index): the first chunk's owner takes them
			for (uint32_t t = threadIdx.x; t < n_over; t += THREADS) {
				const uint32_t* rec = overflow + (size_t)t * OW;
				if (rec[0] == level && (rec[1] >> plan.shift) == bucket) add(rec[1], rec + 2);
			}
		}
	};
	// the same through the hand-pipelined halves (first pass over the queue only; its first two halves were requested above)
	auto stream_pipelined = [&](auto&& add) {
#if !defined(TCNN_HOST_EMU)
		auto add_half = [&](uint32_t first, const rec3_t (&h)[4]) {
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) {
				if (first + u * THREADS >= count) continue;
				const uint32_t rec[3] = {h[u][0], h[u][1], h[u][2]};
				add(rec[0] & PAIR_INDEX_MASK, &rec[1]);
				if (rec[0] & PAIR_HAS_SECOND) add(pair_second_index<D>(lv, rec[0]), &rec[1 + PW]);
			}
		};
		const uint32_t last = count ? count - 1u : 0u;
		// every round but the last re-requests its groups (workgroup-uniform trip count: every wave issues the same loads); the last
		// one only drains -- a request beyond the queue's end is a load instruction and a round trip the lane then has to wait out
		uint32_t round = 0;
		for (; round + 4u * NG * THREADS < count; round += 4u * NG * THREADS) {
			const uint32_t first = round + threadIdx.x;
#pragma unroll
			for (uint32_t k = 0; k < NG; ++k) {
				await_oldest_group(grp[k]);
				add_half(first + 4u * k * THREADS, grp[k]);
				issue_half(grp[k], first + 4u * (NG + k) * THREADS, last);
			}
		}
		// nothing of this lane's may still be on its way into registers the compiler is about to reuse
#pragma unroll
		for (uint32_t k = 0; k < NG; ++k) {
			asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(grp[k][0]), "+v"(grp[k][1]), "+v"(grp[k][2]), "+v"(grp[k][3]) : [n] "n"(4 * (NG - 1 - k)) : "memory");
			if (round < count) add_half(round + threadIdx.x + 4u * k * THREADS, grp[k]);
		}
#endif
		if (inline_overflow && chunk == 0u) {
			for (uint32_t t = threadIdx.x; t < n_over; t += THREADS) {
				const uint32_t* rec = overflow + (size_t)t * OW;
				if (rec[0] == level && (rec[1] >> plan.shift) == bucket) add(rec[1], rec + 2);
			}
		}
	};
	// one gradient pair leaves the slice: plain store for a sole owner, a packed atomic where sample chunks share the slice
	auto store_pair = [&](uint32_t e2, h2 v) {
		if (n_chunks == 1) {
			if (accumulate) v += *(const h2*)(grad + 2 * e2);
			*(h2*)(grad + 2 * e2) = v;
		} else if (v[0] != (half_t)0.0f || v[1] != (half_t)0.0f) {
			atomic_add_h2(grad + 2 * e2, v);  // small tables only: (table size) x (chunks) updates per level
		}
	};


	if (safe) {
		if (!(diag_owner & 1u)) {
			for (uint32_t e = threadIdx.x; e < slice_count * PW / 2; e += THREADS) ((u4*)lds_raw)[e] = u4{0u, 0u, 0u, 0u};  // slice_count is a multiple of 8
		}
		if constexpr (PIPELINED) {
#if !defined(TCNN_HOST_EMU)
			// (only the packed pass consumes them -- and nothing the compiler does not know of may stay in flight otherwise)
#pragma unroll
			for (uint32_t k = 0; k < NG; ++k) issue_half(grp[k], threadIdx.x + 4u * k * THREADS, cap - 1u);
#endif
		} else {
			load_round(threadIdx.x, cap - 1u, first_round);
		}
		unsigned long long* tab = (unsigned long long*)lds_raw;  // [entries][PW]: features 2p (low word) and 2p + 1 (high word)
		__syncthreads();  // the table is clear
		float bound[F];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) bound[f] = 0.0f;
		auto add_packed = [&](uint32_t index, const uint32_t* payload) {
			const uint32_t rel = index & (entries_per_bucket - 1u);
#pragma unroll
			for (uint32_t p = 0; p < PW; ++p) {
				const h2 v = bits_h2(payload[p]);
				const float f0 = (float)v[0], f1 = (float)v[1];
				bound[2 * p] += __builtin_fabsf(f0);
				bound[2 * p + 1] += __builtin_fabsf(f1);
				const int v0 = to_fixed32(f0, sc), v1 = to_fixed32(f1, sc);
				const unsigned long long x = ((unsigned long long)(uint32_t)(v1 + (v0 >> 31)) << 32) | (unsigned long long)(uint32_t)v0;
				if (!(diag_owner & 2u) || x == 0x123456789ull) lds_atomic_add_u64(&tab[rel * PW + p], x);
			}
		};
		if constexpr (PIPELINED) stream_pipelined(add_packed);
		else stream(first_round, add_packed);
		// the bound over the whole workgroup (NaN / Inf anywhere fail the comparison)
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) {
			const float w = wave_sum_f32(bound[f]);
			if (lane_id() == 0) bound_parts[threadIdx.x / WAVE][f] = w;
		}
		__syncthreads();  // also: every atomic of the slice has landed
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) {
			float total = 0.0f;
#pragma unroll
			for (uint32_t w = 0; w < N_WAVES; ++w) total += bound_parts[w][f];
			safe = safe && total < sc.safe_abs_sum();
		}
		if (safe && !(diag_owner & 4u)) {
			auto unpack = [&](uint32_t e2) {
				const long long x = (long long)tab[e2];
				const int s0 = (int)(uint32_t)(unsigned long long)x;
				const int s1 = (int)((x - (long long)s0) >> 32);
				// int32 -> fp32 rounds to nearest even exactly as the fp64 -> fp32 conversion of the wide form does
				return h2{(half_t)sc.down((float)s0), (half_t)sc.down((float)s1)};
			};
			if (n_chunks == 1 && !accumulate && ((uintptr_t)grad & 15u) == 0u) {  // sole owner, overwrite: 16 bytes per lane (slice_count * PW is a multiple of 8)
				for (uint32_t e8 = threadIdx.x; e8 < slice_count * PW / 4; e8 += THREADS) {
					const h2 a = unpack(4 * e8), b = unpack(4 * e8 + 1), c = unpack(4 * e8 + 2), d = unpack(4 * e8 + 3);
					*(h8*)(grad + 8 * e8) = h8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
				}
			} else {
				for (uint32_t e2 = threadIdx.x; e2 < slice_count * PW; e2 += THREADS) store_pair(e2, unpack(e2));
			}
		}
	}
#if defined(TCNN_HOST_EMU)
	if (threadIdx.x == 0) owner_slice_stats[safe ? 0 : 1]++;
#else
	if (!safe && !force_wide && threadIdx.x == 0) atomicAdd(&g_owner_wide_slices, 1ull);
#endif
	if (!safe) {
		// 64 bits per value, `sub` entries at a time (the same LDS): each pass streams the queue again and keeps its own entries
		unsigned long long* tab = (unsigned long long*)lds_raw;  // [sub][F]
		const uint32_t sub = max(8u, (lds_bytes / (F * 8u)) & ~7u);
		for (uint32_t sub_begin = 0; sub_begin < slice_count; sub_begin += sub) {
			const uint32_t sub_count = min(sub, slice_count - sub_begin);
			__syncthreads();  // the table is free (bound test / previous pass's conversion)
			for (uint32_t e = threadIdx.x; e < sub_count * F / 2; e += THREADS) ((u4*)lds_raw)[e] = u4{0u, 0u, 0u, 0u};
			__syncthreads();
			stream(nullptr, [&](uint32_t index, const uint32_t* payload) {
				const uint32_t rel = (index & (entries_per_bucket - 1u)) - sub_begin;
				if (rel >= sub_count) return;
#pragma unroll
				for (uint32_t p = 0; p < PW; ++p) {
					const h2 v = bits_h2(payload[p]);
					lds_atomic_add_u64(&tab[rel * F + 2 * p], (unsigned long long)to_fixed64((float)v[0], sc));
					lds_atomic_add_u64(&tab[rel * F + 2 * p + 1], (unsigned long long)to_fixed64((float)v[1], sc));
				}
			});
			__syncthreads();
			for (uint32_t e2 = threadIdx.x; e2 < sub_count * PW; e2 += THREADS) {
				const long long q0 = ((const long long*)lds_raw)[2 * e2], q1 = ((const long long*)lds_raw)[2 * e2 + 1];
				store_pair(sub_begin * PW + e2, h2{from_fixed64(q0, sc), from_fixed64(q1, sc)});
			}
		}
	}
	if (diag_owner & 8u) {  // (racy on purpose: the list counter is reset by whoever gets here)
		if (threadIdx.x == 0) {
			counters[plan.counter_base[j] + queue] = 0u;
			counters[plan.overflow_counter] = 0u;
		}
		return;
	}
	bucket_owner_epilogue<F, THREADS>(meta, plan, j, queue, inline_overflow, n_over, counters, overflow, grid_gradient);
}

// The workgroups of pass B that own a (bucket, chunk), packed form.  Launched as a 2-D grid -- blockIdx.y = the plan's item, blockIdx.x = the
// workgroup within it -- with everything a workgroup needs to know about its item in ONE descriptor in the kernel arguments: a single
// scalar load round before the queue's first records are requested.  (Round 5 looked the item up through the sliced kernel's plan: blocks per
// item -> block_begin[item] -> kind[item] -> level[item] -> n_slices[item] -> the level's table size -> the slot's queue geometry, six
// dependent loads, 1.6 us of a workgroup's 18.7 by the clock stamps of round 4 -- twice per launch, there are two generations of owners.)
// k_grid_backward_sliced (the other kinds of items, if the plan holds any) skips the bucket items when this kernel runs them.
struct OwnerItem {
	uint32_t level, slot, n_slices, n_blocks;      // n_blocks = n_slices x n_chunks workgroups belong to the item
	uint32_t hashmap_size, fast, offset, capacity;  // the level's table: entries, hashed power-of-two table?, first entry; pairs per queue
	uint32_t n_chunks, n_buckets, counter_base, pad;
	uint64_t queue_base;
};
struct OwnerItems {
	OwnerItem item[MAX_BUCKET_LEVELS];
};
// what bucket_level_packed and the epilogue read of a BucketPlan, for ONE slot (j == 0 indexes it), out of the descriptor
struct OwnerPlanView {
	uint32_t shift, overflow_counter, overflow_capacity, n_owner_blocks, level_sum_base, n_levels, slot;
	uint32_t capacity[1], n_chunks[1], n_buckets[1], counter_base[1];
	uint64_t queue_base[1];
	uint32_t offset;
	TCNN_HOST_DEVICE uint32_t sum_slot(uint32_t) const { return slot; }
	TCNN_HOST_DEVICE uint32_t table_offset(const GridMeta&, uint32_t) const { return offset; }
};
template <uint32_t D, uint32_t F>
__global__ void __launch_bounds__(OWNER_THREADS) k_grid_bucket_owner(const GridMeta meta, const OwnerItems items, const int accumulate, const uint32_t shift,
                                                                      const uint32_t overflow_counter, const uint32_t overflow_capacity, const uint32_t n_owner_blocks,
                                                                      const uint32_t level_sum_base, const uint32_t n_bucket_levels, uint32_t* __restrict__ counters,
                                                                      const uint32_t* queues, const uint32_t* __restrict__ overflow, half_t* __restrict__ grid_gradient,
                                                                      const uint32_t lds_bytes, const int force_wide) {
	TCNN_DYN_LDS(lds_raw);
	const OwnerItem it = items.item[blockIdx.y];
	const uint32_t local_block = blockIdx.x;
	if (local_block >= it.n_blocks) return;
	const uint32_t slice = local_block % it.n_slices, chunk = local_block / it.n_slices;
	Level<D> lv = {};  // (what the owner reads of it: the table's size and kind -- pair_second_index, the slice's extent)
	lv.hashmap_size = it.hashmap_size;
	lv.mask = it.hashmap_size - 1u;
	lv.fast = it.fast != 0u;
	OwnerPlanView view;
	view.shift = shift;
	view.overflow_counter = overflow_counter;
	view.overflow_capacity = overflow_capacity;
	view.n_owner_blocks = n_owner_blocks;
	view.level_sum_base = level_sum_base;
	view.n_levels = n_bucket_levels;
	view.slot = it.slot;
	view.offset = it.offset;
	view.capacity[0] = it.capacity;
	view.n_chunks[0] = it.n_chunks;
	view.n_buckets[0] = it.n_buckets;
	view.counter_base[0] = it.counter_base;
	view.queue_base[0] = it.queue_base;
	if constexpr (F % 2 == 0) {  // (never launched for odd F)
		bucket_level_packed<D, F, OWNER_THREADS>(meta, lv, it.level, 0u, slice, chunk, view, counters, queues, overflow, grid_gradient, accumulate != 0, lds_raw, lds_bytes,
		                                         force_wide != 0);
	}
}

template <uint32_t D, uint32_t F, bool PACKED>
__global__ void __launch_bounds__(SLICED_THREADS) k_grid_backward_sliced(const GridMeta meta, const GridIO io, const SlicePlan plan,
                                                                           const half_t* __restrict__ dL_dy, half_t* __restrict__ grid_gradient,
                                                                           const int accumulate, const BucketPlan bplan,
                                                                           uint32_t* __restrict__ counters, const uint32_t* __restrict__ queues,
                                                                           const uint32_t* __restrict__ overflow) {
	TCNN_DYN_LDS(lds_raw);
	uint32_t item = 0, local_block;
	if (plan.blocks_per_item) {
		// near-uniform plan (the bucketed backward): work item = blockIdx / stride, no search; an item with fewer workgroups
		// than the stride leaves the rest idle (each idle workgroup still claims a CU's LDS for an instant, so the host only
		// picks this when almost nothing is padded)
		item = blockIdx.x / plan.blocks_per_item;
		local_block = blockIdx.x % plan.blocks_per_item;
		if (local_block >= plan.block_begin[item + 1] - plan.block_begin[item]) return;
	} else {
		while (item + 1 < plan.n_items && blockIdx.x >= plan.block_begin[item + 1]) ++item;
		local_block = blockIdx.x - plan.block_begin[item];
	}
	const uint32_t level = plan.level[item], kind = plan.kind[item];
	const uint32_t n_slices = plan.n_slices[item];
	const uint32_t n_chunks = (plan.block_begin[item + 1] - plan.block_begin[item]) / n_slices;
	const uint32_t slice = local_block % n_slices, chunk = local_block / n_slices;

	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	const bool level_off = (float)level > max_level + 1e-3f;  // grid.h:242
	const Level<D> lv = make_level<D>(meta, level);

	if (kind == SLICE_BUCKET) {
		if (bplan.packed_owner) return;  // k_grid_bucket_owner runs them
		bucket_level<D, F>(meta, lv, level, plan.slot[item], slice, chunk, bplan, counters, queues, overflow, grid_gradient, accumulate != 0, lds_raw);
		return;
	}
	if (kind == SLICE_GLOBAL_ATOMIC) {
		// Dense-indexed level too large for the fixed-point path: its corners are memory-adjacent, so a float
		// slice would see all-or-nothing samples (8 serial iterations at 1/16 lane occupancy).  The memory-side
		// atomic units are otherwise idle during this launch: send this level's updates there (tile = slice).
		if (level_off) return;
		half_t* __restrict__ grad = grid_gradient + (size_t)meta.offset[level] * F;
		const uint32_t per_tile = div_round_up(io.n, n_slices);
		const uint32_t begin = slice * per_tile, end = min(begin + per_tile, io.n);
		for (uint32_t i = begin + threadIdx.x; i < end; i += SLICED_THREADS) {
			const Cell<D> c = make_cell<D, false>(lv, io, i);
			half_t g[F];
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) g[f] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
			const uint32_t n_corners = lv.nearest ? 1u : (1u << D);
			for (uint32_t idx = 0; idx < n_corners; ++idx) {
				const half_t wh = lv.nearest ? (half_t)1.0f : to_half_rn(corner_weight<D>(c, idx));
				const uint32_t index = corner_index<D, false>(lv, c, idx);
				if constexpr (F == 1) {
					// a packed atomic on the aligned pair; the partner half gets +0
					const h2 v = (index & 1u) ? h2{(half_t)0.0f, wh * g[0]} : h2{wh * g[0], (half_t)0.0f};
					atomic_add_h2(grad + (index & ~1u), v);
				} else {
					const h2 w2 = h2{wh, wh};
#pragma unroll
					for (uint32_t p = 0; p < F / 2; ++p) atomic_add_h2(grad + (size_t)index * F + 2 * p, w2 * h2{g[2 * p], g[2 * p + 1]});
				}
			}
		}
		return;
	}

	// equal slices of this level's table (level sizes are multiples of 8; the host sized n_slices to fit LDS)
	const uint32_t entries_per_slice = next_multiple(div_round_up(lv.hashmap_size, n_slices), 8u);
	if (kind == SLICE_FIXED64) {
		sliced_level<D, F, Acc::FIX64>(meta, io, lv, level, slice, chunk, n_chunks, entries_per_slice, dL_dy, grid_gradient, accumulate != 0,
		                               level_off, lds_raw);
	} else {
		sliced_level<D, F, PACKED ? Acc::PK16 : Acc::F32>(meta, io, lv, level, slice, chunk, n_chunks, entries_per_slice, dL_dy, grid_gradient,
		                                                   accumulate != 0, level_off, lds_raw);
	}
}

template <typename GRAD_T>
__global__ void k_grid_backward_input(uint32_t n_dims, uint32_t n_features, GridIO io, const GRAD_T* __restrict__ dL_dy,
                                      const float* __restrict__ dy_dx, float* __restrict__ dL_dx, uint32_t dx_stride_i,
                                      uint32_t dx_stride_d) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	float result[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (uint32_t k = 0; k < n_features; ++k) {
		const float dl = (float)dL_dy[(size_t)k * io.stride_k + (size_t)i * io.stride_i];
		for (uint32_t d = 0; d < n_dims; ++d) {
			const float t = dl * dy_dx[((size_t)k * io.n + i) * n_dims + d];
			result[d] = result[d] + t;
		}
	}
	for (uint32_t d = 0; d < n_dims; ++d) dL_dx[(size_t)i * dx_stride_i + (size_t)d * dx_stride_d] = result[d];
}

// second order w.r.t. dL_dy: dL_ddLdy[k][i] = sum_d dy_dx[k][i][d] * ddx[i][d]  (grid.h:623-653)
__global__ void k_grid_backward_backward_dLdoutput(uint32_t n_dims, uint32_t n_features, uint32_t n_to_pad, GridIO io, const float* __restrict__ dy_dx,
                                                   half_t* __restrict__ dL_ddLdy) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	float dd[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (uint32_t d = 0; d < n_dims; ++d) dd[d] = io.ddx[(size_t)i * io.ddx_stride_i + (size_t)d * io.ddx_stride_d];
	for (uint32_t k = 0; k < n_features; ++k) {
		float result = 0.0f;
		for (uint32_t d = 0; d < n_dims; ++d) result += dy_dx[((size_t)k * io.n + i) * n_dims + d] * dd[d];
		dL_ddLdy[(size_t)k * io.stride_k + (size_t)i * io.stride_i] = to_half_rn(result);
	}
	for (uint32_t k = n_features; k < n_features + n_to_pad; ++k) dL_ddLdy[(size_t)k * io.stride_k + (size_t)i * io.stride_i] = (half_t)0.0f;
}

// second order w.r.t. the positions (grid.h:457-620).  With v(corner) = sum_f grid[corner][f] * dL_dy[f] and s_d = +-1
// (right / left corner along d):
//   dL_dx[a] = scale^2 * ( ddx[a] * pos''(a) * sum_c s_a prod_{e != a} w_e v(c)                           (Smoothstep only)
//                        + sum_{b != a} ddx[b] * pos'(b) * pos'(a) * sum_c s_a s_b prod_{e != a, b} w_e v(c) )
// summed over the levels; one thread per sample walks the levels (no atomics, fixed order).
template <uint32_t D, uint32_t F>
__global__ void __launch_bounds__(128) k_grid_backward_backward_input(const GridMeta meta, const GridIO io, const half_t* __restrict__ dL_dy,
                                                                       const half_t* __restrict__ params, float* __restrict__ dL_dx,
                                                                       uint32_t dx_stride_i, uint32_t dx_stride_d) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	constexpr uint32_t N_CORNERS = 1u << D, NP = (F + 1) / 2;
	float x[D], dd[D], out[D];
	load_position<D>(io, i, x);
	load_ddx<D>(io, i, dd);
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) out[d] = 0.0f;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	const bool nearest = meta.interp == (uint32_t)InterpolationType::Nearest;
	for (uint32_t level = 0; level < meta.n_levels && !nearest; ++level) {
		if ((float)level > max_level + 1e-3f) break;  // grid.h:483
		const Level<D> lv = make_level<D>(meta, level);
		const half_t* __restrict__ grid = params + (size_t)meta.offset[level] * F;
		const Cell<D> c = make_cell<D, false>(lv, x);
		float d2[D];  // pos''(d): 0 for Linear, smoothstep'' = 6 - 12 t otherwise (common_device.h:1012-1014)
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) {
			float p = __builtin_fmaf(lv.scale, x[d], 0.5f);
			p -= __builtin_floorf(p);
			d2[d] = lv.smooth ? 6.0f - 12.0f * p : 0.0f;
		}
		float gy[F];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) gy[f] = (float)dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];
		float v[N_CORNERS];
#pragma unroll
		for (uint32_t idx = 0; idx < N_CORNERS; ++idx) {
			h2 val[NP];
			load_features<F>(grid + (size_t)corner_index<D, false>(lv, c, idx) * F, val);
			float acc = 0.0f;
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) acc += (float)val[f / 2][f % 2] * gy[f];
			v[idx] = acc;
		}
		const float s2 = lv.scale * lv.scale;
#pragma unroll
		for (uint32_t a = 0; a < D; ++a) {
			float grad_out = 0.0f;
#pragma unroll
			for (uint32_t idx = 0; idx < N_CORNERS; ++idx) {
				const float sa = ((idx >> a) & 1u) ? 1.0f : -1.0f;
				if (lv.smooth) {  // diagonal of the Hessian
					float wgt = s2 * dd[a] * d2[a] * sa;
#pragma unroll
					for (uint32_t e = 0; e < D; ++e) {
						if (e != a) wgt *= ((idx >> e) & 1u) ? c.w[e][1] : c.w[e][0];
					}
					grad_out += wgt * v[idx];
				}
#pragma unroll
				for (uint32_t b = 0; b < D; ++b) {  // mixed terms
					if (b == a) continue;
					float wgt = s2 * dd[b] * c.derivative[b] * c.derivative[a] * sa * (((idx >> b) & 1u) ? 1.0f : -1.0f);
#pragma unroll
					for (uint32_t e = 0; e < D; ++e) {
						if (e != a && e != b) wgt *= ((idx >> e) & 1u) ? c.w[e][1] : c.w[e][0];
					}
					grad_out += wgt * v[idx];
				}
			}
			out[a] += grad_out;
		}
	}
#pragma unroll
	for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * dx_stride_i + (size_t)d * dx_stride_d] = out[d];
}

template <uint32_t D>
__global__ void k_grid_indices(const GridMeta meta, const GridIO io, uint32_t* __restrict__ indices) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	for (uint32_t level = 0; level < meta.n_levels; ++level) {
		const Level<D> lv = make_level<D>(meta, level);
		for (uint32_t idx = 0; idx < (1u << D); ++idx) {
			uint32_t index;
			if (lv.fast) {
				index = corner_index<D, true>(lv, make_cell<D, true>(lv, io, i), idx);
			} else {
				index = corner_index<D, false>(lv, make_cell<D, false>(lv, io, i), idx);
			}
			indices[((size_t)i * meta.n_levels + level) * (1u << D) + idx] = index;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------

#define TCNN_GRID_DISPATCH_F(D_, MACRO)                                                             \
	switch (meta.n_feat) {                                                                          \
		case 1: MACRO(D_, 1); break;                                                                \
		case 2: MACRO(D_, 2); break;                                                                \
		case 4: MACRO(D_, 4); break;                                                                \
		case 8: MACRO(D_, 8); break;                                                                \
		default: throw std::runtime_error("GridEncoding: n_features_per_level must be 1, 2, 4, or 8."); \
	}

#define TCNN_GRID_DISPATCH(MACRO)                                                                \
	switch (meta.n_dims) {                                                                       \
		case 2: TCNN_GRID_DISPATCH_F(2, MACRO); break;                                           \
		case 3: TCNN_GRID_DISPATCH_F(3, MACRO); break;                                           \
		case 4: TCNN_GRID_DISPATCH_F(4, MACRO); break;                                           \
		default: throw std::runtime_error("GridEncoding: number of input dims must be 2, 3 or 4."); \
	}

// Cuts the (level, tile) items, level-major, into 8 runs of equal cost.  Cost of an item (measured per kind of level,
// profiles/r02_exp_forward.txt): 4 for a level whose table fits a CU's 32 KiB L1 next to the streaming traffic
// (<= 24 KiB), 8 for a hashed level (one L2 line per corner pair), 11 for a larger densely indexed level, each times
// (1 + 1.5 x the share of fetches that miss the L2) for tables beyond the L2 (T = 2^22: 16 MiB per level, 2.4x the
// time of a 2 MiB level).  Falls back to uniform costs if a run would need more than FWD_MAX_SEGMENTS segments.
static ForwardPlan make_forward_plan(const GridMeta& meta, uint32_t n, uint32_t tile_samples) {
	for (int uniform = 0; uniform < 2; ++uniform) {
		ForwardPlan plan = {};
		plan.tiles = div_round_up(n, tile_samples);
		uint64_t total = 0;
		uint32_t cost[MAX_N_LEVELS], n_regions[MAX_N_LEVELS];
		for (uint32_t l = 0; l < meta.n_levels; ++l) {
			const size_t table_bytes = (size_t)(meta.offset[l + 1] - meta.offset[l]) * meta.n_feat * sizeof(half_t);
			const uint32_t entries = meta.offset[l + 1] - meta.offset[l];
			// What a (level, 512-sample tile) item costs, from per-workgroup clock stamps of this kernel on the headline and the T = 2^22 stress shape
			// (scripts/exp_forward_stamps.*, profiles/r04_exp_notes.txt sections 18 and 20; microseconds of workgroup life, halved):
			//   tables that fit the L1 (<= 24 KiB)                                   4
			//   hashed levels: no locality at all                                    8 while the table fits the L2, x (1 + 2.5 miss) beyond it (16 MiB: 24)
			//   dense levels: grid_index's stride arithmetic, but the corners of a   4.6 + 1.15 log2(table bytes / 24 KiB), the table capped at the L2's
			//   cell are neighbours in y and z too                                   3 MiB, x (1 + 0.8 miss) beyond it  (55 KiB: 6, 1 MiB: 11, 7 MiB: 21)
			// miss = share of the line fetches that miss the XCD's 4 MiB L2 (about 3 MiB of it hold the table while outputs stream through).
			// Round 2's weights (dense 11 whatever the size, told from hashed by "no power-of-two size"; miss x 1.5 for both kinds) had the XCDs that
			// hold the dense levels finish 8 us late on the headline (its dense levels ARE powers of two) and 40-87 us EARLY on the stress shape.
			uint64_t dense_entries = 1;  // resolution^D, saturated
			for (uint32_t d = 0; d < meta.n_dims; ++d) dense_entries = std::min<uint64_t>(dense_entries * meta.resolution[l], 1ull << 40);
			const bool hashed = meta.grid_type == (uint32_t)GridType::Hash && (uint64_t)entries < dense_entries;
			const double miss = std::max(0.0, 1.0 - 3.0 * 1048576.0 / (double)table_bytes);
			const double in_l2 = std::min((double)table_bytes, 3.0 * 1048576.0);
			const double base = table_bytes <= 24u * 1024u ? 4.0
			                    : hashed ? 8.0 * (1.0 + 2.5 * miss)
			                             : (4.6 + 1.15 * std::log2(in_l2 / (24.0 * 1024.0))) * (1.0 + 0.8 * miss);
			cost[l] = uniform ? 16u : (uint32_t)(4.0 * base + 0.5);  // (quarter-microsecond units: the cuts fall on whole tiles)
			n_regions[l] = 1u;
			if (EXP_FWD_REGION_LOG2 != 0u && hashed && level_is_fast(meta, l) && entries > (1u << EXP_FWD_REGION_LOG2)) {
				// region-pass timing build (exp_diag.h): the level is walked once per 2^EXP_FWD_REGION_LOG2-entry slice of its table, every pass
				// fetching only the corners inside its slice -- priced like a pass over an L2-resident table with a fraction of the lanes active
				n_regions[l] = entries >> EXP_FWD_REGION_LOG2;
				cost[l] = (uint32_t)(4.0 * EXP_FWD_REGION_COST + 0.5);
			}
			total += (uint64_t)cost[l] * plan.tiles * n_regions[l];
		}
		bool ok = true;
		uint64_t done = 0;  // cost of the items already assigned
		uint32_t xcd = 0;
		for (uint32_t l = 0; l < meta.n_levels && ok; ++l) {
			for (uint32_t region = 0; region < n_regions[l] && ok; ++region) {
				uint32_t t = 0;
				while (t < plan.tiles) {
					// XCD `xcd` takes items while the cost assigned so far stays below its cumulative share
					const uint64_t limit = (total * (xcd + 1) + 7) / 8;
					uint32_t take = (uint32_t)std::min<uint64_t>(plan.tiles - t, (limit - done + cost[l] - 1) / cost[l]);
					if (xcd == 7) take = plan.tiles - t;
					if (take > 0) {
						uint32_t& ns = plan.n_segments[xcd];
						if (ns == FWD_MAX_SEGMENTS) {
							ok = false;
							break;
						}
						plan.segments[xcd][ns++] = {l, t, t + take, meta.offset[l + 1] - meta.offset[l], meta.resolution[l], __builtin_bit_cast(uint32_t, meta.scale[l]),
						                            meta.offset[l], (level_is_fast(meta, l) ? 1u : 0u) | (EXP_FWD_REGION_LOG2 != 0u ? (region << 8) | (n_regions[l] << 16) : 0u)};
						t += take;
						done += (uint64_t)take * cost[l];
					}
					if (done >= limit && xcd < 7) ++xcd;
				}
			}
		}
		if (ok) return plan;
	}
	throw std::runtime_error("grid_forward: could not build the work plan");
}

template <uint32_t D, uint32_t F, uint32_t SPT>
static void launch_forward_tiles(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* params, half_t* out) {
	// (Gathering the levels whose table fits a CU's LDS out of LDS -- one launch per level, the table copied in by every workgroup -- was
	// built in round 3 and measured slower in every arrangement: scripts/exp_grid_forward_lds.patch, profiles/r03_exp_notes.txt.)
	const ForwardPlan plan = make_forward_plan(meta, io.n, GRID_THREADS * SPT);
	uint32_t slots = 0;
	for (uint32_t x = 0; x < 8; ++x) {
		uint32_t n = 0;
		for (uint32_t k = 0; k < plan.n_segments[x]; ++k) n += plan.segments[x][k].tile_end - plan.segments[x][k].tile_begin;
		slots = std::max(slots, n);
	}
	TCNN_LAUNCH((k_grid_forward_tiles<D, F, SPT>), dim3(8u * slots), dim3(GRID_THREADS), 0, stream, meta, io, plan, params, out);
}

void grid_forward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* params, half_t* out, float* dy_dx) {
	if (io.n == 0) return;
	if (!dy_dx && out) {
#ifndef TCNN_FWD_SPT
#define TCNN_FWD_SPT 2  // samples per thread of the tiled gather (a workgroup: 256 x SPT samples of one level)
#endif
#define FWD_TILES(D_, F_) launch_forward_tiles<D_, F_, TCNN_FWD_SPT>(stream, meta, io, params, out);
		TCNN_GRID_DISPATCH(FWD_TILES)
#undef FWD_TILES
		return;
	}
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
#define FWD(D_, F_)                                                                                                                        \
	if (dy_dx) {                                                                                                                           \
		TCNN_LAUNCH((k_grid_forward<D_, F_, true>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, params, out, dy_dx);            \
	} else {                                                                                                                               \
		TCNN_LAUNCH((k_grid_forward<D_, F_, false>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, params, out, (float*)nullptr); \
	}
	TCNN_GRID_DISPATCH(FWD)
#undef FWD
}

static void grid_backward_atomic(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient,
                                 bool accumulate) {
	const size_t n_params = (size_t)meta.offset[meta.n_levels] * meta.n_feat;
	if (!accumulate) {  // grid.h:865-867
		if (hipMemsetAsync(grid_gradient, 0, n_params * sizeof(half_t), stream) != hipSuccess) throw std::runtime_error("grid_backward: memset failed");
	}
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
#define BWD(D_, F_) TCNN_LAUNCH((k_grid_backward_atomic<D_, F_>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, dL_dy, grid_gradient);
	TCNN_GRID_DISPATCH(BWD)
#undef BWD
}

unsigned long long grid_owner_wide_slices() {
#if defined(TCNN_HOST_EMU)
	return owner_slice_stats[1];
#else
	unsigned long long v = 0;
	if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_owner_wide_slices), sizeof(v)) != hipSuccess) throw std::runtime_error("grid_owner_wide_slices: could not read the counter");
	return v;
#endif
}

int& grid_owner_mode() {
	static int mode = 0;
	return mode;
}

// Host-side plan of one sliced / bucketed launch sequence.
struct BackwardPlan {
	SlicePlan slices = {};
	BucketPlan buckets = {};
	uint32_t lds_slice_bytes = 0, blocks = 0;
	std::vector<uint32_t> n_chunks;  // per item
	// workspace layout (bytes from its start)
	size_t n_counters = 0, overflow_offset = 0, workspace_bytes = 0;  // queues at offset 0 of the workspace
};

static BackwardPlan make_backward_plan(const GridMeta& meta, uint32_t n, bool packed, bool bucketed, bool accumulate, uint32_t lds_slice_bytes) {
	const uint32_t F = meta.n_feat;
	packed = packed && (F % 2 == 0);
	if (lds_slice_bytes == 0 || lds_slice_bytes > SLICED_LDS_MAX_BYTES) lds_slice_bytes = SLICED_LDS_BYTES;
	const uint32_t float_entry_bytes = F * (packed ? (uint32_t)sizeof(half_t) : (uint32_t)sizeof(float));
	const uint32_t fixed_entry_bytes = F * (uint32_t)sizeof(unsigned long long);
	lds_slice_bytes = std::max(lds_slice_bytes / fixed_entry_bytes, 8u) * fixed_entry_bytes;
	const uint32_t cap_fixed = lds_slice_bytes / fixed_entry_bytes, cap_float = lds_slice_bytes / float_entry_bytes;  // entries per slice
	constexpr uint32_t MAX_BASES[11] = {0x0, 0xFFFFFFFF, 0xFFFF, 0x659, 0xFF, 0x54, 0x28, 0x17, 0xF, 0xB, 0x9};
	uint32_t bucket_shift = 0;  // buckets hold a power-of-two number of entries (bucket = index >> shift)
	while ((2u << bucket_shift) <= cap_fixed) ++bucket_shift;
	const uint32_t n_corners = meta.interp == (uint32_t)InterpolationType::Nearest ? 1u : (1u << meta.n_dims);
	const uint32_t record_words = 1u + (F + 1u) / 2u;

	BackwardPlan bp;
	bp.lds_slice_bytes = lds_slice_bytes;
	BucketPlan& bk = bp.buckets;
	bk.shift = bucket_shift;
	bk.tiles = div_round_up(n, bucket_spt(meta.n_dims, F) * BUCKET_THREADS);
	uint32_t n_counters = 0, n_zero_blocks = 0;
	uint64_t n_queue_records = 0, n_records = 0;

	// Per level: accumulator kind by expected LDS-atomic density (see the comments above the kernels).
	struct Item {
		uint32_t level, kind, n_slices, n_chunks, slot;
	};
	std::vector<Item> items;
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		const uint32_t entries = meta.offset[l + 1] - meta.offset[l];
		uint64_t dense = 1;  // resolution^D, saturated
		for (uint32_t d = 0; d < meta.n_dims; ++d) dense = meta.resolution[l] <= MAX_BASES[meta.n_dims] ? dense * meta.resolution[l] : ~0ull >> 1;
		const bool hashed = meta.grid_type == (uint32_t)GridType::Hash && (uint64_t)entries < dense;
		const uint32_t n_fixed = div_round_up(entries, cap_fixed);
		const uint32_t n_buckets = div_round_up(entries, 1u << bucket_shift);
		Item it = {l, SLICE_FIXED64, n_fixed, 1u, 0u};
		if (bucketed && bk.n_levels < MAX_BUCKET_LEVELS && n_buckets <= MAX_BUCKETS_PER_LEVEL && entries <= (1u << PAIR_INDEX_BITS)) {
			// corners are derived once, binned by (table slice, sample chunk), accumulated by the queue's owner.
			// Large tables: one owner per slice (plain stores).  Small tables have few slices: the samples are also
			// split so that an owner sees ~32 Ki records; the owners of a slice then combine with packed-half atomics.
			const uint32_t j = bk.n_levels++;
			const uint64_t level_records = (uint64_t)n * n_corners;
			const uint64_t per_bucket = level_records / n_buckets;
			uint32_t n_chunks = per_bucket <= 65536 ? 1u : (uint32_t)std::min<uint64_t>(div_round_up<uint64_t>(per_bucket, 32768), bk.tiles);
			const uint32_t tiles_per_chunk = div_round_up(bk.tiles, n_chunks);
			n_chunks = div_round_up(bk.tiles, tiles_per_chunk);
			const uint64_t level_pairs = (uint64_t)n * std::max(1u, n_corners / 2u);  // queue unit: a pair of records
			const uint64_t expected = level_pairs / ((uint64_t)n_buckets * n_chunks);
			const uint64_t capacity = next_multiple<uint64_t>(2 * expected + 512, 64);
			// (queue positions are multiplied with 24-bit multiplies in the owner pass; the 2^32 records checked below come first for every
			// table with more than a few buckets, and small tables are chunked to ~32 Ki records per queue)
			if (capacity >= (1ull << 24)) throw std::runtime_error("grid_backward: batch too large for the bucketed backward");
			bk.level[j] = (uint8_t)l;
			bk.n_buckets[j] = n_buckets;
			bk.n_chunks[j] = n_chunks;
			bk.tiles_per_chunk[j] = tiles_per_chunk;
			bk.capacity[j] = (uint32_t)capacity;
			bk.counter_base[j] = n_counters;
			bk.queue_base[j] = n_queue_records;
			bk.zero_block_begin[j] = n_zero_blocks;
			if (n_chunks > 1 && !accumulate) n_zero_blocks += div_round_up(entries * F, ZERO_BLOCK_HALVES);
			n_counters += n_buckets * n_chunks;
			n_queue_records += capacity * n_buckets * n_chunks;
			n_records += level_records;
			it.kind = SLICE_BUCKET;
			it.n_slices = n_buckets;
			it.n_chunks = n_chunks;
			it.slot = j;
		} else if (n_fixed <= 8) {
			// small table: every corner of every sample hits the slice(s) -> dense atomics -> fixed point;
			// <= 4 slices also split the SAMPLES over up to 16 workgroups (few flush atomics)
			if (n_fixed <= 4) it.n_chunks = std::max(1u, std::min(16u / n_fixed, div_round_up(n, 2048u)));
		} else if (hashed) {
			// hashed level: corners scatter over the table -> >= 16 float slices see <= 1/16 of them (sparse atomics)
			it.kind = SLICE_FLOAT;
			it.n_slices = std::max(16u, div_round_up(entries, cap_float));
		} else {
			it.kind = SLICE_GLOBAL_ATOMIC;
			it.n_slices = std::max(1u, div_round_up(n, SLICED_THREADS * 4u));  // sample tiles
		}
		items.push_back(it);
	}
	if (n_records > 0xFFFFFFFFull) throw std::runtime_error("grid_backward: batch too large for the bucketed backward");
	// persistent scatter workgroups (four fit a CU's LDS at a time; twice that many are launched)
	bk.wgs_per_level = bk.n_levels ? std::max(1u, std::min(bk.tiles, div_round_up(BUCKET_RESIDENT_WGS, bk.n_levels))) : 1u;
	bk.scatter_blocks = bk.n_levels * bk.wgs_per_level;
	bk.zero_block_begin[bk.n_levels] = n_zero_blocks;
	bk.overflow_counter = n_counters;
	bk.overflow_capacity = (uint32_t)n_records;
	bk.level_sum_base = (n_counters + 2u + 1u) & ~1u;
	bp.n_counters = bk.n_levels ? bk.level_sum_base + 2u * LEVEL_SUM_PARTS * MAX_BUCKET_LEVELS : 0;
	bp.overflow_offset = next_multiple<size_t>(n_queue_records * (2 * record_words - 1) * sizeof(uint32_t), 256);  // n_queue_records counts pairs
	bp.workspace_bytes = bk.n_levels ? bp.overflow_offset + next_multiple<size_t>(n_records * (record_words + 1) * sizeof(uint32_t), 256) : 0;

	// long passes first (bucket owners, float slices), the short work (fixed-point chunks, atomic tiles) fills the tail
	auto is_long = [](const Item& it) { return it.kind == SLICE_FLOAT || it.kind == SLICE_BUCKET; };
	std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) { return is_long(a) > is_long(b); });

	SlicePlan& plan = bp.slices;
	plan.n_items = (uint32_t)items.size();
	uint32_t blocks = 0;
	for (uint32_t p = 0; p < plan.n_items; ++p) {
		const Item& it = items[p];
		plan.block_begin[p] = blocks;
		plan.n_slices[p] = it.n_slices;
		plan.level[p] = (uint8_t)it.level;
		plan.kind[p] = (uint8_t)it.kind;
		plan.slot[p] = (uint8_t)it.slot;
		bp.n_chunks.push_back(it.n_chunks);
		blocks += it.n_slices * it.n_chunks;
		if (it.kind == SLICE_BUCKET) bp.buckets.n_owner_blocks += it.n_slices * it.n_chunks;
	}
	plan.block_begin[plan.n_items] = blocks;
	uint32_t widest = 1;
	for (uint32_t p = 0; p < plan.n_items; ++p) widest = std::max(widest, plan.block_begin[p + 1] - plan.block_begin[p]);
	if ((uint64_t)plan.n_items * widest * 10 <= (uint64_t)blocks * 11) {  // <= 10 % padding: index by arithmetic
		plan.blocks_per_item = widest;
		bp.blocks = plan.n_items * widest;
	} else {
		plan.blocks_per_item = 0;
		bp.blocks = blocks;
	}
	return bp;
}

GridBackwardWorkspace grid_backward_workspace_size(const GridMeta& meta, uint32_t n, GridBackwardMode mode, uint32_t lds_slice_bytes) {
	GridBackwardWorkspace ws;
	if (mode != GridBackwardMode::Bucketed || n == 0) return ws;
	const BackwardPlan bp = make_backward_plan(meta, n, true, true, false, lds_slice_bytes);
	ws.scratch_bytes = bp.workspace_bytes;
	ws.n_counters = bp.n_counters;
	return ws;
}

static void grid_backward_sliced_launches(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient,
                                          bool accumulate, bool packed, bool bucketed, uint32_t lds_slice_bytes, const GridBackwardWorkspace& ws);

// The queue counters are handed back zeroed by the kernels themselves; if the launch sequence is cut short by an error
// they are cleared here, so that the contract ("zero on entry") survives for the next call.
static void grid_backward_sliced(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient,
                                 bool accumulate, bool packed, bool bucketed, uint32_t lds_slice_bytes, const GridBackwardWorkspace& ws) {
	try {
		grid_backward_sliced_launches(stream, meta, io, dL_dy, grid_gradient, accumulate, packed, bucketed, lds_slice_bytes, ws);
	} catch (...) {
		if (bucketed && ws.counters) (void)hipMemsetAsync(ws.counters, 0, ws.n_counters * sizeof(uint32_t), stream);
		throw;
	}
}

static void grid_backward_sliced_launches(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient,
                                          bool accumulate, bool packed, bool bucketed, uint32_t lds_slice_bytes, const GridBackwardWorkspace& ws) {
	const uint32_t F = meta.n_feat;
	packed = packed && (F % 2 == 0);
	const BackwardPlan bp = make_backward_plan(meta, io.n, packed, bucketed, accumulate, lds_slice_bytes);
	lds_slice_bytes = bp.lds_slice_bytes;
	const SlicePlan& plan = bp.slices;
	const BucketPlan& bk = bp.buckets;
	const uint32_t blocks = bp.blocks;
	if (io.ddx) {
		for (uint32_t p = 0; p < plan.n_items; ++p) {
			if (plan.kind[p] != SLICE_BUCKET) throw std::runtime_error("grid_backward: second-order scatter needs every level in the bucketed path (grid_backward() checks this)");
		}
	}
	uint32_t* counters = nullptr;
	uint32_t* queues = nullptr;
	uint32_t* overflow = nullptr;
	if (bk.n_levels) {
		if (!ws.scratch || ws.scratch_bytes < bp.workspace_bytes || !ws.counters || ws.n_counters < bp.n_counters) {
			throw std::runtime_error("grid_backward: workspace too small for the bucketed backward");
		}
		counters = ws.counters;  // zero on entry (contract); the kernels below leave them zeroed again
		queues = (uint32_t*)ws.scratch;
		overflow = (uint32_t*)((unsigned char*)ws.scratch + bp.overflow_offset);
	}
	if (bk.n_levels) {
		if (ws.phase_hook) ws.phase_hook(ws.hook_user, 0, 1);
		// pass A: derive every corner once, bin by owner (+ zero the gradients of chunked levels)
		const uint32_t scatter_blocks = bk.scatter_blocks + bk.zero_block_begin[bk.n_levels];
		uint32_t max_buckets = 0;
		for (uint32_t j = 0; j < bk.n_levels; ++j) max_buckets = std::max(max_buckets, bk.n_buckets[j]);
#define BSCATTER(D_, F_)                                                                                                                      \
	{                                                                                                                                         \
		const uint32_t lds = bucket_spt(D_, F_) * BUCKET_THREADS * ((1u << D_) / 2u) * BucketRecord<F_>::PAIR_WORDS * 4u +                          \
		                     (2u * max_buckets + WAVE + 4u) * 4u;                                                                             \
		if (io.ddx) {                                                                                                                         \
			TCNN_SET_MAX_DYN_LDS((k_grid_bucket_scatter<D_, F_, true>), lds);                                                                 \
			TCNN_LAUNCH((k_grid_bucket_scatter<D_, F_, true>), dim3(scatter_blocks), dim3(BUCKET_THREADS), lds, stream, meta, io, bk, dL_dy,  \
			            counters, queues, overflow, grid_gradient);                                                                           \
		} else {                                                                                                                              \
			TCNN_SET_MAX_DYN_LDS((k_grid_bucket_scatter<D_, F_, false>), lds);                                                                \
			TCNN_LAUNCH((k_grid_bucket_scatter<D_, F_, false>), dim3(scatter_blocks), dim3(BUCKET_THREADS), lds, stream, meta, io, bk, dL_dy, \
			            counters, queues, overflow, grid_gradient);                                                                           \
		}                                                                                                                                     \
	}
		TCNN_GRID_DISPATCH(BSCATTER)
#undef BSCATTER
		if (ws.phase_hook) ws.phase_hook(ws.hook_user, 0, 0);
	}
	if (ws.phase_hook) ws.phase_hook(ws.hook_user, 1, 1);
	for (uint32_t p = 0; p < plan.n_items; ++p) {
		struct { uint32_t level, kind, n_chunks; } it = {plan.level[p], plan.kind[p], bp.n_chunks[p]};
		if (it.kind != SLICE_BUCKET && (it.n_chunks > 1 || it.kind == SLICE_GLOBAL_ATOMIC) && !accumulate) {  // atomically updated levels start from zero
			const uint32_t entries = meta.offset[it.level + 1] - meta.offset[it.level];
			if (hipMemsetAsync(grid_gradient + (size_t)meta.offset[it.level] * F, 0, (size_t)entries * F * sizeof(half_t), stream) != hipSuccess) {
				throw std::runtime_error("grid_backward: memset failed");
			}
		}
	}
	const int acc = accumulate ? 1 : 0;
	// bucket items: the packed owner kernel (even F) unless grid_owner_mode() asks for the 64-bit-per-value form; mode "wide" runs the
	// packed kernel's own 64-bit redo on every slice (tests)
	const int owner_mode = grid_owner_mode();
	BucketPlan bk_launch = bk;
	bk_launch.packed_owner = (bk.n_levels && F % 2 == 0 && owner_mode != 1) ? 1u : 0u;
	bool other_items = false;
	for (uint32_t p = 0; p < plan.n_items; ++p) other_items = other_items || plan.kind[p] != SLICE_BUCKET;
	if (bk_launch.packed_owner) {
		const uint32_t owner_lds = std::max((1u << bk.shift) * F * 4u, 8u * F * 8u);
		const int force_wide = owner_mode == 2 ? 1 : 0;
		// one descriptor per bucket item, in plan order; the grid is (workgroups of the largest item) x (items)
		OwnerItems owner_items = {};
		uint32_t n_owner_items = 0, owner_width = 0;
		for (uint32_t p = 0; p < plan.n_items; ++p) {
			if (plan.kind[p] != SLICE_BUCKET) continue;
			const uint32_t l = plan.level[p], j = plan.slot[p];
			OwnerItem& it = owner_items.item[n_owner_items++];
			it.level = l;
			it.slot = j;
			it.n_slices = plan.n_slices[p];
			it.n_blocks = plan.block_begin[p + 1] - plan.block_begin[p];
			it.hashmap_size = meta.offset[l + 1] - meta.offset[l];
			it.fast = level_is_fast(meta, l) ? 1u : 0u;
			it.offset = meta.offset[l];
			it.capacity = bk.capacity[j];
			it.n_chunks = bk.n_chunks[j];
			it.n_buckets = bk.n_buckets[j];
			it.counter_base = bk.counter_base[j];
			it.queue_base = bk.queue_base[j];
			owner_width = std::max(owner_width, it.n_blocks);
		}
#define BOWNER(D_, F_)                                                                                                                                  \
	if constexpr (F_ % 2 == 0) {                                                                                                                        \
		TCNN_SET_MAX_DYN_LDS((k_grid_bucket_owner<D_, F_>), owner_lds);                                                                                 \
		TCNN_LAUNCH((k_grid_bucket_owner<D_, F_>), dim3(owner_width, n_owner_items), dim3(OWNER_THREADS), owner_lds, stream, meta, owner_items, acc, bk.shift, \
		            bk.overflow_counter, bk.overflow_capacity, bk.n_owner_blocks, bk.level_sum_base, bk.n_levels, counters, (const uint32_t*)queues,    \
		            (const uint32_t*)overflow, grid_gradient, owner_lds, force_wide);                                                                   \
	}
		TCNN_GRID_DISPATCH(BOWNER)
#undef BOWNER
	}
#define BWDS(D_, F_)                                                                                                                   \
	if (packed) {                                                                                                                      \
		if constexpr (F_ % 2 == 0) {                                                                                                   \
			TCNN_SET_MAX_DYN_LDS((k_grid_backward_sliced<D_, F_, true>), lds_slice_bytes);                                             \
			TCNN_LAUNCH((k_grid_backward_sliced<D_, F_, true>), dim3(blocks), dim3(SLICED_THREADS), lds_slice_bytes, stream, meta, io, \
			            plan, dL_dy, grid_gradient, acc, bk_launch, counters, (const uint32_t*)queues, (const uint32_t*)overflow); \
		}                                                                                                                              \
	} else {                                                                                                                           \
		TCNN_SET_MAX_DYN_LDS((k_grid_backward_sliced<D_, F_, false>), lds_slice_bytes);                                                \
		TCNN_LAUNCH((k_grid_backward_sliced<D_, F_, false>), dim3(blocks), dim3(SLICED_THREADS), lds_slice_bytes, stream, meta, io,    \
		            plan, dL_dy, grid_gradient, acc, bk_launch, counters, (const uint32_t*)queues, (const uint32_t*)overflow);     \
	}
	if (!bk_launch.packed_owner || other_items) {
		TCNN_GRID_DISPATCH(BWDS)
	}
#undef BWDS
	if (ws.phase_hook) ws.phase_hook(ws.hook_user, 1, 0);
}

void grid_backward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient, bool accumulate,
                   GridBackwardMode mode, uint32_t lds_slice_bytes, const GridBackwardWorkspace& ws) {
	if (io.n == 0) return;
	if (!grid_gradient) throw std::runtime_error("grid_backward: missing gradient buffer");
	if (io.ddx) {  // second-order scatter
		if (meta.interp == (uint32_t)InterpolationType::Nearest) {  // d(dy_dx)/d(grid) == 0 without interpolation (grid.h:422-425)
			const size_t bytes = (size_t)meta.offset[meta.n_levels] * meta.n_feat * sizeof(half_t);
			if (!accumulate && hipMemsetAsync(grid_gradient, 0, bytes, stream) != hipSuccess) throw std::runtime_error("grid_backward: memset failed");
			return;
		}
		if (mode != GridBackwardMode::Bucketed) throw std::runtime_error("grid_backward: the second-order scatter runs in the bucketed mode only");
		// more than 32 levels, or a level beyond 4096 buckets of the chosen slice size: the second-order weight through the
		// reference's formulation (global atomics) instead
		const BackwardPlan bp = make_backward_plan(meta, io.n, meta.n_feat % 2 == 0, true, accumulate, lds_slice_bytes);
		for (uint32_t p = 0; p < bp.slices.n_items; ++p) {
			if (bp.slices.kind[p] != SLICE_BUCKET) mode = GridBackwardMode::Atomic;
		}
	}
	// stochastic interpolation (one unweighted update per sample and level): the reference's atomic form; the owner-computes
	// passes are built around all 2^D weighted corners
	if (meta.stochastic != 0u && !io.ddx) mode = GridBackwardMode::Atomic;
	switch (mode) {
		case GridBackwardMode::SlicedF32: grid_backward_sliced(stream, meta, io, dL_dy, grid_gradient, accumulate, false, false, lds_slice_bytes, ws); break;
		case GridBackwardMode::SlicedF16: grid_backward_sliced(stream, meta, io, dL_dy, grid_gradient, accumulate, true, false, lds_slice_bytes, ws); break;
		case GridBackwardMode::Atomic:
			if (ws.phase_hook) ws.phase_hook(ws.hook_user, 1, 1);
			grid_backward_atomic(stream, meta, io, dL_dy, grid_gradient, accumulate);
			if (ws.phase_hook) ws.phase_hook(ws.hook_user, 1, 0);
			break;
		case GridBackwardMode::Bucketed:
			grid_backward_sliced(stream, meta, io, dL_dy, grid_gradient, accumulate, true, true, lds_slice_bytes, ws);
			break;
	}
}

void grid_backward_input(hipStream_t stream, uint32_t n_dims, uint32_t n_features, const GridIO& io, const half_t* dL_dy,
                         const float* dy_dx, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_d) {
	if (io.n == 0) return;
	TCNN_LAUNCH(k_grid_backward_input<half_t>, dim3(div_round_up(io.n, 128u)), dim3(128), 0, stream, n_dims, n_features, io, dL_dy, dy_dx,
	            dL_dx, dx_stride_i, dx_stride_d);
}

// ---- fp32 encodings (GridEncodingTemplated<float>) ----
void grid_forward_f32(hipStream_t stream, const GridMeta& meta, const GridIO& io, const float* params, float* out, float* dy_dx) {
	if (io.n == 0) return;
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
#define FWD32(D_, F_)                                                                                                              \
	if (dy_dx) {                                                                                                                   \
		TCNN_LAUNCH((k_grid_forward_f32<D_, F_, true>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, params, out, dy_dx); \
	} else {                                                                                                                       \
		TCNN_LAUNCH((k_grid_forward_f32<D_, F_, false>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, params, out, (float*)nullptr); \
	}
	TCNN_GRID_DISPATCH(FWD32)
#undef FWD32
}
void grid_backward_f32(hipStream_t stream, const GridMeta& meta, const GridIO& io, const float* dL_dy, float* grid_gradient, bool accumulate) {
	if (io.n == 0) return;
	if (!grid_gradient) throw std::runtime_error("grid_backward: missing gradient buffer");
	const size_t n_params = (size_t)meta.offset[meta.n_levels] * meta.n_feat;
	if (!accumulate) {  // grid.h:865-867
		if (hipMemsetAsync(grid_gradient, 0, n_params * sizeof(float), stream) != hipSuccess) throw std::runtime_error("grid_backward: memset failed");
	}
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
#define BWD32(D_, F_) TCNN_LAUNCH((k_grid_backward_atomic_f32<D_, F_>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, dL_dy, grid_gradient);
	TCNN_GRID_DISPATCH(BWD32)
#undef BWD32
}
void grid_backward_input_f32(hipStream_t stream, uint32_t n_dims, uint32_t n_features, const GridIO& io, const float* dL_dy, const float* dy_dx, float* dL_dx,
                             uint32_t dx_stride_i, uint32_t dx_stride_d) {
	if (io.n == 0) return;
	TCNN_LAUNCH(k_grid_backward_input<float>, dim3(div_round_up(io.n, 128u)), dim3(128), 0, stream, n_dims, n_features, io, dL_dy, dy_dx, dL_dx, dx_stride_i,
	            dx_stride_d);
}

void grid_backward_backward_dLdoutput(hipStream_t stream, uint32_t n_dims, uint32_t n_features, uint32_t n_to_pad, const GridIO& io,
                                      const float* dy_dx, half_t* dL_ddLdy) {
	if (io.n == 0) return;
	if (!io.ddx || !dy_dx) throw std::runtime_error("grid second-order pass: dL_ddLdinput and the forward's dy_dx are required");
	TCNN_LAUNCH(k_grid_backward_backward_dLdoutput, dim3(div_round_up(io.n, 128u)), dim3(128), 0, stream, n_dims, n_features, n_to_pad, io, dy_dx, dL_ddLdy);
}

void grid_backward_backward_input(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, const half_t* params,
                                  float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_d) {
	if (io.n == 0) return;
	if (!io.ddx) throw std::runtime_error("grid second-order pass: dL_ddLdinput is required");
	const uint32_t blocks = div_round_up(io.n, 128u);
#define BBI(D_, F_) TCNN_LAUNCH((k_grid_backward_backward_input<D_, F_>), dim3(blocks), dim3(128), 0, stream, meta, io, dL_dy, params, dL_dx, dx_stride_i, dx_stride_d);
	TCNN_GRID_DISPATCH(BBI)
#undef BBI
}

void grid_indices(hipStream_t stream, const GridMeta& meta, const GridIO& io, uint32_t* indices) {
	if (io.n == 0) return;
	const uint32_t blocks = div_round_up(io.n, 128u);
	switch (meta.n_dims) {
		case 2: TCNN_LAUNCH((k_grid_indices<2>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		case 3: TCNN_LAUNCH((k_grid_indices<3>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		case 4: TCNN_LAUNCH((k_grid_indices<4>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		default: throw std::runtime_error("GridEncoding: number of input dims must be 2, 3 or 4.");
	}
}

}  // namespace tcnn_hip
