// mlp_kernels.hip -- see mlp_kernels.h for the structure and the reference lines restated.
//
// Fragment bookkeeping used throughout (lane l of a wave: lr = l & 15, g = l >> 4):
//   x32 operand: 8 halves, k = 32*kb + 8*g + j      x16 operand: 4 halves, k = k0 + 4*g + j
//   accumulator: element r <-> (row = 4*g + r, col = lr)
// LDS tiles:  "sample-major"  tile[s][ld]   (ld = width + 8 halves: 16-byte aligned, bank-skewed rows)
//             "feature-major" tile[k][SP]   (SP = S + 8; in k_mlp_backward SP = S + 16 with the 8-sample column blocks of row k
//                                            XOR-ed with (k >> 3) & 7, see ft_col())
#include "mlp_kernels.h"

#include <stdlib.h>

#include <algorithm>
#include <stdexcept>
#include <string>

namespace tcnn_hip {



constexpr uint32_t mlp_fwd_tile(uint32_t width) { return width == 128 ? 128u : 64u; }
constexpr uint32_t MLP_BWD_TILE = 64;
// Feature-major tiles of k_mlp_backward.  The saved activations arrive sample-major and are transposed on the way into LDS with
// 2-byte stores: lanes that hold consecutive 8-neuron groups of one sample write rows 8 * SP halves apart -- any SP that keeps
// the rows 16-byte aligned puts all of them on ONE bank (16-way conflicts at 128 neurons: 73 % of the kernel's LDS cycles,
// profiles/r02_exp_notes.txt).  Swizzle: the eight 8-sample column blocks of row k are XOR-ed with (k >> 3) & 7, so those lanes
// land in different blocks; vector accesses of 4 or 8 consecutive samples stay contiguous.  With SP = S + 16 the model of the
// hardware's lane groups (scripts/lds_bank_model.py) gives 2-4 cycles per transposing store (was 16-32) and conflict-free 8-byte
// operand reads.
constexpr uint32_t MLP_BWD_SP = MLP_BWD_TILE + 16;
TCNN_DEVICE uint32_t ft_col(uint32_t k, uint32_t sample) { return sample ^ (((k >> 3) & 7u) << 3); }
// The same for row 16 * r16 + lr and column c16 + c4 (c16 a multiple of 16, c4 < 16 a multiple of 4), split so that the part
// that depends on unrolled loop counters folds into an immediate offset and the lane part is computed once:
//   ft_col(16 r16 + lr, c16 + c4) == (c16 ^ 16 (r16 & 3)) + (c4 ^ 8 (lr >> 3))       (the two terms occupy disjoint bits)
TCNN_DEVICE uint32_t ft_col16(uint32_t r16, uint32_t c16) { return c16 ^ ((r16 & 3u) << 4); }
TCNN_DEVICE uint32_t ft_lane4(uint32_t lr, uint32_t c4) { return c4 ^ ((lr >> 3) << 3); }

// =============================================================================================
// forward / inference
// =============================================================================================
template <uint32_t WIDTH, bool SAVE, bool GENERAL>
__global__ void __launch_bounds__(WIDTH / 16 * 64) k_mlp_forward(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params,
                                                                  const half_t* __restrict__ input, half_t* __restrict__ hidden,
                                                                  half_t* __restrict__ output) {
	constexpr uint32_t NW = WIDTH / 16, THREADS = NW * 64, S = mlp_fwd_tile(WIDTH), NT = S / 16;
	TCNN_DYN_LDS(lds_raw);
	const uint32_t ld = (m.in_width > WIDTH ? m.in_width : WIDTH) + 8;
	half_t* buf0 = (half_t*)lds_raw;
	half_t* buf1 = buf0 + S * ld;

	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t act = m.activation, out_act = m.output_activation;
	const uint32_t n_tiles = n / S;

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		// ---- stage the feature-major input [in_width][n] into a sample-major LDS tile
		{
			const uint32_t n_chunks = m.in_width * (S / 8);
			for (uint32_t c = tid; c < n_chunks; c += THREADS) {
				const uint32_t k = c / (S / 8), cc = c % (S / 8);
				const h8 v = *(const h8*)(input + (size_t)k * n + (size_t)tile * S + 8 * cc);
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) buf0[(8 * cc + j) * ld + k] = v[j];
			}
		}
		__syncthreads();

		half_t* cur = buf0;
		half_t* nxt = buf1;
		const half_t* Wl = params;
		uint32_t K = m.in_width;
		for (uint32_t layer = 0; layer <= m.n_hidden_matmuls; ++layer) {
			f4 acc[NT];
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
			const half_t* wrow = Wl + (size_t)(16 * w + lr) * K;  // A operand: this wave's 16 weight rows
			for (uint32_t kb = 0; kb < K / 32; ++kb) {
				const h8 a = *(const h8*)(wrow + 32 * kb + 8 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h8 b = *(const h8*)(cur + (16 * t + lr) * ld + 32 * kb + 8 * g);
					acc[t] = mfma_16x16x32(a, b, acc[t]);
				}
			}
			if (K & 16u) {
				const uint32_t k0 = K & ~31u;
				const h4 a = *(const h4*)(wrow + k0 + 4 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 b = *(const h4*)(cur + (16 * t + lr) * ld + k0 + 4 * g);
					acc[t] = mfma_16x16x16(a, b, acc[t]);
				}
			}
			// accumulator (neuron 16w+4g+r, sample 16t+lr) -> activation -> sample-major store
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 o = act_forward4<GENERAL>(act, acc[t]);
				*(h4*)(nxt + (16 * t + lr) * ld + 16 * w + 4 * g) = o;
			}
			__syncthreads();
			if (SAVE) {  // post-activation hidden state, [layer][n][WIDTH] (fully_fused_mlp.cu:841-854)
				half_t* dst = hidden + ((size_t)layer * n + (size_t)tile * S) * WIDTH;
				for (uint32_t c = tid; c < S * WIDTH / 8; c += THREADS) {
					const uint32_t i = c / (WIDTH / 8), cc = c % (WIDTH / 8);
					*(h8*)(dst + (size_t)i * WIDTH + 8 * cc) = *(const h8*)(nxt + i * ld + 8 * cc);
				}
			}
			half_t* tmp = cur;
			cur = nxt;
			nxt = tmp;
			Wl += (size_t)WIDTH * K;
			K = WIDTH;
		}

		// ---- output layer: padded_out / 16 blocks of 16 outputs; (sample tile, output block) items are spread over the waves
		const uint32_t OUTP = m.padded_out;
		for (uint32_t item = w; item < NT * (OUTP / 16); item += NW) {
			const uint32_t t = item % NT, ob = item / NT;
			f4 acc = zero4();
			const half_t* wrow = Wl + (size_t)(16 * ob + lr) * WIDTH;
#pragma unroll
			for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
				const h8 a = *(const h8*)(wrow + 32 * kb + 8 * g);
				const h8 b = *(const h8*)(cur + (16 * t + lr) * ld + 32 * kb + 8 * g);
				acc = mfma_16x16x32(a, b, acc);
			}
			if constexpr (WIDTH % 32 != 0) {
				constexpr uint32_t k0 = WIDTH & ~31u;
				const h4 a = *(const h4*)(wrow + k0 + 4 * g);
				const h4 b = *(const h4*)(cur + (16 * t + lr) * ld + k0 + 4 * g);
				acc = mfma_16x16x16(a, b, acc);
			}
			const h4 o = act_forward4<GENERAL>(out_act, acc);
			*(h4*)(output + ((size_t)tile * S + 16 * t + lr) * OUTP + 16 * ob + 4 * g) = o;  // (output 16ob+4g+r, sample 16t+lr)
		}
		__syncthreads();
	}
}

// =============================================================================================
// weight transposition (tiny): params [out][in] row-major -> params_t [in][out] per matrix
// =============================================================================================
__global__ void k_mlp_transpose_weights(const MlpMeta m, const half_t* __restrict__ params, half_t* __restrict__ params_t) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m.n_params()) return;
	const uint32_t dst = mlp_transposed_index(m, i);
	params_t[dst] = params[i];
}

// =============================================================================================
// backward (activation gradients + weight gradients)
// =============================================================================================
template <uint32_t WIDTH, uint32_t HM, bool GENERAL>
__global__ void __launch_bounds__(WIDTH / 16 * 64) k_mlp_backward(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params_t,
                                                                   const half_t* __restrict__ input, const half_t* __restrict__ hidden,
                                                                   const half_t* __restrict__ dL_doutput, half_t* __restrict__ dL_dinput,
                                                                   float* __restrict__ partials) {
	constexpr uint32_t NW = WIDTH / 16, THREADS = NW * 64, S = MLP_BWD_TILE, NT = S / 16, NTP = S / 32;
	constexpr uint32_t SP = MLP_BWD_SP, LDW = WIDTH + 8, NB = WIDTH / 16, MAX_INB = MLP_MAX_IN_WIDTH / 16;
	static_assert(S == 64, "ft_col() swizzles the 8 blocks of 8 samples of a 64-sample tile");
	TCNN_DYN_LDS(lds_raw);
	const uint32_t IN = m.in_width, nb_in = IN / 16;
	half_t* xT = (half_t*)lds_raw;                 // [IN][SP]           network input, feature-major
	half_t* hT = xT + IN * SP;                     // [HM+1][WIDTH][SP]  forward activations, feature-major
	half_t* dact0 = hT + (HM + 1) * WIDTH * SP;    // [S][LDW]           dL/d(pre-activation), sample-major
	half_t* dact1 = dact0 + S * LDW;
	half_t* dyT = dact1 + S * LDW;                 // [16][SP]
	half_t* dxT = dyT + 16 * SP;                   // [IN][SP]

	const half_t* wt_in = params_t;                            // [IN][WIDTH]
	const half_t* wt_hid = wt_in + (size_t)IN * WIDTH;         // HM x [WIDTH][WIDTH]
	const half_t* wt_out = wt_hid + (size_t)HM * WIDTH * WIDTH;  // [WIDTH][16]

	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t lane_c4 = ft_lane4(lr, 4 * g);  // this lane's share of the swizzled column of its 8-byte feature-major reads
	const uint32_t act = m.activation;
	const bool want_grads = partials != nullptr, want_dx = dL_dinput != nullptr;
	const uint32_t n_tiles = n / S;

	// fp32 weight-gradient accumulators, live across all tiles of this workgroup
	f4 accI[MAX_INB];
	f4 accH[HM > 0 ? HM : 1][NB];
	f4 accO = zero4();
#pragma unroll
	for (uint32_t b = 0; b < MAX_INB; ++b) accI[b] = zero4();
#pragma unroll
	for (uint32_t j = 0; j < (HM > 0 ? HM : 1); ++j)
#pragma unroll
		for (uint32_t b = 0; b < NB; ++b) accH[j][b] = zero4();

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		// ---- A. stage tiles -----------------------------------------------------------------
		for (uint32_t c = tid; c < IN * (S / 8); c += THREADS) {  // input is already feature-major
			const uint32_t k = c / (S / 8), cc = c % (S / 8);
			*(h8*)(xT + k * SP + ft_col(k, 8 * cc)) = *(const h8*)(input + (size_t)k * n + (size_t)tile * S + 8 * cc);
		}
#pragma unroll
		for (uint32_t l = 0; l <= HM; ++l) {  // saved activations are sample-major: transpose on the way in
			const half_t* src = hidden + ((size_t)l * n + (size_t)tile * S) * WIDTH;
			for (uint32_t c = tid; c < S * WIDTH / 8; c += THREADS) {
				const uint32_t i = c / (WIDTH / 8), cc = c % (WIDTH / 8);
				const h8 v = *(const h8*)(src + (size_t)i * WIDTH + 8 * cc);
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) hT[(l * WIDTH + 8 * cc + j) * SP + ft_col(8 * cc + j, i)] = v[j];
			}
		}
		for (uint32_t c = tid; c < S * 2; c += THREADS) {
			const uint32_t i = c / 2, cc = c % 2;
			const h8 v = *(const h8*)(dL_doutput + ((size_t)tile * S + i) * 16 + 8 * cc);
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) dyT[(8 * cc + j) * SP + i] = v[j];
		}
		__syncthreads();

		// ---- B. output matrix:  dA_last[s][k] = sum_o dY[s][o] W_out[o][k], masked -----------
		h4 da[NT];  // this wave's slice of dL/d(pre-activation): (neuron 16w+lr, samples 16t+4g+r)
		{
			const half_t* hlast = hT + HM * WIDTH * SP;
			const h4 bw = *(const h4*)(wt_out + (size_t)(16 * w + lr) * 16 + 4 * g);
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 a = *(const h4*)(dL_doutput + ((size_t)tile * S + 16 * t + lr) * 16 + 4 * g);
				const f4 acc = mfma_16x16x16(a, bw, zero4());
				const h4 hv = *(const h4*)(hlast + (16 * w + lr) * SP + ft_col16(w, 16 * t) + lane_c4);
				da[t] = act_backward4<GENERAL>(act, acc, hv);  // transfer on post-activation values (common_device.h:363-418)
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) dact0[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = da[t][r];
			}
			if (want_grads) {  // dW_out^T[k][o] += sum_s A_last[k][s] dY[o][s]
#pragma unroll
				for (uint32_t tp = 0; tp < NTP; ++tp) {
					const h8 a = *(const h8*)(hlast + (16 * w + lr) * SP + ft_col(16 * w + lr, 32 * tp + 8 * g));
					const h8 b = *(const h8*)(dyT + lr * SP + 32 * tp + 8 * g);
					accO = mfma_16x16x32(a, b, accO);
				}
			}
		}
		__syncthreads();

		half_t* cur = dact0;
		half_t* nxt = dact1;

		// ---- C. hidden matrices, last to first ----------------------------------------------
#pragma unroll
		for (int j = (int)HM - 1; j >= 0; --j) {
			const half_t* hj = hT + j * WIDTH * SP;  // input activation of hidden matrix j
			if (want_grads) {  // dW_j[out 16w+..][in 16b+..] += sum_s dA[out][s] A_j[in][s]; A operand = own registers
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
#pragma unroll
					for (uint32_t tp = 0; tp < NTP; ++tp) {
						const h8 a = pack8(da[2 * tp], da[2 * tp + 1]);
						const h4 b0 = *(const h4*)(hj + (16 * b + lr) * SP + ft_col16(b, 32 * tp) + lane_c4);
						const h4 b1 = *(const h4*)(hj + (16 * b + lr) * SP + ft_col16(b, 32 * tp + 16) + lane_c4);
						accH[j][b] = mfma_16x16x32(a, pack8(b0, b1), accH[j][b]);
					}
				}
			}
			// dA_j[s][k] = sum_jj dA_{j+1}[s][jj] M_j[jj][k], masked by A_j > 0
			const half_t* wt = wt_hid + (size_t)j * WIDTH * WIDTH + (size_t)(16 * w + lr) * WIDTH;
			f4 acc[NT];
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
			for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
				const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
					acc[t] = mfma_16x16x32(a, bw, acc[t]);
				}
			}
			if constexpr (WIDTH % 32 != 0) {
				constexpr uint32_t k0 = WIDTH & ~31u;
				const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
					acc[t] = mfma_16x16x16(a, bw, acc[t]);
				}
			}
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 hv = *(const h4*)(hj + (16 * w + lr) * SP + ft_col16(w, 16 * t) + lane_c4);
				da[t] = act_backward4<GENERAL>(act, acc[t], hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) nxt[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = da[t][r];
			}
			__syncthreads();
			half_t* tmp = cur;
			cur = nxt;
			nxt = tmp;
		}

		// ---- D. input matrix ----------------------------------------------------------------
		if (want_grads) {
#pragma unroll
			for (uint32_t b = 0; b < MAX_INB; ++b) {
				if (b < nb_in) {
#pragma unroll
					for (uint32_t tp = 0; tp < NTP; ++tp) {
						const h8 a = pack8(da[2 * tp], da[2 * tp + 1]);
						const h4 b0 = *(const h4*)(xT + (16 * b + lr) * SP + ft_col16(b, 32 * tp) + lane_c4);
						const h4 b1 = *(const h4*)(xT + (16 * b + lr) * SP + ft_col16(b, 32 * tp + 16) + lane_c4);
						accI[b] = mfma_16x16x32(a, pack8(b0, b1), accI[b]);
					}
				}
			}
		}
		if (want_dx) {  // dX[s][k] = sum_jj dA_0[s][jj] M_in[jj][k]   (no activation on the network input)
			for (uint32_t sl = w; sl < nb_in; sl += NW) {
				const half_t* wt = wt_in + (size_t)(16 * sl + lr) * WIDTH;
				f4 acc[NT];
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
				for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
					const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
						acc[t] = mfma_16x16x32(a, bw, acc[t]);
					}
				}
				if constexpr (WIDTH % 32 != 0) {
					constexpr uint32_t k0 = WIDTH & ~31u;
					const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
						acc[t] = mfma_16x16x16(a, bw, acc[t]);
					}
				}
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 o = h4{(half_t)acc[t][0], (half_t)acc[t][1], (half_t)acc[t][2], (half_t)acc[t][3]};
					*(h4*)(dxT + (16 * sl + lr) * SP + 16 * t + 4 * g) = o;  // (feature 16sl+lr, samples 16t+4g+r)
				}
			}
		}
		__syncthreads();
		if (want_dx) {
			for (uint32_t c = tid; c < IN * (S / 8); c += THREADS) {
				const uint32_t k = c / (S / 8), cc = c % (S / 8);
				*(h8*)(dL_dinput + (size_t)k * n + (size_t)tile * S + 8 * cc) = *(const h8*)(dxT + k * SP + 8 * cc);
			}
		}
		__syncthreads();
	}

	// ---- fp32 partial weight gradients of this workgroup, same layout as the parameters -------
	if (want_grads) {
		float* P = partials + (size_t)blockIdx.x * m.n_params();
#pragma unroll
		for (uint32_t b = 0; b < MAX_INB; ++b) {
			if (b < nb_in) {
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) P[(size_t)(16 * w + 4 * g + r) * IN + 16 * b + lr] = accI[b][r];
			}
		}
		const size_t off_hid = (size_t)WIDTH * IN;
#pragma unroll
		for (uint32_t j = 0; j < HM; ++j)
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) P[off_hid + (size_t)j * WIDTH * WIDTH + (size_t)(16 * w + 4 * g + r) * WIDTH + 16 * b + lr] = accH[j][b][r];
		const size_t off_out = off_hid + (size_t)HM * WIDTH * WIDTH;
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) P[off_out + (size_t)lr * WIDTH + 16 * w + 4 * g + r] = accO[r];  // accO holds dW_out^T
	}
}

// =============================================================================================
// backward for networks deeper than the register-resident kernels cover (more than 4 hidden layers): the structure
// of the reference (fully_fused_mlp.cu:740-837) -- one pass propagates dL/d(pre-activation) through all layers and
// stores it per layer, then one weight-gradient product per matrix.  Depth is a run-time value here.
// =============================================================================================
template <uint32_t WIDTH, bool GENERAL>
__global__ void __launch_bounds__(WIDTH / 16 * 64) k_mlp_backward_chain(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params_t,
                                                                         const half_t* __restrict__ hidden, const half_t* __restrict__ dL_doutput,
                                                                         half_t* __restrict__ dact_all, half_t* __restrict__ dL_dinput) {
	constexpr uint32_t NW = WIDTH / 16, THREADS = NW * 64, S = MLP_BWD_TILE, NT = S / 16;
	constexpr uint32_t SP = S + 8, LDW = WIDTH + 8;
	TCNN_DYN_LDS(lds_raw);
	const uint32_t IN = m.in_width, nb_in = IN / 16, HM = m.n_hidden_matmuls;
	half_t* hT = (half_t*)lds_raw;        // [WIDTH][SP]  forward activation of the current layer, feature-major (masks)
	half_t* dact0 = hT + WIDTH * SP;      // [S][LDW]     dL/d(pre-activation), sample-major ping-pong
	half_t* dact1 = dact0 + S * LDW;
	half_t* dxT = dact1 + S * LDW;        // [IN][SP]

	const half_t* wt_in = params_t;
	const half_t* wt_hid = wt_in + (size_t)IN * WIDTH;
	const half_t* wt_out = wt_hid + (size_t)HM * WIDTH * WIDTH;
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t act = m.activation;
	const uint32_t n_tiles = n / S;

	auto stage_hidden = [&](uint32_t layer, uint32_t tile) {  // hidden[layer] tile, transposed (consecutive lanes = consecutive neurons)
		const half_t* src = hidden + ((size_t)layer * n + (size_t)tile * S) * WIDTH;
		for (uint32_t c = tid; c < S * (WIDTH / 8); c += THREADS) {
			const uint32_t i = c % S, cc = c / S;
			const h8 v = *(const h8*)(src + (size_t)i * WIDTH + 8 * cc);
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) hT[(8 * cc + j) * SP + i] = v[j];
		}
	};
	auto store_dact = [&](uint32_t layer, uint32_t tile, const half_t* tile_lds) {  // sample-major LDS tile -> dact_all[layer]
		half_t* dst = dact_all + ((size_t)layer * n + (size_t)tile * S) * WIDTH;
		for (uint32_t c = tid; c < S * (WIDTH / 8); c += THREADS) {
			const uint32_t i = c / (WIDTH / 8), cc = c % (WIDTH / 8);
			*(h8*)(dst + (size_t)i * WIDTH + 8 * cc) = *(const h8*)(tile_lds + i * LDW + 8 * cc);
		}
	};

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		stage_hidden(HM, tile);
		__syncthreads();
		{  // output matrix: dA_last[s][k] = sum_o dY[s][o] W_out[o][k], transferred through the last hidden activation
			const uint32_t OUTP = m.padded_out;
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				f4 acc = zero4();
				for (uint32_t ob = 0; ob < OUTP / 16; ++ob) {
					const h4 bw = *(const h4*)(wt_out + (size_t)(16 * w + lr) * OUTP + 16 * ob + 4 * g);
					const h4 a = *(const h4*)(dL_doutput + ((size_t)tile * S + 16 * t + lr) * OUTP + 16 * ob + 4 * g);
					acc = mfma_16x16x16(a, bw, acc);
				}
				const h4 hv = *(const h4*)(hT + (16 * w + lr) * SP + 16 * t + 4 * g);
				const h4 dv = act_backward4<GENERAL>(act, acc, hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) dact0[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = dv[r];
			}
		}
		__syncthreads();
		store_dact(HM, tile, dact0);
		half_t* cur = dact0;
		half_t* nxt = dact1;
		for (int j = (int)HM - 1; j >= 0; --j) {
			stage_hidden((uint32_t)j, tile);  // hT is free: the previous layer's masks were consumed before the barrier above
			__syncthreads();
			const half_t* wt = wt_hid + (size_t)j * WIDTH * WIDTH + (size_t)(16 * w + lr) * WIDTH;
			f4 acc[NT];
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
			for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
				const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
					acc[t] = mfma_16x16x32(a, bw, acc[t]);
				}
			}
			if constexpr (WIDTH % 32 != 0) {
				constexpr uint32_t k0 = WIDTH & ~31u;
				const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
					acc[t] = mfma_16x16x16(a, bw, acc[t]);
				}
			}
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 hv = *(const h4*)(hT + (16 * w + lr) * SP + 16 * t + 4 * g);
				const h4 dv = act_backward4<GENERAL>(act, acc[t], hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) nxt[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = dv[r];
			}
			__syncthreads();
			store_dact((uint32_t)j, tile, nxt);
			half_t* tmp = cur;
			cur = nxt;
			nxt = tmp;
		}
		if (dL_dinput) {  // dX[s][k] = sum_jj dA_0[s][jj] M_in[jj][k]
			for (uint32_t sl = w; sl < nb_in; sl += NW) {
				const half_t* wt = wt_in + (size_t)(16 * sl + lr) * WIDTH;
				f4 acc[NT];
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
				for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
					const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
						acc[t] = mfma_16x16x32(a, bw, acc[t]);
					}
				}
				if constexpr (WIDTH % 32 != 0) {
					constexpr uint32_t k0 = WIDTH & ~31u;
					const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
						acc[t] = mfma_16x16x16(a, bw, acc[t]);
					}
				}
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 o = h4{(half_t)acc[t][0], (half_t)acc[t][1], (half_t)acc[t][2], (half_t)acc[t][3]};
					*(h4*)(dxT + (16 * sl + lr) * SP + 16 * t + 4 * g) = o;
				}
			}
			__syncthreads();
			for (uint32_t c = tid; c < IN * (S / 8); c += THREADS) {
				const uint32_t k = c / (S / 8), cc = c % (S / 8);
				*(h8*)(dL_dinput + (size_t)k * n + (size_t)tile * S + 8 * cc) = *(const h8*)(dxT + k * SP + 8 * cc);
			}
		}
		__syncthreads();
	}
}

// dW[out][in] += sum_s d[s][out] * a[s][in] for ONE matrix.  d: sample-major [n][d_stride] (WO columns used), a: sample-major
// [n][WI] (a_feature_major == 0) or feature-major [WI][n].  Accumulators stay in registers over all tiles of a persistent
// workgroup; P = this workgroup's fp32 slab at the matrix' offset, natural [out][in] layout.  THREADS = WIDTH / 16 * 64.
template <uint32_t WIDTH>
__global__ void __launch_bounds__(WIDTH / 16 * 64) k_mlp_weight_gradient(const uint32_t n, const uint32_t WO, const uint32_t WI, const half_t* __restrict__ d,
                                                                          const uint32_t d_stride, const half_t* __restrict__ a, const int a_feature_major,
                                                                          float* __restrict__ partials, const size_t slab_stride, const size_t matrix_offset) {
	constexpr uint32_t NW = WIDTH / 16, THREADS = NW * 64, S = MLP_BWD_TILE, NTP = S / 32, SP = S + 8, MAXP = 8;
	TCNN_DYN_LDS(lds_raw);
	half_t* dT = (half_t*)lds_raw;  // [WO][SP]
	half_t* aT = dT + WO * SP;      // [WI][SP]
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t n_ob = WO / 16, n_ib = WI / 16, n_pairs = n_ob * n_ib;  // (out block, in block) products, dealt round-robin to the waves
	f4 acc[MAXP];
#pragma unroll
	for (uint32_t q = 0; q < MAXP; ++q) acc[q] = zero4();
	const uint32_t n_tiles = n / S;
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		for (uint32_t c = tid; c < S * (WO / 8); c += THREADS) {  // transpose d (consecutive lanes = consecutive samples' rows, 16 bytes each)
			const uint32_t i = c % S, cc = c / S;
			const h8 v = *(const h8*)(d + ((size_t)tile * S + i) * d_stride + 8 * cc);
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) dT[(8 * cc + j) * SP + i] = v[j];
		}
		if (a_feature_major) {
			for (uint32_t c = tid; c < WI * (S / 8); c += THREADS) {
				const uint32_t k = c / (S / 8), cc = c % (S / 8);
				*(h8*)(aT + k * SP + 8 * cc) = *(const h8*)(a + (size_t)k * n + (size_t)tile * S + 8 * cc);
			}
		} else {
			for (uint32_t c = tid; c < S * (WI / 8); c += THREADS) {
				const uint32_t i = c % S, cc = c / S;
				const h8 v = *(const h8*)(a + ((size_t)tile * S + i) * WI + 8 * cc);
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) aT[(8 * cc + j) * SP + i] = v[j];
			}
		}
		__syncthreads();
#pragma unroll
		for (uint32_t q = 0; q < MAXP; ++q) {
			const uint32_t p = w + q * NW;
			if (p < n_pairs) {
				const uint32_t ob = p / n_ib, ib = p % n_ib;
#pragma unroll
				for (uint32_t tp = 0; tp < NTP; ++tp) {
					const h8 av = *(const h8*)(dT + (16 * ob + lr) * SP + 32 * tp + 8 * g);
					const h8 bv = *(const h8*)(aT + (16 * ib + lr) * SP + 32 * tp + 8 * g);
					acc[q] = mfma_16x16x32(av, bv, acc[q]);
				}
			}
		}
		__syncthreads();
	}
	float* P = partials + (size_t)blockIdx.x * slab_stride + matrix_offset;
#pragma unroll
	for (uint32_t q = 0; q < MAXP; ++q) {
		const uint32_t p = w + q * NW;
		if (p < n_pairs) {
			const uint32_t ob = p / n_ib, ib = p % n_ib;
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) P[(size_t)(16 * ob + 4 * g + r) * WI + 16 * ib + lr] = acc[q][r];
		}
	}
}

// =============================================================================================
// fused training pass: forward + loss + backward of one sample tile without leaving the CU.
// What Trainer::training_step needs from the network (trainer.h:254-357) in ONE kernel: the hidden activations
// never travel to HBM (2 x 64 MB at the headline config), the prediction / dL_doutput are written once for the
// caller's ForwardContext, the encoded input is read once.  Same MFMA fragments and the same rounding points as
// k_mlp_forward -> k_loss -> k_mlp_backward, so the results are bit-identical to the unfused path.
// =============================================================================================
#ifndef TCNN_MLP_TRAIN_MIN_BLOCKS
#define TCNN_MLP_TRAIN_MIN_BLOCKS 2
#endif
template <uint32_t WIDTH, uint32_t HM, bool GENERAL>
__global__ void __launch_bounds__(WIDTH / 16 * 64, WIDTH == 128 ? 1 : TCNN_MLP_TRAIN_MIN_BLOCKS) k_mlp_train(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params,
                                                                const half_t* __restrict__ params_t, const half_t* __restrict__ input,
                                                                const MlpLossArgs la, half_t* __restrict__ output, half_t* __restrict__ dL_doutput,
                                                                half_t* __restrict__ dL_dinput, float* __restrict__ partials,
                                                                float* __restrict__ block_sums) {
	constexpr uint32_t NW = WIDTH / 16, THREADS = NW * 64, S = MLP_BWD_TILE, NT = S / 16, NTP = S / 32;
	constexpr uint32_t SP = S + 8, LDW = WIDTH + 8, NB = WIDTH / 16, MAX_INB = MLP_MAX_IN_WIDTH / 16, LDY = 16 + 8;
	TCNN_DYN_LDS(lds_raw);
	__shared__ float red[THREADS];
	const uint32_t IN = m.in_width, nb_in = IN / 16, ldi = IN + 8;
	half_t* xT = (half_t*)lds_raw;                 // [IN][SP]           network input, feature-major (dW_in operand)
	half_t* hT = xT + IN * SP;                     // [HM+1][WIDTH][SP]  forward activations, feature-major (dW operands, ReLU masks)
	half_t* buf0 = hT + (HM + 1) * WIDTH * SP;     // [S][LDW]           sample-major ping-pong: forward activations, then dL/d(pre-activation)
	half_t* buf1 = buf0 + S * LDW;
	half_t* dyT = buf1 + S * LDW;                  // [16][SP]           dL/dy, feature-major (dW_out operand)
	half_t* dys = dyT + 16 * SP;                   // [S][LDY]           dL/dy, sample-major (dA_last operand)
	half_t* xs = dys + S * LDY;                    // [S][ldi]           network input, sample-major (first layer operand) ...
	half_t* dxT = xs;                              // [IN][SP]           ... later dL/dinput, feature-major (the two never overlap in time)

	const half_t* wt_in = params_t;                             // [IN][WIDTH]
	const half_t* wt_hid = wt_in + (size_t)IN * WIDTH;          // HM x [WIDTH][WIDTH]
	const half_t* wt_out = wt_hid + (size_t)HM * WIDTH * WIDTH; // [WIDTH][16]

	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t act = m.activation, out_act = m.output_activation;
	const bool want_grads = partials != nullptr, want_dx = dL_dinput != nullptr;
	const uint32_t n_tiles = n / S;
	const float n_total = (float)la.n_total;
	float loss_sum = 0.0f;

	f4 accI[MAX_INB];
	f4 accH[HM > 0 ? HM : 1][NB];
	f4 accO = zero4();
#pragma unroll
	for (uint32_t b = 0; b < MAX_INB; ++b) accI[b] = zero4();
#pragma unroll
	for (uint32_t j = 0; j < (HM > 0 ? HM : 1); ++j)
#pragma unroll
		for (uint32_t b = 0; b < NB; ++b) accH[j][b] = zero4();

	// The first PF input chunks of a thread are fetched one tile ahead (registers), and the loss targets of a tile are
	// requested before its forward pass: neither global round trip sits on the tile's critical path.
	constexpr uint32_t PF = 2, NTW = (NT + NW - 1) / NW;
	const uint32_t n_chunks = IN * (S / 8);
	auto load_chunk = [&](uint32_t tile, uint32_t c) {
		const uint32_t k = c % IN, cc = c / IN;
		return *(const h8*)(input + (size_t)k * n + (size_t)tile * S + 8 * cc);
	};
	h8 pf[PF];
#pragma unroll
	for (uint32_t u = 0; u < PF; ++u) {
		const uint32_t c = tid + u * THREADS;
		if (blockIdx.x < n_tiles && c < n_chunks) pf[u] = load_chunk(blockIdx.x, c);
	}

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		// ---- stage the input tile in both layouts (consecutive lanes = consecutive features: conflict-free transposition)
		for (uint32_t c = tid, u = 0; c < n_chunks; c += THREADS, ++u) {
			const uint32_t k = c % IN, cc = c / IN;
			const h8 v = u < PF ? pf[u < PF ? u : 0] : load_chunk(tile, c);
			*(h8*)(xT + k * SP + 8 * cc) = v;
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) xs[(8 * cc + j) * ldi + k] = v[j];
		}
		{
			const uint32_t next = tile + gridDim.x;
#pragma unroll
			for (uint32_t u = 0; u < PF; ++u) {
				const uint32_t c = tid + u * THREADS;
				if (next < n_tiles && c < n_chunks) pf[u] = load_chunk(next, c);
			}
		}
		// this lane's targets (output 4g+r of sample 16t+lr for its output-layer tiles t = w, w + NW, ...)
		float tgt[NTW][4], pdf[NTW][4];
#pragma unroll
		for (uint32_t q = 0; q < NTW; ++q) {
			const uint32_t t = w + q * NW;
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) {
				const uint32_t dim = 4 * g + r;
				const bool live = t < NT && dim < la.dims && !la.external_dL_doutput;
				const size_t target_idx = ((size_t)tile * S + 16 * t + lr) * la.dims + dim;
				tgt[q][r] = live ? la.targets[target_idx] : 0.0f;
				pdf[q][r] = live && la.data_pdf ? la.data_pdf[target_idx] : 1.0f;
			}
		}
		__syncthreads();

		// ================= forward =================
		{
			const half_t* cur = xs;
			uint32_t ldc = ldi;
			half_t* nxt = buf0;
			const half_t* Wl = params;
			uint32_t K = IN;
#pragma unroll
			for (uint32_t layer = 0; layer <= HM; ++layer) {
				f4 acc[NT];
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
				const half_t* wrow = Wl + (size_t)(16 * w + lr) * K;
				for (uint32_t kb = 0; kb < K / 32; ++kb) {
					const h8 a = *(const h8*)(wrow + 32 * kb + 8 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h8 b = *(const h8*)(cur + (16 * t + lr) * ldc + 32 * kb + 8 * g);
						acc[t] = mfma_16x16x32(a, b, acc[t]);
					}
				}
				if (K & 16u) {
					const uint32_t k0 = K & ~31u;
					const h4 a = *(const h4*)(wrow + k0 + 4 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h4 b = *(const h4*)(cur + (16 * t + lr) * ldc + k0 + 4 * g);
						acc[t] = mfma_16x16x16(a, b, acc[t]);
					}
				}
				// accumulator (neuron 16w+4g+r, sample 16t+lr) -> activation -> both layouts
				half_t* hl = hT + layer * WIDTH * SP;
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 o = act_forward4<GENERAL>(act, acc[t]);
#pragma unroll
					for (uint32_t r = 0; r < 4; ++r) hl[(16 * w + 4 * g + r) * SP + 16 * t + lr] = o[r];
					*(h4*)(nxt + (16 * t + lr) * LDW + 16 * w + 4 * g) = o;
				}
				__syncthreads();
				cur = nxt;
				ldc = LDW;
				nxt = nxt == buf0 ? buf1 : buf0;
				Wl += (size_t)WIDTH * K;
				K = WIDTH;
			}
			// ---- output layer + loss: (output 4g+r, sample 16t+lr)
#pragma unroll
			for (uint32_t q = 0; q < NTW; ++q) {
				const uint32_t t = w + q * NW;
				if (t >= NT) break;
				f4 acc = zero4();
				const half_t* wrow = Wl + (size_t)lr * WIDTH;
#pragma unroll
				for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
					const h8 a = *(const h8*)(wrow + 32 * kb + 8 * g);
					const h8 b = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
					acc = mfma_16x16x32(a, b, acc);
				}
				if constexpr (WIDTH % 32 != 0) {
					constexpr uint32_t k0 = WIDTH & ~31u;
					const h4 a = *(const h4*)(wrow + k0 + 4 * g);
					const h4 b = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
					acc = mfma_16x16x16(a, b, acc);
				}
				const h4 o = act_forward4<GENERAL>(out_act, acc);
				const size_t i = (size_t)tile * S + 16 * t + lr;
				h4 gy;
				if (la.external_dL_doutput) {
					gy = *(const h4*)(la.external_dL_doutput + i * 16 + 4 * g);
				} else {
#pragma unroll
					for (uint32_t r = 0; r < 4; ++r) {
						const uint32_t dim = 4 * g + r;
						gy[r] = (half_t)0.0f;
						if (dim < la.dims) {  // relative_l2.h:57-61: padding outputs carry no loss
							float value;
							gy[r] = loss_element<GENERAL>(la.type, (float)o[r], tgt[q][r], pdf[q][r], n_total, la.loss_scale, value);
							loss_sum += value;
						}
					}
				}
				if (output) *(h4*)(output + i * 16 + 4 * g) = o;
				if (dL_doutput) *(h4*)(dL_doutput + i * 16 + 4 * g) = gy;  // the caller's context holds dL/doutput ...
				gy = act_backward4<GENERAL>(out_act, f4{(float)gy[0], (float)gy[1], (float)gy[2], (float)gy[3]}, o);  // ... the backward pass continues from dL/d(pre-activation) (fully_fused_mlp.cu:760-763)
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) dyT[(4 * g + r) * SP + 16 * t + lr] = gy[r];
				*(h4*)(dys + (16 * t + lr) * LDY + 4 * g) = gy;
			}
			__syncthreads();
		}

		// ================= backward (k_mlp_backward steps B-D on the LDS-resident tiles) =================
		half_t* dact0 = buf0;
		half_t* dact1 = buf1;
		h4 da[NT];
		{
			const half_t* hlast = hT + HM * WIDTH * SP;
			const h4 bw = *(const h4*)(wt_out + (size_t)(16 * w + lr) * 16 + 4 * g);
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 a = *(const h4*)(dys + (16 * t + lr) * LDY + 4 * g);
				const f4 acc = mfma_16x16x16(a, bw, zero4());
				const h4 hv = *(const h4*)(hlast + (16 * w + lr) * SP + 16 * t + 4 * g);
				da[t] = act_backward4<GENERAL>(act, acc, hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) dact0[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = da[t][r];
			}
			if (want_grads) {
#pragma unroll
				for (uint32_t tp = 0; tp < NTP; ++tp) {
					const h8 a = *(const h8*)(hlast + (16 * w + lr) * SP + 32 * tp + 8 * g);
					const h8 b = *(const h8*)(dyT + lr * SP + 32 * tp + 8 * g);
					accO = mfma_16x16x32(a, b, accO);
				}
			}
		}
		__syncthreads();

		half_t* cur = dact0;
		half_t* nxt = dact1;
#pragma unroll
		for (int j = (int)HM - 1; j >= 0; --j) {
			const half_t* hj = hT + j * WIDTH * SP;
			if (want_grads) {
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
#pragma unroll
					for (uint32_t tp = 0; tp < NTP; ++tp) {
						const h8 a = pack8(da[2 * tp], da[2 * tp + 1]);
						const h4 b0 = *(const h4*)(hj + (16 * b + lr) * SP + 32 * tp + 4 * g);
						const h4 b1 = *(const h4*)(hj + (16 * b + lr) * SP + 32 * tp + 16 + 4 * g);
						accH[j][b] = mfma_16x16x32(a, pack8(b0, b1), accH[j][b]);
					}
				}
			}
			const half_t* wt = wt_hid + (size_t)j * WIDTH * WIDTH + (size_t)(16 * w + lr) * WIDTH;
			f4 acc[NT];
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
			for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
				const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
					acc[t] = mfma_16x16x32(a, bw, acc[t]);
				}
			}
			if constexpr (WIDTH % 32 != 0) {
				constexpr uint32_t k0 = WIDTH & ~31u;
				const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
					acc[t] = mfma_16x16x16(a, bw, acc[t]);
				}
			}
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 hv = *(const h4*)(hj + (16 * w + lr) * SP + 16 * t + 4 * g);
				da[t] = act_backward4<GENERAL>(act, acc[t], hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) nxt[(16 * t + 4 * g + r) * LDW + 16 * w + lr] = da[t][r];
			}
			__syncthreads();
			half_t* tmp = cur;
			cur = nxt;
			nxt = tmp;
		}

		if (want_grads) {
#pragma unroll
			for (uint32_t b = 0; b < MAX_INB; ++b) {
				if (b < nb_in) {
#pragma unroll
					for (uint32_t tp = 0; tp < NTP; ++tp) {
						const h8 a = pack8(da[2 * tp], da[2 * tp + 1]);
						const h4 b0 = *(const h4*)(xT + (16 * b + lr) * SP + 32 * tp + 4 * g);
						const h4 b1 = *(const h4*)(xT + (16 * b + lr) * SP + 32 * tp + 16 + 4 * g);
						accI[b] = mfma_16x16x32(a, pack8(b0, b1), accI[b]);
					}
				}
			}
		}
		if (want_dx) {
			for (uint32_t sl = w; sl < nb_in; sl += NW) {
				const half_t* wt = wt_in + (size_t)(16 * sl + lr) * WIDTH;
				f4 acc[NT];
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = zero4();
#pragma unroll
				for (uint32_t kb = 0; kb < WIDTH / 32; ++kb) {
					const h8 bw = *(const h8*)(wt + 32 * kb + 8 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h8 a = *(const h8*)(cur + (16 * t + lr) * LDW + 32 * kb + 8 * g);
						acc[t] = mfma_16x16x32(a, bw, acc[t]);
					}
				}
				if constexpr (WIDTH % 32 != 0) {
					constexpr uint32_t k0 = WIDTH & ~31u;
					const h4 bw = *(const h4*)(wt + k0 + 4 * g);
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) {
						const h4 a = *(const h4*)(cur + (16 * t + lr) * LDW + k0 + 4 * g);
						acc[t] = mfma_16x16x16(a, bw, acc[t]);
					}
				}
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) {
					const h4 o = h4{(half_t)acc[t][0], (half_t)acc[t][1], (half_t)acc[t][2], (half_t)acc[t][3]};
					*(h4*)(dxT + (16 * sl + lr) * SP + 16 * t + 4 * g) = o;
				}
			}
		}
		__syncthreads();
		if (want_dx) {
			for (uint32_t c = tid; c < IN * (S / 8); c += THREADS) {
				const uint32_t k = c / (S / 8), cc = c % (S / 8);
				*(h8*)(dL_dinput + (size_t)k * n + (size_t)tile * S + 8 * cc) = *(const h8*)(dxT + k * SP + 8 * cc);
			}
		}
		__syncthreads();
	}

	// ---- this workgroup's share of the loss
	if (block_sums) {
		red[tid] = loss_sum;
		__syncthreads();
		for (uint32_t k = THREADS / 2; k > 0; k >>= 1) {
			if (tid < k) red[tid] += red[tid + k];
			__syncthreads();
		}
		if (tid == 0) block_sums[blockIdx.x] = red[0];
	}

	// ---- fp32 partial weight gradients of this workgroup, same layout as the parameters -------
	if (want_grads) {
		float* P = partials + (size_t)blockIdx.x * m.n_params();
#pragma unroll
		for (uint32_t b = 0; b < MAX_INB; ++b) {
			if (b < nb_in) {
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) P[(size_t)(16 * w + 4 * g + r) * IN + 16 * b + lr] = accI[b][r];
			}
		}
		const size_t off_hid = (size_t)WIDTH * IN;
#pragma unroll
		for (uint32_t j = 0; j < HM; ++j)
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) P[off_hid + (size_t)j * WIDTH * WIDTH + (size_t)(16 * w + 4 * g + r) * WIDTH + 16 * b + lr] = accH[j][b][r];
		const size_t off_out = off_hid + (size_t)HM * WIDTH * WIDTH;
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) P[off_out + (size_t)lr * WIDTH + 16 * w + 4 * g + r] = accO[r];
	}
}

__global__ void __launch_bounds__(256) k_mlp_output_activation_backward(uint32_t n_groups, uint32_t act, const half_t* __restrict__ output,
                                                                         const half_t* __restrict__ dL_doutput, half_t* __restrict__ dL_dpreact) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // 8 halves per thread
	if (i >= n_groups) return;
	const h8 o = *(const h8*)(output + (size_t)i * 8), d = *(const h8*)(dL_doutput + (size_t)i * 8);
	h8 r;
#pragma unroll
	for (uint32_t q = 0; q < 2; ++q) {
		const h4 v = act_backward4<true>(act, f4{(float)d[4 * q], (float)d[4 * q + 1], (float)d[4 * q + 2], (float)d[4 * q + 3]}, h4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]});
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) r[4 * q + j] = v[j];
	}
	*(h8*)(dL_dpreact + (size_t)i * 8) = r;
}

constexpr uint32_t FINALIZE_GROUPS = 32;  // slab groups per block (x 32 parameters = 1024 threads)

__global__ void __launch_bounds__(32 * FINALIZE_GROUPS) k_mlp_finalize_gradients(const MlpMeta m, uint32_t n_params, uint32_t n_partials,
                                                                                 const float* __restrict__ partials, half_t* __restrict__ grads,
                                                                                 int accumulate, uint32_t order) {
	// 32 parameters x 32 slab groups; group g sums slabs g, g + 32, ... with 16 loads in flight (the sum over <= 512 slabs is
	// latency-bound: one round trip for the 512 slabs of a persistent training kernel instead of two); fixed summation order ->
	// deterministic gradients
	__shared__ float red[FINALIZE_GROUPS][32];
	const uint32_t lane = threadIdx.x & 31u, group = threadIdx.x >> 5;
	const uint32_t i = blockIdx.x * 32u + lane;
	float s = 0.0f;
	if (i < n_params) {
		constexpr uint32_t U = 16;
		for (uint32_t b = group; b < n_partials; b += FINALIZE_GROUPS * U) {
			float v[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) {
				const uint32_t bb = b + u * FINALIZE_GROUPS;
				v[u] = bb < n_partials ? partials[(size_t)bb * n_params + i] : 0.0f;
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) s += v[u];
		}
	}
	red[group][lane] = s;
	__syncthreads();
	if (group == 0 && i < n_params) {
		float t = red[0][lane];
#pragma unroll
		for (uint32_t k = 1; k < FINALIZE_GROUPS; ++k) t += red[k][lane];
		// slabs in the register order of k_mlp_train_wave: position i of every slab belongs to parameter mlp_wave_slab_param(m, i)
		const uint32_t param = order == (uint32_t)SlabOrder::WaveRegisters ? mlp_wave_slab_param(m, i) : i;
		if (accumulate) t += (float)grads[param];
		grads[param] = to_half_rn(t);
	}
}

// =============================================================================================
// host launchers
// =============================================================================================
static void check_meta(const MlpMeta& m, uint32_t n) {
	if (m.width != 16 && m.width != 32 && m.width != 64 && m.width != 128) {
		throw std::runtime_error("FullyFusedMLP only supports 16, 32, 64, and 128 neurons, but got " + std::to_string(m.width) + ".");
	}
	if (m.in_width % 16 != 0 || m.in_width == 0 || m.in_width > MLP_MAX_IN_WIDTH) {
		throw std::runtime_error("FullyFusedMLP: input width must be a multiple of 16 and at most " + std::to_string(MLP_MAX_IN_WIDTH) + ".");
	}
	if (m.padded_out % 16 != 0 || m.padded_out == 0 || m.padded_out > MLP_MAX_OUT_WIDTH) {
		throw std::runtime_error("FullyFusedMLP: at most " + std::to_string(MLP_MAX_OUT_WIDTH) + " output dimensions are supported.");
	}
	if (n % BATCH_SIZE_GRANULARITY != 0) throw std::runtime_error("Batch size must be a multiple of 256.");
}

template <uint32_t WIDTH>
static void launch_forward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* hidden,
                           half_t* output) {
	constexpr uint32_t S = mlp_fwd_tile(WIDTH);
	const uint32_t ld = (m.in_width > WIDTH ? m.in_width : WIDTH) + 8;
	const uint32_t lds_bytes = 2 * S * ld * (uint32_t)sizeof(half_t);
	const uint32_t n_tiles = n / S;
	const uint32_t blocks = n_tiles < 2048 ? n_tiles : 2048;
	const bool general = !act_is_simple(m.activation) || !act_is_simple(m.output_activation);
#define TCNN_FWD_LAUNCH(SAVE_, GENERAL_)                                                                                                      \
	TCNN_SET_MAX_DYN_LDS((k_mlp_forward<WIDTH, SAVE_, GENERAL_>), lds_bytes);                                                                 \
	TCNN_LAUNCH((k_mlp_forward<WIDTH, SAVE_, GENERAL_>), dim3(blocks), dim3(WIDTH / 16 * 64), lds_bytes, stream, m, n, params, input, hidden, output);
	if (hidden) {
		if (general) { TCNN_FWD_LAUNCH(true, true) } else { TCNN_FWD_LAUNCH(true, false) }
	} else {
		if (general) { TCNN_FWD_LAUNCH(false, true) } else { TCNN_FWD_LAUNCH(false, false) }
	}
#undef TCNN_FWD_LAUNCH
}

void mlp_forward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* hidden, half_t* output) {
	check_meta(m, n);
	if (n == 0) return;
	if (!hidden && mlp_infer_wave_supported(m, n)) {
		mlp_infer_wave(stream, m, n, params, input, output);
		return;
	}
	switch (m.width) {
		case 16: launch_forward<16>(stream, m, n, params, input, hidden, output); break;
		case 32: launch_forward<32>(stream, m, n, params, input, hidden, output); break;
		case 64: launch_forward<64>(stream, m, n, params, input, hidden, output); break;
		case 128: launch_forward<128>(stream, m, n, params, input, hidden, output); break;
	}
}

void mlp_transpose_weights(hipStream_t stream, const MlpMeta& m, const half_t* params, half_t* params_t) {
	TCNN_LAUNCH(k_mlp_transpose_weights, dim3(div_round_up(m.n_params(), 256u)), dim3(256), 0, stream, m, params, params_t);
}

uint32_t mlp_backward_n_partials(const MlpMeta& m, uint32_t n) {
	(void)m;
	const uint32_t n_tiles = n / MLP_BWD_TILE;
#ifndef TCNN_MLP_PARTIALS
#define TCNN_MLP_PARTIALS 512
#endif
	return n_tiles < TCNN_MLP_PARTIALS ? n_tiles : TCNN_MLP_PARTIALS;  // two persistent workgroups per CU (latency hiding); measured better than 256
}

template <uint32_t WIDTH, uint32_t HM>
static void launch_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params_t, const half_t* input, const half_t* hidden,
                            const half_t* dL_doutput, half_t* dL_dinput, float* partials) {
	constexpr uint32_t S = MLP_BWD_TILE, SP = MLP_BWD_SP, LDW = WIDTH + 8;
	const uint32_t halves = 2 * m.in_width * SP + (HM + 1) * WIDTH * SP + 2 * S * LDW + 16 * SP;
	const uint32_t lds_bytes = halves * (uint32_t)sizeof(half_t);
	const uint32_t blocks = mlp_backward_n_partials(m, n);
	if (!act_is_simple(m.activation)) {
		TCNN_SET_MAX_DYN_LDS((k_mlp_backward<WIDTH, HM, true>), lds_bytes);
		TCNN_LAUNCH((k_mlp_backward<WIDTH, HM, true>), dim3(blocks), dim3(WIDTH / 16 * 64), lds_bytes, stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials);
	} else {
		TCNN_SET_MAX_DYN_LDS((k_mlp_backward<WIDTH, HM, false>), lds_bytes);
		TCNN_LAUNCH((k_mlp_backward<WIDTH, HM, false>), dim3(blocks), dim3(WIDTH / 16 * 64), lds_bytes, stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials);
	}
}

// networks with more than MLP_MAX_HIDDEN_MATMULS_TRAIN + 1 hidden layers: chain kernel + one weight-gradient product per matrix
template <uint32_t WIDTH>
static void launch_backward_deep(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params_t, const half_t* input, const half_t* hidden,
                                 const half_t* dL_doutput, half_t* dL_dinput, float* partials, void* workspace) {
	constexpr uint32_t S = MLP_BWD_TILE, SP = S + 8, LDW = WIDTH + 8, THREADS = WIDTH / 16 * 64;
	if (!workspace) throw std::runtime_error("mlp_backward: networks with more than 4 hidden layers or 16 outputs need a workspace (mlp_backward_workspace_bytes)");
	half_t* dact_all = (half_t*)workspace;  // [n_hidden][n][WIDTH]
	const uint32_t blocks = mlp_backward_n_partials(m, n);
	const uint32_t chain_lds = (WIDTH * SP + 2 * S * LDW + m.in_width * SP) * (uint32_t)sizeof(half_t);
	if (!act_is_simple(m.activation)) {
		TCNN_SET_MAX_DYN_LDS((k_mlp_backward_chain<WIDTH, true>), chain_lds);
		TCNN_LAUNCH((k_mlp_backward_chain<WIDTH, true>), dim3(blocks), dim3(THREADS), chain_lds, stream, m, n, params_t, hidden, dL_doutput, dact_all, dL_dinput);
	} else {
		TCNN_SET_MAX_DYN_LDS((k_mlp_backward_chain<WIDTH, false>), chain_lds);
		TCNN_LAUNCH((k_mlp_backward_chain<WIDTH, false>), dim3(blocks), dim3(THREADS), chain_lds, stream, m, n, params_t, hidden, dL_doutput, dact_all, dL_dinput);
	}
	if (!partials) return;
	const uint32_t HM = m.n_hidden_matmuls, IN = m.in_width;
	const size_t slab = m.n_params(), off_hid = (size_t)WIDTH * IN, off_out = off_hid + (size_t)HM * WIDTH * WIDTH;
	auto product = [&](uint32_t WO, uint32_t WI, const half_t* d, uint32_t d_stride, const half_t* a, int a_fm, size_t offset) {
		const uint32_t lds = (WO + WI) * SP * (uint32_t)sizeof(half_t);
		TCNN_SET_MAX_DYN_LDS((k_mlp_weight_gradient<WIDTH>), lds);
		TCNN_LAUNCH((k_mlp_weight_gradient<WIDTH>), dim3(blocks), dim3(THREADS), lds, stream, n, WO, WI, d, d_stride, a, a_fm, partials, slab, offset);
	};
	// output matrix [16][W]: d = dL/doutput, a = last hidden activation
	product(m.padded_out, WIDTH, dL_doutput, m.padded_out, hidden + (size_t)HM * n * WIDTH, 0, off_out);
	for (uint32_t j = 0; j < HM; ++j) {  // hidden matrix j: d = dL/d(pre-activation of hidden layer j + 1), a = hidden layer j
		product(WIDTH, WIDTH, dact_all + (size_t)(j + 1) * n * WIDTH, WIDTH, hidden + (size_t)j * n * WIDTH, 0, off_hid + (size_t)j * WIDTH * WIDTH);
	}
	product(WIDTH, IN, dact_all, WIDTH, input, 1, 0);  // input matrix: a = the feature-major network input
}

template <uint32_t WIDTH>
static void dispatch_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params_t, const half_t* input, const half_t* hidden,
                              const half_t* dL_doutput, half_t* dL_dinput, float* partials, void* workspace) {
	if (m.padded_out != 16) {  // the register-resident kernels are built for one block of 16 outputs
		launch_backward_deep<WIDTH>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace);
		return;
	}
	switch (m.n_hidden_matmuls) {
		case 0: launch_backward<WIDTH, 0>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials); break;
		case 1: launch_backward<WIDTH, 1>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials); break;
		case 2: launch_backward<WIDTH, 2>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials); break;
		case 3: launch_backward<WIDTH, 3>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials); break;
		default: launch_backward_deep<WIDTH>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace); break;
	}
}

size_t mlp_backward_workspace_bytes(const MlpMeta& m, uint32_t n) {
	const bool layer_by_layer = m.n_hidden_matmuls > MLP_MAX_HIDDEN_MATMULS_TRAIN || m.padded_out != 16;
	return layer_by_layer ? (size_t)(m.n_hidden_matmuls + 1) * n * m.width * sizeof(half_t) : 0;
}

void mlp_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params_t, const half_t* input, const half_t* hidden,
                  const half_t* dL_doutput, half_t* dL_dinput, float* partials, void* workspace) {
	check_meta(m, n);
	if (n == 0) return;
	switch (m.width) {
		case 16: dispatch_backward<16>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace); break;
		case 32: dispatch_backward<32>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace); break;
		case 64: dispatch_backward<64>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace); break;
		case 128: dispatch_backward<128>(stream, m, n, params_t, input, hidden, dL_doutput, dL_dinput, partials, workspace); break;
	}
}

static uint32_t mlp_train_lds_bytes(const MlpMeta& m) {
	const uint32_t S = MLP_BWD_TILE, SP = S + 8, LDW = m.width + 8, LDY = 16 + 8;
	const uint32_t in_region = std::max(S * (m.in_width + 8), m.in_width * SP);  // xs, later dxT
	return (m.in_width * SP + (m.n_hidden_matmuls + 1) * m.width * SP + 2 * S * LDW + 16 * SP + S * LDY + in_region) * (uint32_t)sizeof(half_t);
}

template <uint32_t WIDTH, uint32_t HM>
static void launch_train(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                         const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums) {
	const uint32_t lds_bytes = mlp_train_lds_bytes(m);
	const uint32_t blocks = mlp_backward_n_partials(m, n);
	if (!act_is_simple(m.activation) || !act_is_simple(m.output_activation) || !(la.external_dL_doutput || loss_is_simple(la.type))) {
		TCNN_SET_MAX_DYN_LDS((k_mlp_train<WIDTH, HM, true>), lds_bytes);
		TCNN_LAUNCH((k_mlp_train<WIDTH, HM, true>), dim3(blocks), dim3(WIDTH / 16 * 64), lds_bytes, stream, m, n, params, params_t, input, la, output,
		            dL_doutput, dL_dinput, partials, block_sums);
	} else {  // ReLU / None activations and (Relative)L2: an instance without any out-of-line call in its body
		TCNN_SET_MAX_DYN_LDS((k_mlp_train<WIDTH, HM, false>), lds_bytes);
		TCNN_LAUNCH((k_mlp_train<WIDTH, HM, false>), dim3(blocks), dim3(WIDTH / 16 * 64), lds_bytes, stream, m, n, params, params_t, input, la, output,
		            dL_doutput, dL_dinput, partials, block_sums);
	}
}

template <uint32_t WIDTH>
static void dispatch_train(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                           const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums) {
	switch (m.n_hidden_matmuls) {
		case 0: launch_train<WIDTH, 0>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 1: launch_train<WIDTH, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 2: launch_train<WIDTH, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 3: launch_train<WIDTH, 3>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		default: throw std::runtime_error("mlp_train: unsupported depth (check mlp_train_supported first)");
	}
}

uint32_t mlp_train_n_partials(const MlpMeta& m, uint32_t n, LossType loss) {
	if (mlp_train_wave_supported(m, n, loss)) return mlp_train_wave_n_partials(m, n);
	if (mlp_train_wide_supported(m, n)) return mlp_train_wide_n_partials(n);
	return mlp_backward_n_partials(m, n);
}

bool mlp_train_supported(const MlpMeta& m) {
	// 128-wide networks: k_mlp_train measured no faster than the three-kernel path (the weight-gradient accumulators of four
	// 128 x 128 matrices spill at 64-sample tiles); they have their own kernel (mlp_train_wide.hip) for 32 / 64 inputs
	if (m.width == 128) return mlp_train_wide_supported(m, MLP_BWD_TILE);
	return m.width <= 64 && m.padded_out == 16 && m.n_hidden_matmuls <= MLP_MAX_HIDDEN_MATMULS_TRAIN && mlp_train_lds_bytes(m) + 4096u <= 160u * 1024u;
}

SlabOrder mlp_train(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                    const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums,
                    const MlpF32Input* f32_input) {
	check_meta(m, n);
	if (n == 0) return SlabOrder::Params;
	if (!mlp_train_supported(m)) throw std::runtime_error("mlp_train: unsupported network shape (check mlp_train_supported first)");
	if (!la.external_dL_doutput && !loss_is_elementwise(la.type)) throw std::runtime_error("mlp_train: this loss needs whole output rows; use the stand-alone loss kernel");
	if (f32_input && (la.external_dL_doutput || !mlp_train_f32_input_supported(m, n, la.type))) {
		throw std::runtime_error("mlp_train: no instance reads an fp32 input for this shape (check mlp_train_f32_input_supported first)");
	}
	if (mlp_train_wave_supported(m, n, la.external_dL_doutput ? LossType::L2 : la.type)) {
		mlp_train_wave(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input);
		return SlabOrder::WaveRegisters;
	}
	if (m.width == 128) {
		mlp_train_wide(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums);
		return SlabOrder::Params;
	}
	switch (m.width) {
		case 16: dispatch_train<16>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 32: dispatch_train<32>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 64: dispatch_train<64>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
	}
	return SlabOrder::Params;
}

void mlp_output_activation_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* output, const half_t* dL_doutput,
                                    half_t* dL_dpreact) {
	if (n == 0) return;
	const uint32_t n_groups = n * m.padded_out / 8u;
	TCNN_LAUNCH(k_mlp_output_activation_backward, dim3(div_round_up(n_groups, 256u)), dim3(256), 0, stream, n_groups, m.output_activation, output,
	            dL_doutput, dL_dpreact);
}

void mlp_finalize_gradients(hipStream_t stream, const MlpMeta& m, uint32_t n_partials, const float* partials, half_t* grads, bool accumulate, SlabOrder order) {
	const uint32_t n_params = m.n_params();
	TCNN_LAUNCH(k_mlp_finalize_gradients, dim3(div_round_up(n_params, 32u)), dim3(32 * FINALIZE_GROUPS), 0, stream, m, n_params, n_partials, partials, grads,
	            accumulate ? 1 : 0, (uint32_t)order);
}

}  // namespace tcnn_hip
