// mlp_train_wide.hip -- the fused training pass (forward + loss + backward + weight gradients in one kernel) of 128-neuron networks.
//
// The reference fuses its 128-wide network into one kernel per direction with the weights in shared memory
// (fully_fused_mlp.cu:609-613, 895-898) and leaves the weight gradients to split-K GEMMs over activations it stored in HBM
// (:776-835).  Here one persistent workgroup per CU (8 waves, wave w owns neurons 16w..16w+15 as in mlp_kernels.hip) keeps
//   * the hidden weight matrices in LDS, ONE copy in their natural [out][in] layout, for the whole launch: the forward pass reads
//     a wave's rows as 16-byte MFMA A operands, the backward pass reads the transposed operand (8 consecutive `out` for one `in`)
//     out of the same image with the hardware transpose read (ds_read_b64_tr_b16, lds_read_tr4) -- no second orientation;
//   * every activation tile ONCE, sample-major: it is the next layer's B operand as written, and the transpose read turns it
//     into the "samples in k" operand of the weight-gradient MFMAs and into the per-neuron ReLU masks.  k_mlp_train writes
//     each activation twice (the feature-major copy with 2-byte scattered stores) and re-fetches the weights from L2 for every
//     64-sample tile with the load latency exposed;
//   * all weight gradients in fp32 MFMA accumulators across the tiles of the workgroup (108 registers per lane at three hidden
//     matrices), one slab per workgroup, summed in fixed order by k_mlp_finalize_gradients;
//   * the small input / output matrices as per-wave register fragments loaded once.
// 96 KB of weights leave room for tiles of 32 samples: 11 workgroup barriers per tile, nothing but the encoded input, the
// targets, the prediction and dL/dinput crosses HBM.  Same products, same k order inside every MFMA and the same rounding
// points as k_mlp_forward / k_loss / k_mlp_backward; the fp32 weight-gradient partial sums are grouped per 32 instead of per
// 64 samples.
#include "mlp_kernels.h"

#include <stdlib.h>

#include <stdexcept>

namespace tcnn_hip {

constexpr uint32_t WIDE = 128;       // neurons
constexpr uint32_t WIDE_S = 32;      // samples per tile
constexpr uint32_t WIDE_LD = WIDE + 8;   // row stride (halves) of the weight images: 16-byte aligned, bank-skewed
// Row stride of the sample-major tiles: + 16 halves where the LDS has room for it (32 inputs), + 8 otherwise.  With + 16 the 8 rows a
// half wave touches in a transpose read start 8 banks apart (conflict-free; + 8: two-way) -- scripts/lds_bank_model.py wide: 944 ->
// 760 modelled LDS cycles per tile and wave.  (XOR-swizzled unpadded rows model better still, 664, but the address arithmetic
// costs more than the conflicts: measured 258 vs 223 us, profiles/r02_exp_notes.txt.)
constexpr uint32_t wide_tile_ld(uint32_t kb_in) { return kb_in == 1 ? WIDE + 16 : WIDE + 8; }
constexpr uint32_t WIDE_SPX = WIDE_S + 8;  // row stride of the feature-major input tile
constexpr uint32_t WIDE_LDY = 16 + 8;
constexpr uint32_t WIDE_THREADS = WIDE / 16 * 64;

// this lane's word of a transpose read that yields M[row0 + j][col0 + (lane & 15)], j < 4, from a row-major image with row stride ld
TCNN_DEVICE h4 tr4(const half_t* image, uint32_t row0, uint32_t col0, uint32_t ld, uint32_t lane) {
	const uint32_t i = lane & 15u;
	return lds_read_tr4(image + (row0 + (i >> 2)) * ld + col0 + 4u * (i & 3u));
}
TCNN_DEVICE h8 tr8(const half_t* image, uint32_t row0, uint32_t col0, uint32_t ld, uint32_t lane) {
	return pack8(tr4(image, row0, col0, ld, lane), tr4(image, row0 + 4u, col0, ld, lane));
}

template <uint32_t HM, uint32_t KB_IN, bool GENERAL>
__global__ void __launch_bounds__(WIDE_THREADS, 1) k_mlp_train_wide(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params,
                                                                     const half_t* __restrict__ params_t, const half_t* __restrict__ input,
                                                                     const MlpLossArgs la, half_t* __restrict__ output, half_t* __restrict__ dL_doutput,
                                                                     half_t* __restrict__ dL_dinput, float* __restrict__ partials,
                                                                     float* __restrict__ block_sums) {
	constexpr uint32_t NW = WIDE / 16, THREADS = WIDE_THREADS, S = WIDE_S, NT = S / 16, LD = WIDE_LD, LDA = wide_tile_ld(KB_IN), SPX = WIDE_SPX, LDY = WIDE_LDY, NB = WIDE / 16;
	constexpr uint32_t IN = 32 * KB_IN, NB_IN = IN / 16, KB = WIDE / 32;
	static_assert(NT == 2 && NB_IN <= NW, "one 32-sample tile: two 16-sample blocks; at most one input block per wave");
	TCNN_DYN_LDS(lds_raw);
	half_t* wl = (half_t*)lds_raw;                  // [HM][WIDE][LD]     hidden weight matrices, natural layout
	half_t* hT = wl + HM * WIDE * LD;               // [HM+1][S][LDA]     forward activations, sample-major
	half_t* d0 = hT + (HM + 1) * S * LDA;           // [S][LDA]           dL/d(pre-activation), sample-major, ping ...
	half_t* d1 = d0 + S * LDA;                      // ... pong; the idle one carries dL/dinput [IN][SPX] at the end of a tile
	half_t* xT = d1 + S * LDA;                      // [IN][SPX]          network input, feature-major (as it arrives)
	half_t* dys = xT + IN * SPX;                    // [S][LDY]           dL/d(output pre-activation), sample-major

	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t act = m.activation, out_act = m.output_activation;
	const bool want_grads = partials != nullptr, want_dx = dL_dinput != nullptr;
	const uint32_t n_tiles = n / S;
	const float n_total = (float)la.n_total;
	float loss_sum = 0.0f;

	const half_t* W_hid = params + (size_t)WIDE * IN;           // HM x [WIDE][WIDE]
	const half_t* W_out = W_hid + (size_t)HM * WIDE * WIDE;     // [16][WIDE]
	const half_t* wt_in = params_t;                             // [IN][WIDE]
	const half_t* wt_out = wt_in + (size_t)IN * WIDE + (size_t)HM * WIDE * WIDE;  // [WIDE][16]

	// ---- once per workgroup: hidden matrices -> LDS, this wave's fragments of the input / output matrices -> registers
	for (uint32_t c = tid; c < HM * WIDE * (WIDE / 8); c += THREADS) {
		const uint32_t row = c / (WIDE / 8), cc = c % (WIDE / 8);
		*(h8*)(wl + row * LD + 8 * cc) = *(const h8*)(W_hid + (size_t)row * WIDE + 8 * cc);
	}
	h8 win[KB_IN];  // first layer, A operand: row 16w+lr
#pragma unroll
	for (uint32_t kb = 0; kb < KB_IN; ++kb) win[kb] = *(const h8*)(params + (size_t)(16 * w + lr) * IN + 32 * kb + 8 * g);
	h8 wof[KB];     // output layer, A operand: row lr (used by the waves that own an output tile)
#pragma unroll
	for (uint32_t kb = 0; kb < KB; ++kb) wof[kb] = *(const h8*)(W_out + (size_t)lr * WIDE + 32 * kb + 8 * g);
	const h4 wob = *(const h4*)(wt_out + (size_t)(16 * w + lr) * 16 + 4 * g);  // output layer backward, B operand
	h8 wdx[KB];     // dL/dinput, B operand: row 16w+lr of the transposed input matrix (waves w < NB_IN)
#pragma unroll
	for (uint32_t kb = 0; kb < KB; ++kb) wdx[kb] = w < NB_IN ? *(const h8*)(wt_in + (size_t)(16 * w + lr) * WIDE + 32 * kb + 8 * g) : h8{};

	f4 accI[NB_IN];
	f4 accH[HM > 0 ? HM : 1][NB];
	f4 accO = zero4();
#pragma unroll
	for (uint32_t b = 0; b < NB_IN; ++b) accI[b] = zero4();
#pragma unroll
	for (uint32_t j = 0; j < (HM > 0 ? HM : 1); ++j)
#pragma unroll
		for (uint32_t b = 0; b < NB; ++b) accH[j][b] = zero4();

	// the input chunk of a thread is fetched one tile ahead
	constexpr uint32_t N_CHUNKS = IN * (S / 8);
	static_assert(N_CHUNKS <= THREADS, "one input chunk per thread");
	const uint32_t chunk_k = tid % IN, chunk_cc = tid / IN;
	h8 pf = h8{};
	if (tid < N_CHUNKS && blockIdx.x < n_tiles) pf = *(const h8*)(input + (size_t)chunk_k * n + (size_t)blockIdx.x * S + 8 * chunk_cc);
#if !defined(TCNN_HOST_EMU)
	// (waited for HERE: left pending into the loop, the staging store at the loop's top -- one block for the first and for every later
	// iteration -- carries the wait for it, and on the later iterations that wait sits out dL/dinput's stores, see below)
	asm volatile("" : "+v"(pf));
#endif

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		if (tid < N_CHUNKS) *(h8*)(xT + chunk_k * SPX + 8 * chunk_cc) = pf;
		{
			const uint32_t next = tile + gridDim.x;
			if (tid < N_CHUNKS && next < n_tiles) pf = *(const h8*)(input + (size_t)chunk_k * n + (size_t)next * S + 8 * chunk_cc);
		}
		// this lane's targets (output 4g+r of sample 16w+lr; waves 0..NT-1 own the output tiles)
		float tgt[4], pdf[4];
		h4 gy_external = h4{};
		if (w < NT && la.external_dL_doutput) gy_external = *(const h4*)(la.external_dL_doutput + ((size_t)tile * S + 16 * w + lr) * 16 + 4 * g);
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) {
			const uint32_t dim = 4 * g + r;
			const bool live = w < NT && dim < la.dims && !la.external_dL_doutput;
			const size_t target_idx = ((size_t)tile * S + 16 * w + lr) * la.dims + dim;
			tgt[r] = live ? la.targets[target_idx] : 0.0f;
			pdf[r] = live && la.data_pdf ? la.data_pdf[target_idx] : 1.0f;
		}
		__syncthreads();

		// ================= forward =================
		{  // first layer: B operand = (sample 16t+lr, features 32kb+8g..) out of the feature-major input tile
			f4 acc[NT] = {zero4(), zero4()};
#pragma unroll
			for (uint32_t kb = 0; kb < KB_IN; ++kb) {
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = mfma_16x16x32(win[kb], tr8(xT, 32 * kb + 8 * g, 16 * t, SPX, lane), acc[t]);
			}
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 o = act_forward4<GENERAL>(act, acc[t]);
				*(h4*)(hT + (16 * t + lr) * LDA + 16 * w + 4 * g) = o;  // (neurons 16w+4g.., sample 16t+lr)
			}
		}
		__syncthreads();
#pragma unroll
		for (uint32_t l = 1; l <= HM; ++l) {
			const half_t* cur = hT + (l - 1) * S * LDA;
			const half_t* wrow = wl + ((l - 1) * WIDE + 16 * w + lr) * LD;
			f4 acc[NT] = {zero4(), zero4()};
#pragma unroll
			for (uint32_t kb = 0; kb < KB; ++kb) {
				const h8 a = *(const h8*)(wrow + 32 * kb + 8 * g);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = mfma_16x16x32(a, *(const h8*)(cur + (16 * t + lr) * LDA + 32 * kb + 8 * g), acc[t]);
			}
			half_t* nxt = hT + l * S * LDA;
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 o = act_forward4<GENERAL>(act, acc[t]);
				*(h4*)(nxt + (16 * t + lr) * LDA + 16 * w + 4 * g) = o;
			}
			__syncthreads();
		}
		const half_t* hlast = hT + HM * S * LDA;
		// Every load of the tile -- the next tile's input chunk, the targets, the pdf -- has landed BEFORE the tile's first store is issued, and
		// no load follows the stores.  gfx9 counts loads and stores in ONE counter (vmcnt); a register that a load MAY still be writing on some
		// path (the conditional target loads) is only reused behind "s_waitcnt vmcnt(0)", and the staging of the next tile's chunk at the
		// loop's top sat directly behind dL/dinput's stores: three to five waits per tile that sat out a store's round trip to HBM (one of
		// them between the prediction's store and dL/doutput's) with the workgroup's other waves waiting at the next barrier.
#if !defined(TCNN_HOST_EMU)
		asm volatile("" : "+v"(pf), "+v"(gy_external));
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) asm volatile("" : "+v"(tgt[r]), "+v"(pdf[r]));
#endif
		if (w < NT) {  // output layer + loss: (output 4g+r, sample 16w+lr)
			const uint32_t t = w;
			f4 acc = zero4();
#pragma unroll
			for (uint32_t kb = 0; kb < KB; ++kb) acc = mfma_16x16x32(wof[kb], *(const h8*)(hlast + (16 * t + lr) * LDA + 32 * kb + 8 * g), acc);
			const h4 o = act_forward4<GENERAL>(out_act, acc);
			const size_t i = (size_t)tile * S + 16 * t + lr;
			h4 gy;
			if (la.external_dL_doutput) {
				gy = gy_external;
			} else {
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) {
					const uint32_t dim = 4 * g + r;
					gy[r] = (half_t)0.0f;
					if (dim < la.dims) {  // relative_l2.h:57-61: padding outputs carry no loss
						float value;
						gy[r] = loss_element<GENERAL>(la.type, (float)o[r], tgt[r], pdf[r], n_total, la.loss_scale, value);
						loss_sum += value;
					}
				}
			}
			if (output) *(h4*)(output + i * 16 + 4 * g) = o;
			if (dL_doutput) *(h4*)(dL_doutput + i * 16 + 4 * g) = gy;
			gy = act_backward4<GENERAL>(out_act, f4{(float)gy[0], (float)gy[1], (float)gy[2], (float)gy[3]}, o);  // fully_fused_mlp.cu:760-763
			*(h4*)(dys + (16 * t + lr) * LDY + 4 * g) = gy;
		}
		__syncthreads();

		// ================= backward =================
		h4 da[NT];  // this wave's slice of dL/d(pre-activation): (neuron 16w+lr, samples 16t+4g+r)
#pragma unroll
		for (uint32_t t = 0; t < NT; ++t) {
			const f4 acc = mfma_16x16x16(*(const h4*)(dys + (16 * t + lr) * LDY + 4 * g), wob, zero4());
			const h4 hv = tr4(hlast, 16 * t + 4 * g, 16 * w, LDA, lane);
			da[t] = act_backward4<GENERAL>(act, acc, hv);
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) d0[(16 * t + 4 * g + r) * LDA + 16 * w + lr] = da[t][r];
		}
		if (want_grads)  // dW_out^T[neuron][o] += sum_s A_last[neuron][s] dY[o][s]; both operands transposed out of sample-major tiles
			accO = mfma_16x16x32(tr8(hlast, 8 * g, 16 * w, LDA, lane), tr8(dys, 8 * g, 0, LDY, lane), accO);
		__syncthreads();

		half_t* cur = d0;
		half_t* nxt = d1;
#pragma unroll
		for (int j = (int)HM - 1; j >= 0; --j) {
			const half_t* hj = hT + j * S * LDA;  // input activation of hidden matrix j
			const half_t* Wj = wl + j * WIDE * LD;
			const h8 a_da = pack8(da[0], da[1]);   // k = 8g+r <-> sample 4g+r, k = 8g+4+r <-> sample 16+4g+r
			if (want_grads) {  // dW_j[out 16w+..][in 16b+..] += sum_s dA[out][s] A_j[in][s]
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
					accH[j][b] = mfma_16x16x32(a_da, pack8(tr4(hj, 4 * g, 16 * b, LDA, lane), tr4(hj, 16 + 4 * g, 16 * b, LDA, lane)), accH[j][b]);
			}
			// dA_j[s][k] = sum_jj dA_{j+1}[s][jj] W_j[jj][k]: B operand = rows jj = 32kb+8g.. of column k = 16w+lr of the natural image
			f4 acc[NT] = {zero4(), zero4()};
#pragma unroll
			for (uint32_t kb = 0; kb < KB; ++kb) {
				const h8 bw = tr8(Wj, 32 * kb + 8 * g, 16 * w, LD, lane);
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t) acc[t] = mfma_16x16x32(*(const h8*)(cur + (16 * t + lr) * LDA + 32 * kb + 8 * g), bw, acc[t]);
			}
#pragma unroll
			for (uint32_t t = 0; t < NT; ++t) {
				const h4 hv = tr4(hj, 16 * t + 4 * g, 16 * w, LDA, lane);
				da[t] = act_backward4<GENERAL>(act, acc[t], hv);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) nxt[(16 * t + 4 * g + r) * LDA + 16 * w + lr] = da[t][r];
			}
			__syncthreads();
			half_t* tmp = cur;
			cur = nxt;
			nxt = tmp;
		}

		// ---- input matrix
		if (want_grads) {
			const h8 a_da = pack8(da[0], da[1]);
#pragma unroll
			for (uint32_t b = 0; b < NB_IN; ++b)
				accI[b] = mfma_16x16x32(a_da, pack8(*(const h4*)(xT + (16 * b + lr) * SPX + 4 * g), *(const h4*)(xT + (16 * b + lr) * SPX + 16 + 4 * g)), accI[b]);
		}
		half_t* dxT = nxt;  // [IN][SPX]
		if (want_dx) {
			if (w < NB_IN) {  // dX[s][k] = sum_jj dA_0[s][jj] M_in[jj][k]   (no activation on the network input)
				f4 acc[NT] = {zero4(), zero4()};
#pragma unroll
				for (uint32_t kb = 0; kb < KB; ++kb) {
#pragma unroll
					for (uint32_t t = 0; t < NT; ++t) acc[t] = mfma_16x16x32(*(const h8*)(cur + (16 * t + lr) * LDA + 32 * kb + 8 * g), wdx[kb], acc[t]);
				}
#pragma unroll
				for (uint32_t t = 0; t < NT; ++t)
					*(h4*)(dxT + (16 * w + lr) * SPX + 16 * t + 4 * g) = h4{(half_t)acc[t][0], (half_t)acc[t][1], (half_t)acc[t][2], (half_t)acc[t][3]};
			}
			__syncthreads();
			if (tid < N_CHUNKS) {
				const uint32_t k = tid / (S / 8), cc = tid % (S / 8);
				*(h8*)(dL_dinput + (size_t)k * n + (size_t)tile * S + 8 * cc) = *(const h8*)(dxT + k * SPX + 8 * cc);
			}
		}
		else {
			__syncthreads();  // the next tile's staging overwrites xT, which the dW_in products above read
		}
		// (with dL/dinput the barrier before the copy-out already separates those reads from the next staging; the copy-out itself
		// reads the idle ping-pong buffer, which the next tile writes only after further barriers)
	}

	// ---- this workgroup's share of the loss
	__syncthreads();
	if (block_sums) {
		float* red = (float*)hT;
		red[tid] = loss_sum;
		__syncthreads();
		for (uint32_t k = THREADS / 2; k > 0; k >>= 1) {
			if (tid < k) red[tid] += red[tid + k];
			__syncthreads();
		}
		if (tid == 0) block_sums[blockIdx.x] = red[0];
	}

	// ---- fp32 partial weight gradients of this workgroup, same layout as the parameters
	if (want_grads) {
		float* P = partials + (size_t)blockIdx.x * m.n_params();
#pragma unroll
		for (uint32_t b = 0; b < NB_IN; ++b)
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) P[(size_t)(16 * w + 4 * g + r) * IN + 16 * b + lr] = accI[b][r];
		const size_t off_hid = (size_t)WIDE * IN;
#pragma unroll
		for (uint32_t j = 0; j < HM; ++j)
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) P[off_hid + (size_t)j * WIDE * WIDE + (size_t)(16 * w + 4 * g + r) * WIDE + 16 * b + lr] = accH[j][b][r];
		const size_t off_out = off_hid + (size_t)HM * WIDE * WIDE;
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) P[off_out + (size_t)lr * WIDE + 16 * w + 4 * g + r] = accO[r];  // accO holds dW_out^T
	}
}

// ---------------------------------------------------------------------------------------------------------------------
static uint32_t mlp_train_wide_lds_bytes(const MlpMeta& m) {
	const uint32_t HM = m.n_hidden_matmuls;
	const uint32_t lda = wide_tile_ld(m.in_width / 32u);
	return (HM * WIDE * WIDE_LD + (HM + 3) * WIDE_S * lda + m.in_width * WIDE_SPX + WIDE_S * WIDE_LDY) * (uint32_t)sizeof(half_t);
}

bool mlp_train_wide_supported(const MlpMeta& m, uint32_t n) {
	return m.width == WIDE && m.padded_out == 16 && (m.in_width == 32 || m.in_width == 64) &&
	       m.n_hidden_matmuls <= MLP_MAX_HIDDEN_MATMULS_TRAIN && n % WIDE_S == 0 && mlp_train_wide_lds_bytes(m) <= 160u * 1024u;
}

uint32_t mlp_train_wide_n_partials(uint32_t n) {
#ifndef TCNN_MLP_WIDE_BLOCKS
#define TCNN_MLP_WIDE_BLOCKS 256  // one persistent workgroup per CU (its LDS holds the weights)
#endif
	const uint32_t n_tiles = n / WIDE_S;
	return n_tiles < TCNN_MLP_WIDE_BLOCKS ? n_tiles : TCNN_MLP_WIDE_BLOCKS;
}

template <uint32_t HM, uint32_t KB_IN>
static void launch_train_wide(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                              const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums) {
	const uint32_t lds_bytes = mlp_train_wide_lds_bytes(m);
	const uint32_t blocks = mlp_train_wide_n_partials(n);
	if (!act_is_simple(m.activation) || !act_is_simple(m.output_activation) || !(la.external_dL_doutput || loss_is_simple(la.type))) {
		TCNN_SET_MAX_DYN_LDS((k_mlp_train_wide<HM, KB_IN, true>), lds_bytes);
		TCNN_LAUNCH((k_mlp_train_wide<HM, KB_IN, true>), dim3(blocks), dim3(WIDE_THREADS), lds_bytes, stream, m, n, params, params_t, input, la, output,
		            dL_doutput, dL_dinput, partials, block_sums);
	} else {
		TCNN_SET_MAX_DYN_LDS((k_mlp_train_wide<HM, KB_IN, false>), lds_bytes);
		TCNN_LAUNCH((k_mlp_train_wide<HM, KB_IN, false>), dim3(blocks), dim3(WIDE_THREADS), lds_bytes, stream, m, n, params, params_t, input, la, output,
		            dL_doutput, dL_dinput, partials, block_sums);
	}
}

void mlp_train_wide(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                    const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums) {
	if (!mlp_train_wide_supported(m, n)) throw std::runtime_error("mlp_train_wide: unsupported network shape (check mlp_train_wide_supported first)");
	switch (m.n_hidden_matmuls * 10u + m.in_width / 32u) {
		case 1: launch_train_wide<0, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 2: launch_train_wide<0, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 11: launch_train_wide<1, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 12: launch_train_wide<1, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 21: launch_train_wide<2, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 22: launch_train_wide<2, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 31: launch_train_wide<3, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		case 32: launch_train_wide<3, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums); break;
		default: throw std::runtime_error("mlp_train_wide: no instance for this shape");
	}
}

}  // namespace tcnn_hip
