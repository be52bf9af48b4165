// mlp_train_wave.hip -- the register-resident training kernel of the fully fused MLP (one wavefront per strip of samples).
// Register budget: 256 per wave (two waves per SIMD).  At that budget the compiler keeps every MFMA result in VGPRs; with
// one wave per SIMD (512 registers) it routes them through AGPRs and v_accvgpr_read unless built with
// -mllvm -amdgpu-mfma-vgpr-form -- measured slower either way (profiles/r01_exp_notes.txt).
#include "mlp_kernels.h"

#include <stdlib.h>

#include <stdexcept>

namespace tcnn_hip {

// =============================================================================================
// fused training pass, one wavefront per strip of 32 samples, everything in registers.
//
// k_mlp_train (mlp_kernels.hip) shares a 64-sample tile between the waves of a workgroup: every layer costs each wave a read of
// the whole activation tile from LDS, two LDS copies of what it produced (both layouts) and a workgroup barrier, and
// the PMC counters show the LDS pipe, not MFMA or HBM, bounding it.  Here a wave owns its samples through all layers:
//   * an MFMA accumulator tile (row 4g+r, col lr) IS a 16x16x16 B operand (k = 4g+j, n = lr), and two of them whose
//     rows interleave (rows of block 2p: neurons 32p + 8g + {0..3}, of block 2p+1: 32p + 8g + {4..7}) are a 16x16x32
//     B operand in natural k order.  So layer l+1 consumes layer l's accumulators directly; the permutation lives in
//     which weight ROW each lane of the A operand holds (perm32) and costs nothing.
//   * the weight gradients contract over samples and need both factors with the samples in the operand's k slots,
//     i.e. the transposed tiles.  A 16x16 tile is transposed exactly by one MFMA against the identity
//     (D = A * I with the tile as A operand), no LDS round trip.
//   * the network input arrives feature-major: a 16-byte load per lane is already the "samples in k" operand of
//     dL/dW_in; one MFMA against a selection matrix turns it into the first layer's B operand.  dL/dinput is
//     produced sample-transposed (D = dA^T * W) so that it leaves as 16-byte feature-major stores.
// The weights (both orientations, pre-arranged as operands) are read-only in LDS; nothing a wave computes for its
// samples goes through LDS, which otherwise only serves the final reduction of the four waves' weight-gradient
// accumulators.  Same products and the same rounding points as k_mlp_forward / k_loss / k_mlp_backward: activations,
// outputs and dL/doutput are bit-identical; the output layer's backward MFMA holds the outputs in permuted k slots
// (out_perm) and the fp32 weight-gradient partial sums are grouped per wavefront, so dL/dinput and the weight
// gradients agree to the association order of fp32 sums (a last fp16 bit in ~1e-5 / ~1e-3 of the entries).
// =============================================================================================
TCNN_DEVICE uint32_t perm32(uint32_t b, uint32_t row) { return 32u * (b >> 1) + 8u * (row >> 2) + 4u * (b & 1u) + (row & 3u); }
// Output rows: accumulator row 4g+r of the output layer holds output 4r+g (a 4x4 transpose of the 16 padded outputs, done
// by the weight row each lane of the A operand holds).  The live outputs (< dims, usually 3 or 4) then sit in element
// r = 0 of ALL four lane groups instead of in four elements of one group: the loss is evaluated once per lane, not
// four times on a quarter of the lanes.
TCNN_DEVICE uint32_t out_perm(uint32_t row) { return 4u * (row & 3u) + (row >> 2); }
TCNN_DEVICE h4 to_h4(f4 v) { return h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }

// ReLU / None on packed halves (the GENERAL == false instances).  Same values as activation_device.h's scalar forms:
//   forward  (half)(x > 0 ? x : 0) == max((half)x, 0)   rounding is monotonic; "None" takes the maximum with -inf
//   backward (half)(h > 0 ? v : 0): the fp16 bits of (half)v ANDed with 0xFFFF where h != +-0 and with 0 elsewhere
//            (h >= 0 after ReLU; +0 for masked entries); "None" forces the mask to ones
typedef uint16_t us2 __attribute__((ext_vector_type(2)));
struct PackedAct {
	h2 floor2;            // forward: {0, 0} for ReLU, {-inf, -inf} for None
	uint32_t keep_bits;   // backward: 0 for ReLU, 0x00010001 for None
};
TCNN_DEVICE PackedAct packed_act(uint32_t act) {
	const half_t lo = act == (uint32_t)Activation::ReLU ? (half_t)0.0f : (half_t)-__builtin_inff();
	return {h2{lo, lo}, act == (uint32_t)Activation::ReLU ? 0u : 0x00010001u};
}
template <bool GENERAL>
TCNN_DEVICE h4 act_forward4(uint32_t act, const PackedAct& pa, f4 x) {
	if constexpr (GENERAL) {
		return h4{(half_t)act_forward<true>(act, x[0]), (half_t)act_forward<true>(act, x[1]), (half_t)act_forward<true>(act, x[2]), (half_t)act_forward<true>(act, x[3])};
	} else {
		const h2 a = __builtin_elementwise_max(h2{(half_t)x[0], (half_t)x[1]}, pa.floor2);
		const h2 b = __builtin_elementwise_max(h2{(half_t)x[2], (half_t)x[3]}, pa.floor2);
		return pack4(a, b);
	}
}
TCNN_DEVICE h2 relu_mask(h2 d, h2 forward_value, uint32_t keep_bits) {
	// three instructions per pair (v_and_or_b32: strip the sign, force "keep"; v_pk_min_u16; v_pk_mul_lo_u16) where the sign-extending form took
	// five: the sign-stripped forward value, as an unsigned 16-bit integer, is clamped to {0, 1} and multiplies the gradient's BITS
#if defined(TCNN_HOST_EMU)
	const uint32_t t = (__builtin_bit_cast(uint32_t, forward_value) & 0x7FFF7FFFu) | keep_bits;
	const us2 pass = __builtin_elementwise_min(__builtin_bit_cast(us2, t), us2{1, 1});
	return __builtin_bit_cast(h2, (us2)(__builtin_bit_cast(us2, d) * pass));
#else
	// (as instructions: told that the factor is 0 or 1, the compiler rewrites minimum and product as compares and selects on unpacked halves;
	// and it does not form v_and_or_b32 out of a literal and a scalar register -- one of the two has to be a vector register)
	uint32_t t, pass, masked;
	asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(t) : "v"(__builtin_bit_cast(uint32_t, forward_value)), "s"(0x7FFF7FFFu), "v"(keep_bits));
	asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(pass) : "v"(t));
	asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(masked) : "v"(__builtin_bit_cast(uint32_t, d)), "v"(pass));
	return __builtin_bit_cast(h2, masked);
#endif
}
template <bool GENERAL>
TCNN_DEVICE h4 act_backward4(uint32_t act, const PackedAct& pa, f4 v, h4 forward_value) {
	if constexpr (GENERAL) {
		return h4{(half_t)act_backward<true>(act, v[0], forward_value[0]), (half_t)act_backward<true>(act, v[1], forward_value[1]),
		          (half_t)act_backward<true>(act, v[2], forward_value[2]), (half_t)act_backward<true>(act, v[3], forward_value[3])};
	} else {
		const h4 d = to_h4(v);
		return pack4(relu_mask(__builtin_shufflevector(d, d, 0, 1), __builtin_shufflevector(forward_value, forward_value, 0, 1), pa.keep_bits),
		             relu_mask(__builtin_shufflevector(d, d, 2, 3), __builtin_shufflevector(forward_value, forward_value, 2, 3), pa.keep_bits));
	}
}

// (Relative)L2 gradient with the divisions that cannot change a bit left out: x / pdf when there is no pdf (x / 1), and the
// per-element loss VALUE, of which only the sum is ever used: the caller sums difference * gradient (= 2 n_total value up to
// one rounding) and scales once.  The gradient itself is evaluated exactly as loss_element() does.
// x / n_total is x * (1 / n_total) bit for bit when n_total is a power of two (the usual batch x outputs): no division then.
TCNN_DEVICE half_t loss_gradient_simple(bool relative, bool has_pdf, float prediction, float target, float pdf, float n_total, float inv_n_total_if_exact,
                                        float loss_scale, float& difference_times_gradient) {
	const float difference = prediction - target;
	float gradient = 2 * difference;
	if (relative) gradient = gradient / (prediction * prediction + 0.01f);
	if (has_pdf) gradient = gradient / pdf;
	difference_times_gradient = difference * gradient;
	if (inv_n_total_if_exact != 0.0f) return to_half_rn(loss_scale * gradient * inv_n_total_if_exact);
	return to_half_rn(loss_scale * gradient / n_total);
}

#ifndef TCNN_MLP_WAVE_BLOCKS
#define TCNN_MLP_WAVE_BLOCKS 512  // in units of four waves -- two per SIMD (256 registers per wave): the second wave of a SIMD fills the first one's stalls
#endif
#ifndef TCNN_MLP_WAVE_MIN_BLOCKS
#define TCNN_MLP_WAVE_MIN_BLOCKS 2
#endif
constexpr uint32_t MLP_WAVE_STRIP = 32, MLP_WAVE_THREADS = 256;
// Waves per training workgroup.  The instances that run two waves per SIMD put all EIGHT waves of a CU into one workgroup (round 6): the weights
// are staged once per CU instead of twice, and the launch leaves 256 weight-gradient slabs behind instead of 512 -- half the write burst at the
// kernel's end, half of what the optimizer's launch has to sum (AdamFinalize).  The instance that has a SIMD's registers to itself stays at four.
#ifndef TCNN_MLP_WAVE_WAVES
#define TCNN_MLP_WAVE_WAVES 8  // (4: the geometry of rounds 3-5, two workgroups per CU; kept as a build-time knob for A/B runs)
#endif
constexpr uint32_t mlp_train_wave_waves(uint32_t min_waves) { return min_waves >= 2u ? (uint32_t)TCNN_MLP_WAVE_WAVES : 4u; }

// EXTERNAL: no loss -- dL/doutput comes from la.external_dL_doutput (the backward pass of a module recomputing its forward pass).  A
// compile-time switch: the loss instance is at the register limit (256 of 256 at two waves per SIMD), a run-time branch around the loss spills.
// F32IN: the input is the caller's fp32 sample-major matrix (MlpF32Input: an unpadded Identity encoding folded into the strip loads).  A
// lane's fragment -- eight consecutive samples of one feature -- is eight 4-byte loads then (sample stride IN floats; the four lanes of a quad
// read 16 contiguous bytes, a strip's 8 KiB are fetched once and hit the L1 for the other fragment of the pair), held as fp32 until the
// strip is used so that the loads stay a whole strip ahead, converted with k_identity_forward's arithmetic, and stored once as the encoded
// matrix if the caller wants it for its context.
// FEW_DIMS: at most four (unpadded) outputs -- accumulator row 4g + r holds output 4r + g, so only element r = 0 of a lane carries a target, a
// loss and a pdf then.  A compile-time switch: the other three elements' loads, branches and registers (six target registers live through the
// whole forward pass at the 256-register limit) leave the instruction stream of the usual case; wider outputs run the FEW_DIMS = false instance.
template <uint32_t WIDTH, uint32_t IN, uint32_t HM, bool GENERAL, bool EXTERNAL, uint32_t MIN_WAVES = TCNN_MLP_WAVE_MIN_BLOCKS, bool F32IN = false, bool FEW_DIMS = true>
__global__ void __launch_bounds__(64 * mlp_train_wave_waves(MIN_WAVES), MIN_WAVES) k_mlp_train_wave(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params,
                                                                        const half_t* __restrict__ params_t, const half_t* __restrict__ input,
                                                                        const MlpLossArgs la, half_t* __restrict__ output, half_t* __restrict__ dL_doutput,
                                                                        half_t* __restrict__ dL_dinput, float* __restrict__ partials,
                                                                        float* __restrict__ block_sums, const MlpF32Input fin) {
	constexpr uint32_t NB = WIDTH / 16, NP = WIDTH / 32, FB = IN / 16, FP = IN / 32, NWAVES = mlp_train_wave_waves(MIN_WAVES), THREADS = 64 * NWAVES, HMX = HM > 0 ? HM : 1;
	constexpr uint32_t N_PARAMS = WIDTH * IN + HM * WIDTH * WIDTH + 16 * WIDTH;
	static_assert((NWAVES == 4 || NWAVES == 8) && WIDTH % 32 == 0 && IN % 32 == 0, "the final reduction halves the waves down to two; operands are built from pairs of 16-row tiles");
	constexpr uint32_t N_TILES = NB * FB + HM * NB * NB + NB;  // accumulator tiles per wave
	// Transposes through LDS: the weight-gradient MFMAs need their factors with the SAMPLES in the k slots.  A 16 x 16 tile as the
	// accumulators hold it (lane (g, lr): neurons 4g .. 4g+3 of sample lr) is one 8-byte LDS store per lane into a sample-major image, and
	// gfx950's transposing read (ds_read_b64_tr_b16, lds_read_tr4) hands lane (g, c) the four samples 4g .. 4g+3 of neuron c out of it --
	// exactly what the MFMA against the identity + two conversions produced, on the LDS pipe, which this kernel otherwise leaves idle,
	// instead of on the matrix and vector pipes, which bound it (34 of 126 MFMAs and 68 conversions per strip for the 64-neuron instance).
	// The tiles of a wave are its own: no barrier, the LDS executes a wave's requests in order.  Layer j's activations are stored as soon as
	// the forward pass has them; dL/d(pre-activation) of layer j reuses the region of layer j's activations once those have been read.
	constexpr uint32_t TR_TILES = (HM + 1) * 2 * NB + 2;  // per wave: every layer's activations + the two dL/doutput tiles, 512 bytes each
	constexpr uint32_t EXCH_F4 = (NWAVES / 2 - 1) * N_TILES * 64, TR_F4 = NWAVES * TR_TILES * 32;  // (the last exchange buffer is the weights' region)
	__shared__ f4 exchange[EXCH_F4 > TR_F4 ? EXCH_F4 : TR_F4];
	__shared__ float red[THREADS];
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t act = m.activation, out_act = m.output_activation;
	const bool want_grads = partials != nullptr, want_dx = dL_dinput != nullptr;
	const float n_total = (float)la.n_total;
	const PackedAct pa = packed_act(act);
	// (a template parameter, not a run-time test of la.external_dL_doutput: the run-time form takes the 64-neuron instance from 254
	// registers to 254 + 8 spilled AND puts a taken branch behind the output layer's last MFMA, profiles/r03_mfma_branch_hazard.txt)
	constexpr bool external = EXTERNAL;
	const bool relative = la.type == LossType::RelativeL2, has_pdf = la.data_pdf != nullptr;
	const float inv_n_total = (la.n_total & (la.n_total - 1u)) == 0u && la.n_total != 0u ? 1.0f / n_total : 0.0f;  // exact reciprocal or "divide"

	// ---- weights, both orientations, as MFMA operands (lane lr <-> the row/column perm32 assigns to it), staged once per
	// workgroup in LDS in lane order: a fragment is one conflict-free 16-byte read per lane wherever it is used
	const half_t* W_in = params;                                // [WIDTH][IN]
	const half_t* W_hid = W_in + (size_t)WIDTH * IN;            // HM x [WIDTH][WIDTH]
	const half_t* W_out = W_hid + (size_t)HM * WIDTH * WIDTH;   // [16][WIDTH]
	const half_t* wt_in = params_t;                             // [IN][WIDTH]
	const half_t* wt_hid = wt_in + (size_t)IN * WIDTH;          // HM x [WIDTH][WIDTH]
	const half_t* wt_out = wt_hid + (size_t)HM * WIDTH * WIDTH; // [WIDTH][16]
	constexpr uint32_t F_WINA = 0, F_WHIDA = F_WINA + NB * FP, F_WHIDT = F_WHIDA + HM * NB * NP, F_WOUTA = F_WHIDT + HM * NB * NP,
	                   F_WINB = F_WOUTA + NP, N_FRAG = F_WINB + FB * NP;
	constexpr uint32_t N_FRAG_LDS = N_FRAG > N_TILES ? N_FRAG : N_TILES;  // the region doubles as the second exchange buffer at the end
	__shared__ h8 wfrag[N_FRAG_LDS][64];
	__shared__ h4 wfrag_out_t[NB][64];
	// The first strip's input is requested before anything else and travels while the weights are staged; the wave's fragment loads
	// are all issued before the first one is written to LDS (one fragment at a time the loop paid a full L2 round trip per
	// fragment: 1.7 us of fixed cost per launch, profiles/r03_exp_notes.txt).
	const uint32_t n_strips = n / MLP_WAVE_STRIP, stride = gridDim.x * NWAVES;
	uint32_t strip = blockIdx.x * NWAVES + w;
	h8 xq_next[FB];
	float xraw_next[F32IN ? FB : 1][8];  // F32IN: the next strip's fragments as they arrive (fp32)
	auto request_strip = [&](uint32_t strip_) {
		if constexpr (F32IN) {
#pragma unroll
			for (uint32_t f = 0; f < FB; ++f)
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) xraw_next[f][j] = fin.x[(strip_ * MLP_WAVE_STRIP + 8 * g + j) * IN + perm32(f, lr)];  // (element offsets fit 32 bits: the host checks n)
		} else {
#pragma unroll
			for (uint32_t f = 0; f < FB; ++f) xq_next[f] = *(const h8*)(input + (perm32(f, lr) * n + strip_ * MLP_WAVE_STRIP + 8 * g));
		}
	};
	if (strip < n_strips) request_strip(strip);
	auto fragment_source = [&](uint32_t f) -> const half_t* {
		if (f < F_WHIDA) {
			const uint32_t b = (f - F_WINA) / FP, p = (f - F_WINA) % FP;
			return W_in + (size_t)perm32(b, lr) * IN + 32 * p + 8 * g;
		} else if (f < F_WOUTA) {
			const bool transposed = f >= F_WHIDT;
			const uint32_t e = f - (transposed ? F_WHIDT : F_WHIDA), j = e / (NB * NP), b = e / NP % NB, p = e % NP;
			return (transposed ? wt_hid : W_hid) + (size_t)j * WIDTH * WIDTH + (size_t)perm32(b, lr) * WIDTH + 32 * p + 8 * g;
		} else if (f < F_WINB) {
			return W_out + (size_t)out_perm(lr) * WIDTH + 32 * (f - F_WOUTA) + 8 * g;
		}
		const uint32_t b = (f - F_WINB) / NP, p = (f - F_WINB) % NP;
		return wt_in + (size_t)(16 * b + lr) * WIDTH + 32 * p + 8 * g;
	};
	{
		constexpr uint32_t PER_WAVE = (N_FRAG + NWAVES - 1) / NWAVES;
		constexpr uint32_t n_frag = N_FRAG;
		h8 staged[PER_WAVE];
#pragma unroll
		for (uint32_t k = 0; k < PER_WAVE; ++k) {
			const uint32_t f = w + k * NWAVES;
			staged[k] = *(const h8*)fragment_source(f < n_frag ? f : 0u);  // (an in-range address for the idle slots of the last round)
		}
#pragma unroll
		for (uint32_t k = 0; k < PER_WAVE; ++k) {
			const uint32_t f = w + k * NWAVES;
			if (f < n_frag) wfrag[f][lane] = staged[k];
		}
	}
	for (uint32_t b = w; b < NB; b += NWAVES) {  // k slot 4g+i of the output-layer backward <-> output 4i+g
		const half_t* row = wt_out + (size_t)perm32(b, lr) * 16;
		wfrag_out_t[b][lane] = h4{row[g], row[4 + g], row[8 + g], row[12 + g]};
	}
	// the two sample selections (16x16x32 B operand: column lr picks sample perm32(s, lr)), kept with the weights
	__shared__ h8 sel_frag[2][64];
	if (w < 2) {
		h8 v;
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) v[j] = (half_t)(8 * g + j == perm32(w, lr) ? 1.0f : 0.0f);
		sel_frag[w][lane] = v;
	}
	__syncthreads();
	auto winA = [&](uint32_t b, uint32_t p) { return wfrag[F_WINA + b * FP + p][lane]; };
	auto whidA = [&](uint32_t j, uint32_t b, uint32_t p) { return wfrag[F_WHIDA + (j * NB + b) * NP + p][lane]; };
	auto whidT = [&](uint32_t j, uint32_t b, uint32_t p) { return wfrag[F_WHIDT + (j * NB + b) * NP + p][lane]; };
	auto woutA = [&](uint32_t p) { return wfrag[F_WOUTA + p][lane]; };
	auto winB = [&](uint32_t b, uint32_t p) { return wfrag[F_WINB + b * NP + p][lane]; };
	auto woutT = [&](uint32_t b) { return wfrag_out_t[b][lane]; };
	half_t* const tr = (half_t*)exchange + w * (TR_TILES * 256u);
	// A tile's 64 words (sample s, neuron quad q; 8 bytes each) are laid out for the LDS banks, not row-major: the transposing read only
	// cares which word a lane addresses.  Word (s, q) at 8-byte slot 32 (s >> 3) + 8 ((q + (s >> 3)) & 3) + (s & 7): the 16 lanes of a
	// ds_write_b64 group (one q, all s) and the 32 lanes of a ds_read_b64_tr_b16 group (all q, eight s) each cover their bank window
	// exactly once (row-major, 32-byte rows: four-way conflicts on the stores -- 3.3 M conflict cycles per launch, 2 us slower than the MFMA form).
	auto tr_slot = [](uint32_t s_, uint32_t q_) { return 32u * (s_ >> 3) + 8u * ((q_ + (s_ >> 3)) & 3u) + (s_ & 7u); };
	const uint32_t tr_store_at = 4u * tr_slot(lr, g);                        // lane (g, lr) holds neurons 4g .. 4g+3 of sample lr
	const uint32_t tr_load_at = 4u * tr_slot(4u * g + (lr >> 2), lr & 3u);    // lds_read_tr4: lane i of group g addresses word (sample 4g + (i >> 2), quad i & 3)
	auto tr_put = [&](uint32_t tile, h4 v) { *(h4*)(tr + tile * 256u + tr_store_at) = v; };
	auto tr_get = [&](uint32_t tile) -> h4 { return lds_read_tr4(tr + tile * 256u + tr_load_at); };
	auto tile_h = [&](uint32_t layer, uint32_t s_, uint32_t b_) { return (layer * 2u + s_) * NB + b_; };
	constexpr uint32_t TILE_DY = (HM + 1) * 2 * NB;
	const PackedAct pa_out = packed_act(out_act);

	f4 accI[NB][FB], accH[HMX][NB][NB], accO[NB];
#pragma unroll
	for (uint32_t b = 0; b < NB; ++b) {
		accO[b] = zero4();
#pragma unroll
		for (uint32_t f = 0; f < FB; ++f) accI[b][f] = zero4();
#pragma unroll
		for (uint32_t j = 0; j < HMX; ++j)
#pragma unroll
			for (uint32_t i = 0; i < NB; ++i) accH[j][b][i] = zero4();
	}
	float loss_sum = 0.0f;

	for (; strip < n_strips; strip += stride) {
		asm volatile("" ::: "memory");  // the weight fragments are re-read from LDS where they are used, not hoisted into registers for the whole loop
		const uint32_t base = strip * MLP_WAVE_STRIP;  // element offsets fit 32 bits (the host checks n): scalar base + 32-bit lane offset addressing
		h8 xq[FB];  // lane lr <-> feature perm32(f, lr), k = sample 8g+j
		if constexpr (F32IN) {
#pragma unroll
			for (uint32_t f = 0; f < FB; ++f) {
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) {  // identity.h:60: (T)(x * scale + offset), the two roundings of k_identity_forward
					float t = xraw_next[f][j] * fin.scale;
					t = t + fin.offset;
					xq[f][j] = to_half_rn(t);
				}
				if (fin.enc_out) *(h8*)(fin.enc_out + (perm32(f, lr) * n + base + 8 * g)) = xq[f];
			}
		} else {
#pragma unroll
			for (uint32_t f = 0; f < FB; ++f) xq[f] = xq_next[f];
		}
		if (strip + stride < n_strips) request_strip(strip + stride);
		// this lane's targets: output 4r+g of sample perm32(s, lr)
		constexpr bool few_dims = FEW_DIMS;
		float tgt[2][4], pdf0[2];
		h4 gy_external[2];
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
			pdf0[s] = 1.0f;
			gy_external[s] = h4{};
			if constexpr (external) {  // one 8-byte load of outputs 4g .. 4g + 3 (transposed where it is used)
				gy_external[s] = *(const h4*)(la.external_dL_doutput + ((base + perm32(s, lr)) * 16 + 4 * g));
			} else if (has_pdf && g < la.dims) {
				pdf0[s] = la.data_pdf[(base + perm32(s, lr)) * la.dims + g];
			}
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) {
				const uint32_t dim = 4 * r + g;
				tgt[s][r] = 0.0f;
				if (!external && (r == 0 || !few_dims)) {
					const bool live = dim < la.dims;
					const uint32_t target_idx = (base + perm32(s, lr)) * la.dims + dim;
					tgt[s][r] = live ? la.targets[target_idx] : 0.0f;
				}
			}
		}

		// ================= forward =================
		// The weight fragments of a phase are requested from LDS at the start of the phase BEFORE (WF_AHEAD): read where they are used, every
		// phase opened with a wait on the LDS -- 46 % of a wave's cycles were spent parked in s_waitcnt (profiles/r06_exp_notes.txt).
		// (not in the wide-output instances: with four targets per lane live through the forward pass they are at the 256-register limit)
		constexpr bool WF_AHEAD = FEW_DIMS;
		h4 hp[HM + 1][2][NB];  // layer, sample block, neuron block: (neuron perm32(b, 4g+r), sample perm32(s, lr))
		h8 w_hid[NB][NP];      // the next hidden layer's fragments (forward: whidA, backward: whidT), a phase ahead
		h8 w_out[NP];
		{
			h8 w_in[NB][FP];
			if constexpr (WF_AHEAD) {
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
					for (uint32_t p = 0; p < FP; ++p) w_in[b][p] = winA(b, p);
			}
			h8 xb[2][FP];  // first layer's B operand: k = feature 32p + 8g + j, n = sample perm32(s, lr)
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s) {
				const h8 sel = sel_frag[s][lane];
#pragma unroll
				for (uint32_t p = 0; p < FP; ++p)
					xb[s][p] = pack8(to_h4(mfma_16x16x32(xq[2 * p], sel, zero4())), to_h4(mfma_16x16x32(xq[2 * p + 1], sel, zero4())));
			}
			sched_fence();
			if constexpr (WF_AHEAD) {
				if constexpr (HM > 0) {
#pragma unroll
					for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
						for (uint32_t p = 0; p < NP; ++p) w_hid[b][p] = whidA(0, b, p);
				} else {
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) w_out[p] = woutA(p);
				}
			}
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < FP; ++p) acc = mfma_16x16x32(WF_AHEAD ? w_in[b][p] : winA(b, p), xb[s][p], acc);
					hp[0][s][b] = act_forward4<GENERAL>(act, pa, acc);
					tr_put(tile_h(0, s, b), hp[0][s][b]);
				}
			sched_fence();
		}
#pragma unroll
		for (uint32_t j = 0; j < HM; ++j) {
			h8 w_cur[NB][NP];
			if constexpr (WF_AHEAD) {
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) w_cur[b][p] = w_hid[b][p];
				if (j + 1 < HM) {
#pragma unroll
					for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
						for (uint32_t p = 0; p < NP; ++p) w_hid[b][p] = whidA(j + 1, b, p);
				} else {
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) w_out[p] = woutA(p);
				}
			}
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(WF_AHEAD ? w_cur[b][p] : whidA(j, b, p), pack8(hp[j][s][2 * p], hp[j][s][2 * p + 1]), acc);
					hp[j + 1][s][b] = act_forward4<GENERAL>(act, pa, acc);
					tr_put(tile_h(j + 1, s, b), hp[j + 1][s][b]);
				}
			sched_fence();
		}
		// the backward pass's first fragments travel during the output layer and the loss
		h4 w_out_t[NB];
		if constexpr (WF_AHEAD) {
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) w_out_t[b] = woutT(b);
			if constexpr (HM > 0) {
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) w_hid[b][p] = whidT(HM - 1, b, p);
			}
		}
		// ---- output layer + loss: (output 4r+g, sample perm32(s, lr))
		// Every load of the strip has landed BEFORE its first store is issued, and no load follows the stores.  gfx9 counts loads and stores
		// in ONE counter (vmcnt), and a register a load MAY still be writing -- on any path, taken or not: the targets of outputs 4 .. 15, the
		// pdf -- is only reused behind "s_waitcnt vmcnt(0)".  With the loss of sample block 1 (and the backward pass's first register reuse)
		// behind sample block 0's stores, those waits sat out the stores' round trip to HBM twice per strip: 47 % of a wave's cycles were spent
		// parked in s_waitcnt, whatever the instruction count (profiles/r06_exp_notes.txt).
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
#if !defined(TCNN_HOST_EMU)
			asm volatile("" : "+v"(pdf0[s]));
			if constexpr (external) asm volatile("" : "+v"(gy_external[s]));
#pragma unroll
			for (uint32_t r = 0; r < 4; ++r) asm volatile("" : "+v"(tgt[s][r]));
#endif
		}
		h4 dyp[2], o_[2], gy_[2];
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
			f4 acc = zero4();
#pragma unroll
			for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(WF_AHEAD ? w_out[p] : woutA(p), pack8(hp[HM][s][2 * p], hp[HM][s][2 * p + 1]), acc);
			o_[s] = act_forward4<GENERAL>(out_act, pa_out, acc);
		}
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
			const h4 o = o_[s];
			const uint32_t i = base + perm32(s, lr);
			h4 gy;
			if (external) {  // the same 4 x 4 transpose as the stores below (it is its own inverse)
				gy = wave_rows_transpose4(gy_external[s]);
			} else {
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) {
					const uint32_t dim = 4 * r + g;
					gy[r] = (half_t)0.0f;
					if ((r == 0 || !few_dims) && dim < la.dims) {  // relative_l2.h:57-61: padding outputs carry no loss
						float pdf = pdf0[s];
						if (r > 0 && has_pdf) pdf = la.data_pdf[i * la.dims + dim];  // (more than four outputs with a pdf: rare, fetched where it is used)
						float value;
						if constexpr (GENERAL) gy[r] = loss_element<true>(la.type, (float)o[r], tgt[s][r], pdf, n_total, la.loss_scale, value);
						else gy[r] = loss_gradient_simple(relative, has_pdf, (float)o[r], tgt[s][r], pdf, n_total, inv_n_total, la.loss_scale, value);
						loss_sum += value;
					}
				}
			}
			gy_[s] = gy;
			// fully_fused_mlp.cu:760-763 (an fp16 value through fp32 and back is itself: the packed mask is the scalar form's select)
			if constexpr (GENERAL) {
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) dyp[s][r] = (half_t)act_backward<GENERAL>(out_act, (float)gy[r], o[r]);
			} else {
				dyp[s] = pack4(relu_mask(__builtin_shufflevector(gy, gy, 0, 1), __builtin_shufflevector(o, o, 0, 1), pa_out.keep_bits),
				               relu_mask(__builtin_shufflevector(gy, gy, 2, 3), __builtin_shufflevector(o, o, 2, 3), pa_out.keep_bits));
			}
			tr_put(TILE_DY + s, dyp[s]);
		}
		sched_fence();
		// Lane (g, lr) holds outputs 4r + g of its sample.  Stored as they lie that is four 2-byte stores per matrix whose 64 lanes
		// touch 64 different 32-byte sectors -- measured: 6 us of the kernel's 34 (profiles/r03_exp_notes.txt).  A 4 x 4 transpose
		// over the four lane groups (register moves, no LDS) gives every lane outputs 4g .. 4g + 3: ONE 8-byte store.
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
			const uint32_t i = base + perm32(s, lr);
			if (output) *(h4*)(output + (i * 16 + 4 * g)) = wave_rows_transpose4(o_[s]);
			if (dL_doutput) *(h4*)(dL_doutput + (i * 16 + 4 * g)) = wave_rows_transpose4(gy_[s]);
		}

		sched_fence();
		// ================= backward =================
		wave_lds_sync();  // this wave's tiles are in LDS
		{  // dW_out[output][neuron] += dY * H_last^T  (accumulated unconditionally: a branch around the MFMAs costs the in-place accumulators)
			const h8 dyq = pack8(tr_get(TILE_DY), tr_get(TILE_DY + 1));
			h8 hq[NB];  // the layer's activations with the samples in k: lane lr <-> neuron perm32(b, lr), k = sample 8g+j
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) hq[b] = pack8(tr_get(tile_h(HM, 0, b)), tr_get(tile_h(HM, 1, b)));
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) accO[b] = mfma_16x16x32(dyq, hq[b], accO[b]);
		}
		h4 dap[2][NB];  // dL/d(pre-activation) of the current layer, same tile layout as hp
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) dap[s][b] = act_backward4<GENERAL>(act, pa, mfma_16x16x16(WF_AHEAD ? w_out_t[b] : woutT(b), dyp[s], zero4()), hp[HM][s][b]);
		wave_lds_sync();  // the activations' region has been read: it takes the layer's dL/d(pre-activation)
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) tr_put(tile_h(HM, s, b), dap[s][b]);
		sched_fence();
#pragma unroll
		for (int j = (int)HM - 1; j >= 0; --j) {
			// the previous layer's dL/d(pre-activation) first (it needs nothing from LDS): the tiles just stored travel meanwhile
			h4 prev[2][NB];
			h8 w_cur[NB][NP];
			if constexpr (WF_AHEAD) {
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) w_cur[b][p] = w_hid[b][p];
				if (j > 0) {
#pragma unroll
					for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
						for (uint32_t p = 0; p < NP; ++p) w_hid[b][p] = whidT(j - 1, b, p);
				}
			}
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(WF_AHEAD ? w_cur[b][p] : whidT(j, b, p), pack8(dap[s][2 * p], dap[s][2 * p + 1]), acc);
					prev[s][b] = act_backward4<GENERAL>(act, pa, acc, hp[j][s][b]);
				}
			sched_fence();
			wave_lds_sync();
			{  // dW_hid_j[out][in] += dA_{j+1} * H_j^T
				h8 daq[NB], hq[NB];
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					daq[b] = pack8(tr_get(tile_h(j + 1, 0, b)), tr_get(tile_h(j + 1, 1, b)));
					hq[b] = pack8(tr_get(tile_h(j, 0, b)), tr_get(tile_h(j, 1, b)));
				}
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
					for (uint32_t i = 0; i < NB; ++i) accH[j][b][i] = mfma_16x16x32(daq[b], hq[i], accH[j][b][i]);
			}
			wave_lds_sync();
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					dap[s][b] = prev[s][b];
					tr_put(tile_h(j, s, b), dap[s][b]);
				}
			sched_fence();
		}
		if (want_dx) {  // dX^T = dA_0^T * W_in (below) needs nothing from LDS either: ahead of the last transposes
#pragma unroll
			for (uint32_t f = 0; f < FB; ++f) {
				h4 d[2];
#pragma unroll
				for (uint32_t s = 0; s < 2; ++s) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(pack8(dap[s][2 * p], dap[s][2 * p + 1]), winB(f, p), acc);
					d[s] = to_h4(acc);
				}
				*(h8*)(dL_dinput + ((16 * f + lr) * n + base + 8 * g)) = pack8(d[0], d[1]);
			}
		}
		sched_fence();
		wave_lds_sync();
		{  // dW_in[out][feature] += dA_0 * X^T
			h8 daq[NB];
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) daq[b] = pack8(tr_get(tile_h(0, 0, b)), tr_get(tile_h(0, 1, b)));
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b)
#pragma unroll
				for (uint32_t f = 0; f < FB; ++f) accI[b][f] = mfma_16x16x32(daq[b], xq[f], accI[b][f]);
		}
		wave_lds_sync();  // (the next strip's forward pass overwrites the tiles)
		sched_fence();
	}

	// ---- this workgroup's share of the loss
	if (block_sums) {
		if constexpr (!GENERAL) loss_sum = loss_sum * 0.5f / n_total;  // see loss_gradient_simple
		red[tid] = loss_sum;
		__syncthreads();
		for (uint32_t k = THREADS / 2; k > 0; k >>= 1) {
			if (tid < k) red[tid] += red[tid + k];
			__syncthreads();
		}
		if (tid == 0) block_sums[blockIdx.x] = red[0];
	}

	// ---- fp32 partial weight gradients: the waves are halved down to two -- w += w + NWAVES/2, ..., w += w + 2, i.e. ((0+4)+(2+6)) + ((1+5)+(3+7))
	// for eight waves, (0+2) + (1+3) for four -- exchanged through LDS in register order (tile t of lane l at [t][l]: conflict-free 16-byte
	// accesses; all waves share the lane <-> element map).  The last round splits the tiles between waves 0 and 1, and each writes its half of
	// the workgroup's slab IN REGISTER ORDER (slab position 256 t + 4 lane + r: one coalesced 16-byte store per tile; the summation knows the
	// position -> parameter map, mlp_wave_slab_param).  The parameter-layout slab of rounds 1-2 was 112 scattered 4-byte store instructions
	// issued by wave 0 alone: 3.7 us of the kernel's fixed cost (profiles/r03_exp_notes.txt).
	if (want_grads) {
		auto for_each_tile = [&](auto&& fn) {
			uint32_t t = 0;
#pragma unroll
			for (uint32_t b = 0; b < NB; ++b) {
#pragma unroll
				for (uint32_t f = 0; f < FB; ++f) fn(t++, accI[b][f]);
#pragma unroll
				for (uint32_t j = 0; j < HM; ++j)
#pragma unroll
					for (uint32_t i = 0; i < NB; ++i) fn(t++, accH[j][b][i]);
				fn(t++, accO[b]);
			}
		};
		constexpr uint32_t HALF_TILES = N_TILES / 2;  // wave 0 finishes tiles [0, HALF_TILES), wave 1 the rest
		constexpr uint32_t N_BUF = NWAVES / 2;
		auto buffer = [&](uint32_t k) -> f4* { return k + 1 < N_BUF ? (f4*)exchange + k * (N_TILES * 64u) : (f4*)wfrag; };  // (the weights are not needed any more)
		f4* ex0 = buffer(0);
		f4* ex1 = buffer(N_BUF - 1);
#pragma unroll
		for (uint32_t half = NWAVES / 2; half >= 2; half >>= 1) {
			__syncthreads();
			if (w >= half && w < 2 * half) {
				f4* ex = buffer(w - half);
				for_each_tile([&](uint32_t t, f4& a) { ex[t * 64 + lane] = a; });
			}
			__syncthreads();
			if (w < half) {
				const f4* ex = buffer(w);
				for_each_tile([&](uint32_t t, f4& a) { a += ex[t * 64 + lane]; });
			}
		}
		__syncthreads();
		if (w == 1) for_each_tile([&](uint32_t t, f4& a) { if (t < HALF_TILES) ex0[t * 64 + lane] = a; });
		if (w == 0) for_each_tile([&](uint32_t t, f4& a) { if (t >= HALF_TILES) ex1[t * 64 + lane] = a; });
		__syncthreads();
		float* P = partials + (size_t)blockIdx.x * N_PARAMS;
		if (w == 0) {
			for_each_tile([&](uint32_t t, f4& a) {
				if (t < HALF_TILES) *(f4*)(P + (t * 256u + lane * 4u)) = a + ex0[t * 64 + lane];
			});
		} else if (w == 1) {
			for_each_tile([&](uint32_t t, f4& a) {
				if (t >= HALF_TILES) *(f4*)(P + (t * 256u + lane * 4u)) = ex1[t * 64 + lane] + a;  // (even waves) + (odd waves) here too
			});
		}
	}
}

// =============================================================================================
// inference (Network::inference_mixed_precision_impl, fully_fused_mlp.cu:680-708): the forward half of the kernel above.
// A wave chains the layers of its 32 samples in registers; the only memory traffic is the encoded input in and the
// padded output out.  Natural output rows (accumulator row 4g+r = output 4g+r): 8-byte stores.
// =============================================================================================
#ifndef TCNN_MLP_INFER_BLOCKS
#define TCNN_MLP_INFER_BLOCKS 1024  // four workgroups per CU; the instance needs < 128 registers and 8-12 KiB of LDS
#endif
// F32IN: the input is the caller's fp32 sample-major matrix (MlpF32Input).  Inference needs the input in ONE orientation only -- the first
// layer's B operand, eight consecutive features of a sample per lane -- which is 32 contiguous bytes of that matrix: two 16-byte loads, the
// Identity encoding's arithmetic, and neither the encoding kernel nor the selection MFMAs that turn the feature-major fragments around.
template <uint32_t WIDTH, uint32_t IN, uint32_t HM, bool F32IN = false>
__global__ void __launch_bounds__(MLP_WAVE_THREADS) k_mlp_infer_wave(const MlpMeta m, const uint32_t n, const half_t* __restrict__ params,
                                                                     const half_t* __restrict__ input, half_t* __restrict__ output, const MlpF32Output f32,
                                                                     const MlpF32Input fin) {
	constexpr uint32_t NB = WIDTH / 16, NP = WIDTH / 32, FB = IN / 16, FP = IN / 32, NWAVES = MLP_WAVE_THREADS / 64;
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, lr = lane & 15u, g = lane >> 4;
	const uint32_t out_act = m.output_activation;
	const PackedAct pa = packed_act(m.activation);
	const half_t* W_in = params;
	const half_t* W_hid = W_in + (size_t)WIDTH * IN;
	const half_t* W_out = W_hid + (size_t)HM * WIDTH * WIDTH;
	constexpr uint32_t F_WINA = 0, F_WHIDA = F_WINA + NB * FP, F_WOUTA = F_WHIDA + HM * NB * NP, F_SEL = F_WOUTA + NP, N_FRAG = F_SEL + 2;
	__shared__ h8 wfrag[N_FRAG][64];
	for (uint32_t f = w; f < N_FRAG; f += NWAVES) {
		h8 v;
		if (f < F_WHIDA) {
			const uint32_t b = f / FP, p = f % FP;
			v = *(const h8*)(W_in + (size_t)perm32(b, lr) * IN + 32 * p + 8 * g);
		} else if (f < F_WOUTA) {
			const uint32_t e = f - F_WHIDA, j = e / (NB * NP), b = e / NP % NB, p = e % NP;
			v = *(const h8*)(W_hid + (size_t)j * WIDTH * WIDTH + (size_t)perm32(b, lr) * WIDTH + 32 * p + 8 * g);
		} else if (f < F_SEL) {
			v = *(const h8*)(W_out + (size_t)lr * WIDTH + 32 * (f - F_WOUTA) + 8 * g);
		} else {
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) v[j] = (half_t)(8 * g + j == perm32(f - F_SEL, lr) ? 1.0f : 0.0f);
		}
		wfrag[f][lane] = v;
	}
	__syncthreads();

	const uint32_t n_strips = n / MLP_WAVE_STRIP, stride = gridDim.x * NWAVES;
	for (uint32_t strip = blockIdx.x * NWAVES + w; strip < n_strips; strip += stride) {
		const uint32_t base = strip * MLP_WAVE_STRIP;
		h4 hp[2][NB];
		{
			h8 xb[2][FP];  // first layer's B operand: k = feature 32p + 8g + j, n = sample perm32(s, lr)
			if constexpr (F32IN) {
#pragma unroll
				for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
					for (uint32_t p = 0; p < FP; ++p) {
						const float* src = fin.x + ((base + perm32(s, lr)) * IN + 32 * p + 8 * g);  // (element offsets fit 32 bits: the host checks n)
						const f4 lo = *(const f4*)src, hi = *(const f4*)(src + 4);
#pragma unroll
						for (uint32_t j = 0; j < 8; ++j) {  // identity.h:60: (T)(x * scale + offset), the two roundings of k_identity_forward
							float t = (j < 4 ? lo[j & 3u] : hi[j & 3u]) * fin.scale;
							t = t + fin.offset;
							xb[s][p][j] = to_half_rn(t);
						}
					}
			} else {
				h8 xq[FB];
#pragma unroll
				for (uint32_t f = 0; f < FB; ++f) xq[f] = *(const h8*)(input + (perm32(f, lr) * n + base + 8 * g));
#pragma unroll
				for (uint32_t s = 0; s < 2; ++s) {
					const h8 sel = wfrag[F_SEL + s][lane];
#pragma unroll
					for (uint32_t p = 0; p < FP; ++p)
						xb[s][p] = pack8(to_h4(mfma_16x16x32(xq[2 * p], sel, zero4())), to_h4(mfma_16x16x32(xq[2 * p + 1], sel, zero4())));
				}
			}
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < FP; ++p) acc = mfma_16x16x32(wfrag[F_WINA + b * FP + p][lane], xb[s][p], acc);
					hp[s][b] = act_forward4<false>(m.activation, pa, acc);
				}
		}
#pragma unroll
		for (uint32_t j = 0; j < HM; ++j) {
			h4 nx[2][NB];
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) {
					f4 acc = zero4();
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(wfrag[F_WHIDA + (j * NB + b) * NP + p][lane], pack8(hp[s][2 * p], hp[s][2 * p + 1]), acc);
					nx[s][b] = act_forward4<false>(m.activation, pa, acc);
				}
#pragma unroll
			for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
				for (uint32_t b = 0; b < NB; ++b) hp[s][b] = nx[s][b];
		}
#pragma unroll
		for (uint32_t s = 0; s < 2; ++s) {
			f4 acc = zero4();
#pragma unroll
			for (uint32_t p = 0; p < NP; ++p) acc = mfma_16x16x32(wfrag[F_WOUTA + p][lane], pack8(hp[s][2 * p], hp[s][2 * p + 1]), acc);
			const h4 o = h4{(half_t)act_forward<false>(out_act, acc[0]), (half_t)act_forward<false>(out_act, acc[1]), (half_t)act_forward<false>(out_act, acc[2]),
			                (half_t)act_forward<false>(out_act, acc[3])};
			if (f32.out) {  // the caller's fp32 matrix straight from the registers (the same exact conversion trim_and_cast does)
				const size_t at = (size_t)(base + perm32(s, lr)) * f32.stride_i;
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r)
					if (4 * g + r < f32.dims) f32.out[at + (size_t)(4 * g + r) * f32.stride_j] = (float)o[r];
			} else {
				*(h4*)(output + ((base + perm32(s, lr)) * 16 + 4 * g)) = o;
			}
		}
	}
}

// ---- the register-resident wave-per-strip variant: instantiated for the shapes whose operands fit one wave's registers
bool mlp_train_wave_supported(const MlpMeta& m, uint32_t n, LossType loss) {
	if (m.padded_out != 16 || (m.in_width != 32 && m.in_width != 64)) return false;
	// ReLU / None and (Relative)L2 only: the instances with out-of-line activation / loss calls gain nothing here
	if (!act_is_simple(m.activation) || !act_is_simple(m.output_activation) || !loss_is_simple(loss)) return false;
	if (n > (1u << 26)) return false;  // 32-bit element offsets inside the kernel
	// 64 inputs with two hidden layers (the benchmarks/mlp shape, BASELINE configs[1]): the 144 fp32 weight-gradient accumulators per
	// lane spill at two waves per SIMD (0.077 vs 0.068 ms for the workgroup-tiled kernel, profiles/r02_exp_notes.txt); that instance is
	// built for ONE wave per SIMD instead (VGPRs + AGPRs holding the accumulators, no spill: profiles/r03_exp_notes.txt)
	if (m.in_width == 64) return m.width == 64 && m.n_hidden_matmuls <= 1;
	return (m.width == 64 && m.n_hidden_matmuls <= 1) || (m.width == 32 && m.n_hidden_matmuls <= 2);
}

// The 64-input, two-hidden-layer instance has a SIMD to itself (its 144 accumulators): ONE workgroup per CU is all that is resident, so 256 of
// them are one generation -- with 512 the second generation pays the weight staging, the ramp and the 144-register reduction again
// (mlp_train_fused stage at N = 2^18: 0.0646 -> 0.0562 ms; carrying 64 instead of 32 samples per wave and iteration into the same
// accumulators, for more independent MFMA -> convert chains in the one wave, measured 0.0591 at 256 and 0.0683 at 512 blocks: not kept)
#ifndef TCNN_MLP_WAVE_WIDE_BLOCKS
#define TCNN_MLP_WAVE_WIDE_BLOCKS 256
#endif
uint32_t mlp_train_wave_n_partials(const MlpMeta& m, uint32_t n) {
	const bool one_wave_per_simd = m.in_width == 64 && m.width == 64 && m.n_hidden_matmuls == 1;
	const uint32_t n_waves = mlp_train_wave_waves(one_wave_per_simd ? 1u : TCNN_MLP_WAVE_MIN_BLOCKS);
	const uint32_t wanted = div_round_up(n / MLP_WAVE_STRIP, n_waves);
	const uint32_t most = one_wave_per_simd ? TCNN_MLP_WAVE_WIDE_BLOCKS : div_round_up((uint32_t)TCNN_MLP_WAVE_BLOCKS * 4u, n_waves);
	return wanted < most ? wanted : most;
}

template <uint32_t WIDTH, uint32_t IN, uint32_t HM, uint32_t MIN_WAVES = TCNN_MLP_WAVE_MIN_BLOCKS>
static void launch_train_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                              const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums,
                              const MlpF32Input* f32_input) {
	const uint32_t blocks = mlp_train_wave_n_partials(m, n);
	const MlpF32Input fin = f32_input ? *f32_input : MlpF32Input();
	const bool few = la.external_dL_doutput != nullptr || la.dims <= 4u;  // (no loss, no targets with an external dL/doutput)
#define TCNN_WAVE_LAUNCH(EXTERNAL_, F32IN_, FEW_)                                                                                                              \
	TCNN_LAUNCH((k_mlp_train_wave<WIDTH, IN, HM, false, EXTERNAL_, MIN_WAVES, F32IN_, FEW_>), dim3(blocks), dim3(64 * mlp_train_wave_waves(MIN_WAVES)), 0, stream, m, n, params, params_t, \
	            input, la, output, dL_doutput, dL_dinput, partials, block_sums, fin)
	if (f32_input) {
		if constexpr (IN == 64) {  // the instances mlp_train_f32_input_supported names
			if (few) TCNN_WAVE_LAUNCH(false, true, true); else TCNN_WAVE_LAUNCH(false, true, false);
		} else {
			throw std::runtime_error("mlp_train_wave: no fp32-input instance for this shape");
		}
	} else if (la.external_dL_doutput) {
		TCNN_WAVE_LAUNCH(true, false, true);
	} else if (few) {
		TCNN_WAVE_LAUNCH(false, false, true);
	} else {
		TCNN_WAVE_LAUNCH(false, false, false);
	}
#undef TCNN_WAVE_LAUNCH
}
// fp32 sample-major input (MlpF32Input): the 64-input instances -- two hidden layers (the benchmarks/mlp shape, which has a SIMD's registers to
// itself) and one hidden layer (246 registers at two waves per SIMD with a strip of fp32 fragments in flight, no scratch); the 32-input
// instances serve grid encodings and stay as they are
bool mlp_train_f32_input_supported(const MlpMeta& m, uint32_t n, LossType loss) {
	return m.in_width == 64 && m.width == 64 && m.n_hidden_matmuls <= 1 && n <= (1u << 25) && mlp_train_wave_supported(m, n, loss);
}

void mlp_train_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                    const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums,
                    const MlpF32Input* f32_input) {
	if (!mlp_train_wave_supported(m, n, la.external_dL_doutput ? LossType::L2 : la.type)) throw std::runtime_error("mlp_train_wave: unsupported shape, activation or loss (check mlp_train_wave_supported first)");
	const uint32_t key = (m.in_width == 64 ? 10000u : 0u) + m.width * 10u + m.n_hidden_matmuls;
	switch (key) {
		case 10640: launch_train_wave<64, 64, 0>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		case 10641: launch_train_wave<64, 64, 1, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;  // one wave per SIMD
		case 640: launch_train_wave<64, 32, 0>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		case 641: launch_train_wave<64, 32, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		case 320: launch_train_wave<32, 32, 0>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		case 321: launch_train_wave<32, 32, 1>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		case 322: launch_train_wave<32, 32, 2>(stream, m, n, params, params_t, input, la, output, dL_doutput, dL_dinput, partials, block_sums, f32_input); break;
		default: throw std::runtime_error("mlp_train: no register-resident instance for this shape");
	}
}

// ---- inference instances
bool mlp_infer_wave_supported(const MlpMeta& m, uint32_t n) {
	if (m.padded_out != 16 || (m.in_width != 32 && m.in_width != 64) || n > (1u << 25)) return false;
	if (!act_is_simple(m.activation) || !act_is_simple(m.output_activation)) return false;
	return (m.width == 64 && m.n_hidden_matmuls <= 2) || (m.width == 32 && m.n_hidden_matmuls <= 3 && m.in_width == 32);
}

template <uint32_t WIDTH, uint32_t IN, uint32_t HM>
static void launch_infer_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* output, const MlpF32Output& f32,
                              const MlpF32Input* f32_input) {
	const uint32_t wanted = div_round_up(n / MLP_WAVE_STRIP, MLP_WAVE_THREADS / 64u);
	const uint32_t blocks = wanted < TCNN_MLP_INFER_BLOCKS ? wanted : TCNN_MLP_INFER_BLOCKS;
	if (f32_input) {
		TCNN_LAUNCH((k_mlp_infer_wave<WIDTH, IN, HM, true>), dim3(blocks), dim3(MLP_WAVE_THREADS), 0, stream, m, n, params, input, output, f32, *f32_input);
	} else {
		TCNN_LAUNCH((k_mlp_infer_wave<WIDTH, IN, HM, false>), dim3(blocks), dim3(MLP_WAVE_THREADS), 0, stream, m, n, params, input, output, f32, MlpF32Input());
	}
}

// every inference instance reads an fp32 sample-major input (16-byte aligned rows: in_width is 32 or 64)
bool mlp_infer_f32_input_supported(const MlpMeta& m, uint32_t n) {
	return n <= (1u << 25) && mlp_infer_wave_supported(m, n);
}

void mlp_infer_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* output, const MlpF32Output& f32,
                    const MlpF32Input* f32_input) {
	if (!mlp_infer_wave_supported(m, n)) throw std::runtime_error("mlp_infer_wave: unsupported shape or activation (check mlp_infer_wave_supported first)");
	if (f32_input && (!mlp_infer_f32_input_supported(m, n) || ((uintptr_t)f32_input->x & 15u) != 0u)) throw std::runtime_error("mlp_infer_wave: the fp32 input needs 16-byte alignment and n <= 2^25");
	switch (m.width * 1000u + m.in_width * 10u + m.n_hidden_matmuls) {
		case 64320: launch_infer_wave<64, 32, 0>(stream, m, n, params, input, output, f32, f32_input); break;
		case 64321: launch_infer_wave<64, 32, 1>(stream, m, n, params, input, output, f32, f32_input); break;
		case 64322: launch_infer_wave<64, 32, 2>(stream, m, n, params, input, output, f32, f32_input); break;
		case 64640: launch_infer_wave<64, 64, 0>(stream, m, n, params, input, output, f32, f32_input); break;
		case 64641: launch_infer_wave<64, 64, 1>(stream, m, n, params, input, output, f32, f32_input); break;
		case 64642: launch_infer_wave<64, 64, 2>(stream, m, n, params, input, output, f32, f32_input); break;
		case 32320: launch_infer_wave<32, 32, 0>(stream, m, n, params, input, output, f32, f32_input); break;
		case 32321: launch_infer_wave<32, 32, 1>(stream, m, n, params, input, output, f32, f32_input); break;
		case 32322: launch_infer_wave<32, 32, 2>(stream, m, n, params, input, output, f32, f32_input); break;
		case 32323: launch_infer_wave<32, 32, 3>(stream, m, n, params, input, output, f32, f32_input); break;
		default: throw std::runtime_error("mlp_infer_wave: no instance for this shape");
	}
}

}  // namespace tcnn_hip
