// ref_driver_mlp.cpp -- extern "C" face of the REFERENCE'S OWN fully fused network kernels compiled for the host
// (oracle/build_ref.py; second translation unit of oracle/_ref/libtcnn_ref.so).
// TEST INFRASTRUCTURE: nothing under tiny-cuda-nn_amd/ links, loads or calls this; tests/test_oracle_ref.py uses it to pin the
// restated oracle's network passes (orc_mlp_forward / orc_mlp_backward_ex in their fp16-accumulate mode) bit for bit.
//
// The kernel bodies (kernel_mlp_fused, kernel_mlp_fused_backward and the threadblock_* device functions they call) come from
// /root/reference/src/fully_fused_mlp.cu where they lie (ref_extracted_mlp.inc exists in a temporary directory for the duration of the
// compile).  OURS here: (1) the launch shapes of mlp_fused_forward / mlp_fused_backward (cited per wrapper), (2) the execution model of
// one thread block -- its 32 x WIDTH/16 threads are FIBERS (ucontext) of one host thread; __syncthreads() switches to the next fiber,
// a round ends when every live fiber has arrived at a barrier, so between two barriers the threads run one after the other in
// ascending (threadIdx.y, threadIdx.x) order (any order is a legal schedule of a race-free kernel), (3) nvcuda::wmma as
// oracle/ref_shim/mma.h states it, (4) the block's dynamic shared memory: one host array, poisoned (binary16 NaN pattern 0x7e00 would
// be legal data, so 0xffff -- a NaN no kernel produces) before every block.
// What is NOT the reference here: the tensor core's arithmetic inside one 16x16x16 operation (mma.h says how it is modelled) and the
// CUTLASS GEMMs of the weight gradients / >16-wide output layers / input gradients of narrow inputs (not in /root/reference).
#include <tiny-cuda-nn/common.h>
#define asm
#define volatile(...) ((void)0)
#include <tiny-cuda-nn/common_device.h>
#undef asm
#undef volatile
#include <mma.h>

#include <ucontext.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace tcnn {
// `extern __shared__ __half shmem[]` (fully_fused_mlp.cu:176, 516): the largest launch the wrappers below make needs
// (16 + 128) * (128 + 8) halves (forward, WIDTH 128) or (128 + 16) * (in_width + 8) with in_width <= 1024
__half shmem[160 * 1040];
}  // namespace tcnn

#include "ref_extracted_mlp.inc"

using namespace tcnn;

namespace {

struct BlockFibers {
	static constexpr size_t STACK_BYTES = 1u << 20;
	ucontext_t main_ctx;
	std::vector<ucontext_t> ctx;
	std::vector<char> stacks;
	std::vector<char> finished;
	std::function<void()> body;
	unsigned cur = 0;
	static BlockFibers* active;

	static void entry() {
		BlockFibers* b = active;
		b->body();
		b->finished[b->cur] = 1;  // uc_link returns to main_ctx
	}
	static void yield() {
		BlockFibers* b = active;
		swapcontext(&b->ctx[b->cur], &b->main_ctx);
	}
	// -> false when some threads of the block ended while others waited at a barrier (undefined behaviour on the device)
	bool run(unsigned tx, unsigned ty, std::function<void()> f) {
		const unsigned n = tx * ty;
		body = std::move(f);
		ctx.assign(n, ucontext_t{});
		stacks.resize((size_t)n * STACK_BYTES);
		finished.assign(n, 0);
		for (unsigned t = 0; t < n; ++t) {
			getcontext(&ctx[t]);
			ctx[t].uc_stack.ss_sp = stacks.data() + (size_t)t * STACK_BYTES;
			ctx[t].uc_stack.ss_size = STACK_BYTES;
			ctx[t].uc_link = &main_ctx;
			makecontext(&ctx[t], (void (*)())entry, 0);
		}
		active = this;
		ref_sync_hook = &yield;
		blockDim.x = tx;
		blockDim.y = ty;
		bool ok = true;
		for (;;) {
			unsigned waiting = 0, ended = 0;
			for (unsigned t = 0; t < n; ++t) {
				if (finished[t]) continue;
				cur = t;
				threadIdx.x = t % tx;
				threadIdx.y = t / tx;
				swapcontext(&main_ctx, &ctx[t]);
				if (finished[t]) ++ended; else ++waiting;
			}
			if (waiting == 0) break;
			if (ended != 0) {
				ok = false;
				break;
			}
		}
		ref_sync_hook = nullptr;
		active = nullptr;
		threadIdx.y = 0;
		blockDim.y = 1;
		return ok;
	}
};
BlockFibers* BlockFibers::active = nullptr;

template <typename F>
bool launch_blocks(uint32_t n_blocks, uint32_t tx, uint32_t ty, F&& body) {
	BlockFibers fibers;
	gridDim.x = n_blocks;
	gridDim.y = 1;
	for (uint32_t b = 0; b < n_blocks; ++b) {
		blockIdx.x = b;
		blockIdx.y = 0;
		std::memset((void*)shmem, 0xff, sizeof(shmem));
		if (!fibers.run(tx, ty, body)) return false;
	}
	return true;
}

struct FwdArgs {
	int inference;
	Activation output_activation;
	const __half* input;
	const __half* weights;
	__half* out_intermediate;
	__half* out;
	uint32_t output_stride, batch_size, in_width, out_width, n_hidden_matmuls;
	nvcuda::wmma::layout_t input_layout, output_layout;
};
// mlp_fused_forward, fully_fused_mlp.cu:583-640: N_ITERS = WIDTH >= 256 ? 2 : 8, threads (32, WIDTH / 16), batch / (16 * N_ITERS) blocks,
// batch % (16 * N_ITERS) == 0 and in_width % 16 == 0 are CHECK_THROWs there
template <uint32_t WIDTH, Activation ACT>
int mlp_forward(const FwdArgs& a) {
	constexpr uint32_t N_ITERS = WIDTH >= 256 ? 2 : 8;
	if (a.batch_size % (16 * N_ITERS) != 0 || a.in_width % 16 != 0 || a.in_width > 1024) return 2;
	const uint32_t n_blocks = a.batch_size / (16 * N_ITERS);
	bool ok;
	if (a.inference) {
		ok = launch_blocks(n_blocks, 32, WIDTH / 16, [&] {
			kernel_mlp_fused<WIDTH, N_ITERS, __half, ACT, true>(a.output_activation, a.input, a.weights, a.out_intermediate, a.out, a.output_stride, a.batch_size, a.in_width,
			                                                     a.out_width, a.n_hidden_matmuls, a.input_layout, a.output_layout);
		});
	} else {
		ok = launch_blocks(n_blocks, 32, WIDTH / 16, [&] {
			kernel_mlp_fused<WIDTH, N_ITERS, __half, ACT, false>(a.output_activation, a.input, a.weights, a.out_intermediate, a.out, a.output_stride, a.batch_size, a.in_width,
			                                                      a.out_width, a.n_hidden_matmuls, a.input_layout, a.output_layout);
		});
	}
	return ok ? 0 : 3;
}

struct BwdArgs {
	const __half* dL_doutput;
	const __half* weights;
	__half* out_intermediate;
	const __half* forward;
	__half* dL_dinput;
	const __half* weights_first_layer;
	uint32_t output_stride, batch_size, out_width, n_hidden_matmuls;
	int dL_doutput_row_major;
};
// mlp_fused_backward, fully_fused_mlp.cu:275-313: a row-major (RM) dL_doutput matrix selects the wmma::col_major instance and vice versa
template <uint32_t WIDTH, Activation ACT>
int mlp_backward(const BwdArgs& a) {
	constexpr uint32_t N_ITERS = WIDTH >= 256 ? 2 : 8;
	if (a.batch_size % (16 * N_ITERS) != 0) return 2;
	const uint32_t n_blocks = a.batch_size / (16 * N_ITERS);
	bool ok;
	if (a.dL_doutput_row_major) {
		ok = launch_blocks(n_blocks, 32, WIDTH / 16, [&] {
			kernel_mlp_fused_backward<WIDTH, N_ITERS, ACT, nvcuda::wmma::col_major>(a.dL_doutput, a.weights, a.out_intermediate, a.forward, a.dL_dinput, a.weights_first_layer,
			                                                                        a.output_stride, a.batch_size, a.out_width, a.n_hidden_matmuls);
		});
	} else {
		ok = launch_blocks(n_blocks, 32, WIDTH / 16, [&] {
			kernel_mlp_fused_backward<WIDTH, N_ITERS, ACT, nvcuda::wmma::row_major>(a.dL_doutput, a.weights, a.out_intermediate, a.forward, a.dL_dinput, a.weights_first_layer,
			                                                                        a.output_stride, a.batch_size, a.out_width, a.n_hidden_matmuls);
		});
	}
	return ok ? 0 : 3;
}

// the activations FullyFusedMLP dispatches on (fully_fused_mlp.cu:689-699, 712-722, 793-803); anything else throws there
template <uint32_t WIDTH, typename Fn>
int dispatch_activation(int activation, Fn&& fn) {
	switch ((Activation)activation) {
		case Activation::None: return fn(std::integral_constant<Activation, Activation::None>{});
		case Activation::Exponential: return fn(std::integral_constant<Activation, Activation::Exponential>{});
		case Activation::Sigmoid: return fn(std::integral_constant<Activation, Activation::Sigmoid>{});
		case Activation::ReLU: return fn(std::integral_constant<Activation, Activation::ReLU>{});
		case Activation::LeakyReLU: return fn(std::integral_constant<Activation, Activation::LeakyReLU>{});
		case Activation::Squareplus: return fn(std::integral_constant<Activation, Activation::Squareplus>{});
		case Activation::Softplus: return fn(std::integral_constant<Activation, Activation::Softplus>{});
		case Activation::Tanh: return fn(std::integral_constant<Activation, Activation::Tanh>{});
		default: return 1;
	}
}
template <typename Fn>
int dispatch_width(uint32_t width, Fn&& fn) {
	switch (width) {  // the instances at the end of fully_fused_mlp.cu
		case 16: return fn(std::integral_constant<uint32_t, 16>{});
		case 32: return fn(std::integral_constant<uint32_t, 32>{});
		case 64: return fn(std::integral_constant<uint32_t, 64>{});
		case 128: return fn(std::integral_constant<uint32_t, 128>{});
		default: return 1;
	}
}

}  // namespace

extern "C" {

// FullyFusedMLP::forward_impl / inference_mixed_precision_impl -> mlp_fused_forward -> kernel_mlp_fused (fully_fused_mlp.cu:499-557).
// `weights`: all weight matrices, contiguous, row-major [out][in] (input matrix first).  `input`: in_width x batch, column-major
// (input_row_major = 0; the kernel's mem_row_major) or row-major (1).  `out_intermediate`: [n_hidden][batch][width] (forward only).
// `out`: padded_out x batch with `output_stride`, column-major (output_row_major = 0) or row-major.  out_width = rows of `out` (<= 16 for
// the fused last layer).  Returns 0, 1 (no such instance), 2 (a CHECK_THROW of the host function), 3 (threads diverged at a barrier).
int ref_mlp_fused_forward(uint32_t width, int activation, int output_activation, int inference, const void* input, int input_row_major, const void* weights,
                          void* out_intermediate, void* out, uint32_t output_stride, int output_row_major, uint32_t batch_size, uint32_t in_width, uint32_t out_width,
                          uint32_t n_hidden_layers) {
	using namespace nvcuda::wmma;
	FwdArgs a = {inference, (Activation)output_activation, (const __half*)input, (const __half*)weights, (__half*)out_intermediate, (__half*)out,
	             out ? output_stride : 0u, batch_size, in_width, out ? out_width : 0u, n_hidden_layers - 1,
	             input_row_major ? mem_col_major : mem_row_major, out && output_row_major ? mem_col_major : mem_row_major};
	return dispatch_width(width, [&](auto w) {
		return dispatch_activation<decltype(w)::value>(activation, [&](auto act) { return mlp_forward<decltype(w)::value, decltype(act)::value>(a); });
	});
}

// FullyFusedMLP::backward_impl -> mlp_fused_backward -> kernel_mlp_fused_backward (fully_fused_mlp.cu:152-260).
// `weights_first_layer`: the input matrix; `weights`: weight_matrix_at(0), the first hidden-to-hidden matrix (the output matrix follows
// the n_hidden_matmuls hidden ones); `forward`: forward.hidden.at(0), [n_hidden][batch][width]; `backward_tmp`: same shape, entry k =
// dL/d(pre-activation) of hidden layer n_hidden - 1 - k; dL_dinput (width x batch, column-major) only when in_width == width
// (fully_fused_mlp.cu:788); dL_doutput: out_width x batch with stride, after the output activation's transfer.
int ref_mlp_fused_backward(uint32_t width, int activation, const void* dL_doutput, int dL_doutput_row_major, uint32_t output_stride, const void* weights_first_layer,
                           const void* weights, void* backward_tmp, const void* forward, void* dL_dinput, uint32_t batch_size, uint32_t out_width, uint32_t n_hidden_layers) {
	BwdArgs a = {(const __half*)dL_doutput, (const __half*)weights, (__half*)backward_tmp, (const __half*)forward, (__half*)dL_dinput, (const __half*)weights_first_layer,
	             output_stride, batch_size, out_width, n_hidden_layers - 1, dL_doutput_row_major};
	return dispatch_width(width, [&](auto w) {
		return dispatch_activation<decltype(w)::value>(activation, [&](auto act) { return mlp_backward<decltype(w)::value, decltype(act)::value>(a); });
	});
}

}  // extern "C"
