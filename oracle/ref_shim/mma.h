/* Host stand-in for <mma.h> (nvcuda::wmma): TEST INFRASTRUCTURE (oracle/_ref build only; see oracle/build_ref.py).
 * Lets the reference's fully fused network kernels (src/fully_fused_mlp.cu) be compiled for the CPU so that the restated oracle's
 * network passes can be pinned against the reference's own code: which weights a layer multiplies with, which operand is transposed,
 * where the activation and its transfer sit, what is written to the intermediate / output buffers and in which layout.
 *
 * Model of a 16x16x16 fragment operation:
 *   - every lane of the warp holds the WHOLE 16x16 tile, in logical (row, column) order x[16 * r + c] for all three fragment kinds
 *     (matrix_a: r = m, c = k;  matrix_b: r = k, c = n;  accumulator: r = m, c = n), so the element-wise code the reference runs on
 *     fragments (warp_activation, and warp_activation_backward between an accumulator and a matrix_a fragment,
 *     fully_fused_mlp.cu:107-108, 222-225) pairs the elements it means to pair; on the device the tile is spread over the lanes,
 *     which changes who holds an element, not its value;
 *   - mma_sync: every product a * b of two binary16 values is exact in binary32; d(r, c) = round_to_accumulator_type(c(r, c) +
 *     sum_k a(r, k) * b(k, c)) with the sum taken in binary32 in ascending k.  The tensor core's internal summation order and width
 *     are not specified by the PTX ISA beyond "at least single precision for the products"; this is the same model the oracle's
 *     fp16-accumulate mode states (tcnn_oracle.c, layer_fwd): ONE rounding to binary16 per 16-deep operation.
 * Lanes run one after the other (fibers, oracle/ref_driver.cpp), so the *_sync operations need no communication. */
#pragma once
#include <cuda_fp16.h>

namespace nvcuda {
namespace wmma {

struct matrix_a {};
struct matrix_b {};
struct accumulator {};
struct row_major {};
struct col_major {};
enum layout_t { mem_row_major, mem_col_major };

template <typename Use, int M, int N, int K, typename T, typename Layout = void>
struct fragment {
	static_assert(M == 16 && N == 16 && K == 16, "only the 16x16x16 shape is modelled");
	static constexpr int num_elements = 256;
	T x[256];
};

template <typename Use, typename T, typename Layout, typename V>
inline void fill_fragment(fragment<Use, 16, 16, 16, T, Layout>& f, const V& v) {
	for (int t = 0; t < 256; ++t) f.x[t] = (T)v;
}

/* matrix_a / matrix_b: the layout is part of the fragment type; element (r, c) at p[r * ldm + c] (row_major) or p[c * ldm + r] */
template <typename Use, typename T, typename Layout>
inline void load_matrix_sync(fragment<Use, 16, 16, 16, T, Layout>& f, const T* p, unsigned ldm) {
	static_assert(!std::is_same<Use, accumulator>::value, "accumulator loads name their layout at run time");
	constexpr bool rm = std::is_same<Layout, row_major>::value;
	/* threadblock_layer<BACKWARD> loads `activation_aux` tiles even when it was handed nullptr (the dL/dinput layer,
	 * fully_fused_mlp.cu:107, 258): with Activation::None the tile is never used and a device compiler drops the dead load; here
	 * the load happens, so a source inside the null page yields a tile of NaNs (any use of it would show in the results) */
	if ((uintptr_t)p < (uintptr_t)(1u << 20)) {
		std::memset((void*)f.x, 0xff, sizeof(f.x));
		return;
	}
	for (unsigned r = 0; r < 16; ++r) {
		for (unsigned c = 0; c < 16; ++c) f.x[16 * r + c] = rm ? p[r * ldm + c] : p[c * ldm + r];
	}
}
template <typename T>
inline void load_matrix_sync(fragment<accumulator, 16, 16, 16, T>& f, const T* p, unsigned ldm, layout_t layout) {
	for (unsigned r = 0; r < 16; ++r) {
		for (unsigned c = 0; c < 16; ++c) f.x[16 * r + c] = layout == mem_row_major ? p[r * ldm + c] : p[c * ldm + r];
	}
}
template <typename T>
inline void store_matrix_sync(T* p, const fragment<accumulator, 16, 16, 16, T>& f, unsigned ldm, layout_t layout) {
	for (unsigned r = 0; r < 16; ++r) {
		for (unsigned c = 0; c < 16; ++c) (layout == mem_row_major ? p[r * ldm + c] : p[c * ldm + r]) = f.x[16 * r + c];
	}
}

template <typename T, typename LA, typename LB>
inline void mma_sync(fragment<accumulator, 16, 16, 16, T>& d, const fragment<matrix_a, 16, 16, 16, __half, LA>& a,
                     const fragment<matrix_b, 16, 16, 16, __half, LB>& b, const fragment<accumulator, 16, 16, 16, T>& c) {
	float af[256], bf[256];
	for (int t = 0; t < 256; ++t) {
		af[t] = (float)a.x[t];
		bf[t] = (float)b.x[t];
	}
	for (int r = 0; r < 16; ++r) {
		for (int col = 0; col < 16; ++col) {
			float acc = (float)c.x[16 * r + col];
			for (int k = 0; k < 16; ++k) acc += af[16 * r + k] * bf[16 * k + col];
			d.x[16 * r + col] = (T)acc;
		}
	}
}

}  // namespace wmma
}  // namespace nvcuda
