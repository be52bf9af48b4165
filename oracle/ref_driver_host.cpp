// ref_driver_host.cpp -- extern "C" face of the reference's HOST-side pieces of the path, compiled where they lie (oracle/build_ref.py; third
// translation unit of oracle/_ref/libtcnn_ref.so).  TEST INFRASTRUCTURE, like the rest of oracle/_ref.
//   * GridEncodingTemplated's constructor (encodings/grid.h:673-737): the per-level resolutions and the parameter offset table;
//   * GPUMatrix::initialize_uniform / initialize_xavier_uniform / initialize_siren_uniform[_first] (gpu_matrix.h:275-375) and
//     FullyFusedMLP::initialize_params (src/fully_fused_mlp.cu:868-893): which matrices are drawn, in which order, from which range.
// These are member functions of classes that need the CUDA runtime; build_ref.py extracts the member DEFINITIONS by name and they are
// compiled here inside minimal stand-ins for their classes: only the members those bodies touch exist, `cudaMemcpy` is memcpy (the
// "device" buffer is the caller's host array), CHECK_THROW / CUDA_CHECK_THROW evaluate their argument, fmt::format formats nothing.
#include <tiny-cuda-nn/common.h>
#define asm
#define volatile(...) ((void)0)
#include <tiny-cuda-nn/common_device.h>
#undef asm
#undef volatile
#include <pcg32/pcg32.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CHECK_THROW(x) do { if (!(x)) throw std::runtime_error(#x " failed"); } while (0)
#define CUDA_CHECK_THROW(x) (void)(x)
enum ref_memcpy_kind { cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost };
static inline int cudaMemcpy(void* dst, const void* src, size_t n, ref_memcpy_kind) { std::memcpy(dst, src, n); return 0; }
namespace fmt {
template <typename... A> std::string format(const char* f, A&&...) { return f; }
}  // namespace fmt

namespace tcnn {

template <typename... A> void log_debug(const char*, A&&...) {}
static inline std::string to_string(GridType) { return "GridType"; }

#include "ref_extracted_host_functions.inc"  // powi (common_host.h:361-368), MAX_N_LEVELS / ParamsOffsetTable (multi_level_interface.h:84-88)

// ---- gpu_matrix.h: the members the initialisers touch (gpu_matrix.h:106-253)
template <typename T, MatrixLayout _layout = MatrixLayout::ColumnMajor>
class GPUMatrix {
public:
	GPUMatrix(T* data, uint32_t m, uint32_t n) : m_data{data}, m_rows{m}, m_cols{n} {}
	T* data() const { return m_data; }
	bool is_contiguous() const { return true; }
	uint32_t m() const { return m_rows; }
	uint32_t n() const { return m_cols; }
	uint32_t n_elements() const { return m_rows * m_cols; }
	size_t n_bytes() const { return (size_t)n_elements() * sizeof(T); }
	uint32_t fan_out() const { return m_rows; }  // gpu_matrix.h:235
	uint32_t fan_in() const { return m_cols; }   // gpu_matrix.h:239
#include "ref_extracted_matrix_init.inc"
private:
	T* m_data;
	uint32_t m_rows, m_cols;
};

// ---- networks/fully_fused_mlp.h: the members initialize_params touches
template <typename T, uint32_t WIDTH>
class FullyFusedMLP {
public:
	void initialize_params(pcg32& rnd, float* params_full_precision, float scale = 1);
	uint32_t m_n_hidden_layers, m_n_hidden_matmuls, m_input_width, m_network_width, m_output_width, m_padded_output_width;
	Activation m_activation, m_output_activation;
};
#include "ref_extracted_mlp_init.inc"

// ---- encodings/grid.h: the members the constructor touches (grid.h:1115-1135)
template <typename T, uint32_t N_POS_DIMS, uint32_t N_FEATURES_PER_LEVEL>
class GridEncodingTemplated {
public:
#include "ref_extracted_grid_ctor.inc"
	uint32_t m_n_features, m_n_levels, m_n_params;
	ParamsOffsetTable m_offset_table;
	uint32_t m_log2_hashmap_size, m_base_resolution;
	uint32_t m_n_output_dims;
	float m_per_level_scale;
	bool m_stochastic_interpolation;
	InterpolationType m_interpolation_type;
	GridType m_grid_type;
	bool m_fixed_point_pos;
};

}  // namespace tcnn

using namespace tcnn;

namespace {
template <uint32_t D, uint32_t F>
int grid_table(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, int grid_type, uint32_t* offsets, uint32_t* n_params) {
	GridEncodingTemplated<__half, D, F> g(n_levels * F, log2_hashmap_size, base_resolution, per_level_scale, false, InterpolationType::Linear, (GridType)grid_type, false);
	for (uint32_t l = 0; l <= g.m_n_levels; ++l) offsets[l] = g.m_offset_table.data[l];
	*n_params = g.m_n_params;
	return (int)g.m_n_levels;
}
}  // namespace

extern "C" {

// GridEncodingTemplated<T, D, F>(n_features = n_levels * F, ...) -> offsets[0 .. n_levels], n_params; returns n_levels, -1 for an instance
// that does not exist, -2 when the constructor throws (grid.h:697-699, 719, 734-736)
int ref_grid_offset_table(uint32_t n_dims, uint32_t n_feat, uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, int grid_type,
                          uint32_t* offsets, uint32_t* n_params) {
	try {
#define CASE(D_, F_) if (n_dims == D_ && n_feat == F_) return grid_table<D_, F_>(n_levels, log2_hashmap_size, base_resolution, per_level_scale, grid_type, offsets, n_params);
		CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(2, 8) CASE(3, 1) CASE(3, 2) CASE(3, 4) CASE(3, 8) CASE(4, 1) CASE(4, 2) CASE(4, 4) CASE(4, 8)
#undef CASE
	} catch (...) {
		return -2;
	}
	return -1;
}

// FullyFusedMLP<T, WIDTH>::initialize_params(rnd, params_full_precision, scale): the generator continues from `*state` / `*inc` and its
// state is handed back (the encoding's parameters are drawn next from the same generator, network_with_input_encoding.h:124-130)
int ref_mlp_initialize_params(uint32_t input_width, uint32_t network_width, uint32_t padded_output_width, uint32_t n_hidden_layers, int activation, uint64_t* state,
                              uint64_t* inc, float* params_full_precision, float scale) {
	FullyFusedMLP<__half, 64> net;  // WIDTH is not used by initialize_params: one instance serves all widths
	net.m_input_width = input_width;
	net.m_network_width = network_width;
	net.m_padded_output_width = padded_output_width;
	net.m_n_hidden_layers = n_hidden_layers;
	net.m_n_hidden_matmuls = n_hidden_layers - 1;
	net.m_activation = (Activation)activation;
	pcg32 rnd;
	rnd.state = *state;
	rnd.inc = *inc;
	net.initialize_params(rnd, params_full_precision, scale);
	*state = rnd.state;
	*inc = rnd.inc;
	return 0;
}
// pcg32{seed}: the state pair the two functions above continue from (pcg32.h:60-70)
void ref_pcg32_seed(uint64_t seed, uint64_t* state, uint64_t* inc) {
	pcg32 rnd{seed};
	*state = rnd.state;
	*inc = rnd.inc;
}
// GPUMatrix<float>::initialize_uniform(rnd, low, high), gpu_matrix.h:275-290
void ref_matrix_initialize_uniform(uint64_t* state, uint64_t* inc, uint32_t n, float* data, float low, float high) {
	pcg32 rnd;
	rnd.state = *state;
	rnd.inc = *inc;
	GPUMatrix<float>(data, n, 1).initialize_uniform(rnd, low, high);
	*state = rnd.state;
	*inc = rnd.inc;
}

}  // extern "C"
