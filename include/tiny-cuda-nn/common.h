/*
 * tiny-cuda-nn/common.h -- vocabulary of the C++ facade over the C ABI (include/tcnn_hip.h, libtcnn_hip.so).
 *
 * The headers under include/tiny-cuda-nn/ carry the names of the reference headers they stand in for (common.h,
 * gpu_memory.h, gpu_matrix.h, random.h, loss.h, optimizer.h, network_with_input_encoding.h, trainer.h, config.h) and
 * mirror the part of each that the hot path's callers use, so that host code written like the reference's
 * samples/mlp_learning_an_image.cu:214-311 compiles against them with `hip*` in place of `cuda*`.  Header-only; no
 * kernels or numerics live here -- everything forwards to the C ABI.  Errors surface as std::runtime_error carrying
 * tcnn_last_error(), as in the reference (common_host.h:71-110).
 *
 * Mirrors: common.h:152-176 (GradientMode, MatrixLayout), :243-247 (loss scale, granularity), common_host.h:71-110
 * (CHECK_THROW), :180-200 (linear_kernel), cpp_api.h:62-85 (free functions).
 */
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>  // a caller with device code of its own (linear_kernel below)
#else
#include <hip/hip_runtime_api.h>
#endif
#include <tcnn_hip.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

/* nlohmann::json, under the include path the reference uses (dependencies/json -> <json/json.hpp>), the upstream package
 * path, or a path named by the build (-DTCNN_JSON_HEADER='"/path/json.hpp"').  Without any of them `tcnn::json` is the
 * small value class of json_mini.h (parse / dump / value(key, default) / operator[] / contains / object()). */
#if defined(TCNN_JSON_HEADER)
#include TCNN_JSON_HEADER
#define TCNN_HAS_NLOHMANN_JSON 1
#elif defined(__has_include)
#if __has_include(<json/json.hpp>)
#include <json/json.hpp>
#define TCNN_HAS_NLOHMANN_JSON 1
#elif __has_include(<nlohmann/json.hpp>)
#include <nlohmann/json.hpp>
#define TCNN_HAS_NLOHMANN_JSON 1
#endif
#endif
#if !defined(TCNN_HAS_NLOHMANN_JSON)
#include <tiny-cuda-nn/json_mini.h>
#elif defined(NLOHMANN_JSON_VERSION_MAJOR) && (NLOHMANN_JSON_VERSION_MAJOR > 3 || (NLOHMANN_JSON_VERSION_MAJOR == 3 && NLOHMANN_JSON_VERSION_MINOR >= 8))
#define TCNN_JSON_HAS_BINARY 1 /* json::binary_t, MessagePack bin: the snapshot document can be a json value (trainer.h: serialize) */
#endif

namespace tcnn {

#if defined(TCNN_HAS_NLOHMANN_JSON)
using json = nlohmann::json;
inline std::string json_text(const json& j) { return j.dump(); }
#else
using json = tcnn_hip::Json;
inline std::string json_text(const json& j) { return j.dump(); }
#endif

#define TCNN_STR_(x) #x
#define TCNN_STR(x) TCNN_STR_(x)
#define CHECK_THROW(x) \
	do { if (!(x)) throw std::runtime_error(std::string(__FILE__ ":" TCNN_STR(__LINE__) " check failed " #x)); } while (0)
#define HIP_CHECK_THROW(x)                                                                                                          \
	do {                                                                                                                            \
		hipError_t tcnn_e_ = (x);                                                                                                   \
		if (tcnn_e_ != hipSuccess) throw std::runtime_error(std::string(__FILE__ ":" TCNN_STR(__LINE__) " " #x " failed: ") + hipGetErrorString(tcnn_e_)); \
	} while (0)
#define CUDA_CHECK_THROW(x) HIP_CHECK_THROW(x) /* the reference's spelling, for ported callers */

inline void check(int rc) {
	if (rc != TCNN_OK) throw std::runtime_error(tcnn_last_error());
}
inline void hip_check(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

constexpr uint32_t BATCH_SIZE_GRANULARITY = 256;  // common.h:246
constexpr uint32_t N_THREADS_LINEAR = 128;        // common.h:247
inline uint32_t batch_size_granularity() { return tcnn_batch_size_granularity(); }
template <typename T> constexpr T div_round_up(T v, T d) { return (v + d - 1) / d; }
template <typename T> constexpr T next_multiple(T v, T d) { return div_round_up(v, d) * d; }

enum class GradientMode { Ignore = TCNN_GRADIENT_IGNORE, Overwrite = TCNN_GRADIENT_OVERWRITE, Accumulate = TCNN_GRADIENT_ACCUMULATE };  // common.h:152-156
enum class MatrixLayout { RowMajor = 0, SoA = 0, ColumnMajor = 1, AoS = 1 };                                                            // common.h:166-176
static constexpr MatrixLayout RM = MatrixLayout::RowMajor;
static constexpr MatrixLayout SoA = MatrixLayout::SoA;
static constexpr MatrixLayout CM = MatrixLayout::ColumnMajor;
static constexpr MatrixLayout AoS = MatrixLayout::AoS;

// IEEE binary16 as it crosses the boundary (the reference's __half): storage only on the host side.
struct half {
	uint16_t x = 0;
	half() = default;
	explicit half(float f) {
		uint32_t u;
		std::memcpy(&u, &f, 4);
		const uint32_t sign = (u >> 16) & 0x8000u;
		const int32_t e = (int32_t)((u >> 23) & 0xFF) - 127 + 15;
		uint32_t m = u & 0x7FFFFFu;
		if (((u >> 23) & 0xFF) == 0xFF) { x = (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0)); return; }
		if (e >= 31) { x = (uint16_t)(sign | 0x7C00u); return; }
		if (e <= 0) {
			if (e < -10) { x = (uint16_t)sign; return; }
			m |= 0x800000u;
			const uint32_t shift = (uint32_t)(14 - e), r = m >> shift, rem = m & ((1u << shift) - 1u), half_ulp = 1u << (shift - 1);
			x = (uint16_t)(sign | (r + ((rem > half_ulp || (rem == half_ulp && (r & 1u))) ? 1u : 0u)));
			return;
		}
		const uint32_t r = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1FFFu;
		x = (uint16_t)(sign | (r + ((rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ? 1u : 0u)));
	}
	explicit operator float() const {
		const uint32_t sign = (uint32_t)(x & 0x8000u) << 16, e = (x >> 10) & 0x1Fu, m = x & 0x3FFu;
		uint32_t u;
		if (e == 0) {
			if (m == 0) { u = sign; } else {
				int sh = 0;
				uint32_t mm = m;
				while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
				u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FFu) << 13);
			}
		} else if (e == 31) { u = sign | 0x7F800000u | (m << 13); } else { u = sign | ((e - 15 + 127) << 23) | (m << 13); }
		float f;
		std::memcpy(&f, &u, 4);
		return f;
	}
};
using network_precision_t = half;  // common.h:66-70 with TCNN_HALF_PRECISION

template <typename T> constexpr float default_loss_scale() { return 1.0f; }           // common.h:243
template <> constexpr float default_loss_scale<half>() { return 128.0f; }

// cpp_api.h:62-85 / common_host.h free functions
inline int hip_device() { return tcnn_hip_device(); }
inline void set_hip_device(int device) { check(tcnn_set_hip_device(device)); }
inline int cuda_device() { return hip_device(); }                 // the reference's spelling
inline void set_cuda_device(int device) { set_hip_device(device); }
inline void free_all_gpu_memory_arenas() { tcnn_free_temporary_memory(); }  // gpu_memory.h:688-700
inline bool supports_jit_fusion(int device = -1) { return tcnn_supports_jit_fusion(device) != 0; }  // always false: no RTC path
inline uint32_t cuda_compute_capability() { return 950; }  // gfx950; callers compare it with a minimum architecture

#if defined(__HIPCC__)
// common_host.h:180-200: launches kernel(n_elements, args...) over n_elements threads, 128 per workgroup
template <typename K, typename T, typename... Types>
inline void linear_kernel(K kernel, uint32_t shmem_size, hipStream_t stream, T n_elements, Types... args) {
	if (n_elements <= 0) return;
	hipLaunchKernelGGL(kernel, dim3((uint32_t)div_round_up((uint64_t)n_elements, (uint64_t)N_THREADS_LINEAR)), dim3(N_THREADS_LINEAR), shmem_size, stream, n_elements, args...);
}
#endif

}  // namespace tcnn
