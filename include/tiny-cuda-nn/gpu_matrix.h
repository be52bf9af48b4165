/*
 * tiny-cuda-nn/gpu_matrix.h -- GPUMatrixDynamic<T> / GPUMatrix<T, layout>: the matrix value types of the public
 * signatures (reference gpu_matrix.h:106-490).  m rows (features) x n columns (samples -- the batch); column-major
 * (CM == AoS: one sample's values contiguous) is the default, row-major (RM == SoA) and a stride larger than the
 * leading dimension are honoured by the library (tcnn_matrix_t in tcnn_hip.h).
 */
#pragma once
#include <tiny-cuda-nn/common.h>

namespace tcnn {

template <typename T>
class GPUMatrixDynamic {
public:
	// owning, dense (gpu_matrix.h:128-139)
	GPUMatrixDynamic(uint32_t m, uint32_t n, MatrixLayout layout = CM) : m_rows(m), m_cols(n), m_layout(layout), m_owned(true) {
		m_stride = layout == CM ? m : n;
		if (n_elements() > 0) HIP_CHECK_THROW(hipMalloc(reinterpret_cast<void**>(&m_data), n_bytes()));
	}
	// owning, out of the stream-ordered arena (gpu_matrix.h:141-152: allocate_workspace(stream, ...)): a block of the library's
	// stream-keyed cache, handed back to it on destruction -- no driver allocation in a loop that makes temporaries per iteration
	GPUMatrixDynamic(uint32_t m, uint32_t n, hipStream_t stream, MatrixLayout layout = CM) : m_rows(m), m_cols(n), m_layout(layout), m_owned(true) {
		m_stride = layout == CM ? m : n;
		if (n_elements() > 0) {
			void* p = nullptr;
			check(tcnn_stream_malloc((tcnn_stream_t)stream, n_bytes(), &p, &m_arena_bytes));
			m_data = (T*)p;
			m_arena_stream = stream;
			m_from_arena = true;
		}
	}
	// non-owning view of caller memory (gpu_matrix.h:118-126); stride 0 = dense
	GPUMatrixDynamic(T* data, uint32_t m, uint32_t n, MatrixLayout layout = CM, uint32_t stride = 0)
	    : m_data(data), m_rows(m), m_cols(n), m_stride(stride ? stride : (layout == CM ? m : n)), m_layout(layout), m_owned(false) {}
	GPUMatrixDynamic() = default;
	GPUMatrixDynamic(const GPUMatrixDynamic&) = delete;
	GPUMatrixDynamic& operator=(const GPUMatrixDynamic&) = delete;
	GPUMatrixDynamic(GPUMatrixDynamic&& o) noexcept { *this = std::move(o); }
	GPUMatrixDynamic& operator=(GPUMatrixDynamic&& o) noexcept {
		std::swap(m_data, o.m_data);
		std::swap(m_rows, o.m_rows);
		std::swap(m_cols, o.m_cols);
		std::swap(m_stride, o.m_stride);
		std::swap(m_layout, o.m_layout);
		std::swap(m_owned, o.m_owned);
		std::swap(m_from_arena, o.m_from_arena);
		std::swap(m_arena_stream, o.m_arena_stream);
		std::swap(m_arena_bytes, o.m_arena_bytes);
		return *this;
	}
	virtual ~GPUMatrixDynamic() {
		if (m_owned && m_data) {
			if (m_from_arena) (void)tcnn_stream_free((tcnn_stream_t)m_arena_stream, m_data, m_arena_bytes);
			else (void)hipFree(m_data);
		}
	}

	T* data() const { return m_data; }
	uint32_t rows() const { return m_rows; }
	uint32_t cols() const { return m_cols; }
	uint32_t m() const { return m_rows; }
	uint32_t n() const { return m_cols; }
	uint32_t fan_out() const { return m_rows; }
	uint32_t fan_in() const { return m_cols; }
	uint32_t stride() const { return m_stride; }
	MatrixLayout layout() const { return m_layout; }
	MatrixLayout transposed_layout() const { return m_layout == RM ? CM : RM; }
	bool is_contiguous() const { return m_stride == (m_layout == CM ? m_rows : m_cols); }
	uint32_t n_elements() const { return m_rows * m_cols; }
	size_t n_bytes() const { return (size_t)n_elements() * sizeof(T); }

	// views (gpu_matrix.h:176-212): share the memory, never own it
	GPUMatrixDynamic<T> slice(uint32_t offset_rows, uint32_t new_rows, uint32_t offset_cols, uint32_t new_cols) const {
		T* p = m_data + (m_layout == CM ? (size_t)offset_cols * m_stride + offset_rows : (size_t)offset_rows * m_stride + offset_cols);
		return GPUMatrixDynamic<T>(p, new_rows, new_cols, m_layout, m_stride);
	}
	GPUMatrixDynamic<T> slice_rows(uint32_t offset, uint32_t size) const { return slice(offset, size, 0, m_cols); }
	GPUMatrixDynamic<T> slice_cols(uint32_t offset, uint32_t size) const { return slice(0, m_rows, offset, size); }
	GPUMatrixDynamic<T> transposed() const { return GPUMatrixDynamic<T>(m_data, m_cols, m_rows, transposed_layout(), m_stride); }

	void memset(int value) {
		CHECK_THROW(is_contiguous());
		HIP_CHECK_THROW(hipMemset(m_data, value, n_bytes()));
	}
	void memset_async(hipStream_t stream, int value) {
		CHECK_THROW(is_contiguous());
		HIP_CHECK_THROW(hipMemsetAsync(m_data, value, n_bytes(), stream));
	}
	void copy_from_host(const T* host) {
		CHECK_THROW(is_contiguous());
		HIP_CHECK_THROW(hipMemcpy(m_data, host, n_bytes(), hipMemcpyHostToDevice));
	}
	void copy_from_host(const std::vector<T>& host) {
		if (host.size() < n_elements()) throw std::runtime_error("GPUMatrix::copy_from_host: host buffer too small");
		copy_from_host(host.data());
	}
	std::vector<T> to_cpu_vector() const {
		CHECK_THROW(is_contiguous());
		std::vector<T> out(n_elements());
		HIP_CHECK_THROW(hipMemcpy(out.data(), m_data, n_bytes(), hipMemcpyDeviceToHost));
		return out;
	}

	// the matrix as it crosses the C ABI
	tcnn_matrix_t c_matrix() const {
		return tcnn_matrix_t{(void*)m_data, m_rows, m_cols, m_stride, m_layout == CM ? TCNN_LAYOUT_COLUMN_MAJOR : TCNN_LAYOUT_ROW_MAJOR};
	}

protected:
	T* m_data = nullptr;
	uint32_t m_rows = 0, m_cols = 0, m_stride = 0;
	MatrixLayout m_layout = CM;
	bool m_owned = false;
	bool m_from_arena = false;  // m_data is a block of the library's stream-keyed cache (tcnn_stream_malloc)
	hipStream_t m_arena_stream = nullptr;
	size_t m_arena_bytes = 0;
};

// static layout (gpu_matrix.h:253-330)
template <typename T, MatrixLayout _layout = MatrixLayout::ColumnMajor>
class GPUMatrix : public GPUMatrixDynamic<T> {
public:
	static constexpr MatrixLayout static_layout = _layout;
	static constexpr MatrixLayout static_transposed_layout = _layout == RM ? CM : RM;
	GPUMatrix(uint32_t m, uint32_t n) : GPUMatrixDynamic<T>(m, n, _layout) {}
	GPUMatrix(uint32_t m, uint32_t n, hipStream_t stream) : GPUMatrixDynamic<T>(m, n, stream, _layout) {}
	GPUMatrix(T* data, uint32_t m, uint32_t n, uint32_t stride = 0) : GPUMatrixDynamic<T>(data, m, n, _layout, stride) {}
	GPUMatrix() = default;
	GPUMatrix(GPUMatrix&& o) noexcept : GPUMatrixDynamic<T>(std::move(o)) {}
	GPUMatrix& operator=(GPUMatrix&& o) noexcept {
		GPUMatrixDynamic<T>::operator=(std::move(o));
		return *this;
	}
	explicit GPUMatrix(GPUMatrixDynamic<T>&& o) : GPUMatrixDynamic<T>(std::move(o)) {
		if (this->layout() != _layout) throw std::runtime_error("GPUMatrix must be constructed from a GPUMatrixDynamic with matching layout.");  // gpu_matrix.h:266-270
	}
	GPUMatrix<T, _layout> slice_cols(uint32_t offset, uint32_t size) const {
		return GPUMatrix<T, _layout>(this->m_data + (_layout == CM ? (size_t)offset * this->m_stride : offset), this->m_rows, size, this->m_stride);
	}
};

}  // namespace tcnn
