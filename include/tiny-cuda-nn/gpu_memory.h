/*
 * tiny-cuda-nn/gpu_memory.h -- GPUMemory<T>: an owning device array (reference gpu_memory.h:60-403) for callers of the
 * hot path (the sample keeps its image, coordinates and renders in it, mlp_learning_an_image.cu:156-200).  The library's
 * own scratch memory is a stream-ordered block cache inside libtcnn_hip.so (where the reference has GPUMemoryArena); hosts
 * draw from the same cache with GPUMatrix(m, n, stream) / tcnn_stream_malloc, free_all_gpu_memory_arenas() (common.h) releases it.
 */
#pragma once
#include <tiny-cuda-nn/common.h>

namespace tcnn {

template <typename T>
class GPUMemory {
public:
	GPUMemory() = default;
	explicit GPUMemory(size_t size) { resize(size); }
	GPUMemory(const GPUMemory& other) : GPUMemory(other.m_size) { copy_from_device(other); }
	GPUMemory& operator=(const GPUMemory& other) {
		if (this != &other) {
			resize(other.m_size);
			copy_from_device(other);
		}
		return *this;
	}
	GPUMemory(GPUMemory&& other) noexcept { *this = std::move(other); }
	GPUMemory& operator=(GPUMemory&& other) noexcept {
		std::swap(m_data, other.m_data);
		std::swap(m_size, other.m_size);
		return *this;
	}
	~GPUMemory() { free_memory_noexcept(); }

	void free_memory() {
		if (m_data) HIP_CHECK_THROW(hipFree(m_data));
		m_data = nullptr;
		m_size = 0;
	}
	void resize(size_t size) {  // gpu_memory.h:173-191: contents are not preserved
		if (size == m_size) return;
		free_memory();
		if (size > 0) HIP_CHECK_THROW(hipMalloc(reinterpret_cast<void**>(&m_data), size * sizeof(T)));
		m_size = size;
	}
	void enlarge(size_t size) {  // gpu_memory.h:193-200
		if (size > m_size) resize(size);
	}
	void memset(int value, size_t num_elements, size_t offset = 0) {
		if (num_elements + offset > m_size) throw std::runtime_error("Could not set memory: Number of elements " + std::to_string(num_elements) + "+" + std::to_string(offset) + " larger than allocated memory " + std::to_string(m_size) + ".");
		HIP_CHECK_THROW(hipMemset(m_data + offset, value, num_elements * sizeof(T)));
	}
	void memset(int value) { memset(value, m_size); }
	void memset_async(hipStream_t stream, int value) { HIP_CHECK_THROW(hipMemsetAsync(m_data, value, m_size * sizeof(T), stream)); }

	void copy_from_host(const T* host_data, size_t num_elements) {
		if (num_elements > m_size) throw std::runtime_error("Trying to copy " + std::to_string(num_elements) + " elements, but memory size is only " + std::to_string(m_size) + ".");
		HIP_CHECK_THROW(hipMemcpy(m_data, host_data, num_elements * sizeof(T), hipMemcpyHostToDevice));
	}
	void copy_from_host(const T* host_data) { copy_from_host(host_data, m_size); }
	void copy_from_host(const std::vector<T>& data) {
		if (data.size() < m_size) throw std::runtime_error("Trying to copy " + std::to_string(m_size) + " elements, but vector size is only " + std::to_string(data.size()) + ".");
		copy_from_host(data.data(), m_size);
	}
	void resize_and_copy_from_host(const std::vector<T>& data) {
		resize(data.size());
		copy_from_host(data);
	}
	void copy_to_host(T* host_data, size_t num_elements) const {
		if (num_elements > m_size) throw std::runtime_error("Trying to copy " + std::to_string(num_elements) + " elements, but memory size is only " + std::to_string(m_size) + ".");
		HIP_CHECK_THROW(hipMemcpy(host_data, m_data, num_elements * sizeof(T), hipMemcpyDeviceToHost));
	}
	void copy_to_host(T* host_data) const { copy_to_host(host_data, m_size); }
	void copy_to_host(std::vector<T>& data) const {
		if (data.size() < m_size) data.resize(m_size);
		copy_to_host(data.data(), m_size);
	}
	void copy_from_device(const GPUMemory<T>& other, size_t size) {
		if (size == 0) return;
		if (m_size < size) resize(size);
		HIP_CHECK_THROW(hipMemcpy(m_data, other.m_data, size * sizeof(T), hipMemcpyDeviceToDevice));
	}
	void copy_from_device(const GPUMemory<T>& other) { copy_from_device(other, other.m_size); }

	T* data() const { return m_data; }
	size_t size() const { return m_size; }
	size_t get_num_elements() const { return m_size; }
	size_t get_bytes() const { return m_size * sizeof(T); }
	size_t n_bytes() const { return get_bytes(); }
	size_t bytes() const { return get_bytes(); }

private:
	void free_memory_noexcept() noexcept {
		if (m_data) (void)hipFree(m_data);
		m_data = nullptr;
		m_size = 0;
	}
	T* m_data = nullptr;
	size_t m_size = 0;
};

}  // namespace tcnn
