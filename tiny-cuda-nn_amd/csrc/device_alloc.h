// device_alloc.h -- every device allocation of the library goes through device_malloc / device_free.
//
// Normally these are hipMalloc / hipFree.  TCNN_DEBUG_ALLOC selects a checking allocator (the reference wraps its
// allocations the same way, gpu_memory.h:97-130, and counts them; here the wrapper is a debugging tool for the kernels):
//   TCNN_DEBUG_ALLOC=canary   every block gets a 4 KiB head and tail filled with 0xA5 and a body filled with 0xFF
//                             (NaN as fp16 / fp32, 2^32 - 1 as a count); debug_alloc_check() and device_free() verify the
//                             canaries -- an out-of-bounds WRITE becomes a reported error instead of silent corruption;
//   TCNN_DEBUG_ALLOC=fence    every block is mapped on its own (hipMemAddressReserve / hipMemCreate / hipMemMap) with its
//                             END on the last mapped byte (64-byte granularity) and an unmapped granule on either side:
//                             an out-of-bounds READ or WRITE past the end faults at once, whatever else the process has
//                             mapped.  The <= 63 slack bytes carry the canary.
// In both modes the scratch cache recycles exact-size blocks only and re-poisons a block every time it is handed out, so
// that nothing can depend on what an earlier use left behind.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn_hip {

enum class DebugAlloc : int { Off = 0, Canary = 1, Fence = 2 };

inline DebugAlloc debug_alloc_mode() {
	static const DebugAlloc mode = [] {
		const char* e = getenv("TCNN_DEBUG_ALLOC");
		if (!e || !*e || !strcmp(e, "0") || !strcmp(e, "off")) return DebugAlloc::Off;
		if (!strcmp(e, "fence") || !strcmp(e, "2")) return DebugAlloc::Fence;
		return DebugAlloc::Canary;
	}();
	return mode;
}

// fence mode: alignment of the blocks (TCNN_DEBUG_ALLOC_ALIGN, a power of two >= 16; default 64).  hipMalloc returns 4 KiB-aligned
// blocks, callers may pass any 16-byte aligned pointer: a kernel that silently assumes more shows up at a small value here.
inline size_t debug_alloc_align() {
	static const size_t align = [] {
		const char* e = getenv("TCNN_DEBUG_ALLOC_ALIGN");
		size_t a = e ? (size_t)atol(e) : 64;
		if (a < 16 || (a & (a - 1))) a = 64;
		return a;
	}();
	return align;
}
constexpr unsigned char DEBUG_CANARY_BYTE = 0xA5, DEBUG_POISON_BYTE = 0xFF;
constexpr size_t DEBUG_CANARY_BYTES = 4096;

struct DebugBlock {
	void* base = nullptr;      // hipMalloc'ed block (canary) or reserved address range (fence)
	size_t reserved = 0;       // fence: bytes of the address range
	size_t mapped = 0;         // fence: bytes mapped (starting one granule into the range)
	size_t granule = 0;
	size_t bytes = 0;          // what the caller asked for
	hipMemGenericAllocationHandle_t handle = {};
	int device = 0;
};

class DebugAllocator {
public:
	static DebugAllocator& get() {
		static DebugAllocator a;
		return a;
	}
	void* allocate(size_t bytes) {
		if (bytes == 0) bytes = 1;
		DebugBlock b;
		b.bytes = bytes;
		(void)hipGetDevice(&b.device);
		void* user = nullptr;
		if (debug_alloc_mode() == DebugAlloc::Fence) {
			hipMemAllocationProp prop = {};
			prop.type = hipMemAllocationTypePinned;
			prop.location.type = hipMemLocationTypeDevice;
			prop.location.id = b.device;
			size_t gran = 0;
			check(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum), "hipMemGetAllocationGranularity");
			if (gran == 0) gran = 2u << 20;
			b.granule = gran;
			b.mapped = (bytes + gran - 1) / gran * gran;
			b.reserved = b.mapped + 2 * gran;
			check(hipMemAddressReserve(&b.base, b.reserved, gran, nullptr, 0), "hipMemAddressReserve");
			check(hipMemCreate(&b.handle, b.mapped, &prop, 0), "hipMemCreate");
			char* first = (char*)b.base + gran;
			check(hipMemMap(first, b.mapped, 0, b.handle, 0), "hipMemMap");
			hipMemAccessDesc access = {};
			access.location = prop.location;
			access.flags = hipMemAccessFlagsProtReadWrite;
			check(hipMemSetAccess(first, b.mapped, &access, 1), "hipMemSetAccess");
			const size_t span = (bytes + debug_alloc_align() - 1) / debug_alloc_align() * debug_alloc_align();  // the block ends on the last mapped byte, up to the alignment
			user = first + b.mapped - span;
			check(hipMemset(user, DEBUG_POISON_BYTE, bytes), "hipMemset");
			if (span > bytes) check(hipMemset((char*)user + bytes, DEBUG_CANARY_BYTE, span - bytes), "hipMemset");
		} else {
			check(hipMalloc(&b.base, bytes + 2 * DEBUG_CANARY_BYTES), "hipMalloc");
			user = (char*)b.base + DEBUG_CANARY_BYTES;
			check(hipMemset(b.base, DEBUG_CANARY_BYTE, DEBUG_CANARY_BYTES), "hipMemset");
			check(hipMemset(user, DEBUG_POISON_BYTE, bytes), "hipMemset");
			check(hipMemset((char*)user + bytes, DEBUG_CANARY_BYTE, DEBUG_CANARY_BYTES), "hipMemset");
		}
		check(hipDeviceSynchronize(), "hipDeviceSynchronize");
		std::lock_guard<std::mutex> lock(m_mutex);
		m_blocks[user] = b;
		return user;
	}
	// false: not one of ours
	bool free(void* user) {
		DebugBlock b;
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			auto it = m_blocks.find(user);
			if (it == m_blocks.end()) return false;
			b = it->second;
			m_blocks.erase(it);
		}
		(void)hipDeviceSynchronize();
		std::string report;
		if (!verify(user, b, &report)) {
			fprintf(stderr, "tiny-cuda-nn_amd debug allocator: %s\n", report.c_str());
			std::lock_guard<std::mutex> lock(m_mutex);
			m_freed_corrupt++;
			m_freed_report += report + "\n";
		}
		if (b.reserved) {
			(void)hipMemUnmap((char*)b.base + b.granule, b.mapped);
			(void)hipMemRelease(b.handle);
			// the address range is returned only on request (TCNN_DEBUG_ALLOC_REUSE_VA=1): a freed block's addresses then stay
			// unmapped for the rest of the process, so a use after free faults as well
			static const bool reuse_va = getenv("TCNN_DEBUG_ALLOC_REUSE_VA") && atoi(getenv("TCNN_DEBUG_ALLOC_REUSE_VA")) != 0;
			if (reuse_va) (void)hipMemAddressFree(b.base, b.reserved);
		} else {
			(void)hipFree(b.base);
		}
		return true;
	}
	// number of blocks (live, or freed since the last call) whose canaries were overwritten; details in *report
	size_t check_all(std::string* report) {
		(void)hipDeviceSynchronize();
		std::map<void*, DebugBlock> blocks;
		size_t bad = 0;
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			blocks = m_blocks;
			bad = m_freed_corrupt;
			if (report) *report += m_freed_report;
			m_freed_corrupt = 0;
			m_freed_report.clear();
		}
		for (auto& kv : blocks) {
			std::string r;
			if (!verify(kv.first, kv.second, &r)) {
				++bad;
				if (report) *report += r + "\n";
			}
		}
		return bad;
	}
	size_t n_live() {
		std::lock_guard<std::mutex> lock(m_mutex);
		return m_blocks.size();
	}

private:
	static void check(hipError_t e, const char* what) {
		if (e != hipSuccess) throw std::runtime_error(std::string("debug allocator: ") + what + " failed: " + hipGetErrorString(e));
	}
	static bool region_is(const void* dev, size_t n, unsigned char value, size_t* first_bad) {
		if (n == 0) return true;
		std::vector<unsigned char> host(n);
		if (hipMemcpy(host.data(), dev, n, hipMemcpyDeviceToHost) != hipSuccess) return false;
		for (size_t i = 0; i < n; ++i) {
			if (host[i] != value) {
				*first_bad = i;
				return false;
			}
		}
		return true;
	}
	static bool verify(void* user, const DebugBlock& b, std::string* report) {
		int before = 0;
		(void)hipGetDevice(&before);
		if (before != b.device) (void)hipSetDevice(b.device);
		bool ok = true;
		size_t at = 0;
		char msg[256];
		if (b.reserved) {
			const size_t span = (b.bytes + debug_alloc_align() - 1) / debug_alloc_align() * debug_alloc_align();
			if (!region_is((char*)user + b.bytes, span - b.bytes, DEBUG_CANARY_BYTE, &at)) {
				snprintf(msg, sizeof(msg), "block %p (%zu bytes): byte %zu past its end was overwritten", user, b.bytes, at);
				ok = false;
			}
		} else {
			if (!region_is(b.base, DEBUG_CANARY_BYTES, DEBUG_CANARY_BYTE, &at)) {
				snprintf(msg, sizeof(msg), "block %p (%zu bytes): byte %zu before its start was overwritten", user, b.bytes, DEBUG_CANARY_BYTES - at);
				ok = false;
			} else if (!region_is((char*)user + b.bytes, DEBUG_CANARY_BYTES, DEBUG_CANARY_BYTE, &at)) {
				snprintf(msg, sizeof(msg), "block %p (%zu bytes): byte %zu past its end was overwritten", user, b.bytes, at);
				ok = false;
			}
		}
		if (!ok && report) *report = msg;
		if (before != b.device) (void)hipSetDevice(before);
		return ok;
	}
	std::mutex m_mutex;
	std::map<void*, DebugBlock> m_blocks;
	size_t m_freed_corrupt = 0;
	std::string m_freed_report;
};

inline void* device_malloc(size_t bytes) {
	if (debug_alloc_mode() != DebugAlloc::Off) return DebugAllocator::get().allocate(bytes);
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) throw std::runtime_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
	return p;
}
inline void device_free(void* p) {
	if (!p) return;
	if (debug_alloc_mode() != DebugAlloc::Off && DebugAllocator::get().free(p)) return;
	(void)hipFree(p);
}
template <typename T>
inline T* device_malloc_n(size_t n) {
	return (T*)device_malloc(n * sizeof(T));
}

}  // namespace tcnn_hip
