/*
 * tiny-cuda-nn/random.h -- default_rng_t + generate_random_uniform (reference random.h:39-75, pcg32.h:40-170).
 * The generator is a position in the pcg32 stream of its seed; the device kernel and the draw order are the library's
 * (tcnn_generate_random_uniform), so `default_rng_t rng{1337}` yields the reference's sequence.
 */
#pragma once
#include <tiny-cuda-nn/common.h>

namespace tcnn {

struct pcg32 {
	uint64_t seed, position = 0;
	explicit pcg32(uint64_t seed_) : seed(seed_) {}  // pcg32(initstate), initseq = 1 (pcg32.h:56-59)
	void advance(int64_t delta) { position += (uint64_t)delta; }
};
using default_rng_t = pcg32;

template <typename T, typename RNG>
inline void generate_random_uniform(hipStream_t stream, RNG& rng, size_t n_elements, T* out, T lower = (T)0.0, T upper = (T)1.0) {
	static_assert(sizeof(T) == sizeof(float), "the library draws fp32 values");
	check(tcnn_generate_random_uniform(stream, rng.seed, &rng.position, n_elements, reinterpret_cast<float*>(out), (float)lower, (float)upper));
}
template <typename T, typename RNG>
inline void generate_random_uniform(RNG& rng, size_t n_elements, T* out, T lower = (T)0.0, T upper = (T)1.0) {
	generate_random_uniform<T>(nullptr, rng, n_elements, out, lower, upper);
}

}  // namespace tcnn
