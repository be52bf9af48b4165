// api.hip -- host-side object layer (encoding / network / NetworkWithInputEncoding / Trainer) and
// the C ABI declared in include/tcnn_hip.h.  Mirrors the behaviour of the reference's
//   src/cpp_api.cu:72-174 (type-erased Module), include/tiny-cuda-nn/config.h:46-63,
//   trainer.h:51-503, network_with_input_encoding.h:55-130, grid.h:673-737/1725-1852,
//   src/network.cu:51-138
// for the HashGrid + FullyFusedMLP hot path only.  Everything heavy happens in the kernel files.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>
#include <functional>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../../include/tcnn_hip.h"
#include "adam_device.h"
#include "device_alloc.h"
#include "elementwise_kernels.h"
#include "direct_exchange.h"
#include "grid_kernels.h"
#include "../../include/tiny-cuda-nn/json_mini.h"
#include "mlp_kernels.h"
#include "snapshot_msgpack.h"

namespace tcnn_hip {

#define HIP_CHECK(x)                                                                                        \
	do {                                                                                                    \
		hipError_t e_ = (x);                                                                                \
		if (e_ != hipSuccess) throw std::runtime_error(std::string(#x " failed: ") + hipGetErrorString(e_)); \
	} while (0)

// ------------------------------------------------------------------------------------------------
// logging (common_host.h:46-69) and error state
// ------------------------------------------------------------------------------------------------
static void (*g_log_callback)(int, const char*) = nullptr;
static thread_local std::string g_last_error;

static void log_message(int severity, const std::string& msg) {
	if (g_log_callback) {
		g_log_callback(severity, msg.c_str());
	} else if (severity == TCNN_LOG_WARNING || severity == TCNN_LOG_ERROR) {
		fprintf(stderr, "tiny-cuda-nn_amd %s: %s\n", severity == TCNN_LOG_WARNING ? "warning" : "error", msg.c_str());
	}
}

// ------------------------------------------------------------------------------------------------
// (device, stream)-keyed scratch cache (stands where the reference's per-stream GPUMemoryArena stands,
// gpu_memory.h:405-700; its null-stream arenas are per device, global_gpu_memory_arenas()[cuda_device()]): blocks are
// recycled per stream of one device, so steady-state steps allocate nothing (a precondition for hipGraph capture), a
// recycled block is only ever reused in stream order, and a block never crosses to another GPU (the default stream's
// handle is 0 on every device).
// ------------------------------------------------------------------------------------------------
typedef std::pair<int, hipStream_t> StreamKey;
static StreamKey stream_key(hipStream_t stream) {
	int device = 0;
	(void)hipGetDevice(&device);
	return {device, stream};
}
// synchronises and frees `blocks` with their own device current
template <typename F>
static void for_each_device_of(const std::map<StreamKey, F>& m, const std::function<void(const StreamKey&, const F&)>& fn) {
	int before = 0;
	(void)hipGetDevice(&before);
	for (auto& kv : m) {
		(void)hipSetDevice(kv.first.first);
		(void)hipDeviceSynchronize();
		fn(kv.first, kv.second);
	}
	(void)hipSetDevice(before);
}

class ScratchCache {
public:
	static void* acquire(hipStream_t stream, size_t bytes, size_t* granted) {
		const bool debug = debug_alloc_mode() != DebugAlloc::Off;
		// checking allocator (device_alloc.h): exact sizes, so that a block ends where the request ends, and fresh poison on
		// every hand-out, so that nothing can rely on what an earlier use left in a recycled block
		bytes = debug ? (bytes ? bytes : (size_t)1) : next_multiple(bytes ? bytes : (size_t)1, (size_t)256);
		{
			std::lock_guard<std::mutex> lock(mutex());
			auto& fl = lists()[stream_key(stream)];
			auto it = fl.lower_bound(bytes);
			if (it != fl.end() && it->first <= (debug ? bytes : 2 * bytes)) {
				void* p = it->second;
				*granted = it->first;
				fl.erase(it);
				if (debug) HIP_CHECK(hipMemsetAsync(p, DEBUG_POISON_BYTE, bytes, stream));
				return p;
			}
		}
		void* p = device_malloc(bytes);
		*granted = bytes;
		return p;
	}
	static void release(const StreamKey& key, void* p, size_t bytes) {
		std::lock_guard<std::mutex> lock(mutex());
		lists()[key].emplace(bytes, p);
	}
	static void free_all() {
		std::lock_guard<std::mutex> lock(mutex());
		for_each_device_of<std::multimap<size_t, void*>>(lists(), [](const StreamKey&, const std::multimap<size_t, void*>& blocks) {
			for (auto& b : blocks) device_free(b.second);
		});
		lists().clear();
	}

private:
	static std::mutex& mutex() {
		static std::mutex m;
		return m;
	}
	static std::map<StreamKey, std::multimap<size_t, void*>>& lists() {
		static std::map<StreamKey, std::multimap<size_t, void*>> l;
		return l;
	}
};

// Small per-stream device buffers that are zero whenever no kernel of that stream is using them (the bucketed grid
// backward's queue counters: its kernels hand them back zeroed, so they are cleared exactly once, at allocation).
class ZeroedCounters {
public:
	static uint32_t* get(hipStream_t stream, size_t n) {
		std::lock_guard<std::mutex> lock(mutex());
		auto& slot = slots()[stream_key(stream)];
		if (slot.second < n) {
			if (slot.first) {
				HIP_CHECK(hipStreamSynchronize(stream));
				device_free(slot.first);
				slot = {nullptr, 0};
			}
			const size_t cap = std::max<size_t>(next_multiple<size_t>(n, 1024), 4096);
			void* p = device_malloc(cap * sizeof(uint32_t));
			HIP_CHECK(hipMemset(p, 0, cap * sizeof(uint32_t)));
			HIP_CHECK(hipDeviceSynchronize());
			slot = {(uint32_t*)p, cap};
		}
		return slot.first;
	}
	static void free_all() {
		std::lock_guard<std::mutex> lock(mutex());
		for_each_device_of<std::pair<uint32_t*, size_t>>(slots(), [](const StreamKey&, const std::pair<uint32_t*, size_t>& slot) { device_free(slot.first); });
		slots().clear();
	}

private:
	static std::mutex& mutex() {
		static std::mutex m;
		return m;
	}
	static std::map<StreamKey, std::pair<uint32_t*, size_t>>& slots() {
		static std::map<StreamKey, std::pair<uint32_t*, size_t>> s;
		return s;
	}
};

struct Scratch {
	void* ptr = nullptr;
	size_t bytes = 0;
	StreamKey stream = {0, nullptr};  // (device, stream) the block belongs to
	Scratch() = default;
	Scratch(hipStream_t s, size_t n_bytes) : stream(stream_key(s)) { ptr = ScratchCache::acquire(s, n_bytes, &bytes); }
	Scratch(const Scratch&) = delete;
	Scratch& operator=(const Scratch&) = delete;
	Scratch(Scratch&& o) noexcept { *this = std::move(o); }
	Scratch& operator=(Scratch&& o) noexcept {
		reset();
		ptr = o.ptr;
		bytes = o.bytes;
		stream = o.stream;
		o.ptr = nullptr;
		return *this;
	}
	~Scratch() { reset(); }
	void reset() {
		if (ptr) ScratchCache::release(stream, ptr, bytes);
		ptr = nullptr;
	}
	template <typename T>
	T* as() const { return (T*)ptr; }
};

// ------------------------------------------------------------------------------------------------
// optional per-stage timing with HIP events recorded on the stream the kernels are launched on
// (bench.py's roofline leg; off by default -- no events are recorded unless a trainer enables it)
// ------------------------------------------------------------------------------------------------
enum Stage : int {
	STAGE_GRID_FWD = 0,
	STAGE_MLP_FWD,
	STAGE_LOSS,
	STAGE_MLP_BWD,       // weight transpose + fused backward + finalize
	STAGE_MLP_TRAIN,     // training_step fast path: weight transpose + forward/loss/backward in one kernel + finalize
	STAGE_GRID_BWD_SCATTER,     // bucketed backward pass A: derive the corner records once, bin them by owning slice
	STAGE_GRID_BWD,             // pass B (owners accumulate + store, overflow records included) -- or the whole backward in the sliced / atomic modes
	STAGE_ADAM,
	// the direct exchange's phases (direct_exchange.h); recorded whenever a profiler is on, whatever `only_stage` says: they exist on N > 1 only,
	// where a step is long and the first node run has to explain itself
	STAGE_DX_WAIT_GRADS,   // signal "my gradients are final" + wait for every peer's
	STAGE_DX_REDUCE,       // read the peers' shards over the links, fp32 sum, one rounding
	STAGE_DX_PUSH,         // write the stepped shard into every peer's parameter buffer
	STAGE_DX_WAIT_PARAMS,  // signal "pushed" + wait for every peer's push
	N_STAGES
};
static const char* const STAGE_NAMES[N_STAGES] = {"grid_forward", "mlp_forward", "loss", "mlp_backward", "mlp_train_fused", "grid_backward_scatter", "grid_backward", "adam",
                                                  "exchange_wait_gradients", "exchange_reduce", "exchange_push", "exchange_wait_parameters"};

struct Profiler {
	int only_stage = -1;  // -1: all stages
	std::vector<hipEvent_t> pool;
	size_t next = 0;
	struct Span {
		int stage;
		hipEvent_t a, b;
		bool counts;
	};
	std::vector<Span> spans;
	double total_ms[N_STAGES] = {};
	uint64_t count[N_STAGES] = {};

	hipEvent_t get() {
		if (next == pool.size()) {
			hipEvent_t e;
			HIP_CHECK(hipEventCreate(&e));
			pool.push_back(e);
		}
		return pool[next++];
	}
	void collect() {
		for (auto& s : spans) {
			HIP_CHECK(hipEventSynchronize(s.b));
			float ms = 0.0f;
			HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
			total_ms[s.stage] += ms;
			if (s.counts) count[s.stage]++;
		}
		spans.clear();
		next = 0;
	}
	~Profiler() {
		for (auto e : pool) (void)hipEventDestroy(e);
	}
};
static thread_local Profiler* g_profiler = nullptr;

// grid backward formulation (GridBackwardMode); TCNN_GRID_BACKWARD=sliced_f32|sliced_f16|atomic|bucketed overrides the default
static int initial_grid_backward_mode() {
	const char* e = getenv("TCNN_GRID_BACKWARD");
	if (e && std::string(e) == "atomic") return (int)GridBackwardMode::Atomic;
	if (e && std::string(e) == "sliced_f32") return (int)GridBackwardMode::SlicedF32;
	if (e && std::string(e) == "sliced_f16") return (int)GridBackwardMode::SlicedF16;
	return (int)GridBackwardMode::Bucketed;  // default: derive each corner once, bin by owner, exact fixed-point accumulation
}
static std::atomic<int> g_grid_backward_mode{initial_grid_backward_mode()};
static const uint32_t g_default_lds_slice_bytes = 0u;  // LDS bytes per table slice of the grid backward: 0 = the kernels' default (tcnn_trainer_set_lds_level_budget overrides)

static thread_local bool g_prof_every_stage = false;  // inside the direct exchange: its Adam is one of the exchange's phases
struct ProfScope {
	hipStream_t stream;
	int stage;
	bool counts;  // false: a further piece of a stage that is launched in several parts per step (time adds up, the launch count does not)
	hipEvent_t a = nullptr;
	ProfScope(hipStream_t s, int st, bool counts_ = true) : stream(s), stage(st), counts(counts_) {
		if (g_profiler && (g_profiler->only_stage < 0 || g_profiler->only_stage == st || st >= STAGE_DX_WAIT_GRADS || g_prof_every_stage)) {
			a = g_profiler->get();
			HIP_CHECK(hipEventRecord(a, stream));
		}
	}
	~ProfScope() {
		if (a) {
			hipEvent_t b = g_profiler->get();
			(void)hipEventRecord(b, stream);
			g_profiler->spans.push_back({stage, a, b, counts});
		}
	}
};
// grid_backward reports its kernels one by one (GridBackwardWorkspace::phase_hook); user = the stream
// (false while the level groups after the first are launched: their time adds to the stage, the launch count of the step does not)
static thread_local bool g_phase_hook_counts = true;
static void grid_backward_phase_hook(void* user, int phase, int begin) {
	static thread_local hipEvent_t a = nullptr;
	const int stage = phase == 0 ? STAGE_GRID_BWD_SCATTER : STAGE_GRID_BWD;
	if (!g_profiler || (g_profiler->only_stage >= 0 && g_profiler->only_stage != stage)) return;
	if (begin) {
		a = g_profiler->get();
		HIP_CHECK(hipEventRecord(a, (hipStream_t)user));
	} else if (a) {
		hipEvent_t b = g_profiler->get();
		HIP_CHECK(hipEventRecord(b, (hipStream_t)user));
		g_profiler->spans.push_back({stage, a, b, g_phase_hook_counts});
		a = nullptr;
	}
}

struct ProfilerGuard {
	explicit ProfilerGuard(Profiler* p) { g_profiler = p; }
	~ProfilerGuard() { g_profiler = nullptr; }
};

// ------------------------------------------------------------------------------------------------
// model description
// ------------------------------------------------------------------------------------------------
static uint32_t powi(uint32_t base, uint32_t exponent) {
	uint32_t r = 1;
	for (uint32_t i = 0; i < exponent; ++i) r *= base;
	return r;
}

static const char* to_string(GridType t) { return t == GridType::Hash ? "Hash" : t == GridType::Dense ? "Dense" : "Tiled"; }
static const char* to_string(InterpolationType t) {
	return t == InterpolationType::Nearest ? "Nearest" : t == InterpolationType::Linear ? "Linear" : "Smoothstep";
}
static const char* const ACTIVATION_NAMES[] = {"None", "ReLU", "LeakyReLU", "Exponential", "Sigmoid", "Squareplus", "Softplus", "Tanh"};
static const char* to_string(Activation a) { return ACTIVATION_NAMES[(int)a]; }

static GridType string_to_grid_type(const std::string& s) {  // common_host.cu:112-122
	if (equals_case_insensitive(s, "Hash")) return GridType::Hash;
	if (equals_case_insensitive(s, "Dense")) return GridType::Dense;
	if (equals_case_insensitive(s, "Tiled") || equals_case_insensitive(s, "Tile")) return GridType::Tiled;
	throw std::runtime_error("Invalid grid type: " + s);
}
static InterpolationType string_to_interpolation_type(const std::string& s) {  // common_host.cu:160-170
	if (equals_case_insensitive(s, "Nearest")) return InterpolationType::Nearest;
	if (equals_case_insensitive(s, "Linear")) return InterpolationType::Linear;
	if (equals_case_insensitive(s, "Smoothstep")) return InterpolationType::Smoothstep;
	throw std::runtime_error("Invalid interpolation type: " + s);
}
static Activation string_to_activation(const std::string& s) {  // common_host.cu:70-96
	for (int i = 0; i < 8; ++i) {
		if (equals_case_insensitive(s, ACTIVATION_NAMES[i])) return (Activation)i;
	}
	// SiLU and Sine need stored pre-activations, which FullyFusedMLP does not keep (common_device.h:377-386)
	if (equals_case_insensitive(s, "SiLU") || equals_case_insensitive(s, "Sine")) {
		throw std::runtime_error("Activation '" + s + "' is not supported by FullyFusedMLP (it needs stored pre-activations).");
	}
	throw std::runtime_error("Invalid activation name: " + s);  // common_host.cu:94
}

struct EncodingDesc {
	bool is_grid = false;
	// grid (grid.h:673-737)
	GridMeta grid = {};
	uint32_t log2_hashmap_size = 19, base_resolution = 16;
	float per_level_scale = 2.0f;
	// identity (identity.h:88-93)
	float id_scale = 1.0f, id_offset = 0.0f;
	// one-blob (oneblob.h:168-178): n_bins outputs per input dimension
	bool is_oneblob = false;
	uint32_t n_bins = 0;
	// frequency (frequency.h:106-111): sin and cos of n_frequencies octaves per input dimension
	bool is_frequency = false;
	uint32_t n_frequencies = 0;
	uint32_t n_dims = 0;
	uint32_t n_output_dims = 0;  // before padding
	uint32_t n_params = 0;
	uint32_t padded_output_width = 0;

	uint32_t required_output_alignment() const { return is_grid ? grid.n_feat : 1u; }  // grid.h:1066-1068
	void set_alignment(uint32_t alignment) {  // encoding.h:70-72
		uint32_t a = alignment, b = required_output_alignment();
		uint32_t x = a, y = b;
		while (y) {
			uint32_t t = x % y;
			x = y;
			y = t;
		}
		const uint32_t l = a / x * b;
		padded_output_width = next_multiple(n_output_dims, l);
	}

	Json hyperparams() const {
		Json j = Json::object();
		if (is_grid) {  // grid.h:1115-1132
			j["otype"] = "Grid";
			j["type"] = to_string((GridType)grid.grid_type);
			j["n_levels"] = grid.n_levels;
			j["n_features_per_level"] = grid.n_feat;
			j["base_resolution"] = base_resolution;
			j["per_level_scale"] = per_level_scale;
			j["interpolation"] = to_string((InterpolationType)grid.interp);
			j["hash"] = "CoherentPrime";
			if ((GridType)grid.grid_type == GridType::Hash) j["log2_hashmap_size"] = log2_hashmap_size;
		} else if (is_frequency) {  // frequency.h:200-205
			j["otype"] = "Frequency";
			j["n_frequencies"] = n_frequencies;
		} else if (is_oneblob) {  // oneblob.h:296-301
			j["otype"] = "OneBlob";
			j["n_bins"] = n_bins;
		} else {
			j["otype"] = "Identity";
			j["scale"] = id_scale;
			j["offset"] = id_offset;
		}
		return j;
	}
};

static EncodingDesc create_grid_encoding(uint32_t n_dims, const Json& enc) {  // grid.h:1725-1852
	EncodingDesc e;
	e.is_grid = true;
	e.n_dims = n_dims;
	const std::string hash = enc.value("hash", "CoherentPrime");
	if (!equals_case_insensitive(hash, "CoherentPrime")) throw std::runtime_error("GridEncoding: compiled without " + hash + " hash support.");
	const uint32_t F = enc.value("n_features_per_level", 2u);
	if (F != 1 && F != 2 && F != 4 && F != 8) throw std::runtime_error("GridEncoding: n_features_per_level must be 1, 2, 4, or 8.");
	const uint32_t log2_hashmap_size = enc.value("log2_hashmap_size", 19u);
	const std::string otype = enc.value("otype", "Grid");
	const std::string default_type = equals_case_insensitive(otype, "TiledGrid") ? "Tiled" : (equals_case_insensitive(otype, "DenseGrid") ? "Dense" : "Hash");
	uint32_t n_features;
	if (enc.contains("n_features") || enc.contains("n_grid_features")) {
		n_features = (uint32_t)(enc.contains("n_features") ? enc["n_features"] : enc["n_grid_features"]).as_number();
		if (enc.contains("n_levels")) throw std::runtime_error("GridEncoding: may not specify n_features and n_levels simultaneously (one determines the other)");
	} else {
		n_features = F * enc.value("n_levels", 16u);
	}
	const uint32_t n_levels = n_features / F;
	const GridType grid_type = string_to_grid_type(enc.value("type", default_type));
	const uint32_t base_resolution = enc.value("base_resolution", 16u);
	const float default_scale = grid_type == GridType::Dense ? std::exp(std::log(256.0f / (float)base_resolution) / (float)(n_levels - 1)) : 2.0f;
	const float per_level_scale = enc.value("per_level_scale", default_scale);

	const InterpolationType interp = string_to_interpolation_type(enc.value("interpolation", "Linear"));
	if (n_dims < 2 || n_dims > 4) throw std::runtime_error("GridEncoding: number of input dims must be 2, 3 or 4.");
	if (n_levels > MAX_N_LEVELS) throw std::runtime_error("GridEncoding: m_n_levels=" + std::to_string(n_levels) + " must be at most MAX_N_LEVELS=" + std::to_string(MAX_N_LEVELS));
	if (n_features % F != 0) throw std::runtime_error("GridEncoding: n_features=" + std::to_string(n_features) + " must be a multiple of N_FEATURES_PER_LEVEL=" + std::to_string(F));

	GridMeta& g = e.grid;
	g.n_dims = n_dims;
	g.n_levels = n_levels;
	g.n_feat = F;
	g.grid_type = (uint32_t)grid_type;
	g.interp = (uint32_t)interp;
	g.max_level = 1.0f;
	g.stochastic = enc.value("stochastic_interpolation", false) ? 1u : 0u;  // grid.h:1752
	// grid.h:699-727; the scale / resolution table is computed here once (fp32, same expressions as
	// common_device.h:886-895) and handed to the kernels, see GridMeta.
	const float log2_per_level_scale = std::log2(per_level_scale);
	uint32_t offset = 0;
	for (uint32_t i = 0; i < n_levels; ++i) {
		const float scale = exp2f((float)i * log2_per_level_scale) * (float)base_resolution - 1.0f;
		const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
		g.scale[i] = scale;
		g.resolution[i] = resolution;
		const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
		uint32_t params_in_level = std::pow((float)resolution, (float)n_dims) > (float)max_params ? max_params : powi(resolution, n_dims);
		params_in_level = next_multiple(params_in_level, 8u);
		if (grid_type == GridType::Tiled) {
			params_in_level = std::min(params_in_level, powi(base_resolution, n_dims));
		} else if (grid_type == GridType::Hash) {
			params_in_level = std::min(params_in_level, 1u << log2_hashmap_size);
		}
		g.offset[i] = offset;
		offset += params_in_level;
		log_message(TCNN_LOG_DEBUG, "GridEncoding at level " + std::to_string(i) + ": resolution=" + std::to_string(resolution) +
		                                " params_in_level=" + std::to_string(params_in_level));
	}
	g.offset[n_levels] = offset;
	e.n_params = offset * F;
	e.n_output_dims = n_features;
	e.padded_output_width = n_features;
	e.log2_hashmap_size = log2_hashmap_size;
	e.base_resolution = base_resolution;
	e.per_level_scale = per_level_scale;
	return e;
}

static EncodingDesc create_encoding_desc(uint32_t n_dims, const Json& enc, uint32_t alignment) {  // encoding.cu:131-145
	const std::string name = enc.value("otype", "OneBlob");
	EncodingDesc e;
	if (equals_case_insensitive(name, "Grid") || equals_case_insensitive(name, "HashGrid") || equals_case_insensitive(name, "DenseGrid") ||
	    equals_case_insensitive(name, "TiledGrid")) {
		e = create_grid_encoding(n_dims, enc);
	} else if (equals_case_insensitive(name, "Identity")) {
		e.is_grid = false;
		e.n_dims = n_dims;
		e.id_scale = enc.value("scale", 1.0f);
		e.id_offset = enc.value("offset", 0.0f);
		e.n_output_dims = n_dims;
		e.padded_output_width = n_dims;
	} else if (equals_case_insensitive(name, "Frequency")) {  // encoding.cu:65-67
		e.is_frequency = true;
		e.n_dims = n_dims;
		e.n_frequencies = enc.value("n_frequencies", 12u);
		if (e.n_frequencies == 0 || e.n_frequencies > 32) throw std::runtime_error("FrequencyEncoding: n_frequencies must be in [1, 32]");
		e.n_output_dims = n_dims * e.n_frequencies * 2u;
		e.padded_output_width = e.n_output_dims;
	} else if (equals_case_insensitive(name, "OneBlob")) {  // encoding.cu:118-120
		e.is_oneblob = true;
		e.n_dims = n_dims;
		e.n_bins = enc.value("n_bins", 16u);
		if (e.n_bins == 0 || (e.n_bins & (e.n_bins - 1)) != 0) throw std::runtime_error("Number of bins must be a power of 2");  // oneblob.h:174-176
		e.n_output_dims = n_dims * e.n_bins;
		e.padded_output_width = e.n_output_dims;
	} else {
		throw std::runtime_error("Encoding '" + name + "' not found (this build provides Grid/HashGrid/DenseGrid/TiledGrid, Frequency, OneBlob and Identity)");
	}
	if (alignment > 0) e.set_alignment(alignment);
	return e;
}

struct NetworkDesc {
	MlpMeta mlp = {};
	uint32_t n_output_dims = 0;
	uint32_t n_hidden_layers = 0;
	std::string otype;
	Json hyperparams() const {  // fully_fused_mlp.h:139-147
		Json j = Json::object();
		j["otype"] = "FullyFusedMLP";
		j["activation"] = to_string((Activation)mlp.activation);
		j["output_activation"] = to_string((Activation)mlp.output_activation);
		j["n_neurons"] = mlp.width;
		j["n_hidden_layers"] = n_hidden_layers;
		return j;
	}
};

static NetworkDesc create_network_desc(uint32_t n_input_dims, uint32_t n_output_dims, const Json& net) {  // network.cu:51-138
	const std::string otype = net.value("otype", "MLP");
	const bool known = equals_case_insensitive(otype, "MegakernelMLP") || equals_case_insensitive(otype, "FullyFusedMLP") ||
	                   equals_case_insensitive(otype, "MLP") || equals_case_insensitive(otype, "CutlassMLP");
	if (!known) throw std::runtime_error("Invalid network type: " + otype);
	NetworkDesc d;
	d.otype = otype;
	const uint32_t n_neurons = net.value("n_neurons", 128u);
	if (n_neurons != 16 && n_neurons != 32 && n_neurons != 64 && n_neurons != 128) {
		throw std::runtime_error("FullyFusedMLP only supports 16, 32, 64, and 128 neurons, but got " + std::to_string(n_neurons) +
		                         ". (CutlassMLP's arbitrary widths are not part of this build.)");
	}
	d.n_hidden_layers = net.value("n_hidden_layers", 5u);
	if (d.n_hidden_layers == 0) throw std::runtime_error("FullyFusedMLP requires at least 1 hidden layer (3 layers in total).");
	const Activation act = string_to_activation(net.value("activation", "ReLU"));
	const Activation out_act = string_to_activation(net.value("output_activation", "None"));
	d.n_output_dims = n_output_dims;
	d.mlp.in_width = n_input_dims;
	d.mlp.width = n_neurons;
	d.mlp.padded_out = next_multiple(n_output_dims, 16u);  // fully_fused_mlp.cu:656
	d.mlp.n_hidden_matmuls = d.n_hidden_layers - 1;
	d.mlp.activation = (uint32_t)act;
	d.mlp.output_activation = (uint32_t)out_act;
	if (d.mlp.padded_out > MLP_MAX_OUT_WIDTH) {
		throw std::runtime_error("FullyFusedMLP: more than " + std::to_string(MLP_MAX_OUT_WIDTH) + " output dimensions are not supported by this build.");
	}
	if (n_input_dims % 16 != 0 || n_input_dims > MLP_MAX_IN_WIDTH) {
		throw std::runtime_error("FullyFusedMLP: input width " + std::to_string(n_input_dims) + " must be a multiple of 16 and at most " + std::to_string(MLP_MAX_IN_WIDTH));
	}
	return d;
}

// A NetworkWithInputEncoding (network_with_input_encoding.h:40-130) or a bare encoding.
struct Model {
	uint32_t n_input_dims = 0;
	EncodingDesc enc;
	bool has_network = false;
	NetworkDesc net;
	std::string hyper_json;

	size_t n_mlp_params() const { return has_network ? net.mlp.n_params() : 0; }
	size_t n_params() const { return n_mlp_params() + enc.n_params; }  // network first, then encoding (:115-122)
	uint32_t padded_output_width() const { return has_network ? net.mlp.padded_out : enc.padded_output_width; }
	uint32_t output_width() const { return has_network ? net.n_output_dims : enc.padded_output_width; }
	std::string name() const { return has_network ? "NetworkWithInputEncoding" : (enc.is_grid ? "GridEncoding" : (enc.is_oneblob ? "OneBlobEncoding" : (enc.is_frequency ? "FrequencyEncoding" : "IdentityEncoding"))); }

	void finish() {
		Json j = Json::object();
		if (has_network) {
			j["otype"] = "NetworkWithInputEncoding";
			j["encoding"] = enc.hyperparams();
			j["network"] = net.hyperparams();
			hyper_json = j.dump();
		} else {
			hyper_json = enc.hyperparams().dump();
		}
	}

	// network_with_input_encoding.h:124-130 + fully_fused_mlp.cu:868-893 + grid.h:1076-1079
	void initialize_params(hipStream_t stream, Pcg32& rng, float* params_full_precision, float scale) const {
		if (has_network) {
			std::vector<float> host(n_mlp_params());
			float* p = host.data();
			auto xavier = [&](uint32_t rows, uint32_t cols) {  // gpu_matrix.h:292-307
				const float s = scale * std::sqrt(6.0f / (float)(rows + cols));
				for (size_t i = 0; i < (size_t)rows * cols; ++i) {
					float t = rng.next_float() * 2.0f;
					t = t * s;
					p[i] = t - s;
				}
				p += (size_t)rows * cols;
			};
			xavier(net.mlp.width, net.mlp.in_width);
			for (uint32_t i = 0; i < net.mlp.n_hidden_matmuls; ++i) xavier(net.mlp.width, net.mlp.width);
			xavier(net.mlp.padded_out, net.mlp.width);
			HIP_CHECK(hipMemcpyAsync(params_full_precision, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
		}
		if (enc.n_params > 0) {
			generate_random_uniform(stream, rng, enc.n_params, params_full_precision + n_mlp_params(), -1e-4f * scale, 1e-4f * scale);
		}
	}
};

// Layout of the caller's fp32 input / dL_dinput matrices (GPUMatrixDynamic: row- or column-major with a stride,
// gpu_matrix.h:106-250).  The plain entry points pass dense column-major matrices (one sample's values contiguous); the
// *_matrices entry points set this for the duration of their call (thread-local: one call per thread at a time).
struct IoLayout {
	bool set = false;
	uint32_t in_stride_i = 0, in_stride_d = 0;  // input element (sample i, dim d) at [i * in_stride_i + d * in_stride_d]
	uint32_t dx_stride_i = 0, dx_stride_d = 0;  // the same for dL_dinput
};
static thread_local IoLayout g_io_layout;
static uint32_t in_stride_d() { return g_io_layout.set ? g_io_layout.in_stride_d : 1u; }
static uint32_t dx_stride_d() { return g_io_layout.set ? g_io_layout.dx_stride_d : 1u; }
struct IoLayoutGuard {
	explicit IoLayoutGuard(const IoLayout& l) { g_io_layout = l; }
	~IoLayoutGuard() { g_io_layout = IoLayout(); }
};

static uint32_t in_stride_i(const Model& md) { return g_io_layout.set ? g_io_layout.in_stride_i : md.n_input_dims; }
static uint32_t dx_stride_i(const Model& md) { return g_io_layout.set ? g_io_layout.dx_stride_i : md.n_input_dims; }

static void check_batch(uint32_t n, uint32_t widest = 128) {
	if (n % BATCH_SIZE_GRANULARITY != 0) {  // object.h:170, 217, 298
		throw std::runtime_error("Batch size " + std::to_string(n) + " must be a multiple of " + std::to_string(BATCH_SIZE_GRANULARITY) + ".");
	}
	// the element-wise kernels index (sample, feature) pairs with 32 bits
	if ((uint64_t)n * widest > 0xFFFFFFFFull) {
		throw std::runtime_error("Batch size " + std::to_string(n) + " x " + std::to_string(widest) + " features exceeds 2^32 elements; split the batch.");
	}
}
static uint32_t widest_matrix(const struct Model& md);

struct ForwardCtx {
	hipStream_t stream = nullptr;
	uint32_t n = 0;
	Scratch enc;     // half, feature-major [enc.padded][n]   (network_with_input_encoding.h:76)
	Scratch hidden;  // half [n_hidden][n][width]             (fully_fused_mlp.cu:841-854)
	Scratch dy_dx;   // fp32 [(k*n + i)*D + d]                (grid.h:783-785)
};

// Encoding forward into a feature-major (SoA) or sample-major (AoS) half matrix.
static void encoding_forward(hipStream_t stream, const Model& md, uint32_t n, const float* input, const half_t* enc_params, half_t* out,
                             bool soa, float* dy_dx) {
	const EncodingDesc& e = md.enc;
	const uint32_t stride_k = soa ? n : 1u, stride_i = soa ? 1u : e.padded_output_width;
	ProfScope prof(stream, STAGE_GRID_FWD);
	if (e.is_grid) {
		GridIO io = {input, in_stride_i(md), in_stride_d(), n, stride_k, stride_i};
		grid_forward(stream, e.grid, io, enc_params, out, dy_dx);
		const uint32_t n_to_pad = e.padded_output_width - e.n_output_dims;
		if (n_to_pad > 0) {  // grid.h:757-766: padded dims are zero
			if (soa) {
				HIP_CHECK(hipMemsetAsync(out + (size_t)e.n_output_dims * n, 0, (size_t)n_to_pad * n * sizeof(half_t), stream));
			} else {
				HIP_CHECK(hipMemset2DAsync(out + e.n_output_dims, (size_t)e.padded_output_width * sizeof(half_t), 0, (size_t)n_to_pad * sizeof(half_t), n, stream));
			}
		}
	} else if (e.is_frequency) {
		frequency_forward(stream, n, e.n_dims, e.n_frequencies, e.padded_output_width, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	} else if (e.is_oneblob) {
		oneblob_forward(stream, n, e.n_dims, e.n_bins, e.padded_output_width, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	} else {
		identity_forward(stream, n, e.n_dims, e.padded_output_width, e.id_scale, e.id_offset, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	}
}

// Single-kernel network passes (process-wide; tcnn_set_fused_network_passes(0) turns them off):
//   * training_step: encoding forward, ONE kernel for the network's forward + loss + backward, encoding backward;
//   * forward() + backward() (Trainer and modules): the forward pass saves only the encoded input; the backward pass runs the same
//     kernel with the caller's dL/doutput in place of the loss -- it RECOMPUTES the hidden activations (three small matrix products)
//     instead of reading back what the forward pass would have had to write (2 B x width x layers per sample each way).
// Off: k_mlp_forward (saves the activations) -> k_loss -> k_mlp_backward.  Same results either way (tests/test_emu_kernels.py).
// training_step: an unpadded Identity encoding is evaluated by the network kernel's own input loads (MlpF32Input) where an instance offers it;
// tcnn_set_fused_identity_input(0): always the separate encoding kernel (A/B runs, tests)
static std::atomic<int> g_fused_identity_input{1};
static std::atomic<int> g_fused_network_passes{1};
// training_step(run_optimizer = 1) on one GPU sums the network kernel's weight-gradient slabs inside the optimizer's launch (AdamFinalize);
// tcnn_set_finalize_in_optimizer(0): always k_mlp_finalize_gradients, a launch of its own behind the network kernel (A/B runs, tests)
static std::atomic<int> g_finalize_in_optimizer{1};
static bool backward_recomputes(const Model& md) { return g_fused_network_passes.load() != 0 && md.has_network && mlp_train_supported(md.net.mlp); }

// NetworkWithInputEncoding::forward_impl / inference_mixed_precision_impl (:60-81).  ctx == nullptr: inference.
static void model_forward(hipStream_t stream, const Model& md, uint32_t n, const float* input, half_t* output, const half_t* params,
                          ForwardCtx* ctx, bool prepare_input_gradients, const MlpF32Output* f32 = nullptr) {
	check_batch(n, widest_matrix(md));
	if (n == 0) return;
	if (ctx) {
		ctx->stream = stream;
		ctx->n = n;
	}
	float* dy_dx = nullptr;
	if (ctx && prepare_input_gradients && md.enc.is_grid) {
		ctx->dy_dx = Scratch(stream, (size_t)md.enc.n_output_dims * n * md.n_input_dims * sizeof(float));
		dy_dx = ctx->dy_dx.as<float>();
	}
	if (!md.has_network) {
		encoding_forward(stream, md, n, input, params, output, /*soa=*/false, dy_dx);
		return;
	}
	// inference into the caller's fp32 matrix with an Identity encoding that pads nothing: the inference kernel reads the fp32 input itself
	// (MlpF32Input; no encoding kernel, no encoded matrix)
	const EncodingDesc& e = md.enc;
	if (!ctx && !e.is_grid && !e.is_frequency && !e.is_oneblob && e.n_dims == e.padded_output_width && in_stride_i(md) == e.n_dims && in_stride_d() == 1u &&
	    ((uintptr_t)input & 15u) == 0u && g_fused_identity_input.load() != 0 && mlp_infer_f32_input_supported(md.net.mlp, n)) {
		MlpF32Input f32_input;
		f32_input.x = input;
		f32_input.scale = e.id_scale;
		f32_input.offset = e.id_offset;
		ProfScope prof(stream, STAGE_MLP_FWD);
		// into the caller's fp32 matrix (network->inference) or into the padded 16-bit matrix (a module's inference, cpp_api.h:97)
		mlp_infer_wave(stream, md.net.mlp, n, params, nullptr, f32 ? nullptr : output, f32 ? *f32 : MlpF32Output(), &f32_input);
		return;
	}
	Scratch enc_local;
	Scratch& enc = ctx ? ctx->enc : enc_local;
	enc = Scratch(stream, (size_t)md.enc.padded_output_width * n * sizeof(half_t));
	encoding_forward(stream, md, n, input, params + md.n_mlp_params(), enc.as<half_t>(), /*soa=*/true, dy_dx);
	half_t* hidden = nullptr;
	if (ctx && !backward_recomputes(md)) {
		ctx->hidden = Scratch(stream, (size_t)md.net.n_hidden_layers * n * md.net.mlp.width * sizeof(half_t));
		hidden = ctx->hidden.as<half_t>();
	}
	ProfScope prof(stream, STAGE_MLP_FWD);
	if (f32) {  // (inference_to_f32 below checked that the register-resident inference kernel takes this network)
		mlp_infer_wave(stream, md.net.mlp, n, params, enc.as<half_t>(), nullptr, *f32);
		return;
	}
	mlp_forward(stream, md.net.mlp, n, params, enc.as<half_t>(), hidden, output);
}

// network->inference into the caller's fp32 matrix (object.h:214-271).  Where the register-resident inference kernel runs the network it
// writes the fp32 elements itself; otherwise the padded 16-bit result goes through trim_and_cast as in the reference.
static void inference_to_f32(hipStream_t stream, const Model& md, uint32_t n, const float* input, const half_t* params, float* out, uint32_t stride_i, uint32_t stride_j) {
	const uint32_t padded = md.padded_output_width(), width = md.output_width();
	if (md.has_network && n > 0 && mlp_infer_wave_supported(md.net.mlp, n)) {
		const MlpF32Output f32 = {out, width, stride_i, stride_j};
		model_forward(stream, md, n, input, nullptr, params, nullptr, false, &f32);
		return;
	}
	Scratch tmp(stream, (size_t)padded * n * sizeof(half_t));  // object.h:260
	model_forward(stream, md, n, input, tmp.as<half_t>(), params, nullptr, false);
	trim_and_cast(stream, n, padded, width, tmp.as<half_t>(), out, stride_i, stride_j);  // object.h:269-270
}

// The grid's parameter gradients in groups of consecutive levels, each reported as soon as its kernels are enqueued (data-parallel
// hosts start that group's exchange while the next group is still being computed): `ready(ctx, begin, end)` with the parameter range
// relative to the model's first parameter.
struct LevelGroups {
	uint32_t n_groups = 1;
	void (*ready)(void* ctx, size_t begin, size_t end) = nullptr;
	void* ctx = nullptr;
};
static void encoding_backward(hipStream_t stream, const Model& md, const ForwardCtx& ctx, uint32_t n, float* dL_dinput, const half_t* dL_denc,
                              uint32_t stride_k, uint32_t stride_i, half_t* dL_dparams, bool want_grads, bool accumulate, const float* input,
                              uint32_t lds_level_budget, const LevelGroups* groups = nullptr);
static uint32_t widest_matrix(const Model& md) {
	uint32_t w = std::max(md.enc.padded_output_width, md.n_input_dims);
	if (md.has_network) w = std::max(w, std::max(md.net.mlp.width * md.net.n_hidden_layers, md.net.mlp.padded_out));
	if (md.enc.is_grid) w = std::max(w, md.enc.n_output_dims * md.n_input_dims);  // dy_dx
	return w;
}

// NetworkWithInputEncoding::backward_impl (:83-113) / GridEncodingTemplated::backward_impl (grid.h:817-908)
static void model_backward(hipStream_t stream, const Model& md, const ForwardCtx& ctx, uint32_t n, float* dL_dinput, const half_t* dL_doutput,
                           half_t* dL_dparams, const float* input, const half_t* output, const half_t* params, int gradient_mode,
                           uint32_t lds_level_budget) {
	check_batch(n, widest_matrix(md));
	if (n == 0) return;
	if (ctx.n != n) throw std::runtime_error("backward: batch size does not match the forward context");
	const bool recompute = md.has_network && !ctx.hidden.ptr;  // the context holds the encoded input only (see g_fused_network_passes)
	if (recompute && (!ctx.enc.ptr || !mlp_train_supported(md.net.mlp))) {
		throw std::runtime_error("backward: this context holds no saved activations and the network has no single-kernel backward pass");
	}
	const bool want_grads = gradient_mode != TCNN_GRADIENT_IGNORE && dL_dparams != nullptr;
	const bool accumulate = gradient_mode == TCNN_GRADIENT_ACCUMULATE;
	const EncodingDesc& e = md.enc;
	if (!want_grads && !dL_dinput) return;

	const half_t* dL_denc = dL_doutput;  // bare encoding: gradient of the encoding output, sample-major
	uint32_t stride_k = 1u, stride_i = e.padded_output_width;
	Scratch denc;
	if (md.has_network) {
		const bool need_denc = (want_grads && e.n_params > 0) || dL_dinput;
		ProfScope prof(stream, STAGE_MLP_BWD);
		Scratch params_t(stream, md.n_mlp_params() * sizeof(half_t));
		mlp_transpose_weights(stream, md.net.mlp, params, params_t.as<half_t>());
		const uint32_t n_partials = recompute ? mlp_train_n_partials(md.net.mlp, n, LossType::L2) : mlp_backward_n_partials(md.net.mlp, n);
		Scratch partials;
		if (want_grads) partials = Scratch(stream, (size_t)n_partials * md.n_mlp_params() * sizeof(float));
		if (need_denc) denc = Scratch(stream, (size_t)e.padded_output_width * n * sizeof(half_t));
		if (recompute) {  // forward from the encoded input again, then backward from the caller's dL/doutput, in one kernel
			MlpLossArgs la = {LossType::L2, nullptr, nullptr, md.output_width(), 1.0f, 1u};
			la.external_dL_doutput = dL_doutput;
			const SlabOrder order = mlp_train(stream, md.net.mlp, n, params, params_t.as<half_t>(), ctx.enc.as<half_t>(), la, nullptr, nullptr,
			                                  need_denc ? denc.as<half_t>() : nullptr, want_grads ? partials.as<float>() : nullptr, nullptr);
			if (want_grads) mlp_finalize_gradients(stream, md.net.mlp, n_partials, partials.as<float>(), dL_dparams, accumulate, order);
			if (!need_denc) return;
			dL_denc = denc.as<half_t>();
			stride_k = n;
			stride_i = 1u;
			encoding_backward(stream, md, ctx, n, dL_dinput, dL_denc, stride_k, stride_i, dL_dparams, want_grads, accumulate, input, lds_level_budget);
			return;
		}
		Scratch dpre;  // output activation: continue from dL/d(pre-activation) (fully_fused_mlp.cu:760-763)
		if (md.net.mlp.output_activation != (uint32_t)Activation::None) {
			if (!output) throw std::runtime_error("backward: the network output is required when an output activation is set");
			dpre = Scratch(stream, (size_t)md.padded_output_width() * n * sizeof(half_t));
			mlp_output_activation_backward(stream, md.net.mlp, n, output, dL_doutput, dpre.as<half_t>());
			dL_doutput = dpre.as<half_t>();
		}
		Scratch deep;
		if (const size_t deep_bytes = mlp_backward_workspace_bytes(md.net.mlp, n)) deep = Scratch(stream, deep_bytes);
		mlp_backward(stream, md.net.mlp, n, params_t.as<half_t>(), ctx.enc.as<half_t>(), ctx.hidden.as<half_t>(), dL_doutput,
		             need_denc ? denc.as<half_t>() : nullptr, want_grads ? partials.as<float>() : nullptr, deep.ptr);
		if (want_grads) mlp_finalize_gradients(stream, md.net.mlp, n_partials, partials.as<float>(), dL_dparams, accumulate);
		if (!need_denc) return;
		dL_denc = denc.as<half_t>();
		stride_k = n;
		stride_i = 1u;
	}

	encoding_backward(stream, md, ctx, n, dL_dinput, dL_denc, stride_k, stride_i, dL_dparams, want_grads, accumulate, input, lds_level_budget);
}

// the encoding's share of the backward pass: dL_denc has element (feature k, sample i) at [k * stride_k + i * stride_i]
// levels [a, b) of a grid as a grid of their own: the kernels index dL_dy and the gradients from the first of them
static GridMeta grid_levels(const GridMeta& g, uint32_t a, uint32_t b) {
	GridMeta s = g;
	s.n_levels = b - a;
	for (uint32_t l = 0; l <= b - a; ++l) s.offset[l] = g.offset[a + l] - g.offset[a];
	for (uint32_t l = 0; l < b - a; ++l) {
		s.scale[l] = g.scale[a + l];
		s.resolution[l] = g.resolution[a + l];
	}
	return s;
}

// consecutive levels in `n_groups` groups, cut where the running parameter count passes k / n_groups of the total
static std::vector<std::pair<uint32_t, uint32_t>> split_levels(const GridMeta& g, uint32_t n_groups) {
	const uint32_t L = g.n_levels;
	std::vector<std::pair<uint32_t, uint32_t>> level_ranges;
	for (uint32_t k = 1, a = 0; k <= n_groups && a < L; ++k) {
		uint32_t b = a + 1;
		const uint64_t target = (uint64_t)g.offset[L] * k / n_groups;
		while (b < L && (k == n_groups || g.offset[b] < target)) ++b;
		if (k == n_groups) b = L;
		level_ranges.push_back({a, b});
		a = b;
	}
	return level_ranges;
}

static void encoding_backward(hipStream_t stream, const Model& md, const ForwardCtx& ctx, uint32_t n, float* dL_dinput, const half_t* dL_denc,
                              uint32_t stride_k, uint32_t stride_i, half_t* dL_dparams, bool want_grads, bool accumulate, const float* input,
                              uint32_t lds_level_budget, const LevelGroups* groups) {
	const EncodingDesc& e = md.enc;
	if (e.is_grid) {
		GridIO io = {input, in_stride_i(md), in_stride_d(), n, stride_k, stride_i};
		if (want_grads && e.n_params > 0) {
			half_t* grid_grads = dL_dparams + md.n_mlp_params();
			// Overwrite vs Accumulate (grid.h:865-867) is handled inside: the owner-computes kernel stores
			// whole slices, so the reference's full-table memset is only issued for the atomic A/B mode.
			const GridBackwardMode mode = (GridBackwardMode)g_grid_backward_mode.load();
			if (lds_level_budget == 0) lds_level_budget = g_default_lds_slice_bytes;
			// level groups: only where a level's treatment does not depend on its index among ALL levels (every level switched on,
			// no per-level random stream) and nothing else rides on the pass
			const uint32_t L = e.grid.n_levels, F = e.grid.n_feat;
			uint32_t n_groups = groups ? std::min(std::max(groups->n_groups, 1u), L) : 1u;
			if (e.grid.max_level < 1.0f || e.grid.stochastic != 0u) n_groups = 1;
			const std::vector<std::pair<uint32_t, uint32_t>> level_ranges = split_levels(e.grid, n_groups);
			// one workspace for all groups: the largest any of them asks for (a group of later levels can bucket levels that the plan of
			// the whole grid, which takes the first MAX_BUCKET_LEVELS eligible ones, left to the other kinds)
			GridBackwardWorkspace ws = grid_backward_workspace_size(e.grid, n, mode, lds_level_budget);
			if (level_ranges.size() > 1) {
				ws = GridBackwardWorkspace();
				for (const auto& r : level_ranges) {
					const GridBackwardWorkspace w = grid_backward_workspace_size(grid_levels(e.grid, r.first, r.second), n, mode, lds_level_budget);
					ws.scratch_bytes = std::max(ws.scratch_bytes, w.scratch_bytes);
					ws.n_counters = std::max(ws.n_counters, w.n_counters);
				}
			}
			Scratch queues;
			if (ws.scratch_bytes) {
				queues = Scratch(stream, ws.scratch_bytes);
				ws.scratch = queues.ptr;
				ws.scratch_bytes = queues.bytes;
				ws.counters = ZeroedCounters::get(stream, ws.n_counters);
			}
			ws.phase_hook = grid_backward_phase_hook;  // per-kernel timing when a profiler is attached
			ws.hook_user = (void*)stream;
			if (level_ranges.size() <= 1) {
				grid_backward(stream, e.grid, io, dL_denc, grid_grads, accumulate, mode, lds_level_budget, ws);
				if (groups && groups->ready) groups->ready(groups->ctx, md.n_mlp_params(), md.n_mlp_params() + (size_t)e.grid.offset[L] * F);
			} else {
				struct CountsGuard {
					~CountsGuard() { g_phase_hook_counts = true; }
				} counts_guard;
				for (const auto& r : level_ranges) {
					const uint32_t a = r.first, b = r.second;
					const GridMeta sub = grid_levels(e.grid, a, b);
					g_phase_hook_counts = a == 0;  // one backward pass per step, however many launches it takes
					grid_backward(stream, sub, io, dL_denc + (size_t)a * F * stride_k, grid_grads + (size_t)e.grid.offset[a] * F, accumulate, mode, lds_level_budget, ws);
					if (groups->ready) groups->ready(groups->ctx, md.n_mlp_params() + (size_t)e.grid.offset[a] * F, md.n_mlp_params() + (size_t)e.grid.offset[b] * F);
				}
			}
		}
		if (dL_dinput) {
			if (!ctx.dy_dx.ptr) throw std::runtime_error("backward: dL_dinput requested but forward was not run with prepare_input_gradients");
			grid_backward_input(stream, md.n_input_dims, e.n_output_dims, io, dL_denc, ctx.dy_dx.as<float>(), dL_dinput, dx_stride_i(md), dx_stride_d());
		}
	} else if (dL_dinput && e.is_frequency) {
		frequency_backward(stream, n, e.n_dims, e.n_frequencies, dL_denc, stride_k, stride_i, input, in_stride_i(md), in_stride_d(), dL_dinput, dx_stride_i(md), dx_stride_d());
	} else if (dL_dinput && e.is_oneblob) {
		oneblob_backward(stream, n, e.n_dims, e.n_bins, dL_denc, stride_k, stride_i, input, in_stride_i(md), in_stride_d(), dL_dinput, dx_stride_i(md), dx_stride_d());
	} else if (dL_dinput) {
		identity_backward(stream, n, e.n_dims, e.id_scale, dL_denc, stride_k, stride_i, dL_dinput, dx_stride_i(md), dx_stride_d());
	}
}

// ---- Encoding<float> (create_encoding(..., Precision::Fp32), cpp_api.cu:165-168): a bare encoding whose parameters, output and gradients are
// fp32 and which COMPUTES in fp32, as the reference's instantiation does.  Output sample-major [n][padded] (cpp_api.cu:94-95).
static void encoding_forward_f32(hipStream_t stream, const Model& md, uint32_t n, const float* input, const float* params, float* out, ForwardCtx* ctx,
                                 bool prepare_input_gradients) {
	check_batch(n, widest_matrix(md));
	if (n == 0) return;
	const EncodingDesc& e = md.enc;
	if (ctx) {
		ctx->stream = stream;
		ctx->n = n;
	}
	const uint32_t stride_k = 1u, stride_i = e.padded_output_width;
	if (e.is_grid) {
		float* dy_dx = nullptr;
		if (ctx && prepare_input_gradients) {
			ctx->dy_dx = Scratch(stream, (size_t)e.n_output_dims * n * md.n_input_dims * sizeof(float));
			dy_dx = ctx->dy_dx.as<float>();
		}
		GridIO io = {input, in_stride_i(md), in_stride_d(), n, stride_k, stride_i};
		grid_forward_f32(stream, e.grid, io, params, out, dy_dx);
		const uint32_t n_to_pad = e.padded_output_width - e.n_output_dims;
		if (n_to_pad > 0) HIP_CHECK(hipMemset2DAsync(out + e.n_output_dims, (size_t)e.padded_output_width * sizeof(float), 0, (size_t)n_to_pad * sizeof(float), n, stream));
	} else if (e.is_frequency) {
		frequency_forward(stream, n, e.n_dims, e.n_frequencies, e.padded_output_width, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	} else if (e.is_oneblob) {
		oneblob_forward(stream, n, e.n_dims, e.n_bins, e.padded_output_width, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	} else {
		identity_forward(stream, n, e.n_dims, e.padded_output_width, e.id_scale, e.id_offset, input, in_stride_i(md), in_stride_d(), out, stride_k, stride_i);
	}
}
static void encoding_backward_f32(hipStream_t stream, const Model& md, const ForwardCtx& ctx, uint32_t n, float* dL_dinput, const float* dL_doutput, float* dL_dparams,
                                  const float* input) {
	check_batch(n, widest_matrix(md));
	if (n == 0) return;
	if (ctx.n != n) throw std::runtime_error("backward: batch size does not match the forward context");
	const EncodingDesc& e = md.enc;
	const uint32_t stride_k = 1u, stride_i = e.padded_output_width;
	if (e.is_grid) {
		GridIO io = {input, in_stride_i(md), in_stride_d(), n, stride_k, stride_i};
		if (dL_dparams && e.n_params > 0) grid_backward_f32(stream, e.grid, io, dL_doutput, dL_dparams, /*accumulate=*/false);  // GradientMode::Overwrite, cpp_api.cu:115
		if (dL_dinput) {
			if (!ctx.dy_dx.ptr) throw std::runtime_error("backward: dL_dinput requested but forward was not run with prepare_input_gradients");
			grid_backward_input_f32(stream, md.n_input_dims, e.n_output_dims, io, dL_doutput, ctx.dy_dx.as<float>(), dL_dinput, dx_stride_i(md), dx_stride_d());
		}
	} else if (dL_dinput && e.is_frequency) {
		frequency_backward(stream, n, e.n_dims, e.n_frequencies, dL_doutput, stride_k, stride_i, input, in_stride_i(md), in_stride_d(), dL_dinput, dx_stride_i(md), dx_stride_d());
	} else if (dL_dinput && e.is_oneblob) {
		oneblob_backward(stream, n, e.n_dims, e.n_bins, dL_doutput, stride_k, stride_i, input, in_stride_i(md), in_stride_d(), dL_dinput, dx_stride_i(md), dx_stride_d());
	} else if (dL_dinput) {
		identity_backward(stream, n, e.n_dims, e.id_scale, dL_doutput, stride_k, stride_i, dL_dinput, dx_stride_i(md), dx_stride_d());
	}
}

static Model make_nwie(uint32_t n_input_dims, uint32_t n_output_dims, const Json& encoding, const Json& network) {
	Model md;
	md.n_input_dims = n_input_dims;
	md.enc = create_encoding_desc(n_input_dims, encoding, /*minimum_alignment(network)=*/16);  // network.cu:79-98 -> 16
	md.has_network = true;
	md.net = create_network_desc(md.enc.padded_output_width, n_output_dims, network);
	md.finish();
	return md;
}

}  // namespace tcnn_hip

// =================================================================================================
// C ABI
// =================================================================================================
using namespace tcnn_hip;

struct tcnn_module {
	Model md;
	std::string name;
	uint32_t lds_level_budget = 0;  // 0: default LDS slice size of the sliced grid backward
	// create_encoding(..., Precision::Fp32) (cpp_api.cu:165-174 -> Encoding<float>): parameters, outputs and gradients are fp32 and the
	// first-order passes COMPUTE in fp32 (encoding_forward_f32 / encoding_backward_f32: the grid's kernel_grid<float> formulation, fp32
	// global atomics for its parameter gradients; frequency / one-blob / identity with float values).  Only the second-order pass
	// (backward_backward_input, grid only) still goes through 16-bit stand-ins of the caller's tensors: incoming gradients are scaled
	// into that type's range by the largest power of two <= FP32_GRADIENT_SCALE that keeps max |dL_doutput| * scale <=
	// FP32_GRADIENT_TARGET (found on the device per call) and the results scaled back, exactly.
	bool fp32_io = false;
};
static constexpr float FP32_GRADIENT_SCALE = 1024.0f, FP32_GRADIENT_TARGET = 16384.0f;
// the 16-bit copies an fp32 module works on
struct Fp32Bridge {
	Scratch params, output, dL_doutput, dL_dparams;
	static Scratch to_half(hipStream_t stream, const void* src, size_t n, float scale = 1.0f) {
		Scratch s(stream, std::max<size_t>(n, 1) * sizeof(half_t));
		cast_scaled_f32_to_f16(stream, n, (const float*)src, s.as<half_t>(), scale);
		return s;
	}
	// the gradient entering the backward pass: scaled by a per-call power of two left in `scale_pair` ({scale, 1 / scale} on the device)
	static Scratch gradient_to_half(hipStream_t stream, const void* src, size_t n, Scratch& scale_pair) {
		scale_pair = Scratch(stream, 4 * sizeof(float));
		gradient_scale_from_absmax(stream, n, (const float*)src, scale_pair.as<float>(), FP32_GRADIENT_SCALE, FP32_GRADIENT_TARGET);
		Scratch s(stream, std::max<size_t>(n, 1) * sizeof(half_t));
		cast_scaled_f32_to_f16(stream, n, (const float*)src, s.as<half_t>(), (const float*)scale_pair.as<float>());
		return s;
	}
};
struct tcnn_context {
	ForwardCtx ctx;
};

struct tcnn_train_context {
	ForwardCtx model_ctx;
	Scratch output;       // half [n][padded]
	Scratch dL_doutput;   // half [n][padded]
	const half_t* dL_doutput_ptr = nullptr;  // == dL_doutput or the caller's external_dL_dy
	Scratch block_sums;   // fp32 partial loss sums
	uint32_t n_block_sums = 0;
	uint32_t n = 0;
	hipStream_t stream = nullptr;
};

struct tcnn_trainable_model {
	Model md;
	LossType loss = LossType::RelativeL2;
	AdamHyper adam;
	// wrapper optimizers around Adam (optimizers/ema.h, exponential_decay.h), outermost first
	std::vector<std::string> optimizer_order;  // e.g. {"Ema", "ExponentialDecay"}
	bool ema = false, ema_full_precision = false;
	float ema_decay = 0.99f;
	half_t* params_ema = nullptr;  // custom_weights(): the inference parameters while EMA is on (trainer.h:497-500)
	float* ema_tmp = nullptr;      // fp32 shadow of the average (full_precision)
	bool lr_decay = false;
	float decay_base = 0.1f, lr_factor = 1.0f, base_lr = 0.0f;
	uint32_t decay_interval = 10000, decay_start = 10000, decay_end = 10000000;
	half_t* inference_params() const { return ema ? params_ema : params; }
	// transposed copy of the network weights for the backward kernels; Adam keeps it current, anything else that writes
	// `params` invalidates it
	half_t* params_t = nullptr;
	bool params_t_valid = false;
	// a mutable pointer to `params` has left the library (tcnn_trainer_params / _params_inference): the caller may write
	// through it at any time, so from then on the transposed copy is rebuilt before every pass that needs it
	bool params_exposed = false;
	uint32_t optimizer_step = 0;
	Pcg32 rng;
	void* buffer = nullptr;  // [fp32 master | half params | half grads], trainer.h:76, 489-495
	float* master = nullptr;
	half_t* params = nullptr;
	half_t* grads = nullptr;
	float *m1 = nullptr, *m2 = nullptr;
	uint32_t* steps = nullptr;
	// `steps` holds the counters' deficits instead (elementwise_kernels.h: adam_flip_step_representation) while most
	// table entries are stepped every time; chosen per optimizer step from the last batch size, see choose_step_representation
	int steps_form = ADAM_STEPS_COUNTERS;  // AdamStepsForm of `steps` (+ `step_deficits8` for the byte form)
	uint8_t* step_deficits8 = nullptr;     // n_params bytes
	uint32_t last_batch = 0;
	uint64_t global_batch = 0;
	uint32_t lds_level_budget = 0;  // 0: default LDS slice size of the sliced grid backward
	std::string hyper_json;
	float* loss_scratch = nullptr;  // 1024 + 1 floats
	std::unique_ptr<Profiler> profiler;  // null unless tcnn_trainer_set_profiling enabled it
	// training_step on one GPU: the network kernel's fp32 weight-gradient slabs of THIS step, summed inside the optimizer's launch instead
	// of by a kernel of their own (AdamFinalize); set by training_step_fused for the optimizer step it runs itself, empty otherwise
	AdamFinalize pending_finalize;
	// training_step as ONE graph launch (tcnn_trainer_set_graph_capture; Trainer::training_step runs its passes under CudaGraph::capture_guard,
	// trainer.h:343-350, cuda_graph.h:65-155): every call re-records its launches into a graph, patches the instantiated graph with it and
	// launches that.  `graph_warm`: the shape (batch size and the flags that decide which scratch blocks a step takes) whose step has run
	// once outside a capture, so that the capture finds every block in the stream's cache and allocates nothing.
	bool graph_capture = false;
	hipGraphExec_t graph_exec = nullptr;
	uint64_t graph_warm = ~0ull;
	uint64_t graph_launches = 0, graph_instantiations = 0;
	// data-parallel hosts: called between backward and the optimizer (tcnn_trainer_set_gradient_exchange)
	void (*exchange)(void* user, void* gradients_fp16, size_t n_params, tcnn_stream_t stream) = nullptr;
	void* exchange_user = nullptr;
	// data-parallel hosts that overlap the exchange with the backward pass (tcnn_trainer_set_gradient_ready_callback,
	// tcnn_trainer_set_backward_level_groups, tcnn_trainer_enable_rccl)
	void (*gradients_ready)(void* user, size_t begin, size_t end, tcnn_stream_t stream) = nullptr;
	void* ready_user = nullptr;
	uint32_t backward_level_groups = 1u;
	void* rccl_comm = nullptr;  // ncclComm_t
	int rccl_ranks = 0;
	// gradient exchange over peer-mapped memory (direct_exchange.h; tcnn_trainer_direct_*)
	DirectExchange direct;
	// sharded exchange inside the library (tcnn_trainer_enable_rccl_sharded): reduce-scatter of every ready range -> Adam on this rank's
	// shards -> all-gather of the 16-bit parameters; -1: the all-reduce scheme
	int rccl_rank = -1;
	hipStream_t comm_stream = nullptr;
	std::vector<hipEvent_t> comm_events;
	size_t comm_events_used = 0;
	struct ReducedRange {
		size_t begin, end;
		hipEvent_t done;
		size_t shard = 0;  // sharded scheme: parameters per rank of this range's evenly divided part [begin, begin + shard * ranks); the rest is all-reduced
	};
	std::vector<ReducedRange> reduced;  // this step's ranges whose all-reduce is in flight on comm_stream, in issue order
	hipEvent_t comm_event() {
		if (comm_events_used == comm_events.size()) {
			hipEvent_t e;
			HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			comm_events.push_back(e);
		}
		return comm_events[comm_events_used++];
	}
};

// RCCL, loaded at run time: the library links no collective library, a host that never asks for it never loads one
struct Rccl {
	void* handle = nullptr;
	int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;  // ncclReduceScatter(send, recv, recvcount, type, op, comm, stream)
	int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;           // ncclAllGather(send, recv, sendcount, type, comm, stream)
	int (*group_start)() = nullptr;
	int (*group_end)() = nullptr;
	int (*comm_get_async_error)(void*, int*) = nullptr;  // ncclCommGetAsyncError(comm, ncclResult_t*)
	const char* (*error_string)(int) = nullptr;
	static Rccl& get() {
		static Rccl r = [] {
			Rccl x;
			for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
				x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
				if (x.handle) break;
			}
			if (x.handle) {
				x.all_reduce = (decltype(x.all_reduce))dlsym(x.handle, "ncclAllReduce");
				x.reduce_scatter = (decltype(x.reduce_scatter))dlsym(x.handle, "ncclReduceScatter");
				x.all_gather = (decltype(x.all_gather))dlsym(x.handle, "ncclAllGather");
				x.group_start = (decltype(x.group_start))dlsym(x.handle, "ncclGroupStart");
				x.group_end = (decltype(x.group_end))dlsym(x.handle, "ncclGroupEnd");
				x.comm_get_async_error = (decltype(x.comm_get_async_error))dlsym(x.handle, "ncclCommGetAsyncError");
				x.error_string = (decltype(x.error_string))dlsym(x.handle, "ncclGetErrorString");
			}
			return x;
		}();
		return r;
	}
};
constexpr int RCCL_SUM = 0, RCCL_HALF = 6, RCCL_BFLOAT16 = 9;  // rccl.h: ncclSum, ncclFloat16, ncclBfloat16
constexpr int RCCL_SUCCESS = 0, RCCL_IN_PROGRESS = 7;           // ncclSuccess, ncclInProgress
static void rccl_check(const Rccl& r, int rc, const char* what) {
	if (rc != RCCL_SUCCESS) throw std::runtime_error(std::string(what) + " failed: " + (r.error_string ? r.error_string(rc) : "?"));
}
// Collectives fail ASYNCHRONOUSLY (a peer that died, a link error): RCCL records the error on the communicator and the kernels already
// enqueued would wait forever.  Polled at every point where this library is about to put the compute stream behind a collective.
static void rccl_poll_async_error(tcnn_trainable_model* tm) {
	if (!tm->rccl_comm) return;
	const Rccl& r = Rccl::get();
	if (!r.comm_get_async_error) return;
	int state = RCCL_SUCCESS;
	const int rc = r.comm_get_async_error(tm->rccl_comm, &state);
	if (rc != RCCL_SUCCESS) throw std::runtime_error(std::string("ncclCommGetAsyncError failed: ") + (r.error_string ? r.error_string(rc) : "?"));
	if (state != RCCL_SUCCESS && state != RCCL_IN_PROGRESS) {
		throw std::runtime_error(std::string("RCCL reported an asynchronous error on the communicator: ") + (r.error_string ? r.error_string(state) : "?") +
		                         " (the gradient exchange of this step cannot complete; destroy the communicator and the trainer's rccl hook)");
	}
}

// Gradients [begin, end) of this step are final once the work enqueued on `stream` so far has run: tell the host (callback) and /
// or start their all-reduce on the communication stream, behind an event -- the rest of the backward pass keeps the compute
// stream busy meanwhile.
static void notify_gradients_ready(tcnn_trainable_model* tm, hipStream_t stream, size_t begin, size_t end) {
	if (begin >= end) return;
	if (tm->gradients_ready) tm->gradients_ready(tm->ready_user, begin, end, stream);
	if (tm->rccl_comm) {
		Rccl& r = Rccl::get();
		rccl_poll_async_error(tm);
		hipEvent_t ready = tm->comm_event(), done = tm->comm_event();
		HIP_CHECK(hipEventRecord(ready, stream));
		HIP_CHECK(hipStreamWaitEvent(tm->comm_stream, ready, 0));
		const int type = HALF_IS_BF16 ? RCCL_BFLOAT16 : RCCL_HALF;
		size_t shard = 0;
		if (tm->rccl_rank >= 0) {
			// sharded: the part of the range that divides evenly over the ranks (shards of a multiple of 8 parameters, what the ranged optimizer
			// step needs) is reduce-scattered IN PLACE (recv = send + rank * shard: RCCL's in-place form), the remainder all-reduced
			const size_t P = (size_t)tm->rccl_ranks;
			shard = ((end - begin) / (8 * P)) * 8;
			if (shard) rccl_check(r, r.reduce_scatter(tm->grads + begin, tm->grads + begin + (size_t)tm->rccl_rank * shard, shard, type, RCCL_SUM, tm->rccl_comm, tm->comm_stream), "ncclReduceScatter");
			if (begin + shard * P < end) rccl_check(r, r.all_reduce(tm->grads + begin + shard * P, tm->grads + begin + shard * P, end - begin - shard * P, type, RCCL_SUM, tm->rccl_comm, tm->comm_stream), "ncclAllReduce");
		} else {
			rccl_check(r, r.all_reduce(tm->grads + begin, tm->grads + begin, end - begin, type, RCCL_SUM, tm->rccl_comm, tm->comm_stream), "ncclAllReduce");
		}
		HIP_CHECK(hipEventRecord(done, tm->comm_stream));
		tm->reduced.push_back({begin, end, done, shard});
	}
}

#define TCNN_API_BEGIN try {
#define TCNN_API_END                             \
	}                                            \
	catch (const std::exception& ex) {           \
		g_last_error = ex.what();                \
		log_message(TCNN_LOG_ERROR, ex.what());  \
		return TCNN_ERROR;                       \
	}                                            \
	return TCNN_OK;

extern "C" {

const char* tcnn_last_error(void) { return g_last_error.c_str(); }
uint32_t tcnn_batch_size_granularity(void) { return BATCH_SIZE_GRANULARITY; }
int tcnn_hip_device(void) {
	int d = -1;
	(void)hipGetDevice(&d);
	return d;
}
int tcnn_set_hip_device(int device) {
	TCNN_API_BEGIN
	HIP_CHECK(hipSetDevice(device));
	TCNN_API_END
}
void tcnn_free_temporary_memory(void) {
	ScratchCache::free_all();
	ZeroedCounters::free_all();
}
int tcnn_device_malloc(size_t bytes, void** out) {
	TCNN_API_BEGIN
	*out = device_malloc(bytes);
	TCNN_API_END
}
void tcnn_device_free(void* ptr) { device_free(ptr); }
int tcnn_debug_alloc_mode(void) { return (int)debug_alloc_mode(); }
int tcnn_debug_check_allocations(void) {
	if (debug_alloc_mode() == DebugAlloc::Off) return 0;
	std::string report;
	const size_t bad = DebugAllocator::get().check_all(&report);
	g_last_error = report;
	if (bad) log_message(TCNN_LOG_ERROR, "debug allocator: " + report);
	return (int)bad;
}
int tcnn_set_debug_launches(int enable) {
	debug_launch_flags() = (debug_launch_flags() & ~1) | (enable ? 1 : 0);
	return TCNN_OK;
}
int tcnn_has_networks(void) { return 1; }
// this build's 16-bit type (tcnn_device.h): fp16, or bfloat16 when compiled with -DTCNN_BF16
static constexpr int NATIVE_PRECISION = HALF_IS_BF16 ? TCNN_PRECISION_BF16 : TCNN_PRECISION_FP16;
static const char* const NATIVE_TYPE_NAME = HALF_IS_BF16 ? "__nv_bfloat16" : "__half";  // gpu_memory_json.h type strings
// the loss scale is kept at 128 for bfloat16 as well: harmless for its range, and the exact fixed-point accumulation of
// the grid backward (2^-24 resolution) relies on gradients of that magnitude
float tcnn_default_loss_scale(int precision) { return precision == TCNN_PRECISION_FP32 ? 1.0f : LOSS_SCALE_FP16; }
int tcnn_preferred_precision(void) { return NATIVE_PRECISION; }
int tcnn_supports_jit_fusion(int) { return 0; }
void tcnn_set_log_callback(void (*callback)(int, const char*)) { g_log_callback = callback; }

int tcnn_generate_random_uniform(tcnn_stream_t stream, uint64_t seed, uint64_t* position, size_t n, float* out, float lower, float upper) {
	TCNN_API_BEGIN
	Pcg32 rng{seed};
	if (position && *position) rng.advance((int64_t)*position);
	generate_random_uniform((hipStream_t)stream, rng, n, out, lower, upper);
	if (position) *position += n;
	TCNN_API_END
}

int tcnn_generate_sinusoid_targets(tcnn_stream_t stream, uint32_t n, uint32_t n_input_dims, uint32_t n_output_dims, const float* positions, float* targets) {
	TCNN_API_BEGIN
	if (n_input_dims == 0) throw std::runtime_error("tcnn_generate_sinusoid_targets: n_input_dims must be positive");
	if ((uint64_t)n * n_output_dims > 0xFFFFFFFFull) throw std::runtime_error("tcnn_generate_sinusoid_targets: batch too large");
	sinusoid_targets((hipStream_t)stream, n, n_input_dims, n_output_dims, positions, targets);
	TCNN_API_END
}

int tcnn_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const char* encoding_json, const char* network_json,
                                            tcnn_module_t** out) {
	TCNN_API_BEGIN
	auto m = std::make_unique<tcnn_module>();
	m->md = make_nwie(n_input_dims, n_output_dims, Json::parse(encoding_json), Json::parse(network_json));
	m->name = m->md.name();
	*out = m.release();
	TCNN_API_END
}

int tcnn_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json, tcnn_module_t** out) {
	return tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, "{\"otype\": \"Identity\"}", network_json, out);  // cpp_api.cu:160-162
}

int tcnn_create_encoding(uint32_t n_input_dims, const char* encoding_json, int requested_precision, tcnn_module_t** out) {
	TCNN_API_BEGIN
	if (requested_precision != NATIVE_PRECISION && requested_precision != TCNN_PRECISION_FP32) {
		g_last_error = HALF_IS_BF16 ? "create_encoding: this build (libtcnn_hip_bf16.so) provides bf16 and fp32 encodings"
		                            : "create_encoding: this build provides fp16 and fp32 encodings";
		return TCNN_ERROR_UNSUPPORTED;
	}
	auto m = std::make_unique<tcnn_module>();
	m->fp32_io = requested_precision == TCNN_PRECISION_FP32;
	m->md.n_input_dims = n_input_dims;
	m->md.enc = create_encoding_desc(n_input_dims, Json::parse(encoding_json), /*alignment=*/0);  // cpp_api.cu:165-174
	m->md.has_network = false;
	m->md.finish();
	m->name = m->md.name();
	*out = m.release();
	TCNN_API_END
}

void tcnn_module_destroy(tcnn_module_t* m) { delete m; }

int tcnn_module_inference(tcnn_module_t* m, tcnn_stream_t stream_, uint32_t n, const float* input, void* output, void* params) {
	TCNN_API_BEGIN
	hipStream_t stream = (hipStream_t)stream_;
	if (m->fp32_io) {
		encoding_forward_f32(stream, m->md, n, input, (const float*)params, (float*)output, nullptr, false);
		return TCNN_OK;
	}
	model_forward(stream, m->md, n, input, (half_t*)output, (const half_t*)params, nullptr, false);
	TCNN_API_END
}

int tcnn_module_forward(tcnn_module_t* m, tcnn_stream_t stream_, uint32_t n, const float* input, void* output, void* params,
                        int prepare_input_gradients, tcnn_context_t** ctx) {
	TCNN_API_BEGIN
	hipStream_t stream = (hipStream_t)stream_;
	auto c = std::make_unique<tcnn_context>();
	if (m->fp32_io) {
		encoding_forward_f32(stream, m->md, n, input, (const float*)params, (float*)output, &c->ctx, prepare_input_gradients != 0);
	} else {
		model_forward(stream, m->md, n, input, (half_t*)output, (const half_t*)params, &c->ctx, prepare_input_gradients != 0);
	}
	*ctx = c.release();
	TCNN_API_END
}

int tcnn_module_backward(tcnn_module_t* m, tcnn_stream_t stream, const tcnn_context_t* ctx, uint32_t n, float* dL_dinput, const void* dL_doutput,
                         void* dL_dparams, const float* input, const void* output, const void* params) {
	(void)output;
	TCNN_API_BEGIN
	if (!ctx) throw std::runtime_error("backward: missing forward context");
	if (m->fp32_io) {  // bare encodings only: neither `output` nor the parameters are needed by their first-order backward pass
		(void)params;
		encoding_backward_f32((hipStream_t)stream, m->md, ctx->ctx, n, dL_dinput, (const float*)dL_doutput, (float*)dL_dparams, input);
		return TCNN_OK;
	}
	model_backward((hipStream_t)stream, m->md, ctx->ctx, n, dL_dinput, (const half_t*)dL_doutput, (half_t*)dL_dparams, input, (const half_t*)output,
	               (const half_t*)params,
	               dL_dparams ? TCNN_GRADIENT_OVERWRITE : TCNN_GRADIENT_IGNORE, m->lds_level_budget);  // cpp_api.cu:115
	TCNN_API_END
}

// cpp_api.cu:117-135 -> DifferentiableObject::backward_backward_input, implemented by the grid encoding only in the
// reference (grid.h:910-1042; object.h:468 throws for everything else).
int tcnn_module_backward_backward_input(tcnn_module_t* m, tcnn_stream_t stream_, const tcnn_context_t* ctx, uint32_t n, const float* dL_ddLdinput,
                                        const float* input, const void* dL_doutput, void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput,
                                        const void* params) {
	if (m->md.has_network || !m->md.enc.is_grid) {
		g_last_error = "DifferentiableObject::backward_backward_input_impl: not implemented error";  // object.h:478
		return TCNN_ERROR_UNSUPPORTED;
	}
	TCNN_API_BEGIN
	if (!ctx) throw std::runtime_error("backward_backward_input: missing forward context");
	check_batch(n, widest_matrix(m->md));
	if (n == 0) return TCNN_OK;
	if (ctx->ctx.n != n) throw std::runtime_error("backward_backward_input: batch size does not match the forward context");
	if (!dL_ddLdinput) throw std::runtime_error("backward_backward_input: dL_ddLdinput is required");
	hipStream_t stream = (hipStream_t)stream_;
	const Model& md = m->md;
	const EncodingDesc& e = md.enc;
	// fp32 module: 16-bit stand-ins for the caller's fp32 tensors (see tcnn_module::fp32_io)
	Scratch dy16, p16, dp16, ddy16, gscale;
	void* const dL_dparams_f32 = dL_dparams;
	void* const dL_ddLdoutput_f32 = dL_ddLdoutput;
	const size_t n_out_elems = (size_t)n * e.padded_output_width;
	if (m->fp32_io) {
		if (dL_doutput) {
			dy16 = Fp32Bridge::gradient_to_half(stream, dL_doutput, n_out_elems, gscale);
			dL_doutput = dy16.ptr;
		}
		if (params) {
			p16 = Fp32Bridge::to_half(stream, params, md.n_params());
			params = p16.ptr;
		}
		if (dL_dparams) {
			dp16 = Scratch(stream, std::max<size_t>(md.n_params(), 1) * sizeof(half_t));
			dL_dparams = dp16.ptr;
		}
		if (dL_ddLdoutput) {
			ddy16 = Scratch(stream, std::max<size_t>(n_out_elems, 1) * sizeof(half_t));
			dL_ddLdoutput = ddy16.ptr;
		}
	}
	GridIO io = {input, md.n_input_dims, 1u, n, 1u, e.padded_output_width};  // the bare encoding's output is sample-major (cpp_api.cu:94-95)
	io.ddx = dL_ddLdinput;
	io.ddx_stride_i = md.n_input_dims;
	io.ddx_stride_d = 1u;
	if (dL_ddLdoutput) {  // grid.h:1012-1035
		if (!ctx->ctx.dy_dx.ptr) throw std::runtime_error("backward_backward_input: the forward pass did not prepare input gradients");
		grid_backward_backward_dLdoutput(stream, md.n_input_dims, e.n_output_dims, e.padded_output_width - e.n_output_dims, io, ctx->ctx.dy_dx.as<float>(),
		                                 (half_t*)dL_ddLdoutput);
	}
	if (dL_dparams || dL_dinput) {
		if (!dL_doutput) throw std::runtime_error("backward_backward_input: dL_doutput is required for parameter / input gradients");
	}
	if (dL_dparams && e.n_params > 0) {  // grid.h:942-975, GradientMode::Overwrite
		uint32_t budget = m->lds_level_budget ? m->lds_level_budget : g_default_lds_slice_bytes;
		GridBackwardWorkspace ws = grid_backward_workspace_size(e.grid, n, GridBackwardMode::Bucketed, budget);
		Scratch queues;
		if (ws.scratch_bytes) {
			queues = Scratch(stream, ws.scratch_bytes);
			ws.scratch = queues.ptr;
			ws.scratch_bytes = queues.bytes;
			ws.counters = ZeroedCounters::get(stream, ws.n_counters);
		}
		grid_backward(stream, e.grid, io, (const half_t*)dL_doutput, (half_t*)dL_dparams, false, GridBackwardMode::Bucketed, budget, ws);
	}
	if (dL_dinput) {  // grid.h:977-1010
		grid_backward_backward_input(stream, e.grid, io, (const half_t*)dL_doutput, (const half_t*)params, dL_dinput, md.n_input_dims, 1u);
	}
	if (m->fp32_io) {  // dL_ddLdoutput does not depend on dL_doutput; the other two carry its scale
		if (dL_ddLdoutput_f32) cast_f16_to_f32(stream, n_out_elems, ddy16.as<half_t>(), (float*)dL_ddLdoutput_f32);
		if (dL_dparams_f32) {
			if (gscale.ptr) cast_scaled_f16_to_f32(stream, md.n_params(), dp16.as<half_t>(), (float*)dL_dparams_f32, (const float*)(gscale.as<float>() + 1));
			else cast_f16_to_f32(stream, md.n_params(), dp16.as<half_t>(), (float*)dL_dparams_f32);
		}
		if (dL_dinput && gscale.ptr) scale_f32(stream, (size_t)n * md.n_input_dims, dL_dinput, (const float*)(gscale.as<float>() + 1));
	}
	TCNN_API_END
}
void tcnn_context_destroy(tcnn_context_t* ctx) { delete ctx; }

uint32_t tcnn_module_n_input_dims(const tcnn_module_t* m) { return m->md.n_input_dims; }
uint32_t tcnn_module_n_output_dims(const tcnn_module_t* m) { return m->md.padded_output_width(); }
size_t tcnn_module_n_params(const tcnn_module_t* m) { return m->md.n_params(); }
int tcnn_module_param_precision(const tcnn_module_t* m) { return m->fp32_io ? TCNN_PRECISION_FP32 : NATIVE_PRECISION; }
int tcnn_module_output_precision(const tcnn_module_t* m) { return m->fp32_io ? TCNN_PRECISION_FP32 : NATIVE_PRECISION; }

int tcnn_module_initialize_params(tcnn_module_t* m, size_t seed, float* params_full_precision, float scale) {
	TCNN_API_BEGIN
	Pcg32 rng{(uint64_t)seed};  // cpp_api.cu:139-142
	m->md.initialize_params(nullptr, rng, params_full_precision, scale);
	HIP_CHECK(hipStreamSynchronize(nullptr));
	TCNN_API_END
}

const char* tcnn_module_hyperparams_json(const tcnn_module_t* m) { return m->md.hyper_json.c_str(); }
const char* tcnn_module_name(const tcnn_module_t* m) { return m->name.c_str(); }
int tcnn_module_jit_fusion(const tcnn_module_t*) { return 0; }
int tcnn_module_set_jit_fusion(tcnn_module_t*, int val) {
	if (val) log_message(TCNN_LOG_WARNING, "JIT fusion was requested but this build has no runtime compilation path; the statically fused kernels are used.");
	return TCNN_OK;
}

int tcnn_module_grid_indices(tcnn_module_t* m, tcnn_stream_t stream, uint32_t n, const float* input, uint32_t* indices) {
	TCNN_API_BEGIN
	if (!m->md.enc.is_grid) throw std::runtime_error("grid_indices: module has no grid encoding");
	GridIO io = {input, m->md.n_input_dims, 1u, n, n, 1u};
	grid_indices((hipStream_t)stream, m->md.enc.grid, io, indices);
	TCNN_API_END
}
int tcnn_module_grid_level_n_params(const tcnn_module_t* m, uint32_t level, size_t* out) {
	TCNN_API_BEGIN
	if (!m->md.enc.is_grid || level >= m->md.enc.grid.n_levels) throw std::runtime_error("grid_level_n_params: invalid level");
	*out = m->md.enc.grid.offset[level + 1] - m->md.enc.grid.offset[level];  // multi_level_interface.h level_n_params
	TCNN_API_END
}
int tcnn_module_grid_level_params_offset(const tcnn_module_t* m, uint32_t level, size_t* out) {
	TCNN_API_BEGIN
	if (!m->md.enc.is_grid || level >= m->md.enc.grid.n_levels) throw std::runtime_error("grid_level_params_offset: invalid level");
	*out = m->md.enc.grid.offset[level];
	TCNN_API_END
}

// ------------------------------------------------------------------------------------------------ trainer

static const char* const LOSS_NAMES[N_LOSS_TYPES] = {"L2",   "RelativeL2",   "L1",       "RelativeL1",         "Mape",
                                                    "Smape", "CrossEntropy", "Variance", "RelativeL2Luminance"};  // loss.cu:57-65
static LossType string_to_loss(const std::string& s) {
	for (int i = 0; i < N_LOSS_TYPES; ++i) {
		if (equals_case_insensitive(s, LOSS_NAMES[i])) return (LossType)i;
	}
	throw std::runtime_error("Loss '" + s + "' not found");  // loss.cu:86
}

static void parse_adam(AdamHyper& h, const Json& p) {  // adam.h:221-281
	h.beta1 = p.value("beta1", h.beta1);
	h.beta2 = p.value("beta2", h.beta2);
	h.epsilon = p.value("epsilon", h.epsilon);
	h.learning_rate = p.value("learning_rate", h.learning_rate);
	h.l2_reg = p.value("l2_reg", h.l2_reg);
	h.adabound = p.value("adabound", h.adabound);
	h.relative_weight_decay = p.value("relative_decay", h.relative_weight_decay);
	h.absolute_weight_decay = p.value("absolute_decay", h.absolute_weight_decay);
	h.weight_clipping_magnitude = p.value("clipping_magnitude", h.weight_clipping_magnitude);
	h.gradient_clipping_magnitude = p.value("gradient_clipping_magnitude", h.gradient_clipping_magnitude);
	h.non_matrix_learning_rate_factor = p.value("non_matrix_learning_rate_factor", h.non_matrix_learning_rate_factor);
	h.non_matrix_l2_reg = p.value("non_matrix_l2_reg", h.non_matrix_l2_reg);
	h.optimize_matrix_params = p.value("optimize_matrix_params", h.optimize_matrix_params);
	h.optimize_non_matrix_params = p.value("optimize_non_matrix_params", h.optimize_non_matrix_params);
	h.skip_zero_grad_non_matrix_params = p.value("skip_zero_grad_non_matrix_params", h.skip_zero_grad_non_matrix_params);
}

// optimizer.cu:50-86 for the optimizers of this build: Adam, optionally inside Ema / ExponentialDecay wrappers.
// `creating`: build the chain from the config; otherwise walk the existing chain (update_hyperparams, trainer.h:380-383).
static void apply_optimizer_json(tcnn_trainable_model* tm, const Json& opts, bool creating) {
	Json cur = opts;
	size_t depth = 0;
	for (;;) {
		std::string otype = cur.value("otype", creating ? "Adam" : (depth < tm->optimizer_order.size() ? tm->optimizer_order[depth] : "Adam"));
		if (equals_case_insensitive(otype, "Ema")) {
			if (creating) {
				if (tm->ema) throw std::runtime_error("Optimizer: nested Ema inside Ema is not supported");
				tm->ema = true;
				tm->optimizer_order.push_back("Ema");
			} else if (depth >= tm->optimizer_order.size() || tm->optimizer_order[depth] != "Ema") {
				throw std::runtime_error("update_hyperparams: optimizer structure does not match the trainer's");
			}
			tm->ema_decay = cur.value("decay", tm->ema_decay);
			if (creating) tm->ema_full_precision = cur.value("full_precision", tm->ema_full_precision);
		} else if (equals_case_insensitive(otype, "ExponentialDecay")) {
			if (creating) {
				if (tm->lr_decay) throw std::runtime_error("Optimizer: nested ExponentialDecay inside ExponentialDecay is not supported");
				tm->lr_decay = true;
				tm->optimizer_order.push_back("ExponentialDecay");
			} else if (depth >= tm->optimizer_order.size() || tm->optimizer_order[depth] != "ExponentialDecay") {
				throw std::runtime_error("update_hyperparams: optimizer structure does not match the trainer's");
			}
			tm->decay_base = cur.value("decay_base", tm->decay_base);
			tm->decay_interval = cur.value("decay_interval", tm->decay_interval);
			tm->decay_start = cur.value("decay_start", tm->decay_start);
			tm->decay_end = cur.value("decay_end", tm->decay_end);
			if (tm->decay_interval == 0) throw std::runtime_error("ExponentialDecay: decay_interval must be positive");
		} else if (equals_case_insensitive(otype, "Adam")) {
			const float lr_before = tm->adam.learning_rate;
			parse_adam(tm->adam, cur);
			if (tm->lr_decay) {
				if (creating) {
					tm->base_lr = tm->adam.learning_rate;  // exponential_decay.h:52
				} else if (tm->adam.learning_rate != lr_before) {
					tm->base_lr = tm->adam.learning_rate;  // the nested optimizer's learning rate was set directly
					tm->adam.learning_rate = tm->base_lr * tm->lr_factor;
				}
			}
			return;
		} else {
			throw std::runtime_error("Optimizer '" + otype + "' is not available in this build (supported: Adam, Ema, ExponentialDecay).");
		}
		if (!creating && !cur.contains("nested")) return;
		cur = cur.value("nested", Json::object());
		++depth;
	}
}

static void refresh_hyper_json(tcnn_trainable_model* tm) {  // trainer.h:385-391, adam.h:283-302
	Json o = Json::object();
	o["otype"] = "Adam";
	o["beta1"] = tm->adam.beta1;
	o["beta2"] = tm->adam.beta2;
	o["epsilon"] = tm->adam.epsilon;
	o["learning_rate"] = tm->adam.learning_rate;
	o["l2_reg"] = tm->adam.l2_reg;
	o["adabound"] = tm->adam.adabound;
	o["relative_decay"] = tm->adam.relative_weight_decay;
	o["absolute_decay"] = tm->adam.absolute_weight_decay;
	o["clipping_magnitude"] = tm->adam.weight_clipping_magnitude;
	o["gradient_clipping_magnitude"] = tm->adam.gradient_clipping_magnitude;
	o["non_matrix_learning_rate_factor"] = tm->adam.non_matrix_learning_rate_factor;
	o["non_matrix_l2_reg"] = tm->adam.non_matrix_l2_reg;
	o["optimize_matrix_params"] = tm->adam.optimize_matrix_params;
	o["optimize_non_matrix_params"] = tm->adam.optimize_non_matrix_params;
	o["skip_zero_grad_non_matrix_params"] = tm->adam.skip_zero_grad_non_matrix_params;
	if (tm->lr_decay) o["learning_rate"] = tm->base_lr * tm->lr_factor;
	for (size_t k = tm->optimizer_order.size(); k-- > 0;) {  // wrap inside-out (ema.h:181-188, exponential_decay.h:116-125)
		Json wrapper = Json::object();
		if (tm->optimizer_order[k] == "Ema") {
			wrapper["otype"] = "EMA";
			wrapper["nested"] = o;
			wrapper["decay"] = tm->ema_decay;
			wrapper["full_precision"] = tm->ema_full_precision;
		} else {
			wrapper["otype"] = "ExponentialDecay";
			wrapper["nested"] = o;
			wrapper["decay_base"] = tm->decay_base;
			wrapper["decay_interval"] = tm->decay_interval;
			wrapper["decay_start"] = tm->decay_start;
			wrapper["decay_end"] = tm->decay_end;
		}
		o = wrapper;
	}
	Json l = Json::object();
	l["otype"] = LOSS_NAMES[(int)tm->loss];
	Json j = Json::object();
	j["otype"] = "Trainer";
	j["optimizer"] = o;
	j["loss"] = l;
	tm->hyper_json = j.dump();
}

static void cast_master_to_params(tcnn_trainable_model* tm, hipStream_t stream) {  // trainer.h:409-421
	cast_f32_to_f16(stream, tm->md.n_params(), tm->master, tm->params);
	tm->params_t_valid = false;
	HIP_CHECK(hipMemsetAsync(tm->grads, 0, tm->md.n_params() * sizeof(half_t), stream));  // reset_param_gradients, trainer.h:419
}

// transposed network weights matching `params` (the pointer the pass is about to use)
static const half_t* trainer_params_t(tcnn_trainable_model* tm, hipStream_t stream, const half_t* params, Scratch& local) {
	if (!tm->md.has_network) return nullptr;
	if (params != tm->params || !tm->params_t) {  // EMA / foreign parameters: one-off transposition
		local = Scratch(stream, tm->md.n_mlp_params() * sizeof(half_t));
		mlp_transpose_weights(stream, tm->md.net.mlp, params, local.as<half_t>());
		return local.as<half_t>();
	}
	if (!tm->params_t_valid || tm->params_exposed) {
		mlp_transpose_weights(stream, tm->md.net.mlp, tm->params, tm->params_t);
		tm->params_t_valid = true;
	}
	return tm->params_t;
}

int tcnn_create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, uint32_t seed, tcnn_trainable_model_t** out) {
	TCNN_API_BEGIN
	const Json config = Json::parse(config_json);
	auto tm = std::make_unique<tcnn_trainable_model>();
	// config.h:53-63
	const Json loss_opts = config.value("loss", Json::object());
	const Json optimizer_opts = config.value("optimizer", Json::object());
	const Json network_opts = config.value("network", Json::object());
	const Json encoding_opts = config.value("encoding", Json::object());
	const std::string loss_type = loss_opts.value("otype", "RelativeL2");  // loss.cu:81-83
	tm->loss = string_to_loss(loss_type);
	apply_optimizer_json(tm.get(), optimizer_opts, /*creating=*/true);
	tm->md = make_nwie(n_input_dims, n_output_dims, encoding_opts, network_opts);
	refresh_hyper_json(tm.get());

	// Trainer ctor + initialize_params, trainer.h:51-87
	const size_t n = tm->md.n_params();
	tm->buffer = device_malloc(n * (sizeof(float) + 2 * sizeof(half_t)));
	HIP_CHECK(hipMemset(tm->buffer, 0, n * (sizeof(float) + 2 * sizeof(half_t))));
	tm->master = (float*)tm->buffer;
	tm->params = (half_t*)((char*)tm->buffer + sizeof(float) * n);
	tm->grads = tm->params + n;
	tm->m1 = device_malloc_n<float>(n);  // adam.h:136-156
	tm->m2 = device_malloc_n<float>(n);
	tm->steps = device_malloc_n<uint32_t>(n);
	tm->step_deficits8 = device_malloc_n<uint8_t>(n);
	HIP_CHECK(hipMemset(tm->m1, 0, n * sizeof(float)));
	HIP_CHECK(hipMemset(tm->m2, 0, n * sizeof(float)));
	HIP_CHECK(hipMemset(tm->steps, 0, n * sizeof(uint32_t)));
	if (tm->ema) {  // ema.h:90-102
		tm->params_ema = device_malloc_n<half_t>(n);
		HIP_CHECK(hipMemset(tm->params_ema, 0, n * sizeof(half_t)));
		if (tm->ema_full_precision) {
			tm->ema_tmp = device_malloc_n<float>(n);
			HIP_CHECK(hipMemset(tm->ema_tmp, 0, n * sizeof(float)));
		}
	}
	if (tm->md.has_network) tm->params_t = device_malloc_n<half_t>(tm->md.n_mlp_params());
	tm->loss_scratch = device_malloc_n<float>(1032);
	std::seed_seq seq{seed};
	std::vector<uint32_t> seeds(2);
	seq.generate(std::begin(seeds), std::end(seeds));
	tm->rng = Pcg32{seeds.front()};
	tm->md.initialize_params(nullptr, tm->rng, tm->master, 1.0f);
	cast_master_to_params(tm.get(), nullptr);
	HIP_CHECK(hipDeviceSynchronize());
	*out = tm.release();
	TCNN_API_END
}

void tcnn_trainable_model_destroy(tcnn_trainable_model_t* tm) {
	if (!tm) return;
	(void)hipDeviceSynchronize();
	direct_exchange_close(tm->direct);
	if (tm->direct.own_signals) (void)hipFree(tm->direct.own_signals);
	if (tm->direct.host_error) (void)hipHostFree((void*)tm->direct.host_error);
	device_free(tm->buffer);
	device_free(tm->m1);
	device_free(tm->m2);
	device_free(tm->steps);
	device_free(tm->step_deficits8);
	device_free(tm->params_t);
	device_free(tm->params_ema);
	device_free(tm->ema_tmp);
	device_free(tm->loss_scratch);
	for (hipEvent_t e : tm->comm_events) (void)hipEventDestroy(e);
	if (tm->comm_stream) (void)hipStreamDestroy(tm->comm_stream);
	if (tm->graph_exec) (void)hipGraphExecDestroy(tm->graph_exec);
	delete tm;
}

// The stream-ordered arena of the reference (GPUMemoryArena, gpu_memory.h:405-700; allocate_workspace(stream, bytes)) as the host
// sees it: a block out of the library's stream-keyed cache -- what the library's own scratch memory comes from -- handed back to the
// cache, not to the driver, when the host is done with it.  A block released on a stream is reused by later requests ON THAT STREAM
// only (work queued there is ordered behind its previous user); *granted is what to pass back.
int tcnn_stream_malloc(tcnn_stream_t stream, size_t bytes, void** out, size_t* granted) {
	TCNN_API_BEGIN
	if (!out || !granted) throw std::runtime_error("tcnn_stream_malloc: missing output argument");
	*out = ScratchCache::acquire((hipStream_t)stream, bytes, granted);
	TCNN_API_END
}
int tcnn_stream_free(tcnn_stream_t stream, void* ptr, size_t granted) {
	TCNN_API_BEGIN
	if (ptr) ScratchCache::release(stream_key((hipStream_t)stream), ptr, granted);
	TCNN_API_END
}

// Optimizer<T> on its own (optimizer.h:40-99, optimizers/adam.h:130-219): Adam over weight buffers the HOST owns -- for callers that
// drive forward / loss / backward themselves.  The same kernel and arithmetic as the trainer's optimizer step (counter form of the
// per-parameter step counters).
struct tcnn_optimizer {
	AdamHyper adam;
	uint32_t n = 0, n_matrix = 0, step = 0;
	float *m1 = nullptr, *m2 = nullptr;
	uint32_t* steps = nullptr;
};
int tcnn_create_optimizer(const char* optimizer_json, tcnn_optimizer_t** out) {
	TCNN_API_BEGIN
	const Json opts = Json::parse(optimizer_json ? optimizer_json : "{}");
	const std::string otype = opts.value("otype", "Adam");
	if (!equals_case_insensitive(otype, "Adam")) throw std::runtime_error("Optimizer '" + otype + "' is not available as a stand-alone object in this build (supported: Adam).");
	auto o = std::make_unique<tcnn_optimizer>();
	parse_adam(o->adam, opts);
	*out = o.release();
	TCNN_API_END
}
static void optimizer_release(tcnn_optimizer_t* o) {
	device_free(o->m1);
	device_free(o->m2);
	device_free(o->steps);
	o->m1 = o->m2 = nullptr;
	o->steps = nullptr;
}
// Optimizer::allocate(n_weights, layer_sizes): the first n_matrix_weights parameters are matrix weights (weight decay / l2_reg apply to
// them, adam.h:79-110), the rest (e.g. an encoding's) are not
int tcnn_optimizer_allocate(tcnn_optimizer_t* o, size_t n_weights, size_t n_matrix_weights) {
	TCNN_API_BEGIN
	if (n_weights > 0xFFFFFFFFull || n_matrix_weights > n_weights) throw std::runtime_error("Optimizer::allocate: bad sizes");
	(void)hipDeviceSynchronize();
	optimizer_release(o);
	o->n = (uint32_t)n_weights;
	o->n_matrix = (uint32_t)n_matrix_weights;
	o->step = 0;
	if (o->n) {
		o->m1 = device_malloc_n<float>(o->n);
		o->m2 = device_malloc_n<float>(o->n);
		o->steps = device_malloc_n<uint32_t>(o->n);
		HIP_CHECK(hipMemset(o->m1, 0, (size_t)o->n * sizeof(float)));
		HIP_CHECK(hipMemset(o->m2, 0, (size_t)o->n * sizeof(float)));
		HIP_CHECK(hipMemset(o->steps, 0, (size_t)o->n * sizeof(uint32_t)));
	}
	TCNN_API_END
}
// Optimizer::step (optimizer.h:58): gradients carry the loss scale; weights_full_precision and weights (16-bit) are both updated
int tcnn_optimizer_step(tcnn_optimizer_t* o, tcnn_stream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) {
	TCNN_API_BEGIN
	if (!o->n) return TCNN_OK;
	if (!weights_full_precision || !weights || !gradients) throw std::runtime_error("Optimizer::step: missing buffer");
	++o->step;  // adam.h:159
	adam_step((hipStream_t)stream, o->adam, o->n, o->n_matrix, loss_scale, o->step, weights_full_precision, (half_t*)weights, (half_t*)gradients /* read only: no finalize rides along */, o->m1, o->m2, o->steps);
	TCNN_API_END
}
uint32_t tcnn_optimizer_step_count(const tcnn_optimizer_t* o) { return o->step; }
int tcnn_optimizer_update_hyperparams(tcnn_optimizer_t* o, const char* optimizer_json) {
	TCNN_API_BEGIN
	parse_adam(o->adam, Json::parse(optimizer_json ? optimizer_json : "{}"));
	TCNN_API_END
}
// which = 0 first moments, 1 second moments (fp32), 2 per-parameter step counters (u32); device pointers, n_weights elements each
void* tcnn_optimizer_state(tcnn_optimizer_t* o, int which) { return which == 0 ? (void*)o->m1 : which == 1 ? (void*)o->m2 : which == 2 ? (void*)o->steps : nullptr; }
void tcnn_optimizer_destroy(tcnn_optimizer_t* o) {
	if (!o) return;
	(void)hipDeviceSynchronize();
	optimizer_release(o);
	delete o;
}

// Loss<T>::evaluate (loss.h:42-50) on its own: prediction / gradients are column-major `stride` x n matrices in the library's 16-bit
// type (= sample-major [n][stride]), target / data_pdf `dims` x n fp32, values `stride` x n fp32 (may be null); rows >= dims carry no
// loss (relative_l2.h:57-61).  Normalised by n * dims like the reference's kernels (n_elements / stride * dims).
int tcnn_loss_evaluate(const char* loss_otype, tcnn_stream_t stream, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale, const void* prediction,
                       const float* target, const float* data_pdf, float* values, void* gradients) {
	TCNN_API_BEGIN
	if (!loss_otype || !prediction || !target || !gradients) throw std::runtime_error("Loss::evaluate: missing argument");
	if (stride % 8 != 0 || dims > stride) throw std::runtime_error("Loss::evaluate: the prediction's row count must be a multiple of 8 and at least the target's");
	if ((uint64_t)n * dims > 0xFFFFFFFFull || (uint64_t)n * stride > 0xFFFFFFFFull) throw std::runtime_error("Loss::evaluate: batch too large");
	if (n == 0) return TCNN_OK;
	loss_evaluate((hipStream_t)stream, string_to_loss(loss_otype), n, stride, dims, loss_scale, (const half_t*)prediction, target, data_pdf, values, (half_t*)gradients,
	              nullptr, n * dims);
	TCNN_API_END
}

int tcnn_trainer_forward(tcnn_trainable_model_t* tm, tcnn_stream_t stream_, float loss_scale, uint32_t n, const float* input, const float* target,
                         const float* data_pdf, int use_inference_params, int prepare_input_gradients, const void* external_dL_dy,
                         tcnn_train_context_t** ctx_out) {
	TCNN_API_BEGIN
	ProfilerGuard pg(tm->profiler.get());
	hipStream_t stream = (hipStream_t)stream_;
	const half_t* params = use_inference_params ? tm->inference_params() : tm->params;  // trainer.h:497-500 (EMA weights when Ema is on)
	auto c = std::make_unique<tcnn_train_context>();
	c->n = n;
	c->stream = stream;
	const uint32_t padded = tm->md.padded_output_width();
	c->output = Scratch(stream, (size_t)padded * n * sizeof(half_t));
	model_forward(stream, tm->md, n, input, c->output.as<half_t>(), params, &c->model_ctx, prepare_input_gradients != 0);
	if (external_dL_dy) {  // trainer.h:124-128
		c->dL_doutput_ptr = (const half_t*)external_dL_dy;
	} else {
		if (!target) throw std::runtime_error("Trainer::forward: target must be given unless external_dL_dy is");
		c->dL_doutput = Scratch(stream, (size_t)padded * n * sizeof(half_t));
		c->dL_doutput_ptr = c->dL_doutput.as<half_t>();
		c->n_block_sums = loss_n_blocks(n, padded);
		c->block_sums = Scratch(stream, (size_t)c->n_block_sums * sizeof(float));
		const uint64_t n_total = (tm->global_batch ? tm->global_batch : (uint64_t)n) * tm->md.output_width();
		if (n_total > 0xFFFFFFFFull) throw std::runtime_error("Trainer::forward: batch too large");
		ProfScope prof(stream, STAGE_LOSS);
		loss_evaluate(stream, tm->loss, n, padded, tm->md.output_width(), loss_scale, c->output.as<half_t>(), target, data_pdf, nullptr,
		              c->dL_doutput.as<half_t>(), c->block_sums.as<float>(), (uint32_t)n_total);
	}
	*ctx_out = c.release();
	TCNN_API_END
}

int tcnn_trainer_backward(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_train_context_t* ctx, uint32_t n, const float* input,
                          float* dL_dinput, int use_inference_params, int gradient_mode) {
	TCNN_API_BEGIN
	tm->last_batch = n;
	if (!ctx) throw std::runtime_error("Trainer::backward: missing forward context");
	ProfilerGuard pg(tm->profiler.get());
	model_backward((hipStream_t)stream, tm->md, ctx->model_ctx, n, dL_dinput, ctx->dL_doutput_ptr, tm->grads, input, ctx->output.as<half_t>(),
	               use_inference_params ? tm->inference_params() : tm->params, gradient_mode,
	               tm->lds_level_budget);
	TCNN_API_END
}

// Per-parameter step counters (adam.h:84) as counters or as deficits?  A stepped parameter costs 8 B of counter traffic in
// the first form and 4 B in the second, a skipped (zero-gradient) hash-table entry 0 B and 8 B.  With N samples touching
// 2^D corners per level, an entry of a level with T entries is skipped with probability exp(-N 2^D / T): deficits pay off
// below ~1/3, i.e. for N 2^D >= T at the largest level (the headline: 4 T).  TCNN_ADAM_STEP_DEFICITS=0/1 forces a form.
// The deficits are kept as BYTES (255 = the parameter's counter itself lives in the 32-bit array): one byte of bookkeeping per stepped
// parameter instead of four; the 32-bit deficit form remains for the optimizer step fused into the grid backward and for
// TCNN_ADAM_STEP_DEFICITS=1 (=0: counters, =2: bytes).
static int choose_step_representation(const tcnn_trainable_model* tm) {
	const int deficits = ADAM_STEPS_DEFICITS8;
	if (!tm->md.enc.is_grid) return deficits;  // network weights are stepped every time
	const auto& g = tm->md.enc.grid;
	uint32_t largest = 0;
	for (uint32_t l = 0; l < g.n_levels; ++l) largest = std::max(largest, g.offset[l + 1] - g.offset[l]);
	const uint64_t batch = tm->global_batch ? tm->global_batch : tm->last_batch;  // the reduced gradient covers the global batch
	return (batch << g.n_dims) >= (uint64_t)largest ? deficits : ADAM_STEPS_COUNTERS;
}
// the per-parameter step counters as counters (what snapshots and hosts see); `steps_done` = optimizer steps completed
static void step_counters_to_counter_form(tcnn_trainable_model* tm, hipStream_t stream, uint32_t steps_done) {
	adam_convert_step_representation(stream, (uint32_t)tm->md.n_params(), steps_done, tm->steps, tm->step_deficits8, tm->steps_form, ADAM_STEPS_COUNTERS);
	tm->steps_form = ADAM_STEPS_COUNTERS;
}

// Optimizer::step over a set of parameter ranges [begin, end) (begins multiples of 8).  `advance`: this call opens a new
// optimizer step (step counter, learning-rate schedule, step-counter representation); the other calls of the same step
// (a data-parallel host steps each gradient bucket as soon as it is reduced) continue it.
// opens a new optimizer step: step counter, learning-rate schedule, representation of the per-parameter step counters
static void optimizer_advance(tcnn_trainable_model_t* tm, hipStream_t stream) {
	if (tm->lr_decay) {  // exponential_decay.h:59-70, with step() == the nested optimizer's step count before this step
		const uint32_t step = tm->optimizer_step;
		if (step == 0) tm->lr_factor = 1.0f;
		if (step >= tm->decay_start && (step - tm->decay_start) % tm->decay_interval == 0 && step <= tm->decay_end) tm->lr_factor *= tm->decay_base;
		tm->adam.learning_rate = tm->base_lr * tm->lr_factor;
	}
	++tm->optimizer_step;  // adam.h:159
	const int want = choose_step_representation(tm);
	if (want != tm->steps_form) {
		adam_convert_step_representation(stream, (uint32_t)tm->md.n_params(), tm->optimizer_step - 1u, tm->steps, tm->step_deficits8, tm->steps_form, want);
		tm->steps_form = want;
	}
}

// all-reduces this step started on the communication stream (tcnn_trainer_enable_rccl): `stream` continues behind them
static void await_reduced_gradients(tcnn_trainable_model_t* tm, hipStream_t stream) {
	for (const auto& r : tm->reduced) HIP_CHECK(hipStreamWaitEvent(stream, r.done, 0));
	tm->reduced.clear();
}

// Adam over [begin, end) of the current optimizer step (optimizer_advance opened it), on `stream`
static void adam_range(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale, size_t begin, size_t end, bool counts) {
	// the trainer's 16-bit parameters are its rounded master weights unless a caller holds a pointer to them (params_exposed): Adam need
	// not read the skipped ones back (AdamCore::half_follows_master)
	const size_t n = tm->md.n_params();
	ProfScope prof(stream, STAGE_ADAM, counts);  // a ranged (bucketed) step is ONE optimizer step
	adam_step(stream, tm->adam, (uint32_t)n, (uint32_t)tm->md.n_mlp_params(), loss_scale, tm->optimizer_step, tm->master, tm->params, tm->grads,
	          tm->m1, tm->m2, tm->steps, tm->params_t_valid ? tm->params_t : nullptr, tm->md.has_network ? &tm->md.net.mlp : nullptr, (uint32_t)begin,
	          (uint32_t)end, tm->steps_form, tm->step_deficits8, /*half_follows_master=*/!tm->params_exposed,
	          begin == 0 && tm->pending_finalize.partials ? &tm->pending_finalize : nullptr);
	if (begin == 0) tm->pending_finalize = AdamFinalize();
	if (tm->ema) ema_step(stream, (uint32_t)n, tm->ema_decay, tm->optimizer_step, tm->params, tm->params_ema, tm->ema_tmp, (uint32_t)begin, (uint32_t)end);
}

static void optimizer_step_ranges(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale, size_t n_ranges, const size_t* begins, const size_t* ends,
                                  bool advance, bool opens_profiled_step) {
	const size_t n = tm->md.n_params();
	await_reduced_gradients(tm, stream);
	for (size_t r = 0; r < n_ranges; ++r) {
		if (begins[r] % 8 != 0 || begins[r] > std::min(ends[r], n)) throw std::runtime_error("optimizer_step_range: a range must start at a multiple of 8 and not end before it");
	}
	if (advance) optimizer_advance(tm, stream);
	ProfilerGuard pg(tm->profiler.get());
	for (size_t r = 0; r < n_ranges; ++r) {
		const size_t begin = begins[r], end = std::min(ends[r], n);
		if (begin == end) continue;
		adam_range(tm, stream, loss_scale, begin, end, /*counts=*/opens_profiled_step && r == 0);
	}
}

// One optimizer step == calls whose ranges tile [0, n_params) exactly once, the range with begin == 0 first.
int tcnn_trainer_optimizer_step_range(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, size_t begin, size_t end) {
	TCNN_API_BEGIN
	optimizer_step_ranges(tm, (hipStream_t)stream, loss_scale, 1, &begin, &end, /*advance=*/begin == 0, begin == 0);
	TCNN_API_END
}

// One optimizer step over the union of the given ranges only (a rank that owns a shard of the parameters, ZeRO-1 style:
// the other parameters' optimizer state is left untouched on this rank).
int tcnn_trainer_optimizer_step_ranges(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, size_t n_ranges, const size_t* begins,
                                       const size_t* ends) {
	TCNN_API_BEGIN
	optimizer_step_ranges(tm, (hipStream_t)stream, loss_scale, n_ranges, begins, ends, /*advance=*/true, true);
	TCNN_API_END
}

int tcnn_trainer_optimizer_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale) {
	return tcnn_trainer_optimizer_step_range(tm, stream, loss_scale, 0, tm->md.n_params());
}

// Gradient exchange hook of a data-parallel C/C++ host: called by training_step(run_optimizer = true) between backward and
// the optimizer with the fp16 gradient buffer [network | encoding] and the stream the step runs on; the callback issues
// e.g. ncclAllReduce(grads, grads, n, ncclHalf, ncclSum, comm, stream).  The library itself links no collective library.
int tcnn_trainer_set_gradient_exchange(tcnn_trainable_model_t* tm, void (*exchange)(void* user, void* gradients_fp16, size_t n_params, tcnn_stream_t stream),
                                       void* user) {
	tm->exchange = exchange;
	tm->exchange_user = user;
	return TCNN_OK;
}

// Hosts that overlap the exchange with the backward pass.  `ready(user, begin, end, stream)` is called on the host, during
// training_step, as soon as the kernels that produce the gradients [begin, end) have been enqueued on `stream`: first the network's
// weights [0, n_network_params), then the encoding's levels in `n_groups` groups of consecutive levels (equal parameter counts;
// tcnn_trainer_set_backward_level_groups).  The ranges of one step tile [0, n_params) in ascending order, begins are multiples of 8.
int tcnn_trainer_set_gradient_ready_callback(tcnn_trainable_model_t* tm, void (*ready)(void* user, size_t begin, size_t end, tcnn_stream_t stream), void* user) {
	tm->gradients_ready = ready;
	tm->ready_user = user;
	return TCNN_OK;
}
int tcnn_trainer_set_backward_level_groups(tcnn_trainable_model_t* tm, uint32_t n_groups) {
	tm->backward_level_groups = n_groups ? n_groups : 1u;
	return TCNN_OK;
}
// Data parallelism without a callback: `nccl_comm` is the host's ncclComm_t for this rank (NULL switches it off again).  From then on
// training_step all-reduces (sum) every gradient range on an internal communication stream as soon as it is ready -- RCCL is loaded
// with dlopen at this point, the library does not link it -- and, with run_optimizer, steps each range when its own collective has
// finished.  The host sets the global batch size (tcnn_trainer_set_global_batch_size) so that the sum is the global gradient.
static void enable_rccl(tcnn_trainable_model_t* tm, void* nccl_comm, int n_ranks, int rank) {
	if (nccl_comm) {
		const Rccl& r = Rccl::get();
		if (!r.handle || !r.all_reduce) throw std::runtime_error("tcnn_trainer_enable_rccl: librccl.so could not be loaded");
		if (rank >= 0 && (!r.reduce_scatter || !r.all_gather)) throw std::runtime_error("tcnn_trainer_enable_rccl_sharded: librccl.so lacks ncclReduceScatter / ncclAllGather");
		if (n_ranks < 1 || rank >= n_ranks) throw std::runtime_error("tcnn_trainer_enable_rccl: rank / n_ranks out of range");
		if (!tm->comm_stream) HIP_CHECK(hipStreamCreateWithFlags(&tm->comm_stream, hipStreamNonBlocking));
	}
	tm->rccl_comm = nccl_comm;
	tm->rccl_ranks = n_ranks;
	tm->rccl_rank = nccl_comm ? rank : -1;
	tm->reduced.clear();
}
int tcnn_trainer_enable_rccl(tcnn_trainable_model_t* tm, void* nccl_comm, int n_ranks) {
	TCNN_API_BEGIN
	enable_rccl(tm, nccl_comm, n_ranks, -1);
	TCNN_API_END
}
// The sharded exchange inside the library (what tinycudann/parallel.py's "pipelined_sharded" does from Python): every ready gradient
// range is reduce-scattered; training_step(run_optimizer) then runs Adam on this rank's shard of every range only -- the optimizer, the
// largest HBM consumer of a step, shrinks by the number of ranks; fp32 master weights and Adam's moments of the other shards are never
// touched on this rank -- and all-gathers the 16-bit parameters (and the EMA weights of an Ema optimizer).  Same bytes on the wire as the
// all-reduce scheme; replicas cannot drift (everyone receives the same 16-bit parameters).  `rank`: this process's rank in `nccl_comm`.
int tcnn_trainer_enable_rccl_sharded(tcnn_trainable_model_t* tm, void* nccl_comm, int n_ranks, int rank) {
	TCNN_API_BEGIN
	if (nccl_comm && rank < 0) throw std::runtime_error("tcnn_trainer_enable_rccl_sharded: rank must be >= 0");
	enable_rccl(tm, nccl_comm, n_ranks, rank);
	TCNN_API_END
}

static void direct_exchange_and_step(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale);
// ---- gradient exchange over peer-mapped memory (direct_exchange.h): every rank publishes IPC handles of its trainer buffer and of a small
// signal block, maps its peers', and from then on a step's exchange is: read the peers' shards of the own 1/P of the gradient buffer over all
// links at once, sum in fp32 in rank order, one rounding -> Adam on that shard -> write the stepped parameters into every peer's buffer.
int tcnn_trainer_direct_export(tcnn_trainable_model_t* tm, void* out, size_t capacity, size_t* n_bytes) {
	TCNN_API_BEGIN
	if (n_bytes) *n_bytes = sizeof(DirectExport);
	if (!out) return TCNN_OK;
	if (capacity < sizeof(DirectExport)) throw std::runtime_error("tcnn_trainer_direct_export: buffer too small");
	if (debug_alloc_mode() != DebugAlloc::Off) throw std::runtime_error("tcnn_trainer_direct_export: not available under TCNN_DEBUG_ALLOC (the trainer buffer must be a plain hipMalloc block)");
	if (tm->ema) throw std::runtime_error("tcnn_trainer_direct_export: Ema-wrapped optimizers are not supported by the direct exchange (use the sharded collective scheme)");
	HIP_CHECK(hipDeviceSynchronize());
	direct_exchange_export(tm->direct, tm->buffer, tm->params, tm->grads, tm->md.n_params(), *(DirectExport*)out);
	TCNN_API_END
}
int tcnn_trainer_direct_open(tcnn_trainable_model_t* tm, int rank, int n_ranks, const void* exports, size_t bytes_each) {
	TCNN_API_BEGIN
	if (bytes_each != sizeof(DirectExport) || !exports) throw std::runtime_error("tcnn_trainer_direct_open: exports must be n_ranks records of tcnn_trainer_direct_export's size");
	HIP_CHECK(hipDeviceSynchronize());
	direct_exchange_open(tm->direct, rank, n_ranks, (const DirectExport*)exports, tm->params, tm->grads);
	tm->params_exposed = true;  // peers write this rank's 16-bit parameters from now on: Adam must not re-derive skipped ones from its master weights
	TCNN_API_END
}
int tcnn_trainer_direct_close(tcnn_trainable_model_t* tm) {
	TCNN_API_BEGIN
	HIP_CHECK(hipDeviceSynchronize());
	direct_exchange_close(tm->direct);
	TCNN_API_END
}
// after training_step(run_optimizer = 0): the exchange + optimizer half of the step (training_step(run_optimizer = 1) does the same itself)
int tcnn_trainer_direct_exchange_and_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale) {
	TCNN_API_BEGIN
	if (!tm->direct.active()) throw std::runtime_error("tcnn_trainer_direct_exchange_and_step: tcnn_trainer_direct_open first");
	direct_exchange_and_step(tm, (hipStream_t)stream, loss_scale);
	TCNN_API_END
}
// 0: every wait of the exchange found its peers in time; 1 / 2: a wait for the peers' gradients / parameters timed out (synchronises)
int tcnn_trainer_direct_status(tcnn_trainable_model_t* tm, tcnn_stream_t stream, int* status) {
	TCNN_API_BEGIN
	*status = direct_exchange_status((hipStream_t)stream, tm->direct);
	TCNN_API_END
}

// Link check of an opened exchange (collective: same rounds and seed on every rank, between steps; overwrites the gradient buffer only).
// *mismatches: elements of this rank's buffer that did not hold the expected sum; *status as tcnn_trainer_direct_status.
int tcnn_trainer_direct_selftest(tcnn_trainable_model_t* tm, tcnn_stream_t stream, uint32_t rounds, uint32_t seed, uint64_t* mismatches, int* status) {
	TCNN_API_BEGIN
	if (!tm->direct.active()) throw std::runtime_error("tcnn_trainer_direct_selftest: tcnn_trainer_direct_open first");
	direct_exchange_selftest((hipStream_t)stream, tm->direct, rounds, seed, mismatches, status);
	TCNN_API_END
}

// Adam's state for snapshots / sharded data parallelism: which = 0 first moments (fp32), 1 second moments (fp32),
// 2 per-parameter step counters (u32; *steps_are_deficits tells their representation, see tcnn_trainer_optimizer_step_range).
void* tcnn_trainer_optimizer_state(tcnn_trainable_model_t* tm, int which, int* steps_are_deficits) {
	if (which == 2 && tm->steps_form == ADAM_STEPS_DEFICITS8) {  // the byte form is the library's own business: hosts see counters
		(void)hipDeviceSynchronize();
		try {
			step_counters_to_counter_form(tm, nullptr, tm->optimizer_step);
		} catch (const std::exception& ex) {
			g_last_error = ex.what();
			return nullptr;
		}
		(void)hipDeviceSynchronize();
	}
	if (steps_are_deficits) *steps_are_deficits = tm->steps_form == ADAM_STEPS_DEFICITS32 ? 1 : 0;
	return which == 0 ? (void*)tm->m1 : which == 1 ? (void*)tm->m2 : which == 2 ? (void*)tm->steps : nullptr;
}

// The optimizer half of training_step.  With RCCL enabled every range whose all-reduce was started during the backward pass is
// stepped as soon as ITS collective has finished (the later ones are still on the wire); otherwise the host's exchange hook, then
// one optimizer step.
// reduce over the peers' mapped gradient buffers -> Adam on this rank's shard (the last rank's carries the remainder) -> push the stepped parameters
static void direct_exchange_and_step(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale) {
	DirectExchange& dx = tm->direct;
	direct_exchange_begin_step(dx);
	{
		ProfilerGuard pg(tm->profiler.get());
		{
			ProfScope prof(stream, STAGE_DX_WAIT_GRADS);
			direct_exchange_signal_wait(stream, dx, 0);
		}
		ProfScope prof(stream, STAGE_DX_REDUCE);
		direct_exchange_reduce_own(stream, dx);
	}
	std::vector<size_t> begins, ends;
	if (dx.own_count()) {
		begins.push_back(dx.own_begin);
		ends.push_back(dx.own_end);
	}
	{
		struct EveryStage {  // the exchange's Adam is one of its phases: timed whatever the profiler's stage filter says
			EveryStage() { g_prof_every_stage = true; }
			~EveryStage() { g_prof_every_stage = false; }
		} every_stage;
		optimizer_step_ranges(tm, stream, loss_scale, begins.size(), begins.data(), ends.data(), /*advance=*/true, /*opens_profiled_step=*/true);
	}
	{
		ProfilerGuard pg(tm->profiler.get());
		{
			ProfScope prof(stream, STAGE_DX_PUSH);
			direct_exchange_push_own(stream, dx);
		}
		ProfScope prof(stream, STAGE_DX_WAIT_PARAMS);
		direct_exchange_signal_wait(stream, dx, 1);
	}
	direct_exchange_finish_step(stream, dx);
	tm->params_t_valid = false;  // the transposed network weights were maintained for this rank's shard only
}

static int finish_training_step(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale) {
	if (tm->direct.active()) {
		TCNN_API_BEGIN
		direct_exchange_and_step(tm, stream, loss_scale);
		return TCNN_OK;
		TCNN_API_END
	}
	if (tm->rccl_comm && !tm->reduced.empty() && tm->rccl_rank >= 0) {
		TCNN_API_BEGIN
		const std::vector<tcnn_trainable_model::ReducedRange> ranges = std::move(tm->reduced);
		tm->reduced.clear();
		if (ranges.front().begin != 0) throw std::runtime_error("training_step: the reduced gradient ranges do not start at parameter 0");
		rccl_poll_async_error(tm);
		const Rccl& r = Rccl::get();
		const size_t P = (size_t)tm->rccl_ranks, me = (size_t)tm->rccl_rank;
		// ONE optimizer step over this rank's shards of all ranges (+ the remainders everyone steps), behind all reduce-scatters
		std::vector<size_t> begins, ends;
		for (const auto& rr : ranges) {
			HIP_CHECK(hipStreamWaitEvent(stream, rr.done, 0));
			if (rr.shard) {
				begins.push_back(rr.begin + me * rr.shard);
				ends.push_back(rr.begin + (me + 1) * rr.shard);
			}
			if (rr.begin + rr.shard * P < rr.end) {
				begins.push_back(rr.begin + rr.shard * P);
				ends.push_back(rr.end);
			}
		}
		optimizer_step_ranges(tm, stream, loss_scale, begins.size(), begins.data(), ends.data(), /*advance=*/true, /*opens_profiled_step=*/true);
		// all-gather of the stepped 16-bit parameters, in place (send = recv + rank * shard), on the communication stream behind the optimizer
		hipEvent_t stepped = tm->comm_event(), gathered = tm->comm_event();
		HIP_CHECK(hipEventRecord(stepped, stream));
		HIP_CHECK(hipStreamWaitEvent(tm->comm_stream, stepped, 0));
		const int type = HALF_IS_BF16 ? RCCL_BFLOAT16 : RCCL_HALF;
		if (r.group_start) rccl_check(r, r.group_start(), "ncclGroupStart");
		for (half_t* buf : {tm->params, tm->ema ? tm->params_ema : (half_t*)nullptr}) {
			if (!buf) continue;
			for (const auto& rr : ranges) {
				if (rr.shard) rccl_check(r, r.all_gather(buf + rr.begin + me * rr.shard, buf + rr.begin, rr.shard, type, tm->rccl_comm, tm->comm_stream), "ncclAllGather");
			}
		}
		if (r.group_end) rccl_check(r, r.group_end(), "ncclGroupEnd");
		HIP_CHECK(hipEventRecord(gathered, tm->comm_stream));
		HIP_CHECK(hipStreamWaitEvent(stream, gathered, 0));  // whatever reads the parameters next on the compute stream sees everybody's shards
		tm->params_t_valid = false;  // the transposed network weights were maintained for this rank's shard only
		return TCNN_OK;
		TCNN_API_END
	}
	if (tm->rccl_comm && !tm->reduced.empty()) {
		TCNN_API_BEGIN
		const std::vector<tcnn_trainable_model::ReducedRange> ranges = std::move(tm->reduced);
		tm->reduced.clear();
		if (ranges.front().begin != 0) throw std::runtime_error("training_step: the reduced gradient ranges do not start at parameter 0");
		rccl_poll_async_error(tm);
		for (const auto& r : ranges) {
			HIP_CHECK(hipStreamWaitEvent(stream, r.done, 0));
			optimizer_step_ranges(tm, stream, loss_scale, 1, &r.begin, &r.end, /*advance=*/r.begin == 0, r.begin == 0);
		}
		TCNN_API_END
	}
	if (tm->exchange) tm->exchange(tm->exchange_user, tm->grads, tm->md.n_params(), stream);
	return tcnn_trainer_optimizer_step(tm, stream, loss_scale);
}
// A step that leaves the optimizer to a later call: the compute stream goes behind the all-reduces it started, so that whatever reads the
// gradient buffer next on that stream (the host's own optimizer_step, a copy of param_gradients) sees the reduced values.
static int settle_unstepped_reductions(tcnn_trainable_model_t* tm, hipStream_t stream, bool optimizer_ran) {
	if (optimizer_ran || !tm->rccl_comm) return TCNN_OK;
	TCNN_API_BEGIN
	await_reduced_gradients(tm, stream);
	TCNN_API_END
}
struct ReadyTrampoline {
	tcnn_trainable_model_t* tm;
	hipStream_t stream;
	static void call(void* self, size_t begin, size_t end) {
		auto* t = (ReadyTrampoline*)self;
		notify_gradients_ready(t->tm, t->stream, begin, end);
	}
};
static bool wants_ready_ranges(const tcnn_trainable_model_t* tm) { return tm->gradients_ready || tm->rccl_comm; }

// training_step fast path (g_fused_network_passes): encoding forward, ONE kernel for the network's forward + loss + backward, encoding
// backward.  Same results as forward() + backward() (tests/test_emu_kernels.py); the returned context carries the
// prediction, dL_doutput, the loss and the encoded input, but no hidden activations.
static int training_step_fused(tcnn_trainable_model_t* tm, hipStream_t stream, float loss_scale, uint32_t n, const float* input, const float* target,
                               const float* data_pdf, float* dL_dinput, int use_inference_params, int gradient_mode, bool run_optimizer,
                               const half_t* external_dL_dy, tcnn_train_context_t** ctx_out, bool context_wanted) {
	TCNN_API_BEGIN
	const half_t* params = use_inference_params ? tm->inference_params() : tm->params;
	ProfilerGuard pg(tm->profiler.get());
	const Model& md = tm->md;
	check_batch(n, widest_matrix(md));
	auto c = std::make_unique<tcnn_train_context>();
	c->n = n;
	c->stream = stream;
	ForwardCtx& fc = c->model_ctx;
	fc.stream = stream;
	fc.n = n;
	const uint32_t padded = md.padded_output_width();
	const bool want_grads = gradient_mode != TCNN_GRADIENT_IGNORE;
	const bool accumulate = gradient_mode == TCNN_GRADIENT_ACCUMULATE;
	const EncodingDesc& e = md.enc;
	c->output = Scratch(stream, (size_t)padded * n * sizeof(half_t));
	if (external_dL_dy) {  // trainer.h:124-128: no loss is evaluated, the backward half continues from the caller's gradient
		c->dL_doutput_ptr = external_dL_dy;
	} else {
		c->dL_doutput = Scratch(stream, (size_t)padded * n * sizeof(half_t));
		c->dL_doutput_ptr = c->dL_doutput.as<half_t>();
	}
	const uint64_t n_total = (tm->global_batch ? tm->global_batch : (uint64_t)n) * md.output_width();
	if (n_total > 0xFFFFFFFFull) throw std::runtime_error("Trainer::forward: batch too large");
	if (n == 0) {
		*ctx_out = c.release();
		if (run_optimizer) return tcnn_trainer_optimizer_step(tm, stream, loss_scale);
		return TCNN_OK;
	}

	float* dy_dx = nullptr;
	if (dL_dinput && e.is_grid) {
		fc.dy_dx = Scratch(stream, (size_t)e.n_output_dims * n * md.n_input_dims * sizeof(float));
		dy_dx = fc.dy_dx.as<float>();
	}
	fc.enc = Scratch(stream, (size_t)e.padded_output_width * n * sizeof(half_t));
	// An Identity encoding that pads nothing is `(T)(x * scale + offset)` per element: the register-resident network kernel reads the caller's
	// fp32 matrix itself (MlpF32Input) and leaves the encoded matrix behind for the context -- no transpose kernel, no second pass over the input
	const bool plain_identity = !e.is_grid && !e.is_frequency && !e.is_oneblob && e.n_dims == e.padded_output_width && in_stride_i(md) == e.n_dims && in_stride_d() == 1u;
	const bool input_by_network = plain_identity && !external_dL_dy && g_fused_identity_input.load() != 0 && n <= (1u << 25) &&
	                              mlp_train_f32_input_supported(md.net.mlp, n, tm->loss);
	if (!input_by_network) encoding_forward(stream, md, n, input, params + md.n_mlp_params(), fc.enc.as<half_t>(), /*soa=*/true, dy_dx);

	const bool need_denc = (want_grads && e.n_params > 0) || dL_dinput;
	Scratch denc;
	Scratch partials;
	{
		ProfScope prof(stream, STAGE_MLP_TRAIN);
		Scratch params_t_local;
		const half_t* params_t = trainer_params_t(tm, stream, params, params_t_local);
		const uint32_t n_partials = mlp_train_n_partials(md.net.mlp, n, external_dL_dy ? LossType::L2 : tm->loss);
		if (want_grads) partials = Scratch(stream, (size_t)n_partials * md.n_mlp_params() * sizeof(float));
		if (need_denc) denc = Scratch(stream, (size_t)e.padded_output_width * n * sizeof(half_t));
		MlpLossArgs la = {tm->loss, target, data_pdf, md.output_width(), loss_scale, (uint32_t)n_total};
		la.external_dL_doutput = external_dL_dy;
		if (!external_dL_dy) {  // Trainer::loss(ctx) has something to reduce
			c->n_block_sums = n_partials;
			c->block_sums = Scratch(stream, (size_t)n_partials * sizeof(float));
		}
		MlpF32Input f32_input;
		f32_input.x = input;
		f32_input.scale = e.id_scale;
		f32_input.offset = e.id_offset;
		f32_input.enc_out = context_wanted ? fc.enc.as<half_t>() : nullptr;  // (a context that is handed out holds the encoded input, whoever computed it)
		const SlabOrder order = mlp_train(stream, md.net.mlp, n, params, params_t, fc.enc.as<half_t>(), la, c->output.as<half_t>(),
		                                  external_dL_dy ? nullptr : c->dL_doutput.as<half_t>(), need_denc ? denc.as<half_t>() : nullptr,
		                                  want_grads ? partials.as<float>() : nullptr, external_dL_dy ? nullptr : c->block_sums.as<float>(),
		                                  input_by_network ? &f32_input : nullptr);
		// The slabs' sums: by the first workgroups of the optimizer's launch when this call runs the whole step itself on one GPU (nobody reads
		// the network's gradients between here and there: no exchange, no ready callback, no accumulation) -- by their own kernel otherwise.
		const bool sum_in_optimizer = want_grads && run_optimizer && !accumulate && !tm->exchange && !tm->direct.active() && !wants_ready_ranges(tm) &&
		                              md.n_mlp_params() % 4 == 0 && g_finalize_in_optimizer.load() != 0;
		if (sum_in_optimizer) {
			tm->pending_finalize.partials = partials.as<float>();
			tm->pending_finalize.n_partials = n_partials;
			tm->pending_finalize.order = (uint32_t)order;
		} else if (want_grads) {
			mlp_finalize_gradients(stream, md.net.mlp, n_partials, partials.as<float>(), tm->grads, accumulate, order);
		}
	}
	struct DropPendingFinalize {  // (an exception between here and the optimizer must not leave a dangling slab pointer behind)
		tcnn_trainable_model_t* tm;
		~DropPendingFinalize() { tm->pending_finalize = AdamFinalize(); }
	} drop_pending_finalize = {tm};
	ReadyTrampoline tramp = {tm, stream};
	LevelGroups level_groups = {tm->backward_level_groups, want_grads && wants_ready_ranges(tm) ? &ReadyTrampoline::call : nullptr, &tramp};
	if (level_groups.ready) level_groups.ready(&tramp, 0, md.n_mlp_params());  // the network's gradients: the first range of the step
	if (need_denc) {
		encoding_backward(stream, md, fc, n, dL_dinput, denc.as<half_t>(), n, 1u, tm->grads, want_grads, accumulate, input, tm->lds_level_budget, &level_groups);
	}
	*ctx_out = c.release();
	if (run_optimizer) return finish_training_step(tm, stream, loss_scale);
	TCNN_API_END
}

int tcnn_trainer_training_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, uint32_t n, const float* input, const float* target,
                               const float* data_pdf, int run_optimizer, float* dL_dinput, int use_inference_params, int gradient_mode,
                               const void* external_dL_dy, tcnn_train_context_t** ctx_out) {
	const float loss_scale = LOSS_SCALE_FP16;  // trainer.h:265
	tm->last_batch = n;
	// all-reduces of an earlier step(run_optimizer = false) may still be on the communication stream: this step's backward pass writes
	// the same gradient buffer and re-records the same events, so the compute stream goes behind them first
	try {
		await_reduced_gradients(tm, (hipStream_t)stream);
	} catch (const std::exception& ex) {
		g_last_error = ex.what();
		return TCNN_ERROR;
	}
	tm->comm_events_used = 0;
	if (tm->rccl_comm && tm->rccl_rank >= 0 && !run_optimizer && gradient_mode != TCNN_GRADIENT_IGNORE) {
		// after the reduce-scatter only this rank's shard of the gradient buffer holds sums: a host-side optimizer_step() over everything would be wrong
		g_last_error = "training_step: the sharded exchange of tcnn_trainer_enable_rccl_sharded steps the optimizer inside training_step (run_optimizer must be true)";
		return TCNN_ERROR;
	}
	if (tm->rccl_comm && gradient_mode == TCNN_GRADIENT_ACCUMULATE) {
		// every all-reduce would sum the ranks' ACCUMULATED buffers again: the earlier micro-batches would be counted once per rank and step
		g_last_error = "training_step: GradientMode::Accumulate cannot be combined with tcnn_trainer_enable_rccl (accumulate locally with the communicator switched off and reduce the buffer once)";
		return TCNN_ERROR;
	}
	tcnn_train_context_t* ctx = nullptr;
	if (g_fused_network_passes.load() && tm->md.has_network && mlp_train_supported(tm->md.net.mlp) && (external_dL_dy || (target && loss_is_elementwise(tm->loss)))) {
		// The step as one graph launch (opt-in; the reference's capture_guard).  Not on the null stream (cuda_graph.h:67-69), not inside a
		// capture of the caller's (:72-76), not with an exchange / ready callback (their collectives live on other streams) or a profiler
		// (its events would be recorded into the graph), and not before a plain step of the same shape has filled the stream's scratch cache.
		const uint64_t shape = (uint64_t)n | ((uint64_t)(dL_dinput != nullptr) << 32) | ((uint64_t)(gradient_mode & 3) << 33) | ((uint64_t)(use_inference_params != 0) << 35) |
		                       ((uint64_t)(external_dL_dy != nullptr) << 36) | ((uint64_t)(ctx_out != nullptr) << 37) | ((uint64_t)(run_optimizer != 0) << 38) |
		                       ((uint64_t)(data_pdf != nullptr) << 39);
		bool capture = tm->graph_capture && stream != nullptr && !tm->profiler && !tm->exchange && !tm->direct.active() && !wants_ready_ranges(tm) && n > 0 &&
		               tm->graph_warm == shape && (debug_launch_flags() & 1) == 0;
		if (capture) {
			hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
			if (hipStreamIsCapturing((hipStream_t)stream, &status) != hipSuccess || status != hipStreamCaptureStatusNone) capture = false;
		}
		if (capture && hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
			(void)hipGetLastError();
			capture = false;
		}
		int r = training_step_fused(tm, (hipStream_t)stream, loss_scale, n, input, target, data_pdf, dL_dinput, use_inference_params, gradient_mode,
		                            run_optimizer != 0, (const half_t*)external_dL_dy, &ctx, /*context_wanted=*/ctx_out != nullptr);
		if (capture) {
			hipGraph_t graph = nullptr;
			const hipError_t ended = hipStreamEndCapture((hipStream_t)stream, &graph);  // (always: the stream must leave capture mode, whatever happened)
			if (r == TCNN_OK && (ended != hipSuccess || !graph)) {
				(void)hipGetLastError();
				tm->graph_capture = false;  // the step was recorded, not run, and its host-side effects (step count) have happened: say so loudly, once
				g_last_error = std::string("training_step: capturing the step into a graph failed (") + hipGetErrorString(ended) + "); graph capture is switched off, this step did not run";
				r = TCNN_ERROR;
			}
			if (r == TCNN_OK) {
				if (tm->graph_exec) {  // patch the instantiated graph with this step's arguments (cuda_graph.h:118-138); a new topology re-instantiates
					hipGraphNode_t error_node = nullptr;
					hipGraphExecUpdateResult result = hipGraphExecUpdateSuccess;
					if (hipGraphExecUpdate(tm->graph_exec, graph, &error_node, &result) != hipSuccess || result != hipGraphExecUpdateSuccess) {
						(void)hipGetLastError();
						(void)hipGraphExecDestroy(tm->graph_exec);
						tm->graph_exec = nullptr;
					}
				}
				hipError_t e = hipSuccess;
				if (!tm->graph_exec) {
					e = hipGraphInstantiate(&tm->graph_exec, graph, nullptr, nullptr, 0);
					++tm->graph_instantiations;
				}
				if (e == hipSuccess) e = hipGraphLaunch(tm->graph_exec, (hipStream_t)stream);
				if (e != hipSuccess) {
					(void)hipGetLastError();
					tm->graph_capture = false;
					g_last_error = std::string("training_step: launching the captured step failed (") + hipGetErrorString(e) + "); graph capture is switched off, this step did not run";
					r = TCNN_ERROR;
				} else {
					++tm->graph_launches;
				}
			}
			if (graph) (void)hipGraphDestroy(graph);
		} else if (r == TCNN_OK) {
			tm->graph_warm = shape;
		}
		if (r == TCNN_OK) r = settle_unstepped_reductions(tm, (hipStream_t)stream, run_optimizer != 0);
		if (ctx_out && r == TCNN_OK) {
			*ctx_out = ctx;
		} else {
			delete ctx;
		}
		return r;
	}
	int r = tcnn_trainer_forward(tm, stream, loss_scale, n, input, target, data_pdf, use_inference_params, dL_dinput != nullptr, external_dL_dy, &ctx);
	if (r == TCNN_OK) r = tcnn_trainer_backward(tm, stream, ctx, n, input, dL_dinput, use_inference_params, gradient_mode);
	if (r == TCNN_OK && gradient_mode != TCNN_GRADIENT_IGNORE && wants_ready_ranges(tm)) {  // this path reports the whole buffer at once
		try {
			notify_gradients_ready(tm, (hipStream_t)stream, 0, tm->md.n_params());
		} catch (const std::exception& ex) {
			g_last_error = ex.what();
			r = TCNN_ERROR;
		}
	}
	if (r == TCNN_OK && run_optimizer) r = finish_training_step(tm, (hipStream_t)stream, loss_scale);
	if (r == TCNN_OK) r = settle_unstepped_reductions(tm, (hipStream_t)stream, run_optimizer != 0);
	if (ctx_out && r == TCNN_OK) {
		*ctx_out = ctx;
	} else {
		delete ctx;
	}
	return r;
}

int tcnn_trainer_loss(tcnn_trainable_model_t* tm, tcnn_stream_t stream_, const tcnn_train_context_t* ctx, float* out_loss) {
	TCNN_API_BEGIN
	hipStream_t stream = (hipStream_t)stream_;
	if (!ctx || !ctx->block_sums.ptr) throw std::runtime_error("Trainer::loss: context holds no loss values (external_dL_dy was used)");
	reduce_sum(stream, ctx->block_sums.as<float>(), ctx->n_block_sums, tm->loss_scratch, tm->loss_scratch + 1024);
	HIP_CHECK(hipMemcpyAsync(out_loss, tm->loss_scratch + 1024, sizeof(float), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	TCNN_API_END
}

void tcnn_train_context_destroy(tcnn_train_context_t* ctx) { delete ctx; }
const void* tcnn_train_context_output(const tcnn_train_context_t* ctx) { return ctx->output.ptr; }
const void* tcnn_train_context_dL_doutput(const tcnn_train_context_t* ctx) { return ctx->dL_doutput_ptr; }

int tcnn_network_inference(tcnn_trainable_model_t* tm, tcnn_stream_t stream_, uint32_t n, const float* input, float* output, int use_inference_params) {
	TCNN_API_BEGIN
	hipStream_t stream = (hipStream_t)stream_;
	inference_to_f32(stream, tm->md, n, input, use_inference_params ? tm->inference_params() : tm->params, output, tm->md.output_width(), 1u);
	TCNN_API_END
}

// ---- GPUMatrixDynamic-typed entry points (gpu_matrix.h:106-250): layout + stride of the caller's matrices honoured ----
static IoLayout layout_of(const tcnn_matrix_t* input, const tcnn_matrix_t* dL_dinput, uint32_t n_input_dims, uint32_t n) {
	auto check = [&](const tcnn_matrix_t* m, const char* what) {
		if (!m->data) throw std::runtime_error(std::string(what) + ": null matrix data");
		if (m->m != n_input_dims || m->n != n) throw std::runtime_error(std::string(what) + ": expected a " + std::to_string(n_input_dims) + " x " + std::to_string(n) + " matrix");
		const uint32_t min_stride = m->layout == TCNN_LAYOUT_COLUMN_MAJOR ? m->m : m->n;
		if (m->stride < min_stride) throw std::runtime_error(std::string(what) + ": stride smaller than the matrix' leading dimension");
	};
	IoLayout l;
	l.set = true;
	check(input, "input");
	const bool cm = input->layout == TCNN_LAYOUT_COLUMN_MAJOR;
	l.in_stride_i = cm ? input->stride : 1u;
	l.in_stride_d = cm ? 1u : input->stride;
	l.dx_stride_i = n_input_dims;
	l.dx_stride_d = 1u;
	if (dL_dinput) {
		check(dL_dinput, "dL_dinput");
		const bool dcm = dL_dinput->layout == TCNN_LAYOUT_COLUMN_MAJOR;
		l.dx_stride_i = dcm ? dL_dinput->stride : 1u;
		l.dx_stride_d = dcm ? 1u : dL_dinput->stride;
	}
	return l;
}
// target / data_pdf / external_dL_dy are GPUMatrix<T> (static column-major) in the reference's signatures (trainer.h:254-264)
static const void* dense_cm(const tcnn_matrix_t* m, uint32_t rows, uint32_t n, const char* what) {
	if (!m) return nullptr;
	if (m->layout != TCNN_LAYOUT_COLUMN_MAJOR || m->stride != m->m || m->m != rows || m->n != n) {
		throw std::runtime_error(std::string(what) + ": expected a dense column-major " + std::to_string(rows) + " x " + std::to_string(n) + " matrix");
	}
	return m->data;
}

int tcnn_trainer_training_step_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_matrix_t* input, const tcnn_matrix_t* target,
                                        const tcnn_matrix_t* data_pdf, int run_optimizer, const tcnn_matrix_t* dL_dinput, int use_inference_params,
                                        int gradient_mode, const tcnn_matrix_t* external_dL_dy, tcnn_train_context_t** ctx_out) {
	TCNN_API_BEGIN
	if (!input) throw std::runtime_error("training_step: input is required");
	const uint32_t n = input->n;
	const IoLayoutGuard guard(layout_of(input, dL_dinput, tm->md.n_input_dims, n));
	const void* ext = dense_cm(external_dL_dy, tm->md.padded_output_width(), n, "external_dL_dy");  // trainer.h:125-126
	// with an external gradient the loss is not evaluated: target and data_pdf are ignored, whatever they are (trainer.h:124-128)
	const void* tgt = ext ? (target && target->data ? target->data : nullptr) : dense_cm(target, tm->md.output_width(), n, "target");
	const void* pdf = ext ? nullptr : dense_cm(data_pdf, tm->md.output_width(), n, "data_pdf");
	const int r = tcnn_trainer_training_step(tm, stream, n, (const float*)input->data, (const float*)tgt, (const float*)pdf, run_optimizer,
	                                         dL_dinput ? (float*)dL_dinput->data : nullptr, use_inference_params, gradient_mode, ext, ctx_out);
	if (r != TCNN_OK) return r;
	TCNN_API_END
}

// Trainer::forward / backward (trainer.h:97-148) with GPUMatrixDynamic inputs of either layout
int tcnn_trainer_forward_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, const tcnn_matrix_t* input, const tcnn_matrix_t* target,
                                  const tcnn_matrix_t* data_pdf, int use_inference_params, int prepare_input_gradients, const tcnn_matrix_t* external_dL_dy,
                                  tcnn_train_context_t** ctx_out) {
	TCNN_API_BEGIN
	if (!input) throw std::runtime_error("forward: input is required");
	const uint32_t n = input->n;
	const IoLayoutGuard guard(layout_of(input, nullptr, tm->md.n_input_dims, n));
	const void* ext = dense_cm(external_dL_dy, tm->md.padded_output_width(), n, "external_dL_dy");
	const void* tgt = ext ? (target && target->data ? target->data : nullptr) : dense_cm(target, tm->md.output_width(), n, "target");
	const void* pdf = ext ? nullptr : dense_cm(data_pdf, tm->md.output_width(), n, "data_pdf");
	const int r = tcnn_trainer_forward(tm, stream, loss_scale, n, (const float*)input->data, (const float*)tgt, (const float*)pdf, use_inference_params, prepare_input_gradients, ext, ctx_out);
	if (r != TCNN_OK) return r;
	TCNN_API_END
}
int tcnn_trainer_backward_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_train_context_t* ctx, const tcnn_matrix_t* input,
                                   const tcnn_matrix_t* dL_dinput, int use_inference_params, int gradient_mode) {
	TCNN_API_BEGIN
	if (!input) throw std::runtime_error("backward: input is required");
	const uint32_t n = input->n;
	const IoLayoutGuard guard(layout_of(input, dL_dinput, tm->md.n_input_dims, n));
	const int r = tcnn_trainer_backward(tm, stream, ctx, n, (const float*)input->data, dL_dinput ? (float*)dL_dinput->data : nullptr, use_inference_params, gradient_mode);
	if (r != TCNN_OK) return r;
	TCNN_API_END
}

int tcnn_network_inference_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream_, const tcnn_matrix_t* input, const tcnn_matrix_t* output,
                                    int use_inference_params) {
	TCNN_API_BEGIN
	if (!input || !output) throw std::runtime_error("inference: input and output are required");
	hipStream_t stream = (hipStream_t)stream_;
	const uint32_t n = input->n, width = tm->md.output_width();
	const IoLayoutGuard guard(layout_of(input, nullptr, tm->md.n_input_dims, n));
	if (output->m != width || output->n != n) throw std::runtime_error("inference: output must be n_output_dims x batch_size");  // object.h:221-222
	const bool cm = output->layout == TCNN_LAYOUT_COLUMN_MAJOR;
	if (output->stride < (cm ? output->m : output->n)) throw std::runtime_error("inference: output stride smaller than its leading dimension");
	inference_to_f32(stream, tm->md, n, (const float*)input->data, use_inference_params ? tm->inference_params() : tm->params, (float*)output->data,
	                 cm ? output->stride : 1u, cm ? 1u : output->stride);
	TCNN_API_END
}

size_t tcnn_trainer_n_params(const tcnn_trainable_model_t* tm) { return tm->md.n_params(); }
// (mutable as well: a caller that writes the fp32 master weights through this pointer breaks "the 16-bit parameters are the rounded
// master weights" just like one that writes the 16-bit buffer, so it is treated the same way until tcnn_trainer_params_written)
float* tcnn_trainer_params_full_precision(tcnn_trainable_model_t* tm) {
	tm->params_exposed = true;
	return tm->master;
}
// read-only view of the same memory: the trainer's mode does not change (a logging / checkpointing host must not switch
// Adam's "16-bit weights follow the master weights" shortcut off for good)
const float* tcnn_trainer_params_full_precision_view(const tcnn_trainable_model_t* tm) { return tm->master; }
const void* tcnn_trainer_params_view(const tcnn_trainable_model_t* tm) { return tm->params; }
void* tcnn_trainer_params(tcnn_trainable_model_t* tm) {
	tm->params_exposed = true;  // a mutable pointer leaves the library: assume the caller writes through it, now or later
	return tm->params;
}
void* tcnn_trainer_params_inference(tcnn_trainable_model_t* tm) {
	if (!tm->ema) tm->params_exposed = true;  // the same buffer as `params` (trainer.h:497-500)
	return tm->inference_params();
}
void* tcnn_trainer_param_gradients(tcnn_trainable_model_t* tm) { return tm->grads; }

int tcnn_trainer_set_params_full_precision(tcnn_trainable_model_t* tm, const float* params, size_t n_params, int device_ptr) {
	TCNN_API_BEGIN
	if (n_params != tm->md.n_params()) throw std::runtime_error("Can't set fp params because buffer has the wrong size.");  // trainer.h:410-412
	HIP_CHECK(hipMemcpy(tm->master, params, sizeof(float) * n_params, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
	cast_master_to_params(tm, nullptr);
	HIP_CHECK(hipDeviceSynchronize());
	TCNN_API_END
}

int tcnn_trainer_set_params(tcnn_trainable_model_t* tm, const void* params_fp16, size_t n_params, int device_ptr) {
	TCNN_API_BEGIN
	if (n_params != tm->md.n_params()) throw std::runtime_error("Can't set params because buffer has the wrong size.");  // trainer.h:424-426
	HIP_CHECK(hipMemcpy(tm->params, params_fp16, sizeof(half_t) * n_params, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
	tm->params_t_valid = false;
	cast_f16_to_f32(nullptr, n_params, tm->params, tm->master);
	HIP_CHECK(hipMemsetAsync(tm->grads, 0, n_params * sizeof(half_t), nullptr));  // reset_param_gradients, trainer.h:437
	HIP_CHECK(hipDeviceSynchronize());
	TCNN_API_END
}

// Trainer::serialize / deserialize, trainer.h:442-481 + adam.h:304-325; document layout in snapshot_msgpack.h.
static size_t n_params_of(const tcnn_trainable_model* tm) { return tm->md.n_params(); }

static Snapshot snapshot_shape(const tcnn_trainable_model* tm, bool with_optimizer) {
	const size_t n = tm->md.n_params();
	Snapshot s;
	s.n_params = n;
	s.params_type = NATIVE_TYPE_NAME;
	s.params.size = n * sizeof(half_t);
	s.has_optimizer = with_optimizer;
	s.current_step = tm->optimizer_step;
	s.base_learning_rate = tm->adam.learning_rate;
	s.first_moments.size = s.second_moments.size = n * sizeof(float);
	s.param_steps.size = n * sizeof(uint32_t);
	s.wrappers = tm->optimizer_order;
	if (tm->ema) s.weights_ema.size = n * sizeof(half_t);
	if (tm->lr_decay) {
		s.base_learning_rate = tm->adam.learning_rate;  // Adam's own (already scaled) rate, adam.h:307
		s.decay_learning_rate = tm->base_lr;
		s.decay_learning_rate_factor = tm->lr_factor;
	}
	return s;
}

int tcnn_trainer_serialize(tcnn_trainable_model_t* tm, int serialize_optimizer, void* buffer, size_t capacity, size_t* n_bytes) {
	TCNN_API_BEGIN
	Snapshot s = snapshot_shape(tm, serialize_optimizer != 0);
	const size_t needed = snapshot_encoded_size(s);
	if (n_bytes) *n_bytes = needed;
	if (buffer) {
		if (capacity < needed) throw std::runtime_error("tcnn_trainer_serialize: buffer too small (" + std::to_string(capacity) + " < " + std::to_string(needed) + " bytes)");
		HIP_CHECK(hipDeviceSynchronize());
		std::vector<uint8_t> host_params(s.params.size), host_m1, host_m2, host_steps, host_ema;
		HIP_CHECK(hipMemcpy(host_params.data(), tm->inference_params(), s.params.size, hipMemcpyDeviceToHost));  // trainer.h:448: params_inference
		s.params.data = host_params.data();
		if (s.has_optimizer) {
			host_m1.resize(s.first_moments.size);
			host_m2.resize(s.second_moments.size);
			host_steps.resize(s.param_steps.size);
			HIP_CHECK(hipMemcpy(host_m1.data(), tm->m1, host_m1.size(), hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(host_m2.data(), tm->m2, host_m2.size(), hipMemcpyDeviceToHost));
			if (tm->steps_form == ADAM_STEPS_DEFICITS8) {  // (the next optimizer step picks its representation again)
				step_counters_to_counter_form(tm, nullptr, tm->optimizer_step);
				HIP_CHECK(hipDeviceSynchronize());
			}
			HIP_CHECK(hipMemcpy(host_steps.data(), tm->steps, host_steps.size(), hipMemcpyDeviceToHost));
			if (tm->steps_form == ADAM_STEPS_DEFICITS32) {  // snapshots hold the counters themselves (adam.h:311)
				uint32_t* counters = (uint32_t*)host_steps.data();
				for (size_t i = 0; i < n_params_of(tm); ++i) counters[i] = tm->optimizer_step - counters[i];
			}
			s.first_moments.data = host_m1.data();
			s.second_moments.data = host_m2.data();
			s.param_steps.data = host_steps.data();
			if (tm->ema) {
				host_ema.resize(s.weights_ema.size);
				HIP_CHECK(hipMemcpy(host_ema.data(), tm->params_ema, host_ema.size(), hipMemcpyDeviceToHost));
				s.weights_ema.data = host_ema.data();
			}
		}
		const std::vector<uint8_t> bytes = snapshot_encode(s);
		std::memcpy(buffer, bytes.data(), bytes.size());
	}
	TCNN_API_END
}

int tcnn_trainer_deserialize(tcnn_trainable_model_t* tm, const void* data, size_t n_bytes) {
	TCNN_API_BEGIN
	const Snapshot s = snapshot_decode(static_cast<const uint8_t*>(data), n_bytes);
	const size_t n = tm->md.n_params();
	if (s.params_type == "float") {
		if (s.params.size != n * sizeof(float)) throw std::runtime_error("Can't set fp params because buffer has the wrong size.");  // trainer.h:410-412
		HIP_CHECK(hipMemcpy(tm->master, s.params.data, s.params.size, hipMemcpyHostToDevice));
		cast_master_to_params(tm, nullptr);
	} else if (s.params_type == NATIVE_TYPE_NAME) {
		if (s.params.size != n * sizeof(half_t)) throw std::runtime_error("Can't set params because buffer has the wrong size.");  // trainer.h:424-426
		HIP_CHECK(hipMemcpy(tm->params, s.params.data, s.params.size, hipMemcpyHostToDevice));
		tm->params_t_valid = false;
		cast_f16_to_f32(nullptr, n, tm->params, tm->master);
		HIP_CHECK(hipMemsetAsync(tm->grads, 0, n * sizeof(half_t), nullptr));
	} else {
		throw std::runtime_error(std::string("Trainer: snapshot parameters must be of type float of ") + NATIVE_TYPE_NAME);  // trainer.h:473
	}
	if (s.has_optimizer) {
		if (!s.first_moments.present() || !s.second_moments.present() || s.first_moments.size != n * sizeof(float) || s.second_moments.size != n * sizeof(float))
			throw std::runtime_error("Trainer: optimizer snapshot does not match the number of parameters");
		HIP_CHECK(hipMemcpy(tm->m1, s.first_moments.data, s.first_moments.size, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(tm->m2, s.second_moments.data, s.second_moments.size, hipMemcpyHostToDevice));
		if (s.param_steps.present()) {  // adam.h:317-322: older snapshots carry no per-parameter steps
			if (s.param_steps.size != n * sizeof(uint32_t)) throw std::runtime_error("Trainer: optimizer snapshot does not match the number of parameters");
			HIP_CHECK(hipMemcpy(tm->steps, s.param_steps.data, s.param_steps.size, hipMemcpyHostToDevice));
		} else {
			HIP_CHECK(hipMemset(tm->steps, 0, n * sizeof(uint32_t)));
		}
		tm->steps_form = ADAM_STEPS_COUNTERS;  // the next optimizer step picks the representation again
		tm->optimizer_step = s.current_step;
		tm->adam.learning_rate = s.base_learning_rate;
		if (tm->ema) {  // ema.h:195-204
			if (!s.weights_ema.present() || s.weights_ema.size != n * sizeof(half_t)) throw std::runtime_error("Trainer: EMA snapshot does not match the number of parameters");
			HIP_CHECK(hipMemcpy(tm->params_ema, s.weights_ema.data, s.weights_ema.size, hipMemcpyHostToDevice));
			if (tm->ema_tmp) cast_f16_to_f32(nullptr, n, tm->params_ema, tm->ema_tmp);
		}
		if (tm->lr_decay && s.has_decay) {  // exponential_decay.h:144-148
			tm->base_lr = s.decay_learning_rate;
			tm->lr_factor = s.decay_learning_rate_factor;
		}
		refresh_hyper_json(tm);
	}
	HIP_CHECK(hipDeviceSynchronize());
	TCNN_API_END
}

int tcnn_trainer_update_hyperparams(tcnn_trainable_model_t* tm, const char* json) {
	TCNN_API_BEGIN
	const Json j = Json::parse(json);
	if (j.contains("optimizer")) apply_optimizer_json(tm, j.value("optimizer", Json::object()), /*creating=*/false);
	refresh_hyper_json(tm);
	TCNN_API_END
}
const char* tcnn_trainer_hyperparams_json(tcnn_trainable_model_t* tm) {
	refresh_hyper_json(tm);  // the learning rate moves with the ExponentialDecay schedule
	return tm->hyper_json.c_str();
}
uint32_t tcnn_trainer_optimizer_step_count(const tcnn_trainable_model_t* tm) { return tm->optimizer_step; }
uint32_t tcnn_trainer_padded_output_width(const tcnn_trainable_model_t* tm) { return tm->md.padded_output_width(); }
uint32_t tcnn_trainer_n_mlp_params(const tcnn_trainable_model_t* tm) { return (uint32_t)tm->md.n_mlp_params(); }

// tcnn_trainer_params / _params_inference hand out a mutable pointer: from then on the transposed copy of the network weights is rebuilt
// before every pass.  A caller that has finished writing says so here; the copy is rebuilt once more and then trusted again.
// Which buffer is authoritative after direct writes: the 16-bit parameters (what the kernels compute with, what tcnn_trainer_params
// hands out).  The fp32 master weights are re-derived from them wherever the two disagree -- a parameter whose 16-bit value still IS
// its rounded master weight keeps the master's extra bits -- so a caller that wrote only the master buffer calls
// tcnn_trainer_set_params_full_precision instead.  After this call Adam's "16-bit weights follow the master weights" shortcut holds again.
int tcnn_trainer_params_written(tcnn_trainable_model_t* tm) {
	TCNN_API_BEGIN
	resync_master_from_half(nullptr, tm->md.n_params(), tm->params, tm->master);
	HIP_CHECK(hipDeviceSynchronize());
	tm->params_exposed = false;
	tm->params_t_valid = false;
	TCNN_API_END
}

int tcnn_trainer_set_global_batch_size(tcnn_trainable_model_t* tm, uint64_t global_batch_size) {
	tm->global_batch = global_batch_size;
	return TCNN_OK;
}

int tcnn_trainer_set_profiling(tcnn_trainable_model_t* tm, int enable, int only_stage) {
	TCNN_API_BEGIN
	if (!enable) {
		tm->profiler.reset();
	} else {
		tm->profiler = std::make_unique<Profiler>();
		tm->profiler->only_stage = only_stage;
		// events for the first steps exist before the caller's timed region begins (hipEventCreate costs host time: created on
		// first use they slowed a 20-step measurement by ~5 %)
		const size_t ahead = only_stage >= 0 ? 512 : 2048;
		for (size_t i = 0; i < ahead; ++i) (void)tm->profiler->get();
		tm->profiler->next = 0;
	}
	TCNN_API_END
}
int tcnn_trainer_n_stages(void) { return N_STAGES; }
const char* tcnn_trainer_stage_name(int stage) { return stage >= 0 && stage < N_STAGES ? STAGE_NAMES[stage] : ""; }
int tcnn_trainer_get_stage_times(tcnn_trainable_model_t* tm, double* total_ms, uint64_t* counts) {
	TCNN_API_BEGIN
	if (!tm->profiler) throw std::runtime_error("profiling is not enabled");
	tm->profiler->collect();
	for (int i = 0; i < N_STAGES; ++i) {
		total_ms[i] = tm->profiler->total_ms[i];
		counts[i] = tm->profiler->count[i];
	}
	TCNN_API_END
}
int tcnn_trainer_set_lds_level_budget(tcnn_trainable_model_t* tm, uint32_t bytes) {
	tm->lds_level_budget = bytes;
	return TCNN_OK;
}
int tcnn_set_fused_identity_input(int enable) {
	g_fused_identity_input.store(enable != 0 ? 1 : 0);
	return TCNN_OK;
}
int tcnn_set_finalize_in_optimizer(int enable) {
	g_finalize_in_optimizer.store(enable != 0 ? 1 : 0);
	return TCNN_OK;
}
int tcnn_get_fused_network_passes(void) { return g_fused_network_passes.load(); }
int tcnn_trainer_set_graph_capture(tcnn_trainable_model_t* tm, int enable) {
	TCNN_API_BEGIN
	if (!tm) throw std::runtime_error("tcnn_trainer_set_graph_capture: null model");
	tm->graph_capture = enable != 0;
	if (!enable && tm->graph_exec) {
		HIP_CHECK(hipDeviceSynchronize());
		(void)hipGraphExecDestroy(tm->graph_exec);
		tm->graph_exec = nullptr;
	}
	TCNN_API_END
}
int tcnn_trainer_graph_capture_stats(const tcnn_trainable_model_t* tm, uint64_t* launches, uint64_t* instantiations) {
	TCNN_API_BEGIN
	if (!tm) throw std::runtime_error("tcnn_trainer_graph_capture_stats: null model");
	if (launches) *launches = tm->graph_launches;
	if (instantiations) *instantiations = tm->graph_instantiations;
	TCNN_API_END
}
int tcnn_set_fused_network_passes(int enable) {
	g_fused_network_passes.store(enable != 0 ? 1 : 0);
	return TCNN_OK;
}
int tcnn_grid_owner_wide_slices(uint64_t* out) {
	TCNN_API_BEGIN
	*out = grid_owner_wide_slices();
	TCNN_API_END
}
int tcnn_get_grid_owner_mode(void) { return grid_owner_mode(); }
int tcnn_set_grid_owner_mode(int mode) {
	if (mode < 0 || mode > 2) return TCNN_ERROR;
	grid_owner_mode() = mode;
	return TCNN_OK;
}
int tcnn_get_grid_backward_mode(void) { return g_grid_backward_mode.load(); }
int tcnn_set_grid_backward_mode(int mode) {
	if (mode < 0 || mode > 3) return TCNN_ERROR;
	g_grid_backward_mode.store(mode);
	return TCNN_OK;
}

}  // extern "C"
