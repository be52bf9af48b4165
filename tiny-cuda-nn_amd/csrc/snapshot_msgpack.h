// snapshot_msgpack.h -- the trainer snapshot as MessagePack bytes (host only, no device code).
//
// The reference's Trainer::serialize() returns an nlohmann::json object (trainer.h:442-455):
//   { "n_params": uint, "params_type": "__half"|"float", "params_binary": <binary>,
//     "optimizer": { "current_step": uint, "base_learning_rate": float, "first_moments_binary": <binary>,
//                    "second_moments_binary": <binary>, "param_steps_binary": <binary> } }     (adam.h:304-312)
// and hosts persist it with json::to_msgpack (binary members have no JSON-text form).  This header writes / reads
// exactly that document: keys in lexicographic order (nlohmann's std::map), unsigned integers in the shortest
// positive form, floats as float32 when exactly representable, binaries as bin8/16/32 without subtype -- i.e. the
// bytes nlohmann::json::to_msgpack() produces for the object above, so snapshots move between the two
// implementations.  The reader accepts any key order, skips unknown keys, and also takes the
// {"bytes":[...]} object form of a binary (gpu_memory_json.h:58-67).
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn_hip {

struct SnapshotBlob {
	const uint8_t* data = nullptr;
	size_t size = 0;
	std::vector<uint8_t> owned;  // backing store when the input was a {"bytes":[...]} array
	bool present() const { return data != nullptr; }
};

struct Snapshot {
	uint64_t n_params = 0;
	std::string params_type = "__half";
	SnapshotBlob params;
	bool has_optimizer = false;
	uint32_t current_step = 0;
	float base_learning_rate = 0.f;
	SnapshotBlob first_moments, second_moments, param_steps;
	// wrapper optimizers, outermost first ("Ema" / "ExponentialDecay"): each nests the next one's state under "nested"
	// (ema.h:190-205: {"nested", "weights_ema_binary"}; exponential_decay.h:135-148: {"nested", "learning_rate", "learning_rate_factor"})
	std::vector<std::string> wrappers;
	SnapshotBlob weights_ema;
	bool has_decay = false;
	float decay_learning_rate = 0.f, decay_learning_rate_factor = 1.f;
};

namespace msgpack_detail {

struct Writer {
	std::vector<uint8_t>& out;
	bool count_only = false;  // size query: binary payloads are counted in `skipped`, not copied
	size_t skipped = 0;
	void byte(uint8_t b) { out.push_back(b); }
	template <typename T>
	void big_endian(T v) {
		for (int i = int(sizeof(T)) - 1; i >= 0; --i) out.push_back(uint8_t(v >> (8 * i)));
	}
	void map(uint32_t n) {
		if (n <= 15) byte(0x80 | n);
		else if (n <= 0xffff) { byte(0xde); big_endian<uint16_t>(n); }
		else { byte(0xdf); big_endian<uint32_t>(n); }
	}
	void str(const std::string& s) {
		const size_t n = s.size();
		if (n <= 31) byte(0xa0 | uint8_t(n));
		else if (n <= 0xff) { byte(0xd9); byte(uint8_t(n)); }
		else if (n <= 0xffff) { byte(0xda); big_endian<uint16_t>(uint16_t(n)); }
		else { byte(0xdb); big_endian<uint32_t>(uint32_t(n)); }
		out.insert(out.end(), s.begin(), s.end());
	}
	void uint(uint64_t v) {
		if (v < 128) byte(uint8_t(v));
		else if (v <= 0xff) { byte(0xcc); byte(uint8_t(v)); }
		else if (v <= 0xffff) { byte(0xcd); big_endian<uint16_t>(uint16_t(v)); }
		else if (v <= 0xffffffffull) { byte(0xce); big_endian<uint32_t>(uint32_t(v)); }
		else { byte(0xcf); big_endian<uint64_t>(v); }
	}
	void real(double v) {
		const float f = float(v);
		if (double(f) == v) { uint32_t u; std::memcpy(&u, &f, 4); byte(0xca); big_endian<uint32_t>(u); }
		else { uint64_t u; std::memcpy(&u, &v, 8); byte(0xcb); big_endian<uint64_t>(u); }
	}
	void bin(const uint8_t* p, size_t n) {
		if (n <= 0xff) { byte(0xc4); byte(uint8_t(n)); }
		else if (n <= 0xffff) { byte(0xc5); big_endian<uint16_t>(uint16_t(n)); }
		else if (n <= 0xffffffffull) { byte(0xc6); big_endian<uint32_t>(uint32_t(n)); }
		else throw std::runtime_error("snapshot: binary member exceeds the 4 GiB MessagePack limit");
		if (count_only) skipped += n;
		else out.insert(out.end(), p, p + n);
	}
};

struct Reader {
	const uint8_t* p;
	const uint8_t* end;
	void need(size_t n) const { if (size_t(end - p) < n) throw std::runtime_error("snapshot: truncated MessagePack data"); }
	uint8_t byte() { need(1); return *p++; }
	template <typename T>
	T big_endian() {
		need(sizeof(T));
		T v = 0;
		for (size_t i = 0; i < sizeof(T); ++i) v = T(v << 8) | T(*p++);
		return v;
	}
	uint8_t peek() const { need(1); return *p; }

	// generic value categories
	bool is_map() const { const uint8_t t = peek(); return (t & 0xf0) == 0x80 || t == 0xde || t == 0xdf; }
	bool is_bin() const { const uint8_t t = peek(); return t >= 0xc4 && t <= 0xc6; }
	bool is_array() const { const uint8_t t = peek(); return (t & 0xf0) == 0x90 || t == 0xdc || t == 0xdd; }

	uint32_t map_header() {
		const uint8_t t = byte();
		if ((t & 0xf0) == 0x80) return t & 0x0f;
		if (t == 0xde) return big_endian<uint16_t>();
		if (t == 0xdf) return big_endian<uint32_t>();
		throw std::runtime_error("snapshot: expected a map");
	}
	uint32_t array_header() {
		const uint8_t t = byte();
		if ((t & 0xf0) == 0x90) return t & 0x0f;
		if (t == 0xdc) return big_endian<uint16_t>();
		if (t == 0xdd) return big_endian<uint32_t>();
		throw std::runtime_error("snapshot: expected an array");
	}
	std::string str() {
		const uint8_t t = byte();
		size_t n;
		if ((t & 0xe0) == 0xa0) n = t & 0x1f;
		else if (t == 0xd9) n = byte();
		else if (t == 0xda) n = big_endian<uint16_t>();
		else if (t == 0xdb) n = big_endian<uint32_t>();
		else throw std::runtime_error("snapshot: expected a string");
		need(n);
		std::string s(reinterpret_cast<const char*>(p), n);
		p += n;
		return s;
	}
	double number() {
		const uint8_t t = byte();
		if (t < 0x80) return t;
		if (t >= 0xe0) return int8_t(t);
		switch (t) {
			case 0xcc: return byte();
			case 0xcd: return big_endian<uint16_t>();
			case 0xce: return big_endian<uint32_t>();
			case 0xcf: return double(big_endian<uint64_t>());
			case 0xd0: return int8_t(byte());
			case 0xd1: return int16_t(big_endian<uint16_t>());
			case 0xd2: return int32_t(big_endian<uint32_t>());
			case 0xd3: return double(int64_t(big_endian<uint64_t>()));
			case 0xca: { const uint32_t u = big_endian<uint32_t>(); float f; std::memcpy(&f, &u, 4); return f; }
			case 0xcb: { const uint64_t u = big_endian<uint64_t>(); double d; std::memcpy(&d, &u, 8); return d; }
			default: throw std::runtime_error("snapshot: expected a number");
		}
	}
	uint64_t unsigned_integer() {
		const uint8_t t = peek();
		if (t == 0xcf) { byte(); return big_endian<uint64_t>(); }
		const double v = number();
		if (v < 0) throw std::runtime_error("snapshot: expected a non-negative integer");
		return uint64_t(v);
	}
	void bin(SnapshotBlob& out) {
		const uint8_t t = byte();
		size_t n;
		if (t == 0xc4) n = byte();
		else if (t == 0xc5) n = big_endian<uint16_t>();
		else if (t == 0xc6) n = big_endian<uint32_t>();
		else throw std::runtime_error("snapshot: expected a binary");
		need(n);
		out.data = p;
		out.size = n;
		p += n;
	}
	void skip() {
		const uint8_t t = byte();
		auto adv = [&](size_t n) { need(n); p += n; };
		if (t < 0x80 || t >= 0xe0) return;
		if ((t & 0xf0) == 0x80) { for (uint32_t i = 0, n = t & 0x0f; i < 2 * n; ++i) skip(); return; }
		if ((t & 0xf0) == 0x90) { for (uint32_t i = 0, n = t & 0x0f; i < n; ++i) skip(); return; }
		if ((t & 0xe0) == 0xa0) { adv(t & 0x1f); return; }
		switch (t) {
			case 0xc0: case 0xc2: case 0xc3: return;
			case 0xc4: case 0xd9: adv(byte()); return;
			case 0xc5: case 0xda: adv(big_endian<uint16_t>()); return;
			case 0xc6: case 0xdb: adv(big_endian<uint32_t>()); return;
			case 0xc7: { const size_t n = byte(); adv(n + 1); return; }
			case 0xc8: { const size_t n = big_endian<uint16_t>(); adv(n + 1); return; }
			case 0xc9: { const size_t n = big_endian<uint32_t>(); adv(n + 1); return; }
			case 0xca: case 0xce: case 0xd2: adv(4); return;
			case 0xcb: case 0xcf: case 0xd3: adv(8); return;
			case 0xcc: case 0xd0: adv(1); return;
			case 0xcd: case 0xd1: adv(2); return;
			case 0xd4: adv(2); return;
			case 0xd5: adv(3); return;
			case 0xd6: adv(5); return;
			case 0xd7: adv(9); return;
			case 0xd8: adv(17); return;
			case 0xdc: { for (uint32_t i = 0, n = big_endian<uint16_t>(); i < n; ++i) skip(); return; }
			case 0xdd: { for (uint32_t i = 0, n = big_endian<uint32_t>(); i < n; ++i) skip(); return; }
			case 0xde: { for (uint32_t i = 0, n = big_endian<uint16_t>(); i < 2 * n; ++i) skip(); return; }
			case 0xdf: { for (uint32_t i = 0, n = big_endian<uint32_t>(); i < 2 * n; ++i) skip(); return; }
			default: throw std::runtime_error("snapshot: unsupported MessagePack type byte");
		}
	}
	// a binary member: either a bin, or the {"bytes":[...], "subtype":...} object form (gpu_memory_json.h:58-67)
	void blob(SnapshotBlob& out) {
		if (is_bin()) { bin(out); return; }
		if (!is_map()) throw std::runtime_error("Invalid json type: must be either binary or object");  // gpu_memory_json.h:69
		bool found = false;
		for (uint32_t i = 0, n = map_header(); i < n; ++i) {
			if (str() == "bytes" && is_array()) {
				const uint32_t len = array_header();
				out.owned.resize(len);
				for (uint32_t k = 0; k < len; ++k) out.owned[k] = uint8_t(number());
				out.data = out.owned.data();
				out.size = len;
				found = true;
			} else skip();
		}
		if (!found) throw std::runtime_error("Invalid json type: must be either binary or object");
	}
};

}  // namespace msgpack_detail

namespace msgpack_detail {
inline void write_snapshot(Writer& w, const Snapshot& s);
inline void write_optimizer(Writer& w, const Snapshot& s, size_t depth);
inline void read_optimizer(Reader& r, Snapshot& s);
}

// Bytes snapshot_encode() will produce; only the blob SIZES of `s` are read.
inline size_t snapshot_encoded_size(const Snapshot& s) {
	std::vector<uint8_t> header;
	msgpack_detail::Writer w{header, true};
	msgpack_detail::write_snapshot(w, s);
	return header.size() + w.skipped;
}

inline std::vector<uint8_t> snapshot_encode(const Snapshot& s) {
	std::vector<uint8_t> out;
	out.reserve(128 + s.params.size + s.first_moments.size + s.second_moments.size + s.param_steps.size);
	msgpack_detail::Writer w{out};
	msgpack_detail::write_snapshot(w, s);
	return out;
}

// keys of every map in lexicographic order, as nlohmann's std::map emits them
inline void msgpack_detail::write_optimizer(Writer& w, const Snapshot& s, size_t depth) {
	if (depth < s.wrappers.size()) {
		if (s.wrappers[depth] == "Ema") {
			w.map(2);
			w.str("nested"); write_optimizer(w, s, depth + 1);
			w.str("weights_ema_binary"); w.bin(s.weights_ema.data, s.weights_ema.size);
		} else {  // ExponentialDecay
			w.map(3);
			w.str("learning_rate"); w.real(double(s.decay_learning_rate));
			w.str("learning_rate_factor"); w.real(double(s.decay_learning_rate_factor));
			w.str("nested"); write_optimizer(w, s, depth + 1);
		}
		return;
	}
	w.map(5);  // Adam, adam.h:304-312
	w.str("base_learning_rate"); w.real(double(s.base_learning_rate));
	w.str("current_step"); w.uint(s.current_step);
	w.str("first_moments_binary"); w.bin(s.first_moments.data, s.first_moments.size);
	w.str("param_steps_binary"); w.bin(s.param_steps.data, s.param_steps.size);
	w.str("second_moments_binary"); w.bin(s.second_moments.data, s.second_moments.size);
}

inline void msgpack_detail::write_snapshot(Writer& w, const Snapshot& s) {
	w.map(s.has_optimizer ? 4 : 3);
	w.str("n_params"); w.uint(s.n_params);
	if (s.has_optimizer) {
		w.str("optimizer");
		write_optimizer(w, s, 0);
	}
	w.str("params_binary"); w.bin(s.params.data, s.params.size);
	w.str("params_type"); w.str(s.params_type);
}

// one optimizer state map; wrapper optimizers hold the next one under "nested" -- their own keys are distinct, so the
// fields are collected by name whatever the nesting order
inline void msgpack_detail::read_optimizer(Reader& r, Snapshot& s) {
	for (uint32_t k = 0, m = r.map_header(); k < m; ++k) {
		const std::string okey = r.str();
		if (okey == "current_step") s.current_step = uint32_t(r.unsigned_integer());
		else if (okey == "base_learning_rate") s.base_learning_rate = float(r.number());
		else if (okey == "first_moments_binary") r.blob(s.first_moments);
		else if (okey == "second_moments_binary") r.blob(s.second_moments);
		else if (okey == "param_steps_binary") r.blob(s.param_steps);
		else if (okey == "weights_ema_binary") r.blob(s.weights_ema);
		else if (okey == "learning_rate") { s.decay_learning_rate = float(r.number()); s.has_decay = true; }
		else if (okey == "learning_rate_factor") s.decay_learning_rate_factor = float(r.number());
		else if (okey == "nested" && r.is_map()) read_optimizer(r, s);
		else r.skip();
	}
}

// Blobs point into [data, data + size) unless they came from a {"bytes":[...]} array.
inline Snapshot snapshot_decode(const uint8_t* data, size_t size) {
	msgpack_detail::Reader r{data, data + size};
	Snapshot s;
	bool have_type = false;
	for (uint32_t i = 0, n = r.map_header(); i < n; ++i) {
		const std::string key = r.str();
		if (key == "n_params") s.n_params = r.unsigned_integer();
		else if (key == "params_type") { s.params_type = r.str(); have_type = true; }
		else if (key == "params_binary") r.blob(s.params);
		else if (key == "optimizer") {
			s.has_optimizer = true;
			msgpack_detail::read_optimizer(r, s);
		} else r.skip();
	}
	(void)have_type;  // absent -> the trainer's own parameter type, trainer.h:458
	if (!s.params.present()) throw std::runtime_error("snapshot: missing params_binary");
	return s;
}

}  // namespace tcnn_hip
