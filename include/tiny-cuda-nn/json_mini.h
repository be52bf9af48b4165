// json_mini.h -- a small JSON value (parse + dump) for the configuration surface.
// The reference uses nlohmann::json (dependencies/json/json.hpp) with `.value(key, default)` lookups
// everywhere (network.cu:101-138, grid.h:1727-1755, adam.h:221-281); this restates just that subset.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn_hip {

class Json {
public:
	enum class Type { Null, Bool, Number, String, Array, Object };

	Json() : m_type(Type::Null) {}
	Json(bool b) : m_type(Type::Bool), m_bool(b) {}
	Json(double d) : m_type(Type::Number), m_num(d) {}
	Json(int d) : m_type(Type::Number), m_num((double)d) {}
	Json(uint32_t d) : m_type(Type::Number), m_num((double)d) {}
	Json(float d) : m_type(Type::Number), m_num((double)d), m_is_float(true) {}
	Json(const char* s) : m_type(Type::String), m_str(s) {}
	Json(const std::string& s) : m_type(Type::String), m_str(s) {}

	static Json object() {
		Json j;
		j.m_type = Type::Object;
		return j;
	}
	static Json array() {
		Json j;
		j.m_type = Type::Array;
		return j;
	}

	static Json parse(const std::string& text) {
		size_t pos = 0;
		Json j = parse_value(text, pos);
		skip_ws(text, pos);
		if (pos != text.size()) throw std::runtime_error("JSON: trailing characters at offset " + std::to_string(pos));
		return j;
	}

	Type type() const { return m_type; }
	bool is_object() const { return m_type == Type::Object; }
	bool is_null() const { return m_type == Type::Null; }

	bool contains(const std::string& key) const { return m_type == Type::Object && m_obj.count(key) > 0; }
	const Json& operator[](const std::string& key) const {
		if (m_type != Type::Object) throw std::runtime_error("JSON: not an object (looking up '" + key + "')");
		auto it = m_obj.find(key);
		if (it == m_obj.end()) throw std::runtime_error("JSON: key '" + key + "' not found");
		return it->second;
	}
	Json& operator[](const std::string& key) {
		if (m_type == Type::Null) m_type = Type::Object;
		if (m_type != Type::Object) throw std::runtime_error("JSON: not an object (assigning '" + key + "')");
		if (!m_obj.count(key)) m_keys.push_back(key);
		return m_obj[key];
	}

	double as_number() const {
		if (m_type == Type::Number) return m_num;
		if (m_type == Type::Bool) return m_bool ? 1.0 : 0.0;
		throw std::runtime_error("JSON: value is not a number");
	}
	bool as_bool() const {
		if (m_type == Type::Bool) return m_bool;
		if (m_type == Type::Number) return m_num != 0.0;
		throw std::runtime_error("JSON: value is not a boolean");
	}
	const std::string& as_string() const {
		if (m_type != Type::String) throw std::runtime_error("JSON: value is not a string");
		return m_str;
	}

	float value(const std::string& key, float def) const { return contains(key) ? (float)(*this)[key].as_number() : def; }
	uint32_t value(const std::string& key, uint32_t def) const { return contains(key) ? (uint32_t)(*this)[key].as_number() : def; }
	bool value(const std::string& key, bool def) const { return contains(key) ? (*this)[key].as_bool() : def; }
	std::string value(const std::string& key, const char* def) const { return contains(key) ? (*this)[key].as_string() : std::string(def); }
	std::string value(const std::string& key, const std::string& def) const { return contains(key) ? (*this)[key].as_string() : def; }
	Json value(const std::string& key, const Json& def) const { return contains(key) ? (*this)[key] : def; }

	std::string dump() const {
		std::ostringstream o;
		dump_to(o);
		return o.str();
	}

private:
	Type m_type;
	bool m_bool = false;
	double m_num = 0.0;
	bool m_is_float = false;
	std::string m_str;
	std::vector<Json> m_arr;
	std::map<std::string, Json> m_obj;
	std::vector<std::string> m_keys;  // insertion order for dump()

	void dump_to(std::ostringstream& o) const {
		switch (m_type) {
			case Type::Null: o << "null"; break;
			case Type::Bool: o << (m_bool ? "true" : "false"); break;
			case Type::Number: {
				char buf[64];
				if (m_num == (double)(long long)m_num && m_num > -1e15 && m_num < 1e15 && !m_is_float) {
					snprintf(buf, sizeof(buf), "%lld", (long long)m_num);
				} else {
					snprintf(buf, sizeof(buf), m_is_float ? "%.9g" : "%.17g", m_num);
				}
				o << buf;
				break;
			}
			case Type::String: {
				o << '"';
				for (char c : m_str) {
					if (c == '"' || c == '\\') o << '\\' << c;
					else if (c == '\n') o << "\\n";
					else if (c == '\t') o << "\\t";
					else o << c;
				}
				o << '"';
				break;
			}
			case Type::Array: {
				o << '[';
				for (size_t i = 0; i < m_arr.size(); ++i) {
					if (i) o << ',';
					m_arr[i].dump_to(o);
				}
				o << ']';
				break;
			}
			case Type::Object: {
				o << '{';
				bool first = true;
				for (const auto& k : m_keys) {
					if (!first) o << ',';
					first = false;
					o << '"' << k << "\":";
					m_obj.at(k).dump_to(o);
				}
				o << '}';
				break;
			}
		}
	}

	static void skip_ws(const std::string& s, size_t& p) {
		while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p;
	}

	static Json parse_value(const std::string& s, size_t& p) {
		skip_ws(s, p);
		if (p >= s.size()) throw std::runtime_error("JSON: unexpected end of input");
		const char c = s[p];
		if (c == '{') {
			Json j = Json::object();
			++p;
			skip_ws(s, p);
			if (p < s.size() && s[p] == '}') {
				++p;
				return j;
			}
			while (true) {
				skip_ws(s, p);
				if (p >= s.size() || s[p] != '"') throw std::runtime_error("JSON: expected a string key at offset " + std::to_string(p));
				std::string key = parse_string(s, p);
				skip_ws(s, p);
				if (p >= s.size() || s[p] != ':') throw std::runtime_error("JSON: expected ':' at offset " + std::to_string(p));
				++p;
				j[key] = parse_value(s, p);  // later duplicates win, as in nlohmann (tests/test_grid.cu:40-48 has two "otype")
				skip_ws(s, p);
				if (p < s.size() && s[p] == ',') {
					++p;
					continue;
				}
				if (p < s.size() && s[p] == '}') {
					++p;
					return j;
				}
				throw std::runtime_error("JSON: expected ',' or '}' at offset " + std::to_string(p));
			}
		}
		if (c == '[') {
			Json j = Json::array();
			++p;
			skip_ws(s, p);
			if (p < s.size() && s[p] == ']') {
				++p;
				return j;
			}
			while (true) {
				j.m_arr.push_back(parse_value(s, p));
				skip_ws(s, p);
				if (p < s.size() && s[p] == ',') {
					++p;
					continue;
				}
				if (p < s.size() && s[p] == ']') {
					++p;
					return j;
				}
				throw std::runtime_error("JSON: expected ',' or ']' at offset " + std::to_string(p));
			}
		}
		if (c == '"') return Json(parse_string(s, p));
		if (s.compare(p, 4, "true") == 0) {
			p += 4;
			return Json(true);
		}
		if (s.compare(p, 5, "false") == 0) {
			p += 5;
			return Json(false);
		}
		if (s.compare(p, 4, "null") == 0) {
			p += 4;
			return Json();
		}
		{
			const char* start = s.c_str() + p;
			char* end = nullptr;
			const double d = strtod(start, &end);
			if (end == start) throw std::runtime_error("JSON: unexpected character '" + std::string(1, c) + "' at offset " + std::to_string(p));
			p += (size_t)(end - start);
			return Json(d);
		}
	}

	static std::string parse_string(const std::string& s, size_t& p) {
		std::string out;
		++p;  // opening quote
		while (p < s.size() && s[p] != '"') {
			if (s[p] == '\\' && p + 1 < s.size()) {
				++p;
				switch (s[p]) {
					case 'n': out += '\n'; break;
					case 't': out += '\t'; break;
					case 'r': out += '\r'; break;
					case 'b': out += '\b'; break;
					case 'f': out += '\f'; break;
					default: out += s[p]; break;
				}
			} else {
				out += s[p];
			}
			++p;
		}
		if (p >= s.size()) throw std::runtime_error("JSON: unterminated string");
		++p;  // closing quote
		return out;
	}
};

inline bool equals_case_insensitive(const std::string& a, const std::string& b) {
	if (a.size() != b.size()) return false;
	for (size_t i = 0; i < a.size(); ++i) {
		if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
	}
	return true;
}

}  // namespace tcnn_hip
