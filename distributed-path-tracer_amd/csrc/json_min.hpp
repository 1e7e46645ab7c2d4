// Small recursive-descent JSON reader shared by the glTF loader and the worker-event front-end.
// Written for this library (the reference vendors cgltf / nlohmann-json; neither is used here).
// Numbers are converted like cgltf does: strtod on the token, cast to float on use.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "flat_scene.hpp"

namespace ptx {
namespace {

[[noreturn]] inline void fail(int code, const std::string& m) { throw Error{code, m}; }
constexpr int E_IO = 2, E_PARSE = 3, E_NO_CAMERA = 4;

struct JVal {
	enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
	bool b = false;
	double num = 0;
	std::string str;
	std::vector<JVal> arr;
	std::vector<std::pair<std::string, JVal>> obj;  // insertion order is kept (attribute order matters)

	const JVal* find(const char* key) const {
		if (kind != Obj) return nullptr;
		for (auto& kv : obj) if (kv.first == key) return &kv.second;
		return nullptr;
	}
	bool has(const char* key) const { return find(key) != nullptr; }
	const JVal& at(const char* key) const {
		const JVal* v = find(key);
		if (!v) fail(E_PARSE, std::string("glTF: missing key '") + key + "'");
		return *v;
	}
	const JVal& el(size_t i) const {
		if (kind != Arr || i >= arr.size()) fail(E_PARSE, "glTF: array index out of range");
		return arr[i];
	}
	size_t size() const { return kind == Arr ? arr.size() : 0; }
	float f() const { if (kind != Num) fail(E_PARSE, "glTF: number expected"); return (float)num; }
	int64_t i() const { if (kind != Num) fail(E_PARSE, "glTF: integer expected"); return (int64_t)num; }
	const std::string& s() const { if (kind != Str) fail(E_PARSE, "glTF: string expected"); return str; }
};

class JsonReader {
public:
	explicit JsonReader(const std::string& t) : p_(t.data()), end_(t.data() + t.size()) {}
	JVal parse() {
		JVal v = value();
		ws();
		if (p_ != end_) fail(E_PARSE, "JSON: trailing characters");
		return v;
	}
private:
	const char* p_;
	const char* end_;
	void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++; }
	char peek() { ws(); if (p_ >= end_) fail(E_PARSE, "JSON: unexpected end"); return *p_; }
	void expect(char c) { if (peek() != c) fail(E_PARSE, std::string("JSON: expected '") + c + "'"); p_++; }
	JVal value() {
		char c = peek();
		JVal v;
		if (c == '{') {
			v.kind = JVal::Obj; p_++;
			if (peek() == '}') { p_++; return v; }
			for (;;) {
				std::string k = string();
				expect(':');
				v.obj.emplace_back(std::move(k), value());
				if (peek() == ',') { p_++; continue; }
				expect('}');
				return v;
			}
		}
		if (c == '[') {
			v.kind = JVal::Arr; p_++;
			if (peek() == ']') { p_++; return v; }
			for (;;) {
				v.arr.push_back(value());
				if (peek() == ',') { p_++; continue; }
				expect(']');
				return v;
			}
		}
		if (c == '"') { v.kind = JVal::Str; v.str = string(); return v; }
		if (!strncmp(p_, "true", 4) && end_ - p_ >= 4) { p_ += 4; v.kind = JVal::Bool; v.b = true; return v; }
		if (!strncmp(p_, "false", 5) && end_ - p_ >= 5) { p_ += 5; v.kind = JVal::Bool; return v; }
		if (!strncmp(p_, "null", 4) && end_ - p_ >= 4) { p_ += 4; return v; }
		// number: hand the token to strtod (cgltf: CGLTF_ATOF on a copy of the token)
		const char* q = p_;
		while (q < end_ && (strchr("+-.eE", *q) || (*q >= '0' && *q <= '9'))) q++;
		if (q == p_) fail(E_PARSE, "JSON: unexpected character");
		std::string tok(p_, q);
		v.kind = JVal::Num;
		v.num = strtod(tok.c_str(), nullptr);
		p_ = q;
		return v;
	}
	std::string string() {
		expect('"');
		std::string out;
		while (p_ < end_ && *p_ != '"') {
			if (*p_ == '\\' && p_ + 1 < end_) {
				p_++;
				switch (*p_) {
				case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
				case 'b': out += '\b'; break; case 'f': out += '\f'; break;
				case 'u': {  // BMP code point -> UTF-8
					if (end_ - p_ < 5) fail(E_PARSE, "JSON: bad \\u escape");
					unsigned cp = (unsigned)strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
					p_ += 4;
					if (cp < 0x80) out += (char)cp;
					else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
					else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
					break;
				}
				default: out += *p_;
				}
				p_++;
			} else out += *p_++;
		}
		if (p_ >= end_) fail(E_PARSE, "JSON: unterminated string");
		p_++;
		return out;
	}
};

}  // namespace
}  // namespace ptx
