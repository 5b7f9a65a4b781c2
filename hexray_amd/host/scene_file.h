// .hexray scene-description reader: text -> list of property blocks.
// Behaviour follows the reference parser (reference src/scene.cpp:401-531 for the line grammar,
// :141-240 for the typed getters, :607-654 for the randfloat()/randint() macros), re-implemented
// over std::string.  Typed element construction lives in flatten.cpp.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "linalg.h"

namespace hxh {

struct ParseError : std::runtime_error {
	int line;
	ParseError(int ln, const std::string& what) : std::runtime_error(what), line(ln) {}
};

struct Property {
	int line = 0;
	std::string key, value;
	mutable bool used = false;
};

struct Block {
	std::string cls, name;
	int line_begin = 0, line_end = 0;
	std::vector<Property> props;

	const Property* find(const char* key) const;   // first property with that key, marks it used
	bool get_int(const char* key, int& out, int lo = -2147483647 - 1, int hi = 2147483647) const;
	bool get_bool(const char* key, bool& out) const;
	bool get_float(const char* key, float& out, float lo = -1e17f, float hi = 1e17f) const;
	bool get_double(const char* key, double& out, double lo = -1e120, double hi = 1e120) const;
	bool get_color(const char* key, float out[3], float lo = -1e17f, float hi = 1e17f) const;
	bool get_vec(const char* key, V3& out) const;
	bool get_string(const char* key, std::string& out) const;
	void require(const char* key) const;
	void apply_transform(Xform& T) const;   // every scale / rotate / translate line, in file order
};

// three doubles out of "(a, b, c)"-style text (reference src/scene.cpp:656-666)
void parse_triple(int line, const std::string& text, double& a, double& b, double& c);

std::vector<std::string> split_ws(const std::string& s);

// Parse scene text into blocks.  `overlay` blocks (same grammar) are merged afterwards: a block whose
// class (and name, when given) matches an existing block overrides / appends its properties; an
// unmatched block is appended.  This is how W x H / spp / AA overrides are expressed, since the
// reference reads them only from the scene file (src/scene.cpp:764-778, src/camera.h:57-72).
std::vector<Block> parse_scene_text(const std::string& text, const std::string& origin);
void merge_overlay(std::vector<Block>& blocks, const std::vector<Block>& overlay);

}  // namespace hxh
