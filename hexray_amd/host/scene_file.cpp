// .hexray reader; see scene_file.h.  Grammar notes cite the reference parser they mirror.
#include "scene_file.h"
#include <cctype>
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>

namespace hxh {

std::vector<std::string> split_ws(const std::string& s)
{
	std::vector<std::string> out;
	size_t i = 0, n = s.size();
	while (i < n) {
		while (i < n && isspace((unsigned char)s[i])) i++;
		if (i >= n) break;
		size_t j = i;
		while (j < n && !isspace((unsigned char)s[j])) j++;
		out.emplace_back(s.substr(i, j - i));
		i = j;
	}
	return out;
}

static std::string trimmed(const std::string& s)
{
	size_t b = 0, e = s.size();
	while (e > b && isspace((unsigned char)s[e - 1])) e--;
	while (b < e && isspace((unsigned char)s[b])) b++;
	return s.substr(b, e - b);
}

static std::string unbrace(std::string s)   // '(' ')' ',' -> ' '   (src/scene.cpp:190-197)
{
	for (char& c : s)
		if (c == ',' || c == '(' || c == ')') c = ' ';
	return s;
}

void parse_triple(int line, const std::string& text, double& a, double& b, double& c)
{
	std::string t = unbrace(text);
	if (3 != sscanf(t.c_str(), "%lf%lf%lf", &a, &b, &c)) throw ParseError(line, "Expected three double values");
}

// randfloat(a,b) / randint(a,b) are replaced in place, left-justified in the blanked-out macro text,
// by draws from ONE default-seeded mt19937 shared by the whole parse (src/scene.cpp:607-654).
static void expand_random_macros(int line_no, std::string& s, std::mt19937& gen)
{
	for (int pass = 0; pass < 2; pass++) {
		const char* word = pass == 0 ? "randfloat" : "randint";
		size_t p;
		while ((p = s.find(word)) != std::string::npos) {
			size_t open = s.find('(', p);
			size_t close = open == std::string::npos ? std::string::npos : s.find(')', open);
			if (open == std::string::npos || close == std::string::npos)
				throw ParseError(line_no, std::string(word) + " in inexpected format");
			std::string args = s.substr(open + 1, close - open - 1);
			char text[40];
			if (pass == 0) {
				float lo, hi;
				if (2 != sscanf(args.c_str(), "%f,%f", &lo, &hi)) throw ParseError(line_no, "bad randfloat format (expected: randfloat(<min>, <max>))");
				if (lo > hi) throw ParseError(line_no, "bad randfloat format (min > max)");
				std::uniform_real_distribution<float> d(lo, hi);
				snprintf(text, sizeof text, "%.5f", d(gen));
			} else {
				int lo, hi;
				if (2 != sscanf(args.c_str(), "%d,%d", &lo, &hi)) throw ParseError(line_no, "bad randint format (expected: randint(<min>, <max>))");
				if (lo > hi) throw ParseError(line_no, "bad randint format (min > max)");
				std::uniform_int_distribution<int> d(lo, hi);
				snprintf(text, sizeof text, "%d", d(gen));
			}
			size_t span = close - p + 1, tl = strlen(text);
			if (tl >= span) throw ParseError(line_no, "random macro result does not fit");
			for (size_t k = 0; k < span; k++) s[p + k] = k < tl ? text[k] : ' ';
		}
	}
}

std::vector<Block> parse_scene_text(const std::string& text, const std::string& origin)
{
	std::vector<Block> blocks;
	std::mt19937 macro_rng;   // default seed
	std::istringstream in(text);
	std::string raw;
	int line_no = 0;
	bool in_comment = false, in_block = false;
	while (std::getline(in, raw)) {
		line_no++;
		if (in_comment) {   // closes only on a line that STARTS with "*/" (src/scene.cpp:428-431)
			if (raw.size() >= 2 && raw[0] == '*' && raw[1] == '/') in_comment = false;
			continue;
		}
		size_t cut = raw.find("//");
		size_t hash = raw.find('#');
		if (hash != std::string::npos && (cut == std::string::npos || hash < cut)) cut = hash;
		std::string line = trimmed(cut == std::string::npos ? raw : raw.substr(0, cut));
		if (line.empty()) continue;
		if (line.size() >= 2 && line[0] == '/' && line[1] == '*') {   // opens only at line start (:442-445)
			in_comment = true;
			continue;
		}
		expand_random_macros(line_no, line, macro_rng);
		std::vector<std::string> tok = split_ws(line);
		if (tok.empty()) continue;
		if (!in_block) {
			if (tok.size() == 1)
				throw ParseError(line_no, origin + ": unexpected token `" + tok[0] + "'");
			if (tok.size() > 3 || tok.back() != "{")
				throw ParseError(line_no, origin + ": an object definition should end with a `{'");
			Block b;
			b.cls = tok[0];
			b.name = tok[1];   // "{" for singletons, exactly like the reference (:474-476,484)
			b.line_begin = line_no;
			blocks.push_back(b);
			in_block = true;
		} else if (tok.size() == 1) {
			if (tok[0] != "}") throw ParseError(line_no, origin + ": unexpected token in object definition: `" + tok[0] + "'");
			blocks.back().line_end = line_no;
			in_block = false;
		} else {
			size_t i = tok[0].size();
			while (i < line.size() && isspace((unsigned char)line[i])) i++;
			size_t last = line.size() - 1;
			std::string value;
			if (i < last && line[i] == '"' && line[last] == '"') value = line.substr(i + 1, last - i - 1);
			else value = line.substr(i);
			Property p;
			p.line = line_no;
			p.key = tok[0].substr(0, 127);
			p.value = value.substr(0, 255);
			blocks.back().props.push_back(p);
		}
	}
	if (in_block) throw ParseError(line_no, origin + ": unfinished object definition at EOF");
	return blocks;
}

static bool appends(const std::string& key)
{
	return key == "scale" || key == "rotate" || key == "translate" || key == "layer";
}

void merge_overlay(std::vector<Block>& blocks, const std::vector<Block>& overlay)
{
	for (const Block& o : overlay) {
		Block* target = nullptr;
		for (Block& b : blocks)
			if (b.cls == o.cls && (o.name == "{" || b.name == o.name)) { target = &b; break; }
		if (!target) { blocks.push_back(o); continue; }
		for (const Property& p : o.props) {
			Property* hit = nullptr;
			if (!appends(p.key))
				for (Property& q : target->props)
					if (q.key == p.key) { hit = &q; break; }
			if (hit) hit->value = p.value;
			else target->props.push_back(p);
		}
	}
}

const Property* Block::find(const char* key) const
{
	for (const Property& p : props)
		if (p.key == key) { p.used = true; return &p; }
	return nullptr;
}

bool Block::get_int(const char* key, int& out, int lo, int hi) const
{
	const Property* p = find(key);
	if (!p) return false;
	int x;
	if (1 != sscanf(p->value.c_str(), "%d", &x)) throw ParseError(p->line, "Invalid integer");
	if (x < lo || x > hi) throw ParseError(p->line, "Value outside the allowed bounds");
	out = x;
	return true;
}

bool Block::get_bool(const char* key, bool& out) const
{
	const Property* p = find(key);
	if (!p) return false;
	out = !(p->value == "off" || p->value == "false" || p->value == "0");
	return true;
}

bool Block::get_float(const char* key, float& out, float lo, float hi) const
{
	const Property* p = find(key);
	if (!p) return false;
	float x;
	if (1 != sscanf(p->value.c_str(), "%f", &x)) throw ParseError(p->line, "Invalid float");
	if (x < lo || x > hi) throw ParseError(p->line, "Value outside the allowed bounds");
	out = x;
	return true;
}

bool Block::get_double(const char* key, double& out, double lo, double hi) const
{
	const Property* p = find(key);
	if (!p) return false;
	double x;
	if (1 != sscanf(p->value.c_str(), "%lf", &x)) throw ParseError(p->line, "Invalid double");
	if (x < lo || x > hi) throw ParseError(p->line, "Value outside the allowed bounds");
	out = x;
	return true;
}

bool Block::get_color(const char* key, float out[3], float lo, float hi) const
{
	const Property* p = find(key);
	if (!p) return false;
	std::string t = unbrace(p->value);
	float c[3];
	if (3 != sscanf(t.c_str(), "%f%f%f", &c[0], &c[1], &c[2])) throw ParseError(p->line, "Invalid color");
	for (int i = 0; i < 3; i++)
		if (c[i] < lo || c[i] > hi) throw ParseError(p->line, "Color value outside the allowed bounds");
	out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
	return true;
}

bool Block::get_vec(const char* key, V3& out) const
{
	const Property* p = find(key);
	if (!p) return false;
	std::string t = unbrace(p->value);
	V3 v;
	if (3 != sscanf(t.c_str(), "%lf%lf%lf", &v.x, &v.y, &v.z)) throw ParseError(p->line, "Invalid vector");
	out = v;
	return true;
}

bool Block::get_string(const char* key, std::string& out) const
{
	const Property* p = find(key);
	if (!p) return false;
	out = p->value;
	return true;
}

void Block::require(const char* key) const
{
	if (!find(key)) throw ParseError(line_end, std::string("Required property `") + key + "' not defined");
}

void Block::apply_transform(Xform& T) const
{
	for (const Property& p : props) {
		double a, b, c;
		if (p.key == "scale") { p.used = true; parse_triple(p.line, p.value, a, b, c); T.scale(a, b, c); }
		else if (p.key == "rotate") { p.used = true; parse_triple(p.line, p.value, a, b, c); T.rotate(a, b, c); }
		else if (p.key == "translate") { p.used = true; parse_triple(p.line, p.value, a, b, c); T.translate(V3(a, b, c)); }
	}
}

}  // namespace hxh
