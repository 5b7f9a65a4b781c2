// OpenEXR writer for screenshots of the float frame (the reference's F12 / Bitmap::saveEXR, src/bitmap.cpp:270-288, goes through the
// OpenEXR library: Imf::RgbaOutputFile with WRITE_RGBA, i.e. four 16-bit HALF channels, alpha = 1).  This is a from-scratch writer
// of the same file contents in the simplest valid container: single part, scanline, increasing Y, NO_COMPRESSION, channels A B G R
// in the alphabetical order the format demands.  The library's default for RgbaOutputFile is PIZ compression; the decoded image is the
// same (HALF rounding is the only loss either way), only the file is larger.  exr_reader.cpp reads what this writes (round-trip test).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "image_io.h"

namespace hxh {

uint16_t float_to_half(float f)
{
	uint32_t x;
	memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u;
	const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
	uint32_t man = x & 0x7fffffu;
	if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0u));   // inf / nan
	if (exp >= 31) return (uint16_t)(sign | 0x7c00u);                                                         // overflow -> inf
	if (exp <= 0) {                                                                                            // subnormal half or zero
		if (exp < -10) return (uint16_t)sign;
		man |= 0x800000u;
		const int shift = 14 - exp;
		uint32_t h = man >> shift;
		const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
		if (rem > half || (rem == half && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
	uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
	const uint32_t rem = man & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;   // may carry into the exponent: still the right value (up to inf)
	return (uint16_t)(sign | h);
}

namespace {
struct Out {
	std::vector<uint8_t> b;
	void u8(uint8_t v) { b.push_back(v); }
	void i32(int32_t v) { for (int k = 0; k < 4; k++) b.push_back((uint8_t)((uint32_t)v >> (8 * k))); }
	void u64(uint64_t v) { for (int k = 0; k < 8; k++) b.push_back((uint8_t)(v >> (8 * k))); }
	void f32(float v) { int32_t i; memcpy(&i, &v, 4); i32(i); }
	void str(const char* s) { while (*s) b.push_back((uint8_t)*s++); b.push_back(0); }
	void attr(const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); }
};
}  // namespace

bool save_exr(const std::string& path, const float* rgb, int w, int h, size_t row_stride_bytes)
{
	if (w <= 0 || h <= 0 || !rgb) return false;
	Out o;
	o.i32(20000630);   // magic
	o.i32(2);          // version 2, no flags: single-part scanline
	o.attr("channels", "chlist", 4 * 18 + 1);
	for (const char* c : {"A", "B", "G", "R"}) {
		o.str(c);      // name (2 bytes with the terminator)
		o.i32(1);      // HALF
		o.u8(0); o.u8(0); o.u8(0); o.u8(0);   // pLinear + reserved
		o.i32(1); o.i32(1);                   // x / y sampling
	}
	o.u8(0);
	o.attr("compression", "compression", 1); o.u8(0);   // NO_COMPRESSION
	o.attr("dataWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32(w - 1); o.i32(h - 1);
	o.attr("displayWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32(w - 1); o.i32(h - 1);
	o.attr("lineOrder", "lineOrder", 1); o.u8(0);        // INCREASING_Y
	o.attr("pixelAspectRatio", "float", 4); o.f32(1.0f);
	o.attr("screenWindowCenter", "v2f", 8); o.f32(0.0f); o.f32(0.0f);
	o.attr("screenWindowWidth", "float", 4); o.f32(1.0f);
	o.u8(0);           // end of header
	const size_t table_at = o.b.size(), line_bytes = (size_t)w * 4 * 2, chunk_bytes = 8 + line_bytes;
	for (int y = 0; y < h; y++) o.u64((uint64_t)(table_at + (size_t)h * 8 + (size_t)y * chunk_bytes));
	std::vector<uint16_t> line((size_t)w * 4);
	for (int y = 0; y < h; y++) {
		const float* row = (const float*)((const char*)rgb + (size_t)y * row_stride_bytes);
		for (int x = 0; x < w; x++) {
			line[(size_t)x] = float_to_half(1.0f);                                // A
			line[(size_t)w + x] = float_to_half(row[3 * x + 2]);                  // B
			line[(size_t)2 * w + x] = float_to_half(row[3 * x + 1]);              // G
			line[(size_t)3 * w + x] = float_to_half(row[3 * x + 0]);              // R
		}
		o.i32(y);
		o.i32((int32_t)line_bytes);
		for (uint16_t v : line) { o.u8((uint8_t)v); o.u8((uint8_t)(v >> 8)); }
	}
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) return false;
	const bool ok = fwrite(o.b.data(), 1, o.b.size(), f) == o.b.size();
	return fclose(f) == 0 && ok;
}

}  // namespace hxh
