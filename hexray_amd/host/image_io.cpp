// BMP codec, gamma decode, differentiate, sRGB table; see image_io.h for the reference citations.
#include "image_io.h"
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace hxh {

void Image::reset(int width, int height)
{
	w = h = -1;
	rgb.clear();
	if (width <= 0 || height <= 0) return;
	w = width;
	h = height;
	rgb.assign(size_t(w) * h * 3, 0.0f);
}

void Image::texel(int x, int y, float out[3]) const
{
	if (rgb.empty() || x < 0 || x >= w || y < 0 || y >= h) { out[0] = out[1] = out[2] = 0.0f; return; }
	const float* p = &rgb[(size_t(y) * w + x) * 3];
	out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}

float Image::intensity(int x, int y) const
{
	float c[3];
	texel(x, y, c);
	return (c[0] + c[1] + c[2]) / 3;
}

namespace {
struct FileCloser {
	FILE* f;
	~FileCloser() { if (f) fclose(f); }
};
uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); }
uint16_t le16(const unsigned char* p) { return uint16_t(p[0] | (p[1] << 8)); }
}  // namespace

bool load_bmp(const std::string& path, Image& out)
{
	out.reset(0, 0);
	FILE* fp = fopen(path.c_str(), "rb");
	if (!fp) { printf("loadBMP: Can't open file: `%s'\n", path.c_str()); return false; }
	FileCloser closer{fp};
	unsigned char head[54];
	if (fread(head, 1, 54, fp) != 54) return false;
	if (le16(head) != 19778) { printf("loadBMP: `%s' is not a BMP file.\n", path.c_str()); return false; }
	int data_offset = int(le32(head + 10));
	int width = int(le32(head + 18)), height = int(le32(head + 22));
	int planes = le16(head + 26), bpp = le16(head + 28);
	int colors_used = int(le32(head + 46));
	if (!(bpp == 8 || bpp == 24 || bpp == 32)) { printf("loadBMP: Cannot handle file format at %d bpp.\n", bpp); return false; }
	if (planes != 1) { printf("loadBMP: cannot load multichannel .bmp!\n"); return false; }
	float palette[256][3];
	int pal_entries = 0;
	if (bpp <= 8) {
		pal_entries = colors_used ? colors_used : (1 << bpp);
		for (int i = 0; i < pal_entries; i++) {
			unsigned char q[4];
			if (fread(q, 1, 4, fp) != 4) return false;
			if (i < 256) {   // Color(unsigned rgb): b = byte0, g = byte1, r = byte2, each / 255
				palette[i][2] = q[0] / 255.0f;
				palette[i][1] = q[1] / 255.0f;
				palette[i][0] = q[2] / 255.0f;
			}
		}
	}
	fseek(fp, data_offset - (54 + pal_entries * 4), SEEK_CUR);
	int k = bpp / 8;
	int row_bytes = width * k;
	if (row_bytes % 4 != 0) row_bytes = (row_bytes / 4 + 1) * 4;
	std::vector<unsigned char> row(row_bytes > 0 ? row_bytes : 1);
	out.reset(width, height);
	if (!out.ok()) { printf("loadBMP: cannot allocate memory for bitmap! Check file integrity!\n"); return false; }
	for (int j = height - 1; j >= 0; j--) {   // stored bottom-up
		if (!fread(row.data(), 1, row_bytes, fp)) {
			printf("loadBMP: short read while opening `%s', file is probably incomplete!\n", path.c_str());
			out.reset(0, 0);
			return false;
		}
		float* dst = &out.rgb[size_t(j) * width * 3];
		for (int i = 0; i < width; i++, dst += 3) {
			if (bpp > 8) {
				dst[0] = row[i * k + 2] / 255.0f;
				dst[1] = row[i * k + 1] / 255.0f;
				dst[2] = row[i * k] / 255.0f;
			} else {
				const float* c = palette[row[i * k]];
				dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2];
			}
		}
	}
	return true;
}

static std::string upper_ext(const std::string& path)
{
	size_t dot = path.rfind('.');
	if (path.size() < 2 || dot == std::string::npos) return "";
	std::string e = path.substr(dot + 1);
	for (char& c : e) c = char(toupper((unsigned char)c));
	return e;
}

bool load_image(const std::string& path, Image& out)
{
	std::string e = upper_ext(path);
	if (e == "BMP") return load_bmp(path, out);
	if (e == "EXR") return load_exr(path, out);
	return false;
}

static float srgb_to_linear(float x)   // decompress_sRGB (src/color.h:49-57)
{
	if (x <= 0) return x;
	if (x >= 1) return x;
	if (x <= 0.04045f) return x / 12.92f;
	return powf((x + 0.055f) / 1.055f, 2.4f);
}

// Bitmap::decompressGamma (src/bitmap.cpp:321-338) over a span of channel values
static void decode_span(float* p, size_t n, bool srgb, float gamma)
{
	for (size_t i = 0; i < n; i++) {
		float& c = p[i];
		if (!(c > 0)) continue;
		c = srgb ? srgb_to_linear(c) : powf(c, gamma);
	}
}

static bool eight_bit(const Image& img)
{
	float key[256];
	for (int q = 0; q < 256; q++) key[q] = q / 255.0f;
	for (float c : img.rgb) {
		// range first: converting a NaN, an infinity or a huge texel of a float source to int is undefined behaviour
		if (!(c >= 0.0f && c <= 1.0f)) return false;
		const int q = (int)(c * 255.0f + 0.5f);
		if (!(q >= 0 && q < 256 && key[q] == c)) return false;
	}
	return true;
}

bool gamma_table(const Image& img, float gamma, float table[256])
{
	if (!eight_bit(img)) return false;
	for (int q = 0; q < 256; q++) table[q] = q / 255.0f;
	decode_span(table, 256, fabsf(gamma - 2.2f) < 1e-6f, gamma);
	return true;
}

void decode_gamma(Image& img, float gamma)
{
	// The texels of an 8-bit image are one of 256 values (q / 255.0f): decode each once, through the same loop (this build's -ffast-math
	// lets the compiler vectorise powf, whose lanes round like each other but not like the scalar function: going through the one loop
	// keeps every texel what the straight pass gave it).  An image with any other value (an EXR source) takes the straight pass.
	float val[256];
	if (!gamma_table(img, gamma, val)) { decode_span(img.rgb.data(), img.rgb.size(), fabsf(gamma - 2.2f) < 1e-6f, gamma); return; }
	for (float& c : img.rgb) c = val[(int)(c * 255.0f + 0.5f)];
}

void differentiate(Image& img)
{
	if (!img.ok()) return;
	std::vector<float> d(img.rgb.size(), 0.0f);
	for (int y = 0; y < img.h; y++)
		for (int x = 0; x < img.w; x++) {
			float here = img.intensity(x, y);
			float* o = &d[(size_t(y) * img.w + x) * 3];
			o[0] = here - img.intensity((x + 1) % img.w, y);
			o[1] = here - img.intensity(x, (y + 1) % img.h);
			o[2] = 0;
		}
	img.rgb.swap(d);
}

static unsigned encode_srgb_exact(float x)   // convertTo8bit_sRGB (src/color.h:36-47); note 12.02, not 12.92
{
	const float a = 0.055f;
	if (x <= 0) return 0;
	if (x >= 1) return 255;
	if (x <= 0.0031308f) x = x * 12.02f;
	else x = (1.0f + a) * powf(x, 1.0f / 2.4f) - a;
	return unsigned(int(floorf(x * 255.0f + 0.5f)));
}

SrgbTable::SrgbTable()
{
	for (int i = 0; i <= 4096; i++) lut[i] = (unsigned char)encode_srgb_exact(i / 4096.0f);
}

const SrgbTable& srgb_table()
{
	static const SrgbTable t;
	return t;
}

bool save_bmp(const std::string& path, const float* rgb, int w, int h, size_t row_stride_bytes)
{
	FILE* fp = fopen(path.c_str(), "wb");
	if (!fp) return false;
	FileCloser closer{fp};
	int row_bytes = w * 3;
	if (row_bytes % 4) row_bytes += 4 - (row_bytes % 4);
	unsigned char head[54];
	memset(head, 0, sizeof head);
	auto put32 = [&](int off, uint32_t v) { head[off] = v & 255; head[off + 1] = (v >> 8) & 255; head[off + 2] = (v >> 16) & 255; head[off + 3] = (v >> 24) & 255; };
	head[0] = 'B'; head[1] = 'M';
	put32(2, uint32_t(row_bytes) * h + 54);
	put32(10, 54);
	put32(14, 40);
	put32(18, uint32_t(w));
	put32(22, uint32_t(h));
	head[26] = 1;
	head[28] = 24;
	fwrite(head, 1, 54, fp);
	const SrgbTable& t = srgb_table();
	std::vector<unsigned char> row(row_bytes, 0);
	for (int y = h - 1; y >= 0; y--) {
		const float* src = (const float*)((const char*)rgb + size_t(y) * row_stride_bytes);
		for (int x = 0; x < w; x++) {
			row[x * 3] = (unsigned char)t.encode(src[x * 3 + 2]);
			row[x * 3 + 1] = (unsigned char)t.encode(src[x * 3 + 1]);
			row[x * 3 + 2] = (unsigned char)t.encode(src[x * 3]);
		}
		fwrite(row.data(), 1, row_bytes, fp);
	}
	return true;
}

}  // namespace hxh
