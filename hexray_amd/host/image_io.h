// Float-RGB images on the host: BMP codec, gamma decode, bump-map differentiation, sRGB quantiser.
// Behaviour follows the reference Bitmap (reference src/bitmap.cpp:127-240 BMP, :304-338
// differentiate / decompressGamma) and its sRGB table (src/color.h:36-59, src/sdl.cpp:404-419).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace hxh {

struct Image {
	int w = -1, h = -1;
	std::vector<float> rgb;   // 3 floats per texel, row-major, row 0 = top
	bool ok() const { return !rgb.empty(); }
	void reset(int width, int height);
	// Bitmap::getPixel: black outside the image
	void texel(int x, int y, float out[3]) const;
	float intensity(int x, int y) const;   // (r+g+b)/3 of getPixel
};

bool load_bmp(const std::string& path, Image& out);
bool load_exr(const std::string& path, Image& out);       // scanline HALF/FLOAT, NONE/ZIP/ZIPS/PIZ; see exr_reader.cpp
bool load_image(const std::string& path, Image& out);     // by extension, like Bitmap::loadImage
void decode_gamma(Image& img, float gamma);               // Bitmap::decompressGamma
// The 256 values decode_gamma gives the texels q / 255.0f of an 8-bit image (hxr_bitmap::first_gamma_table); false when `img` holds any other value
bool gamma_table(const Image& img, float gamma, float table[256]);
void differentiate(Image& img);                           // Bitmap::differentiate

// 8-bit sRGB through the 4097-entry table, exactly as Color::toRGB32 does
struct SrgbTable {
	uint8_t lut[4097];
	SrgbTable();
	unsigned encode(float x) const
	{
		if (x <= 0) return 0;
		if (x >= 1) return 255;
		return lut[int(x * 4096.0f)];
	}
};
const SrgbTable& srgb_table();

// 24-bit BMP of a float frame (Bitmap::saveBMP, rows bottom-up, padded to 4 bytes)
bool save_bmp(const std::string& path, const float* rgb, int w, int h, size_t row_stride_bytes);
// OpenEXR file of a float frame (Bitmap::saveEXR, src/bitmap.cpp:270-288: RGBA, 16-bit HALF, alpha 1); see exr_writer.cpp
bool save_exr(const std::string& path, const float* rgb, int w, int h, size_t row_stride_bytes);
uint16_t float_to_half(float f);   // round to nearest even, like Imath's half(float)

}  // namespace hxh
