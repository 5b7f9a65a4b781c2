// OpenEXR scanline reader (placeholder): the reference decodes EXR through the OpenEXR library
// (reference src/bitmap.cpp:242-265), which this image does not have.  Until the PIZ/ZIP decoder
// lands, EXR files fail to load, exactly like a reference build without a usable file: the cubemap
// stays "not loaded" and renders black (reference src/environment.cpp:45,64).
#include "image_io.h"

namespace hxh {

bool load_exr(const std::string&, Image& out)
{
	out.reset(0, 0);
	return false;
}

}  // namespace hxh
