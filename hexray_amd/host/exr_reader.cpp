// OpenEXR scanline reader for the assets the scenes use (RGBA, HALF or FLOAT channels; NONE, ZIPS, ZIP
// or PIZ compression; single part, no tiles).  The reference decodes EXR through the OpenEXR library
// (reference src/bitmap.cpp:242-265: Imf::RgbaInputFile -> Color r,g,b); that library is not part of
// this image, so the decoder below is written from the published file-format description ("OpenEXR File
// Layout", "Technical Introduction to OpenEXR") and the algorithm descriptions of the PIZ codec (16-bit
// wavelet + range-compacting lookup table + canonical Huffman coding).
//
// PARITY NOTE: the reference's own tests hold no decoded texel values and the OpenEXR library cannot be
// run here, so this decoder is UNPINNED against the reference ("Oracle").  It is validated by
// structure (every chunk decodes to exactly its expected size, Huffman / wavelet streams end exactly at
// their boundaries) and by the decoded images being continuous cube-map faces whose edges match their
// neighbours (tests/test_host.py).  Both the oracle and the GPU path consume the SAME decoded texels.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "image_io.h"

namespace hxh {
namespace {

struct Reader {
	const uint8_t* p;
	const uint8_t* end;
	void need(size_t n) const { if (size_t(end - p) < n) throw std::runtime_error("EXR: truncated file"); }
	uint8_t u8() { need(1); return *p++; }
	uint32_t u32() { need(4); uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); p += 4; return v; }
	int32_t i32() { return int32_t(u32()); }
	uint64_t u64() { uint64_t lo = u32(); uint64_t hi = u32(); return lo | (hi << 32); }
	std::string str() { std::string s; for (;;) { uint8_t c = u8(); if (!c) break; s += char(c); } return s; }
};

struct Channel {
	std::string name;
	int type;      // 0 UINT, 1 HALF, 2 FLOAT
	int xs, ys;
	int bytes() const { return type == 1 ? 2 : 4; }
};

float half_to_float(uint16_t h)
{
	uint32_t sign = uint32_t(h & 0x8000) << 16, exp = (h >> 10) & 31, man = h & 1023, bits;
	if (exp == 0) {
		if (man == 0) bits = sign;
		else {   // subnormal: renormalise
			int e = -1;
			do { man <<= 1; e++; } while (!(man & 1024));
			bits = sign | uint32_t(127 - 15 - e) << 23 | (man & 1023) << 13;
		}
	} else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
	else bits = sign | (exp + 112) << 23 | man << 13;
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

// ---------------------------------------------------------------- zlib "stored / fixed / dynamic" inflate (ZIP, ZIPS blocks)
struct BitIn {
	const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0;
	int bit() { if (!n) { if (p >= end) throw std::runtime_error("EXR: zlib stream truncated"); acc = *p++; n = 8; } int b = acc & 1; acc >>= 1; n--; return b; }
	uint32_t bits(int k) { uint32_t v = 0; for (int i = 0; i < k; i++) v |= uint32_t(bit()) << i; return v; }
};
struct HuffTable {
	std::vector<uint16_t> count = std::vector<uint16_t>(16, 0), symbol;
	void build(const uint8_t* len, int n) {
		std::fill(count.begin(), count.end(), 0);
		for (int i = 0; i < n; i++) count[len[i]]++;
		count[0] = 0;
		uint16_t offs[16]; offs[1] = 0;
		for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
		symbol.assign(n, 0);
		for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = uint16_t(i);
	}
	int decode(BitIn& in) const {
		int code = 0, first = 0, index = 0;
		for (int l = 1; l < 16; l++) {
			code |= in.bit();
			int c = count[l];
			if (code - c < first) return symbol[index + (code - first)];
			index += c; first += c; first <<= 1; code <<= 1;
		}
		throw std::runtime_error("EXR: bad zlib code");
	}
};
void inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
	if (n < 2) throw std::runtime_error("EXR: zlib header");
	BitIn in{src + 2, src + n};
	static const uint16_t lbase[] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
	static const uint16_t lext[] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
	static const uint16_t dbase[] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
	static const uint16_t dext[] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
	out.clear();
	out.reserve(expect);
	for (int last = 0; !last;) {
		last = in.bit();
		int type = int(in.bits(2));
		if (type == 0) {
			in.n = 0;
			if (in.end - in.p < 4) throw std::runtime_error("EXR: zlib stored block");
			size_t len = in.p[0] | (in.p[1] << 8);
			in.p += 4;
			if (size_t(in.end - in.p) < len) throw std::runtime_error("EXR: zlib stored block");
			out.insert(out.end(), in.p, in.p + len);
			in.p += len;
			continue;
		}
		HuffTable lit, dist;
		uint8_t lens[320];
		if (type == 1) {
			int i = 0;
			for (; i < 144; i++) lens[i] = 8;
			for (; i < 256; i++) lens[i] = 9;
			for (; i < 280; i++) lens[i] = 7;
			for (; i < 288; i++) lens[i] = 8;
			lit.build(lens, 288);
			for (i = 0; i < 30; i++) lens[i] = 5;
			dist.build(lens, 30);
		} else if (type == 2) {
			int nlen = int(in.bits(5)) + 257, ndist = int(in.bits(5)) + 1, ncode = int(in.bits(4)) + 4;
			static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
			uint8_t cl[19] = {0};
			for (int i = 0; i < ncode; i++) cl[order[i]] = uint8_t(in.bits(3));
			HuffTable ct;
			ct.build(cl, 19);
			int idx = 0;
			while (idx < nlen + ndist) {
				int sym = ct.decode(in);
				if (sym < 16) lens[idx++] = uint8_t(sym);
				else {
					int rep, val = 0;
					if (sym == 16) { if (!idx) throw std::runtime_error("EXR: zlib lengths"); val = lens[idx - 1]; rep = 3 + int(in.bits(2)); }
					else if (sym == 17) rep = 3 + int(in.bits(3));
					else rep = 11 + int(in.bits(7));
					if (idx + rep > nlen + ndist) throw std::runtime_error("EXR: zlib lengths");
					while (rep--) lens[idx++] = uint8_t(val);
				}
			}
			lit.build(lens, nlen);
			dist.build(lens + nlen, ndist);
		} else throw std::runtime_error("EXR: zlib block type");
		for (;;) {
			int sym = lit.decode(in);
			if (sym < 256) out.push_back(uint8_t(sym));
			else if (sym == 256) break;
			else {
				sym -= 257;
				if (sym >= 29) throw std::runtime_error("EXR: zlib length symbol");
				size_t len = lbase[sym] + in.bits(lext[sym]);
				int ds = dist.decode(in);
				if (ds >= 30) throw std::runtime_error("EXR: zlib distance symbol");
				size_t d = dbase[ds] + in.bits(dext[ds]);
				if (d > out.size()) throw std::runtime_error("EXR: zlib distance");
				for (size_t i = 0; i < len; i++) out.push_back(out[out.size() - d]);
			}
		}
	}
}

// ZIP / ZIPS post-processing: undo the byte predictor, then re-interleave the two halves
void unzip_block(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
	std::vector<uint8_t> tmp;
	inflate_zlib(src, n, tmp, expect);
	if (tmp.size() != expect) throw std::runtime_error("EXR: ZIP block size");
	for (size_t i = 1; i < tmp.size(); i++) tmp[i] = uint8_t(tmp[i - 1] + tmp[i] - 128);
	out.resize(expect);
	size_t half = (expect + 1) / 2;
	for (size_t i = 0, a = 0, b = half; i < expect;) {
		out[i++] = tmp[a++];
		if (i < expect) out[i++] = tmp[b++];
	}
}

// ---------------------------------------------------------------- PIZ
const int HUF_ENCBITS = 16, HUF_DECBITS = 14;
const int HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1, HUF_DECSIZE = 1 << HUF_DECBITS, HUF_DECMASK = HUF_DECSIZE - 1;
const int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;

struct HufDec { int len = 0; std::vector<int> longs; int lit = 0; };

struct HufBits {
	const uint8_t* in; const uint8_t* end; uint64_t c = 0; int lc = 0;
	void get_char() { if (in >= end) throw std::runtime_error("EXR: Huffman stream truncated"); c = (c << 8) | *in++; lc += 8; }
	uint32_t bits(int n) { while (lc < n) get_char(); lc -= n; return uint32_t((c >> lc) & ((1u << n) - 1)); }
};

// code lengths -> canonical codes: entry = length | code << 6
void canonical_codes(std::vector<uint64_t>& hcode)
{
	uint64_t n[59] = {0};
	for (int i = 0; i < HUF_ENCSIZE; i++) n[hcode[i]]++;
	uint64_t c = 0;
	for (int i = 58; i > 0; --i) { uint64_t nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
	for (int i = 0; i < HUF_ENCSIZE; i++) { int l = int(hcode[i]); if (l > 0) hcode[i] = uint64_t(l) | (n[l]++ << 6); }
}

void huf_uncompress(const uint8_t* src, size_t n, uint16_t* raw, size_t n_raw)
{
	if (n == 0) { if (n_raw) throw std::runtime_error("EXR: empty Huffman block"); return; }
	if (n < 20) throw std::runtime_error("EXR: Huffman header");
	auto rd = [&](int o) { return uint32_t(src[o] | (src[o + 1] << 8) | (src[o + 2] << 16) | (uint32_t(src[o + 3]) << 24)); };
	int im = int(rd(0)), iM = int(rd(4));
	uint32_t n_bits = rd(12);
	if (im < 0 || im >= HUF_ENCSIZE || iM < 0 || iM >= HUF_ENCSIZE) throw std::runtime_error("EXR: Huffman symbol range");
	std::vector<uint64_t> hcode(HUF_ENCSIZE, 0);
	HufBits hb{src + 20, src + n};
	for (int i = im; i <= iM; i++) {   // packed 6-bit code lengths with zero-run escapes
		uint32_t l = hb.bits(6);
		hcode[i] = l;
		if (l == uint32_t(LONG_ZEROCODE_RUN)) {
			int run = int(hb.bits(8)) + SHORTEST_LONG_RUN;
			if (i + run > iM + 1) throw std::runtime_error("EXR: Huffman table run");
			while (run--) hcode[i++] = 0;
			i--;
		} else if (l >= uint32_t(SHORT_ZEROCODE_RUN)) {
			int run = int(l) - SHORT_ZEROCODE_RUN + 2;
			if (i + run > iM + 1) throw std::runtime_error("EXR: Huffman table run");
			while (run--) hcode[i++] = 0;
			i--;
		}
	}
	const uint8_t* data = hb.in;   // the table is byte aligned at its end
	canonical_codes(hcode);
	if (size_t(src + n - data) < (size_t(n_bits) + 7) / 8) throw std::runtime_error("EXR: Huffman data truncated");

	std::vector<HufDec> dec(HUF_DECSIZE);
	for (int s = im; s <= iM; s++) {
		uint64_t c = hcode[s] >> 6;
		int l = int(hcode[s] & 63);
		if (c >> l) throw std::runtime_error("EXR: Huffman code longer than its length");
		if (l > HUF_DECBITS) {
			HufDec& pl = dec[c >> (l - HUF_DECBITS)];
			if (pl.len) throw std::runtime_error("EXR: Huffman long code clashes with a short one");
			pl.longs.push_back(s);
		} else if (l) {
			size_t base = size_t(c << (HUF_DECBITS - l));
			for (size_t i = 0; i < (size_t(1) << (HUF_DECBITS - l)); i++) {
				HufDec& pl = dec[base + i];
				if (pl.len || !pl.longs.empty()) throw std::runtime_error("EXR: Huffman short code clash");
				pl.len = l;
				pl.lit = s;
			}
		}
	}
	const int rlc = iM;   // the run-length symbol
	uint16_t* out = raw;
	uint16_t* oe = raw + n_raw;
	HufBits d{data, data + (size_t(n_bits) + 7) / 8};
	auto emit = [&](int sym) {
		if (sym == rlc) {
			if (d.lc < 8) d.get_char();
			d.lc -= 8;
			unsigned cs = unsigned((d.c >> d.lc) & 255);
			if (out == raw || out + cs > oe) throw std::runtime_error("EXR: Huffman run out of range");
			uint16_t s = out[-1];
			while (cs--) *out++ = s;
		} else if (out < oe) *out++ = uint16_t(sym);
		else throw std::runtime_error("EXR: Huffman output overflow");
	};
	while (d.in < d.end) {
		d.get_char();
		while (d.lc >= HUF_DECBITS) {
			const HufDec& pl = dec[(d.c >> (d.lc - HUF_DECBITS)) & HUF_DECMASK];
			if (pl.len) { d.lc -= pl.len; emit(pl.lit); }
			else {
				if (pl.longs.empty()) throw std::runtime_error("EXR: Huffman code not in table");
				size_t j = 0;
				for (; j < pl.longs.size(); j++) {
					int l = int(hcode[pl.longs[j]] & 63);
					while (d.lc < l && d.in < d.end) d.get_char();
					if (d.lc >= l && (hcode[pl.longs[j]] >> 6) == ((d.c >> (d.lc - l)) & ((uint64_t(1) << l) - 1))) {
						d.lc -= l;
						emit(pl.longs[j]);
						break;
					}
				}
				if (j == pl.longs.size()) throw std::runtime_error("EXR: Huffman long code not found");
			}
		}
	}
	int i = (8 - int(n_bits)) & 7;   // padding bits of the last byte
	d.c >>= i;
	d.lc -= i;
	while (d.lc > 0) {
		const HufDec& pl = dec[(d.c << (HUF_DECBITS - d.lc)) & HUF_DECMASK];
		if (!pl.len) throw std::runtime_error("EXR: Huffman tail");
		d.lc -= pl.len;
		emit(pl.lit);
	}
	if (out != oe) throw std::runtime_error("EXR: Huffman block decoded to the wrong size");
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
	int16_t ls = int16_t(l), hs = int16_t(h);
	int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
	a = uint16_t(int16_t(ai));
	b = uint16_t(int16_t(ai - hi));
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
	int m = l, d = h;
	int bb = (m - (d >> 1)) & 0xffff;
	int aa = (d + bb - 0x8000) & 0xffff;
	b = uint16_t(bb);
	a = uint16_t(aa);
}

// inverse 2-D Haar-like wavelet over one channel plane (nx x ny samples, strides ox / oy in uint16 units)
void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx)
{
	const bool w14 = mx < (1 << 14);
	int n = nx > ny ? ny : nx, p = 1, p2;
	while (p <= n) p <<= 1;
	p >>= 1;
	p2 = p;
	p >>= 1;
	auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
	while (p >= 1) {
		uint16_t* py = in;
		uint16_t* ey = in + ptrdiff_t(oy) * (ny - p2);
		ptrdiff_t oy1 = ptrdiff_t(oy) * p, oy2 = ptrdiff_t(oy) * p2, ox1 = ptrdiff_t(ox) * p, ox2 = ptrdiff_t(ox) * p2;
		uint16_t i00, i01, i10, i11;
		for (; py <= ey; py += oy2) {
			uint16_t* px = py;
			uint16_t* ex = py + ptrdiff_t(ox) * (nx - p2);
			for (; px <= ex; px += ox2) {
				uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
				dec(*px, *p10, i00, i10);
				dec(*p01, *p11, i01, i11);
				dec(i00, i01, *px, *p01);
				dec(i10, i11, *p10, *p11);
			}
			if (nx & p) {
				uint16_t* p10 = px + oy1;
				dec(*px, *p10, i00, *p10);
				*px = i00;
			}
		}
		if (ny & p) {
			uint16_t* px = py;
			uint16_t* ex = py + ptrdiff_t(ox) * (nx - p2);
			for (; px <= ex; px += ox2) {
				uint16_t* p01 = px + ox1;
				dec(*px, *p01, i00, *p01);
				*px = i00;
			}
		}
		p2 = p;
		p >>= 1;
	}
}

void unpiz_block(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect, const std::vector<Channel>& chans, int width, int lines)
{
	Reader r{src, src + n};
	std::vector<uint8_t> bitmap(8192, 0);
	r.need(4);
	int min_nz = r.p[0] | (r.p[1] << 8), max_nz = r.p[2] | (r.p[3] << 8);
	r.p += 4;
	if (max_nz >= 8192) throw std::runtime_error("EXR: PIZ bitmap range");
	if (min_nz <= max_nz) {
		r.need(size_t(max_nz - min_nz + 1));
		memcpy(&bitmap[min_nz], r.p, size_t(max_nz - min_nz + 1));
		r.p += max_nz - min_nz + 1;
	}
	std::vector<uint16_t> lut(65536, 0);
	int k = 0;
	for (int i = 0; i < 65536; i++)
		if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = uint16_t(i);
	uint16_t max_value = uint16_t(k - 1);
	int length = r.i32();
	if (length < 0 || size_t(length) > size_t(r.end - r.p)) throw std::runtime_error("EXR: PIZ Huffman length");
	size_t n_short = expect / 2;
	std::vector<uint16_t> tmp(n_short);
	huf_uncompress(r.p, size_t(length), tmp.data(), n_short);
	// per channel: all lines of the block are contiguous in tmp
	std::vector<uint16_t*> start(chans.size());
	uint16_t* cur = tmp.data();
	for (size_t c = 0; c < chans.size(); c++) {
		int size = chans[c].bytes() / 2;
		start[c] = cur;
		for (int j = 0; j < size; j++) wav2_decode(cur + j, width, size, lines, width * size, max_value);
		cur += size_t(width) * lines * size;
	}
	for (uint16_t& v : tmp) v = lut[v];
	out.resize(expect);
	uint16_t* dst = reinterpret_cast<uint16_t*>(out.data());
	for (int y = 0; y < lines; y++)
		for (size_t c = 0; c < chans.size(); c++) {
			size_t cnt = size_t(width) * (chans[c].bytes() / 2);
			memcpy(dst, start[c], cnt * 2);
			dst += cnt;
			start[c] += cnt;
		}
}

}  // namespace

bool load_exr(const std::string& path, Image& out)
{
	out.reset(0, 0);
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	std::vector<uint8_t> file;
	uint8_t buf[65536];
	for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) file.insert(file.end(), buf, buf + n);
	fclose(f);
	try {
		Reader r{file.data(), file.data() + file.size()};
		if (r.u32() != 20000630u) throw std::runtime_error("EXR: bad magic");
		uint32_t version = r.u32();
		if ((version & 0xff) != 2 || (version & 0x1e00)) throw std::runtime_error("EXR: tiled / multi-part / deep files are not supported");
		std::vector<Channel> chans;
		int compression = -1, line_order = 0;
		int dw[4] = {0, 0, -1, -1};
		for (;;) {
			std::string name = r.str();
			if (name.empty()) break;
			std::string type = r.str();
			int size = r.i32();
			if (size < 0) throw std::runtime_error("EXR: attribute size");
			r.need(size_t(size));
			Reader a{r.p, r.p + size};
			r.p += size;
			if (name == "channels") {
				for (;;) {
					Channel c;
					c.name = a.str();
					if (c.name.empty()) break;
					c.type = a.i32();
					a.u32();   // pLinear + reserved
					c.xs = a.i32();
					c.ys = a.i32();
					if (c.type != 1 && c.type != 2) throw std::runtime_error("EXR: UINT channels are not supported");
					if (c.xs != 1 || c.ys != 1) throw std::runtime_error("EXR: subsampled channels are not supported");
					chans.push_back(c);
				}
			} else if (name == "compression") compression = a.u8();
			else if (name == "dataWindow") for (int& v : dw) v = a.i32();
			else if (name == "lineOrder") line_order = a.u8();
		}
		(void)line_order;   // chunks carry their y; the offset table gives random access either way
		int width = dw[2] - dw[0] + 1, height = dw[3] - dw[1] + 1;
		if (width <= 0 || height <= 0 || width > 16384 || height > 16384 || chans.empty()) throw std::runtime_error("EXR: bad header");
		int lines_per_chunk;
		switch (compression) {
			case 0: case 2: lines_per_chunk = 1; break;   // NONE, ZIPS
			case 3: lines_per_chunk = 16; break;          // ZIP
			case 4: lines_per_chunk = 32; break;          // PIZ
			default: throw std::runtime_error("EXR: compression method " + std::to_string(compression) + " is not supported");
		}
		size_t line_bytes = 0;
		for (const Channel& c : chans) line_bytes += size_t(width) * c.bytes();
		int n_chunks = (height + lines_per_chunk - 1) / lines_per_chunk;
		std::vector<uint64_t> offsets(n_chunks);
		for (uint64_t& o : offsets) o = r.u64();
		int ci[3] = {-1, -1, -1};   // R, G, B positions in the (alphabetical) channel list
		for (size_t c = 0; c < chans.size(); c++) {
			if (chans[c].name == "R") ci[0] = int(c);
			else if (chans[c].name == "G") ci[1] = int(c);
			else if (chans[c].name == "B") ci[2] = int(c);
		}
		if (ci[0] < 0 && ci[1] < 0 && ci[2] < 0) {   // luminance-only files: Y feeds all three
			for (size_t c = 0; c < chans.size(); c++) if (chans[c].name == "Y") ci[0] = ci[1] = ci[2] = int(c);
		}
		out.reset(width, height);
		std::vector<uint8_t> block;
		for (int k = 0; k < n_chunks; k++) {
			if (offsets[k] + 8 > file.size()) throw std::runtime_error("EXR: chunk offset");
			Reader c{file.data() + offsets[k], file.data() + file.size()};
			int y0 = c.i32() - dw[1];
			int data_size = c.i32();
			if (y0 < 0 || y0 >= height || data_size < 0) throw std::runtime_error("EXR: chunk header");
			c.need(size_t(data_size));
			int lines = std::min(lines_per_chunk, height - y0);
			size_t expect = line_bytes * lines;
			const uint8_t* data = c.p;
			if (size_t(data_size) == expect || compression == 0) {
				if (size_t(data_size) < expect) throw std::runtime_error("EXR: raw chunk too small");
				block.assign(data, data + expect);
			} else if (compression == 4) unpiz_block(data, size_t(data_size), block, expect, chans, width, lines);
			else unzip_block(data, size_t(data_size), block, expect);
			for (int y = 0; y < lines; y++) {
				const uint8_t* row = block.data() + size_t(y) * line_bytes;
				std::vector<size_t> ch_off(chans.size());
				size_t o = 0;
				for (size_t cc = 0; cc < chans.size(); cc++) { ch_off[cc] = o; o += size_t(width) * chans[cc].bytes(); }
				float* dst = &out.rgb[size_t(y0 + y) * width * 3];
				for (int x = 0; x < width; x++)
					for (int q = 0; q < 3; q++) {
						float v = 0;
						if (ci[q] >= 0) {
							const Channel& ch = chans[ci[q]];
							const uint8_t* s = row + ch_off[ci[q]] + size_t(x) * ch.bytes();
							if (ch.type == 1) v = half_to_float(uint16_t(s[0] | (s[1] << 8)));
							else memcpy(&v, s, 4);
						}
						dst[x * 3 + q] = v;
					}
			}
		}
		return true;
	} catch (const std::exception& e) {
		fprintf(stderr, "%s: %s\n", path.c_str(), e.what());
		out.reset(0, 0);
		return false;
	}
}

}  // namespace hxh
