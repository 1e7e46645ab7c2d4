// Minimal PNG reader for glTF textures (host). The reference loads images with stb_image (vendored there, not used
// here): image::image::load, LIB/image/image.cpp:23-54, req_comp = 0, i.e. the file's own channel count, 8 bits per
// channel. This reader produces the same pixels for PNGs (Adam7-interlaced ones included): grey / grey+alpha / RGB / RGBA / palette,
// 1-16 bits, tRNS; 16-bit samples keep their high byte, sub-byte grey is scaled to 0..255, palette is expanded to
// RGB (RGBA with tRNS) — the conversions stb_image applies. Inflate is zlib's.
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "flat_scene.hpp"

namespace ptx {
namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
	int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
[[noreturn]] void bad(const std::string& path, const char* why) { throw Error{3, "PNG '" + path + "': " + why}; }
}  // namespace

void read_png(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw Error{2, "Failed to load image to memory: " + path};   // image.cpp:44-45
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (file.size() < 8 || memcmp(file.data(), sig, 8)) bad(path, "not a PNG (only PNG textures are supported)");
	uint32_t depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, plte, trns;
	bool have_hdr = false;
	for (size_t p = 8; p + 12 <= file.size();) {
		uint32_t len = be32(&file[p]);
		const uint8_t* type = &file[p + 4];
		const uint8_t* data = &file[p + 8];
		if (p + 12 + (size_t)len > file.size()) bad(path, "truncated chunk");
		if (!memcmp(type, "IHDR", 4)) {
			if (len != 13) bad(path, "bad IHDR");
			W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
			have_hdr = true;
		} else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
		else if (!memcmp(type, "IEND", 4)) break;
		p += 12 + (size_t)len;
	}
	if (!have_hdr || !W || !H) bad(path, "missing IHDR");
	if (W > 65536u || H > 65536u) bad(path, "image dimensions out of range");   // before any allocation sized by them
	if (interlace > 1) bad(path, "unknown interlace method");
	uint32_t src_ch;
	switch (ctype) {
	case 0: src_ch = 1; break; case 2: src_ch = 3; break; case 3: src_ch = 1; break; case 4: src_ch = 2; break; case 6: src_ch = 4; break;
	default: bad(path, "unknown colour type");
	}
	if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) bad(path, "bad bit depth");
	const size_t bpp_bits = (size_t)src_ch * depth, fb = std::max<size_t>(1, bpp_bits / 8);
	auto row_bytes = [&](uint32_t w) { return (w * bpp_bits + 7) / 8; };
	// the image is one pass, or the seven Adam7 passes (each a reduced image of the pixels x0 + i dx, y0 + j dy), one after the other
	struct Pass { uint32_t x0, y0, dx, dy, w, h; };
	std::vector<Pass> passes;
	if (!interlace) passes.push_back({0, 0, 1, 1, W, H});
	else {
		static const uint32_t xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dxs[7] = {8, 8, 4, 4, 2, 2, 1}, dys[7] = {8, 8, 8, 4, 4, 2, 2};
		for (int k = 0; k < 7; k++) {
			const uint32_t w = (W - xs[k] + dxs[k] - 1) / dxs[k], h = (H - ys[k] + dys[k] - 1) / dys[k];
			if (W > xs[k] && H > ys[k] && w && h) passes.push_back({xs[k], ys[k], dxs[k], dys[k], w, h});
		}
	}
	size_t raw_size = 0;
	for (const Pass& ps : passes) raw_size += (row_bytes(ps.w) + 1) * ps.h;
	// a header cannot demand more memory than the file could possibly fill: deflate expands at most 1032 : 1 (IHDR allows 65536 x 65536
	// at 64 bits per pixel — 32 GiB zero-filled before inflate would fail), and the readers cap an image at 2^28 pixels
	if ((uint64_t)W * H > ((uint64_t)1 << 28) || raw_size > idat.size() * 1032 + 64) bad(path, "image data too short for the declared size");
	std::vector<uint8_t> raw(raw_size);
	uLongf raw_len = (uLongf)raw.size();
	if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) bad(path, "inflate failed");
	const bool has_trns = !trns.empty();
	C = ctype == 3 ? (has_trns ? 4u : 3u) : src_ch + ((has_trns && (ctype == 0 || ctype == 2)) ? 1u : 0u);
	out.assign((size_t)W * H * C, 255);
	auto sample16 = [&](const uint8_t* cur, size_t idx) -> uint32_t {   // idx-th sample of the row, any depth
		if (depth == 8) return cur[idx];
		if (depth == 16) return ((uint32_t)cur[2 * idx] << 8) | cur[2 * idx + 1];
		const size_t bit = idx * depth;
		return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
	};
	const uint32_t scale = depth < 8 ? 255u / ((1u << depth) - 1u) : 1u;
	size_t at = 0;
	for (const Pass& ps : passes) {
		const size_t stride = row_bytes(ps.w);
		// unfilter in place (rows keep their filter byte)
		std::vector<uint8_t> prev(stride, 0);
		for (uint32_t y = 0; y < ps.h; y++) {
			uint8_t* row = &raw[at + (stride + 1) * y];
			const uint8_t ft = row[0];
			uint8_t* cur = row + 1;
			for (size_t i = 0; i < stride; i++) {
				const int a = i >= fb ? cur[i - fb] : 0, b = prev[i], c = i >= fb ? prev[i - fb] : 0;
				int v = cur[i];
				switch (ft) {
				case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
				default: bad(path, "bad filter type");
				}
				cur[i] = (uint8_t)v;
			}
			memcpy(prev.data(), cur, stride);
		}
		for (uint32_t y = 0; y < ps.h; y++) {
			const uint8_t* cur = &raw[at + (stride + 1) * y + 1];
			uint8_t* drow = &out[(size_t)(ps.y0 + y * ps.dy) * W * C];
			for (uint32_t x = 0; x < ps.w; x++) {
				uint8_t* dst = drow + (size_t)(ps.x0 + x * ps.dx) * C;
				if (ctype == 3) {
					const uint32_t k = sample16(cur, x);
					if (3 * k + 2 >= plte.size()) bad(path, "palette index out of range");
					dst[0] = plte[3 * k]; dst[1] = plte[3 * k + 1]; dst[2] = plte[3 * k + 2];
					if (has_trns) dst[3] = k < trns.size() ? trns[k] : 255;
					continue;
				}
				uint32_t s[4] = {0, 0, 0, 0};
				for (uint32_t c = 0; c < src_ch; c++) s[c] = sample16(cur, (size_t)x * src_ch + c);
				for (uint32_t c = 0; c < src_ch; c++) dst[c] = depth == 16 ? (uint8_t)(s[c] >> 8) : (uint8_t)(s[c] * scale);
				if (has_trns && ctype == 0 && trns.size() >= 2) dst[1] = s[0] == (((uint32_t)trns[0] << 8) | trns[1]) ? 0 : 255;
				if (has_trns && ctype == 2 && trns.size() >= 6) {
					const bool eq = s[0] == (((uint32_t)trns[0] << 8) | trns[1]) && s[1] == (((uint32_t)trns[2] << 8) | trns[3]) && s[2] == (((uint32_t)trns[4] << 8) | trns[5]);
					dst[3] = eq ? 0 : 255;
				}
			}
		}
		at += (stride + 1) * ps.h;
	}
}

}  // namespace ptx
