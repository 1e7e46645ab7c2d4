// Radiance .hdr (RGBE) reader (host). The reference loads HDR images with stb_image's stbi_loadf (image::image::load,
// LIB/image/image.cpp:23-54: `hdr = stbi_is_hdr(path)`, req_comp = 0) and keeps the floats (image::read returns them as they are,
// :124-141). This reader produces the same floats: header "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w" orientation
// only (what stb accepts), new-style run-length scanlines or flat RGBE quadruples, and per pixel
//     rgb = byte * ldexp(1.0f, e - (128 + 8))   (exact in binary32),   black when e == 0;   three channels.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "flat_scene.hpp"

namespace ptx {
namespace {
[[noreturn]] void bad(const std::string& path, const char* why, int code = 3) { throw Error{code, "HDR '" + path + "': " + why}; }
}

bool is_hdr_file(const std::string& path) {
	std::ifstream f(path, std::ios::binary);
	char sig[12] = {0};
	f.read(sig, 11);
	return !memcmp(sig, "#?RADIANCE\n", 11) || !memcmp(sig, "#?RGBE\n", 7);
}

void read_hdr(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<float>& out) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw Error{2, "Failed to load image to memory: " + path};
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	size_t p = 0;
	auto token = [&]() {   // one header line (without its newline), at most 1023 characters as stb reads it
		std::string t;
		while (p < file.size() && file[p] != '\n') {
			if (t.size() < 1023) t.push_back((char)file[p]);
			p++;
		}
		if (p < file.size()) p++;
		return t;
	};
	const std::string magic = token();
	if (magic != "#?RADIANCE" && magic != "#?RGBE") bad(path, "not a Radiance HDR file");
	bool valid = false;
	for (;;) {
		const std::string t = token();
		if (t.empty()) break;
		if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
		if (p >= file.size()) break;
	}
	if (!valid) bad(path, "unsupported format (FORMAT=32-bit_rle_rgbe expected)", 7);
	const std::string dim = token();
	if (dim.compare(0, 3, "-Y ") != 0) bad(path, "unsupported data layout", 7);
	char* endp = nullptr;
	const long h = strtol(dim.c_str() + 3, &endp, 10);
	while (*endp == ' ') endp++;
	if (strncmp(endp, "+X ", 3) != 0) bad(path, "unsupported data layout", 7);
	const long w = strtol(endp + 3, nullptr, 10);
	if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || (uint64_t)w * (uint64_t)h > (1ull << 28)) bad(path, "bad image size");
	W = (uint32_t)w; H = (uint32_t)h; C = 3;
	{   // the header alone must not be able to demand gigabytes: even the best run-length coding spends 4 + 4 * 2 * ceil(W / 127) bytes
		// per scanline (flat data: 4 per pixel)
		const uint64_t per_line = (W < 8 || W >= 32768) ? (uint64_t)W * 4 : 4 + 8 * (((uint64_t)W + 126) / 127);
		if ((uint64_t)(file.size() - p) < std::min<uint64_t>(per_line, (uint64_t)W * 4) * H) bad(path, "truncated file");
	}
	out.assign((size_t)W * H * 3, 0.f);
	auto get8 = [&]() -> int { return p < file.size() ? file[p++] : 0; };
	auto convert = [&](float* o, const uint8_t* in) {
		if (in[3] != 0) {
			const float f1 = (float)std::ldexp(1.0f, (int)in[3] - (128 + 8));
			o[0] = in[0] * f1; o[1] = in[1] * f1; o[2] = in[2] * f1;
		} else o[0] = o[1] = o[2] = 0.f;
	};
	auto flat_from = [&](size_t first_pixel) {   // the rest of the image as plain RGBE quadruples
		for (size_t i = first_pixel; i < (size_t)W * H; i++) {
			uint8_t rgbe[4];
			for (int k = 0; k < 4; k++) rgbe[k] = (uint8_t)get8();
			convert(&out[3 * i], rgbe);
		}
	};
	if (W < 8 || W >= 32768) { flat_from(0); return; }
	std::vector<uint8_t> scan((size_t)W * 4);
	for (uint32_t j = 0; j < H; j++) {
		const int c1 = get8(), c2 = get8(), len_hi = get8();
		if (c1 != 2 || c2 != 2 || (len_hi & 0x80)) {
			// not run-length encoded: these bytes ARE the first pixel, and everything after them is flat (only legal on the first scanline)
			if (j != 0) bad(path, "corrupt: mixed scanline encodings");
			const uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len_hi, (uint8_t)get8()};
			convert(&out[0], rgbe);
			flat_from(1);
			return;
		}
		const int len = (len_hi << 8) | get8();
		if ((uint32_t)len != W) bad(path, "corrupt: invalid decoded scanline length");
		for (int k = 0; k < 4; k++) {
			uint32_t i = 0;
			while (i < W) {
				const uint32_t nleft = W - i;
				int count = get8();
				if (count > 128) {   // a run
					const uint8_t value = (uint8_t)get8();
					count -= 128;
					if (count == 0 || (uint32_t)count > nleft) bad(path, "corrupt: bad RLE data");
					for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = value;
				} else {             // a dump
					if (count == 0 || (uint32_t)count > nleft) bad(path, "corrupt: bad RLE data");
					for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = (uint8_t)get8();
				}
			}
		}
		for (uint32_t i = 0; i < W; i++) convert(&out[3 * ((size_t)j * W + i)], &scan[(size_t)i * 4]);
	}
}

}  // namespace ptx
