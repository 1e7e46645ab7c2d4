// JPEG reader for glTF textures (host). The reference loads images with stb_image v2.30 (vendored there, not used here):
// image::image::load, LIB/image/image.cpp:23-54, req_comp = 0. A JPEG decoder is only "the same texture" if every texel is the same
// byte, and JPEG decoders legitimately differ in the inverse DCT, the chroma upsampling filter and the YCbCr -> RGB arithmetic. This
// reader therefore restates the arithmetic stb_image performs — entropy decoding is lossless and needs no such care:
//   * dequantisation while decoding, in 16-bit (coefficient * table entry truncated to short);
//   * the integer inverse DCT: 12-bit fixed-point constants, column pass rounded to 2 extra bits (+512 >> 10), row pass
//     (+65536 + (128 << 17)) >> 17, clamped to 0..255 (the SSE2 path x86-64 builds take is constructed to give the same bits);
//   * chroma upsampling: h2v1 (3:1 taps, +2 >> 2), h1v2 (3:1, +2 >> 2), h2v2 (9:3:3:1 as two 3:1 passes, +8 >> 4), others by replication;
//   * YCbCr -> RGB in 20-bit fixed point with 12-bit constants and the chroma term of G truncated to 16 bits first, as the SIMD path does;
//   * JFIF / Adobe APP14 handling of the colour transform, grey images stay 1 channel, 3-component images give 3 channels.
// Supported: baseline and extended-sequential Huffman (SOF0 / SOF1, 8-bit) and progressive (SOF2) JPEGs with 1 or 3 components,
// restart intervals, any sampling factors up to 4. CMYK / YCCK (4 components), 12-bit and arithmetic-coded files: PTX_ERR_UNSUPPORTED.
// Pinned against the compiled reference on its own Sponza textures and on small synthetic files (tests/golden/jpeg_vectors.npz).
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "flat_scene.hpp"

namespace ptx {
namespace {

[[noreturn]] void bad(const std::string& path, const char* why, int code = 3) { throw Error{code, "JPEG '" + path + "': " + why}; }

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run may step past the block: those writes land here
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
	// canonical code: for each length 1..16, the first code, the index of its first symbol, and one past the largest code (left-aligned to 16 bits)
	uint32_t maxcode[18];
	int32_t delta[17];
	uint16_t firstcode[17];
	uint8_t values[256];
	uint8_t sizes[257];
	bool present = false;
	void build(const std::string& path, const uint8_t counts[16]) {
		int k = 0;
		for (int i = 0; i < 16; i++)
			for (int j = 0; j < counts[i]; j++) {
				if (k >= 256) bad(path, "bad Huffman table");
				sizes[k++] = (uint8_t)(i + 1);
			}
		sizes[k] = 0;
		uint32_t code = 0;
		k = 0;
		for (int len = 1; len <= 16; len++) {
			delta[len] = k - (int32_t)code;
			firstcode[len] = (uint16_t)code;
			while (sizes[k] == len) { k++; code++; }
			if (code > (1u << len)) bad(path, "bad Huffman code lengths");
			maxcode[len] = code << (16 - len);
			code <<= 1;
		}
		maxcode[17] = 0xFFFFFFFFu;
		present = true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
	int x = 0, y = 0, w2 = 0, h2 = 0;   // size in samples, padded size
	std::vector<uint8_t> data;          // [h2][w2] decoded samples
	std::vector<int16_t> coeff;         // progressive: [blocks][64]
	int coeff_w = 0, coeff_h = 0;
};

struct Decoder {
	const std::string& path;
	const uint8_t* p;
	const uint8_t* end;
	uint32_t W = 0, H = 0;
	int n_comp = 0;
	bool progressive = false, jfif = false;
	int app14_transform = -1, rgb_ids = 0;
	Component comp[4];
	Huff hdc[4], hac[4];
	uint16_t dequant[4][64] = {};
	bool dq_present[4] = {false, false, false, false};
	int h_max = 1, v_max = 1, mcu_w = 0, mcu_h = 0, mcu_x = 0, mcu_y = 0;
	int restart_interval = 0, todo = 0;
	// scan state
	int scan_n = 0, order[4];
	int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
	// bit reader
	uint32_t code_buffer = 0;
	int code_bits = 0;
	uint8_t marker = 0xFF;
	bool nomore = false;

	Decoder(const std::string& path_, const std::vector<uint8_t>& file) : path(path_), p(file.data()), end(file.data() + file.size()) {}

	int get8() { return p < end ? *p++ : 0; }
	int get16() { int a = get8(); return (a << 8) | get8(); }

	void grow() {
		do {
			uint32_t b = nomore ? 0 : (uint32_t)get8();
			if (b == 0xFF) {
				int c = get8();
				while (c == 0xFF) c = get8();   // fill bytes
				if (c != 0) { marker = (uint8_t)c; nomore = true; return; }
			}
			code_buffer |= b << (24 - code_bits);
			code_bits += 8;
		} while (code_bits <= 24);
	}
	int bits(int n) {   // n in 0..16
		if (n == 0) return 0;
		if (code_bits < n) grow();
		if (code_bits < n) return 0;   // the entropy data ended at a marker: stb_image v2.30 reads zeros from there on, and so does this reader
		const uint32_t k = code_buffer >> (32 - n);
		code_buffer <<= n;
		code_bits -= n;
		return (int)k;
	}
	int bit() { return bits(1); }
	// receive n bits and sign-extend the JPEG way (a leading 0 bit means negative: value - (2^n - 1))
	int extend_receive(int n) {
		if (n == 0) return 0;
		if (code_bits < n) grow();
		if (code_bits < n) return 0;   // out of bits: the value 0, not the sign extension of n zero bits
		const int v = bits(n);
		return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
	}
	int decode(const Huff& h) {
		if (code_bits < 16) grow();
		const uint32_t top = code_buffer >> 16;
		int len = 1;
		while (top >= h.maxcode[len]) len++;
		if (len > 16 || len > code_bits) bad(path, "bad Huffman code");   // a code that needs bits the stream no longer has
		const int idx = (int)(code_buffer >> (32 - len)) + h.delta[len];
		if (idx < 0 || idx >= 256) bad(path, "bad Huffman code");
		code_buffer <<= len;
		code_bits -= len;
		return h.values[idx];
	}
	void reset_entropy() {
		code_bits = 0; code_buffer = 0; nomore = false; marker = 0xFF;
		for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
		todo = restart_interval ? restart_interval : 0x7FFFFFFF;
		eob_run = 0;
	}

	// A corrupt stream can run the DC predictor out of int range, or its dequantised value out of the 16-bit coefficient: refused
	// (as stb_image v2.30 refuses them: "bad delta", "can't merge dc and ac") instead of overflowing a signed integer
	int checked_dc(int pred, int diff, int factor) {
		const long long dc = (long long)pred + diff;
		if (dc > 0x7FFFFFFFll || dc < -0x80000000ll) bad(path, "corrupt JPEG: DC predictor out of range");
		const long long v = dc * factor;
		if (v > 32767 || v < -32768) bad(path, "corrupt JPEG: DC coefficient out of range");
		return (int)dc;
	}
	void need_quant(const Component& c) {
		if (!dq_present[c.tq]) bad(path, "component uses a quantisation table that was not defined");
	}
	// ---------------- baseline block: Huffman -> dequantised coefficients in natural order
	void decode_block(int16_t data[64], Component& c) {
		const Huff& dc = hdc[c.hd];
		const Huff& ac = hac[c.ha];
		const uint16_t* dq = dequant[c.tq];
		memset(data, 0, 64 * sizeof(int16_t));
		const int t = decode(dc);
		if (t > 15) bad(path, "bad DC size");
		const int diff = t ? extend_receive(t) : 0;
		c.dc_pred = checked_dc(c.dc_pred, diff, (int)dq[0]);
		data[0] = (int16_t)(c.dc_pred * dq[0]);
		for (int k = 1; k < 64;) {
			const int rs = decode(ac), s = rs & 15, r = rs >> 4;
			if (s == 0) {
				if (rs != 0xF0) break;   // end of block
				k += 16;
			} else {
				k += r;
				const int zig = kZigzag[k++];
				data[zig] = (int16_t)(extend_receive(s) * dq[zig]);
			}
		}
	}
	// ---------------- progressive: DC scans
	void decode_block_prog_dc(int16_t data[64], Component& c) {
		if (spec_end != 0) bad(path, "can't merge DC and AC");
		if (succ_high == 0) {
			memset(data, 0, 64 * sizeof(int16_t));
			const int t = decode(hdc[c.hd]);
			if (t > 15) bad(path, "bad DC size");
			const int diff = t ? extend_receive(t) : 0;
			c.dc_pred = checked_dc(c.dc_pred, diff, 1 << succ_low);
			data[0] = (int16_t)(c.dc_pred * (1 << succ_low));
		} else if (bit()) {
			data[0] = (int16_t)(data[0] + (1 << succ_low));
		}
	}
	// ---------------- progressive: AC scans (first pass and refinement)
	void decode_block_prog_ac(int16_t data[64], const Huff& ac) {
		if (spec_start == 0) bad(path, "can't merge DC and AC");
		if (succ_high == 0) {
			const int shift = succ_low;
			if (eob_run) { eob_run--; return; }
			int k = spec_start;
			do {
				const int rs = decode(ac), s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) {
						eob_run = 1 << r;
						if (r) eob_run += bits(r);
						eob_run--;
						break;
					}
					k += 16;
				} else {
					k += r;
					const int zig = kZigzag[k++];
					data[zig] = (int16_t)(extend_receive(s) * (1 << shift));
				}
			} while (k <= spec_end);
		} else {
			const int16_t b = (int16_t)(1 << succ_low);
			auto refine = [&](int16_t& v) {
				if (v != 0 && bit() && (v & b) == 0) v = (int16_t)(v > 0 ? v + b : v - b);
			};
			if (eob_run) {
				eob_run--;
				for (int k = spec_start; k <= spec_end; k++) refine(data[kZigzag[k]]);
				return;
			}
			int k = spec_start;
			do {
				const int rs = decode(ac);
				int s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) {
						eob_run = (1 << r) - 1;
						if (r) eob_run += bits(r);
						r = 64;   // run to the end of the band, refining on the way
					}
					// r = 15: skip 16 zero-history coefficients
				} else {
					if (s != 1) bad(path, "bad Huffman code");
					s = bit() ? b : -b;
				}
				while (k <= spec_end) {
					int16_t& v = data[kZigzag[k++]];
					if (v != 0) {
						if (bit() && (v & b) == 0) v = (int16_t)(v > 0 ? v + b : v - b);
					} else {
						if (r == 0) { v = (int16_t)s; break; }
						r--;
					}
				}
			} while (k <= spec_end);
		}
	}

	// ---------------- the integer inverse DCT (see header comment); coefficients already dequantised
	static inline uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
	static void idct_block(uint8_t* out, int stride, const int16_t d[64]) {
		constexpr auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
		int val[64];
		auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int& x0, int& x1, int& x2, int& x3, int& t0, int& t1, int& t2, int& t3) {
			int p2 = s2, p3 = s6;
			int p1 = (p2 + p3) * f2f(0.5411961f);
			t2 = p1 + p3 * f2f(-1.847759065f);
			t3 = p1 + p2 * f2f(0.765366865f);
			p2 = s0; p3 = s4;
			t0 = (p2 + p3) * 4096;
			t1 = (p2 - p3) * 4096;
			x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
			t0 = s7; t1 = s5; t2 = s3; t3 = s1;
			p3 = t0 + t2;
			int p4 = t1 + t3;
			p1 = t0 + t3;
			p2 = t1 + t2;
			const int p5 = (p3 + p4) * f2f(1.175875602f);
			t0 = t0 * f2f(0.298631336f);
			t1 = t1 * f2f(2.053119869f);
			t2 = t2 * f2f(3.072711026f);
			t3 = t3 * f2f(1.501321110f);
			p1 = p5 + p1 * f2f(-0.899976223f);
			p2 = p5 + p2 * f2f(-2.562915447f);
			p3 = p3 * f2f(-1.961570560f);
			p4 = p4 * f2f(-0.390180644f);
			t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
		};
		for (int i = 0; i < 8; i++) {   // columns
			const int16_t* c = d + i;
			int* v = val + i;
			if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
				const int dc = c[0] * 4;
				v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
			} else {
				int x0, x1, x2, x3, t0, t1, t2, t3;
				pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
				x0 += 512; x1 += 512; x2 += 512; x3 += 512;
				v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
				v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
				v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
				v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
			}
		}
		for (int i = 0; i < 8; i++) {   // rows
			const int* v = val + 8 * i;
			uint8_t* o = out + (size_t)stride * i;
			int x0, x1, x2, x3, t0, t1, t2, t3;
			pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
			const int bias = 65536 + (128 << 17);
			x0 += bias; x1 += bias; x2 += bias; x3 += bias;
			o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
			o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
			o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
			o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
		}
	}

	// ---------------- markers
	void read_dqt(int len) {
		len -= 2;
		while (len > 0) {
			const int q = get8(), prec = q >> 4, t = q & 15;
			if ((prec != 0 && prec != 1) || t > 3) bad(path, "bad DQT");
			for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
			dq_present[t] = true;
			len -= prec ? 129 : 65;
		}
		if (len != 0) bad(path, "bad DQT length");
	}
	void read_dht(int len) {
		len -= 2;
		while (len > 0) {
			const int q = get8(), tc = q >> 4, th = q & 15;
			if (tc > 1 || th > 3) bad(path, "bad DHT header");
			uint8_t counts[16];
			int n = 0;
			for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); n += counts[i]; }
			if (n > 256) bad(path, "bad DHT header");
			Huff& h = tc ? hac[th] : hdc[th];
			for (int i = 0; i < n; i++) h.values[i] = (uint8_t)get8();
			h.build(path, counts);
			len -= 17 + n;
		}
		if (len != 0) bad(path, "bad DHT length");
	}
	void read_sof(int len) {
		if (len < 11) bad(path, "bad SOF length");
		if (get8() != 8) bad(path, "only 8-bit JPEGs are supported", 7);
		H = (uint32_t)get16(); W = (uint32_t)get16();
		if (!H || !W) bad(path, "zero-sized image");
		n_comp = get8();
		if (n_comp == 4) bad(path, "CMYK / YCCK JPEGs are not supported", 7);
		if (n_comp != 1 && n_comp != 3) bad(path, "bad component count");
		if (len != 8 + 3 * n_comp) bad(path, "bad SOF length");
		rgb_ids = 0;
		static const char rgb[3] = {'R', 'G', 'B'};
		for (int i = 0; i < n_comp; i++) {
			Component& c = comp[i];
			c.id = get8();
			if (n_comp == 3 && c.id == rgb[i]) rgb_ids++;
			const int q = get8();
			c.h = q >> 4; c.v = q & 15;
			if (!c.h || c.h > 4 || !c.v || c.v > 4) bad(path, "bad sampling factor");
			c.tq = get8();
			if (c.tq > 3) bad(path, "bad quantisation table index");
		}
		for (int i = 0; i < n_comp; i++) { h_max = std::max(h_max, comp[i].h); v_max = std::max(v_max, comp[i].v); }
		for (int i = 0; i < n_comp; i++)
			if (h_max % comp[i].h || v_max % comp[i].v) bad(path, "bad sampling factors");
		mcu_w = h_max * 8; mcu_h = v_max * 8;
		mcu_x = ((int)W + mcu_w - 1) / mcu_w; mcu_y = ((int)H + mcu_h - 1) / mcu_h;
		if ((uint64_t)W * H > (1ull << 28)) bad(path, "image too large");
		{   // a header must not be able to demand gigabytes from a tiny file: every 8 x 8 block costs at least one bit of entropy-coded data
			uint64_t blocks = 0;
			for (int i = 0; i < n_comp; i++) blocks += (uint64_t)mcu_x * mcu_y * comp[i].h * comp[i].v;
			if ((uint64_t)(end - p) * 8 < blocks) bad(path, "truncated file");
		}
		for (int i = 0; i < n_comp; i++) {
			Component& c = comp[i];
			c.x = ((int)W * c.h + h_max - 1) / h_max;
			c.y = ((int)H * c.v + v_max - 1) / v_max;
			c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
			c.data.assign((size_t)c.w2 * c.h2, 0);
			if (progressive) {
				c.coeff_w = c.w2 / 8; c.coeff_h = c.h2 / 8;
				c.coeff.assign((size_t)c.w2 * c.h2, 0);
			}
		}
	}
	void read_sos(int len) {
		scan_n = get8();
		if (scan_n < 1 || scan_n > 4 || scan_n > n_comp) bad(path, "bad SOS component count");
		if (len != 6 + 2 * scan_n) bad(path, "bad SOS length");
		for (int i = 0; i < scan_n; i++) {
			const int id = get8(), q = get8();
			int which = 0;
			while (which < n_comp && comp[which].id != id) which++;
			if (which == n_comp) bad(path, "bad SOS component");
			comp[which].hd = q >> 4; comp[which].ha = q & 15;
			if (comp[which].hd > 3 || comp[which].ha > 3) bad(path, "bad Huffman table index");
			order[i] = which;
		}
		spec_start = get8(); spec_end = get8();
		const int aa = get8();
		succ_high = aa >> 4; succ_low = aa & 15;
		if (progressive) {
			if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) bad(path, "bad SOS");
		} else {
			if (spec_start != 0 || succ_high != 0 || succ_low != 0) bad(path, "bad SOS");
			spec_end = 63;
		}
	}
	void need_tables(const Component& c, bool dc, bool ac) {
		if ((dc && !hdc[c.hd].present) || (ac && !hac[c.ha].present)) bad(path, "scan uses a Huffman table that was not defined");
	}
	// after each MCU (or block of a single-component scan): restart interval handling
	bool after_unit() {
		if (--todo <= 0) {
			if (code_bits < 24) grow();
			if (!(marker >= 0xD0 && marker <= 0xD7)) return false;   // no restart marker where one is due: the scan ends
			reset_entropy();
		}
		return true;
	}
	void decode_scan() {
		reset_entropy();
		int16_t blk[64];
		if (!progressive) {
			for (int i = 0; i < scan_n; i++) { need_tables(comp[order[i]], true, true); need_quant(comp[order[i]]); }
			if (scan_n == 1) {   // non-interleaved: the component's own blocks, row by row, only those that cover the image
				Component& c = comp[order[0]];
				const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
				for (int j = 0; j < h; j++)
					for (int i = 0; i < w; i++) {
						decode_block(blk, c);
						idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, blk);
						if (!after_unit()) return;
					}
			} else {
				for (int j = 0; j < mcu_y; j++)
					for (int i = 0; i < mcu_x; i++) {
						for (int k = 0; k < scan_n; k++) {
							Component& c = comp[order[k]];
							for (int y = 0; y < c.v; y++)
								for (int x = 0; x < c.h; x++) {
									const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
									decode_block(blk, c);
									idct_block(&c.data[(size_t)c.w2 * y2 + x2], c.w2, blk);
								}
						}
						if (!after_unit()) return;
					}
			}
		} else {
			if (scan_n == 1) {
				Component& c = comp[order[0]];
				need_tables(c, spec_start == 0 && succ_high == 0, spec_start != 0);
				const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
				for (int j = 0; j < h; j++)
					for (int i = 0; i < w; i++) {
						int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
						if (spec_start == 0) decode_block_prog_dc(d, c);
						else decode_block_prog_ac(d, hac[c.ha]);
						if (!after_unit()) return;
					}
			} else {
				for (int k = 0; k < scan_n; k++) need_tables(comp[order[k]], succ_high == 0, false);
				for (int j = 0; j < mcu_y; j++)
					for (int i = 0; i < mcu_x; i++) {
						for (int k = 0; k < scan_n; k++) {
							Component& c = comp[order[k]];
							for (int y = 0; y < c.v; y++)
								for (int x = 0; x < c.h; x++) {
									const int x2 = i * c.h + x, y2 = j * c.v + y;
									decode_block_prog_dc(&c.coeff[64 * ((size_t)x2 + (size_t)y2 * c.coeff_w)], c);   // interleaved progressive scans are DC scans
								}
						}
						if (!after_unit()) return;
					}
			}
		}
	}
	void finish_progressive() {
		for (int n = 0; n < n_comp; n++) {
			Component& c = comp[n];
			const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
			need_quant(c);
			const uint16_t* dq = dequant[c.tq];
			for (int j = 0; j < h; j++)
				for (int i = 0; i < w; i++) {
					int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
					for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * dq[k]);
					idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, d);
				}
		}
	}

	void decode_image() {
		if (get8() != 0xFF || get8() != 0xD8) bad(path, "not a JPEG");
		bool have_sof = false, have_scan = false;
		int m = next_marker();
		for (;;) {
			if (m == 0xD9) break;                         // EOI
			if (m < 0) { if (have_scan) break; bad(path, "truncated file"); }
			if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
				if (have_sof) bad(path, "more than one frame");
				progressive = m == 0xC2;
				read_sof(get16());
				have_sof = true;
			} else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
				bad(path, "lossless / hierarchical / arithmetic-coded JPEGs are not supported", 7);
			} else if (m == 0xC4) read_dht(get16());
			else if (m == 0xDB) read_dqt(get16());
			else if (m == 0xDD) { if (get16() != 4) bad(path, "bad DRI length"); restart_interval = get16(); }
			else if (m == 0xDA) {
				if (!have_sof) bad(path, "scan before the frame header");
				read_sos(get16());
				decode_scan();
				have_scan = true;
				if (marker != 0xFF) { m = marker; marker = 0xFF; if (m >= 0xD0 && m <= 0xD7) m = next_marker(); continue; }   // the entropy decoder ran into a marker
				// otherwise skip to the next marker (trailing bytes of the scan)
			} else if (m == 0xE0) {                         // APP0: JFIF
				int len = get16() - 2;
				if (len >= 5) {
					static const char tag[5] = {'J', 'F', 'I', 'F', 0};
					bool ok = true;
					for (int i = 0; i < 5; i++) ok &= get8() == tag[i];
					len -= 5;
					if (ok) jfif = true;
				}
				skip(len);
			} else if (m == 0xEE) {                         // APP14: Adobe
				int len = get16() - 2;
				if (len >= 12) {
					static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
					bool ok = true;
					for (int i = 0; i < 6; i++) ok &= get8() == tag[i];
					len -= 6;
					if (ok) { get8(); get16(); get16(); app14_transform = get8(); len -= 6; }
				}
				skip(len);
			} else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
				skip(get16() - 2);
			} else if (m >= 0xD0 && m <= 0xD7) {
				// stray restart marker
			} else bad(path, "unknown marker");
			m = next_marker();
		}
		if (!have_sof || !have_scan) bad(path, "no image data");
		if (progressive) finish_progressive();
	}
	void skip(int n) {
		if (n < 0) bad(path, "bad segment length");
		p = (end - p) < n ? end : p + n;
	}
	int next_marker() {
		// scan forward to the next 0xFF xx (xx != 0, != 0xFF)
		while (p < end) {
			if (*p++ != 0xFF) continue;
			while (p < end && *p == 0xFF) p++;
			if (p >= end) return -1;
			const int m = *p++;
			if (m != 0) return m;
		}
		return -1;
	}

	// ---------------- upsampling + colour conversion -> interleaved pixels
	static inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
	static inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }
	void resample_row(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w, int hs, int vs) const {
		if (hs == 1 && vs == 1) { memcpy(out, near_, (size_t)w); return; }
		if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = div4(3 * near_[i] + far_[i] + 2); return; }
		if (hs == 2 && vs == 1) {
			const uint8_t* in = near_;
			if (w == 1) { out[0] = out[1] = in[0]; return; }
			out[0] = in[0];
			out[1] = div4(in[0] * 3 + in[1] + 2);
			int i;
			for (i = 1; i < w - 1; i++) {
				const int n = 3 * in[i] + 2;
				out[i * 2] = div4(n + in[i - 1]);
				out[i * 2 + 1] = div4(n + in[i + 1]);
			}
			out[i * 2] = div4(in[w - 2] * 3 + in[w - 1] + 2);
			out[i * 2 + 1] = in[w - 1];
			return;
		}
		if (hs == 2 && vs == 2) {
			if (w == 1) { out[0] = out[1] = div4(3 * near_[0] + far_[0] + 2); return; }
			int t1 = 3 * near_[0] + far_[0], t0;
			out[0] = div4(t1 + 2);
			for (int i = 1; i < w; i++) {
				t0 = t1;
				t1 = 3 * near_[i] + far_[i];
				out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
				out[i * 2] = div16(3 * t1 + t0 + 8);
			}
			out[w * 2 - 1] = div4(t1 + 2);
			return;
		}
		for (int i = 0; i < w; i++)      // any other ratio: nearest neighbour
			for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];
	}
	void output(uint32_t& C, std::vector<uint8_t>& out) {
		const int n = n_comp >= 3 ? 3 : 1;
		C = (uint32_t)n;
		const bool is_rgb = n_comp == 3 && (rgb_ids == 3 || (app14_transform == 0 && !jfif));
		out.assign((size_t)W * H * n, 0);
		struct Res { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } res[3];
		for (int k = 0; k < n_comp; k++) {
			Res& r = res[k];
			r.hs = h_max / comp[k].h; r.vs = v_max / comp[k].v;
			r.ystep = r.vs >> 1;
			r.w_lores = ((int)W + r.hs - 1) / r.hs;
			r.ypos = 0;
			r.line0 = r.line1 = comp[k].data.data();
			r.buf.assign((size_t)W + 16 + (size_t)r.hs * 2, 0);
		}
		auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
		const int c_r_cr = fixed(1.40200f), c_g_cr = -fixed(0.71414f), c_g_cb = -fixed(0.34414f), c_b_cb = fixed(1.77200f);
		for (uint32_t j = 0; j < H; j++) {
			const uint8_t* row[3] = {nullptr, nullptr, nullptr};
			for (int k = 0; k < n_comp; k++) {
				Res& r = res[k];
				const bool y_bot = r.ystep >= (r.vs >> 1);
				resample_row(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
				row[k] = r.buf.data();
				if (++r.ystep >= r.vs) {
					r.ystep = 0;
					r.line0 = r.line1;
					if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
				}
			}
			uint8_t* o = &out[(size_t)W * n * j];
			if (n == 1) { memcpy(o, row[0], W); continue; }
			if (is_rgb) {
				for (uint32_t i = 0; i < W; i++) { o[3 * i] = row[0][i]; o[3 * i + 1] = row[1][i]; o[3 * i + 2] = row[2][i]; }
				continue;
			}
			for (uint32_t i = 0; i < W; i++) {
				const int y_fixed = (row[0][i] << 20) + (1 << 19);
				const int cb = row[1][i] - 128, cr = row[2][i] - 128;
				int r = y_fixed + cr * c_r_cr;
				int g = y_fixed + cr * c_g_cr + (int)((uint32_t)(cb * c_g_cb) & 0xffff0000u);
				int b = y_fixed + cb * c_b_cb;
				r >>= 20; g >>= 20; b >>= 20;
				o[3 * i] = clamp8(r); o[3 * i + 1] = clamp8(g); o[3 * i + 2] = clamp8(b);
			}
		}
	}
};

}  // namespace

void read_jpeg(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw Error{2, "Failed to load image to memory: " + path};   // image.cpp:44-45
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	Decoder d(path, file);
	d.decode_image();
	W = d.W; H = d.H;
	d.output(C, out);
}

// image::image::load dispatch (stb_image picks the decoder by content, not by extension)
void read_image(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out) {
	uint8_t sig[4] = {0, 0, 0, 0};
	{
		std::ifstream f(path, std::ios::binary);
		if (!f) throw Error{2, "Failed to load image to memory: " + path};
		f.read((char*)sig, 4);
	}
	if (sig[0] == 0xFF && sig[1] == 0xD8) return read_jpeg(path, W, H, C, out);
	return read_png(path, W, H, C, out);   // reports anything else as "not a PNG"
}

TexRec load_texture(FlatScene& s, const std::string& path, bool srgb) {
	uint32_t W = 0, H = 0, C = 0;
	if (is_hdr_file(path)) {
		std::vector<float> px;
		read_hdr(path, W, H, C, px);
		if (s.texels_f.size() + px.size() > 0x3FFFFFFFull) throw Error{3, "more than 4 GiB of float texels"};
		const TexRec t{W, H, C | (srgb ? kTexSrgb : 0u) | kTexFloat, (uint32_t)s.texels_f.size()};
		s.texels_f.insert(s.texels_f.end(), px.begin(), px.end());
		return t;
	}
	std::vector<uint8_t> px;
	read_image(path, W, H, C, px);
	if (s.texels.size() + px.size() > 0xFFFFFFFFull) throw Error{3, "more than 4 GiB of texels"};
	const TexRec t{W, H, C | (srgb ? kTexSrgb : 0u), (uint32_t)s.texels.size()};
	s.texels.insert(s.texels.end(), px.begin(), px.end());
	while (s.texels.size() % 16) s.texels.push_back(0);
	return t;
}

}  // namespace ptx
