// Minimal C++ host over the mirror class: what path-tracer-core/src/main.cpp + worker.cpp reduce to once the
// Lambda / S3 plumbing (out of scope) is taken away:  ptx_render_cli <scene.gltf> <out.png> [W H spp bounces]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "ptx_renderer.hpp"

int main(int argc, char** argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: %s <scene.gltf> <out.png> [W H spp bounces]\n", argv[0]);
		return 1;
	}
	try {
		core::renderer r(0);
		if (argc >= 7) {
			r.resolution = {(uint32_t)std::atoi(argv[3]), (uint32_t)std::atoi(argv[4])};
			r.sample_count = (uint32_t)std::atoi(argv[5]);
			r.bounce_count = (uint8_t)std::atoi(argv[6]);
		} else {
			r.sample_count = 64;
		}
		r.load_gltf(argv[1]);
		auto t0 = std::chrono::steady_clock::now();
		std::vector<uint8_t> png = r.render();
		double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		std::ofstream(argv[2], std::ios::binary).write((const char*)png.data(), (std::streamsize)png.size());
		std::printf("{\"W\": %u, \"H\": %u, \"spp\": %u, \"bounces\": %u, \"seconds\": %.4f, \"png_bytes\": %zu}\n", r.resolution.x, r.resolution.y,
		            r.sample_count, (unsigned)r.bounce_count, s, png.size());
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 2;
	}
	return 0;
}
