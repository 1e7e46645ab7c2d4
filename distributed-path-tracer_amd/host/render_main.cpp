// Minimal C++ host over the mirror class: what path-tracer-core/src/main.cpp + worker.cpp reduce to once the
// Lambda / S3 plumbing (out of scope) is taken away:
//   ptx_render_cli <scene.gltf> <out.png> [W H spp bounces]
//   ptx_render_cli --event <event.json> <local scene root dir> <out.png>     (the worker's Lambda event, main.cpp:9-25)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "ptx_renderer.hpp"

int main(int argc, char** argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: %s <scene.gltf> <out.png> [W H spp bounces]\n", argv[0]);
		return 1;
	}
	if (argc == 5 && std::string(argv[1]) == "--event") {
		ptx_ctx* ctx = nullptr;
		ptx_scene* scene = nullptr;
		ptx_render_cfg cfg{};
		ptx_worker_event ev{};
		auto die = [&](const char* what) { std::fprintf(stderr, "error: %s: %s\n", what, ptx_last_error()); return 2; };
		if (ptx_ctx_create(0, &ctx) != PTX_OK) return die("ptx_ctx_create");
		if (ptx_worker_event_load(ctx, argv[2], argv[3], &scene, &cfg, &ev) != PTX_OK) return die("ptx_worker_event_load");
		std::vector<float> accum((size_t)cfg.W * cfg.H * 4, 0.f);
		ptx_render_stats st{};
		if (ptx_render(scene, &cfg, accum.data(), &st) != PTX_OK) return die("ptx_render");
		std::vector<uint8_t> rgba((size_t)cfg.W * cfg.H * 4);
		if (ptx_tonemap_encode(ctx, accum.data(), cfg.W, cfg.H, cfg.spp, rgba.data()) != PTX_OK) return die("ptx_tonemap_encode");
		uint8_t* png = nullptr;
		size_t n = 0;
		if (ptx_encode_png(rgba.data(), cfg.W, cfg.H, &png, &n) != PTX_OK) return die("ptx_encode_png");
		std::ofstream(argv[4], std::ios::binary).write((const char*)png, (std::streamsize)n);   // worker.cpp:101-104 uploads this as <scene_root>test.png
		ptx_free(png);
		std::printf("{\"worker_id\": \"%s\", \"num_workers\": %d, \"W\": %u, \"H\": %u, \"spp\": %u, \"bounces\": %u, \"rays\": %llu, \"kernel_ms\": %.3f}\n",
		            ev.worker_id, ev.num_workers, cfg.W, cfg.H, cfg.spp, cfg.bounces, (unsigned long long)st.rays, st.kernel_ms);
		ptx_scene_destroy(scene);
		ptx_ctx_destroy(ctx);
		return 0;
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
