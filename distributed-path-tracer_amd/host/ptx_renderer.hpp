// Host-side C++ mirror of the reference's renderer interface (core::renderer,
// path-tracer-core/path_tracer_lib/path_tracer/core/renderer.hpp:15-36) over the C ABI of include/ptx.h:
// same public field names and defaults, load_gltf(path), render() -> PNG bytes, exceptions for errors
// (renderer.cpp:73-74,97-98 throw std::runtime_error). This is the binding a maintainer of the reference adds;
// see INTEGRATION.md. Header-only; link with libptx_hip.so.
#pragma once
#include <ptx.h>

#include <cstdint>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

namespace core {

class renderer {
public:
	static constexpr uint32_t no_sun_light = 0xFFFFFFFFu;   // renderer.hpp:19

	struct uvec2 { uint32_t x, y; };
	uvec2 resolution{1920, 1080};          // renderer.hpp:21
	uint32_t thread_count = 0;             // kept for source compatibility; the GPU grid replaces util::thread_pool
	uint32_t sample_count = 10000;
	uint8_t bounce_count = 4;
	std::filesystem::path environment;     // PNG environment map (the reference holds a loaded image::texture, renderer.hpp:28); empty = none
	float environment_factor[3] = {1, 1, 1};
	bool transparent_background = false;   // debug path of the reference; not built: render() throws if set
	uint32_t camera_index = 0;
	uint32_t sun_light_index = 0;
	uint8_t visualize_kd_tree_depth = 0;   // debug path; not built
	uint64_t seed = 0x5EED;                // key of the counter-based RNG (the reference seeds from random_device)

	explicit renderer(int device = 0) { check(ptx_ctx_create(device, &ctx_)); }
	renderer(const renderer&) = delete;
	renderer& operator=(const renderer&) = delete;
	~renderer() {
		ptx_scene_destroy(scene_);
		ptx_ctx_destroy(ctx_);
	}

	void load_gltf(const std::filesystem::path& path) {
		ptx_load_opts o{camera_index, sun_light_index};
		ptx_scene_destroy(scene_);
		scene_ = nullptr;
		env_set_.clear();
		check(ptx_scene_load_gltf(ctx_, path.string().c_str(), &o, &scene_));
	}

	// radiance sums [H][W][4]; stats optional
	std::vector<float> render_accum(ptx_render_stats* stats = nullptr) const {
		if (!scene_) throw std::runtime_error("render() before load_gltf()");
		if (transparent_background || visualize_kd_tree_depth) throw std::runtime_error("transparent_background / visualize_kd_tree_depth are not built");
		if (environment.string() != env_set_) {   // (re)load the map only when the field changed
			check(ptx_scene_set_environment(scene_, environment.empty() ? nullptr : environment.string().c_str(), 1));
			env_set_ = environment.string();
		}
		ptx_render_cfg c{};
		c.W = resolution.x; c.H = resolution.y; c.spp = sample_count; c.bounces = bounce_count;
		for (int k = 0; k < 3; k++) c.env[k] = environment_factor[k];
		c.seed_lo = (uint32_t)seed; c.seed_hi = (uint32_t)(seed >> 32);
		std::vector<float> accum((size_t)c.W * c.H * 4, 0.f);
		check(ptx_render(scene_, &c, accum.data(), stats));
		return accum;
	}

	std::vector<uint8_t> render() const {   // renderer.cpp:334-428: PNG bytes (RGBA8, ACES tonemap, sRGB)
		std::vector<float> accum = render_accum();
		std::vector<uint8_t> rgba((size_t)resolution.x * resolution.y * 4);
		check(ptx_tonemap_encode(ctx_, accum.data(), resolution.x, resolution.y, sample_count, rgba.data()));
		uint8_t* png = nullptr;
		size_t n = 0;
		check(ptx_encode_png(rgba.data(), resolution.x, resolution.y, &png, &n));
		std::vector<uint8_t> out(png, png + n);
		ptx_free(png);
		return out;
	}

private:
	static void check(int rc) {
		if (rc != PTX_OK) throw std::runtime_error(std::string(ptx_last_error()));
	}
	ptx_ctx* ctx_ = nullptr;
	ptx_scene* scene_ = nullptr;
	mutable std::string env_set_;
};

}  // namespace core
