// What bounds a tree walk's fetches on gfx950: distinct cache LINES or lane-LOADS? Every active lane runs its own dependent chain through
// a table of 128-byte-aligned 128-byte records (the next record index comes out of the data), and per step loads
//   mode 1: one 16-byte piece of the record            (a KD child pair)
//   mode 3: three consecutive 16-byte pieces            (a 48-byte triangle record)
//   mode 8: all eight 16-byte pieces                    (a 3-level treelet in one cache line)
//   mode 2: one piece of this record + one piece of a second, unrelated record (two lines per step)
// for table sizes from L2-resident to 160 MB, at 5 waves per SIMD, 20 and 64 active lanes. Prints steps per second (chip) and ns per step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_bench2 tools/gather_bench2.hip && tools/bin/gather_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

template <int MODE>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ data, uint32_t n_mask, int steps, uint32_t lanes_on, uint32_t* __restrict__ out) {
	const uint32_t lane = threadIdx.x & 63u;
	if (lane >= lanes_on) return;
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & n_mask;
	uint32_t acc = 0;
	for (int s = 0; s < steps; s++) {
		const uint4* rec = data + (size_t)idx * 8;
		uint4 v = rec[0];
		if (MODE == 3) { const uint4 a = rec[1], b = rec[2]; acc += a.y + b.y; }
		if (MODE == 8) { for (int k = 1; k < 8; k++) acc += rec[k].y; }
		if (MODE == 2) { const uint4 a = data[(size_t)((idx * 40503u + 17u) & n_mask) * 8]; acc += a.y; }
		acc += v.y;
		idx = v.x & n_mask;
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int n_cu = prop.multiProcessorCount;
	const size_t max_recs = (size_t)1 << 21;   // 256 MiB of 128-byte records
	std::vector<uint4> h(max_recs * 8);
	std::mt19937 rng(1);
	for (size_t i = 0; i < max_recs; i++) { const uint32_t nx = rng(); for (int k = 0; k < 8; k++) h[i * 8 + k] = make_uint4(nx, (uint32_t)i + k, 0, 0); }
	uint4* d; uint32_t* out;
	hipMalloc(&d, h.size() * sizeof(uint4));
	hipMalloc(&out, (size_t)n_cu * 16 * 256 * 4);
	hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	printf("%10s %5s %5s %5s %14s %12s\n", "table", "mode", "waves", "lanes", "Gsteps/s", "ns/step/wave");
	const int steps = 1000;
	for (int logn : {13, 15, 18, 20, 21}) {   // 1 MiB, 4 MiB (one XCD's L2), 32 MiB, 128 MiB, 256 MiB
		for (int mode : {1, 3, 8, 2}) {
			for (int wps : {5}) {
				for (uint32_t lanes : {20u, 64u}) {
					const uint32_t mask = ((uint32_t)1 << logn) - 1u;
					const int grid = n_cu * wps;
					auto run = [&](int st) {
						if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(grid), dim3(256), 0, 0, d, mask, st, lanes, out);
						if (mode == 3) hipLaunchKernelGGL(k_chase<3>, dim3(grid), dim3(256), 0, 0, d, mask, st, lanes, out);
						if (mode == 8) hipLaunchKernelGGL(k_chase<8>, dim3(grid), dim3(256), 0, 0, d, mask, st, lanes, out);
						if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(grid), dim3(256), 0, 0, d, mask, st, lanes, out);
					};
					run(100);
					hipEventRecord(e0);
					run(steps);
					hipEventRecord(e1);
					hipEventSynchronize(e1);
					float ms = 0; hipEventElapsedTime(&ms, e0, e1);
					const double n = (double)grid * 4 * lanes * steps;
					printf("%7zu MiB %5d %5d %5u %14.2f %12.1f\n", (((size_t)1 << logn) * 128) >> 20, mode, wps, lanes, n / ms / 1e6, ms * 1e6 / steps);
					fflush(stdout);
				}
			}
		}
	}
	return 0;
}
