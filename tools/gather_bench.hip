// Dependent random gathers on gfx950: what a wave that chases pointers through a large array can get out of the memory system.
// Every active lane runs its own chain idx = data[idx].x (one 16-byte load per step, the shape of a KD-tree walk: fetch a node
// pair, decide, fetch the next), over arrays of several sizes (L2-resident ... HBM), at several occupancies and lane counts.
// Prints lane-loads per second for the whole chip and the time one dependent step takes for a wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_bench tools/gather_bench.hip && tools/bin/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ data, uint32_t n_mask, int steps, uint32_t lanes_on, uint32_t* __restrict__ out) {
	const uint32_t lane = threadIdx.x & 63u;
	if (lane >= lanes_on) return;
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & n_mask;
	uint32_t acc = 0;
	for (int s = 0; s < steps; s++) {
		const uint4 v = data[idx];
		acc += v.y;
		idx = v.x & n_mask;
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int n_cu = prop.multiProcessorCount;
	const size_t max_elems = (size_t)1 << 26;   // 1 GiB of uint4
	std::vector<uint4> h(max_elems);
	std::mt19937 rng(1);
	for (size_t i = 0; i < max_elems; i++) h[i] = make_uint4(rng(), (uint32_t)i, 0, 0);
	uint4* d; uint32_t* out;
	hipMalloc(&d, max_elems * sizeof(uint4));
	hipMalloc(&out, (size_t)n_cu * 16 * 256 * 4);
	hipMemcpy(d, h.data(), max_elems * sizeof(uint4), hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	printf("%10s %6s %6s %14s %14s\n", "bytes", "waves", "lanes", "Gloads/s", "ns/step/wave");
	const int steps = 2000;
	for (int logn : {9, 10, 12, 16, 18, 21, 23, 26}) {   // 8 KiB, 16 KiB, 64 KiB (vector L1 is 32 KiB), 1 MiB, 4 MiB, 32 MiB, 128 MiB, 1 GiB
		for (int wps : {1, 2, 4, 6, 8}) {                // waves per SIMD = 256-thread blocks per CU
			for (uint32_t lanes : {4u, 16u, 32u, 64u}) {
				const uint32_t mask = ((uint32_t)1 << logn) - 1u;
				const int grid = n_cu * wps;
				hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, d, mask, 200, lanes, out);
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, d, mask, steps, lanes, out);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms = 0; hipEventElapsedTime(&ms, e0, e1);
				const double loads = (double)grid * 4 * lanes * steps;
				printf("%10zu %6d %6u %14.2f %14.1f\n", ((size_t)1 << logn) * 16, wps, lanes, loads / ms / 1e6, ms * 1e6 / steps);
			}
		}
	}
	return 0;
}
