// gfx950: does global_load_lds_dwordx4 place lane l's 16 bytes at LDS base + 16 l, and do masked-off lanes leave their slots alone?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_dma_check tools/lds_dma_check.hip && tools/bin/lds_dma_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4* __restrict__ src, float4* dst, int on) {
	__shared__ float4 buf[4][2][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	buf[wave][1][lane] = make_float4(-1.f, -1.f, -1.f, -1.f);
	__syncthreads();
	if (lane < on) __builtin_amdgcn_global_load_lds(src + (blockIdx.x * 4 + wave) * 64 + lane, &buf[wave][1][0], 16, 0, 0);
	__builtin_amdgcn_s_waitcnt(0);
	asm volatile("" ::: "memory");
	dst[(blockIdx.x * 4 + wave) * 64 + lane] = buf[wave][1][63 - lane];
}
int main() {
	const int n = 8 * 256;
	std::vector<float4> h(n), r(n);
	for (int i = 0; i < n; i++) h[i] = make_float4((float)i, i + 0.25f, i + 0.5f, i + 0.75f);
	float4 *d, *o;
	hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
	hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
	int bad = 0;
	for (int on : {64, 40}) {
		hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, d, o, on);
		hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
		for (int i = 0; i < n; i++) {
			const int w = i / 64, l = i % 64, s = 63 - l;
			const float4 want = s < on ? h[w * 64 + s] : make_float4(-1.f, -1.f, -1.f, -1.f);
			if (r[i].x != want.x || r[i].y != want.y || r[i].z != want.z || r[i].w != want.w) { if (bad < 5) printf("on %d i %d got %g %g want %g %g\n", on, i, r[i].x, r[i].w, want.x, want.w); bad++; }
		}
	}
	printf("lds dma check: %d mismatches\n", bad);
	return bad != 0;
}
