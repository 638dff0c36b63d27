// micro-benchmark: latency of the building blocks of the scalarised entropy decoder on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_lds_chain(uint32_t *out, int iters, unsigned long long *cycles) {
	__shared__ uint32_t tab[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (i * 2654435761u) >> 20;
	__syncthreads();
	uint32_t x = 1;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) x = (uint32_t) __builtin_amdgcn_readfirstlane((int) tab[x & 4095]);
	unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[blockIdx.x] = x; cycles[blockIdx.x] = t1 - t0; }
}
__global__ void k_salu_chain(uint32_t *out, int iters, unsigned long long *cycles) {
	uint32_t x = (uint32_t) __builtin_amdgcn_readfirstlane((int) blockIdx.x + 12345);
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) { x = x * 1664525u + 1013904223u; x ^= x >> 13; x = (x << 5) | (x >> 27); x += i; }
	unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[blockIdx.x] = x; cycles[blockIdx.x] = t1 - t0; }
}
__global__ void k_lds_vec_chain(uint32_t *out, int iters, unsigned long long *cycles) {
	__shared__ uint32_t tab[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (i * 2654435761u) >> 20;
	__syncthreads();
	uint32_t x = threadIdx.x == 0 ? 1 : 1;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) x = tab[x & 4095];
	unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[blockIdx.x] = x; cycles[blockIdx.x] = t1 - t0; }
}
__global__ void k_global_chain(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cycles) {
	uint32_t x = 1;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) x = tab[x & 4095];
	unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[blockIdx.x] = x; cycles[blockIdx.x] = t1 - t0; }
}
template <typename F> void run(const char *name, F launch, int iters, unsigned long long *dcyc) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	launch(); hipDeviceSynchronize();
	hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	unsigned long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
	printf("%-22s %8.1f ns/iter  %8.1f counter-ticks/iter  (%.3f ms)\n", name, ms * 1e6 / iters, (double) c / iters, ms);
}
int main() {
	uint32_t *out; unsigned long long *cyc; uint32_t *gtab;
	hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 4096 * 8); hipMalloc(&gtab, 4096 * 4);
	std::vector<uint32_t> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 2654435761u) >> 20;
	hipMemcpy(gtab, h.data(), 4096 * 4, hipMemcpyHostToDevice);
	const int iters = 1000000;
	for (int blocks : {1, 512}) {
		printf("blocks = %d (64 threads each)\n", blocks);
		run("lds + readfirstlane", [&] { hipLaunchKernelGGL(k_lds_chain, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); }, iters, cyc);
		run("lds vector chain", [&] { hipLaunchKernelGGL(k_lds_vec_chain, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); }, iters, cyc);
		run("salu chain (7 ops)", [&] { hipLaunchKernelGGL(k_salu_chain, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); }, iters, cyc);
		run("global(L1) chain", [&] { hipLaunchKernelGGL(k_global_chain, dim3(blocks), dim3(64), 0, 0, gtab, out, iters, cyc); }, iters, cyc);
	}
	return 0;
}
